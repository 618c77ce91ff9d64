// node.cpp — one host process driving every GPU of a node (include/pwaf.h: pwaf_node_*).
//
// north_star's host is ONE process (pingoo/server.rs:40-47,76: rules, lists and GeoIP are built once and shared read-only by
// every listener); requests are independent, so a batch is cut into contiguous 64-aligned slabs, one per device, each evaluated by
// that device's engine replica on its own host thread and stream. There is no data-path exchange between devices: the only
// cross-device result is the sum of the four action counters, added up on the host (the process-per-GPU mode of bench.py does the
// same with an RCCL all-reduce). Verdicts land directly in the caller's output array at the slab's position.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pwaf.h"

namespace pwaf {
int fail(int code, const std::string &msg);  // engine.cpp: sets the thread's last-error text
}

struct pwaf_node {
    std::vector<pwaf_engine *> engines;
    std::vector<int> devices;
};

extern "C" {

void pwaf_node_shard_bounds(uint32_t n, uint32_t rank, uint32_t world, uint32_t *lo, uint32_t *hi) {
    // contiguous slabs, balanced to within one 64-request group, aligned to 64 so that no bit-column group straddles two devices
    const uint64_t groups = ((uint64_t)n + 63) / 64;
    const uint64_t lo_g = world ? groups * rank / world : 0, hi_g = world ? groups * (rank + 1) / world : 0;
    *lo = (uint32_t)std::min<uint64_t>(n, lo_g * 64);
    *hi = (uint32_t)std::min<uint64_t>(n, hi_g * 64);
}

int pwaf_node_create(const pwaf_rule_desc *rules, size_t n_rules, const pwaf_list_desc *lists, size_t n_lists, const pwaf_geoip_table *geoip, const pwaf_options *opts,
                     const int *devices, size_t n_devices, pwaf_node **out, pwaf_compile_error *err) {
    if (!out || !devices || n_devices == 0) return pwaf::fail(PWAF_E_INVALID_ARG, "pwaf_node_create needs at least one device");
    pwaf_options o;
    memset(&o, 0, sizeof o);
    o.struct_size = sizeof o;
    if (opts) {
        if (opts->struct_size != sizeof(pwaf_options)) return pwaf::fail(PWAF_E_INVALID_ARG, "pwaf_options.struct_size mismatch");
        o = *opts;
    }
    pwaf_node *nd = new pwaf_node();
    for (size_t k = 0; k < n_devices; k++) {
        o.device = devices[k];
        pwaf_engine *e = nullptr;
        int rc = pwaf_engine_create(rules, n_rules, lists, n_lists, geoip, &o, &e, err);  // tables are replicated: tens of MB per device
        if (rc) {
            pwaf_node_destroy(nd);
            return rc;
        }
        nd->engines.push_back(e);
        nd->devices.push_back(devices[k]);
    }
    *out = nd;
    return PWAF_OK;
}

void pwaf_node_destroy(pwaf_node *nd) {
    if (!nd) return;
    for (pwaf_engine *e : nd->engines) pwaf_engine_destroy(e);
    delete nd;
}

size_t pwaf_node_device_count(const pwaf_node *nd) { return nd ? nd->engines.size() : 0; }
pwaf_engine *pwaf_node_engine(const pwaf_node *nd, size_t i) { return (nd && i < nd->engines.size()) ? nd->engines[i] : nullptr; }

int pwaf_node_tune(pwaf_node *nd, const pwaf_batch *sample) {
    if (!nd) return pwaf::fail(PWAF_E_INVALID_ARG, "node is NULL");
    for (pwaf_engine *e : nd->engines) {
        int rc = pwaf_engine_tune(e, sample);
        if (rc) return rc;
    }
    return PWAF_OK;
}

int pwaf_node_evaluate_batch(pwaf_node *nd, const pwaf_batch *in, pwaf_verdict *out, pwaf_counts *counts) {
    if (!nd || !in || !out) return pwaf::fail(PWAF_E_INVALID_ARG, "NULL argument");
    if (in->struct_size != sizeof(pwaf_batch)) return pwaf::fail(PWAF_E_INVALID_ARG, "pwaf_batch.struct_size mismatch");
    if (in->memory != PWAF_MEM_HOST) return pwaf::fail(PWAF_E_INVALID_ARG, "pwaf_node_evaluate_batch shards HOST batches (device-resident batches already live on one device)");
    const uint32_t world = (uint32_t)nd->engines.size(), n = in->n;
    if (counts) memset(counts, 0, sizeof *counts);
    if (n == 0) return PWAF_OK;
    std::vector<int> rcs(world, PWAF_OK);
    std::vector<std::string> msgs(world);
    std::vector<pwaf_counts> part(world);
    auto work = [&](uint32_t r) {
        uint32_t lo, hi;
        pwaf_node_shard_bounds(n, r, world, &lo, &hi);
        memset(&part[r], 0, sizeof(pwaf_counts));
        if (hi <= lo) return;
        // a view of the slab: offsets and fixed-width columns advanced to request `lo`; arenas are shared (offsets stay absolute, the
        // engine copies only the slab's own bytes)
        pwaf_batch sub = *in;
        sub.n = hi - lo;
        for (int f = 0; f < PWAF_N_FIELDS; f++) sub.field[f].offsets = in->field[f].offsets + lo;
        sub.ip = in->ip + (size_t)lo * 16;
        sub.ip_is_v6 = in->ip_is_v6 + lo;
        sub.port = in->port + lo;
        sub.flags = in->flags + lo;
        if (in->asn) sub.asn = in->asn + lo;
        if (in->country) sub.country = in->country + lo;
        std::vector<pwaf_strcol> hdr;
        if (in->headers && in->n_headers) {
            hdr.assign(in->headers, in->headers + in->n_headers);
            for (auto &c : hdr)
                if (c.offsets) c.offsets += lo;
            sub.headers = hdr.data();
        }
        rcs[r] = pwaf_evaluate_batch(nd->engines[r], &sub, out + lo, &part[r]);
        if (rcs[r]) msgs[r] = pwaf_last_error();  // (the error text is thread-local: carry it to the caller's thread)
    };
    std::vector<std::thread> th;
    for (uint32_t r = 1; r < world; r++) th.emplace_back(work, r);
    work(0);
    for (auto &t : th) t.join();
    for (uint32_t r = 0; r < world; r++)
        if (rcs[r]) return pwaf::fail(rcs[r], "device " + std::to_string(nd->devices[r]) + ": " + msgs[r]);
    if (counts)
        for (uint32_t r = 0; r < world; r++)
            for (int a = 0; a < 4; a++) counts->by_action[a] += part[r].by_action[a];
    return PWAF_OK;
}

}  // extern "C"
