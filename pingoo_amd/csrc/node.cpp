// node.cpp — one host process driving every GPU of a node (include/pwaf.h: pwaf_node_*).
//
// north_star's host is ONE process (pingoo/server.rs:40-47,76: rules, lists and GeoIP are built once and shared read-only by
// every listener); requests are independent, so a batch is cut into contiguous 64-aligned slabs, one per device, each evaluated by
// that device's engine replica on its own host thread and stream. There is no data-path exchange between devices: the only
// cross-device result is the sum of the four action counters, added up on the host (the process-per-GPU mode of bench.py does the
// same with an RCCL all-reduce). Verdicts land directly in the caller's output array at the slab's position.
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <memory>
#include <vector>

#include "../../include/pwaf.h"

namespace pwaf {
int fail(int code, const std::string &msg);  // engine.cpp: sets the thread's last-error text
}

// One persistent host thread per device (round 3; round 2 spawned std::threads per call): a call posts one job per device and waits
// for all of them. A job runs on the thread that owns the device for the node's lifetime, so the HIP runtime's per-thread
// current-device state is set once.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, stop = false, done = true;
    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_job || stop; });
            if (stop && !has_job) return;
            std::function<void()> j = std::move(job);
            has_job = false;
            lk.unlock();
            j();
            lk.lock();
            done = true;
            cv.notify_all();
        }
    }
    void post(std::function<void()> j) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        job = std::move(j);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
    }
};

struct pwaf_node {
    std::vector<pwaf_engine *> engines;
    std::vector<int> devices;
    std::vector<std::unique_ptr<Worker>> workers;  // one per device but the first (the caller's thread serves device 0)
    std::mutex call_mu;                            // one node-level call at a time posts to the workers
    ~pwaf_node() {
        for (auto &w : workers) {
            {
                std::lock_guard<std::mutex> lk(w->mu);
                w->stop = true;
            }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
        }
    }
    // runs work(r) for every device r: r = 0 on the calling thread, the others on their persistent threads
    void for_each_device(const std::function<void(uint32_t)> &work) {
        std::lock_guard<std::mutex> lk(call_mu);
        const uint32_t world = (uint32_t)engines.size();
        for (uint32_t r = 1; r < world; r++) workers[r - 1]->post([&work, r] { work(r); });
        work(0);
        for (uint32_t r = 1; r < world; r++) workers[r - 1]->wait();
    }
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
    bool partial = false;  // PWAF_OPT_LENIENT dropped a rule (PWAF_W_PARTIAL, a positive status: the engine exists)
    for (size_t k = 0; k < n_devices; k++) {
        o.device = devices[k];
        pwaf_engine *e = nullptr;
        int rc = pwaf_engine_create(rules, n_rules, lists, n_lists, geoip, &o, &e, err);  // tables are replicated: tens of MB per device
        if (rc < 0) {
            if (e) pwaf_engine_destroy(e);
            pwaf_node_destroy(nd);
            return rc;
        }
        partial = partial || rc == PWAF_W_PARTIAL;
        nd->engines.push_back(e);
        nd->devices.push_back(devices[k]);
    }
    for (size_t k = 1; k < n_devices; k++) {
        nd->workers.emplace_back(new Worker());
        Worker *w = nd->workers.back().get();
        w->th = std::thread([w] { w->run(); });
    }
    *out = nd;
    return partial ? PWAF_W_PARTIAL : PWAF_OK;
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
    nd->for_each_device(work);
    for (uint32_t r = 0; r < world; r++)
        if (rcs[r]) return pwaf::fail(rcs[r], "device " + std::to_string(nd->devices[r]) + ": " + msgs[r]);
    if (counts)
        for (uint32_t r = 0; r < world; r++)
            for (int a = 0; a < 4; a++) counts->by_action[a] += part[r].by_action[a];
    return PWAF_OK;
}

// Device-resident multi-GPU entry point (round 3): every device already holds its slab of the request stream (pwaf_node_shard_bounds
// says which) — no host staging, no copy: slab r is enqueued on device r's engine replica and stream by that device's persistent
// thread (an enqueue costs ~0.2 ms of host time: in parallel, not one after the other). Asynchronous like pwaf_evaluate_device:
// verdicts and per-device counters are valid once the caller has synchronised the streams (or calls pwaf_node_synchronize).
int pwaf_node_evaluate_device(pwaf_node *nd, const pwaf_batch *const *batches, pwaf_verdict *const *outs, pwaf_counts *const *counts, void *const *streams) {
    if (!nd || !batches || !outs) return pwaf::fail(PWAF_E_INVALID_ARG, "NULL argument");
    const uint32_t world = (uint32_t)nd->engines.size();
    for (uint32_t r = 0; r < world; r++) {
        if (!batches[r]) return pwaf::fail(PWAF_E_INVALID_ARG, "batches[" + std::to_string(r) + "] is NULL (pass a batch with n = 0 for an idle device)");
        if (batches[r]->struct_size != sizeof(pwaf_batch)) return pwaf::fail(PWAF_E_INVALID_ARG, "pwaf_batch.struct_size mismatch");
        if (batches[r]->memory != PWAF_MEM_DEVICE) return pwaf::fail(PWAF_E_INVALID_ARG, "pwaf_node_evaluate_device takes DEVICE-resident slabs (host batches: pwaf_node_evaluate_batch)");
        if (batches[r]->n && !outs[r]) return pwaf::fail(PWAF_E_INVALID_ARG, "outs[" + std::to_string(r) + "] is NULL");
    }
    std::vector<int> rcs(world, PWAF_OK);
    std::vector<std::string> msgs(world);
    nd->for_each_device([&](uint32_t r) {
        if (batches[r]->n == 0) return;
        rcs[r] = pwaf_evaluate_device(nd->engines[r], batches[r], outs[r], counts ? counts[r] : nullptr, nullptr, nullptr, streams ? streams[r] : nullptr);
        if (rcs[r]) msgs[r] = pwaf_last_error();
    });
    for (uint32_t r = 0; r < world; r++)
        if (rcs[r]) return pwaf::fail(rcs[r], "device " + std::to_string(nd->devices[r]) + ": " + msgs[r]);
    return PWAF_OK;
}

// Waits for every device of the node and reports (like pwaf_engine_device_status per device) whether some device-resident batch
// since the last call ran out of scan scratch.
int pwaf_node_synchronize(pwaf_node *nd) {
    if (!nd) return pwaf::fail(PWAF_E_INVALID_ARG, "node is NULL");
    const uint32_t world = (uint32_t)nd->engines.size();
    std::vector<int> rcs(world, PWAF_OK);
    std::vector<std::string> msgs(world);
    nd->for_each_device([&](uint32_t r) {
        rcs[r] = pwaf_engine_device_status(nd->engines[r]);
        if (rcs[r]) msgs[r] = pwaf_last_error();
    });
    for (uint32_t r = 0; r < world; r++)
        if (rcs[r]) return pwaf::fail(rcs[r], "device " + std::to_string(nd->devices[r]) + ": " + msgs[r]);
    return PWAF_OK;
}

// The path's only exchange between devices, for a single-process host: the four action counters (32 bytes per device) summed over
// xGMI by RCCL — `comms[r]` is the caller's ncclComm_t of device r (one communicator spanning the node's devices, e.g. from
// ncclCommInitAll), the all-reduce is enqueued in place on counts[r] / streams[r] inside one RCCL group. librccl is loaded at run time
// (it is only needed by callers that pass communicators: hosts that sum the counters themselves never touch it).
int pwaf_node_allreduce_counts(pwaf_node *nd, void *const *comms, pwaf_counts *const *counts, void *const *streams) {
    if (!nd || !comms || !counts) return pwaf::fail(PWAF_E_INVALID_ARG, "NULL argument");
    const uint32_t world = (uint32_t)nd->engines.size();
    for (uint32_t r = 0; r < world; r++)
        if (!comms[r] || !counts[r]) return pwaf::fail(PWAF_E_INVALID_ARG, "comms / counts must hold one entry per device");
    typedef int (*group_fn)(void);
    typedef int (*allreduce_fn)(const void *, void *, size_t, int, int, void *, void *);
    static void *lib = nullptr;
    static group_fn g_start = nullptr, g_end = nullptr;
    static allreduce_fn allreduce = nullptr;
    static std::mutex load_mu;
    {
        std::lock_guard<std::mutex> lk(load_mu);
        if (!lib) {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (lib) break;
            }
            if (!lib) return pwaf::fail(PWAF_E_UNSUPPORTED, "librccl.so could not be loaded");
            g_start = (group_fn)dlsym(lib, "ncclGroupStart");
            g_end = (group_fn)dlsym(lib, "ncclGroupEnd");
            allreduce = (allreduce_fn)dlsym(lib, "ncclAllReduce");
        }
        if (!g_start || !g_end || !allreduce) return pwaf::fail(PWAF_E_UNSUPPORTED, "librccl.so lacks ncclGroupStart / ncclGroupEnd / ncclAllReduce");
    }
    constexpr int kNcclUint64 = 5, kNcclSum = 0;  // rccl.h: ncclDataType_t / ncclRedOp_t
    int rc = g_start();
    for (uint32_t r = 0; r < world && rc == 0; r++) rc = allreduce(counts[r], counts[r], 4, kNcclUint64, kNcclSum, comms[r], streams ? streams[r] : nullptr);
    const int rc2 = g_end();
    if (rc || rc2) return pwaf::fail(PWAF_E_DEVICE, "RCCL all-reduce of the action counters failed (ncclResult " + std::to_string(rc ? rc : rc2) + ")");
    return PWAF_OK;
}

}  // extern "C"
