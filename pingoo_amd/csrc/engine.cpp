// engine.cpp — the C ABI of libpwaf.so (include/pwaf.h): engine lifetime, device tables, batch
// marshalling and the launch sequence. Construction mirrors where the reference builds its
// read-only rule state once (pingoo/server.rs:40-47,76) and hands it to every listener
// (server.rs:111-134); evaluation replaces the inline loop at http_listener.rs:196-264.
//
// There is NO CPU evaluation path in this library: without a working HIP device every evaluate call
// returns PWAF_E_DEVICE (the host may then fail open, as the reference does on rule errors —
// pingoo/rules.rs:41-45).
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "frontend.h"
#include "kernels.h"
#include "program.h"

using namespace pwaf;

struct pwaf_program {
    std::unique_ptr<Program> p;
    std::vector<uint8_t> dump;  // lazily built
};

namespace {
thread_local std::string g_last_error = "";
}
namespace pwaf {
int fail(int code, const std::string &msg) {  // also used by loaders.cpp
    g_last_error = msg;
    return code;
}
}  // namespace pwaf
using pwaf::fail;
namespace {
#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess) return fail(PWAF_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return PWAF_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(PWAF_E_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
        cap = want;
        return PWAF_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Page-locked host memory the device reads / writes directly: staging of small host batches, status words on their way back.
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return PWAF_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocPortable);
        if (e != hipSuccess) return fail(PWAF_E_NOMEM, std::string("hipHostMalloc failed: ") + hipGetErrorString(e));
        cap = want;
        return PWAF_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <class T>
int upload(DevBuf &b, const std::vector<T> &v, size_t pad_bytes = 0) {
    size_t bytes = v.size() * sizeof(T);
    int rc = b.reserve(bytes + pad_bytes + 16);
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpy(b.p, v.data(), bytes, hipMemcpyHostToDevice));
    if (pad_bytes + 16) HIP_TRY(hipMemset((char *)b.p + bytes, 0, pad_bytes + 16));
    return PWAF_OK;
}

// flat form of a DFA for list-driven walks (lscan_kernel): next state | 0x8000 when entering it emits; lists indexed by state
struct FlatDev {
    DevBuf flat, flat_classmap, emit_off, emit_list, end_off, end_list, delta;
    uint32_t n_states = 0, n_classes = 0;
    bool scalar_mode = false;  // the table reads scalar values (dfa.cpp): the class image holds the scalar map behind the byte map
    uint32_t ill_class = 0;
    uint32_t n_full = 0, n_delta = 0;  // LDS layout of the list scan: rows [0, n_full), then n_delta 8-byte delta records (states n_full ..)
    void release() {
        for (DevBuf *b : {&flat, &flat_classmap, &emit_off, &emit_list, &end_off, &end_list, &delta}) b->release();
        n_states = 0;
    }
};

struct DevGroup {
    DevBuf tab, classmap, special, list_off, list;
    uint32_t n_states, stride, n_classes, n_hot, start_emit, emit_base, special_base, atom_base, n_local;
    bool scalar_mode = false;
    uint32_t ill_class = 0;
    uint8_t field;
    uint32_t chunks = 1;  // 16-byte chunks per scan iteration (2 for fields whose sampled mean length is >= 48 bytes)
    int gate = -1;  // >= 0: list-driven pass (behind a bigram prefilter, or gated by prefilter factors): index of its request list
    bool filtered = false;  // the list comes from filter_kernel + compact_kernel
    int share_owner = -1;   // a gap pass whose factors all belong to this filtered pass: it walks the owner's list (no list of its own)
    uint32_t shared_bits = 0;  // owner: list bits of the gap passes sharing its list
    int need_slot = -1;     // owner: which need-mask array
    int visit_slot = -1;    // a gap pass: which visited bitmap (its records are valid only where it walked)
    bool identity = false;  // a plain pass over a SHORT field (`method`): walked by the list-scan kernel with the identity list
    GroupFilter filter;     // the prefilter in use (Program's, or rebuilt from a traffic sample by pwaf_engine_tune)
    DevBuf ftable;
    // flat form of the DFA for list-driven walks (lscan_kernel): of the pass's every atom (fl), and — a pass with a confirm tier and
    // atoms of both kinds — of its non-literal atoms alone (rt: DfaGroup::rtier), which is what confirmed candidates walk
    FlatDev fl, rt;
    bool short_lit = false;  // every atom is an anchored literal of <= 8 bytes: evaluated by the attribute kernel, the pass is never walked
    // confirm tier in use (assign_lists uploads it with the filter table): ConfirmTable::head / entries / bytes / classes
    DevBuf c_head, c_entries, c_bytes, c_classes;
    bool confirm = false;      // the pass's candidates go through confirm_kernel
    bool confirm_walk = false; // ... and those with a confirmed regex factor through the DFA (rt if built, else fl)
};

}  // namespace

// Everything one in-flight batch needs besides the immutable tables: device scratch, host-batch staging buffers, its own stream for
// the synchronous entry points, the side stream of the attribute kernel and the events that order them. An engine owns a small
// ring of these, so that callers on several threads / streams overlap (one batch's copies under another's kernels) instead of
// queueing behind one set of buffers; a context's `done` event makes its next user wait (on the device) for the previous one.
struct Scratch {
    std::mutex mu;  // held while a call enqueues on this context (device calls) or for the whole call (host batches: staging is reused)
    hipStream_t stream = nullptr, side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, done = nullptr;
    bool used = false;
    hipStream_t last = nullptr;  // the stream of the context's latest batch
    bool last_own = false;       // ... which was the context's own stream (a host batch)
    DevBuf status;            // one sticky word: bit 0 = some batch on this context exhausted the overflow pool (cleared when reported)
    uint64_t pool_entries = 0;  // overflow pool size in use (grown when a batch exhausted it)
    struct View { void *p = nullptr; };  // a part of zero_block / args (not owned)
    View res_cols;  // residual_kernel: the batch's string columns as device arrays of pointers
    bool retry = false;   // this run repeats a batch that exhausted the overflow pool: its residual rules' execution errors were counted by the first run
    DevBuf err_sink;      // ... and land here
    DevBuf ipres;  // ipres_kernel -> attr_kernel: (GeoIP class, membership set) of every request
    DevBuf rec, pool, gate_lists, attr;
    DevBuf verdict_spill;     // the sparse column file's per-wave spill arrays (kernels.h: VerdictArgs::spill)
    DevBuf res_words;  // specialized residual program: [words][n] match bits per (request, rule)
    // What a batch needs ZEROED lives in one block (one memset per batch instead of four): the control words ([0] pool allocator, [1]
    // status word, then one length per list slot and one pair count per filtered pass), the candidate bitmaps of the filtered passes,
    // the visited bitmaps of the gap passes and the walk bitmaps of the confirm tier.
    DevBuf zero_block;
    View ctrl, cand_bits, visit_bits, walk_bits;
    DevBuf chunk_bits, cand_cnt;  // filter_kernel's chunk bitmaps and per-slab counts
    DevBuf need;                           // per sharing owner: gap-pass mask of every entry of its candidate list
    DevBuf pairs;                          // per filtered pass with a confirm tier: resolve_kernel's (request, flagged chunk) pairs
    DevBuf zero_off;                       // n + 1 zero offsets: the column of a header the batch does not carry
    // Launch descriptors of a batch (kernels.h: FilterArgs / ConfirmArgs / ListScanArgs per pass, the residual kernel's column pointers):
    // built on the host before the first launch and uploaded ONCE — the host writes them into a page-locked slot and one copy launch
    // on the batch's stream moves them to `args` (stream-ordered like the by-value store launches it replaces: a 4096-rule set over 64
    // header fields took 22 of those per batch). A slot is rewritten only after the copy launch that read it has run (its event);
    // with kArgSlots slots a caller can be that many batches ahead of the device before it has to wait.
    static constexpr uint32_t kArgSlots = 8;
    DevBuf args;
    PinBuf arg_slot[kArgSlots];
    hipEvent_t arg_ev[kArgSlots] = {};
    bool arg_pending[kArgSlots] = {};
    uint32_t arg_next = 0;
    std::vector<DevBuf> stage_field_data, stage_field_off;  // n_fields each
    DevBuf stage_ip, stage_v6, stage_port, stage_flags, stage_asn, stage_country, stage_out, stage_counts;
    // A SMALL host batch (the micro-batcher's, pwaf_evaluate_one's) travels as ONE block: every column packed into page-locked memory,
    // one asynchronous copy in, one out (pwaf_evaluate_batch). pin_status: the status words on their way back (never a pageable target:
    // a device-to-host copy into pageable memory is a blocking staged copy).
    PinBuf pin_in, pin_out, pin_status;
    DevBuf packed;
    // Streams and events are created on first use: every HIP stream takes a share of the few hardware queues of its priority
    // class, and a context (or its own stream) a caller never uses must not cost the caller's streams their concurrency (measured:
    // three idle contexts' streams made two caller streams share one queue — no overlap between two batches in flight).
    // The side stream (attribute kernel) is created at the LOWEST stream priority: priority classes have hardware queues of their
    // own, so it can never share a queue with the caller's (normal-priority) stream — when it did, the attribute kernel ran in
    // front of the prefilter instead of beside it (measured: +0.45 ms per 10M-request batch).
    int ensure(bool own_stream) {
        if (!side) {
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
#ifdef PWAF_PROFILING
            if (getenv("PWAF_SIDE_PRIORITY_HIGH")) least = greatest;  // timing experiment
#endif
            if (hipStreamCreateWithPriority(&side, hipStreamNonBlocking, least) != hipSuccess || hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess)
                return fail(PWAF_E_DEVICE, "hipStreamCreate / hipEventCreate failed");
            for (hipEvent_t &ev : arg_ev)
                if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(PWAF_E_DEVICE, "hipEventCreate failed");
        }
        if (own_stream && !stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return fail(PWAF_E_DEVICE, "hipStreamCreate failed");
        return PWAF_OK;
    }
    void release() {
        for (DevBuf *b : {&status, &ipres, &rec, &res_words, &err_sink, &pool, &verdict_spill, &zero_block, &gate_lists, &attr, &chunk_bits, &cand_cnt, &need, &pairs, &zero_off, &args, &stage_ip, &stage_v6, &stage_port, &stage_flags,
                          &stage_asn, &stage_country, &stage_out, &stage_counts})
            b->release();
        for (PinBuf &b : arg_slot) b.release();
        for (hipEvent_t ev : arg_ev)
            if (ev) (void)hipEventDestroy(ev);
        for (auto &b : stage_field_data) b.release();
        for (auto &b : stage_field_off) b.release();
        packed.release();
        for (PinBuf *b : {&pin_in, &pin_out, &pin_status}) b->release();
        if (stream) (void)hipStreamDestroy(stream);
        if (side) (void)hipStreamDestroy(side);
        for (hipEvent_t ev : {ev_fork, ev_join, done})
            if (ev) (void)hipEventDestroy(ev);
    }
};
static constexpr size_t kContexts = 3;

struct pwaf_engine {
    pwaf_program prog;
    int device = 0;
    std::vector<std::unique_ptr<Scratch>> ctx;  // kContexts
    size_t next_ctx = 0;
    std::vector<DevGroup> groups;
    DevBuf num_atoms, bit_atoms /* (source word, bit) -> column */, trig_off, trig_rules, always_rules, country_luts /* transposed: [676][cc_words] */, rules, lits, set_masks;
    uint32_t cmp_vars = 0;  // (VerdictArgs::cmp_vars)
    uint32_t cc_words = 1, n_cmp_atoms = 0, n_bit_atoms = 0, n_trig = 0, n_lazy = 0;
    DevBuf lazy_atoms;  // (VerdictArgs::lazy)
    std::vector<uint32_t> lazy_vars;  // (VerdictArgs::lazy_var)
    uint32_t n_short = 0;  // short-literal atoms (kernels.h: ShortAtom) of the one field handled that way
    int short_field = -1;
    DevBuf short_atoms;
    uint32_t class_words = 1, acmp_words = 0, geo_default = 0, n_classes = 1;
    DevBuf class_rows, dir_esc, leaf_root, geo_leaf_root;
    std::vector<uint32_t> host_cc_masks, host_iu_masks1;  // kept for building the per-record rows
    std::vector<std::pair<uint32_t, uint32_t>> host_acmp;  // (operator 0: ==, 1: <=; constant) of the client.asn comparisons
    std::vector<int64_t> host_iu_vals1;
    DevBuf iu_vals[2], iu_masks[2];
    uint32_t iu_n[2] = {0, 0}, iu_words[2] = {1, 1};
    DevBuf ip_root4, ip_root6, ip_nodes, geo_root4, geo_root6, geo_nodes, geo_recs;
    std::mutex mu;  // guards the context ring and table rebuilds (pwaf_engine_tune)
    std::mutex prof_mu;  // while profiling is on, calls enqueue one at a time: the event / timing tables below are per engine
    DevBuf residual_errors;  // per residual rule: requests whose evaluation ended in an execution error (accumulated; pwaf_engine_rule_errors)
    std::string residual_note;  // why the residual rules are interpreted although specialization was asked for (pwaf_engine_residual_fallback)
    JitKernel residual_jit;  // the specialized residual program (residual_jit.cpp + rtc.cpp); function == nullptr: the rules are interpreted
    DevBuf residual_blob, geo_rec_root4, geo_rec_root6, geo_rec_nodes;  // residual rules: the program image; the GeoIP trie with RECORD leaves (client.asn / country values)
    DevBuf pass_base, colmask, dir24 /* build-time only: released once compressed */, dir_chunks, dir_vals, dir_summary;
    uint32_t dir_sum_shift = 0, dir_common = 0;  // (VerdictArgs::dir_summary)
    uint32_t n_need = 0;   // sharing owners (need-mask arrays per batch)
    uint32_t n_visit = 0;  // gap passes (visited bitmaps per batch)
    std::vector<double> mean_len;  // per field, from the tuning sample (0 = unknown)
    uint32_t n_ungated = 0, n_gated = 0, n_filtered = 0;
    std::vector<uint8_t> owns_factors;  // per pass: some of its atoms are prefilter factors of gap passes
    uint32_t n_gap = 0;                 // gated gap passes (list slots [0, kGapLists), one factor-mask bit each)
    DevBuf pass_table;                  // PassInfo per pass
    std::vector<uint32_t> hlen_fields;  // header columns whose length some rule compares (comparison variable 7 + k)
    uint32_t n_fields = PWAF_N_FIELDS;  // 5 + header columns
    // profiling
    int profiling = 0;  // 0 off, 1 every kernel, 2 only the launches that stream request bytes (pwaf_engine_set_profiling)
    std::vector<hipEvent_t> ev;
    std::vector<pwaf_kernel_time> times;
    std::vector<std::pair<size_t, size_t>> time_ev;  // (begin, end) event index of each entry of `times`
    size_t n_timed = 0;
    uint32_t n_cus = 256;
};

namespace {

int check_opts(const pwaf_options *o, pwaf_options &out) {
    memset(&out, 0, sizeof out);
    out.struct_size = sizeof out;
    out.device = -1;
    if (!o) return PWAF_OK;
    if (o->struct_size != sizeof(pwaf_options)) return fail(PWAF_E_INVALID_ARG, "pwaf_options.struct_size mismatch");
    out = *o;
    return PWAF_OK;
}

// which verdict kernel (kernels.h: verdict_shape's mode): the entry list unless an A/B flag asks for an earlier column file
uint32_t verdict_mode(uint32_t flags) {
    const bool tiny = (flags & PWAF_OPT_TINY_VERDICT_SLOTS) != 0;
    if (flags & PWAF_OPT_DENSE_VERDICT) return 0u;
    if (flags & PWAF_OPT_SPARSE_VERDICT) return tiny ? 2u : 1u;
    return tiny ? 4u : 3u;
}

void put_err(pwaf_compile_error *dst, const pwaf_compile_error &src) {
    if (dst) *dst = src;
    g_last_error = src.message;
}

// A table's class lookups on the device: the 256-byte class map of the BYTES, 16 bytes of padding, then — scalar mode — the image of the
// scalar-value map (csrc/utf8.h; classes renumbered like the byte map when the table permutes its columns).
static constexpr size_t kUmapAt = 272;
std::vector<uint8_t> class_image(const DfaGroup &g, const std::vector<uint32_t> *cpos, uint32_t &ill_class) {
    std::vector<uint8_t> img(kUmapAt, 0);
    for (int b = 0; b < 256; b++) img[(size_t)b] = (uint8_t)(cpos ? (*cpos)[g.classmap[b]] : g.classmap[b]);
    ill_class = 0;
    if (g.umap.on()) {
        ScalarMap m = g.umap;
        if (cpos) {
            for (auto &c : m.stage2) c = (uint8_t)(*cpos)[c];
            m.ill_class = (uint8_t)(*cpos)[m.ill_class];
        }
        ill_class = m.ill_class;
        const std::vector<uint8_t> u = scalar_map_image(m);
        img.insert(img.end(), u.begin(), u.end());
    }
    return img;
}

// Builds the device form of one DFA group: see the cell encoding in kernels.h. `visits` (optional, one count per state) is a
// traffic profile from pwaf_engine_tune: the LDS-resident ("hot") rows are then the most visited states instead of the
// shallowest ones. The result of a scan never depends on which rows are hot.
int build_device_group(const DfaGroup &g, uint32_t lds_hot_budget, DevGroup &d, const std::vector<uint64_t> *visits = nullptr,
                       const std::vector<uint64_t> *class_freq = nullptr) {
    const uint32_t C = g.n_classes, stride = C + 3, stride2 = stride * 2;
    // Class numbering = cell position inside a row. An LDS lookup conflicts when two lanes hit different dwords of one bank
    // (32 banks x 4 B = 128 B: DESIGN.md §6); lanes that sit in the SAME row — the usual case, most walks hover around the
    // start state — conflict exactly when their classes are a multiple of 64 cells apart. With more than 64 classes some
    // positions have no alias inside the row: given a traffic profile, the most frequent classes get those.
    std::vector<uint32_t> cpos(C);
    for (uint32_t c = 0; c < C; c++) cpos[c] = c;
    if (class_freq && class_freq->size() >= C && C > 64) {
        std::vector<uint32_t> by_freq(C), slots;
        for (uint32_t c = 0; c < C; c++) by_freq[c] = c;
        std::stable_sort(by_freq.begin(), by_freq.end(), [&](uint32_t x, uint32_t y) { return (*class_freq)[x] > (*class_freq)[y]; });
        // positions ordered by how many positions of the row share their bank (p mod 64): alone first, then pairs, then triples
        for (uint32_t share = 1; share <= 5; share++)
            for (uint32_t p = 0; p < C; p++) {
                const uint32_t r = p % 64, n_share = (C - 1 - r) / 64 + 1;
                if (n_share == share) slots.push_back(p);
            }
        for (uint32_t k = 0; k < C; k++) cpos[by_freq[k]] = slots[k];
    }
    if (g.n_states > kMaxDfaStates) return fail(PWAF_E_UNSUPPORTED, "DFA has more than 32767 states");
    std::vector<uint32_t> list_off{0};
    std::vector<uint16_t> list;
    auto add_list = [&](const std::vector<uint16_t> &src, uint32_t b, uint32_t e) -> uint32_t {
        list.insert(list.end(), src.begin() + b, src.begin() + e);
        list_off.push_back((uint32_t)list.size());
        return (uint32_t)list_off.size() - 1;  // 1 + id
    };
    std::vector<uint32_t> emit_id(g.n_states, 0);
    for (uint32_t s = 0; s < g.n_states; s++)
        if (g.emit_off[s + 1] > g.emit_off[s]) emit_id[s] = add_list(g.emit_list, g.emit_off[s], g.emit_off[s + 1]);
    // what entering the start state emits is recorded when a request starts (start_emit); re-entering it adds nothing
    const uint32_t start_emit = emit_id[0];
    emit_id[0] = 0;

    // Row order: the start state, then by visit count (profile) or BFS depth (the DFA builder's state order).
    std::vector<uint32_t> order(g.n_states);
    for (uint32_t s = 0; s < g.n_states; s++) order[s] = s;
    if (visits && visits->size() == g.n_states)
        std::stable_sort(order.begin() + 1, order.end(), [&](uint32_t x, uint32_t y) { return (*visits)[x] > (*visits)[y]; });

    // Cell value space (uint16): [0, emit_base) plain hot rows, [emit_base, (n_hot+1)*stride) hot rows that emit + the sentinel
    // row, the rest indexes `special` (one entry per cold state). Shrink the hot set until everything fits.
    const uint32_t budget = std::min<uint32_t>(lds_hot_budget, 131070u);
    uint32_t n_hot = budget > 2 * stride2 ? (budget - stride2) / stride2 : 1;
    n_hot = std::max(1u, std::min(n_hot, g.n_states));
    for (;;) {
        const uint64_t need = (uint64_t)(n_hot + 1) * stride + (g.n_states - n_hot);
        if (need <= 65535) break;
        if (n_hot == 1) return fail(PWAF_E_UNSUPPORTED, "DFA too large for the 16-bit cell space");
        const uint32_t over = (uint32_t)(need - 65535);
        n_hot = std::max(1u, n_hot - std::max(1u, over / (stride - 1) + 1));  // an evicted row frees `stride` cells and adds one special
    }
    // hot rows: plain ones first, then the emitting ones (one compare against emit_base finds both "emits" and "special")
    std::stable_partition(order.begin() + 1, order.begin() + n_hot, [&](uint32_t s) { return emit_id[s] == 0; });
    uint32_t n_plain = n_hot;
    for (uint32_t q = 0; q < n_hot; q++)
        if (emit_id[order[q]]) { n_plain = q; break; }
    std::vector<uint32_t> pos(g.n_states);
    for (uint32_t q = 0; q < g.n_states; q++) pos[order[q]] = q;
    const uint32_t emit_base = n_plain * stride, special_base = (n_hot + 1) * stride;
    std::vector<SpecialCell> special;
    for (uint32_t q = n_hot; q < g.n_states; q++) special.push_back({q * stride2, emit_id[order[q]]});
    auto cell_of = [&](uint32_t t) -> uint16_t { return pos[t] < n_hot ? (uint16_t)(pos[t] * stride) : (uint16_t)(special_base + (pos[t] - n_hot)); };

    std::vector<uint16_t> tab((size_t)g.n_states * stride, 0);
    for (uint32_t q = 0; q < g.n_states; q++) {
        const uint32_t s = order[q];
        uint16_t *row = &tab[(size_t)q * stride];
        for (uint32_t c = 0; c < C; c++) row[cpos[c]] = cell_of(g.trans[(size_t)s * C + c]);
        row[C] = q < n_hot ? (uint16_t)(q * stride) : (uint16_t)0xFFFF;  // STAY
        if (g.end_off[s + 1] > g.end_off[s]) {
            // END cell: like the EMIT cell — 0x8000 | atom for a single match (whole-string equality, the usual case: no list
            // walk, no memory access when a request finishes), else 1 + list id
            if (g.end_off[s + 1] - g.end_off[s] == 1 && g.end_list[g.end_off[s]] < 0x8000u) {
                row[C + 1] = (uint16_t)(0x8000u | g.end_list[g.end_off[s]]);
            } else {
                const uint32_t id1 = add_list(g.end_list, g.end_off[s], g.end_off[s + 1]);
                if (id1 >= 0x8000u) return fail(PWAF_E_UNSUPPORTED, "too many match lists in one DFA group");
                row[C + 1] = (uint16_t)id1;
            }
        }
        if (emit_id[s]) {
            // EMIT cell of a hot row: 0x8000 | atom for a single match, else 1 + list id
            const uint32_t b = list_off[emit_id[s] - 1], en = list_off[emit_id[s]];
            if (en - b == 1 && list[b] < 0x8000u) row[C + 2] = (uint16_t)(0x8000u | list[b]);
            else if (emit_id[s] < 0x8000u) row[C + 2] = (uint16_t)emit_id[s];
            else return fail(PWAF_E_UNSUPPORTED, "too many match lists in one DFA group");
        }
    }
    if (special.empty()) special.push_back({0, 0});
    d.n_states = g.n_states;
    d.stride = stride;
    d.n_classes = C;
    d.n_hot = n_hot;
    d.start_emit = start_emit;
    d.emit_base = emit_base;
    d.special_base = special_base;
    d.atom_base = g.atom_base;
    d.n_local = g.n_local;
    d.field = g.field;
    int rc;
    if ((rc = upload(d.tab, tab, 16))) return rc;
    const std::vector<uint8_t> cm = class_image(g, &cpos, d.ill_class);
    d.scalar_mode = g.umap.on();
    if ((rc = upload(d.classmap, cm))) return rc;
    if ((rc = upload(d.special, special))) return rc;
    if ((rc = upload(d.list_off, list_off))) return rc;
    if ((rc = upload(d.list, list))) return rc;
    return PWAF_OK;
}

// The flat form of a group for lscan_kernel. States are renumbered — start state first, then by visits of the tuning sample
// (discovery order without one) — so that the rows lscan_kernel stages in LDS are the ones its walks spend their steps in.
int build_flat_group(const DfaGroup &g, FlatDev &d, bool wide, const std::vector<uint64_t> *visits = nullptr) {
    const uint32_t S = g.n_states, C = g.n_classes;
    d.n_states = S;
    d.n_classes = C;
    std::vector<uint32_t> order(S), pos(S);
    for (uint32_t s = 0; s < S; s++) order[s] = s;
    if (visits && visits->size() == S) std::stable_sort(order.begin() + 1, order.end(), [&](uint32_t x, uint32_t y) { return (*visits)[x] > (*visits)[y]; });
    // DELTA rows. A row that is not LDS-resident costs an L2 round trip per step, and with 64 walks in lockstep some lane is in
    // such a row in nearly every group of steps once a twentieth of the steps are (hostile traffic: near misses of the rule
    // literals, deep in the patterns' prefix chains). But such states are the cheap kind: a state deep inside one literal differs
    // from a shallow state — the one its failure transitions lead back to — in one or two cells. A state within TWO cells of one
    // of the hottest rows is therefore kept in LDS as an 8-byte record (base row, two exception cells) instead of a row of 100-150
    // bytes: measured on the 1k-rule set, all of the hostile stream's steps outside the resident rows are in such states. The rows /
    // records split maximises the sample visits covered (no sample: the states covered).
    const uint32_t row_bytes = 2u * (C + 3u), budget = list_hot_bytes(list_shape(wide ? 2u : 0u));
    const uint32_t cap_rows = std::min<uint32_t>(S, (budget - 48u) / row_bytes);
    d.n_full = cap_rows;
    d.n_delta = 0;
    std::vector<uint64_t> delta_rec;
    if (S > cap_rows && cap_rows >= 8 && C <= 255) {
        const uint32_t B = std::min<uint32_t>(cap_rows, 512u);  // candidate base rows: the hottest ones
        struct Near { uint16_t base; uint8_t n, c[2]; };
        std::vector<Near> near(S, Near{0, 255, {0, 0}});
        for (uint32_t q = B; q < S; q++) {
            const uint32_t s = order[q];
            const uint16_t *rs = &g.trans[(size_t)s * C];
            for (uint32_t b = 0; b < B && near[s].n != 0; b++) {
                const uint16_t *rb = &g.trans[(size_t)order[b] * C];
                uint32_t nd = 0;
                uint8_t cc[2] = {0, 0};
                for (uint32_t c = 0; c < C && nd <= 2; c++)
                    if (rs[c] != rb[c]) { if (nd < 2) cc[nd] = (uint8_t)c; nd++; }
                if (nd <= 2 && nd < near[s].n) near[s] = Near{(uint16_t)b, (uint8_t)nd, {cc[0], cc[1]}};
            }
        }
        auto weight = [&](uint32_t s) { return (visits && visits->size() == S ? (double)(*visits)[s] : 0.0) + 1e-3; };
        double best_score = -1;
        uint32_t best_n = cap_rows;
        for (uint32_t n = cap_rows;; n = n >= B + 16 ? n - 16 : B) {
            const uint64_t space = (uint64_t)budget - (uint64_t)n * row_bytes;
            uint64_t room = space > 48 ? (space - 48) / 8 : 0;
            double score = 0;
            for (uint32_t q = 0; q < n; q++) score += weight(order[q]);
            for (uint32_t q = n; q < S && room; q++)
                if (near[order[q]].n <= 2) { score += weight(order[q]); room--; }
            if (score > best_score) { best_score = score; best_n = n; }
            if (n == B) break;
        }
        // rows [0, n_full), then the records (in visit order, as many as fit), then everything else
        std::vector<uint32_t> full(order.begin(), order.begin() + best_n), recs, rest;
        uint64_t room = (uint64_t)budget - (uint64_t)best_n * row_bytes > 48 ? ((uint64_t)budget - (uint64_t)best_n * row_bytes - 48) / 8 : 0;
        for (uint32_t q = best_n; q < S; q++) {
            if (near[order[q]].n <= 2 && room) { recs.push_back(order[q]); room--; }
            else rest.push_back(order[q]);
        }
        d.n_full = best_n;
        d.n_delta = (uint32_t)recs.size();
        order = full;
        order.insert(order.end(), recs.begin(), recs.end());
        order.insert(order.end(), rest.begin(), rest.end());
        for (uint32_t q = 0; q < S; q++) pos[order[q]] = q;
        // (a class that STAYS — a continuation byte of scalar mode — enters nothing: its cell, the state itself, carries no emit flag)
        auto cell_of = [&](uint32_t t, uint32_t c) { return (uint16_t)(pos[t] | ((g.emit_off[(size_t)t + 1] != g.emit_off[t] && !(c < g.class_stays.size() && g.class_stays[c])) ? 0x8000u : 0u)); };
        for (uint32_t s : recs) {
            const Near &nr = near[s];
            const uint16_t *rs = &g.trans[(size_t)s * C];
            const uint8_t c1 = nr.n >= 1 ? nr.c[0] : 0, c2 = nr.n >= 2 ? nr.c[1] : c1;
            // (no exception: both slots repeat the base row's own cell of class 0)
            const uint16_t t1 = cell_of(rs[c1], c1), t2 = cell_of(rs[c2], c2);
            delta_rec.push_back((uint64_t)nr.base | ((uint64_t)c1 << 16) | ((uint64_t)c2 << 24) | ((uint64_t)t1 << 32) | ((uint64_t)t2 << 48));
        }
    }
    for (uint32_t q = 0; q < S; q++) pos[order[q]] = q;
    // row = C transition cells (next state | 0x8000 when entering it emits) + one EMIT cell: what entering THIS state emits —
    // 0 = nothing, 0x8000 | local atom = exactly one atom (the common case: settled in registers by the kernel), else 1 + the state's
    // index into emit_off (a list). The cell rides with the row into LDS: round 2 called the out-of-line list walk (three dependent
    // global loads, ~2 us for the whole wave) for every match of every lane — benign candidates are mostly true hits, so a wave of
    // 64 candidates stalled on the order of a hundred times per walk.
    // ... one STAY cell (= the state itself, unflagged): what a lane past its field's end "reads", so that no step is conditional,
    // and one END cell (1 = the field ending in this state emits): a finished walk learns it from the row instead of two global loads.
    const uint32_t stride = C + 3;
    std::vector<uint16_t> flat((size_t)S * stride);
    std::vector<uint32_t> emit_off(1, 0), end_off(1, 0);
    std::vector<uint16_t> emit_list, end_list;
    for (uint32_t q = 0; q < S; q++) {
        const uint32_t s = order[q];
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t t = g.trans[(size_t)s * C + c];
            const bool stays = c < g.class_stays.size() && g.class_stays[c];  // (a continuation byte of scalar mode: the state itself, entering nothing)
            flat[(size_t)q * stride + c] = (uint16_t)(pos[t] | ((g.emit_off[(size_t)t + 1] != g.emit_off[t] && !stays) ? 0x8000u : 0u));
        }
        const uint32_t ne = g.emit_off[(size_t)s + 1] - g.emit_off[s];
        flat[(size_t)q * stride + C] = ne == 0 ? (uint16_t)0 : (ne == 1 && g.emit_list[g.emit_off[s]] < 0x7FFFu) ? (uint16_t)(0x8000u | g.emit_list[g.emit_off[s]]) : (uint16_t)1;
        flat[(size_t)q * stride + C + 1] = (uint16_t)q;
        flat[(size_t)q * stride + C + 2] = g.end_off[(size_t)s + 1] != g.end_off[s] ? (uint16_t)1 : (uint16_t)0;  // the END cell: the field ending in this state emits (a list)
        emit_list.insert(emit_list.end(), g.emit_list.begin() + g.emit_off[s], g.emit_list.begin() + g.emit_off[(size_t)s + 1]);
        emit_off.push_back((uint32_t)emit_list.size());
        end_list.insert(end_list.end(), g.end_list.begin() + g.end_off[s], g.end_list.begin() + g.end_off[(size_t)s + 1]);
        end_off.push_back((uint32_t)end_list.size());
    }
    const std::vector<uint8_t> cm = class_image(g, nullptr, d.ill_class);
    d.scalar_mode = g.umap.on();
    int rc;
    if ((rc = upload(d.flat, flat, 16))) return rc;  // (the LDS staging copies whole 16-byte units)
    if (delta_rec.empty()) delta_rec.push_back(0);
    if ((rc = upload(d.delta, delta_rec, 16))) return rc;
    if ((rc = upload(d.flat_classmap, cm))) return rc;
    if ((rc = upload(d.emit_off, emit_off))) return rc;
    if ((rc = upload(d.emit_list, emit_list))) return rc;
    if ((rc = upload(d.end_off, end_off))) return rc;
    return upload(d.end_list, end_list);
}

int validate_batch_header(const pwaf_batch *b) {
    if (!b) return fail(PWAF_E_INVALID_ARG, "batch is NULL");
    if (b->struct_size != sizeof(pwaf_batch)) return fail(PWAF_E_INVALID_ARG, "pwaf_batch.struct_size mismatch");
    if (b->memory != PWAF_MEM_HOST && b->memory != PWAF_MEM_DEVICE) return fail(PWAF_E_INVALID_ARG, "pwaf_batch.memory is neither HOST nor DEVICE");
    if (b->n == 0) return PWAF_OK;
    for (int f = 0; f < PWAF_N_FIELDS; f++)
        if (!b->field[f].data || !b->field[f].offsets) return fail(PWAF_E_INVALID_ARG, "pwaf_batch.field column is NULL");
    if (!b->ip || !b->ip_is_v6 || !b->port || !b->flags) return fail(PWAF_E_INVALID_ARG, "pwaf_batch numeric column is NULL");
    if ((b->asn == nullptr) != (b->country == nullptr)) return fail(PWAF_E_INVALID_ARG, "pwaf_batch.asn and .country must be given together");
    return PWAF_OK;
}

// the trie part of the kernel arguments (shared by the per-batch pipeline and the one-off DIR-24 table build)
void set_trie_args(const pwaf_engine *e, VerdictArgs &v) {
    const Program &P = *e->prog.p;
    // a family without prefixes walks an all-leaf root: the kernels never test the pointers
    const uint32_t *leaf = (const uint32_t *)e->leaf_root.p;
    v.ip_root4 = P.ipset_trie.root4.empty() ? leaf : (const uint32_t *)e->ip_root4.p;
    v.ip_root6 = P.ipset_trie.root6.empty() ? leaf : (const uint32_t *)e->ip_root6.p;
    v.ip_nodes = (const uint32_t *)e->ip_nodes.p;
    v.set_masks = (const uint32_t *)e->set_masks.p;
    v.set_words = P.set_words;
    v.n_ip_lists = P.n_ip_lists;
    const uint32_t *geo_leaf = (const uint32_t *)e->geo_leaf_root.p;
    v.geo_root4 = P.geo_trie.root4.empty() ? geo_leaf : (const uint32_t *)e->geo_root4.p;
    v.geo_root6 = P.geo_trie.root6.empty() ? geo_leaf : (const uint32_t *)e->geo_root6.p;
    v.geo_nodes = (const uint32_t *)e->geo_nodes.p;
    v.has_geo = P.has_geo ? 1u : 0u;
    v.dir24 = (const uint32_t *)e->dir24.p;
    v.dir_chunks = (const uint32_t *)e->dir_chunks.p;
    v.dir_vals = (const uint32_t *)e->dir_vals.p;
    v.dir_esc = (const uint2 *)e->dir_esc.p;
    v.dir_summary = (const uint32_t *)e->dir_summary.p;
    v.dir_sum_shift = e->dir_sum_shift;
    v.dir_common = e->dir_common;
    v.class_rows = (const uint32_t *)e->class_rows.p;
    v.class_words = e->class_words;
    v.acmp_words = e->acmp_words;
    v.geo_default = e->geo_default;
    const size_t n_sets = P.set_words ? P.set_masks.size() / P.set_words : 1;
    v.ipres_packed = (e->n_classes <= 65536u && n_sets <= 65536u) ? 1u : 0u;
}

// Decides which passes are list-driven and uploads what that needs: a pass behind a bigram prefilter walks the filter's candidate
// list, a gated gap pass the list fed by its prefilter factors (owned by earlier passes). Called at creation and again when
// pwaf_engine_tune has rebuilt the filters from a traffic sample.
int assign_lists(pwaf_engine *e) {
    const Program &P = *e->prog.p;
    int rc;
    e->n_gated = e->n_filtered = e->n_gap = 0;
    e->owns_factors.assign(P.groups.size(), 0);
    std::vector<uint32_t> colmask(P.n_cols, 0);
    // list slots: gated gap passes own slots [0, kGapLists) — their index is also their bit in the factor masks — and every
    // filtered pass one slot after those
    for (size_t k = 0; k < P.groups.size(); k++) {
        DevGroup &d = e->groups[k];
        d.gate = -1;
        d.filtered = false;
        d.confirm = d.confirm_walk = false;
        const bool gap = !P.groups[k].filter_cols.empty();
        if (gap) {
            if (e->n_gap >= kGapLists) continue;  // (beyond 32 gap passes the rest simply walk every request)
            d.gate = (int)e->n_gap;
            for (uint32_t c : P.groups[k].filter_cols) colmask[c] |= 1u << e->n_gap;
            e->n_gap++;
        } else if (d.filter.enabled) {
            d.gate = (int)(kGapLists + e->n_filtered);
            d.filtered = true;
            e->n_filtered++;
            if ((rc = upload(d.ftable, d.filter.table))) return rc;
            // the confirm tier built with this filter (program.h: ConfirmTable)
            const ConfirmTable &ct = d.filter.confirm;
            d.confirm = ct.enabled;
            d.confirm_walk = d.confirm && ct.has_walk;
            if (d.confirm && ((rc = upload(d.c_head, ct.head)) || (rc = upload(d.c_entries, ct.entries)) || (rc = upload(d.c_bytes, ct.bytes)) || (rc = upload(d.c_classes, ct.classes)))) return rc;
        }
    }
    e->n_gated = e->n_gap || e->n_filtered ? kGapLists + e->n_filtered : 0;
    for (size_t k = 0; k < P.groups.size(); k++)
        for (uint32_t c = P.groups[k].atom_base; c < P.groups[k].atom_base + P.groups[k].n_local; c++)
            if (colmask[c]) e->owns_factors[k] = 1;
    // gap passes whose factors all live in ONE filtered pass walk that pass's candidate list (no atomics, no list of their own)
    e->n_need = e->n_visit = 0;
    for (auto &d : e->groups) { d.share_owner = -1; d.shared_bits = 0; d.need_slot = -1; d.visit_slot = -1; d.identity = false; }
    for (size_t k = 0; k < P.groups.size(); k++) {
        DevGroup &d = e->groups[k];
        if (d.gate >= 0 && !d.filtered) {
            d.visit_slot = d.gate;  // (a gap pass's visited bitmap is indexed by its list slot: enqueueing sets the bit)
            e->n_visit = std::max(e->n_visit, (uint32_t)d.gate + 1u);
        }
        // a plain pass over a field of a few bytes: the streaming DFA kernel's per-request machinery costs more than the walk itself
        const double ml = P.groups[k].field < e->mean_len.size() ? e->mean_len[P.groups[k].field] : 0.0;
        if (d.gate < 0 && (P.groups[k].field == PWAF_FIELD_METHOD ? (ml == 0 || ml < 12) : (ml > 0 && ml < 12))) d.identity = true;
    }
    // A pass whose atoms are all anchored literals of <= 8 bytes (`method == "POST"`: the only thing rules ask of the method) is not
    // walked at all: the attribute kernel compares the field's first 8 bytes (one field per engine: the first pass that qualifies).
    {
        std::vector<ShortAtom> sa;
        e->short_field = -1;
        for (size_t k = 0; k < P.groups.size(); k++) {
            DevGroup &d = e->groups[k];
            d.short_lit = false;
            const DfaGroup &g = P.groups[k];
            if (d.gate >= 0 || e->owns_factors[k] || g.field >= PWAF_N_FIELDS || e->short_field >= 0 || (P.flags & PWAF_OPT_NO_PREFILTER)) continue;
            std::vector<ShortAtom> mine;
            for (uint32_t l = 0; l < g.atoms.size(); l++) {
                const Atom &at = P.atoms[g.atoms[l]];
                std::string lit;
                bool exact = false;
                if (!at.pattern || !short_literal_atom(*at.pattern, lit, exact)) { mine.clear(); break; }
                uint8_t b[8] = {0};
                memcpy(b, lit.data(), lit.size());
                ShortAtom x{};
                x.col = g.atom_base + l;
                x.len_exact = (uint32_t)lit.size() | (exact ? 0x100u : 0u);
                memcpy(&x.lit_lo, b, 4);
                memcpy(&x.lit_hi, b + 4, 4);
                mine.push_back(x);
            }
            if (mine.empty()) continue;
            d.short_lit = true;
            d.identity = false;
            e->short_field = (int)g.field;
            sa = mine;
        }
        e->n_short = (uint32_t)sa.size();
        if (!sa.empty() && (rc = upload(e->short_atoms, sa))) return rc;
    }
    for (size_t k = 0; k < P.groups.size(); k++) {
        DevGroup &d = e->groups[k];
        if (d.gate < 0 || d.filtered) continue;
        int owner = -1;
        bool single = true;
        for (uint32_t c : P.groups[k].filter_cols) {
            int o = -1;
            for (size_t q = 0; q < P.groups.size(); q++)
                if (c >= P.groups[q].atom_base && c < P.groups[q].atom_base + P.groups[q].n_local) o = (int)q;
            if (o < 0 || (owner >= 0 && o != owner)) single = false;
            owner = o;
        }
        // (an owner with a confirm tier shares its WALK list — a literal hit that calls for a sharing gap pass sends the request through the
        // walk — unless it never walks: then it enqueues)
        if (!single || owner < 0 || !e->groups[owner].filtered || (e->groups[owner].confirm && !e->groups[owner].confirm_walk)) continue;
        d.share_owner = owner;
        e->groups[owner].shared_bits |= 1u << d.gate;
        if (e->groups[owner].need_slot < 0) e->groups[owner].need_slot = (int)e->n_need++;
    }
    {
        // the verdict kernel's pass table: first column + where the pass's visited bitmap lives
        std::vector<PassInfo> pt(e->groups.size() + 2);
        pt[e->groups.size()] = PassInfo{P.fcmp_base, 0u};  // the pseudo pass of the field-against-field atoms (dense records)
        // ... and the one of the residual rules (right after it, or in its place when there are no such atoms)
        // (dense records from the interpreter kernel; none when the specialized program runs: the verdict kernel reads its result words)
        pt[e->groups.size() + (P.fcmp.empty() ? 0u : 1u)] = PassInfo{P.residual_base, e->residual_jit.function ? (3u << 24) : 0u};
        uint32_t fi = 0;
        for (size_t k = 0; k < e->groups.size(); k++) {
            const DevGroup &d = e->groups[k];
            pt[k].base = d.atom_base;
            pt[k].kind_slot = 0;
            if (d.short_lit) {
                pt[k].kind_slot = 3u << 24;  // no records at all
            } else if (d.filtered) {
                // (a pass with heads writes records outside its candidate list too: its records are zeroed and read densely)
                if (d.filter.heads.empty()) pt[k].kind_slot = (1u << 24) | fi;
                fi++;
            } else if (d.visit_slot >= 0) {
                pt[k].kind_slot = (2u << 24) | (uint32_t)d.visit_slot;
            }
        }
        if ((rc = upload(e->pass_table, pt))) return rc;
    }
    return upload(e->colmask, colmask);
}

// The scratch context of a call, locked. The batch will run on `caller` (device batches; NULL is HIP's default stream) or, with
// `own`, on the context's own stream (host batches). Preference: the context whose latest batch ran on the SAME stream (stream
// order already protects the buffers: a single-stream caller keeps using one context, and one context's worth of memory), then
// a fresh one, then the next of the ring — only in that last case must the stream wait (on the device) for the previous user.
// (Deliberately no "whichever context is idle by hipEventQuery": the choice must not depend on timing.)
Scratch &acquire_context(pwaf_engine *e, bool own, hipStream_t caller, std::unique_lock<std::mutex> &held, bool &must_wait) {
    must_wait = false;
    for (int pass = 0; pass < 2; pass++)
        for (auto &c : e->ctx) {
            std::unique_lock<std::mutex> l(c->mu, std::try_to_lock);
            if (!l.owns_lock()) continue;
            const bool ok = pass == 0 ? (c->used && (own ? c->last_own : (!c->last_own && c->last == caller))) : !c->used;
            if (ok) { held = std::move(l); return *c; }
        }
    size_t k;
    {
        std::lock_guard<std::mutex> lock(e->mu);
        k = e->next_ctx;
        e->next_ctx = (e->next_ctx + 1) % e->ctx.size();
    }
    held = std::unique_lock<std::mutex>(e->ctx[k]->mu);
    must_wait = e->ctx[k]->used;
    return *e->ctx[k];
}

int run_pipeline(pwaf_engine *e, Scratch &S, const pwaf_batch &db /* device pointers */, pwaf_verdict *d_out, pwaf_counts *d_counts, uint32_t *d_match_idx,
                 uint32_t *d_n_matches, hipStream_t stream, bool totals_known = false, const std::vector<uint32_t> *col_begin = nullptr, bool sync_status = false) {
    const Program &P = *e->prog.p;
    const uint32_t n = db.n, n_groups = (n + 63) / 64;
    if (n == 0) return PWAF_OK;
    const uint32_t residual_pass = (uint32_t)e->groups.size() + (P.fcmp.empty() ? 0u : 1u);
    const uint32_t n_passes = residual_pass + (P.n_residual ? 1u : 0u);  // (+ the pseudo passes of the field-against-field atoms and of the residual rules)
    int rc;
    // scratch sized for the worst case the 288 GB part can afford: one 4-byte hit record per (pass, request) and an overflow
    // pool of 8 entries per request (exhaustion is reported through the status word, never silently)
    // (a batch that exhausts the pool sets the context's sticky status word; the synchronous entry points then grow the pool to
    // what the batch asked for — the allocator keeps counting past the cap — and run the batch again)
    const uint64_t pool_cap64 = std::max<uint64_t>(std::max<uint64_t>(1u << 20, (uint64_t)n * 8), S.pool_entries);
    const uint32_t pool_cap = (uint32_t)std::min<uint64_t>(pool_cap64, 0x7FFFFFF0u);
    S.pool_entries = pool_cap;
    // two sticky status words per context: [0] device-resident batches (reported and cleared by pwaf_engine_device_status),
    // [1] the synchronous entry points (read and cleared by their own retry loop) — a synchronous batch on the same context can no
    // longer swallow the overflow bit of an earlier asynchronous one
    if (!S.status.p) {
        if ((rc = S.status.reserve(8))) return rc;
        HIP_TRY(hipMemsetAsync(S.status.p, 0, 8, stream));
    }
    uint32_t *const status_word = (uint32_t *)S.status.p + (sync_status ? 1 : 0);
    if ((rc = S.rec.reserve((size_t)std::max(1u, n_passes) * n * 4))) return rc;
    // the attribute kernel's output: per group a header and room for EVERY non-scan atom (worst case: no overflow path), plus 64
    // pairs of slack so the verdict kernel may read a full wave's worth unconditionally
    const uint32_t pair_stride = std::max(1u, e->n_bit_atoms + e->n_cmp_atoms + e->n_short);
    if ((rc = S.attr.reserve(((size_t)n_groups * pair_stride + 64) * 16 + (size_t)n_groups * 4))) return rc;
    if ((rc = S.pool.reserve((size_t)pool_cap * sizeof(PoolEntry)))) return rc;
    const size_t n_slots = (size_t)std::max(kGapLists, e->n_gated);
    const size_t ctrl_words = 2 + n_slots + e->n_filtered;  // [0] pool allocator, [1] status word, one length per list slot, one pair count per filtered pass
    // visited bitmaps of the list-driven passes (one bit per request, whole 64-request groups): zeroed per batch — 1/32 of what
    // zeroing the hit records themselves would write. They share ONE zeroed block with the control words and the confirm tier's walk bitmaps.
    const uint32_t bit_words = 2 * n_groups;
    uint32_t n_conf = 0;
    for (const DevGroup &d : e->groups) n_conf += (d.filtered && d.confirm) ? 1u : 0u;
    bool zero_deferred = false;
    size_t zero_bytes = 0;
    int arg_slot_used = -1;
    {
        size_t zb = 0;
        auto part = [&](size_t bytes) { const size_t at = zb; zb = (zb + bytes + 255) & ~(size_t)255; return at; };
        const size_t z_ctrl = part(4 * ctrl_words), z_cand = part((size_t)e->n_filtered * bit_words * 4), z_visit = part((size_t)e->n_visit * bit_words * 4),
                     z_walk = part((size_t)n_conf * bit_words * 4);
        if ((rc = S.zero_block.reserve(zb))) return rc;
        // (cleared by the descriptor upload's launch when nothing needs it before that: no pass that streams every request — those run first and
        // count into the control words)
        zero_deferred = e->n_filtered != 0;
        for (const DevGroup &d : e->groups) zero_deferred = zero_deferred && (d.gate >= 0 || d.identity || d.short_lit);
        zero_bytes = zb;
        if (!zero_deferred) HIP_TRY(hipMemsetAsync(S.zero_block.p, 0, zb, stream));
        char *const zbase = (char *)S.zero_block.p;
        S.ctrl.p = zbase + z_ctrl;
        S.cand_bits.p = zbase + z_cand;
        S.visit_bits.p = zbase + z_visit;
        S.walk_bits.p = zbase + z_walk;
    }
    // string columns by field id: the five fixed fields, then one column per header name the rule set mentions (EXTENSION); a header
    // the batch does not carry reads as the empty string for every request
    std::vector<pwaf_strcol> cols(e->n_fields);
    std::vector<uint32_t> col_bytes(e->n_fields, 0);
    for (int f = 0; f < PWAF_N_FIELDS; f++) {
        cols[(size_t)f] = db.field[f];
        col_bytes[(size_t)f] = db.field_bytes[f];
    }
    bool missing_header = false;
    for (uint32_t f = PWAF_N_FIELDS; f < e->n_fields; f++) {
        const uint32_t k = f - PWAF_N_FIELDS;
        if (k < db.n_headers && db.headers != nullptr && db.headers[k].data != nullptr && db.headers[k].offsets != nullptr) {
            cols[f] = db.headers[k];
            col_bytes[f] = db.header_bytes ? db.header_bytes[k] : 0u;
        } else {
            missing_header = true;
        }
    }
    if (missing_header) {
        if (S.zero_off.cap < (size_t)(n + 1) * 4 + PWAF_ARENA_PAD) {
            if ((rc = S.zero_off.reserve((size_t)(n + 1) * 4 + PWAF_ARENA_PAD))) return rc;
            HIP_TRY(hipMemsetAsync(S.zero_off.p, 0, S.zero_off.cap, stream));
        }
        for (uint32_t f = PWAF_N_FIELDS; f < e->n_fields; f++)
            if (cols[f].data == nullptr) {
                cols[f].data = (const uint8_t *)S.zero_off.p;
                cols[f].offsets = (const uint32_t *)S.zero_off.p;
                col_bytes[f] = 0;
            }
    }
    std::vector<uint8_t> col_known(e->n_fields, totals_known ? 1 : 0);
    for (uint32_t f = 0; f < e->n_fields; f++)
        if (col_bytes[f] != 0 || cols[f].data == (const uint8_t *)S.zero_off.p) col_known[f] = 1;
    if (e->n_gated && (rc = S.gate_lists.reserve((size_t)e->n_gated * n * 4))) return rc;
    if (e->n_need && (rc = S.need.reserve((size_t)e->n_need * n * 4))) return rc;

    // Profiling: HIP events on the launch stream. On the main stream the event that ends one kernel also starts the next
    // (half the events; the few microseconds of launch gap or memset in between are charged to the later kernel).
    // (evaluate calls are re-entrant, the timing tables are not — a profiled call holds this lock while it enqueues)
    std::unique_lock<std::mutex> prof_lock;
    if (e->profiling) prof_lock = std::unique_lock<std::mutex>(e->prof_mu);
    size_t ev_i = e->profiling ? e->n_timed : 0;
    long last_main = -1;  // index of the last event recorded on `stream` during this call
    auto record = [&](hipStream_t on) -> int {
        if (e->ev.size() < ev_i + 1) {
            hipEvent_t x;
            HIP_TRY(hipEventCreate(&x));
            e->ev.push_back(x);
        }
        HIP_TRY(hipEventRecord(e->ev[ev_i], on));
        ev_i++;
        return PWAF_OK;
    };
    long open_begin = -1;
    bool stream_mark = false;  // the marks of a launch that streams request bytes (level 2 records no others)
    auto mark = [&](const char *name, uint64_t alg_bytes, hipStream_t on = nullptr) -> int {
#ifdef PWAF_PROFILING
        static const bool sync_each = getenv("PWAF_SYNC_EACH") != nullptr;  // debugging aid: which launch faults
        if (sync_each && name) {
            hipError_t se = hipStreamSynchronize(on ? on : stream);
            fprintf(stderr, "[pwaf] %s done: %s\n", name, hipGetErrorString(se));
        }
#endif
        if (!e->profiling || (e->profiling == 2 && !stream_mark)) return PWAF_OK;
        int rc2;
        if (!name) {  // begin
            if (!on && last_main >= 0) { open_begin = last_main; return PWAF_OK; }
            if ((rc2 = record(on ? on : stream))) return rc2;
            open_begin = (long)ev_i - 1;
            if (!on) last_main = open_begin;
            return PWAF_OK;
        }
        if ((rc2 = record(on ? on : stream))) return rc2;
        if (!on) last_main = (long)ev_i - 1;
        pwaf_kernel_time t{};
        snprintf(t.name, sizeof t.name, "%s", name);
        t.alg_bytes = alg_bytes;
        e->times.push_back(t);
        e->time_ev.push_back({(size_t)open_begin, ev_i - 1});
        return PWAF_OK;
    };

    VerdictArgs v{};
    v.n = n;
    v.n_groups = n_groups;
    v.force_global_tables = (P.flags & PWAF_OPT_GLOBAL_VERDICT_TABLES) ? 1u : 0u;
    v.sparse_mode = verdict_mode(P.flags);
    if (v.sparse_mode >= 3) {
        // the entry list: entries per wave in LDS, and the per-wave spill array (by column) for a group that appends more
        const VerdictShape vs = verdict_shape(P.n_cols, (uint32_t)P.rules.size(), e->n_trig, (uint32_t)P.lits.size(), v.force_global_tables != 0, (int)v.sparse_mode, n_passes);
        v.v_cap = vs.v_cap;
        if ((rc = S.verdict_spill.reserve((size_t)verdict_blocks_sp(vs, e->n_cus) * vs.waves * P.n_cols * 8))) return rc;
        v.spill = (unsigned long long *)S.verdict_spill.p;
    } else if (v.sparse_mode) {
        // the sparse column file: value slots per wave, and the per-wave spill array for a group that dirties more columns than that
        const VerdictShape vs = verdict_shape(P.n_cols, (uint32_t)P.rules.size(), e->n_trig, (uint32_t)P.lits.size(), v.force_global_tables != 0, (int)v.sparse_mode, n_passes);
        v.v_cap = vs.v_cap;
        if (vs.v_cap < P.n_cols) {
            if ((rc = S.verdict_spill.reserve((size_t)verdict_blocks_sp(vs, e->n_cus) * vs.waves * (P.n_cols - vs.v_cap) * 8))) return rc;
            v.spill = (unsigned long long *)S.verdict_spill.p;
        }
    }
#ifdef PWAF_PROFILING
    {
        static const uint32_t skip = getenv("PWAF_DEBUG_SKIP") ? (uint32_t)strtoul(getenv("PWAF_DEBUG_SKIP"), nullptr, 0) : 0u;
        v.debug_skip = skip;  // timing experiments only: results are wrong when non-zero
        static const bool attr_prio = getenv("PWAF_ATTR_PRIO") != nullptr;
        if (attr_prio) v.debug_skip |= 0x80000000u;
    }
#endif
    for (int f = 0; f < PWAF_N_FIELDS; f++) v.off[f] = db.field[f].offsets;
    v.ip = db.ip;
    v.ip_is_v6 = db.ip_is_v6;
    v.port = db.port;
    v.flags = db.flags;
    v.asn = db.asn;
    v.country = db.country;
    v.n_cols = P.n_cols;
    v.n_passes = n_passes;
    v.rec = (const uint32_t *)S.rec.p;
    v.passes = (const PassInfo *)e->pass_table.p;
    if (P.n_residual && e->residual_jit.function) {
        v.res_words = (P.n_residual + 31u) / 32u;
        v.res_base = P.residual_base;
        if ((rc = S.res_words.reserve((size_t)v.res_words * n * 4))) return rc;
        v.res_match = (const uint32_t *)S.res_words.p;
    }
    v.cand_bits = (const uint32_t *)S.cand_bits.p;
    v.visit_bits = (const uint32_t *)S.visit_bits.p;
    v.bit_words = bit_words;
    // A pass with heads writes records outside its candidate list too: zeroed here, read densely. (A confirm tier MERGES hits into zeroed
    // records: resolve_kernel zeroes the records of the requests that have a flagged chunk — the only ones a hit can land in.) Passes are
    // laid out so that the filtered ones are neighbours: runs of passes to zero take ONE memset each.
    for (size_t k = 0; k < e->groups.size();) {
        auto zeroed = [&](size_t q) { return e->groups[q].filtered && !e->groups[q].filter.heads.empty(); };
        if (!zeroed(k)) { k++; continue; }
        size_t k1 = k;
        while (k1 < e->groups.size() && zeroed(k1)) k1++;
        HIP_TRY(hipMemsetAsync((uint32_t *)S.rec.p + k * (size_t)n, 0, (k1 - k) * (size_t)n * 4, stream));
        k = k1;
    }
    v.n_hlen = (uint32_t)e->hlen_fields.size();
    for (size_t k = 0; k < e->hlen_fields.size(); k++) v.hoff[k] = cols[e->hlen_fields[k]].offsets;
    v.gpairs = (uint4 *)S.attr.p;
    v.ghdr = (uint32_t *)((char *)S.attr.p + ((size_t)n_groups * pair_stride + 64) * 16);
    v.pair_stride = pair_stride;
    v.attr_blocks = e->n_cus;
    v.pool = (const PoolEntry *)S.pool.p;
    v.cmp = (const CmpAtomDev *)e->num_atoms.p;
    v.n_cmp = e->n_cmp_atoms;
    v.cmp_vars = e->cmp_vars;
    v.lazy = (const CmpAtomDev *)e->lazy_atoms.p;
    v.n_lazy = e->n_lazy;
    v.n_lazy_var = (uint32_t)e->lazy_vars.size();
    for (size_t k = 0; k < e->lazy_vars.size(); k++) v.lazy_var[k] = e->lazy_vars[k];
    v.n_short = e->n_short;
    v.short_atoms = (const ShortAtom *)e->short_atoms.p;
    if (e->n_short) {
        v.short_data = cols[(size_t)e->short_field].data;
        v.short_off = cols[(size_t)e->short_field].offsets;
    }
    v.n_trig = e->n_trig;
    v.n_lits = (uint32_t)P.lits.size();
    v.bit_col = (const uint32_t *)e->bit_atoms.p;
    v.trig_off = (const uint32_t *)e->trig_off.p;
    v.trig_rules = (const uint16_t *)e->trig_rules.p;
    v.always_rules = (const uint32_t *)e->always_rules.p;
    for (int var = 0; var < 2; var++) {
        v.iu_vals[var] = (const int64_t *)e->iu_vals[var].p;
        v.iu_masks[var] = (const uint32_t *)e->iu_masks[var].p;
        v.iu_n[var] = e->iu_n[var];
        v.iu_words[var] = e->iu_words[var];
    }
    v.country_masks = (const uint32_t *)e->country_luts.p;
    v.cc_words = e->cc_words;
    v.rules = (const DevRule *)e->rules.p;
    v.n_rules = (uint32_t)P.rules.size();
    v.lits = (const uint32_t *)e->lits.p;
    set_trie_args(e, v);
    if ((rc = S.ipres.reserve((size_t)n * (v.ipres_packed ? 4 : 8) + 64))) return rc;
    v.ipres = (uint32_t *)S.ipres.p;
    v.out = d_out;
    v.counts = (unsigned long long *)d_counts;
    v.match_idx = d_match_idx;
    v.n_matches = d_n_matches;
    // The attribute path — ipres_kernel (address lookups: one scattered 16-byte gather per request, bound by the texture addresser) and
    // attr_kernel (rows, transposes, comparisons: bound by vector-ALU issue) — does not depend on the scans. Placement (measured on
    // MI355X, 10M requests; `PWAF_PLACEMENT` in profiling builds):
    //   0  ipres on the side stream from the START of the batch, attr_kernel on the main stream after the prefilter    1.838 ms / step
    //   1  both on the main stream after the prefilter (every kernel alone)                                              1.861
    //   2  both on the side stream, forked after the prefilter                                        <- default          1.779
    //   3  both on the side stream from the start                                                                         1.762
    // 3 is 1 % faster than 2 but slows the prefilter launch itself from 0.73 to 0.95 ms (0 to 0.86): the streaming kernel is what the
    // roofline is quoted on, so it runs undisturbed.
    // Kernels that run beside each other take about as long as one after the other here (the step is the SUM of what its kernels
    // cost alone, give or take 2 %), so the placement only matters where the two really use different units.
    int placement = 2;
#ifdef PWAF_PROFILING
    static const int forced_placement = getenv("PWAF_PLACEMENT") ? atoi(getenv("PWAF_PLACEMENT")) : (getenv("PWAF_ATTR_INLINE") ? 1 : getenv("PWAF_ATTR_EARLY") ? 3 : 2);
    placement = forced_placement;
#endif
    bool ipres_launched = false, attr_launched = false;
    auto launch_ipres_stage = [&](hipStream_t q) -> int {
        if (ipres_launched) return PWAF_OK;
        ipres_launched = true;
        int rc2, he = 0;
        if (q != stream) {
            HIP_TRY(hipEventRecord(S.ev_fork, stream));
            HIP_TRY(hipStreamWaitEvent(q, S.ev_fork, 0));
        }
        if ((rc2 = mark(nullptr, 0, q != stream ? q : nullptr))) return rc2;
#ifdef PWAF_PROFILING
        static const bool skip_ipres = getenv("PWAF_SKIP_IPRES") != nullptr || getenv("PWAF_SKIP_ATTR") != nullptr;  // timing experiments only: every address resolves to class 0 / set 0
        if (skip_ipres) HIP_TRY(hipMemsetAsync(v.ipres, 0, (size_t)n * (v.ipres_packed ? 4 : 8), q));
        else
#endif
        he = launch_ipres(v, q);
        if (he) return fail(PWAF_E_DEVICE, std::string("ipres kernel launch failed: ") + hipGetErrorString((hipError_t)he));
        if ((rc2 = mark("ipres", 0xFAu, q != stream ? q : nullptr))) return rc2;
        if (q != stream) HIP_TRY(hipEventRecord(S.ev_join, q));
        return PWAF_OK;
    };
    auto launch_attr_stage = [&](hipStream_t q) -> int {
        if (attr_launched) return PWAF_OK;
        attr_launched = true;
        int rc2, he = 0;
        if (q == stream && (placement == 0)) HIP_TRY(hipStreamWaitEvent(stream, S.ev_join, 0));  // the side stream's ipres results
        if ((rc2 = mark(nullptr, 0, q != stream ? q : nullptr))) return rc2;
#ifdef PWAF_PROFILING
        static const bool skip_attr = getenv("PWAF_SKIP_ATTR") != nullptr;  // timing experiments only: every non-scan predicate reads false
        if (skip_attr) HIP_TRY(hipMemsetAsync(v.ghdr, 0, (size_t)n_groups * 4, q));
        else
#endif
        he = launch_attr(v, q);
        if (he) return fail(PWAF_E_DEVICE, std::string("attribute kernel launch failed: ") + hipGetErrorString((hipError_t)he));
        if ((rc2 = mark("attr", 0xFEu, q != stream ? q : nullptr))) return rc2;
        if (q != stream) HIP_TRY(hipEventRecord(S.ev_join, q));
        return PWAF_OK;
    };
    // batch start: what forks here
    if (placement == 0 && (rc = launch_ipres_stage(S.side))) return rc;
    if (placement == 3 && ((rc = launch_ipres_stage(S.side)) || (rc = launch_attr_stage(S.side)))) return rc;
    // after the prefilter (or, without one, before the list scans)
    auto launch_attr_side = [&]() -> int {
        int rc2;
        if (placement == 0) return launch_attr_stage(stream);
        if (placement == 1) { if ((rc2 = launch_ipres_stage(stream))) return rc2; return launch_attr_stage(stream); }
        if (placement == 2) { if ((rc2 = launch_ipres_stage(S.side))) return rc2; return launch_attr_stage(S.side); }
        return PWAF_OK;
    };

    static const char *fn[5] = {"host", "url", "path", "method", "user_agent"};
    auto scan_args = [&](size_t gi) -> ScanArgs {
        const DevGroup &d = e->groups[gi];
        ScanArgs a{};
        if (e->n_gap && e->owns_factors[gi]) {
            // this pass owns prefilter factors: it feeds the gated gap passes' request lists as requests finish
            a.colmask_local = (const uint32_t *)e->colmask.p + d.atom_base;
            a.n_local = d.n_local;
            a.gate_lists = (uint32_t *)S.gate_lists.p;
            a.gate_count = (uint32_t *)S.ctrl.p + 2;
        }
        a.data = cols[d.field].data;
        a.off = cols[d.field].offsets;
        a.n = n;
        a.tab = (const uint16_t *)d.tab.p;
        a.classmap = (const uint8_t *)d.classmap.p;
        a.umap = d.scalar_mode ? (const uint8_t *)d.classmap.p + kUmapAt : nullptr;
        a.ill_class = d.ill_class;
        a.special = (const SpecialCell *)d.special.p;
        a.list_off = (const uint32_t *)d.list_off.p;
        a.list = (const uint16_t *)d.list.p;
        a.start_emit = d.start_emit;
        a.emit_base = d.emit_base;
        a.n_cus = e->n_cus;
        a.chunks = d.chunks;
        a.special_base = d.special_base;
        a.n_states = d.n_states;
        a.stride = d.stride;
        a.n_classes = d.n_classes;
        a.n_hot = d.n_hot;
        a.rec = (uint32_t *)S.rec.p + gi * (size_t)n;
        a.pool = (PoolEntry *)S.pool.p;
        a.pool_count = (uint32_t *)S.ctrl.p;
        a.pool_cap = pool_cap;
        a.status = status_word;
        return a;
    };
    // (the shape is per PHASE: the filtered passes' candidate lists are long, the gap passes' short)
    uint32_t list_variant[2] = {2, 0};  // long candidate lists: 1024 threads over 144 KiB of hot rows (measured: benign 0.169 -> 0.143 ms, adversarial 4.43 -> 3.87 ms); short gap lists: 3 x 48 KiB
#ifdef PWAF_PROFILING
    static const uint32_t forced_shape = getenv("PWAF_LIST_SHAPE") ? (uint32_t)atoi(getenv("PWAF_LIST_SHAPE")) : 0u;  // timing experiments (same results): phase 0 | phase 1 << 4
    if (getenv("PWAF_LIST_SHAPE")) {
        list_variant[0] = forced_shape & 15u;
        list_variant[1] = forced_shape >> 4;
    }
#endif
    {
        // With a confirm tier on every filtered pass the phase-0 lists are short walk lists: the small workgroup shape (512 threads,
        // 48 KiB) starts on whatever wave slots the attribute kernels leave free — the 1024-thread / 144 KiB shape had to wait for a whole
        // free CU (measured: 0.04 ms alone, 0.17 ms beside the side stream).
        bool all_confirm = e->n_filtered != 0;
        for (const DevGroup &d : e->groups) all_confirm = all_confirm && (!d.filtered || d.confirm);
#ifdef PWAF_PROFILING
        if (!getenv("PWAF_LIST_SHAPE"))
#endif
        if (all_confirm) list_variant[0] = 0;
    }
    const ListShape lshapes[2] = {list_shape(list_variant[0]), list_shape(list_variant[1])};
    auto list_args = [&](size_t gi, const ListShape &lshape, bool full_table = false) -> ListScanArgs {
        const DevGroup &d = e->groups[gi];
        ListScanArgs a{};
        const DevGroup &src = d.share_owner >= 0 ? e->groups[(size_t)d.share_owner] : d;  // whose list this pass walks
        if (!d.identity) {
            a.req_list = (const uint32_t *)S.gate_lists.p + (size_t)src.gate * n;
            a.n_list = (const uint32_t *)S.ctrl.p + 2 + src.gate;
        }
        if (d.visit_slot >= 0) a.visited = (uint32_t *)S.visit_bits.p + (size_t)d.visit_slot * bit_words;
        if (d.share_owner >= 0) {
            a.need_in = (const uint32_t *)S.need.p + (size_t)src.need_slot * n;
            a.need_bit = (uint32_t)d.gate;
        }
        if (e->n_gap && e->owns_factors[gi]) {
            a.colmask_local = (const uint32_t *)e->colmask.p + d.atom_base;
            a.n_local = d.n_local;
            a.gate_lists = (uint32_t *)S.gate_lists.p;
            a.gate_count = (uint32_t *)S.ctrl.p + 2;
            a.enq_bits = (uint32_t *)S.visit_bits.p;  // (a gap pass's visited bitmap, indexed by its list slot: set when the request is enqueued)
            a.enq_words = bit_words;
            if (d.need_slot >= 0) {
                a.need_out = (uint32_t *)S.need.p + (size_t)d.need_slot * n;
                a.shared_bits = d.shared_bits;
            }
        }
        if (d.confirm) a.merge_rec = 1;  // the R-tier walk of a pass with a confirm tier: its list is confirm_kernel's walk list, the walk starts from the record it merged
        a.behind_filter = d.filtered ? 1u : 0u;
        a.data = cols[d.field].data;
        a.off = cols[d.field].offsets;
        a.n = n;
        const FlatDev &F = (d.confirm && d.rt.n_states && !full_table) ? d.rt : d.fl;  // (a confirmed candidate walks the DFA of the pass's non-literal atoms)
        a.flat = (const uint16_t *)F.flat.p;
        a.classmap = (const uint8_t *)F.flat_classmap.p;
        a.umap = F.scalar_mode ? (const uint8_t *)F.flat_classmap.p + kUmapAt : nullptr;
        a.ill_class = F.ill_class;
        a.n_classes = F.n_classes;
        {
            // rows [0, n_full) and the delta records behind them, when this launch's LDS share holds the layout the tables were built for
            const uint32_t row_bytes = 2u * (F.n_classes + 3u), hb = list_hot_bytes(lshape);
            if (F.n_delta && (uint64_t)F.n_full * row_bytes + 48u + 8ull * F.n_delta <= hb) {
                a.n_hot = F.n_full;
                a.n_delta = F.n_delta;
                a.delta = (const uint64_t *)F.delta.p;
            } else {
                a.n_hot = std::min<uint32_t>(F.n_delta ? F.n_full : F.n_states, (hb - 48u) / row_bytes);  // (states are in visit order: the first rows are the hot ones; 48 bytes stay free for the sentinel cell and lscan_async's dummy record)
            }
        }
        a.emit_off = (const uint32_t *)F.emit_off.p;
        a.emit_list = (const uint16_t *)F.emit_list.p;
        a.end_off = (const uint32_t *)F.end_off.p;
        a.end_list = (const uint16_t *)F.end_list.p;
        a.rec = (uint32_t *)S.rec.p + gi * (size_t)n;
        a.pool = (PoolEntry *)S.pool.p;
        a.pool_count = (uint32_t *)S.ctrl.p;
        a.pool_cap = pool_cap;
        a.status = status_word;
        a.n_cus = e->n_cus;
        return a;
    };
    // ---- 1. plain passes: the DFA walks every request (short fields go through the list-scan kernel below) ----
    for (size_t gi = 0; gi < e->groups.size(); gi++) {
        const DevGroup &d = e->groups[gi];
        if (d.gate >= 0 || d.identity || d.short_lit) continue;
        const ScanArgs a = scan_args(gi);
        char nm[48];
        if (d.field < PWAF_N_FIELDS) snprintf(nm, sizeof nm, "scan_%s_g%zu", fn[d.field], gi);
        else snprintf(nm, sizeof nm, "scan_hdr%u_g%zu", d.field - PWAF_N_FIELDS, gi);
        stream_mark = true;
        if ((rc = mark(nullptr, 0))) return rc;
        int he = launch_scan(a, stream);
        if (he) return fail(PWAF_E_DEVICE, std::string("scan kernel launch failed: ") + hipGetErrorString((hipError_t)he));
        if ((rc = mark(nm, (uint64_t)col_bytes[d.field] + 4ull * (n + 1)))) return rc;  // algorithmic bytes: the field's bytes + its offsets (0 bytes when the arena size is unknown)
        stream_mark = false;
    }
    // ---- 2. the descriptors of every launch of the batch (prefilter, resolve, confirm tier, list scans, residual kernel): built here,
    //         uploaded ONCE ----
    std::vector<uint32_t> &totals = col_bytes;
    // the flag-density switch (kernels.h: ListScanArgs::dense_flag): per pass with a confirm tier, the device word that says "walked whole this batch"
    std::vector<const uint32_t *> dense_flag_of(e->groups.size(), nullptr);
    std::vector<uint32_t> dense_thresh_of(e->groups.size(), 0);
    std::vector<uint32_t> filter_index(e->groups.size(), 0);  // a filtered pass's index among the filtered passes (its candidate / valid bitmap)
    const bool dense_switch = !(P.flags & PWAF_OPT_NO_DENSE_SWITCH);
    std::vector<FilterArgs> fall;        // every filtered pass, in pass order
    std::vector<FilterArgs> by_stride[2];
    std::vector<ConfirmArgs> call;
    std::vector<ListScanArgs> la[2];
    uint64_t alg_bytes[3] = {0, 0, 0};  // per sampling stride
    if (e->n_filtered) {
        // arena sizes: a host batch's offsets were read while staging; a device batch says so itself or is asked (one small copy)
        bool ask = false;
        for (const DevGroup &d : e->groups)
            if (d.filtered && !col_known[d.field]) ask = true;
        if (ask) {
            for (const DevGroup &d : e->groups)
                if (d.filtered && !col_known[d.field]) HIP_TRY(hipMemcpyAsync(&totals[d.field], cols[d.field].offsets + n, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        const uint32_t words = (n + 31) / 32, n_cblocks = (words + kCompactWords - 1) / kCompactWords;
        uint64_t sub_entries = 0, n_slabs_all = 0;
        for (const DevGroup &d : e->groups)
            if (d.filtered) {
                const uint64_t slabs = ((uint64_t)totals[d.field] + kStreamSlab - 1) / kStreamSlab - (col_begin ? (*col_begin)[d.field] / kStreamSlab : 0u);
                n_slabs_all += slabs;
                sub_entries += slabs * (kStreamSlab / 512);  // chunk bitmap words
            }
        if ((rc = S.chunk_bits.reserve((size_t)sub_entries * 4))) return rc;
        if ((rc = S.cand_cnt.reserve((size_t)(2 * n_slabs_all + (uint64_t)e->n_filtered * n_cblocks) * 4))) return rc;  // per pass: flag counts per slab, pair-list starts per slab, candidates per compact workgroup
        uint32_t fi = 0;
        uint64_t sub_at = 0, cnt_at = 0, pair_at = 0;
        {
            uint64_t pair_bytes = 0;
            for (const DevGroup &d : e->groups)
                if (d.filtered && d.confirm) {
                    const uint64_t slabs = ((uint64_t)totals[d.field] + kStreamSlab - 1) / kStreamSlab - (col_begin ? (*col_begin)[d.field] / kStreamSlab : 0u);
                    pair_bytes += std::min<uint64_t>(0xFFFFFFF0u, slabs * (kStreamSlab / 16) + n + 64) * sizeof(uint2);
                }
            if (pair_bytes && (rc = S.pairs.reserve((size_t)pair_bytes))) return rc;
        }
        for (size_t gi = 0; gi < e->groups.size(); gi++) {
            const DevGroup &d = e->groups[gi];
            if (!d.filtered) continue;
            fall.emplace_back();
            FilterArgs &f = fall.back();
            f = FilterArgs{};
            f.data = cols[d.field].data;
            f.off = cols[d.field].offsets;
            f.n = n;
            f.total = totals[d.field];
            f.init = d.filter.init;
            f.mul = d.filter.mul;
            f.stride = d.filter.stride;
#ifdef PWAF_PROFILING
            static const uint32_t fdebug = getenv("PWAF_FILTER_DEBUG_SKIP") ? (uint32_t)atoi(getenv("PWAF_FILTER_DEBUG_SKIP")) : 0u;
            f.debug = fdebug;
#endif
            f.table = (const uint32_t *)d.ftable.p;
            f.n_heads = (uint32_t)std::min<size_t>(2, d.filter.heads.size());
            for (uint32_t h = 0; h < f.n_heads; h++) {
                const FilterHead &fh = d.filter.heads[h];
                uint8_t lit[16] = {0}, msk[16] = {0};
                memcpy(lit, fh.bytes, fh.len);
                memset(msk, 0xFF, fh.len);
                memcpy(f.head_w[h], lit, 16);
                memcpy(f.head_m[h], msk, 16);
                f.head_len[h] = fh.len | ((uint32_t)fh.exact << 8);
                f.head_code[h] = h == 0 ? (uint32_t)fh.local + 1u : ((uint32_t)fh.local + 1u) << 15;
            }
            f.slab0 = col_begin ? (*col_begin)[d.field] / kStreamSlab : 0u;
            const uint32_t slabs = (uint32_t)(((uint64_t)f.total + kStreamSlab - 1) / kStreamSlab) - f.slab0;
            f.rec = (uint32_t *)S.rec.p + gi * (size_t)n;
            f.chunk_bits = (uint32_t *)S.chunk_bits.p + sub_at;
            f.sub_count = (uint32_t *)S.cand_cnt.p + cnt_at;
            f.pair_base = f.sub_count + slabs;
            f.block_count = f.pair_base + slabs;
            f.bitmap = (uint32_t *)S.cand_bits.p + (size_t)fi * bit_words;
            filter_index[gi] = fi;
            f.list = (uint32_t *)S.gate_lists.p + (size_t)d.gate * n;
            f.list_count = (uint32_t *)S.ctrl.p + 2 + d.gate;
            if (d.confirm) {
                // (request, flagged chunk) pairs: at most one per chunk of the arena plus one per request (a chunk that straddles two requests counts for both)
                f.pair_cap = (uint32_t)std::min<uint64_t>(0xFFFFFFF0u, (uint64_t)slabs * (kStreamSlab / 16) + n + 64);
                f.pairs = (uint2 *)((char *)S.pairs.p + pair_at);
                f.pair_count = (uint32_t *)S.ctrl.p + 2 + n_slots + fi;
                pair_at += (uint64_t)f.pair_cap * sizeof(uint2);
                if (dense_switch) {
                    // more than half of the arena's 16-byte chunks flagged: confirming them one by one costs more than walking every request
                    // (measured, round 5: the saturated stream took 22 ms in the confirm tier against 9 ms for the plain DFA over the same bytes;
                    // the hostile stream of the 1k-rule set flags a quarter of its chunks and stays on the confirm tier: 1.3 ms against 3.9)
                    f.dense_flag = f.pair_count;  // (the count filter_kernel leaves there, against dense_thresh)
                    f.dense_thresh = (uint32_t)std::min<uint64_t>(0xFFFFFFFEu, (uint64_t)slabs * (kStreamSlab / 16) / 2);
                    dense_flag_of[gi] = f.dense_flag;
                    dense_thresh_of[gi] = f.dense_thresh;
                }
            }
            f.first_block = 0;
            sub_at += (uint64_t)slabs * (kStreamSlab / 512);
            cnt_at += 2 * slabs + n_cblocks;
            alg_bytes[f.stride == 2 ? 2 : 1] += (uint64_t)f.total + 4ull * (n + 1);
            fi++;
        }
        // the fused filter launch takes the stride-1 passes and the stride-2 passes as two tables, first_block numbered within each
        for (const FilterArgs &f : fall) {
            std::vector<FilterArgs> &sub = by_stride[f.stride == 2 ? 1 : 0];
            const uint32_t block = sub.empty() ? 0u : sub.back().first_block + ((uint32_t)(((uint64_t)sub.back().total + kStreamSlab - 1) / kStreamSlab) - sub.back().slab0 + kFilterWaves - 1) / kFilterWaves;
            sub.push_back(f);
            sub.back().first_block = block;
        }
        // confirm tier: what the flagged chunks of every candidate really hold (literal atoms decided; walk flags)
        uint32_t wi = 0;
        fi = 0;
        for (size_t gi = 0; gi < e->groups.size(); gi++) {
            const DevGroup &d = e->groups[gi];
            if (!d.filtered) continue;
            const FilterArgs &f = fall[fi++];
            if (!d.confirm) continue;
            ConfirmArgs c{};
            c.data = f.data;
            c.off = f.off;
            c.n = n;
            c.total = f.total;
            c.pairs = f.pairs;
            c.pair_count = f.pair_count;
            c.pair_cap = f.pair_cap;
            c.mul = f.mul;
            c.stride = f.stride;
            c.init = f.init;
            c.ftable = f.table;
            c.c_head = (const uint32_t *)d.c_head.p;
            c.c_entries = (const ConfirmEntry *)d.c_entries.p;
            c.c_bytes = (const uint8_t *)d.c_bytes.p;
            c.c_classes = (const uint32_t *)d.c_classes.p;
            c.n_entries = (uint32_t)d.filter.confirm.entries.size();
            c.n_bytes = (uint32_t)d.filter.confirm.bytes.size() & ~3u;
            c.n_class_words = (uint32_t)d.filter.confirm.classes.size();
            c.rec = f.rec;
            c.valid_bits = f.bitmap;
            c.dense_flag = f.dense_flag;
            c.dense_thresh = f.dense_thresh;
            c.walk_bits = (uint32_t *)S.walk_bits.p + (size_t)wi++ * bit_words;
            if (d.confirm_walk) {
                c.walk_list = f.list;
                c.walk_count = f.list_count;
            }
            c.pool = (PoolEntry *)S.pool.p;
            c.pool_count = (uint32_t *)S.ctrl.p;
            c.pool_cap = pool_cap;
            c.status = status_word;
            if (e->n_gap && e->owns_factors[gi]) {
                c.colmask_local = (const uint32_t *)e->colmask.p + d.atom_base;
                c.n_local = d.n_local;
                c.gate_lists = (uint32_t *)S.gate_lists.p;
                c.gate_count = (uint32_t *)S.ctrl.p + 2;
                c.enq_bits = (uint32_t *)S.visit_bits.p;
                c.enq_words = bit_words;
                c.shared_bits = d.shared_bits;
            }
            call.push_back(c);
        }
    }
    // list-driven DFA passes: first those behind a prefilter (they may feed the gap passes' lists), then the gap passes
    for (int phase = 0; phase < 2; phase++)
        for (size_t gi = 0; gi < e->groups.size(); gi++) {
            const DevGroup &d = e->groups[gi];
            if (d.identity ? phase != 0 : (d.gate < 0 || d.filtered != (phase == 0))) continue;
            if (phase == 0 && d.confirm && dense_flag_of[gi] != nullptr) {
                // the pass's dense alternative: EVERY request through the full table (the whole-pass walk of PWAF_OPT_NO_CONFIRM, over the
                // identity list), records and valid bits written for all; gets work only when the device set the pass's flag
                ListScanArgs a = list_args(gi, lshapes[phase], /*full_table=*/true);
                a.req_list = nullptr;
                a.n_list = nullptr;
                a.merge_rec = 0;
                a.visited = (uint32_t *)S.cand_bits.p + (size_t)filter_index[gi] * bit_words;
                a.dense_flag = dense_flag_of[gi];
                a.dense_thresh = dense_thresh_of[gi];
                a.dense_mode = 1;
                la[phase].push_back(a);
            }
            if (d.confirm && !d.confirm_walk) continue;  // every atom of the pass is a literal the confirm tier decided: nothing to walk
#ifdef PWAF_PROFILING
            static const bool skip_identity = getenv("PWAF_SKIP_IDENTITY") != nullptr;  // timing experiment (wrong results)
            if (skip_identity && d.identity) continue;
#endif
            ListScanArgs a = list_args(gi, lshapes[phase]);
            if (phase == 0 && d.confirm && dense_flag_of[gi] != nullptr) {  // the R-tier walk over the confirm tier's walk list: idle when the pass is walked whole
                a.dense_flag = dense_flag_of[gi];
                a.dense_thresh = dense_thresh_of[gi];
                a.dense_mode = 2;
            } else if (d.share_owner >= 0 && dense_flag_of[(size_t)d.share_owner] != nullptr) {  // a gap pass riding the owner's list through need masks
                a.dense_flag = dense_flag_of[(size_t)d.share_owner];
                a.dense_thresh = dense_thresh_of[(size_t)d.share_owner];
                a.dense_mode = 3;
            }
            la[phase].push_back(a);
        }
    ColPtrChunk ptrs{};  // residual kernel: the batch's string columns as arrays of pointers
    if (P.n_residual) {
        const size_t nc = e->n_fields;
        ptrs.count = (uint32_t)(2 * nc);
        for (size_t f = 0; f < nc; f++) { ptrs.p[f] = cols[f].data; ptrs.p[nc + f] = cols[f].offsets; }
    }
    // device layout of `args`: [all filtered passes] [stride-1 passes] [stride-2 passes] [confirm passes] [list passes of phase 0] [of
    // phase 1] [column pointers] — the uploaded part — then the work-item plans the plan kernels write (confirm: count + 2 words; list
    // scans: per phase 2 * count + 1)
    const uint32_t nf = (uint32_t)fall.size(), nc_conf = (uint32_t)call.size();
    const FilterArgs *d_all = nullptr, *d_s1 = nullptr, *d_s2 = nullptr;
    const ConfirmArgs *d_c = nullptr;
    const ListScanArgs *d_la[2] = {nullptr, nullptr};
    uint32_t *c_plan = nullptr, *l_plan = nullptr;
    {
        size_t ab = 0;
        auto part = [&](size_t bytes) { const size_t at = ab; ab = (ab + bytes + 255) & ~(size_t)255; return at; };
        const size_t a_all = part((size_t)nf * sizeof(FilterArgs)), a_s1 = part(by_stride[0].size() * sizeof(FilterArgs)), a_s2 = part(by_stride[1].size() * sizeof(FilterArgs)),
                     a_c = part((size_t)nc_conf * sizeof(ConfirmArgs)), a_l0 = part((la[0].size() + 1) * sizeof(ListScanArgs)), a_l1 = part((la[1].size() + 1) * sizeof(ListScanArgs)),
                     a_ptrs = part((size_t)ptrs.count * sizeof(void *));
        const size_t up_bytes = ab;
        const size_t a_cplan = part(((size_t)nc_conf + 2) * 4), a_lplan = part((2 * (la[0].size() + la[1].size()) + 4) * 4);
        if ((rc = S.args.reserve(ab))) return rc;
        char *const abase = (char *)S.args.p;
        d_all = (const FilterArgs *)(abase + a_all);
        d_s1 = (const FilterArgs *)(abase + a_s1);
        d_s2 = (const FilterArgs *)(abase + a_s2);
        d_c = (const ConfirmArgs *)(abase + a_c);
        d_la[0] = (const ListScanArgs *)(abase + a_l0);
        d_la[1] = (const ListScanArgs *)(abase + a_l1);
        S.res_cols.p = abase + a_ptrs;
        c_plan = (uint32_t *)(abase + a_cplan);
        l_plan = (uint32_t *)(abase + a_lplan);
        if (up_bytes) {
            const uint32_t slot = S.arg_next++ % Scratch::kArgSlots;
            if (S.arg_pending[slot]) HIP_TRY(hipEventSynchronize(S.arg_ev[slot]));  // (the copy launch that read this slot last has run: returns at once unless the caller is kArgSlots batches ahead)
            if ((rc = S.arg_slot[slot].reserve(up_bytes))) return rc;
            char *const hb = (char *)S.arg_slot[slot].p;
            auto put = [&](size_t at, const void *src, size_t bytes) { if (bytes) memcpy(hb + at, src, bytes); };
            put(a_all, fall.data(), (size_t)nf * sizeof(FilterArgs));
            put(a_s1, by_stride[0].data(), by_stride[0].size() * sizeof(FilterArgs));
            put(a_s2, by_stride[1].data(), by_stride[1].size() * sizeof(FilterArgs));
            put(a_c, call.data(), (size_t)nc_conf * sizeof(ConfirmArgs));
            put(a_l0, la[0].data(), la[0].size() * sizeof(ListScanArgs));
            put(a_l1, la[1].data(), la[1].size() * sizeof(ListScanArgs));
            put(a_ptrs, ptrs.p, (size_t)ptrs.count * sizeof(void *));
            int he = upload_args_block(hb, S.args.p, up_bytes, zero_deferred ? S.zero_block.p : nullptr, zero_deferred ? zero_bytes : 0, stream);
            if (he) return fail(PWAF_E_DEVICE, std::string("descriptor upload failed: ") + hipGetErrorString((hipError_t)he));
            zero_deferred = false;
            arg_slot_used = (int)slot;  // (its event is recorded at the END of the batch: a record between two kernels keeps the second from starting while the first drains)
        }
        if (zero_deferred) {  // (nothing to upload: the block is cleared the old way)
            HIP_TRY(hipMemsetAsync(S.zero_block.p, 0, zero_bytes, stream));
            zero_deferred = false;
        }
    }
    // ---- 2a. bigram prefilters of every filtered pass in one launch (the arenas as flat byte streams), hit segments -> candidate
    //          bitmaps / pair lists ----
    if (e->n_filtered) {
        int he;
        stream_mark = true;
        if ((rc = mark(nullptr, 0))) return rc;
        he = launch_filter(by_stride[0].data(), (uint32_t)by_stride[0].size(), d_s1, by_stride[1].data(), (uint32_t)by_stride[1].size(), d_s2, stream);
        if (he) return fail(PWAF_E_DEVICE, std::string("filter kernel launch failed: ") + hipGetErrorString((hipError_t)he));
        // algorithmic bytes: every streamed arena once + its offsets. The mark's name tells the bench which strides the launch mixed.
        if ((rc = mark(by_stride[1].empty() ? "filter_s1" : by_stride[0].empty() ? "filter_s2" : "filter_mix", alg_bytes[1] + alg_bytes[2]))) return rc;
        stream_mark = false;
#ifdef PWAF_PROFILING
        static const bool attr_after_compact = getenv("PWAF_ATTR_AFTER_COMPACT") != nullptr;  // timing experiment
        if (!attr_after_compact)
#endif
        if ((rc = launch_attr_side())) return rc;
        if ((rc = mark(nullptr, 0))) return rc;
        he = launch_resolve(fall.data(), nf, d_all, stream);
        bool any_list = false;  // (a pass with a confirm tier has no candidate bitmap to turn into a list: its flagged chunks are the work list)
        for (const DevGroup &d : e->groups) any_list = any_list || (d.filtered && !d.confirm);
        if (!he && any_list) he = launch_compact(fall.data(), nf, d_all, stream);
        if (he) return fail(PWAF_E_DEVICE, std::string("resolve / compact kernel launch failed: ") + hipGetErrorString((hipError_t)he));
        if ((rc = mark("resolve+compact", 0xFCu))) return rc;
        // ---- 2b. confirm tier ----
        if (!call.empty()) {
            if ((rc = mark(nullptr, 0))) return rc;
            he = launch_confirm(call.data(), nc_conf, d_c, c_plan, e->n_cus, stream);
            if (he) return fail(PWAF_E_DEVICE, std::string("confirm kernel launch failed: ") + hipGetErrorString((hipError_t)he));
            if ((rc = mark("confirm", 0xF8u))) return rc;
        }
    }
    if ((rc = launch_attr_side())) return rc;  // (no filtered pass: beside the list scans / the verdict kernel's predecessors)
    // ---- 3. list-driven DFA passes ----
    {
        uint32_t *plan_at = l_plan;  // work-item prefix sums, one set per phase
        for (int phase = 0; phase < 2; phase++) {
            const uint32_t cnt = (uint32_t)la[phase].size();
            if (!cnt) continue;
            if ((rc = mark(nullptr, 0))) return rc;
            int he = launch_scan_gated(la[phase].data(), cnt, d_la[phase], plan_at, lshapes[phase], stream);
            plan_at += 2 * cnt + 1;  // (prefix sums, then the entries per work item of every pass)
            if (he) return fail(PWAF_E_DEVICE, std::string("gated scan kernel launch failed: ") + hipGetErrorString((hipError_t)he));
            char nm[48];
            snprintf(nm, sizeof nm, "lscan_x%u", cnt);
            if ((rc = mark(nm, 0xFDu))) return rc;
        }
    }
    if (!P.fcmp.empty()) {
        // field-against-field atoms: the last (pseudo) pass; its hit records are written for every request
        FcmpArgs fa{};
        std::vector<uint32_t> slots;  // field ids in use, by slot
        for (size_t k = 0; k < P.fcmp.size(); k++) {
            uint32_t sl[2];
            const uint32_t fld[2] = {P.fcmp[k].a, P.fcmp[k].b};
            for (int q = 0; q < 2; q++) {
                size_t at = std::find(slots.begin(), slots.end(), fld[q]) - slots.begin();
                if (at == slots.size()) slots.push_back(fld[q]);
                sl[q] = (uint32_t)at;
            }
            fa.atoms[k] = P.fcmp[k].op | (sl[0] << 8) | (sl[1] << 16);
        }
        for (size_t q = 0; q < slots.size() && q < kMaxFcmpFields; q++) {
            fa.data[q] = cols[slots[q]].data;
            fa.off[q] = cols[slots[q]].offsets;
        }
        fa.n = n;
        fa.n_atoms = (uint32_t)P.fcmp.size();
        fa.rec = (uint32_t *)S.rec.p + e->groups.size() * (size_t)n;
        fa.pool = (PoolEntry *)S.pool.p;
        fa.pool_count = (uint32_t *)S.ctrl.p;
        fa.pool_cap = pool_cap;
        fa.status = status_word;
        if ((rc = mark(nullptr, 0))) return rc;
        int he = launch_fcmp(fa, stream);
        if (he) return fail(PWAF_E_DEVICE, std::string("field comparison kernel launch failed: ") + hipGetErrorString((hipError_t)he));
        if ((rc = mark("fcmp", 0xFBu))) return rc;
    }
    if (P.n_residual) {
        // rules the column compiler could not take: interpreted per request (residual.h), results = the hit records of the last pseudo pass
        ResidualArgs ra{};
        const size_t nc = e->n_fields;
        ra.data = (const uint8_t *const *)S.res_cols.p;
        ra.off = (const uint32_t *const *)((const char *)S.res_cols.p + nc * 8);
        ra.blob = (const uint8_t *)e->residual_blob.p;
        ra.n = n;
        ra.n_rules = P.n_residual;
        ra.ip = db.ip;
        ra.ip_is_v6 = db.ip_is_v6;
        ra.port = db.port;
        ra.asn = db.asn;
        ra.country = db.country;
        ra.has_geo = (P.has_geo && P.residual_needs_geo) ? 1u : 0u;
        ra.geo_root4 = (const uint32_t *)e->geo_rec_root4.p;
        ra.geo_root6 = (const uint32_t *)e->geo_rec_root6.p;
        ra.geo_nodes = (const uint32_t *)e->geo_rec_nodes.p;
        ra.geo_recs = (const GeoRec *)e->geo_recs.p;
        ra.rec = (uint32_t *)S.rec.p + (size_t)residual_pass * n;
        ra.pool = (PoolEntry *)S.pool.p;
        ra.pool_count = (uint32_t *)S.ctrl.p;
        ra.pool_cap = pool_cap;
        ra.status = status_word;
        ra.rule_errors = (unsigned long long *)e->residual_errors.p;
        if (S.retry) {  // (pwaf_engine_rule_errors counts REQUESTS: a batch run again for a larger pool must not count its errors twice)
            if ((rc = S.err_sink.reserve((size_t)P.n_residual * 8))) return rc;
            ra.rule_errors = (unsigned long long *)S.err_sink.p;
        }
        if ((rc = mark(nullptr, 0))) return rc;
        if (e->residual_jit.function) {
            // the SPECIALIZED form: the same programs as straight-line device code (compiled at creation); the verdict kernel reads the result words
            ResidualJitArgs ja{};
            ja.data = ra.data;
            ja.off = ra.off;
            ja.blob = ra.blob;
            ja.n = n;
            ja.n_rules = P.n_residual;
            ja.ip = ra.ip;
            ja.ip_is_v6 = ra.ip_is_v6;
            ja.port = ra.port;
            ja.asn = ra.asn;
            ja.country = ra.country;
            ja.has_geo = ra.has_geo;
            ja.geo_root4 = ra.geo_root4;
            ja.geo_root6 = ra.geo_root6;
            ja.geo_nodes = ra.geo_nodes;
            ja.geo_recs = ra.geo_recs;
            ja.match_words = (uint32_t *)S.res_words.p;
            ja.rule_errors = ra.rule_errors;
            int hj = launch_residual_jit(e->residual_jit, ja, e->n_cus, stream);
            if (hj) return fail(PWAF_E_DEVICE, std::string("specialized residual kernel launch failed: ") + hipGetErrorString((hipError_t)hj));
            if ((rc = mark("residual_jit", 0xF9u))) return rc;
        } else {
            int he2 = launch_residual(ra, stream);
            if (he2) return fail(PWAF_E_DEVICE, std::string("residual kernel launch failed: ") + hipGetErrorString((hipError_t)he2));
            if ((rc = mark("residual", 0xF9u))) return rc;
        }
    }
    if (placement >= 2) HIP_TRY(hipStreamWaitEvent(stream, S.ev_join, 0));  // (the attribute kernel ran on the side stream)
    if ((rc = mark(nullptr, 0))) return rc;
    int he = launch_verdict(v, stream);
    if (he) return fail(PWAF_E_DEVICE, std::string("verdict kernel launch failed: ") + hipGetErrorString((hipError_t)he));
    if ((rc = mark("verdict", 0xFFu))) return rc;
#ifdef PWAF_PROFILING
    {
        static const bool dump_counts = getenv("PWAF_DUMP_COUNTS") != nullptr;  // debugging aid: the device-side list lengths of this batch
        if (dump_counts) {
            std::vector<uint32_t> cw(ctrl_words);
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipMemcpy(cw.data(), S.ctrl.p, 4 * ctrl_words, hipMemcpyDeviceToHost));
            fprintf(stderr, "[pwaf] n %u pool %u status %u |", n, cw[0], cw[1]);
            for (size_t k = 0; k < e->groups.size(); k++) {
                const DevGroup &d = e->groups[k];
                if (d.gate >= 0) fprintf(stderr, " pass %zu field %u %s list %u", k, d.field, d.filtered ? (d.confirm ? "confirm" : "filtered") : "gap", cw[2 + (size_t)d.gate]);
            }
            for (uint32_t k = 0; k < e->n_filtered; k++) fprintf(stderr, " pairs[%u] %u", k, cw[2 + n_slots + k]);
            fprintf(stderr, "\n");
        }
    }
#endif
    if (arg_slot_used >= 0) {
        HIP_TRY(hipEventRecord(S.arg_ev[arg_slot_used], stream));  // (the copy launch that read the slot ran before everything recorded here)
        S.arg_pending[arg_slot_used] = true;
    }
    e->n_timed = ev_i;
    return PWAF_OK;
}

}  // namespace

extern "C" {

uint32_t pwaf_abi_version(void) { return PWAF_ABI_VERSION; }
const char *pwaf_last_error(void) { return g_last_error.c_str(); }

int pwaf_compile_expression(const char *expression, char *errbuf, size_t errbuf_len) {
    Syntax syn;
    std::string err;
    bool ok = expression && parse_expression(expression, syn, err);
    if (!ok) {
        std::string m = "Expression is not valid: " + (expression ? err : std::string("null expression"));
        if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", m.c_str());
        return fail(PWAF_E_SYNTAX, m);
    }
    return PWAF_OK;
}

int pwaf_validate_expression(const char *expression, char *errbuf, size_t errbuf_len) {
    auto bad = [&](const std::string &why) {
        std::string m = "Expression is not valid: " + why;
        if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", m.c_str());
        return fail(PWAF_E_SYNTAX, m);
    };
    if (!expression || !*expression) return bad("expression is empty");  // rules/rules.rs:56-58
    Syntax syn;
    std::string err;
    if (!parse_expression(expression, syn, err)) return bad(err);
    if (syn.uses_in) return bad("unknown operator: in");  // rules/rules.rs:69-71
    return PWAF_OK;
}

int pwaf_program_compile(const pwaf_rule_desc *rules, size_t n_rules, const pwaf_list_desc *lists, size_t n_lists, const pwaf_geoip_table *geoip,
                         const pwaf_options *opts, pwaf_program **out, pwaf_compile_error *err) {
    if (!out || (n_rules && !rules) || (n_lists && !lists)) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    pwaf_options o;
    int rc = check_opts(opts, o);
    if (rc) return rc;
    CompileInput in{rules, n_rules, lists, n_lists, geoip, o};
    pwaf_compile_error ce{};
    ce.rule_index = 0xFFFFFFFFu;
    std::unique_ptr<Program> p;
    try {
        rc = compile_program(in, p, ce);
    } catch (const std::bad_alloc &) {
        ce.code = rc = PWAF_E_NOMEM;
        snprintf(ce.message, sizeof ce.message, "out of memory while compiling the rule set");
    } catch (const std::exception &ex) {
        ce.code = rc = PWAF_E_UNSUPPORTED;
        snprintf(ce.message, sizeof ce.message, "internal compiler error: %s", ex.what());
    }
    if (rc) {
        put_err(err, ce);
        return rc;
    }
    auto *pp = new pwaf_program();
    pp->p = std::move(p);
    *out = pp;
    for (auto &st : pp->p->rule_status)
        if (st.first != PWAF_OK) return PWAF_W_PARTIAL;  // (PWAF_OPT_LENIENT: created, but some rule is not evaluated)
    return PWAF_OK;
}

void pwaf_program_destroy(pwaf_program *p) { delete p; }

size_t pwaf_program_dump(const pwaf_program *p, uint8_t *buf, size_t cap) {
    if (!p) return 0;
    auto *mp = const_cast<pwaf_program *>(p);
    if (mp->dump.empty()) mp->dump = dump_program(*p->p);
    if (buf && cap) memcpy(buf, mp->dump.data(), std::min(cap, mp->dump.size()));
    return mp->dump.size();
}
int pwaf_program_rule_status(const pwaf_program *p, uint32_t i, char *msg, size_t msg_len) {
    if (msg && msg_len) msg[0] = 0;
    if (!p || i >= p->p->rule_status.size()) return PWAF_E_INVALID_ARG;
    if (msg && msg_len) snprintf(msg, msg_len, "%s", p->p->rule_status[i].second.c_str());
    return p->p->rule_status[i].first;
}
size_t pwaf_program_warning_count(const pwaf_program *p) { return p ? p->p->warnings.size() : 0; }
const char *pwaf_program_warning(const pwaf_program *p, size_t i) { return (p && i < p->p->warnings.size()) ? p->p->warnings[i].c_str() : ""; }
int pwaf_program_stats(const pwaf_program *p, pwaf_stats *out) {
    if (!p || !out) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    *out = p->p->stats;
    return PWAF_OK;
}

int pwaf_engine_create(const pwaf_rule_desc *rules, size_t n_rules, const pwaf_list_desc *lists, size_t n_lists, const pwaf_geoip_table *geoip,
                       const pwaf_options *opts, pwaf_engine **out, pwaf_compile_error *err) {
    if (!out) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    pwaf_program *pp = nullptr;
    int rc = pwaf_program_compile(rules, n_rules, lists, n_lists, geoip, opts, &pp, err);
    if (rc < 0) return rc;
    const int partial = rc;  // PWAF_W_PARTIAL or PWAF_OK
    std::unique_ptr<pwaf_engine> e(new pwaf_engine());
    e->prog.p = std::move(pp->p);
    delete pp;
    const Program &P = *e->prog.p;
    auto dev_fail = [&](int code) {
        if (err) {
            err->code = code;
            err->rule_index = 0xFFFFFFFFu;
            snprintf(err->message, sizeof err->message, "%s", g_last_error.c_str());
        }
        pwaf_engine_destroy(e.release());
        return code;
    };
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev == 0) {
        fail(PWAF_E_DEVICE, std::string("no HIP device available: ") + (he != hipSuccess ? hipGetErrorString(he) : "device count is 0"));
        return dev_fail(PWAF_E_DEVICE);
    }
    int dev = opts && opts->device >= 0 ? opts->device : -1;
    if (dev >= 0) {
        if (hipSetDevice(dev) != hipSuccess) { fail(PWAF_E_DEVICE, "hipSetDevice failed"); return dev_fail(PWAF_E_DEVICE); }
    } else if (hipGetDevice(&dev) != hipSuccess) {
        fail(PWAF_E_DEVICE, "hipGetDevice failed");
        return dev_fail(PWAF_E_DEVICE);
    }
    e->device = dev;
    if (hipSetDevice(dev) != hipSuccess) { fail(PWAF_E_DEVICE, "hipSetDevice failed"); return dev_fail(PWAF_E_DEVICE); }
    if (int ke = configure_kernels(dev)) {
        fail(PWAF_E_DEVICE, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: ") + hipGetErrorString((hipError_t)ke));
        return dev_fail(PWAF_E_DEVICE);
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) e->n_cus = (uint32_t)cus;
    }
    for (size_t k = 0; k < kContexts; k++) e->ctx.emplace_back(new Scratch());
    e->n_fields = PWAF_N_FIELDS + (uint32_t)P.header_names.size();
    e->mean_len.assign(e->n_fields, 0.0);
    e->groups.resize(P.groups.size());
    std::vector<uint32_t> pass_base;
    for (size_t k = 0; k < P.groups.size(); k++) {
        if ((rc = build_device_group(P.groups[k], P.lds_hot_budget, e->groups[k]))) return dev_fail(rc);
        if ((rc = build_flat_group(P.groups[k], e->groups[k].fl, P.groups[k].filter.enabled && P.groups[k].filter_cols.empty()))) return dev_fail(rc);
        if (P.groups[k].rtier && (rc = build_flat_group(*P.groups[k].rtier, e->groups[k].rt, true))) return dev_fail(rc);
        pass_base.push_back(P.groups[k].atom_base);
    }
    if ((rc = upload(e->pass_base, pass_base))) return dev_fail(rc);
    for (size_t k = 0; k < P.groups.size(); k++) e->groups[k].filter = P.groups[k].filter;
    if (P.n_residual && !(opts && (opts->flags & PWAF_OPT_NO_RESIDUAL_JIT))) {
        // The SPECIALIZED form of the residual rules (before the pass table is built: their pseudo pass then has no records): translate,
        // compile for this device, load. Any failure leaves the interpreter in charge (same verdicts) and says so: pwaf_engine_residual_fallback.
        std::string text, why;
        std::vector<char> code;
        hipDeviceProp_t prop;
        bool ok = P.n_residual <= kMaxJitRules;
        if (!ok) why = "more than " + std::to_string(kMaxJitRules) + " residual rules";
        ok = ok && rvm_jit_program(P.residual_blob.data(), P.residual_blob.size(), text, why);
        if (ok && hipGetDeviceProperties(&prop, e->device) != hipSuccess) { ok = false; why = "hipGetDeviceProperties failed"; }
        ok = ok && rtc_compile(text, prop.gcnArchName, code, why) && jit_load(code, e->residual_jit, why);
        if (!ok) e->residual_note = "residual rules are interpreted per request, not specialized: " + why;  // (the ENGINE's: this device, this hiprtc — not the program's)
    }
    if ((rc = assign_lists(e.get()))) return dev_fail(rc);
#define UP(buf, vec)                                     \
    if ((rc = upload(e->buf, vec))) return dev_fail(rc);
    {
        // integer-set atoms: merge all sets tested against one variable into a sorted union with membership rows, so the
        // device does one binary search per request and variable instead of one per predicate
        std::vector<NumAtomDev> atoms = P.num_atoms;
        for (int var = 0; var < 2; var++) {
            std::map<int64_t, std::vector<uint32_t>> member;
            uint32_t n_sets = 0;
            for (auto &d : atoms) {
                if (d.kind != ATOM_INTSET || d.var != var) continue;
                for (uint32_t k = d.ref; k < d.ref2; k++) member[P.int_pool[k]].push_back(n_sets);
                d.ref = n_sets++;  // from now on: bit index in the variable's membership row
            }
            if (n_sets > 128) { fail(PWAF_E_UNSUPPORTED, "more than 128 integer-set predicates on one client variable"); return dev_fail(PWAF_E_UNSUPPORTED); }
            e->iu_words[var] = std::max(1u, (n_sets + 31) / 32);
            e->iu_n[var] = (uint32_t)member.size();
            std::vector<int64_t> vals;
            std::vector<uint32_t> masks(e->iu_words[var], 0);  // row 0: the value is in no set
            for (auto &kv : member) {
                vals.push_back(kv.first);
                std::vector<uint32_t> row(e->iu_words[var], 0);
                for (uint32_t b : kv.second) row[b >> 5] |= 1u << (b & 31);
                masks.insert(masks.end(), row.begin(), row.end());
            }
            if (var == 1) { e->host_iu_vals1 = vals; e->host_iu_masks1 = masks; }
            UP(iu_vals[var], vals)
            UP(iu_masks[var], masks)
        }
        // split: membership atoms become register bit tests (source word, bit); the rest are comparisons
        std::vector<NumAtomDev> cmp_src;
        std::vector<uint32_t> bit_atoms;
        if (P.n_cols >= (1u << 20)) { fail(PWAF_E_UNSUPPORTED, "more than 2^20 predicate columns"); return dev_fail(PWAF_E_UNSUPPORTED); }
        if (P.set_words > kSetWordsMax) { fail(PWAF_E_UNSUPPORTED, "more than 512 ip lists"); return dev_fail(PWAF_E_UNSUPPORTED); }
        if (P.country_luts.size() > 256) { fail(PWAF_E_UNSUPPORTED, "more than 256 distinct client.country predicates"); return dev_fail(PWAF_E_UNSUPPORTED); }
        for (auto &d : atoms) {
            uint32_t src;
            if (d.kind == ATOM_IPSET) src = d.ref >> 5;
            else if (d.kind == ATOM_COUNTRY) src = kSrcCc + (d.ref >> 5);
            else if (d.kind == ATOM_INTSET) src = (d.var == 0 ? kSrcPort : kSrcAsn) + (d.ref >> 5);
            else {
                cmp_src.push_back(d);
                continue;
            }
            bit_atoms.push_back(d.col | ((d.ref & 31u) << 20) | (src << 25));
        }
        e->n_bit_atoms = (uint32_t)bit_atoms.size();
        // (source word, bit) -> column
        std::vector<uint32_t> bit_col(kSrcWords * 32, 0);
        for (uint32_t d : bit_atoms) bit_col[(d >> 25) * 32 + ((d >> 20) & 31u)] = d & 0xFFFFFu;
        // Comparison atoms in the canonical form the kernel evaluates: variable (0-4 field lengths, 5 remote_port, 6 asn) against
        // a 32-bit constant with == or <=. Lengths, ports and ASNs are unsigned 32-bit, so constants outside [0, 2^32) fold to
        // "never" (the atom is dropped: its column stays zero) or "always" (<= 0xFFFFFFFF); `v < c` becomes `v <= c - 1`.
        struct Canon { uint32_t vi, op, col, c; };
        std::vector<Canon> canon;
        for (const NumAtomDev &d : cmp_src) {
            uint32_t vi;
            if (d.kind == ATOM_LEN && d.var >= PWAF_N_FIELDS) {
                // length of a header column: comparison variable 7 + k for the k-th such column
                size_t slot = std::find(e->hlen_fields.begin(), e->hlen_fields.end(), (uint32_t)d.var) - e->hlen_fields.begin();
                if (slot == e->hlen_fields.size()) e->hlen_fields.push_back(d.var);
                if (slot >= kMaxHeaderLens) { fail(PWAF_E_UNSUPPORTED, "length() of more than 8 distinct headers is compared"); return dev_fail(PWAF_E_UNSUPPORTED); }
                vi = 7u + (uint32_t)slot;
            } else {
                vi = d.kind == ATOM_LEN ? d.var : 5u + d.var;
                if (vi > 6) { fail(PWAF_E_UNSUPPORTED, "comparison atom on an unknown variable"); return dev_fail(PWAF_E_UNSUPPORTED); }
            }
            int64_t c = d.c;
            uint32_t op;  // 0: ==, 1: <=
            if (d.op == OP_EQ) {
                if (c < 0 || c > 0xFFFFFFFFll) continue;
                op = 0;
            } else {
                if (d.op == OP_LT) {
                    if (c <= 0) continue;  // v < c with c <= 0: never
                    c -= 1;
                }
                if (c < 0) continue;
                if (c > 0xFFFFFFFFll) c = 0xFFFFFFFFll;
                op = 1;
            }
            canon.push_back({vi, op, d.col, (uint32_t)c});
        }
        std::stable_sort(canon.begin(), canon.end(), [](const Canon &x, const Canon &y) { return x.vi != y.vi ? x.vi < y.vi : x.op < y.op; });
        if (canon.size() > 65535) { fail(PWAF_E_UNSUPPORTED, "more than 65535 comparison predicates"); return dev_fail(PWAF_E_UNSUPPORTED); }
        // POLARITY (round 6). `user_agent.length() >= 256` reaches here as NOT(length <= 255), `path.length() > 20` as NOT(length <= 20): atoms that
        // hold for nearly every request and only ever appear negated — a (column, mask) pair per group for nothing, and gate A's term
        // [NOT(length <= 255)] has no positive literal at all: the gate was a candidate rule in EVERY group. An atom whose literals are mostly
        // negations is evaluated COMPLEMENTED on the device (flag 0x80 of its code: `v > c`, `v != c`) and every literal of it toggles its
        // negation in the device copy of the literals: same truth table, rare columns, and gate A becomes a triggered rule. (client.asn
        // comparisons keep their polarity: an engine-resolved record answers them through class-row bits computed below.)
        std::vector<uint8_t> flipped(P.n_cols, 0);
        if (verdict_mode(P.flags) >= 3u && !(P.flags & PWAF_OPT_EAGER_CMP)) {
            std::vector<uint32_t> n_pos(P.n_cols, 0), n_neg(P.n_cols, 0);
            for (const uint32_t lit : P.lits) ((lit & LIT_NEG) ? n_neg : n_pos)[lit & LIT_ATOM_MASK]++;
            for (const Canon &cn : canon)
                if (cn.vi != 6u && n_neg[cn.col] > n_pos[cn.col]) flipped[cn.col] = 1;
        }
        std::vector<uint32_t> L = P.lits;  // the literals as the device evaluates them
        for (uint32_t &lit : L)
            if (flipped[lit & LIT_ATOM_MASK]) lit ^= LIT_NEG;
        std::vector<CmpAtomDev> cmp_atoms;
        for (const Canon &cn : canon) {
            cmp_atoms.push_back({cn.col | (((2 * cn.vi + cn.op) | (flipped[cn.col] ? 0x80u : 0u)) << 24), cn.c});
            if (cn.vi == 6) {
                // client.asn against a constant is a function of the GeoIP record: bit j of the class row's comparison words
                // (source words 24..27) when the engine resolves the record itself
                const uint32_t j = (uint32_t)e->host_acmp.size();
                if (j >= 128) { fail(PWAF_E_UNSUPPORTED, "more than 128 distinct client.asn comparisons"); return dev_fail(PWAF_E_UNSUPPORTED); }
                bit_col[(kSrcAcmp + j / 32) * 32 + (j & 31)] = cn.col;
                e->host_acmp.push_back({cn.op, cn.c});
            }
        }
        e->acmp_words = ((uint32_t)e->host_acmp.size() + 31) / 32;
        UP(bit_atoms, bit_col)
        // Trigger lists: a rule can match only if one of its DNF terms is true; a term with a positive literal needs that
        // column to be non-zero. Per term pick the positive literal least likely to be set (scan < membership < comparison <
        // TRUE) and file the rule under that column; terms made of negations only make the rule an unconditional candidate.
        // (rule indices travel as 16-bit values inside the verdict kernel; 0xFFF0.. is kept for the pseudo rules of the two gates)
        if (P.rules.size() > 65519 || n_rules > 65519) { fail(PWAF_E_UNSUPPORTED, "more than 65519 rules"); return dev_fail(PWAF_E_UNSUPPORTED); }
        // lower = rarer: scan atoms by how specific their pattern is (shortest possible match), then memberships, then
        // comparisons (often true for most requests), then the constant TRUE column
        std::vector<uint32_t> rank(P.n_cols, 100);
        for (auto &at : P.atoms)
            if (at.kind == ATOM_SCAN && at.id < P.n_cols) rank[at.id] = 64 - std::min<uint32_t>(at.min_len, 64);
        rank[0] = 300;
        for (auto &d : cmp_src) rank[d.col] = 200;
        for (uint32_t d : bit_atoms) rank[d & 0xFFFFFu] = 60;
        std::vector<std::vector<uint16_t>> by_col(P.n_cols);
        std::vector<uint32_t> always((P.rules.size() + 31) / 32 + 1, 0);
        std::vector<uint8_t> in_untriggered_term(P.n_cols, 0);  // the column stands in a term made of negations only: its rule is a candidate in EVERY group
        for (size_t r = 0; r < P.rules.size(); r++) {
            const DevRule &dr = P.rules[r];
            int best = -1;
            bool term_open = false;
            uint32_t term_first = dr.lit_off;
            for (uint32_t k = dr.lit_off; k < dr.lit_off + dr.lit_cnt; k++) {
                const uint32_t lit = L[k];
                if (!term_open) { best = -1; term_open = true; term_first = k; }
                if (!(lit & LIT_NEG)) {
                    const int c = (int)(lit & LIT_ATOM_MASK);
                    if (best < 0 || rank[c] < rank[best]) best = c;
                }
                if (lit & LIT_TERM_END) {
                    if (best < 0) {
                        always[r >> 5] |= 1u << (r & 31);
                        for (uint32_t q = term_first; q <= k; q++) in_untriggered_term[L[q] & LIT_ATOM_MASK] = 1;
                    } else if (by_col[best].empty() || by_col[best].back() != (uint16_t)r) by_col[best].push_back((uint16_t)r);
                    term_open = false;
                }
            }
        }
        std::vector<uint32_t> trig_off(1, 0);
        std::vector<uint16_t> trig_rules;
        for (uint32_t c = 0; c < P.n_cols; c++) {
            trig_rules.insert(trig_rules.end(), by_col[c].begin(), by_col[c].end());
            trig_off.push_back((uint32_t)trig_rules.size());
        }
        e->n_trig = (uint32_t)trig_rules.size();
        // LAZY comparison atoms (round 6; program.h: LIT_LAZY). `path.length() > 20`, `remote_port >= 1024` hold for somebody in nearly every
        // group of 64 requests, yet almost all of them only ever stand beside a rarer literal (`lit && path.length() > K`): the attribute kernel
        // spent a third of its time evaluating them for every group, and the verdict kernel filed a pair for each. An atom that is NO rule's
        // trigger (and stands in no term of negations only) is only needed once such a rule is a candidate whose other literals hold for somebody: the verdict kernel then compares the
        // 64 requests' values itself. (client.asn comparisons stay eager: engine-resolved records answer them through class-row bits.)
        std::vector<CmpAtomDev> lazy_atoms;
        std::vector<uint32_t> lazy_of(P.n_cols, 0xFFFFFFFFu);
        if (verdict_mode(P.flags) >= 3u && !(P.flags & PWAF_OPT_EAGER_CMP)) {
            // (up to two variables: the verdict kernel keeps a group's raw values of the lazy variables in two registers)
            std::vector<CmpAtomDev> eager;
            for (const CmpAtomDev &ca : cmp_atoms) {
                const uint32_t col = ca.col & 0xFFFFFFu, code = ca.col >> 24, vi = (code & 0x7Fu) / 2u;
                // (an atom of a term without a trigger — `user_agent.length() >= 256` is NOT(length <= 255): gate A — would be evaluated lazily in every group)
                bool lazy = vi != 6u && by_col[col].empty() && !in_untriggered_term[col] && ca.c <= 0xFFFFu;  // (the constant travels inside the literal word)
                uint32_t slot = 0;
                if (lazy) {
                    slot = (uint32_t)(std::find(e->lazy_vars.begin(), e->lazy_vars.end(), vi) - e->lazy_vars.begin());
                    if (slot == e->lazy_vars.size()) {
                        if (slot < 2) e->lazy_vars.push_back(vi);
                        else lazy = false;
                    }
                }
                if (lazy) {
                    lazy_of[col] = ca.c | ((code & 1u) << 16) | (slot << 17) | ((code & 0x80u) ? 1u << 18 : 0u);
                    lazy_atoms.push_back({((2u * slot + (code & 1u)) | (code & 0x80u)) << 24, ca.c});  // (kept for pwaf_engine_stats-style introspection: the kernel reads the literal word)
                } else {
                    eager.push_back(ca);
                }
            }
            cmp_atoms.swap(eager);
        }
        e->n_cmp_atoms = (uint32_t)cmp_atoms.size();
        e->cmp_vars = 0;
        for (const CmpAtomDev &ca : cmp_atoms) e->cmp_vars |= 1u << std::min(31u, ((ca.col >> 24) & 0x7Fu) / 2u);
        e->n_lazy = (uint32_t)lazy_atoms.size();
        if (lazy_atoms.empty()) lazy_atoms.push_back({0, 0});
        UP(num_atoms, cmp_atoms)
        UP(lazy_atoms, lazy_atoms)
        {
            std::vector<uint32_t> dev_lits = L;
            for (uint32_t &lit : dev_lits) {
                const uint32_t j = lazy_of[lit & LIT_ATOM_MASK];
                if (j != 0xFFFFFFFFu) lit = (lit & (LIT_NEG | LIT_TERM_END)) | LIT_LAZY | j;
            }
            UP(lits, dev_lits)
        }
        UP(trig_off, trig_off)
        UP(trig_rules, trig_rules)
        UP(always_rules, always)
        if (verdict_shape(P.n_cols, (uint32_t)P.rules.size(), e->n_trig, (uint32_t)P.lits.size(), (P.flags & PWAF_OPT_GLOBAL_VERDICT_TABLES) != 0,
                          (int)verdict_mode(P.flags), (uint32_t)P.groups.size() + 2u).waves == 0 || P.n_cols >= 65536u) {
            fail(PWAF_E_UNSUPPORTED, "too many distinct predicates for one LDS column file (160 KiB)");
            return dev_fail(PWAF_E_UNSUPPORTED);
        }
    }
    {
        // transpose the per-predicate 676-bit country tables into per-country membership words (one gather per request)
        const uint32_t n_luts = (uint32_t)P.country_luts.size();
        e->cc_words = std::max(1u, (n_luts + 31) / 32);
        std::vector<uint32_t> masks((size_t)676 * e->cc_words, 0);
        for (uint32_t t = 0; t < n_luts; t++)
            for (uint32_t c = 0; c < 676; c++)
                if (P.country_luts[t][c]) masks[(size_t)c * e->cc_words + (t >> 5)] |= 1u << (t & 31);
        e->host_cc_masks = masks;
        UP(country_luts, masks)
    }
    UP(rules, P.rules)
    UP(set_masks, P.set_masks)
    UP(ip_root4, P.ipset_trie.root4)
    UP(ip_root6, P.ipset_trie.root6)
    UP(ip_nodes, P.ipset_trie.nodes)
    {
        // GeoIP classes: everything that depends on the record — country-table bits, asn-set bits, asn comparisons — as one row per
        // record; records with equal rows share a CLASS (a few hundred for a real rule set), class 0 is the all-zero row. The
        // device trie's leaves carry class ids, so a request gathers one small cache-resident row instead of a per-record one.
        e->class_words = std::max(1u, e->cc_words + e->iu_words[1] + e->acmp_words);
        std::map<std::vector<uint32_t>, uint32_t> class_of;
        std::vector<uint32_t> rows(e->class_words, 0);  // class 0
        class_of.emplace(rows, 0u);
        std::vector<uint32_t> rec_class(P.geo_recs.size(), 0);
        for (size_t r = 0; r < P.geo_recs.size(); r++) {
            std::vector<uint32_t> row(e->class_words, 0);
            const uint32_t c0 = (P.geo_recs[r].country & 0xFFu) - 'A', c1 = (P.geo_recs[r].country >> 8) - 'A';
            const uint32_t cidx = (c0 < 26u && c1 < 26u) ? c0 * 26u + c1 : 23u * 26u + 23u;
            for (uint32_t w = 0; w < e->cc_words; w++) row[w] = e->host_cc_masks[(size_t)cidx * e->cc_words + w];
            const uint32_t asn = P.geo_recs[r].asn;
            auto it = std::lower_bound(e->host_iu_vals1.begin(), e->host_iu_vals1.end(), (int64_t)asn);
            const size_t mrow = (it != e->host_iu_vals1.end() && *it == (int64_t)asn) ? (size_t)(it - e->host_iu_vals1.begin()) + 1 : 0;
            for (uint32_t w = 0; w < e->iu_words[1]; w++) row[e->cc_words + w] = e->host_iu_masks1[mrow * e->iu_words[1] + w];
            for (size_t j = 0; j < e->host_acmp.size(); j++)
                if (e->host_acmp[j].first == 0 ? asn == e->host_acmp[j].second : asn <= e->host_acmp[j].second)
                    row[e->cc_words + e->iu_words[1] + j / 32] |= 1u << (j & 31);
            auto ins = class_of.emplace(row, (uint32_t)class_of.size());
            if (ins.second) rows.insert(rows.end(), row.begin(), row.end());
            rec_class[r] = ins.first->second;
        }
        e->n_classes = (uint32_t)class_of.size();
        e->geo_default = rec_class[0];
        UP(class_rows, rows)
        auto remap = [&](const std::vector<uint32_t> &src) {
            std::vector<uint32_t> out(src);
            for (auto &x : out)
                if (x & TRIE_LEAF) x = TRIE_LEAF | rec_class[x & ~TRIE_LEAF];
            return out;
        };
        const std::vector<uint32_t> r4 = remap(P.geo_trie.root4), r6 = remap(P.geo_trie.root6), nd = remap(P.geo_trie.nodes);
        UP(geo_root4, r4)
        UP(geo_root6, r6)
        UP(geo_nodes, nd)
        const std::vector<uint32_t> leaf(65536, TRIE_LEAF);
        UP(leaf_root, leaf)
        // (a GeoIP family without prefixes: every address reads the DEFAULT record's class — not class 0: `["XX"].contains(client.country)` or
        // `client.asn < N` hold for the default record. Round 6: an IPv6 client against a table of IPv4 prefixes read class 0 and such a rule
        // failed open; found by tests/test_gpu_paths.py: test_lazy_comparison_atoms_agree_with_eager_ones_and_the_oracle)
        const std::vector<uint32_t> geo_leaf(65536, TRIE_LEAF | e->geo_default);
        UP(geo_leaf_root, geo_leaf)
    }
    if (P.n_residual) {
        UP(residual_blob, P.residual_blob)
        {
            const std::vector<unsigned long long> zero(P.n_residual, 0ull);
            UP(residual_errors, zero)
        }
        if (P.has_geo && P.residual_needs_geo) {
            // client.asn / client.country VALUES (the class trie above only keeps which predicates hold): the trie with record leaves
            const std::vector<uint32_t> leaf0(65536, TRIE_LEAF);  // record 0 = the default {0, "XX"}
            UP(geo_rec_root4, (P.geo_trie.root4.empty() ? leaf0 : P.geo_trie.root4))
            UP(geo_rec_root6, (P.geo_trie.root6.empty() ? leaf0 : P.geo_trie.root6))
            UP(geo_rec_nodes, P.geo_trie.nodes)
            UP(geo_recs, P.geo_recs)
        }
    }
#undef UP
    if ((P.has_geo && !P.geo_trie.root4.empty()) || (P.n_ip_lists && !P.ipset_trie.root4.empty())) {
        // DIR-24-8: one 64 MiB table (2^24 x 4 B) so that an IPv4 address resolves its GeoIP class AND its ip-list membership set with
        // a single gather; built on the device from the two tries that were just uploaded (first launch counts the escapes)
        VerdictArgs tv{};
        set_trie_args(e.get(), tv);
        DevBuf cnt;
        if ((rc = cnt.reserve(4))) return dev_fail(rc);
        uint32_t n_esc = 0;
        bool ok = hipMemset(cnt.p, 0, 4) == hipSuccess && launch_dir24(tv, nullptr, nullptr, cnt.p, nullptr) == 0 &&
                  hipMemcpy(&n_esc, cnt.p, 4, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok) ok = e->dir24.reserve((size_t)4 << 24) == PWAF_OK && e->dir_esc.reserve((size_t)std::max(1u, n_esc) * 8) == PWAF_OK;
        if (ok) ok = hipMemset(cnt.p, 0, 4) == hipSuccess && launch_dir24(tv, e->dir24.p, e->dir_esc.p, cnt.p, nullptr) == 0 && hipDeviceSynchronize() == hipSuccess;
        cnt.release();
        if (!ok) { fail(PWAF_E_DEVICE, "DIR-24 table build failed"); return dev_fail(PWAF_E_DEVICE); }
        // Compressed for the lookups (round 3): 10M uniformly random addresses against the flat 64 MiB table are 10M misses to HBM (the
        // batch's own 3 GB of streaming flush every cache level in between). Consecutive /24s mostly share their entry (a /20 prefix
        // covers 16 of them), so per /16 the 256 entries are stored as RUNS, 32 /24s per 16-byte record (8 records = one 128-byte line
        // per /16): {bitmap of the /24s of this group where a run starts, the entry in force when the group begins, the entry of the
        // first run that starts inside it, where the entries of further runs live in dir_vals}. One 16-byte gather answers every lookup
        // whose /24 lies in the carried-in run or in the first run of its group (nearly all: ~12 runs per /16 over 8 groups); 8 MiB for
        // any table: L2 / Infinity-Cache resident. (Measured: the lookup kernel is bound by the NUMBER of scattered load instructions —
        // the texture addresser takes them one lane-address at a time — not by bytes or latency: four loads per lookup from one line
        // took as long as four from different lines.)
        {
            std::vector<uint32_t> d24((size_t)1 << 24);
            if (hipMemcpy(d24.data(), e->dir24.p, d24.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { fail(PWAF_E_DEVICE, "DIR-24 download failed"); return dev_fail(PWAF_E_DEVICE); }
            std::vector<uint32_t> chunks((size_t)65536 * kDirChunkWords, 0), vals;
            for (uint32_t x = 0; x < 65536; x++) {
                const uint32_t *en = &d24[(size_t)x << 8];
                for (uint32_t w = 0; w < 8; w++) {
                    uint32_t *rec = &chunks[(size_t)x * kDirChunkWords + 4 * w];
                    uint32_t bm = 0, n_starts = 0;
                    for (uint32_t j = 32 * w; j < 32 * w + 32; j++)
                        if (j == 0 || en[j] != en[j - 1]) {
                            bm |= 1u << (j & 31);
                            if (n_starts == 1) rec[3] = (uint32_t)vals.size();
                            if (n_starts == 0) rec[2] = en[j];
                            else vals.push_back(en[j]);
                            n_starts++;
                        }
                    rec[0] = bm;
                    rec[1] = w ? en[32 * w - 1] : 0u;  // (group 0 always starts a run at its first /24)
                }
            }
            if (vals.empty()) vals.push_back(0);
            if ((rc = upload(e->dir_chunks, chunks)) || (rc = upload(e->dir_vals, vals))) return dev_fail(rc);
            // The summary bitmap (round 6; kernels.h: VerdictArgs::dir_summary). Most of the address space holds ONE entry — for a WAF
            // rule set the one that says "no list holds this address and no rule asks about its GeoIP record" — so one bit per block
            // of /24s answers most lookups from a bitmap small enough to stay in every XCD's L2. Granularity: /20 ... /24 blocks,
            // whichever minimises (fraction of the space that still needs the table) + (bitmap bytes / 8 MiB); no summary when more
            // than half of the blocks need the table anyway. Skipped with PWAF_OPT_NO_DIR_SUMMARY (A/B runs and tests: same results).
            if (!(P.flags & PWAF_OPT_NO_DIR_SUMMARY)) {
                std::unordered_map<uint32_t, uint64_t> run_len;
                for (size_t x = 0; x < d24.size();) {
                    size_t y = x + 1;
                    while (y < d24.size() && d24[y] == d24[x]) y++;
                    run_len[d24[x]] += y - x;
                    x = y;
                }
                uint32_t common = 0;
                uint64_t best_len = 0;
                for (const auto &kv : run_len)
                    if (kv.second > best_len || (kv.second == best_len && kv.first < common)) { common = kv.first; best_len = kv.second; }
                double best_cost = 1e9;
                uint32_t best_shift = 0;
                std::vector<uint32_t> best_bits;
                for (uint32_t shift = 0; shift <= 4; shift++) {
                    const size_t n_blk = d24.size() >> shift;
                    std::vector<uint32_t> bits(n_blk / 32, 0);
                    uint64_t set = 0;
                    for (size_t b = 0; b < n_blk; b++) {
                        bool other = false;
                        for (size_t j = b << shift; j < ((b + 1) << shift) && !other; j++) other = d24[j] != common;
                        if (other) { bits[b >> 5] |= 1u << (b & 31); set++; }
                    }
                    const double cost = (double)set / (double)n_blk + (double)(n_blk / 8) / (8.0 * 1024 * 1024);
                    if (cost < best_cost && set * 2 <= n_blk) { best_cost = cost; best_shift = shift; best_bits.swap(bits); }
                }
                if (!best_bits.empty()) {
                    if ((rc = upload(e->dir_summary, best_bits))) return dev_fail(rc);
                    e->dir_sum_shift = best_shift;
                    e->dir_common = common;
                }
            }
            e->dir24.release();
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) { fail(PWAF_E_DEVICE, "device synchronize failed after table upload"); return dev_fail(PWAF_E_DEVICE); }
    *out = e.release();
    return partial;
}

void pwaf_engine_destroy(pwaf_engine *e) {
    if (!e) return;
    for (auto &g : e->groups) { for (DevBuf *b : {&g.tab, &g.classmap, &g.special, &g.list_off, &g.list, &g.ftable, &g.c_head, &g.c_entries, &g.c_bytes, &g.c_classes}) b->release(); g.fl.release(); g.rt.release(); }
    for (DevBuf *b : {&e->num_atoms, &e->lazy_atoms, &e->bit_atoms, &e->trig_off, &e->trig_rules, &e->always_rules, &e->iu_vals[0], &e->iu_vals[1], &e->iu_masks[0], &e->iu_masks[1], &e->country_luts, &e->rules, &e->lits,
                      &e->set_masks, &e->ip_root4, &e->ip_root6, &e->ip_nodes, &e->geo_root4, &e->geo_root6, &e->geo_nodes, &e->geo_recs, &e->residual_blob, &e->residual_errors, &e->geo_rec_root4, &e->geo_rec_root6, &e->geo_rec_nodes, &e->pass_base, &e->colmask, &e->dir24, &e->dir_chunks, &e->dir_vals, &e->dir_summary, &e->class_rows,
                      &e->dir_esc, &e->leaf_root, &e->geo_leaf_root, &e->pass_table})
        b->release();
    if (e->residual_jit.module) {  // (a module belongs to the device it was loaded on)
        int cur = -1;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        (void)hipSetDevice(e->device);
        jit_release(e->residual_jit);
        if (have) (void)hipSetDevice(cur);
    }
    for (auto &c : e->ctx) c->release();
    for (auto ev : e->ev) (void)hipEventDestroy(ev);
    delete e;
}

int pwaf_engine_residual_mode(const pwaf_engine *e) {
    if (!e || e->prog.p->n_residual == 0) return 0;
    return e->residual_jit.function ? 2 : 1;
}
const char *pwaf_engine_residual_fallback(const pwaf_engine *e) { return e ? e->residual_note.c_str() : ""; }
size_t pwaf_program_residual_source(const pwaf_program *p, int kind, char *buf, size_t cap) {
    if (!p || p->p->n_residual == 0) return 0;
    std::string text, why;
    const std::vector<uint8_t> &blob = p->p->residual_blob;
    if (!(kind == 0 ? rvm_specialize(blob.data(), blob.size(), text, why) : rvm_jit_program(blob.data(), blob.size(), text, why))) {
        (void)fail(PWAF_E_UNSUPPORTED, why);
        return 0;
    }
    if (buf && cap) {
        const size_t k = std::min(cap - 1, text.size());
        memcpy(buf, text.data(), k);
        buf[k] = '\0';
    }
    return text.size();
}
long pwaf_program_residual_compile(const pwaf_program *p, const char *arch, char *err, size_t err_len) {
    if (err && err_len) err[0] = '\0';
    if (!p || !arch) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    if (p->p->n_residual == 0) return 0;
    std::string text, why;
    std::vector<char> code;
    const std::vector<uint8_t> &blob = p->p->residual_blob;
    if (!rvm_jit_program(blob.data(), blob.size(), text, why) || !rtc_compile(text, arch, code, why)) {
        if (err && err_len) snprintf(err, err_len, "%s", why.c_str());
        return fail(PWAF_E_UNSUPPORTED, why);
    }
    return (long)code.size();
}

const pwaf_program *pwaf_engine_program(const pwaf_engine *e) { return e ? &e->prog : nullptr; }
uint32_t pwaf_program_header_count(const pwaf_program *p) { return p ? (uint32_t)p->p->header_names.size() : 0u; }
const char *pwaf_program_header_name(const pwaf_program *p, uint32_t i) { return (p && i < p->p->header_names.size()) ? p->p->header_names[i].c_str() : ""; }
uint32_t pwaf_engine_header_count(const pwaf_engine *e) { return e ? pwaf_program_header_count(&e->prog) : 0u; }
const char *pwaf_engine_header_name(const pwaf_engine *e, uint32_t i) { return e ? pwaf_program_header_name(&e->prog, i) : ""; }
void *pwaf_engine_stream(const pwaf_engine *e) {
    if (!e || e->ctx.empty() || hipSetDevice(e->device) != hipSuccess) return nullptr;
    Scratch &S = *e->ctx[0];
    std::lock_guard<std::mutex> lock(S.mu);
    return S.ensure(true) == PWAF_OK ? (void *)S.stream : nullptr;
}

int pwaf_engine_stats(const pwaf_engine *e, pwaf_stats *out) {
    if (!e || !out) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    *out = e->prog.p->stats;
    // (the prefilters in use are the engine's: pwaf_engine_tune rebuilds them from traffic)
    out->n_filtered_groups = out->n_confirm_literals = 0;
    for (size_t k = 0; k < e->groups.size(); k++) {
        if (e->groups[k].filtered) out->n_filtered_groups++;
        if (e->groups[k].confirm) out->n_confirm_literals += e->prog.p->groups[k].n_confirm_literals;
    }
    return PWAF_OK;
}

int pwaf_engine_rule_errors(pwaf_engine *e, uint64_t *counts, size_t n_rules) {
    if (!e || (!counts && n_rules)) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    const Program &P = *e->prog.p;
    for (size_t k = 0; k < n_rules; k++) counts[k] = 0;
    if (P.n_residual == 0) return PWAF_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());  // (the counters of every batch enqueued so far)
    std::vector<unsigned long long> dev(P.n_residual);
    HIP_TRY(hipMemcpy(dev.data(), e->residual_errors.p, dev.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < dev.size() && k < P.residual_rule.size(); k++)
        if (P.residual_rule[k] < n_rules) counts[P.residual_rule[k]] += dev[k];
    return PWAF_OK;
}

int pwaf_program_confirm_field(const pwaf_program *p, uint32_t group, const uint8_t *bytes, size_t len, uint32_t arena_offset, uint16_t *atoms, size_t cap, size_t *n_atoms,
                               int *flagged, int *walk) {
    if (!p || !p->p || group >= p->p->groups.size() || (!bytes && len) || !n_atoms || !flagged || !walk) return fail(PWAF_E_INVALID_ARG, "bad argument");
    const GroupFilter &f = p->p->groups[group].filter;
    if (!f.enabled) return fail(PWAF_E_INVALID_ARG, "the pass has no prefilter");
    if (len > 0x7FFFFFF0u - arena_offset) return fail(PWAF_E_INVALID_ARG, "field too long");
    std::vector<uint8_t> arena((size_t)arena_offset + len + 2 * PWAF_ARENA_PAD, (uint8_t)'~');  // (what lies around the field must not matter)
    if (len) memcpy(arena.data() + arena_offset, bytes, len);
    std::vector<uint16_t> lits;
    bool fl = false;
    const bool wk = confirm_field_host(f, arena.data(), arena_offset, arena_offset + (uint32_t)len, lits, &fl);
    *n_atoms = std::min(cap, lits.size());
    for (size_t k = 0; k < *n_atoms; k++) atoms[k] = lits[k];
    *flagged = fl ? 1 : 0;
    *walk = wk ? 1 : 0;
    return PWAF_OK;
}

int pwaf_evaluate_device(pwaf_engine *e, const pwaf_batch *in, pwaf_verdict *out, pwaf_counts *counts, uint32_t *match_idx, uint32_t *n_matches, void *stream) {
    if (!e || !out) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    int rc = validate_batch_header(in);
    if (rc) return rc;
    if (in->memory != PWAF_MEM_DEVICE) return fail(PWAF_E_INVALID_ARG, "pwaf_evaluate_device needs a DEVICE batch");
    if ((match_idx == nullptr) != (n_matches == nullptr)) return fail(PWAF_E_INVALID_ARG, "match_idx and n_matches must be given together");
    HIP_TRY(hipSetDevice(e->device));
    // Re-entrant: the call takes the next scratch context of the ring; the caller's stream first waits (on the device) for whoever
    // used that context before, and leaves its own completion event behind. Calls on different streams therefore overlap.
    std::unique_lock<std::mutex> lock;
    bool must_wait;
    Scratch &S = acquire_context(e, false, (hipStream_t)stream, lock, must_wait);
    if ((rc = S.ensure(false))) return rc;
    if (must_wait) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, S.done, 0));
    rc = run_pipeline(e, S, *in, out, counts, match_idx, n_matches, (hipStream_t)stream);
    S.used = true;
    S.last = (hipStream_t)stream;
    S.last_own = false;
    HIP_TRY(hipEventRecord(S.done, (hipStream_t)stream));
    return rc;
}

int pwaf_evaluate_batch(pwaf_engine *e, const pwaf_batch *in, pwaf_verdict *out, pwaf_counts *counts) {
    if (!e || !out) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    int rc = validate_batch_header(in);
    if (rc) return rc;
    if (in->n == 0) {
        if (counts && in->memory == PWAF_MEM_HOST) memset(counts, 0, sizeof *counts);
        return PWAF_OK;
    }
    HIP_TRY(hipSetDevice(e->device));
    // the context is held for the whole call (its staging buffers and its stream are reused); other threads take other contexts
    std::unique_lock<std::mutex> lock;
    bool must_wait;
    Scratch &S = acquire_context(e, true, nullptr, lock, must_wait);
    if ((rc = S.ensure(true))) return rc;
    hipStream_t s = S.stream;
    if (must_wait) HIP_TRY(hipStreamWaitEvent(s, S.done, 0));
    // Runs the pipeline and waits ONCE: the status words travel back (into page-locked memory) behind the batch, together with whatever
    // `after` enqueues (the host batch's verdicts). A batch that exhausted the overflow pool is run again with a pool of the size it
    // asked for. `counts_zero`: the first attempt's counters arrive zeroed (they are part of a staged block).
    if ((rc = S.pin_status.reserve(16))) return rc;
    volatile uint32_t *const st = (volatile uint32_t *)S.pin_status.p;
    auto run_checked = [&](const pwaf_batch &db, pwaf_verdict *d_out, pwaf_counts *d_counts, bool known, const std::vector<uint32_t> *begins, const std::function<int()> &after,
                           bool counts_zero) -> int {
        for (int attempt = 0;; attempt++) {
            if (d_counts && !(counts_zero && attempt == 0)) HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof *d_counts, s));
            S.retry = attempt > 0;
            int r = run_pipeline(e, S, db, d_out, d_counts, nullptr, nullptr, s, known, begins, true);
            S.retry = false;
            S.used = true;
            S.last = s;
            S.last_own = true;
            if (r) {
                (void)hipEventRecord(S.done, s);
                (void)hipStreamSynchronize(s);  // (whatever was enqueued may still read the caller's buffers)
                return r;
            }
            HIP_TRY(hipMemcpyAsync((void *)&st[0], (uint32_t *)S.status.p + 1, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync((void *)&st[1], S.ctrl.p, 4, hipMemcpyDeviceToHost, s));  // overflow entries the batch asked for
            if (after && (r = after())) {
                (void)hipEventRecord(S.done, s);
                (void)hipStreamSynchronize(s);
                return r;
            }
            HIP_TRY(hipEventRecord(S.done, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (!st[0]) return PWAF_OK;
            HIP_TRY(hipMemsetAsync((uint32_t *)S.status.p + 1, 0, 4, s));
            if (attempt >= 2 || st[1] >= 0x7FFFFFF0u) return fail(PWAF_E_NOMEM, "scan overflow pool exhausted: verdicts of this batch are incomplete");
            S.pool_entries = (uint64_t)st[1] + st[1] / 4 + 1024;
        }
    };
    if (in->memory == PWAF_MEM_DEVICE) return run_checked(*in, out, counts, false, nullptr, nullptr, false);
    // HOST batch: stage, validate what a device cannot report (while the copies are in flight, when the caller's memory is page-locked:
    // pwaf_host_alloc / pwaf_host_register), run, copy back
    const uint32_t n = in->n;
    // host view of every string column by field id (five fields + the header columns the batch carries; the others read as "")
    const uint32_t n_hdr_in = in->headers ? std::min<uint32_t>(in->n_headers, e->n_fields - PWAF_N_FIELDS) : 0u;
    auto host_col = [&](uint32_t f) -> const pwaf_strcol * {
        if (f < PWAF_N_FIELDS) return &in->field[f];
        const uint32_t k = f - PWAF_N_FIELDS;
        return (k < n_hdr_in && in->headers[k].data && in->headers[k].offsets) ? &in->headers[k] : nullptr;
    };
    auto validate = [&]() -> int {
        for (uint32_t f = 0; f < e->n_fields; f++) {
            const pwaf_strcol *c = host_col(f);
            if (!c) continue;
            const uint32_t *o = c->offsets;
            uint32_t bad = 0;
            for (uint32_t i = 0; i < n; i++) bad |= (uint32_t)(o[i + 1] < o[i]);  // (branch-free: the loop vectorizes)
            if (bad) return fail(PWAF_E_BATCH, "field offsets are not monotone");
        }
        if (in->country) {
            uint32_t bad = 0;
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t c0 = (in->country[i] & 0xFFu) - 'A', c1 = ((uint32_t)in->country[i] >> 8) - 'A';
                bad |= (uint32_t)(c0 > 25u) | (uint32_t)(c1 > 25u);
            }
            if (bad) return fail(PWAF_E_BATCH, "country is not two letters A-Z (pingoo/geoip.rs:128-142)");
        }
        return PWAF_OK;
    };
    // the ends must be ordered before any size is computed from them (the full check follows)
    for (uint32_t f = 0; f < e->n_fields; f++) {
        const pwaf_strcol *c = host_col(f);
        if (c && c->offsets[n] < c->offsets[0]) return fail(PWAF_E_BATCH, "field offsets are not monotone");
    }
    pwaf_batch db = *in;
    db.memory = PWAF_MEM_DEVICE;
    std::vector<pwaf_strcol> hdr_cols(e->n_fields - PWAF_N_FIELDS, pwaf_strcol{nullptr, nullptr});
    std::vector<uint32_t> hdr_bytes(e->n_fields - PWAF_N_FIELDS, 0);
    std::vector<uint32_t> col_begin(e->n_fields, 0);
    auto set_col = [&](uint32_t f, const uint8_t *d_data, const uint32_t *d_off, uint32_t hi) {
        const pwaf_strcol dc{d_data, d_off};
        if (f < PWAF_N_FIELDS) {
            db.field[f] = dc;
            db.field_bytes[f] = hi;
        } else {
            hdr_cols[f - PWAF_N_FIELDS] = dc;
            hdr_bytes[f - PWAF_N_FIELDS] = hi;
        }
    };
    auto finish_headers = [&]() {
        db.headers = hdr_cols.empty() ? nullptr : hdr_cols.data();
        db.header_bytes = hdr_bytes.empty() ? nullptr : hdr_bytes.data();
        db.n_headers = (uint32_t)hdr_cols.size();
    };

    // ---- small batches: ONE packed block ----
    // [offsets + arena of every column] [ip] [v6] [port] [flags] [asn] [country] [counters] | [verdicts], each 256-byte aligned. The
    // micro-batcher's batches and pwaf_evaluate_one's single requests paid ~25 small copies from pageable memory (each a blocking staged
    // copy) and two waits per batch; now: memcpy into page-locked memory, one copy in, one out, one wait.
    static constexpr size_t kPackMax = 1u << 20;
    size_t need = 0;
    auto take = [&](size_t bytes) { const size_t at = need; need = (need + bytes + 255) & ~(size_t)255; return at; };
    bool packable = true;
    std::vector<size_t> at_off(e->n_fields, 0), at_data(e->n_fields, 0);
    for (uint32_t f = 0; f < e->n_fields && packable; f++) {
        const pwaf_strcol *c = host_col(f);
        if (!c) continue;
        if (c->offsets[0] != 0) { packable = false; break; }  // (a slab view of a larger arena keeps its positions: the column-by-column path)
        at_off[f] = take((size_t)(n + 1) * 4);
        at_data[f] = take((size_t)c->offsets[n] + PWAF_ARENA_PAD);
        if (need > kPackMax) packable = false;
    }
    const size_t at_ip = take((size_t)n * 16), at_v6 = take(n), at_port = take((size_t)n * 2), at_flags = take(n);
    const size_t at_asn = in->asn ? take((size_t)n * 4) : 0, at_cc = in->asn ? take((size_t)n * 2) : 0;
    const size_t at_counts = take(sizeof(pwaf_counts));  // (arrive zeroed with the inputs, return with the verdicts)
    const size_t in_bytes = need;
    const size_t at_out = take((size_t)n * sizeof(pwaf_verdict));
    static const bool no_pack = getenv("PWAF_NO_PACKED_STAGING") != nullptr;  // A/B: the column-by-column path for every batch
    if (packable && need <= kPackMax && !no_pack) {
        if ((rc = validate())) return rc;
        if ((rc = S.pin_in.reserve(in_bytes)) || (rc = S.pin_out.reserve(need - at_counts)) || (rc = S.packed.reserve(need))) return rc;
        uint8_t *const h = (uint8_t *)S.pin_in.p;
        const uint8_t *const d = (const uint8_t *)S.packed.p;
        for (uint32_t f = 0; f < e->n_fields; f++) {
            const pwaf_strcol *c = host_col(f);
            if (!c) continue;
            const uint32_t hi = c->offsets[n];
            memcpy(h + at_off[f], c->offsets, (size_t)(n + 1) * 4);
            if (hi) memcpy(h + at_data[f], c->data, hi);
            memset(h + at_data[f] + hi, 0, PWAF_ARENA_PAD);
            set_col(f, d + at_data[f], (const uint32_t *)(d + at_off[f]), hi);
        }
        finish_headers();
        memcpy(h + at_ip, in->ip, (size_t)n * 16);
        memcpy(h + at_v6, in->ip_is_v6, n);
        memcpy(h + at_port, in->port, (size_t)n * 2);
        memcpy(h + at_flags, in->flags, n);
        db.ip = d + at_ip;
        db.ip_is_v6 = d + at_v6;
        db.port = (const uint16_t *)(d + at_port);
        db.flags = d + at_flags;
        if (in->asn) {
            memcpy(h + at_asn, in->asn, (size_t)n * 4);
            memcpy(h + at_cc, in->country, (size_t)n * 2);
            db.asn = (const uint32_t *)(d + at_asn);
            db.country = (const uint16_t *)(d + at_cc);
        }
        memset(h + at_counts, 0, sizeof(pwaf_counts));
        HIP_TRY(hipMemcpyAsync(S.packed.p, h, in_bytes, hipMemcpyHostToDevice, s));
        pwaf_verdict *const d_out = (pwaf_verdict *)((char *)S.packed.p + at_out);
        pwaf_counts *const d_counts = (pwaf_counts *)((char *)S.packed.p + at_counts);
        const size_t back = need - at_counts;
        rc = run_checked(db, d_out, d_counts, true, &col_begin, [&]() -> int {
            HIP_TRY(hipMemcpyAsync(S.pin_out.p, d_counts, back, hipMemcpyDeviceToHost, s));
            return PWAF_OK;
        }, true);
        if (rc) return rc;
        memcpy(out, (const char *)S.pin_out.p + (at_out - at_counts), (size_t)n * sizeof(pwaf_verdict));
        if (counts) memcpy(counts, S.pin_out.p, sizeof(pwaf_counts));
        return PWAF_OK;
    }

    // ---- large batches: column by column, straight from the caller's buffers (page-locked ones are read by the copy engine while this
    //      thread goes on: pwaf_host_alloc / pwaf_host_register) ----
    if (S.stage_field_data.size() < e->n_fields) { S.stage_field_data.resize(e->n_fields); S.stage_field_off.resize(e->n_fields); }
    auto bail = [&](int code) { (void)hipStreamSynchronize(s); return code; };  // (copies in flight read the caller's buffers)
    for (uint32_t f = 0; f < e->n_fields; f++) {
        const pwaf_strcol *c = host_col(f);
        if (!c) continue;
        const uint32_t *o = c->offsets;
        const size_t lo = o[0], hi = o[n];
        col_begin[f] = (uint32_t)lo;
        // offsets are used unchanged on the device: the arena keeps its positions, but only the batch's own bytes [off[0], off[n]) travel
        // (a slab view of a larger batch — pwaf_node_evaluate_batch — does not re-send what lies before it)
        if ((rc = S.stage_field_data[f].reserve(hi + PWAF_ARENA_PAD))) return bail(rc);
        if ((rc = S.stage_field_off[f].reserve((size_t)(n + 1) * 4))) return bail(rc);
        if (hi > lo) HIP_TRY(hipMemcpyAsync((char *)S.stage_field_data[f].p + lo, c->data + lo, hi - lo, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync((char *)S.stage_field_data[f].p + hi, 0, PWAF_ARENA_PAD, s));
        HIP_TRY(hipMemcpyAsync(S.stage_field_off[f].p, o, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, s));
        set_col(f, (const uint8_t *)S.stage_field_data[f].p, (const uint32_t *)S.stage_field_off[f].p, (uint32_t)hi);
    }
    finish_headers();
    auto stage = [&](DevBuf &b, const void *src, size_t bytes, const void **dst) -> int {
        int r = b.reserve(bytes);
        if (r) return r;
        HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s));
        *dst = b.p;
        return PWAF_OK;
    };
    if ((rc = stage(S.stage_ip, in->ip, (size_t)n * 16, (const void **)&db.ip))) return bail(rc);
    if ((rc = stage(S.stage_v6, in->ip_is_v6, n, (const void **)&db.ip_is_v6))) return bail(rc);
    if ((rc = stage(S.stage_port, in->port, (size_t)n * 2, (const void **)&db.port))) return bail(rc);
    if ((rc = stage(S.stage_flags, in->flags, n, (const void **)&db.flags))) return bail(rc);
    if (in->asn) {
        if ((rc = stage(S.stage_asn, in->asn, (size_t)n * 4, (const void **)&db.asn))) return bail(rc);
        if ((rc = stage(S.stage_country, in->country, (size_t)n * 2, (const void **)&db.country))) return bail(rc);
    }
    if ((rc = S.stage_out.reserve((size_t)n * sizeof(pwaf_verdict)))) return bail(rc);
    if ((rc = S.stage_counts.reserve(sizeof(pwaf_counts)))) return bail(rc);
    if ((rc = validate())) return bail(rc);
    return run_checked(db, (pwaf_verdict *)S.stage_out.p, (pwaf_counts *)S.stage_counts.p, true, &col_begin, [&]() -> int {
        HIP_TRY(hipMemcpyAsync(out, S.stage_out.p, (size_t)n * sizeof(pwaf_verdict), hipMemcpyDeviceToHost, s));
        if (counts) HIP_TRY(hipMemcpyAsync(counts, S.stage_counts.p, sizeof(pwaf_counts), hipMemcpyDeviceToHost, s));
        return PWAF_OK;
    }, false);
}

// Page-locked host memory for batch columns: the copy engine reads it directly (no staging copy inside the runtime, the calling thread
// goes on while the copy runs). A listener that parses requests into such arenas hands pwaf_evaluate_batch its bytes at PCIe speed.
int pwaf_host_alloc(size_t bytes, void **out) {
    if (!out) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    *out = nullptr;
    hipError_t he = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable);
    if (he != hipSuccess) { *out = nullptr; return fail(PWAF_E_NOMEM, std::string("hipHostMalloc failed: ") + hipGetErrorString(he)); }
    return PWAF_OK;
}
void pwaf_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}
int pwaf_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable));
    return PWAF_OK;
}
int pwaf_host_unregister(void *p) {
    if (!p) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    HIP_TRY(hipHostUnregister(p));
    return PWAF_OK;
}

// The host half of tuning (no device involved): walks every pass over the sample and rebuilds the bigram prefilters for this traffic.
// Shared by pwaf_engine_tune (which then rebuilds the device tables) and pwaf_program_tune (host-only: the tuned filters replace the
// program's, so that the table dump shows them — CPU tests interpret tuned tables without a GPU).
namespace {
struct TuneOut {
    std::vector<std::vector<uint64_t>> visits, class_freq;
    std::vector<std::vector<uint64_t>> rvisits;  // per pass with an R tier: state visits of the sample's CONFIRMED candidates in that DFA
    std::vector<GroupFilter> filters;  // per pass
    std::vector<double> mean_len;      // per field (0 = the sample does not carry it)
    std::vector<uint32_t> chunks;      // per pass: 16-byte chunks per scan_kernel iteration
};
int tune_host(const Program &P, const pwaf_batch *sample, TuneOut &T) {
    const uint32_t n_fields = PWAF_N_FIELDS + (uint32_t)P.header_names.size();
    const uint32_t n = (uint32_t)std::min<uint64_t>(sample->n, 65536);
    // string column of a field id in the sample (header columns the sample does not carry are left untuned)
    auto sample_col = [&](uint32_t f) -> const pwaf_strcol * {
        if (f < PWAF_N_FIELDS) return &sample->field[f];
        const uint32_t k = f - PWAF_N_FIELDS;
        return (sample->headers && k < sample->n_headers && sample->headers[k].data && sample->headers[k].offsets) ? &sample->headers[k] : nullptr;
    };
    // host walk of every pass over the sample: how often each DFA state is the current state, and for how many requests each
    // pattern holds (patterns that hold for most traffic must not sit behind the bigram prefilter)
    std::vector<std::vector<uint64_t>> &visits = T.visits, &class_freq = T.class_freq;
    std::vector<std::vector<uint64_t>> atom_hits(P.groups.size());
    visits.assign(P.groups.size(), {});
    T.rvisits.assign(P.groups.size(), {});
    class_freq.assign(P.groups.size(), {});
    if (T.filters.size() != P.groups.size()) return fail(PWAF_E_INVALID_ARG, "tune: filters must be pre-filled with the filters in use");
    T.mean_len.assign(n_fields, 0.0);
    T.chunks.assign(P.groups.size(), 0u);  // 0 = the sample does not carry the pass's column: keep
    for (size_t k = 0; k < P.groups.size(); k++) {
        const DfaGroup &g = P.groups[k];
        std::vector<uint64_t> &v = visits[k];
        v.assign(g.n_states, 0);
        std::vector<uint64_t> &cf = class_freq[k];
        cf.assign(256, 0);
        std::vector<uint64_t> &ah = atom_hits[k];
        ah.assign(g.n_local, 0);
        std::vector<uint32_t> stamp(g.n_local, 0);
        const pwaf_strcol *sc = sample_col(g.field);
        if (!sc) continue;
        const uint8_t *data = sc->data;
        const uint32_t *off = sc->offsets;
        for (uint32_t i = 0; i < n; i++) {
            if (off[i + 1] < off[i]) return fail(PWAF_E_BATCH, "sample offsets are not monotonic");
            auto note = [&](const std::vector<uint32_t> &o, const std::vector<uint16_t> &l, uint32_t st) {
                for (uint32_t q = o[st]; q < o[st + 1]; q++)
                    if (stamp[l[q]] != i + 1) { stamp[l[q]] = i + 1; ah[l[q]]++; }
            };
            uint32_t s = 0;
            note(g.emit_off, g.emit_list, s);
            for (uint32_t p = off[i]; p < off[i + 1]; p++) {
                const uint32_t cl = dfa_class_at(g, data + off[i], p - off[i], off[i + 1] - off[i]);  // (scalar mode: the scalar's class at a lead byte)
                cf[cl]++;
                s = g.trans[(size_t)s * g.n_classes + cl];
                v[s]++;
                if (g.emit_off[s + 1] != g.emit_off[s]) note(g.emit_off, g.emit_list, s);
            }
            note(g.end_off, g.end_list, s);
        }
    }
    // bigram prefilters rebuilt for this traffic: window choice and bucketing use the sample's bigram distribution, heads are the
    // anchored literals the sample actually satisfies; a filter that would flag more than 40 % of the sample is dropped (the pass
    // then walks every request, as without a filter)
    if (!(P.flags & PWAF_OPT_NO_PREFILTER)) {
        std::vector<std::vector<double>> pair_prob(n_fields);
        std::vector<double> &mean_len = T.mean_len;
        for (uint32_t f = 0; f < n_fields; f++) {
            const pwaf_strcol *c = sample_col(f);
            pair_prob[f].assign(65536, 0.0);
            if (!c) continue;
            std::vector<uint64_t> cnt(65536, 0);
            uint64_t tot = 0;
            const uint8_t *data = c->data;
            const uint32_t *off = c->offsets;
            for (uint32_t i = 0; i < n; i++)
                for (uint32_t p = off[i]; p + 1 < off[i + 1]; p++) { cnt[filter_fold(data[p]) | (filter_fold(data[p + 1]) << 8)]++; tot++; }
            if (tot)
                for (uint32_t b = 0; b < 65536; b++) pair_prob[f][b] = (double)cnt[b] / (double)tot;
            mean_len[f] = (double)(off[n] - off[0]) / (double)n;
        }
        // (the device samples the bigrams of a stride-2 pass at the even bytes of the ARENA: a field's phase is its offset's parity)
        auto flagged = [&](const GroupFilter &f, const pwaf_strcol *sc, uint32_t i) {
            const uint32_t *off = sc->offsets;
            return filter_candidate_arena(f, sc->data, off[i], off[i + 1], off[n]);  // (the sample's arena carries no slack)
        };
        auto sample_rate = [&](const GroupFilter &f, const pwaf_strcol *sc) {
            uint64_t c = 0;
            for (uint32_t i = 0; i < n; i++) c += flagged(f, sc, i) ? 1u : 0u;
            return (double)c / (double)n;
        };
        // Stride 2 halves the table lookups per byte: a stride-1 pass is bound by them (LDS), a stride-2 pass by HBM. A pass takes it
        // when its factors stay selective with two to four sampled bigrams per alignment: at most two points more of the sample
        // flagged than at stride 1 (a candidate costs about ten times a filtered byte), never above 25 %. Every pass decides for
        // itself: both strides run in ONE launch with their workgroups interleaved (kernels.hip: filter_kernel).
        // PWAF_OPT_FILTER_STRIDE2 takes it wherever it can be built.
        std::vector<GroupFilter> alt(P.groups.size());
        for (size_t k = 0; k < P.groups.size(); k++) {
            const DfaGroup &g = P.groups[k];
            FilterHints h;
            h.pair_prob = pair_prob[g.field].data();
            h.atom_hits = &atom_hits[k];
            h.n_requests = n;
            h.mean_len = mean_len[g.field];
            GroupFilter &gf = T.filters[k];
            const pwaf_strcol *sc = sample_col(g.field);
            if (!sc) continue;  // (a pass on a column the sample does not carry keeps the filter it has)
            build_group_filter(P.atoms, g, &h, gf, 1);
            if (!gf.enabled) continue;
            gf.est_candidate_rate = sample_rate(gf, sc);
            GroupFilter &g2 = alt[k];
            build_group_filter(P.atoms, g, &h, g2, 2);
            const bool forced = (P.flags & PWAF_OPT_FILTER_STRIDE2) != 0;
#ifdef PWAF_PROFILING
            static const long s2_mask = getenv("PWAF_STRIDE2_FIELDS") ? strtol(getenv("PWAF_STRIDE2_FIELDS"), nullptr, 0) : -1;  // timing experiments: fields that may take stride 2
            const bool s2_allowed = ((s2_mask >> g.field) & 1) != 0;
#else
            const bool s2_allowed = true;
#endif
            // The pass takes stride 2 when that flags at most two points more of the sample than stride 1 (never above 25 %).
            if (g2.enabled) {
                g2.est_candidate_rate = sample_rate(g2, sc);
                g2.enabled = forced ? g2.est_candidate_rate <= 0.4 : (g2.est_candidate_rate <= gf.est_candidate_rate + 0.02 && g2.est_candidate_rate <= 0.25);
            }
            const double plain_rate = g2.est_candidate_rate;
            const bool plain_ok = g2.enabled;
            // The same pass with EXTENDED windows (filter.cpp, Model::best_window: a window with fewer than four sampled bigrams reaches
            // one bigram beyond its factor on either side, and short windows get buckets of their own). What surrounds a factor in
            // THIS traffic decides whether that pays, so it is measured: a pass that takes stride 2 anyway takes whichever form flags
            // less of the sample; a pass that qualifies ONLY with extended windows (the URL pass of the 1k-rule set: "../" made 34 % of
            // the sample a candidate at stride 2, 2.8 % extended, 1.75 % at stride 1) takes stride 2 only when that costs next to no
            // candidates — measured on MI355X: with the other three arenas at stride 2 the launch is bound by HBM either
            // way (0.607 ms all stride 2, 0.599 ms mixed), while the extra candidates cost the confirm tier 0.09 ms on benign traffic and
            // 2.2 ms on the hostile stream (near misses survive half the bigrams far more often).
            {
                GroupFilter g2x;
                build_group_filter(P.atoms, g, &h, g2x, 2, true);
                if (g2x.enabled) {
                    g2x.est_candidate_rate = sample_rate(g2x, sc);
                    const bool take = forced ? (g2x.est_candidate_rate <= 0.4 && (!plain_ok || g2x.est_candidate_rate < plain_rate))
                                      : plain_ok ? g2x.est_candidate_rate < plain_rate
                                                 : (g2x.est_candidate_rate <= gf.est_candidate_rate * 1.2 + 0.001 && g2x.est_candidate_rate <= 0.25);
#ifdef PWAF_PROFILING
                    if (getenv("PWAF_TUNE_DEBUG")) fprintf(stderr, "[tune] pass %zu field %d: stride 1 flags %.4f of the sample, stride 2 %.4f (%s), with extended windows %.4f (%s)\n", k, g.field, gf.est_candidate_rate, plain_rate, plain_ok ? "ok" : "no", g2x.est_candidate_rate, take ? "taken" : "not taken");
#endif
                    if (take) g2 = std::move(g2x);
                }
            }
            if (!s2_allowed) g2.enabled = false;
#ifdef PWAF_PROFILING
            if (getenv("PWAF_TUNE_DEBUG")) fprintf(stderr, "[tune] pass %zu field %d: stride 1 flags %.4f of the sample (%zu heads), stride 2 %s %.4f, mean field length %.1f%s%s\n", k, g.field, gf.est_candidate_rate, gf.heads.size(), g2.enabled ? "taken:" : "not taken:", g2.est_candidate_rate, mean_len[g.field], g2.note.empty() ? "" : " — ", g2.note.c_str());
#endif
        }
        for (size_t k = 0; k < P.groups.size(); k++) {
            const DfaGroup &g = P.groups[k];
            GroupFilter &gf = T.filters[k];
            const pwaf_strcol *sc = sample_col(g.field);
            if (!sc || !gf.enabled) continue;
            if (alt[k].enabled) gf = alt[k];
            const uint8_t *data = sc->data;
            const uint32_t *off = sc->offsets;
            if (gf.est_candidate_rate > 0.4) {
                gf.enabled = false;
                gf.heads.clear();
                gf.note = "the filter flags more than 40 % of the sample";
                continue;
            }
            // The DFA of a filtered pass only ever walks the filter's candidates, whose states (deep inside pattern prefixes) are
            // not the ones average traffic visits: its LDS-resident rows are chosen from the candidates' walks alone — with a confirm
            // tier, from the walks of the candidates in which a regex factor was CONFIRMED, through the DFA they take (the R tier).
            std::vector<uint64_t> &v = visits[k];
            std::fill(v.begin(), v.end(), 0);
            std::vector<uint64_t> &rv = T.rvisits[k];
            if (g.rtier) rv.assign(g.rtier->n_states, 0);
            std::vector<uint8_t> padded;  // (confirm.h reads a few bytes past a factor: the caller's host arena carries no slack)
            if (gf.confirm.enabled) {
                padded.assign(data, data + off[n]);
                padded.resize(padded.size() + 2 * PWAF_ARENA_PAD, 0);
            }
            std::vector<uint16_t> lits;
            for (uint32_t i = 0; i < n; i++) {
                bool walk;
                if (gf.confirm.enabled) {
                    lits.clear();
                    walk = confirm_field_host(gf, padded.data(), off[i], off[i + 1], lits);
                } else {
                    walk = flagged(gf, sc, i);
                }
                if (!walk) continue;
                uint32_t st = 0;
                for (uint32_t p = off[i]; p < off[i + 1]; p++) {
                    st = g.trans[(size_t)st * g.n_classes + dfa_class_at(g, data + off[i], p - off[i], off[i + 1] - off[i])];
                    v[st]++;
                }
                if (g.rtier) {
                    const DfaGroup &r = *g.rtier;
                    uint32_t rs = 0;
                    for (uint32_t p = off[i]; p < off[i + 1]; p++) {
                        rs = r.trans[(size_t)rs * r.n_classes + dfa_class_at(r, data + off[i], p - off[i], off[i + 1] - off[i])];
                        rv[rs]++;
                    }
                }
            }
        }
    }
    for (size_t k = 0; k < P.groups.size(); k++) {
        const pwaf_strcol *sc = sample_col(P.groups[k].field);
        if (!sc) continue;
        const uint32_t *off = sc->offsets;
        const uint64_t total = (uint64_t)(off[n] - off[0]);
        const uint32_t t4 = 80u, t2 = 48u;  // mean field length from which a lane takes 4 / 2 chunks per iteration (measured)
        T.chunks[k] = total >= (uint64_t)t4 * n ? 4u : total >= (uint64_t)t2 * n ? 2u : 1u;
    }
    return PWAF_OK;
}
}  // namespace

int pwaf_program_tune(pwaf_program *p, const pwaf_batch *sample) {
    if (!p || !p->p) return fail(PWAF_E_INVALID_ARG, "program is NULL");
    int rc = validate_batch_header(sample);
    if (rc) return rc;
    if (sample->memory != PWAF_MEM_HOST) return fail(PWAF_E_INVALID_ARG, "pwaf_program_tune needs a HOST-memory sample");
    if (sample->n == 0) return PWAF_OK;
    TuneOut T;
    for (const DfaGroup &g : p->p->groups) T.filters.push_back(g.filter);
    if ((rc = tune_host(*p->p, sample, T))) return rc;
    for (size_t k = 0; k < p->p->groups.size(); k++) p->p->groups[k].filter = T.filters[k];
    p->dump.clear();
    return PWAF_OK;
}

int pwaf_engine_tune(pwaf_engine *e, const pwaf_batch *sample) {
    if (!e) return fail(PWAF_E_INVALID_ARG, "engine is NULL");
    int rc = validate_batch_header(sample);
    if (rc) return rc;
    if (sample->memory != PWAF_MEM_HOST) return fail(PWAF_E_INVALID_ARG, "pwaf_engine_tune needs a HOST-memory sample");
    // no evaluate call may enqueue while the tables are rebuilt (calls already enqueued are waited for below)
    std::vector<std::unique_lock<std::mutex>> ctx_locks;
    for (auto &c : e->ctx) ctx_locks.emplace_back(c->mu);
    std::lock_guard<std::mutex> lock(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    const Program &P = *e->prog.p;
    if (sample->n == 0) return PWAF_OK;
    TuneOut T;
    for (const DevGroup &d : e->groups) T.filters.push_back(d.filter);
    if ((rc = tune_host(P, sample, T))) return rc;
    bool rows_only = false;
#ifdef PWAF_PROFILING
    rows_only = getenv("PWAF_TUNE_ROWS_ONLY") != nullptr;  // timing experiment: keep the prefilters, take only the sample's state visits (which rows are LDS-resident)
#endif
    for (size_t k = 0; k < P.groups.size() && !rows_only; k++) {
        e->groups[k].filter = T.filters[k];
        if (T.chunks[k]) e->groups[k].chunks = T.chunks[k];
    }
    for (uint32_t f = 0; f < e->n_fields && f < T.mean_len.size(); f++)
        if (T.mean_len[f] > 0) e->mean_len[f] = T.mean_len[f];
    HIP_TRY(hipDeviceSynchronize());  // no launch may still be reading the tables that are about to be replaced
    for (size_t k = 0; k < P.groups.size(); k++)
        if ((rc = build_device_group(P.groups[k], P.lds_hot_budget, e->groups[k], &T.visits[k], &T.class_freq[k])) ||
            (rc = build_flat_group(P.groups[k], e->groups[k].fl, e->groups[k].filter.enabled && P.groups[k].filter_cols.empty(), &T.visits[k])) ||
            (P.groups[k].rtier && (rc = build_flat_group(*P.groups[k].rtier, e->groups[k].rt, true, T.rvisits[k].empty() ? nullptr : &T.rvisits[k]))))
            return rc;
    if ((rc = assign_lists(e))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return PWAF_OK;
}

int pwaf_engine_device_status(pwaf_engine *e) {
    if (!e) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    uint32_t any = 0;
    for (auto &c : e->ctx) {
        std::lock_guard<std::mutex> lock(c->mu);
        if (!c->status.p) continue;
        uint32_t status = 0;
        HIP_TRY(hipMemcpy(&status, c->status.p, 4, hipMemcpyDeviceToHost));
        if (status) {
            HIP_TRY(hipMemset(c->status.p, 0, 4));  // sticky until reported
            uint32_t asked = 0;  // the allocator keeps counting past the cap: what the last batch on this context needed
            if (c->ctrl.p) HIP_TRY(hipMemcpy(&asked, c->ctrl.p, 4, hipMemcpyDeviceToHost));
            c->pool_entries = std::max<uint64_t>(c->pool_entries * 2, (uint64_t)asked + asked / 4 + 1024);
        }
        any |= status;
    }
    if (any) return fail(PWAF_E_NOMEM, "scan overflow pool exhausted: verdicts of a device-resident batch since the last status call are incomplete "
                                       "(evaluate it again: the pool has been grown; the synchronous entry points retry by themselves)");
    return PWAF_OK;
}

int pwaf_evaluate_one(pwaf_engine *e, const pwaf_request *r, pwaf_verdict *out) {
    if (!e || !r || !out) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    if (r->n_headers && !r->headers) return fail(PWAF_E_INVALID_ARG, "n_headers without a headers array");
    const uint32_t n_hdr = e->n_fields - PWAF_N_FIELDS;  // header columns the rule set reads
    const uint32_t n_cols = PWAF_N_FIELDS + n_hdr;
    std::vector<const char *> ptr{r->host, r->url, r->path, r->method, r->user_agent};
    std::vector<uint32_t> len{r->host_len, r->url_len, r->path_len, r->method_len, r->user_agent_len};
    for (uint32_t k = 0; k < n_hdr; k++) {
        const bool have = k < r->n_headers;
        ptr.push_back(have ? r->headers[k].data : nullptr);
        len.push_back(have ? r->headers[k].len : 0u);
    }
    std::vector<std::vector<uint8_t>> arena(n_cols);
    std::vector<uint32_t> offs(2 * (size_t)n_cols);
    std::vector<pwaf_strcol> hcols(n_hdr);
    std::vector<uint32_t> hbytes(n_hdr);
    pwaf_batch b{};
    b.struct_size = sizeof b;
    b.n = 1;
    b.memory = PWAF_MEM_HOST;
    for (uint32_t f = 0; f < n_cols; f++) {
        arena[f].assign(len[f] + PWAF_ARENA_PAD, 0);
        if (len[f]) {
            if (!ptr[f]) return fail(PWAF_E_INVALID_ARG, "NULL field with non-zero length");
            memcpy(arena[f].data(), ptr[f], len[f]);
        }
        offs[2 * f] = 0;
        offs[2 * f + 1] = len[f];
        pwaf_strcol &c = f < PWAF_N_FIELDS ? b.field[f] : hcols[f - PWAF_N_FIELDS];
        c.data = arena[f].data();
        c.offsets = &offs[2 * f];
        if (f >= PWAF_N_FIELDS) hbytes[f - PWAF_N_FIELDS] = len[f];
    }
    if (n_hdr) {
        b.n_headers = n_hdr;
        b.headers = hcols.data();
        b.header_bytes = hbytes.data();
    }
    uint16_t port = r->port, country = (uint16_t)(r->country[0] | r->country[1] << 8);
    uint8_t v6 = r->ip_is_v6, flags = r->flags;
    uint32_t asn = r->asn;
    b.ip = r->ip;
    b.ip_is_v6 = &v6;
    b.port = &port;
    b.flags = &flags;
    if (r->has_geoip) {
        b.asn = &asn;
        b.country = &country;
    }
    return pwaf_evaluate_batch(e, &b, out, nullptr);
}

int pwaf_engine_set_profiling(pwaf_engine *e, int on) {
    if (!e) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lock(e->mu);
    std::lock_guard<std::mutex> plk(e->prof_mu);
    e->profiling = on < 0 ? 0 : on > 2 ? 1 : on;
    e->times.clear();  // (re)starting a measurement window: kernel_times() reports every launch since this call
    e->time_ev.clear();
    e->n_timed = 0;
    return PWAF_OK;
}

int pwaf_engine_kernel_times(pwaf_engine *e, pwaf_kernel_time *out, int cap) {
    if (!e || !out) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lock(e->mu);
    if (e->n_timed < 2 || e->times.empty()) return 0;
    int n = 0;
    for (size_t k = 0; k < e->times.size() && k < e->time_ev.size() && n < cap; k++, n++) {
        float ms = 0;
        HIP_TRY(hipEventSynchronize(e->ev[e->time_ev[k].second]));
        HIP_TRY(hipEventElapsedTime(&ms, e->ev[e->time_ev[k].first], e->ev[e->time_ev[k].second]));
        out[n] = e->times[k];
        out[n].ms = ms;
    }
    return n;
}

// ---- host-side field derivation ---------------------------------------------------------------------
size_t pwaf_derive_path(const uint8_t *p, size_t len) {
    // get_path: uri.path().trim_end_matches('/')  (pingoo/services/http_utils.rs:114-116)
    while (len && p[len - 1] == '/') len--;
    return len;
}

static bool visible_ascii(const uint8_t *p, size_t n) {
    // HeaderValue::to_str succeeds only if every byte is '\t' or 0x20..=0x7E
    for (size_t i = 0; i < n; i++)
        if (p[i] != '\t' && (p[i] < 0x20 || p[i] > 0x7E)) return false;
    return true;
}
static void trim_span(const uint8_t *p, size_t n, bool unicode_ws, size_t *start, size_t *len) {
    auto ws = [&](uint8_t c) { return c == ' ' || c == '\t' || (unicode_ws && c >= 0x0A && c <= 0x0D); };
    size_t b = 0, e = n;
    while (b < e && ws(p[b])) b++;
    while (e > b && ws(p[e - 1])) e--;
    *start = b;
    *len = e - b;
}

void pwaf_derive_user_agent(const uint8_t *hdr, size_t len, int present, size_t *out_start, size_t *out_len) {
    // pingoo/listeners/http_listener.rs:159-165
    *out_start = 0;
    *out_len = 0;
    if (!present || !visible_ascii(hdr, len)) return;
    size_t s, l;
    trim_span(hdr, len, false, &s, &l);
    if (l > 256) return;  // heapless::String<256>::from_str fails -> unwrap_or_default() == ""
    *out_start = s;
    *out_len = l;
}

void pwaf_derive_host(const uint8_t *uri_host, size_t uri_host_len, int uri_host_present, const uint8_t *host_hdr, size_t host_hdr_len,
                      int host_hdr_present, int *out_from_header, size_t *out_start, size_t *out_len) {
    // pingoo/listeners/http_listener.rs:284-296
    *out_from_header = 0;
    *out_start = 0;
    *out_len = 0;
    size_t s, l;
    if (uri_host_present) {
        trim_span(uri_host, uri_host_len, true, &s, &l);
        if (l > 256) return;
        *out_start = s;
        *out_len = l;
        return;
    }
    if (!host_hdr_present) return;
    *out_from_header = 1;
    if (!visible_ascii(host_hdr, host_hdr_len)) return;
    trim_span(host_hdr, host_hdr_len, false, &s, &l);
    if (l > 256) return;
    *out_start = s;
    *out_len = l;
}

}  // extern "C"
