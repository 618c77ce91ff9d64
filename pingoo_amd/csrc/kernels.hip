// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the WAF batch matcher.
//
// Data model: every predicate ("atom") of the compiled rule set is a COLUMN. The verdict kernel
// handles requests in GROUPS of 64 (one wavefront); for a group a column is one 64-bit word whose bit r says
// "atom holds for request 64*g + r", so rule evaluation is bit-parallel over 64 requests per ALU op with one
// lane per RULE — instead of one interpreter walk per rule per request (pingoo/rules.rs:37-51,
// http_listener.rs:251-264).
//
//   scan_kernel     one launch per DFA group (ideally one per request field). Every lane walks ONE request's field
//                   through the multi-pattern DFA and PULLS the next request of its wave's slab when it is done
//                   (ballot + prefix popcount), so ragged field lengths do not idle lanes. Transition rows of
//                   the shallow ("hot") states live in LDS; rows of deep states are read from the L2-resident
//                   table. What a request matched is written as one 4-byte hit record (two atoms inline, more
//                   through an overflow chain): lanes never share state, no atomics on the common path.
//   verdict_kernel  per group: turns the hit records into LDS column words, derives the numeric columns (lengths,
//                   port, ASN, country table, ip-list membership via the radix trie, GeoIP via the LPM trie) with
//                   wave ballots, evaluates every rule's DNF with one lane per rule, resolves first-match-wins,
//                   writes verdicts, action counters and the compacted index list of non-Allow requests.
//
// No MFMA: this is byte/integer work bounded by LDS lookups per input byte and HBM streaming.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace pwaf {

static constexpr int kScanThreads = 1024;
static constexpr int kScanWaves = kScanThreads / 64;
static constexpr int kVerdictThreads = 256;
static constexpr int kVerdictWaves = kVerdictThreads / 64;
static constexpr uint32_t kNone = 0xFFFFFFFFu;

// Explicit address spaces: a generic pointer that may be LDS or global makes the compiler emit FLAT loads for the table
// lookups (one flat_load per input byte through the texture path instead of ds_read_u16 — measured 5x slower).
#define PWAF_LDS __attribute__((address_space(3)))
#define PWAF_GLOBAL __attribute__((address_space(1)))
typedef const PWAF_LDS uint16_t *lds_u16_ptr;
typedef const PWAF_GLOBAL uint16_t *glb_u16_ptr;
typedef const PWAF_LDS uint32_t *lds_u32_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));  // 16 bytes at any byte address (gfx950 global loads need no alignment)

uint32_t scan_lds_bytes(uint32_t n_hot, uint32_t stride) { return (((n_hot + 1) * stride * 2 + 15) & ~15u) + 1024; }


// -------------------------------------------------------------------------------------------------
// scan
// -------------------------------------------------------------------------------------------------
struct Hits {
    uint32_t a0, a1;  // local atom + 1, 0 = empty
    uint32_t ovf;     // head of the overflow chain, kNone = not overflowed
};

// The slow path is ONE out-of-line function that takes and returns the per-request hit state BY VALUE: passing `Hits` or the
// kernel arguments by reference would pin them in scratch memory, and every scratch access in the main loop is a vector
// memory operation whose s_waitcnt vmcnt(0) also drains the prefetched chunk (measured: 5 us per iteration).
struct SlowCtx {
    const uint32_t *list_off;
    const uint16_t *list;
    PoolEntry *pool;
    uint32_t *pool_count;
    uint32_t *status;
    uint32_t pool_cap;
};

__device__ __forceinline__ Hits pool_push(const SlowCtx &c, uint32_t atom, Hits h) {
    const uint32_t idx = atomicAdd(c.pool_count, 1u);
    if (idx >= c.pool_cap) {
        atomicOr(c.status, 1u);
        return h;
    }
    c.pool[idx].atom = atom;
    c.pool[idx].next = h.ovf;
    h.ovf = idx;
    return h;
}

__device__ __forceinline__ Hits record_atom(const SlowCtx &c, uint32_t atom, Hits h) {
    if (h.ovf == kNone) {
        if (h.a0 == atom + 1 || h.a1 == atom + 1) return h;
        if (h.a0 == 0) { h.a0 = atom + 1; return h; }
        if (h.a1 == 0) { h.a1 = atom + 1; return h; }
        const uint32_t x0 = h.a0 - 1, x1 = h.a1 - 1;
        h = pool_push(c, x0, h);
        h = pool_push(c, x1, h);
        return pool_push(c, atom, h);
    }
    // overflowed: the chain holds every atom of this request; de-duplicate against it (this lane is its only writer)
    for (uint32_t i = h.ovf; i != kNone;) {
        const uint32_t at = __hip_atomic_load(&c.pool[i].atom, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (at == atom) return h;
        i = __hip_atomic_load(&c.pool[i].next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return pool_push(c, atom, h);
}

__device__ __noinline__ Hits emit_list(const uint32_t *list_off, const uint16_t *list, PoolEntry *pool, uint32_t *pool_count, uint32_t *status,
                                       uint32_t pool_cap, uint32_t id, Hits h) {
    const SlowCtx c{list_off, list, pool, pool_count, status, pool_cap};
    const uint32_t b = list_off[id], e = list_off[id + 1];
    for (uint32_t k = b; k < e; k++) h = record_atom(c, list[k], h);
    return h;
}
// A finished request whose hits include prefilter factors is appended to the lists of the gated passes those factors guard
// (rare: one atomic per request and gated pass).
__device__ __noinline__ void enqueue_gated(const uint32_t *colmask_local, const PoolEntry *pool, uint32_t *gate_lists, uint32_t *gate_count, uint32_t n,
                                           uint32_t r, Hits h) {
    uint32_t need = 0;
    if (h.ovf != kNone) {
        for (uint32_t i = h.ovf; i != kNone;) {
            need |= colmask_local[__hip_atomic_load(&pool[i].atom, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)];
            i = __hip_atomic_load(&pool[i].next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        if (h.a0) need |= colmask_local[h.a0 - 1];
        if (h.a1) need |= colmask_local[h.a1 - 1];
    }
    while (need) {
        const uint32_t g = (uint32_t)__builtin_ctz(need);
        need &= need - 1;
        gate_lists[(size_t)g * n + atomicAdd(&gate_count[g], 1u)] = r;
    }
}

#define PWAF_EMIT(id) h = emit_list(a.list_off, a.list, a.pool, a.pool_count, a.status, a.pool_cap, (id), h)

template <bool INDIRECT>
__global__ __launch_bounds__(kScanThreads) void scan_kernel(ScanArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t stride2 = a.stride * 2;
    const uint32_t hot_bytes = a.n_hot * stride2;            // sentinel row starts here
    const uint32_t hot_elems = hot_bytes >> 1;
    const uint32_t tab_bytes = (hot_bytes + stride2 + 15) & ~15u;
    const PWAF_LDS unsigned char *ltab = (const PWAF_LDS unsigned char *)lds;
    // 256 x uint32: byte class (the cell index inside a row). One dword per byte value: bytes that differ by less than
    // 32 never share an LDS bank, so lower-case text (the bulk of URLs) reads it conflict-free.
    lds_u32_ptr cls2 = (lds_u32_ptr)(lds + tab_bytes);
    const PWAF_GLOBAL unsigned char *gtab = (const PWAF_GLOBAL unsigned char *)a.tab;
    const PWAF_GLOBAL unsigned char *gdata = (const PWAF_GLOBAL unsigned char *)a.data;
    const PWAF_GLOBAL uint32_t *goff = (const PWAF_GLOBAL uint32_t *)a.off;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    // stage the hot rows, the sentinel row and the byte-class map into LDS (coalesced 16 B per lane)
    for (uint32_t i = tid * 16; i < tab_bytes; i += kScanThreads * 16) {
        uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);  // sentinel cells: "special"
        if (i + 16 <= hot_bytes) v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(a.tab) + i);
        *reinterpret_cast<uint4 *>(lds + i) = v;
    }
    __syncthreads();
    if (hot_bytes & 15) {  // the last, partially covered 16-byte slot of the hot rows
        const uint32_t base = hot_bytes & ~15u;
        if (tid < (hot_bytes & 15) / 2) reinterpret_cast<uint16_t *>(lds + base)[tid] = a.tab[base / 2 + tid];
    }
    if (tid < 256) reinterpret_cast<uint32_t *>(lds + tab_bytes)[tid] = a.classmap[tid];
    __syncthreads();

    const uint32_t stay_col = a.n_classes, end_col = a.n_classes + 1;
    const uint32_t special_base = a.special_base;  // cells >= special_base index the special table
    // this wave's slab of work items: contiguous, 64-aligned so offset blocks are whole. A work item is request i, or —
    // for a gated pass — entry i of the list of requests whose prefilter fired (its length lives on the device).
    const uint32_t n_items = INDIRECT ? min(*a.n_list, a.n) : a.n;
    const uint32_t total_waves = gridDim.x * kScanWaves;
    const uint32_t per_wave = (((n_items + total_waves - 1) / total_waves) + 63) & ~63u;
    const uint32_t gw = blockIdx.x * kScanWaves + wave;
    const uint32_t w0 = min(n_items, gw * per_wave), w1 = min(n_items, w0 + per_wave);
    if (w0 >= w1) return;

    const unsigned long long lt_mask = (1ull << lane) - 1;
    const uint32_t start_emit = a.start_emit;

    uint32_t next = w0, blk = w0;
    uint32_t o_id = 0, n_id = 0;  // INDIRECT: request index of each block entry
    auto load_off = [&](uint32_t base, uint32_t &lo, uint32_t &hi, uint32_t &id) {
        uint32_t i = min(base + lane, n_items - 1);  // base + lane < n_items + 63; clamp keeps the load in bounds
        if (INDIRECT) i = a.req_list[i];
        id = i;
        lo = goff[i];
        hi = goff[i + 1];
    };
    uint32_t o_lo, o_hi, n_lo = 0, n_hi = 0, n_base = kNone;
    load_off(blk, o_lo, o_hi, o_id);
    // Software pipeline: while a lane chews on the 16 bytes in `w`, the 16 bytes it will need in the NEXT iteration are
    // already in flight in `wn` — either the next chunk of the same field or, when this is the field's last chunk, the
    // first chunk of the request the lane has just pulled (r2/p2/end2). HBM/L2 latency hides behind 16 DFA steps.
    uint32_t r = kNone, p = 0, end = 0;           // current request
    uint32_t row = 0;                             // current row as a CELL value: its uint16 index in LDS, always even
                                                  // (== hot_elems: parked on the sentinel row, the real row is cold)
    uint32_t crow = 0;                            // byte offset of the current row in the full table while cold
    uint32_t r2 = kNone, p2 = 0, end2 = 0;        // request pulled ahead
    Hits h{0, 0, kNone};
    u32x4 w = {0, 0, 0, 0}, wn = {0, 0, 0, 0};

    for (;;) {
        // offsets of the NEXT block of 64 work items: re-requested every iteration (L1 hits) instead of once per block inside a
        // branch, because a load whose result crosses a branch merge forces an immediate s_waitcnt vmcnt(0) — which would
        // also drain the chunk prefetch
        uint32_t f_lo, f_hi, f_id;
        const uint32_t f_base = blk + 64;
        load_off(f_base, f_lo, f_hi, f_id);
        // ---- 1. pull ahead: lanes on their last chunk (or idle) take the next request of the slab ----
        const bool last = r == kNone || p + 16 >= end;
        const unsigned long long want = __ballot(last && r2 == kNone);
        if (want != 0 && next < w1) {
            const uint32_t avail = min(w1 - next, blk + 64 - next);
            const uint32_t rank = (uint32_t)__builtin_popcountll(want & lt_mask);
            const bool take = last && r2 == kNone && rank < avail;
            const uint32_t j = take ? next + rank - blk : 0;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)o_lo);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)o_hi);
            uint32_t rid = next + rank;
            if (INDIRECT) rid = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)o_id);
            if (take) {
                r2 = rid;
                p2 = lo;
                end2 = hi;
            }
            next += min((uint32_t)__builtin_popcountll(want), avail);
            if (next == blk + 64 && next < w1) {
                blk += 64;
                if (n_base == blk) {  // requested during an earlier iteration for this very block
                    o_lo = n_lo;
                    o_hi = n_hi;
                    o_id = n_id;
                } else {              // two block switches in consecutive iterations (very short fields): fetch now
                    load_off(blk, o_lo, o_hi, o_id);
                }
            }
        }
        if (__ballot(r != kNone || r2 != kNone) == 0) break;
        {
            // Unconditional (no branch, so no wait is forced here): lanes with nothing to fetch read the arena's first bytes.
            // Unaligned 16-byte load; arenas carry PWAF_ARENA_PAD slack.
            const bool have = last ? (r2 != kNone && p2 < end2) : true;
            const uint32_t np = have ? (last ? p2 : p + 16) : 0u;
            wn = *reinterpret_cast<const PWAF_GLOBAL u32x4_u *>(gdata + np);
        }

        // ---- 2. 16 bytes of every active lane's field ----
        const bool act = r != kNone && p < end;
        const uint32_t cnt = act ? min(16u, end - p) : 0u;
        const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
        uint32_t c2[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t c = cls2[(wd[k >> 2] >> ((k & 3) * 8)) & 0xFFu];  // byte classes do not depend on the state
            c2[k] = (uint32_t)k < cnt ? c : stay_col;                         // past the end: the STAY column
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t prev = row;
            row = *reinterpret_cast<lds_u16_ptr>(ltab + ((prev + c2[k]) << 1));  // below special_base: the next row's cell — done
            if (row >= special_base) {
                // rare: the target row is cold and/or emits, or this lane is parked on the sentinel row (current row cold)
                uint32_t cell = row;
                row = prev;
                if ((uint32_t)k < cnt) {
                    if (prev == hot_elems) cell = *reinterpret_cast<glb_u16_ptr>(gtab + crow + (c2[k] << 1));  // the real row, from L2
                    if (cell >= special_base) {
                        const SpecialCell sp = a.special[cell - special_base];
                        if (sp.next_off < hot_bytes) row = sp.next_off >> 1;
                        else { row = hot_elems; crow = sp.next_off; }
                        if (sp.emit) PWAF_EMIT(sp.emit - 1);
                    } else {
                        row = cell;
                    }
                }
            }
        }
        p += cnt;

        // ---- 3. finished requests: end-of-field matches, the hit record, then switch to the pulled-ahead request ----
        if (r != kNone && p >= end) {
            const uint32_t e = row == hot_elems ? (uint32_t)*reinterpret_cast<glb_u16_ptr>(gtab + crow + (end_col << 1))
                                                : (uint32_t)*reinterpret_cast<lds_u16_ptr>(ltab + ((row + end_col) << 1));
            if (e) PWAF_EMIT(e - 1);
            a.rec[r] = h.ovf != kNone ? (REC_OVERFLOW | h.ovf) : (h.a0 | (h.a1 << 15));
            if (a.colmask_local != nullptr && (h.a0 | (h.ovf + 1u)) != 0) enqueue_gated(a.colmask_local, a.pool, a.gate_lists, a.gate_count, a.n, r, h);
            r = kNone;
        }
        if (r == kNone && r2 != kNone) {
            r = r2;
            p = p2;
            end = end2;
            r2 = kNone;
            row = 0;
            h = Hits{0, 0, kNone};
            if (start_emit) PWAF_EMIT(start_emit - 1);
        }
        w = wn;
        n_lo = f_lo;
        n_hi = f_hi;
        n_id = f_id;
        n_base = f_base;
    }
}

int launch_scan(const ScanArgs &a, void *stream) {
    uint32_t lds = scan_lds_bytes(a.n_hot, a.stride);
    static thread_local uint32_t configured[2] = {0, 0};
    const int variant = a.req_list != nullptr;
    if (lds > configured[variant]) {
        const void *fn = variant ? reinterpret_cast<const void *>(scan_kernel<true>) : reinterpret_cast<const void *>(scan_kernel<false>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        configured[variant] = lds;
    }
    if (a.n == 0) return 0;
    if (variant) {
        // gated pass: the list length is only known on the device; lists are short (the prefilter is rare), so a modest
        // fixed grid is enough and idle workgroups exit at once
        hipLaunchKernelGGL(scan_kernel<true>, dim3(64), dim3(kScanThreads), lds, (hipStream_t)stream, a);
        return (int)hipGetLastError();
    }
    // at least 256 requests per wave so that work-pulling has something to balance; at most two rounds of one
    // workgroup per CU: long slabs keep the pull queue busy until the very end
    uint32_t waves = (a.n + 255) / 256;
    uint32_t blocks = (waves + kScanWaves - 1) / kScanWaves;
    if (blocks > 512) blocks = 512;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(scan_kernel<false>, dim3(blocks), dim3(kScanThreads), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// verdict
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ip_byte(const uint32_t w[4], uint32_t k) {
    uint32_t word = k < 4 ? w[0] : k < 8 ? w[1] : k < 12 ? w[2] : w[3];
    return (word >> ((k & 3) * 8)) & 0xFFu;
}

// 16-bit root, then 8-bit strides. Returns the leaf value (0 when the family has no table).
__device__ __forceinline__ uint32_t trie_lookup(const uint32_t *root4, const uint32_t *root6, const uint32_t *nodes, const uint32_t w[4], bool v6) {
    const uint32_t *root = v6 ? root6 : root4;
    if (root == nullptr) return 0;
    uint32_t e = root[(ip_byte(w, 0) << 8) | ip_byte(w, 1)];
    uint32_t k = 2;
    while (!(e & TRIE_LEAF)) {
        e = nodes[(size_t)e * 256 + ip_byte(w, k)];
        k++;
    }
    return e & ~TRIE_LEAF;
}

__device__ __forceinline__ bool cmp_u32(uint32_t v, uint32_t op, uint32_t c) {
    return op == OP_EQ ? v == c : op == OP_LT ? v < c : v <= c;  // only EQ / LT / LE reach the device; operands fit 32 bits
}

// LDS per wave: the column file (one 64-request word per atom), a bitmap of non-zero columns, a bitmap of candidate rules
// and the ordered candidate list.
__host__ __device__ static inline uint32_t verdict_wave_lds(uint32_t n_cols, uint32_t n_rules) {
    const uint32_t colw = (n_cols + 31) / 32, rulew = (n_rules + 31) / 32;
    return ((n_cols * 8 + colw * 4 + rulew * 4 + n_rules * 2) + 15) & ~15u;
}

__global__ __launch_bounds__(kVerdictThreads) void verdict_kernel(VerdictArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t colw = (a.n_cols + 31) / 32, rulew = (a.n_rules + 31) / 32;
    unsigned char *mine = lds + (size_t)wave * verdict_wave_lds(a.n_cols, a.n_rules);
    unsigned long long *col = reinterpret_cast<unsigned long long *>(mine);
    uint32_t *colnz = reinterpret_cast<uint32_t *>(mine + (size_t)a.n_cols * 8);
    uint32_t *rulebm = colnz + colw;
    uint16_t *cand = reinterpret_cast<uint16_t *>(rulebm + rulew);
    const unsigned long long mybit = 1ull << lane;
    const unsigned long long lt_mask = mybit - 1;
    unsigned long long cnt_block = 0, cnt_captcha = 0, cnt_bypass = 0, cnt_allow = 0;  // wave-uniform tallies

    // a lane marks "atom c holds for my request": its bit in the column word, and the column in the non-zero bitmap
    auto set_col = [&](uint32_t c) {
        atomicOr(&col[c], mybit);
        atomicOr(&colnz[c >> 5], 1u << (c & 31));
    };

    for (uint32_t g = blockIdx.x * kVerdictWaves + wave; g < a.n_groups; g += gridDim.x * kVerdictWaves) {
        const uint32_t i = g * 64 + lane;
        const bool valid = i < a.n;
        const unsigned long long valid_mask = __ballot(valid);

        // 1. clear the column file and the bitmaps; column 0 is the constant TRUE; rules that can match with every column
        //    zero (a term made of negations only) are always candidates
        for (uint32_t k = lane; k < a.n_cols; k += 64) col[k] = 0;
        for (uint32_t k = lane; k < colw; k += 64) colnz[k] = k == 0 ? 1u : 0u;
        for (uint32_t k = lane; k < rulew; k += 64) rulebm[k] = a.always_rules[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) col[0] = ~0ull;

        // 2. scan results: each lane marks the columns its hit records name. The records of 8 passes are requested together
        //    (independent loads, one wait) before any of them is examined.
        for (uint32_t pb = 0; pb < a.n_passes && !(a.debug_skip & 1u); pb += 8) {
            uint32_t rv[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t ps = min(pb + (uint32_t)q, a.n_passes - 1);
                rv[q] = valid ? a.rec[(size_t)ps * a.n + i] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (pb + (uint32_t)q >= a.n_passes) break;
                if (__ballot(rv[q] != 0) == 0) continue;  // nobody in the group matched anything in this pass
                const uint32_t base = a.pass_base_v[pb + q];  // kernel argument: no memory round trip
                if (rv[q] & REC_OVERFLOW) {
                    for (uint32_t k = rv[q] & ~REC_OVERFLOW; k != kNone;) {
                        const PoolEntry pe = a.pool[k];
                        set_col(base + pe.atom);
                        k = pe.next;
                    }
                }
                // inline atoms: lanes that name the SAME atom (frequent atoms such as a browser User-Agent prefix are named by
                // most of the 64 requests) are folded into one column update instead of 64 serialised LDS atomics
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const uint32_t x = (rv[q] & REC_OVERFLOW) ? 0u : (half == 0 ? rv[q] & 0x7FFFu : (rv[q] >> 15) & 0x7FFFu);
                    unsigned long long todo = __ballot(x != 0);
                    while (todo) {
                        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
                        const uint32_t xa = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)__builtin_amdgcn_readfirstlane(leader));
                        const unsigned long long same = __ballot(x == xa);
                        todo &= ~same;
                        if (lane == leader) {
                            const uint32_t c = base + xa - 1;
                            atomicOr(&col[c], same);
                            atomicOr(&colnz[c >> 5], 1u << (c & 31));
                        }
                    }
                }
            }
        }

        // 3. this lane's request: lengths, address, port, GeoIP record, ip-list membership
        uint32_t len[PWAF_N_FIELDS] = {0, 0, 0, 0, 0};
        uint32_t ipw[4] = {0, 0, 0, 0};
        bool v6 = false;
        uint32_t port = 0, flags = 0, asn = 0, country = (uint32_t)'X' | ((uint32_t)'X' << 8);
        uint32_t set_id = 0, geo_rec = 0;
        if (valid && !(a.debug_skip & 2u)) {
#pragma unroll
            for (int f = 0; f < PWAF_N_FIELDS; f++) len[f] = a.off[f][i + 1] - a.off[f][i];
            const uint4 raw = *reinterpret_cast<const uint4 *>(a.ip + (size_t)i * 16);
            ipw[0] = raw.x; ipw[1] = raw.y; ipw[2] = raw.z; ipw[3] = raw.w;
            v6 = a.ip_is_v6[i] != 0;
            port = a.port[i];
            flags = a.flags[i];
            // The two radix tries (GeoIP record, ip-list membership set) are walked TOGETHER, level by level, so that their
            // dependent loads overlap instead of queueing behind each other.
            bool geo_walk = false;
            if (a.asn != nullptr) {
                asn = a.asn[i];
                country = a.country[i];
            } else if (a.has_geo) {
                // GeoipDB::lookup (pingoo/geoip.rs:73-91): loopback / multicast are "not found"
                if (!v6) {
                    const uint32_t b0 = ipw[0] & 0xFFu;
                    geo_walk = !(b0 == 127u || (b0 & 0xF0u) == 0xE0u);
                } else {
                    const bool loopback = ipw[0] == 0 && ipw[1] == 0 && ipw[2] == 0 && ipw[3] == 0x01000000u;
                    geo_walk = !(loopback || (ipw[0] & 0xFFu) == 0xFFu);
                }
            }
            const uint32_t *groot = v6 ? a.geo_root6 : a.geo_root4, *iroot = v6 ? a.ip_root6 : a.ip_root4;
            const uint32_t top = (ip_byte(ipw, 0) << 8) | ip_byte(ipw, 1);
            uint32_t eg = TRIE_LEAF, ei = TRIE_LEAF;  // leaf 0: no record / member of nothing
            if (geo_walk && groot != nullptr) eg = groot[top];
            if (a.n_ip_lists && iroot != nullptr) ei = iroot[top];
            for (uint32_t k = 2; !((eg & ei) & TRIE_LEAF); k++) {
                const uint32_t byte = ip_byte(ipw, k);
                const uint32_t ng = (eg & TRIE_LEAF) ? eg : a.geo_nodes[(size_t)eg * 256 + byte];
                const uint32_t ni = (ei & TRIE_LEAF) ? ei : a.ip_nodes[(size_t)ei * 256 + byte];
                eg = ng;
                ei = ni;
            }
            geo_rec = eg & ~TRIE_LEAF;
            set_id = ei & ~TRIE_LEAF;
        }
        const unsigned long long verified_mask = __ballot(valid && (flags & PWAF_FLAG_CAPTCHA_VERIFIED));

        // 3a. membership atoms (ip lists, country tables, integer sets). The request's membership words are gathered once;
        //     each SET BIT is one atom that holds for this request, translated to its column through bit_col. Work is
        //     proportional to the number of memberships (rare), not to the number of lists / predicates.
        if (valid && !(a.debug_skip & 2u)) {
            const bool from_row = a.asn == nullptr;  // asn / country come from the engine's own GeoIP record (or its default)
            auto mark_word = [&](uint32_t src, uint32_t word) {
                while (word) {
                    const uint32_t bit = (uint32_t)__builtin_ctz(word);
                    word &= word - 1;
                    const uint32_t c = a.bit_col[src * 32 + bit];
                    if (c) set_col(c);
                }
            };
            if (from_row) {
                // per GeoIP record the engine has precomputed everything that depends on (asn, country)
                const uint32_t *row = a.geo_rows + (size_t)geo_rec * a.geo_row_words;
                asn = row[0];
                country = row[1];
                for (uint32_t wv = 0; wv < a.cc_words; wv++) mark_word(8 + wv, row[2 + wv]);
                for (uint32_t wv = 0; wv < a.iu_words[1]; wv++) mark_word(20 + wv, row[2 + a.cc_words + wv]);
            } else {
                const uint32_t c0 = (country & 0xFFu) - 'A', c1 = (country >> 8) - 'A';
                const uint32_t cidx = (c0 < 26u && c1 < 26u) ? c0 * 26u + c1 : 23u * 26u + 23u;  // invalid input is treated as "XX"
                for (uint32_t wv = 0; wv < a.cc_words; wv++) mark_word(8 + wv, a.country_masks[(size_t)cidx * a.cc_words + wv]);
            }
#pragma unroll
            for (int var = 0; var < 2; var++) {
                if (a.iu_n[var] == 0 || (var == 1 && from_row)) continue;
                // ONE binary search per request over the union of every set tested against this variable; the hit's row
                // says which sets contain the value (the reference scans each list per rule: pingoo/lists.rs:119-121)
                const long long v = var == VAR_PORT ? (long long)port : (long long)asn;
                uint32_t lo = 0, hi = a.iu_n[var];
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (a.iu_vals[var][mid] < v) lo = mid + 1;
                    else hi = mid;
                }
                if (lo < a.iu_n[var] && a.iu_vals[var][lo] == v)
                    for (uint32_t wv = 0; wv < a.iu_words[var]; wv++) mark_word(16 + 4 * var + wv, a.iu_masks[var][(size_t)(lo + 1) * a.iu_words[var] + wv]);
            }
            if (a.n_ip_lists && set_id)
                for (uint32_t wv = 0; wv < a.set_words; wv++) mark_word(wv, a.set_masks[(size_t)set_id * a.set_words + wv]);
        }
        // 3b. comparison atoms (lengths, port, asn against constants): few. Descriptors are fetched 64 at a time, one per lane,
        //     and broadcast with v_readlane; atom j's 64-request ballot is parked in lane j, one ds_write_b64 per chunk.
        for (uint32_t base = 0; base < a.n_num_atoms && !(a.debug_skip & 4u); base += 64) {
            uint32_t m_col = 0, m_meta = 0, m_clo = 0;
            if (base + lane < a.n_num_atoms) {
                const NumAtomDev d = a.num_atoms[base + lane];
                m_col = d.col;
                m_meta = (uint32_t)d.kind | ((uint32_t)d.var << 8) | ((uint32_t)d.op << 16);
                m_clo = (uint32_t)(unsigned long long)d.c;  // the engine folds constants outside [0, 2^32) away
            }
            const uint32_t cntd = min(64u, a.n_num_atoms - base);
            uint32_t acc_lo = 0, acc_hi = 0;
            for (uint32_t j = 0; j < cntd; j++) {
                const uint32_t meta = __builtin_amdgcn_readlane(m_meta, j);
                const uint32_t kind = meta & 0xFFu, var = (meta >> 8) & 0xFFu, op = meta >> 16;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane(m_clo, j);
                uint32_t v;
                if (kind == ATOM_LEN) v = var == 0 ? len[0] : var == 1 ? len[1] : var == 2 ? len[2] : var == 3 ? len[3] : len[4];
                else v = var == VAR_PORT ? port : asn;
                const unsigned long long m = __ballot(cmp_u32(v, op, c) && valid);
                acc_lo = lane == j ? (uint32_t)m : acc_lo;
                acc_hi = lane == j ? (uint32_t)(m >> 32) : acc_hi;
            }
            if (lane < cntd && (acc_lo | acc_hi)) {
                col[m_col] = ((unsigned long long)acc_hi << 32) | acc_lo;
                atomicOr(&colnz[m_col >> 5], 1u << (m_col & 31));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        // 4. candidate rules: a rule can only match some request of this group if one of its terms has a non-zero positive
        //    column (trigger lists, one chosen literal per term) or consists of negations only (always_rules).
        for (uint32_t wv = lane; wv < colw && !(a.debug_skip & 8u); wv += 64) {
            uint32_t nz = colnz[wv];
            while (nz) {
                const uint32_t c = wv * 32 + (uint32_t)__builtin_ctz(nz);
                nz &= nz - 1;
                for (uint32_t k = a.trig_off[c]; k < a.trig_off[c + 1]; k++) {
                    const uint32_t r = a.trig_rules[k];
                    atomicOr(&rulebm[r >> 5], 1u << (r & 31));
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ordered compaction of the rule bitmap into the candidate list (ascending rule index = evaluation order)
        uint32_t n_cand = 0;
        for (uint32_t wb = 0; wb < rulew && !(a.debug_skip & 16u); wb += 64) {
            const uint32_t word = wb + lane < rulew ? rulebm[wb + lane] : 0u;
            uint32_t pc = (uint32_t)__builtin_popcount(word), incl = pc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= (uint32_t)d) incl += up;
            }
            uint32_t pos = n_cand + incl - pc, wrd = word;
            while (wrd) {
                cand[pos++] = (uint16_t)((wb + lane) * 32 + (uint32_t)__builtin_ctz(wrd));
                wrd &= wrd - 1;
            }
            n_cand += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        // 5. evaluate candidates: one lane per rule, 64 requests per ALU op; first match (lowest rule index) wins
        unsigned long long pending = valid_mask;
        uint32_t my_action = PWAF_ACTION_ALLOW, my_rule = PWAF_RULE_NONE;
        for (uint32_t base = 0; base < n_cand && pending != 0 && !(a.debug_skip & 32u); base += 64) {
            unsigned long long fire = 0;
            uint32_t eff_u = 0, eff_v = 0, pub = 0;
            if (base + lane < n_cand) {
                const DevRule dr = a.rules[cand[base + lane]];
                eff_u = dr.eff_unverified;
                eff_v = dr.eff_verified;
                pub = dr.public_idx;
                unsigned long long acc_or = 0, acc_and = ~0ull;
                for (uint32_t k = dr.lit_off; k < dr.lit_off + dr.lit_cnt; k++) {
                    const uint32_t lit = a.lits[k];
                    unsigned long long c = col[lit & LIT_ATOM_MASK];
                    if (lit & LIT_NEG) c = ~c;
                    acc_and &= c;
                    if (lit & LIT_TERM_END) {
                        acc_or |= acc_and;
                        acc_and = ~0ull;
                    }
                }
                // a match only decides when the rule's action list yields an effect for that client
                fire = acc_or & pending & ((eff_u ? ~verified_mask : 0ull) | (eff_v ? verified_mask : 0ull));
            }
            unsigned long long firing_lanes = __ballot(fire != 0);
            while (firing_lanes) {
                const uint32_t j = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_ctzll(firing_lanes));
                firing_lanes &= firing_lanes - 1;
                const uint32_t flo = __builtin_amdgcn_readlane((uint32_t)fire, j);
                const uint32_t fhi = __builtin_amdgcn_readlane((uint32_t)(fire >> 32), j);
                const unsigned long long newly = (((unsigned long long)fhi << 32) | flo) & pending;
                pending &= ~newly;
                const uint32_t ju = __builtin_amdgcn_readlane(eff_u, j), jv = __builtin_amdgcn_readlane(eff_v, j);
                const uint32_t jp = __builtin_amdgcn_readlane(pub, j);
                if (newly & mybit) {
                    my_action = (verified_mask & mybit) ? jv : ju;
                    my_rule = jp;
                }
            }
        }

        // 6. outputs
        if (valid) {
            uint2 v;
            v.x = my_action;  // action in byte 0, pad bytes zero
            v.y = my_rule;
            *reinterpret_cast<uint2 *>(&a.out[i]) = v;
        }
        const unsigned long long m_block = __ballot(valid && my_action == PWAF_ACTION_BLOCK);
        const unsigned long long m_captcha = __ballot(valid && my_action == PWAF_ACTION_CAPTCHA);
        const unsigned long long m_bypass = __ballot(valid && my_action == PWAF_ACTION_BYPASS);
        cnt_block += (unsigned)__builtin_popcountll(m_block);
        cnt_captcha += (unsigned)__builtin_popcountll(m_captcha);
        cnt_bypass += (unsigned)__builtin_popcountll(m_bypass);
        cnt_allow += (unsigned)__builtin_popcountll(valid_mask & ~(m_block | m_captcha | m_bypass));
        if (a.match_idx != nullptr) {
            // compaction of non-Allow requests: wave ballot + prefix popcount, one atomic per group
            const unsigned long long hit = m_block | m_captcha | m_bypass;
            if (hit) {
                uint32_t basei = 0;
                if (lane == 0) basei = atomicAdd(a.n_matches, (uint32_t)__builtin_popcountll(hit));
                basei = __builtin_amdgcn_readfirstlane(basei);
                if (hit & mybit) a.match_idx[basei + (uint32_t)__builtin_popcountll(hit & lt_mask)] = i;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (a.counts != nullptr && lane == 0) {
        if (cnt_allow) atomicAdd(&a.counts[PWAF_ACTION_ALLOW], cnt_allow);
        if (cnt_block) atomicAdd(&a.counts[PWAF_ACTION_BLOCK], cnt_block);
        if (cnt_captcha) atomicAdd(&a.counts[PWAF_ACTION_CAPTCHA], cnt_captcha);
        if (cnt_bypass) atomicAdd(&a.counts[PWAF_ACTION_BYPASS], cnt_bypass);
    }
}

uint32_t verdict_lds_bytes(uint32_t n_cols, uint32_t n_rules) { return kVerdictWaves * verdict_wave_lds(n_cols, n_rules); }

int launch_verdict(const VerdictArgs &a, void *stream) {
    uint32_t lds = verdict_lds_bytes(a.n_cols, a.n_rules);
    static thread_local uint32_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(verdict_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        configured = lds;
    }
    uint32_t blocks = (a.n_groups + kVerdictWaves - 1) / kVerdictWaves;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(verdict_kernel, dim3(blocks), dim3(kVerdictThreads), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // namespace pwaf
