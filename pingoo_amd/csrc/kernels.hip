// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the WAF batch matcher.
//
// Data model: every predicate ("atom") of the compiled rule set is a COLUMN. The verdict kernel
// handles requests in GROUPS of 64 (one wavefront); for a group a column is one 64-bit word whose bit r says
// "atom holds for request 64*g + r", so rule evaluation is bit-parallel over 64 requests per ALU op with one
// lane per RULE — instead of one interpreter walk per rule per request (pingoo/rules.rs:37-51,
// http_listener.rs:251-264).
//
//   filter_kernel    the launch that streams the request bytes: every string pass whose patterns all have a literal factor sits behind
//                    a bucketed shift-or BIGRAM PREFILTER. A field's arena is one flat byte stream; per byte one independent lookup of
//                    a 16 KiB LDS table and two vector ops; the requests overlapping a completed window become CANDIDATES
//                    (resolve_kernel: flagged chunks -> bitmap; bitcount / compact_kernel: bitmap -> dense ascending list).
//   lscan_kernel     every list-driven pass of a phase in one launch: the candidates of the filtered passes, then the gap passes
//                    (patterns with wide gaps, visited only by requests whose prefilter factor matched). One listed request per lane
//                    through the pass's DFA, hot rows in LDS. What a request matched is written as one 4-byte hit record (two atoms
//                    inline, more through an overflow chain): lanes never share state, no atomics on the common path.
//   scan_kernel<CH>  a pass without a usable prefilter: the round-1 design, a DFA over EVERY request (one persistent workgroup per CU,
//                    lanes pull the next request of their wave's slab when done, hot rows in LDS, cold rows from the L2-resident table).
//   fcmp_kernel      predicates between two request fields (one lane per request compares the byte ranges).
//   attr_kernel      on the engine's low-priority side stream, forked after the filter launches: everything that is not a string scan — GeoIP record and
//                    ip-list membership (DIR-24-8 table + radix tries), country / integer-set membership, length / port / asn
//                    comparisons — reduced per 64-request group to (column, 64-request mask) pairs with wave ballots.
//   verdict_kernel   per group: turns hit records and pairs into LDS column words, finds the candidate rules through trigger lists,
//                    evaluates each candidate's DNF with one lane per rule, resolves first-match-wins, writes verdicts, action
//                    counters and the compacted index list of non-Allow requests.
//
// No MFMA: this is byte/integer work bounded by LDS lookups per input byte and HBM streaming (filter_kernel runs
// at 0.47 of the HBM roofline, bound by the bank conflicts of its table gather).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <set>

#include "confirm.h"
#include "utf8.h"
#include "kernels.h"
#include "residual.h"

namespace pwaf {

static constexpr int kScanThreads = 1024;
static constexpr int kScanWaves = kScanThreads / 64;
static constexpr uint32_t kNone = 0xFFFFFFFFu;

// Explicit address spaces: a generic pointer that may be LDS or global makes the compiler emit FLAT loads for the table
// lookups (one flat_load per input byte through the texture path instead of ds_read_u16 — measured 5x slower).
#define PWAF_LDS __attribute__((address_space(3)))
#define PWAF_GLOBAL __attribute__((address_space(1)))
typedef const PWAF_LDS uint16_t *lds_u16_ptr;
typedef const PWAF_GLOBAL uint16_t *glb_u16_ptr;
typedef const PWAF_LDS uint32_t *lds_u32_ptr;
typedef const PWAF_LDS uint8_t *lds_u8_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));  // 16 bytes at any byte address (gfx950 global loads need no alignment)

static constexpr uint32_t kClsBytes = 1024;  // byte-class map at LDS offset 0: 256 x uint32
// The walk is speculative (see scan_body): a lane that has just read a special cell keeps using it as an address until the
// group of 4 steps is checked, so every address a 16-bit cell can form — (0xFFFF << 1) + a class offset — must stay inside the
// allocation.
uint32_t scan_lds_bytes(uint32_t n_hot, uint32_t stride, uint32_t n_gate_atoms) {
    const uint32_t need = (((n_hot + 1) * stride * 2 + 15) & ~15u) + kClsBytes + n_gate_atoms * 4;
    const uint32_t reach = ((0xFFFFu << 1) + stride * 2 + kClsBytes + 15) & ~15u;
    return need > reach ? need : reach;
}


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
// One careful DFA step of one lane (out of line, by value): used when the speculative group walk met an emitting row, a cold
// row or a parked lane. `t` is the cell the lane read from LDS for this step.
struct Walk {
    uint32_t row, crow;
    Hits h;
};
// (Arguments are kept few and scalar: beyond the register budget of the calling convention the compiler passes `Hits` through
// scratch memory.)
__device__ __noinline__ Walk careful_step(const PWAF_GLOBAL unsigned char *gtab, const SpecialCell *special, const uint32_t *list_off, const uint16_t *list,
                                          PoolEntry *pool, uint32_t *ctrl /* [0] pool allocator, [1] status */, uint32_t pool_cap, uint32_t hot_emit /* hot_elems | emit_base << 16 */,
                                          uint32_t special_col /* special_base | emit_col2 << 16 */, uint32_t prev, uint32_t crow, uint32_t c2k, uint32_t t, uint32_t a0,
                                          uint32_t a1, uint32_t ovf) {
    const SlowCtx c{list_off, list, pool, ctrl, ctrl + 1, pool_cap};
    const uint32_t hot_elems = hot_emit & 0xFFFFu, emit_base = hot_emit >> 16, special_base = special_col & 0xFFFFu, emit_col2 = special_col >> 16;
    const Hits h{a0, a1, ovf};
    uint32_t cell = t;
    if (prev == hot_elems) cell = *reinterpret_cast<glb_u16_ptr>(gtab + crow + c2k);  // parked: the real (cold) row, from L2
    Walk w{cell, crow, h};
    if (cell >= special_base) {
        // the target row is cold: park on the sentinel row and remember where the real row lives
        const SpecialCell sp = special[cell - special_base];
        w.row = hot_elems;
        w.crow = sp.next_off;
        if (sp.emit) {
            const uint32_t b = list_off[sp.emit - 1], e = list_off[sp.emit];
            for (uint32_t k = b; k < e; k++) w.h = record_atom(c, list[k], w.h);
        }
    } else if (cell >= emit_base) {
        // a hot row that emits: its EMIT cell (LDS) names the match
        const uint32_t code = *reinterpret_cast<lds_u16_ptr>((uintptr_t)((cell << 1) + emit_col2 + kClsBytes));
        if (code & 0x8000u) {
            w.h = record_atom(c, code & 0x7FFFu, w.h);
        } else {
            const uint32_t b = list_off[code - 1], e = list_off[code];
            for (uint32_t k = b; k < e; k++) w.h = record_atom(c, list[k], w.h);
        }
    }
    return w;
}
// The hit state a walk starts from when the request's record already holds hits (ListScanArgs::merge_rec).
__device__ __forceinline__ Hits hits_of_record(const uint32_t rv) {
    if (rv & REC_OVERFLOW) return Hits{0, 0, rv & ~REC_OVERFLOW};
    return Hits{rv & 0x7FFFu, (rv >> 15) & 0x7FFFu, kNone};
}
// The gap passes a finished request's hits call for: OR of the per-atom masks.
__device__ __noinline__ uint32_t gate_mask(const uint32_t *colmask_local, const PoolEntry *pool, Hits h) {
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
    return need;
}
// A finished request whose hits include prefilter factors is appended to the lists of the gated passes those factors guard
// (rare: one atomic per request and gated pass).
__device__ __noinline__ void enqueue_mask(uint32_t *gate_lists, uint32_t *gate_count, uint32_t n, uint32_t r, uint32_t need) {
    while (need) {
        const uint32_t g = (uint32_t)__builtin_ctz(need);
        need &= need - 1;
        gate_lists[(size_t)g * n + atomicAdd(&gate_count[g], 1u)] = r;
    }
}
// ... once per request and gap pass when the pass's enqueue bitmap is given (a list holds n entries; several chunks of one request — or
// its confirm tier and its walk — may all find a factor).
__device__ __noinline__ void enqueue_mask_once(uint32_t *gate_lists, uint32_t *gate_count, uint32_t n, uint32_t r, uint32_t need, uint32_t *enq_bits, uint32_t enq_words) {
    while (need) {
        const uint32_t g = (uint32_t)__builtin_ctz(need);
        need &= need - 1;
        if (enq_bits != nullptr) {
            const uint32_t bit = 1u << (r & 31u);
            if (atomicOr(&enq_bits[(size_t)g * enq_words + (r >> 5)], bit) & bit) continue;
        }
        gate_lists[(size_t)g * n + atomicAdd(&gate_count[g], 1u)] = r;
    }
}
__device__ __forceinline__ void enqueue_gated(const uint32_t *colmask_local, const PoolEntry *pool, uint32_t *gate_lists, uint32_t *gate_count, uint32_t n,
                                              uint32_t r, Hits h) {
    enqueue_mask(gate_lists, gate_count, n, r, gate_mask(colmask_local, pool, h));
}

// A pass descriptor read from the device-resident table into SCALAR registers (the address is wave-uniform; through a plain
// reference every use inside the hot loops became a reload from memory — the compiler must assume the kernel's own stores alias it —
// and the filter launch went from 0.74 to 0.85 ms).
// The wave's index in its workgroup, as a value the compiler KNOWS to be wave-uniform (threadIdx.x >> 6 is, but is not provably so:
// everything derived from it — slab and group numbers, loop bounds, base addresses — would be computed per lane in vector registers
// and every branch on it would be an exec-mask branch).
__device__ __forceinline__ uint32_t wave_index() { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Scalar mode of a table (dfa.cpp, utf8.h): does a 16-byte window hold a LEAD byte (>= 0xC0: bits 7 and 6 set)? Asked once per window
// and wave (ASCII traffic: never), so that the class fix-up below costs the common case ten vector instructions per 16 bytes.
__device__ __forceinline__ bool window_has_lead(const u32x4 w) {
    const uint32_t m = (w.x & (w.x << 1)) | (w.y & (w.y << 1)) | (w.z & (w.z << 1)) | (w.w & (w.w << 1));
    return (m & 0x80808080u) != 0u;
}
// The class of the scalar value whose lead byte b0 sits at arena position `at` of a field that ends at `end` (rare path: one unaligned
// load of the three bytes behind it, two dependent table loads).
__device__ __noinline__ uint32_t lead_class(const uint8_t *umap, const uint32_t ill_class, const PWAF_GLOBAL unsigned char *gdata, const uint32_t b0, const uint32_t at, const uint32_t end) {
    const uint32_t next = *reinterpret_cast<const PWAF_GLOBAL uint32_t __attribute__((aligned(1))) *>(gdata + at + 1u);  // (PWAF_ARENA_PAD covers the read behind the arena's end)
    return utf8_class(umap, ill_class, b0, next, end - at);
}

// scalar mode: a window with any byte beyond ASCII takes the class fix-up (lead bytes AND continuation bytes are looked at)
__device__ __forceinline__ bool window_has_high(const u32x4 w) { return ((w.x | w.y | w.z | w.w) & 0x80808080u) != 0u; }
// The class of a CONTINUATION byte at arena position `at` of the field [start, end): the table's own (every transition stays: the
// sequence's symbol was read at its lead byte) when a well-formed sequence holds it, else the ill-formed class (utf8.h: utf8_cont_covered).
__device__ __noinline__ uint32_t cont_class(const uint32_t own_class, const uint32_t ill_class, const PWAF_GLOBAL unsigned char *gdata, const uint32_t at, const uint32_t start, const uint32_t end) {
    const uint32_t back = min(3u, at - start);
    uint32_t prev = 0;
    for (uint32_t d = 1; d <= back; d++) prev |= (uint32_t)gdata[at - d] << (8u * (d - 1u));
    const uint32_t self_next = *reinterpret_cast<const PWAF_GLOBAL uint32_t __attribute__((aligned(1))) *>(gdata + at);  // (PWAF_ARENA_PAD covers the read behind the arena's end)
    return utf8_cont_covered(prev, self_next, back, end - at) ? own_class : ill_class;
}

template <class T>
__device__ __forceinline__ T load_descriptor(const T *p) {
    static_assert(sizeof(T) % 4 == 0, "descriptor size");
    typedef uint32_t __attribute__((may_alias)) word_t;  // (the descriptor's members are read and written as plain words)
    uint32_t w[sizeof(T) / 4];
    const word_t *s = reinterpret_cast<const word_t *>(p);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; i++) w[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s[i]);
    T out;
    __builtin_memcpy(&out, w, sizeof(T));
    return out;
}

#define PWAF_EMIT(id) h = emit_list(a.list_off, a.list, a.pool, a.pool_count, a.status, a.pool_cap, (id), h)

// CH = 16-byte chunks a lane walks per loop iteration. 2 halves the per-byte cost of the pull / finish logic (a third of the
// vector instructions at CH = 1) but idles a finished lane for up to 31 bytes instead of 15: it pays for long fields (URL,
// User-Agent), not for short ones (host, method). The engine picks it per pass from the tuning sample's mean field length.
// WIDE: a row has more than 127 byte classes, so a class's byte offset inside a row (class * 2) no longer fits the 256 x u8 class
// table; the table is then 256 x u32 (rare: it takes patterns that tell ~128 byte values apart).
template <int CH, bool WIDE>
__device__ __forceinline__ void scan_body(const ScanArgs &a) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t stride2 = a.stride * 2;
    const uint32_t hot_bytes = a.n_hot * stride2;            // sentinel row starts here
    const uint32_t hot_elems = hot_bytes >> 1;
    const uint32_t tab_bytes = (hot_bytes + stride2 + 15) & ~15u;
    // LDS: [0, kClsBytes) byte -> BYTE offset of its class cell inside a row (class * 2; one dword per byte value: a plain 32-bit load needs no masking), then the
    // hot rows + sentinel row. The kernel has no static LDS, so the dynamic segment starts at LDS address 0 (checked below) and
    // lookups use plain integer addresses: the region bases fold into the ds_read offset fields, a class lookup is one SDWA
    // shift + ds_read_b32, a DFA step is v_lshl_add (cell * 2 + class offset) + ds_read_u16.
    if ((uint32_t)(uintptr_t)(PWAF_LDS unsigned char *)lds != 0u) __builtin_trap();
    // the attribute kernel shares the CUs: scan waves issue first, its waves take the slots they leave (default priority 0)
    __builtin_amdgcn_s_setprio(3);
    const PWAF_GLOBAL unsigned char *gtab = (const PWAF_GLOBAL unsigned char *)a.tab;
    const PWAF_GLOBAL unsigned char *gdata = (const PWAF_GLOBAL unsigned char *)a.data;
    const PWAF_GLOBAL uint32_t *goff = (const PWAF_GLOBAL uint32_t *)a.off;
    const uint32_t tid = threadIdx.x, wave = wave_index(), lane = tid & 63;

    // stage the hot rows, the sentinel row and the byte-class map into LDS (coalesced 16 B per lane)
    for (uint32_t i = tid * 16; i < tab_bytes; i += kScanThreads * 16) {
        uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);  // sentinel cells: "special"
        if (i + 16 <= hot_bytes) v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(a.tab) + i);
        *reinterpret_cast<uint4 *>(lds + kClsBytes + i) = v;
    }
    __syncthreads();
    if (hot_bytes & 15) {  // the last, partially covered 16-byte slot of the hot rows
        const uint32_t base = hot_bytes & ~15u;
        if (tid < (hot_bytes & 15) / 2) reinterpret_cast<uint16_t *>(lds + kClsBytes + base)[tid] = a.tab[base / 2 + tid];
    }
    if (tid < 256) {
        if (WIDE) reinterpret_cast<uint32_t *>(lds)[tid] = a.classmap[tid] * 2u;
        else lds[tid] = (unsigned char)(a.classmap[tid] * 2u);
    }
    // a pass that owns prefilter factors keeps its atom -> gated-pass bitmask map in LDS too (one lookup per hit of a finished request)
    const uint32_t gate_base = kClsBytes + tab_bytes;
    if (a.colmask_local != nullptr)
        for (uint32_t i = tid; i < a.n_local; i += kScanThreads) reinterpret_cast<uint32_t *>(lds + gate_base)[i] = a.colmask_local[i];
    __syncthreads();

    const uint32_t end_col = a.n_classes + 1;
    // the STAY cell's byte offset, pinned in a vector register: as a scalar it would be re-materialised (v_mov) before every
    // one of the 16 selects, because v_cndmask cannot read a second scalar next to VCC
    uint32_t stay2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(stay2) : "s"(a.n_classes * 2u));
    const uint32_t emit_base = a.emit_base;        // cells >= emit_base: the row emits, or (>= special_base) is cold
    const uint32_t special_base = a.special_base;  // cells >= special_base index the special table
    const uint32_t emit_col2 = (a.n_classes + 2) * 2u;
    // this wave's slab of requests: contiguous, split evenly (rounding 1221 requests per wave up to 1280 would cost 5 % of every scan)
    const uint32_t n_items = a.n;
    const uint32_t total_waves = gridDim.x * kScanWaves;
    const uint32_t per_wave = (n_items + total_waves - 1) / total_waves;
    const uint32_t gw = blockIdx.x * kScanWaves + wave;
    const uint32_t w0 = min(n_items, gw * per_wave), w1 = min(n_items, w0 + per_wave);
    if (w0 >= w1) return;

    const unsigned long long lt_mask = (1ull << lane) - 1;
    const uint32_t start_emit = a.start_emit;

    uint32_t next = w0, blk = w0;
    auto load_off = [&](uint32_t base, uint32_t &lo, uint32_t &hi) {
        const uint32_t i = min(base + lane, n_items - 1);  // base + lane < n_items + 63; clamp keeps the load in bounds
        lo = goff[i];
        hi = goff[i + 1];
    };
    uint32_t o_lo, o_hi, n_lo = 0, n_hi = 0, n_base = kNone;
    load_off(blk, o_lo, o_hi);
    // Software pipeline: while a lane chews on the 16 bytes in `w`, the 16 bytes it will need in the NEXT iteration are
    // already in flight in `wn` — either the next chunk of the same field or, when this is the field's last chunk, the
    // first chunk of the request the lane has just pulled (r2/p2/end2). HBM/L2 latency hides behind 16 DFA steps.
    uint32_t r = kNone, p = 0, end = 0;           // current request
    uint32_t row = 0;                             // current row as a CELL value: its uint16 index in LDS, always even
                                                  // (== hot_elems: parked on the sentinel row, the real row is cold)
    uint32_t crow = 0;                            // byte offset of the current row in the full table while cold
    uint32_t r2 = kNone, p2 = 0, end2 = 0;        // request pulled ahead
    Hits h{0, 0, kNone};
    u32x4 w[CH], wn[CH];
#pragma unroll
    for (int q = 0; q < CH; q++) w[q] = wn[q] = u32x4{0, 0, 0, 0};
    constexpr uint32_t kStep = 16u * CH;  // bytes per iteration

    for (;;) {
        // offsets of the NEXT block of 64 work items: re-requested every iteration (L1 hits) instead of once per block inside a
        // branch, because a load whose result crosses a branch merge forces an immediate s_waitcnt vmcnt(0) — which would
        // also drain the chunk prefetch
        uint32_t f_lo, f_hi;
        const uint32_t f_base = blk + 64;
        load_off(f_base, f_lo, f_hi);
        // ---- 1. pull ahead: lanes on their last chunk (or idle) take the next request of the slab ----
        const bool last = r == kNone || p + kStep >= end;
        const unsigned long long want = __ballot(last && r2 == kNone);
        if (want != 0 && next < w1) {
            const uint32_t avail = min(w1 - next, blk + 64 - next);
            const uint32_t rank = (uint32_t)__builtin_popcountll(want & lt_mask);
            const bool take = last && r2 == kNone && rank < avail;
            const uint32_t j = take ? next + rank - blk : 0;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)o_lo);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)o_hi);
            if (take) {
                r2 = next + rank;
                p2 = lo;
                end2 = hi;
            }
            next += min((uint32_t)__builtin_popcountll(want), avail);
            if (next == blk + 64 && next < w1) {
                blk += 64;
                if (n_base == blk) {  // requested during an earlier iteration for this very block
                    o_lo = n_lo;
                    o_hi = n_hi;
                } else {              // two block switches in consecutive iterations (very short fields): fetch now
                    load_off(blk, o_lo, o_hi);
                }
            }
        }
        if (__ballot(r != kNone || r2 != kNone) == 0) break;
        {
            // Unconditional (no branch, so no wait is forced here): lanes with nothing to fetch read the arena's first bytes.
            // Unaligned 16-byte load; arenas carry PWAF_ARENA_PAD slack.
            const bool have = last ? (r2 != kNone && p2 < end2) : true;
            const uint32_t np = have ? (last ? p2 : p + kStep) : 0u;
            const uint32_t nend = last ? end2 : end;  // end of the field the next iteration works on
#pragma unroll
            for (int q = 0; q < CH; q++) {
                // a further chunk is fetched only if the field reaches it: reads never go more than PWAF_ARENA_PAD past a field's end
                const uint32_t at = (q == 0 || (have && np + 16u * q < nend)) ? np + 16u * q : 0u;
                wn[q] = *reinterpret_cast<const PWAF_GLOBAL u32x4_u *>(gdata + at);
            }
        }

        // ---- 2. 16 * CH bytes of every active lane's field ----
        const bool act = r != kNone && p < end;
        const uint32_t cnt_all = act ? min(kStep, end - p) : 0u;
#pragma unroll
        for (int q = 0; q < CH; q++) {
        const uint32_t cnt = cnt_all > 16u * q ? min(16u, cnt_all - 16u * q) : 0u;  // valid bytes of this chunk
        const uint32_t wd[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
        uint32_t c2[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t byte = (wd[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            // byte classes do not depend on the state: all 16 lookups are issued before the dependent chain starts
            const uint32_t c = WIDE ? *reinterpret_cast<lds_u32_ptr>((uintptr_t)(byte << 2)) : (uint32_t)*reinterpret_cast<lds_u8_ptr>((uintptr_t)byte);
            c2[k] = (uint32_t)k < cnt ? c : stay2;  // past the end: the STAY cell
        }
        if (a.umap != nullptr && __ballot(cnt != 0u && window_has_high(w[q])) != 0ull) {  // scalar mode, a byte beyond ASCII in somebody's chunk (rare)
            const uint32_t f_start = r != kNone ? a.off[r] : 0u;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t byte = (wd[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
                if (byte >= 0xC0u && (uint32_t)k < cnt) c2[k] = 2u * lead_class(a.umap, a.ill_class, gdata, byte, p + 16u * (uint32_t)q + (uint32_t)k, end);
                else if (byte >= 0x80u && (uint32_t)k < cnt) c2[k] = cont_class(c2[k], 2u * a.ill_class, gdata, p + 16u * (uint32_t)q + (uint32_t)k, f_start, end);
            }
        }
        // The 16 steps run in groups of 4 with ONE check per group: inside a group every lane chains lookup to lookup
        // (v_lshl_add + ds_read_u16, nothing else on the dependent path), treating whatever it reads as the next row. Only if
        // some lane read a cell >= emit_base (it entered an emitting row, left the hot set, or is parked on a cold row) are
        // the group's 4 steps re-walked carefully by the lanes concerned; all other lanes keep their speculative result.
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4) {
            const uint32_t t1 = *reinterpret_cast<lds_u16_ptr>((uintptr_t)((row << 1) + c2[k0] + kClsBytes));
            const uint32_t t2 = *reinterpret_cast<lds_u16_ptr>((uintptr_t)((t1 << 1) + c2[k0 + 1] + kClsBytes));
            const uint32_t t3 = *reinterpret_cast<lds_u16_ptr>((uintptr_t)((t2 << 1) + c2[k0 + 2] + kClsBytes));
            const uint32_t t4 = *reinterpret_cast<lds_u16_ptr>((uintptr_t)((t3 << 1) + c2[k0 + 3] + kClsBytes));
            if (max(max(t1, t2), max(t3, t4)) >= emit_base) {
                // Entering a hot row that emits does not break the chain (its cell is a real row), so the speculative cells stay
                // usable until the lane meets a special cell; the usual case — a match that is already in the request's record, or
                // fits a free slot — is settled from LDS and registers alone.
                uint32_t cur = row;
                bool chain = true;
                auto redo = [&](const uint32_t ts, const uint32_t cls, const bool in_range) {
                    const uint32_t t = chain ? ts : (uint32_t)*reinterpret_cast<lds_u16_ptr>((uintptr_t)((cur << 1) + cls + kClsBytes));
                    if (in_range) {  // (past the end the row does not change)
                        bool slow = t >= special_base;
                        if (!slow && t >= emit_base) {
                            const uint32_t code = *reinterpret_cast<lds_u16_ptr>((uintptr_t)((t << 1) + emit_col2 + kClsBytes));
                            const uint32_t x = (code & 0x7FFFu) + 1;
                            if (!(code & 0x8000u) || h.ovf != kNone) slow = true;  // a list of matches, or the record has overflowed
                            else if (h.a0 == x || h.a1 == x) {}
                            else if (h.a0 == 0) h.a0 = x;
                            else if (h.a1 == 0) h.a1 = x;
                            else slow = true;
                        }
                        if (slow) {
                            const Walk wk = careful_step(gtab, a.special, a.list_off, a.list, a.pool, a.pool_count, a.pool_cap, hot_elems | (emit_base << 16),
                                                         special_base | (emit_col2 << 16), cur, crow, cls, t, h.a0, h.a1, h.ovf);
                            chain = chain && wk.row == t;  // still on a real hot row: the cells read after it remain valid
                            cur = wk.row;
                            crow = wk.crow;
                            h = wk.h;
                        } else {
                            cur = t;
                        }
                    } else {
                        chain = chain && t < special_base;  // (a parked lane past its end reads the sentinel: nothing to reuse)
                    }
                };
                redo(t1, c2[k0], (uint32_t)k0 < cnt);
                redo(t2, c2[k0 + 1], (uint32_t)(k0 + 1) < cnt);
                redo(t3, c2[k0 + 2], (uint32_t)(k0 + 2) < cnt);
                redo(t4, c2[k0 + 3], (uint32_t)(k0 + 3) < cnt);
                row = cur;
            } else {
                row = t4;
            }
        }
        }  // chunks
        p += cnt_all;

        // ---- 3. finished requests: end-of-field matches, the hit record, then switch to the pulled-ahead request ----
        if (r != kNone && p >= end) {
            const uint32_t e = row == hot_elems ? (uint32_t)*reinterpret_cast<glb_u16_ptr>(gtab + crow + (end_col << 1))
                                                : (uint32_t)*reinterpret_cast<lds_u16_ptr>((uintptr_t)(((row + end_col) << 1) + kClsBytes));
            if (e & 0x8000u) {  // a single end-of-field match: settled in registers unless the record is full
                const uint32_t x = (e & 0x7FFFu) + 1;
                if (h.ovf == kNone && (h.a0 == x || h.a1 == x)) {
                } else if (h.ovf == kNone && h.a0 == 0) {
                    h.a0 = x;
                } else if (h.ovf == kNone && h.a1 == 0) {
                    h.a1 = x;
                } else {
                    const SlowCtx sc{a.list_off, a.list, a.pool, a.pool_count, a.status, a.pool_cap};
                    h = record_atom(sc, x - 1, h);
                }
            } else if (e) {
                PWAF_EMIT(e - 1);
            }
            a.rec[r] = h.ovf != kNone ? (REC_OVERFLOW | h.ovf) : (h.a0 | (h.a1 << 15));
            if (a.colmask_local != nullptr && (h.a0 | (h.ovf + 1u)) != 0) {
                // does any hit of this request gate a later pass? (LDS lookups; the enqueue itself is rare and out of line)
                uint32_t need = 1;  // overflowed record: let the slow path walk the chain
                if (h.ovf == kNone) {
                    need = *reinterpret_cast<lds_u32_ptr>((uintptr_t)(gate_base + (h.a0 - 1) * 4));
                    if (h.a1) need |= *reinterpret_cast<lds_u32_ptr>((uintptr_t)(gate_base + (h.a1 - 1) * 4));
                }
                if (need) enqueue_gated(a.colmask_local, a.pool, a.gate_lists, a.gate_count, a.n, r, h);
            }
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
#pragma unroll
        for (int q = 0; q < CH; q++) w[q] = wn[q];
        n_lo = f_lo;
        n_hi = f_hi;
        n_base = f_base;
    }
}

template <int CH, bool WIDE>
__global__ __launch_bounds__(kScanThreads) void scan_kernel(ScanArgs a) { scan_body<CH, WIDE>(a); }

// lscan_kernel: every list-driven pass of a phase in ONE persistent launch. One listed request per lane, walked byte by byte: per step
// one LDS read for the byte's class (independent of the state: issued ahead) and one dependent read of the cell from the LDS copy of
// the table's hottest rows. (The wave-lockstep walk of scan_kernel, built for streaming EVERY request, took 0.6 ms on the same
// lists: every lane a candidate, its slow paths fire in every group of steps.)
//
// Cold rows: a cell outside the LDS copy lives in the L2-resident flat table (~1 us away) and the wave waits for it. (Round 3 tried
// PARKING such lanes — stop consuming bytes, issue all of a window's cold loads together at its end, resume one byte further: at most
// one round trip per 16-byte window instead of one per step. Measured slower everywhere: benign 0.234 vs 0.177 ms, adversarial 5.3 vs
// 3.9 ms for the filtered passes of the 1k-rule set. The extra selects per step cost more than the waits they save — on adversarial
// traffic the walk is bound by VALU + LDS issue, ~10 instructions per byte and lane, not by latency — and a lane deep in rarely
// visited states, which parks at every byte, became twice as slow and sets the wave's time.)
//
// Work distribution (round 3): the list lengths are only known on the device. lscan_plan_kernel turns them into a prefix sum of work
// items (kListThreads entries of one pass each); the scan launch is a persistent grid in which workgroup b takes the contiguous items
// [total * b / G, total * (b + 1) / G) — consecutive items are mostly of one pass, whose hot rows are staged once. (Round 2 launched
// ceil(n / 512) x passes workgroups, nearly all of which found their list exhausted: 53k launches-and-exits for the 69 filtered
// passes of the 4096-rule set.)
// A work item is one WAVE's share of a list: `epi` entries (a power of two up to 64, per pass: plan[count + 1 + pass]). A long list
// packs 64 walks into a wave; a SHORT one — the few requests whose regex factor the confirm tier found, a gap pass's requests — is
// spread one or a few walks per wave over the waves the launch has: the 64 walks of a wave advance in lockstep and every emit or cold
// cell of one lane sends the whole wave through the slow path of its group of steps, so a wave of 64 true hits (what a dense walk list
// is made of) crawls while the rest of the chip idles (measured: 0.07 -> 0.5 ms for ~10k walks when the lists became dense).
// the list a pass walks in this batch (kernels.h: ListScanArgs::dense_flag)
__device__ __forceinline__ ListOf list_of(const ListScanArgs &a) {
    ListOf l{a.req_list, a.req_list != nullptr ? min(*a.n_list, a.n) : a.n};
    if (a.dense_mode != 0u) {
        const bool dense = *a.dense_flag > a.dense_thresh;
        if (a.dense_mode == 1u && !dense) l.n_l = 0u;
        if (a.dense_mode == 2u && dense) l.n_l = 0u;
        if (a.dense_mode == 3u && dense) l = ListOf{nullptr, a.n};
    }
    return l;
}

// (Round 6: the entries per item also have a floor common to the launch's passes — the power of two with which ALL the passes' entries fit the launch's
// waves once, 16 at most. A pass's own rule alone let six gap passes that ride one walk list of 16k requests make 4 063 four-entry items EACH: 24k items on
// 6 144 waves, four nearly empty items one behind the other per wave, every one the whole dependent chain of a walk — 0.084 ms for 97k entries, most of them
// skipped by their need bit. With the common floor they are 16-entry items, one per wave: lscan_x6 0.084 -> 0.039 ms alone, 0.11 -> 0.05 in the step; the
// 4096-rule set's gap launch 0.094 -> 0.051, its hostile stream's R-tier launch 0.236 -> 0.107. Measured and dropped: the floor counted against the waves of ONE
// workgroup per CU — what is resident beside the attribute kernel — with the items packed into the first workgroups: fewer, fuller waves walk in lockstep
// and are slower — a 1.25M-request share's gap launch 0.033 -> 0.044 ms, nothing gained at 10M.)
__global__ __launch_bounds__(256) void lscan_plan_kernel(GatedTable b, uint32_t *plan /* [2 count + 1] */, uint32_t n_waves) {
    __shared__ uint32_t part[256];
    __shared__ unsigned long long all_entries;
    const uint32_t t = threadIdx.x;
    uint32_t items = 0;
    const uint32_t n_l = t < b.count ? list_of(b.g[t]).n_l : 0u;
    if (t == 0) all_entries = 0ull;
    __syncthreads();
    if (n_l != 0u) atomicAdd(&all_entries, (unsigned long long)n_l);
    __syncthreads();
    if (t < b.count) {
        const uint32_t room = n_waves > b.count ? n_waves - b.count : 1u;  // (every pass may end in a partly filled item)
        const unsigned long long per_wave = (all_entries + room - 1u) / room;
        uint32_t epi = 1;
        while (epi < 16u && epi < per_wave) epi <<= 1;
        while (epi < 64u * kListWalks && (n_l + epi - 1) / epi > n_waves) epi <<= 1;
        items = (n_l + epi - 1) / epi;
        plan[b.count + 1 + t] = epi;
    }
    part[t] = items;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive scan (count <= 250)
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (t < b.count) plan[t] = part[t] - items;
    if (t == 0) plan[b.count] = part[255];
}

// One step of a walk whose state may lie outside the LDS-resident rows: a DELTA record (kernels.h: ListScanArgs::delta — base row and
// two exception cells, 8 bytes in LDS) or, failing that, the L2-resident flat table. `c` = the byte's class (n_classes + 1: past the
// field's end, the state stays).
__device__ __forceinline__ uint32_t list_step_slow(const uint16_t *hot, const PWAF_GLOBAL uint16_t *flat, const uint64_t *delta, const uint32_t state, const uint32_t c,
                                                   const uint32_t stride, const uint32_t n_hot, const uint32_t n_delta) {
    if (state < n_hot) return hot[state * stride + c];
    const uint32_t k = state - n_hot;
    if (k < n_delta) {
        if (c + 2u == stride) return state;  // the STAY cell
        const uint64_t rec = delta[k];
        const uint32_t lo = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
        if (c == ((lo >> 16) & 0xFFu)) return hi & 0xFFFFu;
        if (c == (lo >> 24)) return hi >> 16;
        return hot[(lo & 0xFFFFu) * stride + c];
    }
    return flat[state * stride + c];
}

// A LONG list (more than an eighth of the batch: hostile traffic — near misses of the rule literals in most requests), walked with the
// cold steps taken TOGETHER. In the lockstep loop below a group of four steps in which ANY of the 64 lanes meets a cold cell (a row
// that is not LDS-resident: an L2 round trip) is re-walked step by step by the whole wave, and although a lane spends under a tenth
// of its steps in cold rows, some lane of the wave does nearly always: measured on the hostile stream of the 1k-rule set, 28 of the
// 36 groups of a wave's walk take the slow path, and those round trips are 85 % of the scan's 3.6 ms. Here every lane keeps its own
// position: per iteration a lane whose next four cells are LDS-resident takes them (the wave's cheap iteration), a lane that is not
// waits — and when half the wave is waiting (or nobody else can move) ONE slow iteration takes the blocked lanes through their
// group, cold cells from L2, while the others wait: a round trip then serves thirty lanes instead of one.
template <uint32_t THREADS>
__device__ __forceinline__ void lscan_async(const ListScanArgs &a, const uint16_t *hot, const unsigned char *cls, const uint64_t *delta, uint32_t it, const uint32_t it_end,
                                            const uint32_t first, const uint32_t n_l, const uint32_t hot_elems, const uint32_t epi) {
    const uint32_t ncls = a.n_classes, stride = ncls + 3u;
    const PWAF_GLOBAL unsigned char *gdata = (const PWAF_GLOBAL unsigned char *)a.data;
    const PWAF_GLOBAL uint16_t *flat = (const PWAF_GLOBAL uint16_t *)a.flat;
    auto record_emit = [&](const uint32_t st, Hits &hh) {
        const uint32_t ei = st * stride + ncls;
        const uint32_t code = ei < hot_elems ? (uint32_t)hot[ei] : (uint32_t)flat[ei];
        const uint32_t x = (code & 0x7FFFu) + 1u;
        bool slow = !(code & 0x8000u) || hh.ovf != kNone;
        if (!slow) {
            if (hh.a0 == x || hh.a1 == x) {}
            else if (hh.a0 == 0) hh.a0 = x;
            else if (hh.a1 == 0) hh.a1 = x;
            else slow = true;
        }
        if (slow) hh = emit_list(a.emit_off, a.emit_list, a.pool, a.pool_count, a.status, a.pool_cap, st, hh);
    };
    for (it += wave_index(); it < it_end; it += THREADS / 64u) {  // (work items are one wave's share of the list: this wave's)
        const uint32_t li = (it - first) * epi + (threadIdx.x & 63u);
        bool live = (threadIdx.x & 63u) < epi && li < n_l;
        if (live && a.need_in != nullptr) live = ((a.need_in[li] >> a.need_bit) & 1u) != 0;  // (a sharing gap pass: none of its factors fired here)
        const uint32_t r = live ? (a.req_list != nullptr ? a.req_list[li] : li) : 0u;
        if (live && a.visited != nullptr) atomicOr(&a.visited[r >> 5], 1u << (r & 31u));
        uint32_t p = live ? a.off[r] : 0u;           // the next byte to read
        const uint32_t end = live ? a.off[r + 1] : 0u;
        uint32_t state = 0, wp = p;                  // wp: where the window in `w` begins (p - wp is a multiple of 4, below 16 after the reload)
        Hits h{0, 0, kNone};
        uint32_t need_init = 0;  // what the record's hits (filter heads, confirm tier) already enqueued
        if (live && a.merge_rec) {
            h = hits_of_record(a.rec[r]);
            if (a.colmask_local != nullptr && (h.a0 | (h.ovf + 1u)) != 0) need_init = gate_mask(a.colmask_local, a.pool, h);
        }
        if (live && a.emit_off[1] != a.emit_off[0]) h = emit_list(a.emit_off, a.emit_list, a.pool, a.pool_count, a.status, a.pool_cap, 0u, h);
        u32x4 w = *reinterpret_cast<const PWAF_GLOBAL u32x4_u *>(gdata + wp);
        u32x4 wn = *reinterpret_cast<const PWAF_GLOBAL u32x4_u *>(gdata + (wp + 16u < end ? wp + 16u : 0u));
        for (;;) {
            const bool active = p < end;
            const unsigned long long am = __ballot(active);
            if (am == 0) break;
            if (active && p - wp >= 16u) {  // this lane's window is used up: the prefetched one takes its place, the one after is requested
                w = wn;
                wp += 16u;
                wn = *reinterpret_cast<const PWAF_GLOBAL u32x4_u *>(gdata + (wp + 16u < end ? wp + 16u : 0u));
            }
            const uint32_t j = (p - wp) >> 2;
            const uint32_t wd = j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w;
            const uint32_t cnt = active ? min(4u, end - p) : 0u;
            uint32_t c[4], t[4], sv[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) c[k] = cls[(wd >> (k * 8)) & 0xFFu];
            asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));  // (the four class lookups stay unconditional: left alone the compiler sinks each under its "k < cnt" — sixteen exec-mask branches per window, which doubled the walk's time)
            if (a.umap != nullptr && __ballot(active && (wd & 0x80808080u) != 0u) != 0ull) {  // scalar mode, a byte beyond ASCII somewhere (rare): the scalar's class at a lead byte, a stray continuation byte is ill-formed
                const uint32_t f_start = active ? a.off[r] : 0u;
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t byte = (wd >> (k * 8)) & 0xFFu;
                    if (byte >= 0xC0u && k < cnt) c[k] = lead_class(a.umap, a.ill_class, gdata, byte, p + k, end);
                    else if (byte >= 0x80u && k < cnt) c[k] = cont_class(c[k], a.ill_class, gdata, p + k, f_start, end);
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) c[k] = k < cnt ? c[k] : ncls + 1u;
#pragma unroll
            // (A branch-free step that serves row states and delta-record states alike — record, then the cell of the row or of the
            // record's base row, then the two exceptions: two dependent LDS reads for every lane — was measured: 4.9 ms against 3.1
            // for the hostile stream's filtered passes. States without a row in LDS, records included, wait for the slow iteration.)
            for (uint32_t k = 0; k < 4; k++) {
                t[k] = *reinterpret_cast<lds_u16_ptr>((uintptr_t)min(__umul24(k == 0 ? state : sv[k - 1], 2u * stride) + 2u * c[k], 2u * hot_elems));
                sv[k] = t[k] & 0x7FFFu;
            }
            const uint32_t any = t[0] | t[1] | t[2] | t[3], top = max(max(t[0], t[1]), max(t[2], t[3]));
            const bool blocked = active && top == 0xFFFFu;
            const unsigned long long bm = __ballot(blocked);
            const uint32_t n_blocked = (uint32_t)__builtin_popcountll(bm);
            if (n_blocked < 32u && bm != am) {
                // the cheap iteration: every lane whose four cells are LDS-resident takes them
                if (active && !blocked) {
                    if (any & 0x8000u) {
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++)
                            if (t[k] & 0x8000u) record_emit(sv[k], h);
                    }
                    state = sv[3];
                    p += 4u;
                }
            } else if (blocked) {
                // the slow iteration: the blocked lanes' group step by step, cold cells from the L2-resident table
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t tt = list_step_slow(hot, flat, delta, state, c[k], stride, a.n_hot, a.n_delta);
                    state = tt & 0x7FFFu;
                    if (tt & 0x8000u) record_emit(state, h);
                }
                p += 4u;
            }
        }
        if (live) {
            if (a.end_off[state + 1] != a.end_off[state]) h = emit_list(a.end_off, a.end_list, a.pool, a.pool_count, a.status, a.pool_cap, state, h);
            a.rec[r] = h.ovf != kNone ? (REC_OVERFLOW | h.ovf) : (h.a0 | (h.a1 << 15));
            if (a.colmask_local != nullptr) {
                uint32_t need = 0;
                if ((h.a0 | (h.ovf + 1u)) != 0) need = gate_mask(a.colmask_local, a.pool, h);
                if (a.need_out != nullptr) a.need_out[li] = need;
                need &= ~(a.shared_bits | need_init);
                if (need) enqueue_mask_once(a.gate_lists, a.gate_count, a.n, r, need, a.enq_bits, a.enq_words);
            }
        }
    }
}

// (waves per SIMD: three 512-thread workgroups per CU need 6, i.e. at most 80 vector registers)
template <uint32_t THREADS>
__global__ __launch_bounds__(THREADS, THREADS == 512 ? 6 : 4) void lscan_kernel(GatedTable b, const uint32_t *plan, uint32_t hot_bytes) {
    extern __shared__ uint32_t lscan_lds[];  // [hot_bytes / 4] hot rows + 16 bytes for the sentinel cell, then the 256-byte class map
    __builtin_amdgcn_s_setprio(3);  // on the critical path, beside the attribute kernel's background waves
    if ((uint32_t)(uintptr_t)(PWAF_LDS unsigned char *)lscan_lds != 0u) __builtin_trap();  // (no static LDS in this kernel: the walks address the hot rows by plain integer offsets)
    const uint32_t total = plan[b.count];
    const uint32_t it0 = (uint32_t)((uint64_t)total * blockIdx.x / gridDim.x), it1 = (uint32_t)((uint64_t)total * (blockIdx.x + 1) / gridDim.x);
    const unsigned char *cls = reinterpret_cast<const unsigned char *>(lscan_lds + (hot_bytes + 16u) / 4);
    const uint16_t *hot = reinterpret_cast<const uint16_t *>(lscan_lds);
    for (uint32_t it = it0; it < it1;) {
        // the pass of item `it`: the last p with plan[p] <= it (uniform)
        uint32_t ps = 0;
        for (uint32_t lo = 0, hi = b.count; lo + 1 < hi;) {
            const uint32_t mid = (lo + hi) >> 1;
            if (plan[mid] <= it) lo = mid;
            else hi = mid;
            ps = lo;
        }
        const uint32_t first = plan[ps], it_end = min(it1, plan[ps + 1]), epi = plan[b.count + 1 + ps];
        ListScanArgs a = load_descriptor(&b.g[ps]);
        const ListOf lst = list_of(a);
        a.req_list = lst.req_list;  // (the flag-density switch may have turned a shared list into "every request")
        const uint32_t ncls = a.n_classes, stride = ncls + 3u;  // row: ncls transitions, the EMIT cell, the STAY cell, the END cell
        const uint32_t hot_elems = a.n_hot * stride;
        uint64_t *delta_lds = reinterpret_cast<uint64_t *>(lscan_lds) + (hot_elems * 2u + 2u + 15u) / 16u * 2u;  // (16-byte aligned, behind the sentinel cell)
        __syncthreads();  // (every wave is done with the previous pass's rows)
        {
            // 16 bytes per lane and load, four loads in flight (a dword-per-lane loop of dependent load -> store pairs took a dozen L2
            // round trips per workgroup: as long as the walk of a short list). The flat table is padded to whole 16-byte units.
            const uint4 *src = reinterpret_cast<const uint4 *>(a.flat);
            uint4 *dst = reinterpret_cast<uint4 *>(lscan_lds);
            const uint32_t units = (hot_elems * 2u + 15u) / 16u;
            for (uint32_t k = threadIdx.x; k < units; k += THREADS * 4u) {
                uint4 v[4];
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) v[q] = k + q * THREADS < units ? src[k + q * THREADS] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (uint32_t q = 0; q < 4; q++)
                    if (k + q * THREADS < units) dst[k + q * THREADS] = v[q];
            }
            if (threadIdx.x < 64) lscan_lds[(hot_bytes + 16u) / 4 + threadIdx.x] = reinterpret_cast<const uint32_t *>(a.classmap)[threadIdx.x];  // class map
            // the delta records, behind the rows and the sentinel cell (the engine sized rows + 32 + records to fit hot_bytes)
            for (uint32_t k = threadIdx.x; k < a.n_delta; k += THREADS) delta_lds[k] = a.delta[k];
        }
        __syncthreads();
        // the SENTINEL cell right behind the hot rows (after the barrier: the staging loop's last 16-byte unit may cover it): every
        // index beyond the hot rows is clamped onto it and reads 0xFFFF = "this cell is cold" (no state has id 0x7FFF)
        if (threadIdx.x == 0) reinterpret_cast<uint16_t *>(lscan_lds)[hot_elems] = 0xFFFFu;
        __syncthreads();
        const uint32_t n_l = lst.n_l;
        // (the gap passes' lists stay with the lockstep loop: measured 0.25 -> 0.40 ms with this one on the hostile stream — their walks
        // are short and mostly LDS-resident, and the asynchronous iteration costs more per group)
#ifdef PWAF_PROFILING
        const bool long_list = (b.debug & 1u) ? false : (b.debug & 2u) ? true : (a.behind_filter != 0u && (uint64_t)n_l * 8u >= a.n);  // timing experiments: never / always the asynchronous loop (same results)
#else
        const bool long_list = a.behind_filter != 0u && (uint64_t)n_l * 8u >= a.n;
#endif
        if (long_list) {  // (uniform) a prefilter's candidate list that holds more than an eighth of the batch: lscan_async
            lscan_async<THREADS>(a, hot, cls, delta_lds, it, it_end, first, n_l, hot_elems, epi);
            it = it_end;
            continue;
        }
        const PWAF_GLOBAL unsigned char *gdata = (const PWAF_GLOBAL unsigned char *)a.data;
        const PWAF_GLOBAL uint16_t *flat = (const PWAF_GLOBAL uint16_t *)a.flat;
        // A work item is ONE WAVE's worth of list entries (round 4; it used to be a workgroup's: a short list — the few requests whose
        // regex factor the confirm tier found, a gap pass's requests — then sat in a handful of full waves on a handful of CUs, every
        // wave as slow as the unluckiest of its 64 walks, while the rest of the chip idled: 0.07 -> 0.6 ms when the lists became dense).
        for (uint32_t iw = it + wave_index(); iw < it_end; iw += THREADS / 64u) {
            // kListWalks listed requests per lane, walked in lockstep. Measured with 2 (round 3), in the hope that a lane's two cold cells
            // travelling to L2 together would help hostile traffic (near misses of the rule literals drive most walks deep into states
            // that are not LDS-resident): adversarial filtered passes 3.87 -> 3.41 ms, but benign 0.143 -> 0.204 ms and the gap
            // passes 0.087 -> 0.117 (adversarial 0.24 -> 0.49): a single wave issues an instruction every few cycles at best, so two
            // interleaved chains cost a step twice the issue slots while the list's longest walk — the kernel's critical path on
            // benign traffic — gets no shorter. One walk per lane and as many waves as the LDS copy allows is the better trade.
            uint32_t li[kListWalks], r[kListWalks], p[kListWalks], end[kListWalks], state[kListWalks], need_init[kListWalks];
            bool live[kListWalks];
            Hits h[kListWalks];
            u32x4 w[kListWalks];
#pragma unroll
            for (uint32_t u = 0; u < kListWalks; u++) {
                li[u] = (iw - first) * epi + u * 64u + (threadIdx.x & 63u);
                live[u] = (threadIdx.x & 63u) + u * 64u < epi && li[u] < n_l;
                if (live[u] && a.need_in != nullptr) live[u] = ((a.need_in[li[u]] >> a.need_bit) & 1u) != 0;  // (a sharing gap pass: none of its factors fired here)
                r[u] = live[u] ? (a.req_list != nullptr ? a.req_list[li[u]] : li[u]) : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < kListWalks; u++) {
                if (live[u] && a.visited != nullptr) atomicOr(&a.visited[r[u] >> 5], 1u << (r[u] & 31));
                p[u] = live[u] ? a.off[r[u]] : 0u;
                end[u] = live[u] ? a.off[r[u] + 1] : 0u;
                state[u] = 0;
                h[u] = Hits{0, 0, kNone};
                need_init[u] = 0;
                if (live[u] && a.merge_rec) {  // (the walk of a pass with a confirm tier: the record holds the heads' and the literals' hits)
                    h[u] = hits_of_record(a.rec[r[u]]);
                    if (a.colmask_local != nullptr && (h[u].a0 | (h[u].ovf + 1u)) != 0) need_init[u] = gate_mask(a.colmask_local, a.pool, h[u]);
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < kListWalks; u++) {
                if (live[u] && a.emit_off[1] != a.emit_off[0]) h[u] = emit_list(a.emit_off, a.emit_list, a.pool, a.pool_count, a.status, a.pool_cap, 0u, h[u]);
                w[u] = *reinterpret_cast<const PWAF_GLOBAL u32x4_u *>(gdata + p[u]);
            }
            // what entering state `st` emits rides in its row: a single atom that is already in the record, or fits a free slot, is
            // settled in registers (no memory access at all for a hot row)
            auto record_emit = [&](const uint32_t st, Hits &hh) {
                const uint32_t ei = st * stride + ncls;
                const uint32_t code = ei < hot_elems ? (uint32_t)hot[ei] : (uint32_t)flat[ei];
                const uint32_t x = (code & 0x7FFFu) + 1u;
                bool slow = !(code & 0x8000u) || hh.ovf != kNone;
                if (!slow) {
                    if (hh.a0 == x || hh.a1 == x) {}
                    else if (hh.a0 == 0) hh.a0 = x;
                    else if (hh.a1 == 0) hh.a1 = x;
                    else slow = true;
                }
                if (slow) hh = emit_list(a.emit_off, a.emit_list, a.pool, a.pool_count, a.status, a.pool_cap, st, hh);
            };
            // Per byte the walk is mad -> min -> ds_read -> and, in groups of four steps checked once: a step into a cold row reads the
            // sentinel (and stays there for the rest of the group), a step into an emitting state carries bit 15; bytes past the field's
            // end take the STAY cell, so no step is conditional. Only a group that met a cold cell is re-walked step by step (both walks
            // of the lane together: their L2 loads are issued back to back); emits are recorded after the group, off the chain. The next
            // 16-byte window of each walk is in flight while this one is walked. (Round 2: ~25 instructions and three branches per byte,
            // and a full global round trip per window on the critical path.)
            for (;;) {
                bool more = false;
#pragma unroll
                for (uint32_t u = 0; u < kListWalks; u++) more = more || p[u] < end[u];
                if (!more) break;
                u32x4 wn[kListWalks];
                uint32_t cnt[kListWalks];
                bool lead_here = false;  // scalar mode: some walk's window holds a byte beyond ASCII (wave-uniform after the ballot)
#pragma unroll
                for (uint32_t u = 0; u < kListWalks; u++) {
                    const uint32_t pn = p[u] + 16u;
                    wn[u] = *reinterpret_cast<const PWAF_GLOBAL u32x4_u *>(gdata + (pn < end[u] ? pn : 0u));
                    cnt[u] = p[u] < end[u] ? min(16u, end[u] - p[u]) : 0u;
                    if (a.umap != nullptr) lead_here = lead_here || (cnt[u] != 0u && window_has_high(w[u]));
                }
                const bool fix_classes = a.umap != nullptr && __ballot(lead_here) != 0ull;
#pragma unroll
                for (uint32_t k0 = 0; k0 < 16; k0 += 4) {
                    uint32_t c[kListWalks][4], t[kListWalks][4], sv[kListWalks][4];
#pragma unroll
                    for (uint32_t u = 0; u < kListWalks; u++) {
                        const uint32_t wd = k0 == 0 ? w[u].x : k0 == 4 ? w[u].y : k0 == 8 ? w[u].z : w[u].w;
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) {
                            const uint32_t cl = cls[(wd >> (k * 8)) & 0xFFu];
                            c[u][k] = k0 + k < cnt[u] ? cl : ncls + 1u;
                        }
                        if (fix_classes) {  // (uniform, rare) a lead byte reads as the class of the scalar value it begins; its continuation bytes stay, a stray one is ill-formed
#pragma unroll
                            for (uint32_t k = 0; k < 4; k++) {
                                const uint32_t byte = (wd >> (k * 8)) & 0xFFu;
                                if (byte >= 0xC0u && k0 + k < cnt[u]) c[u][k] = lead_class(a.umap, a.ill_class, gdata, byte, p[u] + k0 + k, end[u]);
                                else if (byte >= 0x80u && k0 + k < cnt[u]) c[u][k] = cont_class(c[u][k], a.ill_class, gdata, p[u] + k0 + k, live[u] ? a.off[r[u]] : 0u, end[u]);
                            }
                        }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
#pragma unroll
                        for (uint32_t u = 0; u < kListWalks; u++) {
                            // per step: v_mad_u32_u24 (a 32-bit multiply is quarter rate) -> v_min -> ds_read_u16 -> v_and, all in BYTE offsets
                    t[u][k] = *reinterpret_cast<lds_u16_ptr>((uintptr_t)min(__umul24(k == 0 ? state[u] : sv[u][k - 1], 2u * stride) + 2u * c[u][k], 2u * hot_elems));  // (the hot rows start at LDS address 0)
                            sv[u][k] = t[u][k] & 0x7FFFu;
                        }
                    }
                    uint32_t any = 0, top = 0;
#pragma unroll
                    for (uint32_t u = 0; u < kListWalks; u++)
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) { any |= t[u][k]; top = max(top, t[u][k]); }
                    if (any & 0x8000u) {
                        if (top == 0xFFFFu) {
                            // a cold cell: the group step by step for both walks, cold cells from the L2-resident table
#pragma unroll
                            for (uint32_t k = 0; k < 4; k++) {
#pragma unroll
                                for (uint32_t u = 0; u < kListWalks; u++) {
                                    const uint32_t tt = list_step_slow(hot, flat, delta_lds, state[u], c[u][k], stride, a.n_hot, a.n_delta);
                                    state[u] = tt & 0x7FFFu;
                                    if (tt & 0x8000u) record_emit(state[u], h[u]);
                                }
                            }
                        } else {
#pragma unroll
                            for (uint32_t u = 0; u < kListWalks; u++) {
#pragma unroll
                                for (uint32_t k = 0; k < 4; k++)
                                    if (t[u][k] & 0x8000u) record_emit(sv[u][k], h[u]);
                                state[u] = sv[u][3];
                            }
                        }
                    } else {
#pragma unroll
                        for (uint32_t u = 0; u < kListWalks; u++) state[u] = sv[u][3];
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < kListWalks; u++) {
                    if (p[u] < end[u]) p[u] += 16u;
                    w[u] = wn[u];
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < kListWalks; u++) {
                if (!live[u]) continue;
                if (a.end_off[state[u] + 1] != a.end_off[state[u]]) h[u] = emit_list(a.end_off, a.end_list, a.pool, a.pool_count, a.status, a.pool_cap, state[u], h[u]);
                a.rec[r[u]] = h[u].ovf != kNone ? (REC_OVERFLOW | h[u].ovf) : (h[u].a0 | (h[u].a1 << 15));
                if (a.colmask_local != nullptr) {
                    uint32_t need = 0;
                    if ((h[u].a0 | (h[u].ovf + 1u)) != 0) need = gate_mask(a.colmask_local, a.pool, h[u]);
                    if (a.need_out != nullptr) a.need_out[li[u]] = need;
                    need &= ~(a.shared_bits | need_init[u]);
                    if (need) enqueue_mask_once(a.gate_lists, a.gate_count, a.n, r[u], need, a.enq_bits, a.enq_words);
                }
            }
        }
        it = it_end;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// confirm tier (confirm.h; program.h: ConfirmTable)
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void confirm_plan_kernel(ConfirmTableDev b, uint32_t *plan /* [count + 1] */) {
    __shared__ uint32_t part[256];
    const uint32_t t = threadIdx.x;
    uint32_t items = 0;
    if (t < b.count) {
        const ConfirmArgs *pa = &b.c[t];
        const bool dense = pa->dense_flag != nullptr && *pa->dense_flag > pa->dense_thresh;  // (walked whole this batch: nothing to confirm)
        items = dense ? 0u : (min(*pa->pair_count, pa->pair_cap) + kConfirmThreads - 1) / kConfirmThreads;
    }
    part[t] = items;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (t < b.count) plan[t] = part[t] - items;
    if (t == 0) plan[b.count] = part[255];
}

// A literal atom confirmed for request r, merged into its hit record. Several lanes — other flagged chunks of the same request, in other
// workgroups on other XCDs — may be merging at once, so every access to the shared words is a read-modify-write atomic (performed at
// the device's coherence point; a returned atomic has completed when its value arrives, which orders an overflow entry's words before
// the compare-and-swap that publishes it): no fences — an agent-scope fence writes back and invalidates the whole L2, and two per hit
// took this kernel from 0.3 to 1.5 ms and its neighbours with it. An overflow chain grows like a lock-free stack (entries allocated for
// an attempt that lost its compare-and-swap are leaked: rare, and the pool is sized per batch).
__device__ __noinline__ void merge_atom(PoolEntry *pool, uint32_t *pool_count, uint32_t *status, uint32_t pool_cap, uint32_t *rec, uint32_t atom) {
    auto rmw_read = [](uint32_t *p) { return atomicOr(p, 0u); };
    auto publish = [&](uint32_t idx, uint32_t at, uint32_t next) {
        const uint32_t t0 = atomicExch(&pool[idx].atom, at), t1 = atomicExch(&pool[idx].next, next);
        asm volatile("" ::"v"(t0), "v"(t1));  // (both have been performed before anything below is issued)
    };
    uint32_t old = rmw_read(rec);
    for (;;) {
        uint32_t upd;
        const uint32_t x = atom + 1u;
        if (!(old & REC_OVERFLOW)) {
            const uint32_t a0 = old & 0x7FFFu, a1 = (old >> 15) & 0x7FFFu;
            if (a0 == x || a1 == x) return;
            if (a0 == 0u) upd = old | x;
            else if (a1 == 0u) upd = old | (x << 15);
            else {  // a third atom: the record becomes a chain of all three
                const uint32_t idx = atomicAdd(pool_count, 3u);
                if (idx + 3u > pool_cap) { atomicOr(status, 1u); return; }  // (the host runs the batch again with the pool the allocator asked for)
                publish(idx, a0 - 1u, kNone);
                publish(idx + 1u, a1 - 1u, idx);
                publish(idx + 2u, atom, idx + 1u);
                upd = REC_OVERFLOW | (idx + 2u);
            }
        } else {
            const uint32_t head = old & ~REC_OVERFLOW;
            for (uint32_t i = head; i != kNone; i = rmw_read(&pool[i].next))
                if (rmw_read(&pool[i].atom) == atom) return;
            const uint32_t idx = atomicAdd(pool_count, 1u);
            if (idx >= pool_cap) { atomicOr(status, 1u); return; }
            publish(idx, atom, head);
            upd = REC_OVERFLOW | idx;
        }
        const uint32_t seen = atomicCAS(rec, old, upd);
        if (seen == old) return;
        old = seen;
    }
}

// One lane per (request, flagged chunk) pair; a persistent grid over device-computed work items (one workgroup's worth of pairs of
// one pass each), like the list scan: consecutive items are mostly of one pass, whose tables are staged in LDS once — the filter
// table and the confirm head table (16 KiB each) and, when they fit kConfirmPoolBytes, the comparison tables (entries, value / mask
// bytes, classes); a pass whose tables do not fit reads them from the L2-resident originals through the same (generic) pointers.
//
// Why pairs. A request-per-lane version (this round's first) looped over the request's flagged chunks, their completed windows and
// the windows' entries: 64 lanes in lockstep take the PRODUCT of each level's longest trip, every trip a chain of memory round
// trips, and a wave is as slow as its unluckiest request — measured 0.32 - 0.64 ms for the ~350k benign candidates of a 10M-request
// batch and 4 - 13 ms for the hostile stream's 11M (profiles/r4_confirm_v1_*). A pair is one chunk of text and the one or two
// windows that completed in it: the same small amount of work in every lane, and nothing to do afterwards unless something was
// confirmed.
__global__ __launch_bounds__(kConfirmThreads, 8) void confirm_kernel(ConfirmTableDev b, const uint32_t *plan) {
    __shared__ uint32_t head[kFilterEntries], ftab[kFilterEntries];  // the confirm table's head words; the pass's filter table
    __shared__ uint32_t pool[kConfirmPoolBytes / 4];                 // entries | bytes | classes of the pass (when they fit)
    constexpr uint32_t kConfirmQueue = 1024;
    __shared__ uint32_t app_cnt, app_base, app_queue[kConfirmQueue];  // the workgroup's walk-list appends of the current pass
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x == 0) app_cnt = 0;
    const uint32_t total = plan[b.count];
    const uint32_t it0 = (uint32_t)((uint64_t)total * blockIdx.x / gridDim.x), it1 = (uint32_t)((uint64_t)total * (blockIdx.x + 1) / gridDim.x);
    for (uint32_t it = it0; it < it1;) {
        uint32_t ps = 0;
        for (uint32_t lo = 0, hi = b.count; lo + 1 < hi;) {  // the pass of item `it`: the last p with plan[p] <= it (uniform)
            const uint32_t mid = (lo + hi) >> 1;
            if (plan[mid] <= it) lo = mid;
            else hi = mid;
            ps = lo;
        }
        const uint32_t first_item = plan[ps], it_end = min(it1, plan[ps + 1]);
        const ConfirmArgs a = load_descriptor(&b.c[ps]);
        __syncthreads();  // (every wave is done with the previous pass's tables)
        {
            // 16 bytes per lane and load (two 16 KiB tables)
            const uint4 *sh = reinterpret_cast<const uint4 *>(a.c_head), *sf = reinterpret_cast<const uint4 *>(a.ftable);
            constexpr uint32_t kPer = kFilterEntries / 4 / kConfirmThreads;
            uint4 vh[kPer], vf[kPer];
#pragma unroll
            for (uint32_t q = 0; q < kPer; q++) { vh[q] = sh[q * kConfirmThreads + threadIdx.x]; vf[q] = sf[q * kConfirmThreads + threadIdx.x]; }
#pragma unroll
            for (uint32_t q = 0; q < kPer; q++) {
                reinterpret_cast<uint4 *>(head)[q * kConfirmThreads + threadIdx.x] = vh[q];
                reinterpret_cast<uint4 *>(ftab)[q * kConfirmThreads + threadIdx.x] = vf[q];
            }
        }
        // the comparison tables: LDS when they fit (generic pointers: the same comparison code reads either copy)
        const uint32_t e_words = a.n_entries * 3u, b_words = a.n_bytes / 4u, c_words = a.n_class_words;
        const bool in_lds = (uint64_t)e_words + b_words + c_words <= kConfirmPoolBytes / 4;
        const ConfirmEntry *t_entries = a.c_entries;
        const uint8_t *t_bytes = a.c_bytes;
        const uint32_t *t_classes = a.c_classes;
        if (in_lds) {
            const uint32_t *se = reinterpret_cast<const uint32_t *>(a.c_entries), *sb = reinterpret_cast<const uint32_t *>(a.c_bytes);
            for (uint32_t k = threadIdx.x; k < e_words; k += kConfirmThreads) pool[k] = se[k];
            for (uint32_t k = threadIdx.x; k < b_words; k += kConfirmThreads) pool[e_words + k] = sb[k];
            for (uint32_t k = threadIdx.x; k < c_words; k += kConfirmThreads) pool[e_words + b_words + k] = a.c_classes[k];
            t_entries = reinterpret_cast<const ConfirmEntry *>(pool);
            t_bytes = reinterpret_cast<const uint8_t *>(pool + e_words);
            t_classes = pool + e_words + b_words;
        }
        __syncthreads();
        const ConfirmView cv{nullptr, t_entries, t_bytes, t_classes, a.mul, a.stride, a.init};
        const uint32_t n_p = min(*a.pair_count, a.pair_cap);
        // A lane's chain of dependent loads — pair -> the request's offsets, the chunk's text — is issued one work item AHEAD: the
        // next item's offsets and text (and the pair of the one after) travel while this item's windows are compared. (The kernel is
        // bound by those round trips, not by instructions: 65 % of its wave cycles were waits with everything issued on demand.)
        auto pair_of = [&](const uint32_t item) -> uint2 {
            const uint32_t k = (item - first_item) * kConfirmThreads + threadIdx.x;
            return (item < it_end && k < n_p) ? a.pairs[k] : make_uint2(kNone, 0u);  // (kNone: no pair for this lane)
        };
        uint2 pr_next = pair_of(it), pr_after = pair_of(it + 1u);
        uint32_t fs_next = 0u, fe_next = 0u;
        ConfirmBytes tb_next{};
        if (pr_next.x != kNone) {
            confirm_load64(reinterpret_cast<const uint8_t *>(a.off + pr_next.x), fs_next, fe_next);  // (the request's two offsets: one scattered load)
            tb_next = confirm_chunk_bytes(a.data, pr_next.y, a.total + PWAF_ARENA_PAD);
        }
        for (; it < it_end; it++) {
            const uint2 pr = pr_next;
            const bool live = pr.x != kNone;
            uint32_t r = live ? pr.x : 0u;  // the request that owns the chunk's first byte; a chunk that holds a field boundary also speaks for the next one(s)
            uint32_t fs = fs_next, fe = fe_next;
            const ConfirmBytes tb = tb_next;
            pr_next = pr_after;
            if (pr_next.x != kNone) {
                confirm_load64(reinterpret_cast<const uint8_t *>(a.off + pr_next.x), fs_next, fe_next);
                tb_next = confirm_chunk_bytes(a.data, pr_next.y, a.total + PWAF_ARENA_PAD);
            }
            pr_after = pair_of(it + 2u);
            ConfirmChunk ch{0u, 0ull};
            if (live) ch = confirm_windows_of(cv, tb, 0u, 0xFFFFFFFFu, pr.y, [&](const uint32_t bin) { return ftab[bin]; });  // (every window of the chunk: whose field it lies in is settled below)
            // the completed windows' entries, one comparison per lane and iteration (a lane advances to ITS next entry: the wave runs
            // as long as its longest lane, not the product of the two loops' longest trips)
            uint32_t mask = ch.mask, cnt = 0, j = 0, e0 = 0, pos = 0, widx = 0;
            bool walk = false;  // of request r (the one the lane is at)
            auto settle = [&](const uint32_t rq, const uint32_t need_lits, const bool wk, const bool aggregated) -> bool {
                // what the comparisons found for request rq: gap passes its literal hits call for, the walk list. Returns "append rq to
                // the walk list through the workgroup's aggregated atomic" (the lane's LAST request); an earlier request of the same
                // chunk (rare: the chunk holds a field boundary) is appended directly.
                bool w2 = wk;
                if (need_lits & a.shared_bits) w2 = true;  // a gap pass that shares the walk list learns of the request through the walk
                const uint32_t direct = need_lits & ~a.shared_bits;
                if (direct) enqueue_mask_once(a.gate_lists, a.gate_count, a.n, rq, direct, a.enq_bits, a.enq_words);
                if (!w2 || a.walk_list == nullptr) return false;
                const uint32_t bit = 1u << (rq & 31u);
                if (atomicOr(&a.walk_bits[rq >> 5], bit) & bit) return false;  // already listed
                atomicOr(&a.valid_bits[rq >> 5], bit);
                if (aggregated) return true;
                a.walk_list[atomicAdd(a.walk_count, 1u)] = rq;
                return false;
            };
            uint32_t need = 0;  // gap passes the literal hits of request r call for
            for (;;) {
                if (__ballot(j < cnt || mask != 0u) == 0) break;
                while (j >= cnt && mask != 0u) {
                    const uint32_t k = (uint32_t)__builtin_ctz(mask);
                    mask &= mask - 1u;
                    pos = pr.y * 16u + k;
                    const uint32_t bin = confirm_bin_of(ch, widx++, a.data, pos, a.mul);
                    while (pos >= fe && r + 1u < a.n) {  // the window lies in a later field of the chunk
                        settle(r, need, walk, false);
                        need = 0;
                        walk = false;
                        r++;
                        fs = fe;
                        fe = a.off[r + 1];
                    }
                    if (pos >= fe) continue;  // the bigram starts in the arena's slack (one that starts on the field's last byte may be the window of a short factor: confirm.h)
                    const uint32_t hd = head[bin];
                    e0 = hd & 0xFFFFFu;
                    cnt = hd >> 20;
                    j = 0;
                }
                if (j < cnt) {
                    const uint32_t res = in_lds ? confirm_entry<3>(t_entries, t_bytes, t_classes, e0 + j, a.data, fs, fe, pos)
                                                : confirm_entry<1>(t_entries, t_bytes, t_classes, e0 + j, a.data, fs, fe, pos);
                    j++;
                    if (res == 2u) {
                        walk = true;
                    } else if (res & 1u) {
                        const uint32_t atom = res >> 8;
                        merge_atom(a.pool, a.pool_count, a.status, a.pool_cap, a.rec + r, atom);
                        atomicOr(&a.valid_bits[r >> 5], 1u << (r & 31u));
                        if (a.colmask_local != nullptr) need |= a.colmask_local[atom];
                    }
                }
            }
            // Nothing confirmed (the common case by far): the lane is done and has written nothing.
            const bool append = live && settle(r, need, walk, true);
            // The walk list: the lane parks the request in the workgroup's LDS queue (an LDS atomic, no barrier); the queue is flushed with
            // ONE atomic on the list's length when the workgroup is done with the pass (a returned same-address atomic per request is
            // what DESIGN.md 4.1 measured at 0.5 ms per batch; a barrier per work item cost the 16 waves their independence).
            if (append) {
                const uint32_t slot = atomicAdd(&app_cnt, 1u);
                if (slot < kConfirmQueue) app_queue[slot] = r;
                else a.walk_list[atomicAdd(a.walk_count, 1u)] = r;  // (queue full: directly — rare)
            }
        }
        __syncthreads();
        {
            const uint32_t queued = min(app_cnt, kConfirmQueue);
            if (threadIdx.x == 0 && queued != 0) app_base = atomicAdd(a.walk_count, queued);
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < queued; k += kConfirmThreads) a.walk_list[app_base + k] = app_queue[k];
            __syncthreads();
            if (threadIdx.x == 0) app_cnt = 0;
        }
    }
}

int launch_scan(const ScanArgs &a, void *stream) {
    const uint32_t lds = scan_lds_bytes(a.n_hot, a.stride, a.colmask_local ? a.n_local : 0u);
    const bool wide = a.n_classes > 127;
    const int v = (a.chunks == 4 ? 2 : a.chunks == 2 ? 1 : 0) + (wide ? 3 : 0);
    const void *fns[6] = {reinterpret_cast<const void *>(scan_kernel<1, false>), reinterpret_cast<const void *>(scan_kernel<2, false>),
                          reinterpret_cast<const void *>(scan_kernel<4, false>), reinterpret_cast<const void *>(scan_kernel<1, true>),
                          reinterpret_cast<const void *>(scan_kernel<2, true>), reinterpret_cast<const void *>(scan_kernel<4, true>)};
    const void *fn = fns[v];
    if (a.n == 0) return 0;
    // at least 256 requests per wave so that work-pulling has something to balance
    uint32_t waves = (a.n + 255) / 256;
    uint32_t blocks = (waves + kScanWaves - 1) / kScanWaves;
#ifdef PWAF_PROFILING
    static const uint32_t forced_cap = getenv("PWAF_SCAN_BLOCKS") ? (uint32_t)atoi(getenv("PWAF_SCAN_BLOCKS")) : 0u;
#else
    const uint32_t forced_cap = 0;
#endif
    const uint32_t cap = forced_cap ? forced_cap : std::max(1u, a.n_cus);  // one persistent workgroup per CU: the table is staged once
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    void *args[] = {const_cast<ScanArgs *>(&a)};
    hipError_t e = hipLaunchKernel(fn, dim3(blocks), dim3(kScanThreads), args, lds, (hipStream_t)stream);
    return (int)(e != hipSuccess ? e : hipGetLastError());
}

// The batch's launch descriptors, host to device in ONE launch: `src` is page-locked host memory (device-visible: the lanes read it
// over the link, 16 bytes each), `dst` device memory; both 16-byte aligned, bytes rounded up to 16 by the caller's 256-byte parts.
// The same launch clears the batch's zeroed block (control words, candidate / visited / walk bitmaps) when the caller hands it over: a memset
// of its own was one more launch (8 us) in front of every batch.
__global__ __launch_bounds__(256) void copy_args_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16, uint4 *__restrict__ zero, uint32_t z16) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < z16; i += gridDim.x * 256u) zero[i] = make_uint4(0u, 0u, 0u, 0u);
}
int upload_args_block(const void *host_pinned, void *dev, size_t bytes, void *zero, size_t zero_bytes, void *stream) {
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16), z16 = (uint32_t)(zero_bytes / 16);  // (the zeroed block is a multiple of 256 bytes)
    if (n16 == 0 && z16 == 0) return 0;
    const uint32_t blocks = std::max(std::min<uint32_t>((n16 + 255) / 256, 64u), std::min<uint32_t>((z16 + 2047) / 2048, 1024u));
    hipLaunchKernelGGL(copy_args_kernel, dim3(std::max(1u, blocks)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)host_pinned, (uint4 *)dev, n16, (uint4 *)zero, z16);
    return (int)hipGetLastError();
}

int launch_confirm(const ConfirmArgs *host, uint32_t count, const ConfirmArgs *dev, uint32_t *plan, uint32_t n_cus, void *stream) {
    if (count == 0 || host[0].n == 0) return 0;
    if (count > 256) return (int)hipErrorInvalidValue;
    ConfirmTableDev b{dev, count};
    hipLaunchKernelGGL(confirm_plan_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, b, plan);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    // persistent grid: 2 workgroups of 1024 per CU (76 KiB of LDS each: 32 waves per CU), never more than the items a full batch could produce
    uint64_t max_items = 0;
    for (uint32_t k = 0; k < count; k++) max_items += ((uint64_t)host[k].pair_cap + kConfirmThreads - 1) / kConfirmThreads;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(max_items, (uint64_t)std::max(1u, n_cus) * 2u);
    const uint32_t *cplan = plan;
    hipLaunchKernelGGL(confirm_kernel, dim3(blocks), dim3(kConfirmThreads), 0, (hipStream_t)stream, b, cplan);
    return (int)hipGetLastError();
}

static const void *lscan_fn(bool wide) { return wide ? reinterpret_cast<const void *>(lscan_kernel<1024>) : reinterpret_cast<const void *>(lscan_kernel<512>); }

int launch_scan_gated(const ListScanArgs *host, uint32_t count, const ListScanArgs *dev, uint32_t *plan, const ListShape &shape, void *stream) {
    if (count == 0 || host[0].n == 0) return 0;
    if (count > 256) return (int)hipErrorInvalidValue;
    GatedTable b{dev, count, 0u};
#ifdef PWAF_PROFILING
    static const uint32_t async_mode = getenv("PWAF_LSCAN_ASYNC") ? (uint32_t)atoi(getenv("PWAF_LSCAN_ASYNC")) : 0u;
    b.debug = async_mode;
#endif
    // persistent grid: what the chip holds at this LDS size (the work items are split evenly over it)
    const uint32_t blocks = std::max(1u, host[0].n_cus) * shape.wg_per_cu;
    hipLaunchKernelGGL(lscan_plan_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, b, plan, blocks * (shape.threads / 64u));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const uint32_t *cplan = plan;
    uint32_t hot_bytes = list_hot_bytes(shape);
    void *args[] = {&b, &cplan, &hot_bytes};
    const void *fn = lscan_fn(shape.threads == 1024);
    e = hipLaunchKernel(fn, dim3(blocks), dim3(shape.threads), args, shape.hot_bytes + 16 + 256, (hipStream_t)stream);
    return (int)(e != hipSuccess ? e : hipGetLastError());
}

// LDS budget of the list scan: 160 KiB per CU shared by wg_per_cu workgroups. Default 3 x (48 KiB, 512 threads) = 24 waves per CU;
// PWAF_LIST_SHAPE (profiling builds) tries the others.
uint32_t list_hot_bytes(const ListShape &shape) { return shape.hot_bytes; }

ListShape list_shape(uint32_t variant) {
    switch (variant) {
        case 1: return ListShape{512, 72u * 1024u, 2};     // 16 waves per CU, 1.5x the rows
        case 2: return ListShape{1024, 144u * 1024u, 1};   // 16 waves per CU, 3x the rows
        case 3: return ListShape{512, 32u * 1024u, 4};     // 32 waves per CU
        default: return ListShape{kListThreads, kListHotBytes, 3};
    }
}

// -------------------------------------------------------------------------------------------------
// bigram prefilter
// -------------------------------------------------------------------------------------------------
// filter_kernel: the launch that streams the request bytes (kernels.h: the arena as one flat byte stream). Per input byte ONE
// independent LDS lookup — table[hash(fold(b[i]), fold(b[i+1]))] — and two vector ops: state = (state << 8) | mask, seen &= state.
// Nothing on the per-byte path depends on a previous lookup (the DFA walk it replaces chains v_lshl_add -> ds_read_u16 per byte plus
// a class lookup) and nothing depends on where requests begin or end. A zero bit in the top byte of `seen` after a 16-byte chunk =
// "some position of the chunk completed a window of some bucket"; the requests overlapping such chunks become CANDIDATES and are
// walked by the pass's DFA afterwards (lscan_kernel); every other request provably matches no pattern of the pass.
//
// Hash of a position = 12 bits of the 16-bit product fold(pair) * mul (program.h: filter_bin): two positions per v_pk_mul_lo_u16.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// STRIDE = 1: a bigram at every byte. STRIDE = 2: only at the even bytes of the arena stream (FilterArgs::stride, chosen per pass
// by the host: half the lookups; every factor is in the table once per alignment).
// The stream is laid out in ROWS: one iteration = 4 rows of 1 KiB, lane l of row q holds the 16-byte chunk at b + 1024 q + 16 l.
// Every load instruction is one contiguous KiB (round 2 gave every lane its own 64-byte segment: 64 lines per load instruction and
// two more scattered dword loads per lane for the bytes around the segment — the texture addresser, not LDS, bounded that kernel:
// 0.73 ms against 0.68 on the 10M-request batch, and stride 2 was no faster than stride 1); what a chunk needs from its
// neighbours travels through the wave instead of memory:
//   * the byte after the chunk (second half of its last bigram) = the first dword of lane l + 1's chunk (DPP wave_shl:1; lane 63:
//     lane 0 of the next row, a scalar),
//   * the shift-or state it starts from = what the three last bigrams of lane l - 1's chunk leave behind, which that lane computes
//     from its OWN lookups (two v_lshl_or_b32 — the state forgets everything older than four bigrams) and hands on (DPP
//     wave_shr:1; lane 0: lane 63 of the previous row, a scalar carried from row to row and iteration to iteration) —
//     no warm-up lookups at all.
// A row's registers are reloaded with the NEXT iteration's row as soon as its lookups are issued: 16 data registers instead of 36.
template <bool HEADS, int STRIDE>
__device__ __forceinline__ void filter_rows(const FilterArgs &a, const uint32_t blk, const uint32_t wave, const uint32_t lane) {
    const PWAF_GLOBAL unsigned char *gdata = (const PWAF_GLOBAL unsigned char *)a.data;
    const uint32_t rel = blk * kFilterWaves + wave, slab = a.slab0 + rel;
    const uint64_t base64 = (uint64_t)slab * kStreamSlab;
    if (base64 >= a.total) return;
    const uint32_t total = a.total, base0 = (uint32_t)base64, slab_end = (uint32_t)min<uint64_t>(total, base64 + kStreamSlab);
    unsigned long long *my_bits = reinterpret_cast<unsigned long long *>(a.chunk_bits) + (size_t)rel * (kStreamSlab / 1024);  // one 64-bit word per row
    uint32_t n_hit = 0;  // wave-uniform: flagged chunks of the slab
    const uint32_t mul2 = a.mul | (a.mul << 16);
    constexpr uint32_t kRow = 1024;

    auto load_row = [&](const uint32_t b, const uint32_t q) {
        const uint32_t p = b + kRow * q + 16u * lane;
        return *reinterpret_cast<const PWAF_GLOBAL u32x4_u *>(gdata + (p < total ? p : 0u));
    };
    auto dword_at = [&](const uint32_t p) {  // (wave-uniform address)
        return filter_fold4(*reinterpret_cast<const PWAF_GLOBAL uint32_t __attribute__((aligned(1))) *>(gdata + p));
    };
    // LDS byte offset of a bin = (16-bit product >> 4) * 4, for the two products of one v_pk_mul_lo_u16: ONE mask for both halves,
    // then one SDWA shift per half (v_lshrrev_b32 reading WORD_0 / WORD_1 zero-extended) — three instructions per two lookups where
    // a shift and a mask per lookup took four (a quarter of the loop's vector instructions).
    auto slot_lo = [](const uint32_t masked) {
        uint32_t r;
        asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(masked));
        return r;
    };
    auto slot_hi = [](const uint32_t masked) {
        uint32_t r;
        asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(masked));
        return r;
    };
    auto bins4 = [&](const uint32_t x, const uint32_t next, uint32_t (&mm)[4]) {  // the four bigrams that start in (case-folded) dword x
        const uint32_t z = __builtin_amdgcn_alignbit(next, x, 8);
        const uint32_t hx = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, x) * __builtin_bit_cast(u16x2, mul2));
        const uint32_t hz = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, z) * __builtin_bit_cast(u16x2, mul2));
#ifdef PWAF_PROFILING
        if (a.debug & 1u) { mm[0] = hx | 0xFF000000u; mm[1] = hz | 0xFF000000u; mm[2] = (hx >> 3) | 0xFF000000u; mm[3] = (hz >> 5) | 0xFF000000u; return; }
#endif
        const uint32_t kx = hx & 0xFFF0FFF0u, kz = hz & 0xFFF0FFF0u;
        mm[0] = *reinterpret_cast<lds_u32_ptr>((uintptr_t)slot_lo(kx));
        mm[1] = *reinterpret_cast<lds_u32_ptr>((uintptr_t)slot_lo(kz));
        mm[2] = *reinterpret_cast<lds_u32_ptr>((uintptr_t)slot_hi(kx));
        mm[3] = *reinterpret_cast<lds_u32_ptr>((uintptr_t)slot_hi(kz));
    };
    auto bins2 = [&](const uint32_t x, uint32_t (&mm)[2]) {  // the two bigrams at the even bytes of (case-folded) dword x
        const uint32_t hx = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, x) * __builtin_bit_cast(u16x2, mul2));
#ifdef PWAF_PROFILING
        if (a.debug & 1u) { mm[0] = hx | 0xFF000000u; mm[1] = (hx >> 3) | 0xFF000000u; return; }
#endif
        const uint32_t kx = hx & 0xFFF0FFF0u;
        mm[0] = *reinterpret_cast<lds_u32_ptr>((uintptr_t)slot_lo(kx));
        mm[1] = *reinterpret_cast<lds_u32_ptr>((uintptr_t)slot_hi(kx));
    };
    auto push = [](uint32_t &st, const uint32_t m) { asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(st) : "v"(st), "v"(m)); };

    // heads: the wave walks the offsets column alongside the bytes; rq = first request that starts at or after the current byte.
    // Software-pipelined so that nothing on this path waits for memory: the offsets of the next 128 requests are requested one
    // iteration ahead; the first 16 bytes of the requests that start inside the current iteration (bytes this wave has just
    // streamed: cache hits) are requested BEFORE the rows are looked up and compared after them. (The first version — a loop of
    // dependent loads per iteration — cost 0.20 ms of a 0.88 ms launch.)
    uint32_t rq = 0, ho[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, hn[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    auto load_offs = [&](const uint32_t r0) {
        // off[n] = the arena's size, which no iteration's limit exceeds: an index clamped to n reads as "starts beyond". A request's
        // end is its right-hand neighbour's start: one load per request, the neighbour's through DPP (lane 63: the next block's first).
        ho[0] = a.off[min(r0 + lane, a.n)];
        ho[1] = a.off[min(r0 + 64u + lane, a.n)];
        const uint32_t e1 = a.off[min(r0 + 128u, a.n)];  // (wave-uniform address)
        hn[0] = (uint32_t)__builtin_amdgcn_update_dpp(__builtin_amdgcn_readfirstlane((int)ho[1]), (int)ho[0], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
        hn[1] = (uint32_t)__builtin_amdgcn_update_dpp(__builtin_amdgcn_readfirstlane((int)e1), (int)ho[1], 0x130, 0xF, 0xF, false);
    };
    auto head_record = [&](const u32x4 w, const uint32_t flen) {
        uint32_t hrec = 0;
#pragma unroll
        for (int hq = 0; hq < 2; hq++) {
            if ((uint32_t)hq < a.n_heads) {
                const uint32_t diff = ((w.x ^ a.head_w[hq][0]) & a.head_m[hq][0]) | ((w.y ^ a.head_w[hq][1]) & a.head_m[hq][1]) |
                                      ((w.z ^ a.head_w[hq][2]) & a.head_m[hq][2]) | ((w.w ^ a.head_w[hq][3]) & a.head_m[hq][3]);
                const uint32_t hl = a.head_len[hq] & 0xFFu;
                const bool len_ok = (a.head_len[hq] >> 8) ? flen == hl : flen >= hl;
                if (diff == 0 && len_ok) hrec |= a.head_code[hq];
            }
        }
        return hrec;
    };
    const bool heads = HEADS && a.n_heads != 0;
    if (heads) {
        // lower bound of base0 in off[0, n): a 64-ary search like resolve_kernel's — every lane probes one of 64 evenly spaced offsets, a ballot narrows
        // the range 64-fold (4 dependent loads for 10M requests where the binary search took 23)
        uint32_t lo = 0, hi = a.n;  // the answer lies in [lo, hi]
        while (lo < hi) {
            const uint32_t span = hi - lo, step = (span + 63) / 64;
            const uint32_t probe = lo + lane * step;
            const bool ok = probe >= hi || a.off[probe] >= base0;  // monotone in the lane index
            const unsigned long long m = __ballot(ok);
            if (m == 0) {  // all 64 probes fail: the answer lies beyond the last one
                lo += 63u * step + 1u;
                continue;
            }
            const uint32_t f = (uint32_t)__builtin_ctzll(m);
            const uint32_t nh = min(hi, lo + f * step);  // probe f holds (or is past the range)
            lo = f == 0 ? lo : lo + (f - 1u) * step + 1u;  // probe f - 1 fails
            hi = nh;
        }
        rq = lo;
        load_offs(rq);
    }

    // the state the slab's first chunk starts from: the three sampled bigrams before the slab (none at the very start of the arena)
    uint32_t carry = a.init;  // wave-uniform
    if (STRIDE == 1) {
        if (base0 >= 4) {
            uint32_t mw[4];
            bins4(dword_at(base0 - 4), dword_at(base0), mw);
            push(carry, mw[1]); push(carry, mw[2]); push(carry, mw[3]);
        }
    } else if (base0 >= 8) {
        uint32_t m0[2], m1[2];
        bins2(dword_at(base0 - 8), m0);
        bins2(dword_at(base0 - 4), m1);
        push(carry, m0[1]); push(carry, m1[0]); push(carry, m1[1]);
    }
    carry = __builtin_amdgcn_readfirstlane(carry);
    // the dword after the slab (second byte of the slab's last bigram)
    const uint32_t after = (STRIDE == 1 && slab_end < total) ? __builtin_amdgcn_readfirstlane(dword_at(slab_end)) : 0u;

    u32x4 w[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) w[q] = load_row(base0, q);
    for (uint32_t b = base0; b < slab_end; b += kStreamIter) {
        const bool more = b + kStreamIter < slab_end;  // wave-uniform
        u32x4 hw[2];
        uint32_t hlen[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
        uint32_t rq_here = rq;
#ifdef PWAF_PROFILING
        if (heads && !(a.debug & 4u)) {
#else
        if (heads) {
#endif
            const uint32_t lim = min(b + kStreamIter, slab_end);
            const bool in0 = ho[0] < lim;
            const uint32_t c0 = (uint32_t)__builtin_popcountll(__ballot(in0));
            if (in0) {  // (only the lanes that hold a request of this iteration issue an address)
                hw[0] = *reinterpret_cast<const PWAF_GLOBAL u32x4_u *>(gdata + ho[0]);
                hlen[0] = hn[0] - ho[0];
            }
            uint32_t c1 = 0;
            if (c0 == 64) {  // (wave-uniform: the second gather is issued only when more than 64 requests start in these 4 KiB)
                const bool in1 = ho[1] < lim;
                c1 = (uint32_t)__builtin_popcountll(__ballot(in1));
                if (in1) {
                    hw[1] = *reinterpret_cast<const PWAF_GLOBAL u32x4_u *>(gdata + ho[1]);
                    hlen[1] = hn[1] - ho[1];
                }
            }
            rq += c0 + c1;
            if (c0 + c1 == 128) {
                for (;;) {  // more than 128 requests start within 4 KiB (very short fields): the rest synchronously
                    const uint32_t idx = rq + lane;
                    const uint32_t s = a.off[min(idx, a.n)];
                    const bool in = s < lim;
                    const uint32_t cnt = (uint32_t)__builtin_popcountll(__ballot(in));
                    if (in) {
                        const uint32_t hrec = head_record(*reinterpret_cast<const PWAF_GLOBAL u32x4_u *>(gdata + s), a.off[idx + 1] - s);
                        if (hrec) a.rec[idx] = hrec;
                    }
                    rq += cnt;
                    if (cnt < 64) break;
                }
            }
            load_offs(rq);  // for the next iteration
        }

        unsigned long long hm[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t x0 = filter_fold4(w[q].x), x1 = filter_fold4(w[q].y), x2 = filter_fold4(w[q].z), x3 = filter_fold4(w[q].w);  // (program.h: letters lose their case, nothing below 0x40 moves)
            uint32_t st, seen = 0xFFFFFFFFu, tail;
            if (STRIDE == 1) {
                // lane 63's next dword: lane 0 of the next row (row 0 already holds the NEXT iteration's chunk; the slab's last row: `after`)
                const uint32_t first_next = (q == 3 && !more) ? after : filter_fold4(__builtin_amdgcn_readfirstlane(w[(q + 1) & 3].x));
                const uint32_t x4 = (uint32_t)__builtin_amdgcn_update_dpp((int)first_next, (int)x0, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
                uint32_t m[16];
                {
                    uint32_t mm[4];
                    bins4(x0, x1, mm); m[0] = mm[0]; m[1] = mm[1]; m[2] = mm[2]; m[3] = mm[3];
                    bins4(x1, x2, mm); m[4] = mm[0]; m[5] = mm[1]; m[6] = mm[2]; m[7] = mm[3];
                    bins4(x2, x3, mm); m[8] = mm[0]; m[9] = mm[1]; m[10] = mm[2]; m[11] = mm[3];
                    bins4(x3, x4, mm); m[12] = mm[0]; m[13] = mm[1]; m[14] = mm[2]; m[15] = mm[3];
                }
#ifdef PWAF_PROFILING
                if (more && !(a.debug & 2u)) w[q] = load_row(b + kStreamIter, q);
#else
                if (more) w[q] = load_row(b + kStreamIter, q);
#endif
                tail = m[13]; push(tail, m[14]); push(tail, m[15]);
                st = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)tail, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
#pragma unroll
                for (int i = 0; i < 16; i++) { push(st, m[i]); seen &= st; }
            } else {
                uint32_t m[8];
                {
                    uint32_t mm[2];
                    bins2(x0, mm); m[0] = mm[0]; m[1] = mm[1];
                    bins2(x1, mm); m[2] = mm[0]; m[3] = mm[1];
                    bins2(x2, mm); m[4] = mm[0]; m[5] = mm[1];
                    bins2(x3, mm); m[6] = mm[0]; m[7] = mm[1];
                }
#ifdef PWAF_PROFILING
                if (more && !(a.debug & 2u)) w[q] = load_row(b + kStreamIter, q);
#else
                if (more) w[q] = load_row(b + kStreamIter, q);
#endif
                tail = m[5]; push(tail, m[6]); push(tail, m[7]);
                st = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)tail, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
#pragma unroll
                for (int i = 0; i < 8; i++) { push(st, m[i]); seen &= st; }
            }
            carry = __builtin_amdgcn_readlane(tail, 63);
            const bool hit = ((~seen) & 0xFF000000u) != 0 && b + kRow * q + 16u * lane < total;
            hm[q] = __ballot(hit);  // bit l = the row's chunk l: the row's word of the pass's chunk bitmap as it is
            __builtin_amdgcn_sched_barrier(0);
        }
        // the iteration's four words: one 32-byte store
        if (lane < 4) my_bits[(b - base0) / kRow + lane] = lane == 0 ? hm[0] : lane == 1 ? hm[1] : lane == 2 ? hm[2] : hm[3];
        n_hit += (uint32_t)(__builtin_popcountll(hm[0]) + __builtin_popcountll(hm[1]) + __builtin_popcountll(hm[2]) + __builtin_popcountll(hm[3]));

        if (heads) {
#pragma unroll
            for (uint32_t q = 0; q < 2; q++) {
                if (hlen[q] != 0xFFFFFFFFu) {
                    const uint32_t hrec = head_record(hw[q], hlen[q]);
                    if (hrec) a.rec[rq_here + 64u * q + lane] = hrec;
                }
            }
        }
    }
    // (the arena's last slab: the words of the iterations it does not have read as "no chunk flagged")
    for (uint32_t wd = (slab_end - base0 + kStreamIter - 1) / kStreamIter * 4 + lane; wd < kStreamSlab / kRow; wd += 64) my_bits[wd] = 0;
    if (lane == 0) {
        a.sub_count[rel] = n_hit;
        // where the slab's pairs begin in the pass's pair list: one returned atomic per slab, here at the slab's end (round 6; until then a scan
        // kernel of its own between this launch and resolve_kernel: 7 us + a launch gap on the critical path. The same atomic taken by
        // resolve_kernel's waves as they START held every one of them up for its turn — here the slabs end spread over the launch). The order
        // of a pass's pairs no longer follows the arena; nothing reads it as ordered.
        if (a.pairs != nullptr) a.pair_base[rel] = n_hit ? atomicAdd(a.pair_count, n_hit) : 0u;
    }
}

// ONE launch for every filtered pass, both strides: stride-1 passes are bound by the LDS lookups, stride-2 passes by HBM, so their
// workgroups are INTERLEAVED in proportion (block i is the floor(i * n1 / N)-th stride-1 block if that count steps at i, else the
// next stride-2 block) and every CU holds both kinds at any time — one launch per stride ran them one after the other, each
// phase leaving the other resource idle.
template <bool HEADS>
__global__ __launch_bounds__(kFilterWaves * 64) void filter_kernel(FilterMix M) {
    extern __shared__ __align__(16) unsigned char lds[];
    __builtin_amdgcn_s_setprio(3);
    const uint32_t n_all = M.blocks1 + M.blocks2;
    const uint32_t before = (uint32_t)(((uint64_t)blockIdx.x * M.blocks1) / n_all), upto = (uint32_t)(((uint64_t)(blockIdx.x + 1) * M.blocks1) / n_all);
    const bool one = upto > before;  // a stride-1 block
    const uint32_t blk = one ? before : blockIdx.x - before;
    const FilterArgs *tab = one ? M.f1 : M.f2;
    // which pass of its class this workgroup belongs to (uniform: the last pass whose first_block is <= blk)
    uint32_t k = 0;
    for (uint32_t lo = 0, hi = one ? M.count1 : M.count2; lo + 1 < hi;) {
        const uint32_t mid = (lo + hi) >> 1;
        if (blk >= tab[mid].first_block) lo = mid;
        else hi = mid;
        k = lo;
    }
    const FilterArgs a = load_descriptor(&tab[k]);
    const uint32_t tid = threadIdx.x, wave = wave_index(), lane = tid & 63;
    // (Bank-private replicas of the table — 4 copies, each lane group of 8 with 8 banks of its own — were measured SLOWER: 1.19 ms
    // against 0.89 ms. A ds_read_b32 takes as many cycles as its most loaded bank over all 32 lanes, and the maximum over four
    // groups of 8-in-8 is hardly below 32-in-32, while 64 KiB per workgroup halves the occupancy.)
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.table);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (uint32_t i = tid; i < kFilterEntries / 4; i += kFilterWaves * 64) dst[i] = src[i];
    }
    __syncthreads();
    // (no static LDS in this kernel: the table starts at LDS address 0 and lookups use plain integer addresses)
    if ((uint32_t)(uintptr_t)(PWAF_LDS unsigned char *)lds != 0u) __builtin_trap();
    if (one) filter_rows<HEADS, 1>(a, blk - a.first_block, wave, lane);
    else filter_rows<HEADS, 2>(a, blk - a.first_block, wave, lane);
}

template <uint32_t PARTS>
__global__ void resolve_kernel(FilterTable B);  // (defined below, next to the wave scan it uses)

// bitcount_kernel / compact_kernel: candidate bitmap -> dense ascending request list. Two launches: candidates per workgroup
// (kCompactWords bitmap words each), then every workgroup sums the counts before it (a few hundred values) and writes its part.
__global__ __launch_bounds__(256) void bitcount_kernel(FilterTable B) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint32_t red[256];
    const FilterArgs a = load_descriptor(&B.f[blockIdx.y]);
    const uint32_t words = (a.n + 31) / 32, w0 = blockIdx.x * kCompactWords;
    if (w0 >= words) return;
    uint32_t c = 0;
    for (uint32_t w = w0 + threadIdx.x; w < min(words, w0 + kCompactWords); w += 256) c += (uint32_t)__builtin_popcount(a.bitmap[w]);
    red[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t h = 128; h > 0; h >>= 1) {
        if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.block_count[blockIdx.x] = red[0];
}

__global__ void compact_kernel(FilterTable B);  // (defined below, next to the wave scan it uses)

static uint32_t filter_blocks(const FilterArgs *host, uint32_t count, bool *heads) {
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t slabs = (uint32_t)(((uint64_t)host[k].total + kStreamSlab - 1) / kStreamSlab) - host[k].slab0;
        blocks += (slabs + kFilterWaves - 1) / kFilterWaves;
        *heads = *heads || host[k].n_heads != 0;
    }
    return blocks;
}

int launch_filter(const FilterArgs *host1, uint32_t count1, const FilterArgs *dev1, const FilterArgs *host2, uint32_t count2, const FilterArgs *dev2, void *stream) {
    // the passes of each stride, first_block numbered within their class by the caller
    bool heads = false;
    FilterMix m{dev1, dev2, count1, count2, filter_blocks(host1, count1, &heads), filter_blocks(host2, count2, &heads)};
    if (m.blocks1 + m.blocks2 == 0) return 0;
    void *args[] = {&m};
    const void *fn = heads ? reinterpret_cast<const void *>(filter_kernel<true>) : reinterpret_cast<const void *>(filter_kernel<false>);
    hipError_t e = hipLaunchKernel(fn, dim3(m.blocks1 + m.blocks2), dim3(kFilterWaves * 64), args, kFilterEntries * 4, (hipStream_t)stream);
    return (int)(e != hipSuccess ? e : hipGetLastError());
}

int launch_resolve(const FilterArgs *host, uint32_t count, const FilterArgs *dev, void *stream) {
    uint32_t max_slabs = 0;
    uint64_t all_slabs = 0;
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t slabs = (uint32_t)(((uint64_t)host[k].total + kStreamSlab - 1) / kStreamSlab) - host[k].slab0;
        max_slabs = max(max_slabs, slabs);
        all_slabs += slabs;
    }
    if (count == 0 || max_slabs == 0) return 0;
    FilterTable t{dev, count};
    void *args[] = {&t};
    // four waves per slab when one per slab would leave the chip's wave slots empty (resolve_kernel: PARTS)
    const char *forced = getenv("PWAF_RESOLVE_PARTS");  // (measurement / test switch, read per launch: 1 or 4)
    const bool split = forced ? atoi(forced) == 4 : all_slabs < 4096u;  // (measured: 1.25M requests = 2 700 slabs 55 -> 40 us; 2.5M = 5 400 slabs no gain; 10M = 21 600 slabs 123 -> 190 us)
    hipError_t e = split ? hipLaunchKernel(reinterpret_cast<const void *>(resolve_kernel<4>), dim3(max_slabs, count), dim3(256), args, 0, (hipStream_t)stream)
                         : hipLaunchKernel(reinterpret_cast<const void *>(resolve_kernel<1>), dim3((max_slabs + 3) / 4, count), dim3(256), args, 0, (hipStream_t)stream);
    return (int)(e != hipSuccess ? e : hipGetLastError());
}

int launch_compact(const FilterArgs *host, uint32_t count, const FilterArgs *dev, void *stream) {
    if (count == 0 || host[0].n == 0) return 0;
    const uint32_t words = (host[0].n + 31) / 32, n_blocks = (words + kCompactWords - 1) / kCompactWords;
    FilterTable t{dev, count};
    void *args[] = {&t};
    hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(bitcount_kernel), dim3(n_blocks, count), dim3(256), args, 0, (hipStream_t)stream);
    if (e == hipSuccess) e = hipLaunchKernel(reinterpret_cast<const void *>(compact_kernel), dim3(n_blocks, count), dim3(256), args, 0, (hipStream_t)stream);
    return (int)(e != hipSuccess ? e : hipGetLastError());
}

// fcmp_kernel: field-against-field atoms (one string field of a request against another: ==, contains, starts_with, ends_with, or a
// comparison of their lengths). One lane per request; rare in rule sets, so plain byte loops (both fields were just streamed: cache hits).
__global__ __launch_bounds__(256) void fcmp_kernel(FcmpArgs a) {
    const SlowCtx ctx{nullptr, nullptr, a.pool, a.pool_count, a.status, a.pool_cap};
    for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < a.n; r += gridDim.x * 256u) {
        Hits h{0, 0, kNone};
        for (uint32_t k = 0; k < a.n_atoms; k++) {
            const uint32_t op = a.atoms[k] & 0xFFu, sa = (a.atoms[k] >> 8) & 0xFFu, sb = (a.atoms[k] >> 16) & 0xFFu;
            const uint32_t a0 = a.off[sa][r], la = a.off[sa][r + 1] - a0, b0 = a.off[sb][r], lb = a.off[sb][r + 1] - b0;
            const uint8_t *x = a.data[sa] + a0, *y = a.data[sb] + b0;
            bool holds = false;
            auto same = [&](const uint8_t *p, const uint8_t *q, uint32_t len) {
                for (uint32_t i = 0; i < len; i++)
                    if (p[i] != q[i]) return false;
                return true;
            };
            switch (op) {
                case FC_EQ: holds = la == lb && same(x, y, la); break;
                case FC_STARTS: holds = la >= lb && same(x, y, lb); break;
                case FC_ENDS: holds = la >= lb && same(x + (la - lb), y, lb); break;
                case FC_CONTAINS:
                    if (la >= lb)
                        for (uint32_t s = 0; s + lb <= la && !holds; s++) holds = same(x + s, y, lb);
                    break;
                case FC_LEN_EQ: holds = la == lb; break;
                case FC_LEN_LT: holds = la < lb; break;
                default: holds = la <= lb; break;
            }
            if (holds) h = record_atom(ctx, k, h);
        }
        a.rec[r] = h.ovf != kNone ? (REC_OVERFLOW | h.ovf) : (h.a0 | (h.a1 << 15));
    }
}

int launch_fcmp(const FcmpArgs &a, void *stream) {
    if (a.n == 0 || a.n_atoms == 0) return 0;
    hipLaunchKernelGGL(fcmp_kernel, dim3(std::min<uint32_t>((a.n + 255) / 256, 4096u)), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// residual_kernel: the rules no column form exists for (residual.h), interpreted with one lane per request. The interpreter's value
// stack and heap live in the lane's private memory: this is the slow path by design (a rule set without such rules never launches it).
__global__ __launch_bounds__(128) void residual_kernel(ResidualArgs a) {
    const SlowCtx ctx{nullptr, nullptr, a.pool, a.pool_count, a.status, a.pool_cap};
    rvm::Machine m;
    m.blob = a.blob;
    m.h = reinterpret_cast<const rvm::Header *>(a.blob);
    m.q.data = a.data;
    m.q.off = a.off;
    for (uint32_t r = blockIdx.x * 128u + threadIdx.x; r < a.n; r += gridDim.x * 128u) {
        m.q.r = r;
        m.q.ip = a.ip + (size_t)r * 16;
        m.q.v6 = a.ip_is_v6[r];
        m.q.port = a.port[r];
        uint32_t asn = 0, country = (uint32_t)'X' | ((uint32_t)'X' << 8);  // the default record {0, "XX"} (http_listener.rs:148-156)
        if (a.asn != nullptr) {
            asn = a.asn[r];
            const uint32_t cc = a.country[r], c0 = (cc & 0xFFu) - 'A', c1 = (cc >> 8) - 'A';
            if (c0 < 26u && c1 < 26u) country = cc;  // (invalid input is treated as "XX", like the attribute kernel does)
        } else if (a.has_geo) {
            // GeoipDB::lookup (pingoo/geoip.rs:73-91): loopback / multicast are "not found"
            const uint8_t *ip = m.q.ip;
            bool walk;
            if (!m.q.v6) walk = !(ip[0] == 127u || (ip[0] & 0xF0u) == 0xE0u);
            else {
                bool loopback = ip[15] == 1;
                for (int k = 0; k < 15; k++) loopback = loopback && ip[k] == 0;
                walk = !(loopback || ip[0] == 0xFFu);
            }
            if (walk) {
                uint32_t e = (m.q.v6 ? a.geo_root6 : a.geo_root4)[((uint32_t)ip[0] << 8) | ip[1]];
                for (uint32_t k = 2; !(e & TRIE_LEAF); k++) e = a.geo_nodes[(size_t)e * 256 + ip[k]];
                const GeoRec g = a.geo_recs[e & ~TRIE_LEAF];
                asn = g.asn;
                country = g.country;
            }
        }
        m.q.asn = asn;
        m.q.country = country;
        Hits h{0, 0, kNone};
        for (uint32_t k = 0; k < a.n_rules; k++) {
            const uint32_t res = rvm::run_rule(m, k);
            if (res == 1u) h = record_atom(ctx, k, h);
            // execution errors are COUNTED per rule (the reference logs each one, pingoo/rules.rs:41-45; here: pwaf_engine_rule_errors):
            // one atomic per wave and rule that saw any
            const unsigned long long em = __ballot(res == 2u);
            if (em != 0 && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(em)) atomicAdd(&a.rule_errors[k], (unsigned long long)__builtin_popcountll(em));
        }
        a.rec[r] = h.ovf != kNone ? (REC_OVERFLOW | h.ovf) : (h.a0 | (h.a1 << 15));
    }
}

int launch_residual(const ResidualArgs &a, void *stream) {
    if (a.n == 0 || a.n_rules == 0) return 0;
    hipLaunchKernelGGL(residual_kernel, dim3(std::min<uint32_t>((a.n + 127) / 128, 8192u)), dim3(128), 0, (hipStream_t)stream, a);
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
// LDS per wave: the column file (one 64-request word per atom), a bitmap of non-zero columns, a bitmap of candidate rules
// and the ordered candidate list.
// Bitwise OR over the 64 lanes, in the vector ALU (DPP row shifts + the two cross-row broadcasts): no LDS, no memory.
__device__ __forceinline__ uint32_t wave_or(uint32_t x) {
#define PWAF_DPP_OR(ctrl, rows) x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, (ctrl), (rows), 0xF, true)
    PWAF_DPP_OR(0x111, 0xF);  // row_shr:1
    PWAF_DPP_OR(0x112, 0xF);  // row_shr:2
    PWAF_DPP_OR(0x114, 0xF);  // row_shr:4
    PWAF_DPP_OR(0x118, 0xF);  // row_shr:8   -> lane 15 of each row holds its row
    PWAF_DPP_OR(0x142, 0xA);  // row_bcast:15 into rows 1 and 3
    PWAF_DPP_OR(0x143, 0xC);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave
#undef PWAF_DPP_OR
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// A wave-uniform 64-bit mask into ONE lane of a register pair (v_writelane_b32): `x = lane == l ? m : x` without the compare of the
// lane id, the two selects and the moves of the scalar pair into vector registers — two vector instructions instead of seven where a
// kernel is bound by vector-ALU issue (attr_kernel: one such parking per present membership bit and per comparison atom). gfx9 lets
// a VALU instruction read ONE scalar register, M0 not counted: the lane select travels in M0. (No compiler builtin for v_writelane.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // (M0 is a reserved register: naming it as clobbered is what is meant)
__device__ __forceinline__ void park64(uint32_t &lo, uint32_t &hi, const unsigned long long m, const uint32_t l) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %4, m0"
                 : "+v"(lo), "+v"(hi)
                 : "s"((uint32_t)m), "s"(l), "s"((uint32_t)(m >> 32))
                 : "m0");
}
#pragma clang diagnostic pop

// Inclusive prefix sum over the 64 lanes, same DPP network (no LDS round trips, unlike __shfl_up).
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x) {
#define PWAF_DPP_ADD(ctrl, rows) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, (ctrl), (rows), 0xF, true)
    PWAF_DPP_ADD(0x111, 0xF);  // row_shr:1
    PWAF_DPP_ADD(0x112, 0xF);  // row_shr:2
    PWAF_DPP_ADD(0x114, 0xF);  // row_shr:4
    PWAF_DPP_ADD(0x118, 0xF);  // row_shr:8   -> inclusive scan inside each row of 16
    PWAF_DPP_ADD(0x142, 0xA);  // row_bcast:15: rows 1 and 3 add the total of the row before
    PWAF_DPP_ADD(0x143, 0xC);  // row_bcast:31: rows 2 and 3 add the total of rows 0-1
#undef PWAF_DPP_ADD
    return x;
}

// resolve_kernel: one wave per slab turns the slab's flagged chunks into candidate REQUESTS. Request-driven: the flagged 16-byte
// chunks of the slab become a bitmap in LDS (8192 bits) with per-word prefix counts, then the wave walks the requests that overlap
// the slab — 64 per step, offsets read coalesced — and a request is a candidate when a flagged chunk lies within its bytes extended
// by what a window may reach back (three sampled bigrams) and forward (one byte): two rank queries. Slabs without a hit leave at
// once. (The first version searched the offsets per HIT: 23 dependent loads each and an atomic per marked request — 0.07 ms on
// benign traffic, 1.4 ms when most segments are flagged.)
// PARTS: waves per slab (round 6). A slab's wave walks every request that overlaps its 128 KiB — 64 per step, one round trip each; when the batch's
// slabs do not fill the chip (a 1.25M-request share: 2 700) the launch takes what its longest walk takes, 44 us. With PARTS = 4 every wave ranks the
// whole slab's bitmap (1 KiB) but walks only the requests of ITS quarter and lists only the flagged chunks of its quarter — their places in the pair
// list are ranks within the slab, as before.
template <uint32_t PARTS>
__global__ __launch_bounds__(256) void resolve_kernel(FilterTable B) {
    __builtin_amdgcn_s_setprio(3);
    constexpr uint32_t kChunks = kStreamSlab / 16, kWords = kChunks / 32;  // 8192 chunk bits = 256 words per slab
    constexpr uint32_t kResolveSparse = 128;  // flagged chunks up to which a slab is resolved chunk by chunk (two rounds of 64 lanes)
    __shared__ uint32_t s_bits[4][kWords], s_rank[4][kWords];
    const FilterArgs *pa = &B.f[blockIdx.y];
    if ((uint64_t)(pa->slab0 + blockIdx.x * 4 / PARTS) * kStreamSlab >= pa->total) return;  // (the whole workgroup is past the pass's last slab)
    const FilterArgs a = load_descriptor(pa);
    if (a.dense_flag != nullptr && *a.dense_flag > a.dense_thresh) return;  // (the pass is walked whole this batch: no pairs, no records to reset)
    const uint32_t wave = wave_index(), rel = (blockIdx.x * 4 + wave) / PARTS, part = (blockIdx.x * 4 + wave) % PARTS, slab = a.slab0 + rel, lane = threadIdx.x & 63;
    constexpr uint32_t kPartChunks = kChunks / PARTS;
    if ((uint64_t)slab * kStreamSlab >= a.total) return;
    const uint32_t cnt = a.sub_count[rel];
    if (cnt == 0) return;
    // a slab with few flagged chunks is resolved chunk by chunk (below) by ONE wave, whatever PARTS: the others of its four leave
    const bool sparse = a.pairs != nullptr && cnt <= kResolveSparse;
    if (sparse && part != 0u) return;
    const uint32_t lo_c = sparse ? 0u : part * kPartChunks, hi_c = sparse ? kChunks : lo_c + kPartChunks;  // the wave's chunks of the slab, slab-relative
    const uint32_t *slab_bits = a.chunk_bits + (size_t)rel * kWords;
    uint32_t *bits = s_bits[wave], *rank = s_rank[wave];
    // 1. the slab's part of the pass's chunk bitmap (filter_kernel wrote it, one 64-bit word per KiB row), prefix popcounts per word
#pragma unroll
    for (uint32_t q = 0; q < kWords / 64; q++) bits[q * 64 + lane] = slab_bits[q * 64 + lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    {
        uint32_t w[kWords / 64], mine = 0;
#pragma unroll
        for (uint32_t q = 0; q < kWords / 64; q++) {
            w[q] = bits[lane * (kWords / 64) + q];
            mine += (uint32_t)__builtin_popcount(w[q]);
        }
        uint32_t before = wave_scan_add(mine) - mine;
#pragma unroll
        for (uint32_t q = 0; q < kWords / 64; q++) {
            rank[lane * (kWords / 64) + q] = before;
            before += (uint32_t)__builtin_popcount(w[q]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    auto rank_of = [&](const uint32_t c) {  // flagged chunks among the slab's chunks [0, c), c <= kChunks
        if (c >= kChunks) return rank[kWords - 1] + (uint32_t)__builtin_popcount(bits[kWords - 1]);
        return rank[c >> 5] + (uint32_t)__builtin_popcount(bits[c >> 5] & ((1u << (c & 31u)) - 1u));
    };
    // 2. the requests overlapping the slab's bytes (extended by the reach of a window), in order
    const uint32_t back = 3u * a.stride;
    const uint64_t slab_b0 = (uint64_t)slab * kStreamSlab, b0 = slab_b0 + (uint64_t)lo_c * 16u;  // (b0: the first byte of the wave's part)
    const uint32_t c_first = (uint32_t)(slab_b0 / 16);  // global index of the slab's first chunk
    // first request with off[r + 1] + back > b0: a 64-ary search — every lane probes one of 64 evenly spaced offsets, a ballot
    // narrows the range 64-fold per step (4 dependent loads for 10M requests; the binary search this replaces took 23, about
    // 10 us of every wave's life in a kernel that is nothing but latency)
    uint32_t lo = 0, hi = a.n;  // the answer lies in [lo, hi]; index a.n stands for "none" (treated as satisfying the predicate)
    while (lo < hi) {
        const uint32_t span = hi - lo, step = (span + 63) / 64;
        const uint32_t probe = lo + lane * step;
        const bool ok = probe >= hi || (uint64_t)a.off[probe + 1] + back > b0;  // monotone in the lane index
        const unsigned long long m = __ballot(ok);
        if (m == 0) {  // all 64 probes fail: the answer lies beyond the last one
            lo += 63u * step + 1u;
            continue;
        }
        const uint32_t f = (uint32_t)__builtin_ctzll(m);
        const uint32_t nh = min(hi, lo + f * step);  // probe f holds (or is past the range)
        lo = f == 0 ? lo : lo + (f - 1u) * step + 1u;  // probe f - 1 fails
        hi = nh;
    }
    const uint64_t b1 = slab_b0 + (uint64_t)hi_c * 16u;  // requests starting at or beyond b1 + 16 cannot be reached by this part's chunks
    // A pass with a confirm tier lists its flagged chunks instead of marking candidates: every flagged chunk of the slab becomes ONE pair
    // {the request that owns the chunk's first byte, chunk}. The slab's pairs take a contiguous part of the pass's pair list — it begins
    // where filter_kernel's atomic on the list's length put it as the slab ended (pair_base; the same atomic taken HERE, ~20k returned
    // same-address atomics as these waves start, held every wave up for its turn) — and a chunk's place in it is its rank among the
    // slab's flagged chunks (the prefix counts above): no second walk.
    const uint32_t pair_base = a.pairs != nullptr ? a.pair_base[rel] : 0u;
    const uint32_t begin = __builtin_amdgcn_readfirstlane(a.off[0]);  // != 0: a slab view of a larger arena
    // A slab view keeps absolute arena positions and the filter streams whole slabs: what lies before off[0] in the view's first slab —
    // another view's requests, stale staging bytes — may have flagged chunks that NO request of this batch owns. Their slots of the pair
    // list are filled with "no pair" here (confirm_kernel skips those; left unwritten they would be read as stale {request, chunk}
    // pairs: ADVICE r4). Only the chunks that end at or before off[0]: the chunk that holds off[0] belongs to the first request.
    if (part == 0 && a.pairs != nullptr && (uint64_t)begin > slab_b0) {
        const uint32_t n_before = min((uint32_t)(((uint64_t)begin - slab_b0) >> 4), kChunks);  // chunks [0, n_before) of this slab lie wholly before off[0]
        for (uint32_t w = lane; w * 32u < n_before; w += 64) {
            uint32_t bw = bits[w];
            if (n_before - w * 32u < 32u) bw &= (1u << (n_before - w * 32u)) - 1u;
            const uint32_t word = bits[w];
            while (bw) {
                const uint32_t bit = (uint32_t)__builtin_ctz(bw);
                bw &= bw - 1u;
                const uint32_t at = pair_base + rank[w] + (uint32_t)__builtin_popcount(word & ((1u << bit) - 1u));
                if (at < a.pair_cap) a.pairs[at] = make_uint2(kNone, 0u);
            }
        }
    }
    // CHUNK-DRIVEN (round 6): a slab with few flagged chunks — a benign 10M-request batch of the 1k-rule set leaves 404k pairs in 21 600 slabs, 19 per
    // slab — does not walk its 1 300 - 2 900 requests, 64 per dependent round trip, to find the owners of 19 chunks: lane k takes the slab's k-th flagged
    // chunk (the rank table gives the word, the word the bit), finds the request that holds the chunk's first byte by a binary search of the offsets
    // (within 4096 requests of the slab's first: a dozen dependent loads, all lanes at once), writes the pair at ITS place (pair_base + k) and resets the
    // records of the requests that overlap the chunk. A slab the bounds do not fit (fields shorter than 32 bytes on average, runs of empty fields) falls
    // through to the request-driven walk below, which writes the same values.
    if (sparse) {
        bool redo = false;
        for (uint32_t k = lane; k < cnt; k += 64) {
            uint32_t wlo = 0, whi = kWords;  // the last word w with rank[w] <= k holds the k-th flagged chunk
            while (wlo + 1u < whi) {
                const uint32_t mid = (wlo + whi) >> 1;
                if (rank[mid] <= k) wlo = mid;
                else whi = mid;
            }
            uint32_t word = bits[wlo];
            for (uint32_t skip = k - rank[wlo]; skip != 0u; skip--) word &= word - 1u;
            const uint32_t chunk = c_first + wlo * 32u + (uint32_t)__builtin_ctz(word), byte0 = chunk * 16u;
            if (byte0 + 16u <= begin) continue;  // (wholly before off[0]: "no pair", written above)
            const uint32_t key = max(byte0, begin);  // (the chunk that holds off[0] belongs to the first request that has bytes)
            uint32_t l = lo, h = min(a.n, lo + 4096u);  // the first request with off[r + 1] > key, in [l, h]
            const uint32_t h0 = h;
            while (l < h) {
                const uint32_t mid = (l + h) >> 1;
                if (a.off[mid + 1] > key) h = mid;
                else l = mid + 1u;
            }
            if (l == h0) {  // not within the bound (or no request at all holds the byte: cannot be, the filter flags no chunk at or beyond off[n])
                if (h0 < a.n) { redo = true; continue; }
                if (pair_base + k < a.pair_cap) a.pairs[pair_base + k] = make_uint2(kNone, 0u);
                continue;
            }
            if (pair_base + k < a.pair_cap) a.pairs[pair_base + k] = make_uint2(l, chunk);
            if (a.n_heads == 0) {  // the requests with a byte in the chunk (their records are merged into by confirm_kernel: see below)
                uint32_t r = l, steps = 0;
                for (; r < a.n && steps < 48u; r++, steps++) {
                    const uint32_t s = a.off[r];
                    if (s >= byte0 + 16u) break;
                    if (a.off[r + 1] > s) a.rec[r] = 0u;
                }
                if (steps == 48u) redo = true;  // (a run of empty fields)
            }
        }
        if (__ballot(redo) == 0ull) return;
    }
    {
    // (software-pipelined: the offsets of the next 64 requests are in flight while these are ranked)
    uint32_t s_n = 0xFFFFFFFFu, e_n = 0xFFFFFFFFu;
    {
        const uint32_t r = lo + lane;
        if (r < a.n) { s_n = a.off[r]; e_n = a.off[r + 1]; }
    }
    for (uint32_t rb = lo; rb < a.n; rb += 64) {
        const uint32_t r = rb + lane;
        const bool live = r < a.n;
        const uint32_t s = s_n, e = e_n;
        {
            const uint32_t r2 = r + 64;
            s_n = e_n = 0xFFFFFFFFu;
            if (r2 < a.n) { s_n = a.off[r2]; e_n = a.off[r2 + 1]; }
        }
        if (a.pairs != nullptr) {
            // A request with a flagged chunk among its own gets its hit record ZEROED here (confirm_kernel merges hits into it; the verdict
            // kernel only reads records whose valid bit a hit or a walk set — which implies a flagged chunk): one store per such request
            // instead of a memset of the pass's every record (69 passes x 4 MB per batch of the 4096-rule set). A pass with filter
            // heads keeps the host's memset: its records are written outside the flagged requests too.
            if (a.n_heads == 0 && live && e > s) {
                const uint32_t g_lo = s >> 4, g_hi = (e - 1u) >> 4;
                if (g_hi >= c_first + lo_c && g_lo < c_first + hi_c) {
                    const uint32_t y0 = g_lo > c_first + lo_c ? g_lo - c_first : lo_c, y1 = min(g_hi - c_first, hi_c - 1u);
                    if (rank_of(y1 + 1u) != rank_of(y0)) a.rec[r] = 0u;
                }
            }
            // the flagged chunks whose first byte lies in this request (clipped to the slab). The chunk that holds off[0] when off[0] is
            // not a multiple of 16 (a slab view) begins before every request: it belongs to the first request that has bytes, which is
            // the one with s == off[0] < e (it used to have no owner, and a factor completing in the view's first bytes was
            // never confirmed).
            if (live && e > s) {
                const uint32_t f_lo = s == begin ? s >> 4 : (s + 15u) >> 4, f_hi = (e - 1u) >> 4;
                if (f_lo <= f_hi && f_hi >= c_first + lo_c && f_lo < c_first + hi_c) {
                    const uint32_t x0 = f_lo > c_first + lo_c ? f_lo - c_first : lo_c, x1 = min(f_hi - c_first, hi_c - 1u);
                    for (uint32_t w = x0 >> 5; w <= (x1 >> 5); w++) {
                        const uint32_t word = bits[w];
                        uint32_t bw = word;
                        if (w == (x0 >> 5)) bw &= ~0u << (x0 & 31u);
                        if (w == (x1 >> 5)) bw &= ~0u >> (31u - (x1 & 31u));
                        while (bw) {
                            const uint32_t bit = (uint32_t)__builtin_ctz(bw);
                            bw &= bw - 1u;
                            const uint32_t at = pair_base + rank[w] + (uint32_t)__builtin_popcount(word & ((1u << bit) - 1u));
                            if (at < a.pair_cap) a.pairs[at] = make_uint2(r, c_first + w * 32u + bit);
                        }
                    }
                }
            }
            if (__ballot(live && (uint64_t)s >= b1) != 0) break;  // (offsets ascend: no later request owns a byte of this slab)
            continue;
        }
        bool mark = false;
        if (live && (uint64_t)s < b1 + 16) {
            // chunks j with 16 j - back < e and 16 j + 16 >= s, clipped to the slab
            const uint32_t j_lo = s == 0 ? 0u : (s - 1u) / 16u, j_hi = (uint32_t)(((uint64_t)e + back - 1u) / 16u);
            const uint32_t x0 = j_lo > c_first + lo_c ? j_lo - c_first : lo_c;
            if (j_hi >= c_first + lo_c && x0 < hi_c) {
                const uint32_t x1 = min(j_hi - c_first, hi_c - 1u);
                mark = x0 <= x1 && rank_of(x1 + 1u) != rank_of(x0);
            }
        }
        // the wave's 64 verdicts as at most three bitmap words (rb is not word-aligned in general)
        const unsigned long long m = __ballot(mark);
        if (m != 0) {
            const uint32_t sh = rb & 31u, w0 = rb >> 5;
            const uint32_t parts[3] = {(uint32_t)(m << sh), (uint32_t)(sh ? m >> (32u - sh) : m >> 32), sh ? (uint32_t)(m >> (64u - sh)) : 0u};
            if (lane < 3 && parts[lane] != 0) atomicOr(&a.bitmap[w0 + lane], parts[lane]);
        }
        if (__ballot(live && (uint64_t)s >= b1 + 16) != 0) break;  // (offsets ascend: nothing further overlaps)
    }
    }
}

// compact_kernel: a workgroup turns kCompactWords bitmap words into its part of the ascending request list. Each wave owns 512
// consecutive words and takes them 64 at a time — lane = word — so that the lanes of one store instruction write neighbouring list
// entries (the first version gave every thread 8 consecutive words: its stores were 20 entries apart per lane and it took
// 180 us next to the attribute kernel, 8 us alone).
__global__ __launch_bounds__(256) void compact_kernel(FilterTable B) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint32_t red[256];
    const FilterArgs a = load_descriptor(&B.f[blockIdx.y]);
    const uint32_t words = (a.n + 31) / 32, w0 = blockIdx.x * kCompactWords, tid = threadIdx.x, wave = wave_index(), lane = tid & 63;
    if (w0 >= words) return;
    const uint32_t n_blocks = (words + kCompactWords - 1) / kCompactWords;
    uint32_t acc = 0;
    for (uint32_t bq = tid; bq < blockIdx.x; bq += 256) acc += a.block_count[bq];
    red[tid] = acc;
    __syncthreads();
    for (uint32_t h = 128; h > 0; h >>= 1) {
        if (tid < h) red[tid] += red[tid + h];
        __syncthreads();
    }
    const uint32_t base = red[0];
    __syncthreads();
    constexpr uint32_t kSteps = kCompactWords / 256;  // 64-word steps per wave
    const uint32_t wave_w0 = w0 + wave * (kSteps * 64);
    uint32_t wv[kSteps], mine = 0;
#pragma unroll
    for (uint32_t q = 0; q < kSteps; q++) {
        const uint32_t w = wave_w0 + q * 64 + lane;
        wv[q] = w < words ? a.bitmap[w] : 0u;
        mine += (uint32_t)__builtin_popcount(wv[q]);
    }
    const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add(mine), 63);
    if (lane == 0) red[wave] = wave_total;
    __syncthreads();
    uint32_t pos0 = base;
    for (uint32_t k = 0; k < wave; k++) pos0 += red[k];
#pragma unroll
    for (uint32_t q = 0; q < kSteps; q++) {
        uint32_t w = wv[q];
        const uint32_t pc = (uint32_t)__builtin_popcount(w), incl = wave_scan_add(pc);
        uint32_t pos = pos0 + incl - pc;
        const uint32_t r0 = (wave_w0 + q * 64 + lane) * 32;
        while (w) {
            a.list[pos++] = r0 + (uint32_t)__builtin_ctz(w);
            w &= w - 1;
        }
        pos0 += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (blockIdx.x == n_blocks - 1 && tid == 255) *a.list_count = pos0;
}


__host__ __device__ static inline uint32_t verdict_wave_lds(uint32_t n_cols, uint32_t n_rules) {
    const uint32_t colw = (n_cols + 31) / 32, rulew = (n_rules + 31) / 32;
    return ((n_cols * 8 + colw * 4 + rulew * 4 + n_rules * 2) + 15) & ~15u;  // columns, bitmaps, candidates (also: the list of non-zero columns)
}

// SPARSE column file (verdict_kernel<.., SP = true>): of a rule set's ~1000 - 4500 columns a group of 64 requests dirties a hundred or two,
// yet the dense file costs 8 bytes per COLUMN and wave — 9 KiB of the 1k-rule set's 11.5 KiB per wave, which is what held the kernel at
// one workgroup of 11 waves per CU (under 3 waves per SIMD; 57 % of its wave cycles waiting: VERDICT r4). The sparse file keeps, per 32
// columns, the dirty bits and the number of dirty columns before the word (8 bytes), and one 64-bit value per DIRTY column, at the
// column's rank among the dirty ones: ~4 KiB per wave whatever the rule set, i.e. two workgroups of 12 waves per CU. A group is marked
// twice — bits first (the ranks need all of them), values second — and a value read is two dependent LDS reads instead of one. A group
// with more dirty columns than the value array holds (v_cap) keeps the rest in a per-wave spill array in global memory (same results).
__host__ __device__ static inline uint32_t verdict_wave_lds_sp(uint32_t n_cols, uint32_t n_rules, uint32_t v_cap) {
    const uint32_t colw = (n_cols + 31) / 32, rulew = (n_rules + 31) / 32;
    return ((colw * 8 + v_cap * 8 + rulew * 4 + n_rules * 2) + 15) & ~15u;  // {dirty bits, rank} per 32 columns, values, rule bitmap, candidates
}
__host__ __device__ static inline uint32_t verdict_v_cap(uint32_t n_cols, bool tiny) {
    const uint32_t v = tiny ? 8u : 256u + n_cols / 16u;
    return v < n_cols ? v : n_cols;
}

// LDS-resident copies of the small read-mostly program tables (LT = true): trigger lists, rule headers and DNF literals are
// gathered several times per 64-request group, and each gather from L2 is a ~1 us round trip that the few waves a CU can hold
// (the column files fill LDS) cannot hide.
struct VerdictTables {
    uint32_t bitcol, trig_off, trig_rules, rules, lits, pub, end;  // byte offsets from the start of the table region
};
__host__ __device__ static inline VerdictTables verdict_tables(uint32_t n_cols, uint32_t n_rules, uint32_t n_trig, uint32_t n_lits, bool lt) {
    VerdictTables t;
    t.bitcol = 0;
    t.trig_off = 0;
    t.trig_rules = t.trig_off + (lt ? ((n_cols + 1) * 2 + 3) & ~3u : 0u);
    t.rules = t.trig_rules + (lt ? (n_trig * 2 + 7) & ~7u : 0u);
    t.lits = t.rules + (lt ? n_rules * 8 : 0u);
    t.pub = t.lits + (lt ? n_lits * 4 : 0u);
    t.end = t.pub + (lt ? (n_rules * 2 + 3) & ~3u : 0u);
    return t;
}

// BR = 64-pass bitmap registers (1: up to 64 passes, else kMaxPasses)
template <bool LT, int BR, bool SP>
__global__ __launch_bounds__(SP ? 768 : 1024, (SP && BR == 1) ? 6 : 1) void verdict_kernel(VerdictArgs a) {  // (sparse, up to 64 passes: 6 waves per SIMD = two 12-wave workgroups per CU)
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t tid = threadIdx.x, wave = wave_index(), lane = tid & 63, n_waves = blockDim.x >> 6;
#ifdef PWAF_PROFILING
    const uint32_t dbg_skip = a.debug_skip;  // section switches for timing experiments (wrong results when set): -DPWAF_PROFILING builds only
#else
    constexpr uint32_t dbg_skip = 0;
#endif
    const uint32_t colw = (a.n_cols + 31) / 32, rulew = (a.n_rules + 31) / 32;
    const uint32_t v_cap = a.v_cap;  // (SP) values held in LDS; dirty columns of higher rank: the wave's spill array
    const uint32_t wave_bytes = SP ? verdict_wave_lds_sp(a.n_cols, a.n_rules, v_cap) : verdict_wave_lds(a.n_cols, a.n_rules);
    unsigned char *mine = lds + (size_t)wave * wave_bytes;
    // dense: col[n_cols] | colnz[colw]; sparse: nz[colw] = {dirty bits, dirty columns before the word} | vals[v_cap]
    unsigned long long *col = reinterpret_cast<unsigned long long *>(mine);
    uint32_t *colnz = reinterpret_cast<uint32_t *>(col + a.n_cols);
    uint32_t *nz = reinterpret_cast<uint32_t *>(mine);
    unsigned long long *vals = reinterpret_cast<unsigned long long *>(mine + (size_t)colw * 8);
    unsigned long long *spill = (SP && a.spill != nullptr) ? a.spill + (size_t)(blockIdx.x * n_waves + wave) * (a.n_cols - v_cap) : nullptr;
    uint32_t *rulebm = SP ? reinterpret_cast<uint32_t *>(vals + v_cap) : colnz + colw;
    uint16_t *cand = reinterpret_cast<uint16_t *>(rulebm + rulew);
    unsigned char *tables = lds + (size_t)n_waves * wave_bytes;
    const VerdictTables vt = verdict_tables(a.n_cols, a.n_rules, a.n_trig, a.n_lits, LT);
    uint16_t *l_trig_off = reinterpret_cast<uint16_t *>(tables + vt.trig_off);
    uint16_t *l_trig_rules = reinterpret_cast<uint16_t *>(tables + vt.trig_rules);
    uint2 *l_rules = reinterpret_cast<uint2 *>(tables + vt.rules);
    uint32_t *l_lits = reinterpret_cast<uint32_t *>(tables + vt.lits);
    uint16_t *l_pub = reinterpret_cast<uint16_t *>(tables + vt.pub);  // public rule index, 16 bits (the pseudo rules' 0xFFFFFFFx ids keep their low half)
    if (LT) {
        for (uint32_t k = tid; k <= a.n_cols; k += blockDim.x) l_trig_off[k] = (uint16_t)a.trig_off[k];
        for (uint32_t k = tid; k < a.n_trig; k += blockDim.x) l_trig_rules[k] = a.trig_rules[k];
        for (uint32_t k = tid; k < a.n_rules; k += blockDim.x) {
            const DevRule dr = a.rules[k];  // header without the public index (only read when the rule decides a request)
            l_rules[k] = make_uint2(dr.lit_off, dr.lit_cnt | ((uint32_t)dr.eff_unverified << 16) | ((uint32_t)dr.eff_verified << 24));
            l_pub[k] = (uint16_t)dr.public_idx;
        }
        for (uint32_t k = tid; k < a.n_lits; k += blockDim.x) l_lits[k] = a.lits[k];
    }
    __syncthreads();
    const unsigned long long mybit = 1ull << lane;
    const unsigned long long lt_mask = mybit - 1;
    unsigned long long cnt_block = 0, cnt_captcha = 0, cnt_bypass = 0, cnt_allow = 0;  // wave-uniform tallies

    // group-invariant, fetched once per wave: this lane's word of the always-candidate bitmap
    const uint32_t h_always = lane < rulew ? a.always_rules[lane] : 0u;

    // A group's inputs — the hit records of up to kPre passes, the captcha flag, the attribute kernel's pair count and first 64
    // pairs — are requested ONE GROUP AHEAD: with ~9 waves per CU nothing else hides the ~2 us those first-touch loads take.
    // Which passes: a list-driven (sparse) pass has a visited bitmap, and in most groups of 64 requests most such passes visited
    // nobody (a 4096-rule set over 64 header fields has ~90 passes, a handful of them non-empty per group). The group's 64 visited
    // bits of EVERY pass are requested TWO groups ahead, lane q holding pass q's word (dense passes: all ones); one group ahead a
    // ballot names the non-empty passes and only those get a record load — limited to the lanes whose bit is set.
    constexpr int kPre = kVerdictPre;
    constexpr int kBitRegs = BR;  // 64 passes per register
    struct Bits {
        unsigned long long w[kBitRegs];  // lane l of w[r]: the visited bits of pass r * 64 + l for the group's 64 requests
    };
    struct Inputs {
        uint32_t rv[kPre];    // the hit records of the group's first kPre non-empty passes ...
        uint32_t base[kPre];  // ... and their first columns (wave-uniform; kNone = slot unused)
        Bits bits;            // for the passes left over
        unsigned long long rest[kBitRegs];  // wave-uniform: non-empty passes that got no slot (fetched inside the group: rare)
        uint32_t flags, n_pairs;
        uint4 pair0;
        uint32_t res0;  // the first result word of the specialized residual program (VerdictArgs::res_match)
    };
    // group-invariant, per lane: the bitmap and first column of "my" pass in each register
    const uint32_t *my_bits[kBitRegs];
    uint32_t my_base[kBitRegs];
    bool my_live[kBitRegs];
#pragma unroll
    for (int r = 0; r < kBitRegs; r++) {
        const uint32_t ps = (uint32_t)r * 64 + lane;
        my_live[r] = ps < a.n_passes;
        my_bits[r] = nullptr;
        my_base[r] = 0;
        if (my_live[r]) {
            const PassInfo pi = a.passes[ps];
            const uint32_t kind = pi.kind_slot >> 24, slot = pi.kind_slot & 0xFFFFFFu;
            if (kind == 3) my_live[r] = false;  // a short-literal pass: its columns arrive as attribute pairs, it has no records
            my_bits[r] = kind == 1 ? a.cand_bits + (size_t)slot * a.bit_words : kind == 2 ? a.visit_bits + (size_t)slot * a.bit_words : nullptr;
            my_base[r] = pi.base;
        }
    }
    auto request_bits = [&](const uint32_t g, Bits &bt) {
        const uint32_t gg = min(g, a.n_groups - 1);
#pragma unroll
        for (int r = 0; r < kBitRegs; r++) {
            // (the bitmap of a batch whose size is not a multiple of 64 is padded to whole groups by the engine)
            bt.w[r] = !my_live[r] ? 0ull : my_bits[r] != nullptr ? *reinterpret_cast<const unsigned long long *>(my_bits[r] + 2 * (size_t)gg) : ~0ull;
        }
    };
    auto request_inputs = [&](const uint32_t g, const Bits &bt, Inputs &in) {
        const uint32_t i = g * 64 + lane;
        const bool valid = g < a.n_groups && i < a.n;
        unsigned long long nz[kBitRegs];
#pragma unroll
        for (int r = 0; r < kBitRegs; r++) nz[r] = (uint32_t)r * 64 < a.n_passes ? __ballot(bt.w[r] != 0) : 0ull;
#pragma unroll
        for (int q = 0; q < kPre; q++) {
            uint32_t ps = kNone, base = kNone;
            unsigned long long word = 0;
#pragma unroll
            for (int r = 0; r < kBitRegs; r++) {
                if (ps == kNone && nz[r] != 0) {  // wave-uniform
                    const int l = __builtin_ctzll(nz[r]);
                    nz[r] &= nz[r] - 1;
                    ps = (uint32_t)r * 64 + (uint32_t)l;
                    word = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(bt.w[r] >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bt.w[r], l);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)my_base[r], l);
                }
            }
            in.base[q] = base;
            in.rv[q] = (ps != kNone && valid && ((word >> lane) & 1ull)) ? a.rec[(size_t)ps * a.n + i] : 0u;
        }
#pragma unroll
        for (int r = 0; r < kBitRegs; r++) in.rest[r] = nz[r];
        in.bits = bt;
        in.flags = valid ? (uint32_t)a.flags[i] : 0u;
        const uint32_t gg = min(g, a.n_groups - 1);
        in.n_pairs = a.ghdr[gg];
        in.pair0 = a.gpairs[(size_t)gg * a.pair_stride + lane];  // (the buffer is padded: reading past the group's count is harmless)
        in.res0 = (a.res_words && valid) ? a.res_match[i] : 0u;
    };
    if (!SP) {
        for (uint32_t k = lane; k < a.n_cols; k += 64) col[k] = 0;  // the column file starts clean; afterwards groups clean up after themselves
        for (uint32_t k = lane; k < colw; k += 64) colnz[k] = 0;
    }
    const uint32_t g_stride = gridDim.x * n_waves;
    Inputs cur;
    Bits b_nxt;
    {
        Bits b_cur;
        request_bits(blockIdx.x * n_waves + wave, b_cur);
        request_inputs(blockIdx.x * n_waves + wave, b_cur, cur);
        request_bits(blockIdx.x * n_waves + wave + g_stride, b_nxt);
    }

    for (uint32_t g = blockIdx.x * n_waves + wave; g < a.n_groups; g += g_stride) {
        const uint32_t i = g * 64 + lane;
        const bool valid = i < a.n;
        const unsigned long long valid_mask = __ballot(valid);
        Inputs nxt;
        request_inputs(g + g_stride, b_nxt, nxt);
        request_bits(g + 2 * g_stride, b_nxt);

        // 1. clear the column file and the bitmaps; column 0 is the constant TRUE; rules that can match with every column
        //    zero (a term made of negations only) are always candidates
        //    (dense: only the columns the previous group dirtied are cleared — the non-zero bitmap names them; sparse: the dirty bits)
        if (!SP) {
            for (uint32_t wv = lane; wv < colw; wv += 64) {
                uint32_t nzw = colnz[wv];
                colnz[wv] = wv == 0 ? 1u : 0u;
                while (nzw) {
                    col[wv * 32 + (uint32_t)__builtin_ctz(nzw)] = 0;
                    nzw &= nzw - 1;
                }
            }
        } else {
            for (uint32_t wv = lane; wv < colw; wv += 64) nz[2 * wv] = wv == 0 ? 1u : 0u;
        }
        for (uint32_t k = lane; k < rulew; k += 64) rulebm[k] = k < 64 ? h_always : a.always_rules[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (!SP && lane == 0) col[0] = ~0ull;

        const uint32_t flags = cur.flags;
        const uint4 *pairs = a.gpairs + (size_t)g * a.pair_stride;
        const uint32_t n_pairs = cur.n_pairs;
        const uint4 pair0 = cur.pair0;
        const unsigned long long verified_mask = __ballot(valid && (flags & PWAF_FLAG_CAPTCHA_VERIFIED));

        // Where column c's 64-request word lives. Dense: the column file. Sparse: at the column's rank among the group's dirty columns
        // (in LDS below v_cap, in the wave's spill array beyond) — valid once the ranks are known (phase 1).
        auto slot_of = [&](const uint32_t c) {
            const uint32_t w = c >> 5, bits = nz[2 * w], before = nz[2 * w + 1];
            return before + (uint32_t)__builtin_popcount(bits & ((1u << (c & 31u)) - 1u));
        };
        auto word_at = [&](const uint32_t slot) -> unsigned long long * { return slot < v_cap ? vals + slot : spill + (slot - v_cap); };
        // PHASE 0 (sparse only): dirty bits. PHASE 1 (sparse): values. PHASE 2: the dense file, bits and values at once.
        auto mark_all = [&](auto phase_tag) {
            constexpr int PH = decltype(phase_tag)::value;
            auto put = [&](const uint32_t c, const unsigned long long m) {  // (one lane speaks for column c)
                if (PH == 0) atomicOr(&nz[2 * (c >> 5)], 1u << (c & 31u));
                else if (PH == 1) atomicOr(word_at(slot_of(c)), m);
                else {
                    atomicOr(&col[c], m);
                    atomicOr(&colnz[c >> 5], 1u << (c & 31));
                }
            };
            // 2. scan results: each lane marks the columns its hit records name
            auto mark_hits = [&](const uint32_t rv, const uint32_t base) {
                if (__ballot(rv != 0) == 0) return;  // nobody in the group matched anything in this pass
                if (rv & REC_OVERFLOW) {
                    for (uint32_t k = rv & ~REC_OVERFLOW; k != kNone;) {
                        const PoolEntry pe = a.pool[k];
                        put(base + pe.atom, mybit);
                        k = pe.next;
                    }
                }
                // inline atoms: lanes that name the SAME atom (frequent atoms such as a browser User-Agent prefix are named by
                // most of the 64 requests) are folded into one column update instead of 64 serialised LDS atomics
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const uint32_t x = (rv & REC_OVERFLOW) ? 0u : (half == 0 ? rv & 0x7FFFu : (rv >> 15) & 0x7FFFu);
                    unsigned long long todo = __ballot(x != 0);
                    while (todo) {
                        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
                        const uint32_t xa = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)__builtin_amdgcn_readfirstlane(leader));
                        const unsigned long long same = __ballot(x == xa);
                        todo &= ~same;
                        if (lane == leader) put(base + xa - 1, same);
                    }
                }
            };
            if (!(dbg_skip & 1u)) {
#pragma unroll
                for (int q = 0; q < kPre; q++) {
                    if (cur.base[q] == kNone) break;
                    mark_hits(cur.rv[q], cur.base[q]);
                }
                // more than kPre non-empty passes in this group (many dense passes, or a burst of candidates): fetched here
#pragma unroll
                for (int r = 0; r < kBitRegs; r++) {
                    unsigned long long m = cur.rest[r];
                    while (m) {
                        const int l = __builtin_ctzll(m);
                        m &= m - 1;
                        const uint32_t ps = (uint32_t)r * 64 + (uint32_t)l;
                        const unsigned long long word = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(cur.bits.w[r] >> 32), l) << 32) |
                                                        (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cur.bits.w[r], l);
                        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)my_base[r], l);
                        mark_hits((valid && ((word >> lane) & 1ull)) ? a.rec[(size_t)ps * a.n + i] : 0u, base);
                    }
                }
            }
            // 2b. residual rules evaluated by their specialized program: one result word per request and 32 rules, transposed here into the
            //     rules' column words (a ballot per rule that matched anybody in the group)
            for (uint32_t w = 0; w < a.res_words; w++) {
                const uint32_t mw = w == 0 ? cur.res0 : (valid ? a.res_match[(size_t)w * a.n + i] : 0u);
                if (__ballot(mw != 0u) == 0ull) continue;
                uint32_t any = mw;
                for (uint32_t d = 1; d < 64u; d <<= 1) any |= (uint32_t)__shfl_xor((int)any, (int)d, 64);
                any = (uint32_t)__builtin_amdgcn_readfirstlane((int)any);
                for (; any; any &= any - 1u) {
                    const uint32_t k = (uint32_t)__builtin_ctz(any);
                    const unsigned long long who = __ballot(((mw >> k) & 1u) != 0u);
                    if (lane == 0) put(a.res_base + 32u * w + k, who);
                }
            }
            // 3. everything that is not a string scan (memberships, comparisons) arrives from the attribute kernel as ready-made
            //    column words: one lane per pair (one pair per atom and group, and no scan pass owns these columns: a plain store)
            if (!(dbg_skip & 2u)) {
                for (uint32_t p = lane; p < n_pairs; p += 64) {
                    const uint4 pr = p < 64 ? pair0 : pairs[p];
                    const unsigned long long v = ((unsigned long long)pr.w << 32) | pr.z;
                    if (PH == 0) atomicOr(&nz[2 * (pr.x >> 5)], 1u << (pr.x & 31u));
                    else if (PH == 1) {
                        const uint32_t sl = slot_of(pr.x);
                        if (sl < v_cap) vals[sl] = v;
                        else atomicOr(spill + (sl - v_cap), v);
                    }
                    else {
                        col[pr.x] = v;
                        atomicOr(&colnz[pr.x >> 5], 1u << (pr.x & 31));
                    }
                }
            }
        };
        uint32_t n_dirty = 0;  // (sparse) dirty columns of the group
        if (SP) {
            mark_all(std::integral_constant<int, 0>{});
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // the ranks: dirty columns before each word (the prefix sum also lists the dirty columns for step 4)
            for (uint32_t wb = 0; wb < colw; wb += 64) {
                const uint32_t bits = wb + lane < colw ? nz[2 * (wb + lane)] : 0u;
                const uint32_t pc = (uint32_t)__builtin_popcount(bits), incl = wave_scan_add(pc);
                if (wb + lane < colw) nz[2 * (wb + lane) + 1] = n_dirty + incl - pc;
                n_dirty += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
            // (column 0 = TRUE is dirty by construction and ranks first; a spilled word is reset with a read-modify-write, like everything
            // that touches it afterwards: performed at L2 in order — a plain store could still be on its way when the first OR arrives)
            for (uint32_t k = lane; k < n_dirty; k += 64) {
                if (k < v_cap) vals[k] = k == 0 ? ~0ull : 0ull;
                else atomicExch(spill + (k - v_cap), 0ull);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            mark_all(std::integral_constant<int, 1>{});
        } else {
            mark_all(std::integral_constant<int, 2>{});
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // the bitmap of non-zero columns (dense: colnz; sparse: the bit words of nz) and a column's word, for the steps below
        auto nz_bits = [&](const uint32_t wv) { return SP ? nz[2 * wv] : colnz[wv]; };
        auto col_word = [&](const uint32_t c) -> unsigned long long {
            if (!SP) return col[c];
            const uint32_t w = c >> 5, bits = nz[2 * w], before = nz[2 * w + 1];
            if (!((bits >> (c & 31u)) & 1u)) return 0ull;
            const uint32_t sl = before + (uint32_t)__builtin_popcount(bits & ((1u << (c & 31u)) - 1u));
            return sl < v_cap ? vals[sl] : __hip_atomic_load(spill + (sl - v_cap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a spilled word: from L2, where the ORs were performed)
        };

        // 4. candidate rules: a rule can only match some request of this group if one of its terms has a non-zero positive
        //    column (trigger lists, one chosen literal per term) or consists of negations only (always_rules).
        //    The non-zero columns are first LISTED (a prefix sum over the bitmap words' popcounts: the candidate array is free until
        //    the compaction below), then one lane takes one column. (Round 4 gave a lane a whole bitmap word: the ~60 attribute columns
        //    of a group sit in a handful of neighbouring words, so a few lanes walked eight to sixteen columns one after the other,
        //    each a chain of three dependent LDS reads, while the other sixty waited.)
        if (!(dbg_skip & 8u)) {
            uint32_t n_nz = 0;
            for (uint32_t wb = 0; wb < colw; wb += 64) {
                uint32_t nzv = wb + lane < colw ? nz_bits(wb + lane) : 0u;
                const uint32_t pc = (uint32_t)__builtin_popcount(nzv), incl = wave_scan_add(pc);
                uint32_t pos = n_nz + incl - pc;
                while (nzv) {
                    if (pos < a.n_rules) cand[pos] = (uint16_t)((wb + lane) * 32 + (uint32_t)__builtin_ctz(nzv));  // (n_cols < 65536: launch_verdict)
                    pos++;
                    nzv &= nzv - 1;
                }
                n_nz += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (n_nz <= a.n_rules) {
                for (uint32_t j = lane; j < n_nz; j += 64) {
                    const uint32_t c = cand[j];
                    const uint32_t kb = LT ? (uint32_t)l_trig_off[c] : a.trig_off[c], ke = LT ? (uint32_t)l_trig_off[c + 1] : a.trig_off[c + 1];
                    for (uint32_t k = kb; k < ke; k++) {
                        const uint32_t r = LT ? (uint32_t)l_trig_rules[k] : (uint32_t)a.trig_rules[k];
                        atomicOr(&rulebm[r >> 5], 1u << (r & 31));
                    }
                }
            } else {  // (more non-zero columns than the list holds: word by word)
                for (uint32_t wv = lane; wv < colw; wv += 64) {
                    uint32_t nzv = nz_bits(wv);
                    while (nzv) {
                        const uint32_t c = wv * 32 + (uint32_t)__builtin_ctz(nzv);
                        nzv &= nzv - 1;
                        const uint32_t kb = LT ? (uint32_t)l_trig_off[c] : a.trig_off[c], ke = LT ? (uint32_t)l_trig_off[c + 1] : a.trig_off[c + 1];
                        for (uint32_t k = kb; k < ke; k++) {
                            const uint32_t r = LT ? (uint32_t)l_trig_rules[k] : (uint32_t)a.trig_rules[k];
                            atomicOr(&rulebm[r >> 5], 1u << (r & 31));
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ordered compaction of the rule bitmap into the candidate list (ascending rule index = evaluation order)
        uint32_t n_cand = 0;
        for (uint32_t wb = 0; wb < rulew && !(dbg_skip & 16u); wb += 64) {
            const uint32_t word = wb + lane < rulew ? rulebm[wb + lane] : 0u;
            const uint32_t pc = (uint32_t)__builtin_popcount(word), incl = wave_scan_add(pc);
            uint32_t pos = n_cand + incl - pc, wrd = word;
            while (wrd) {
                cand[pos++] = (uint16_t)((wb + lane) * 32 + (uint32_t)__builtin_ctz(wrd));
                wrd &= wrd - 1;
            }
            n_cand += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        // 5. evaluate candidates: one lane per rule, 64 requests per ALU op. Each candidate's 64-request match word goes to LDS;
        //    then every REQUEST lane scans the words in rule order for its own bit (broadcast reads), so first-match-wins needs
        //    no lane-to-lane traffic
        unsigned long long pending = valid_mask;
        bool undecided = valid;
        uint32_t my_action = PWAF_ACTION_ALLOW, my_rule = PWAF_RULE_NONE;
        for (uint32_t base = 0; base < n_cand && pending != 0 && !(dbg_skip & 32u); base += 64) {
            unsigned long long fire = 0;
            if (base + lane < n_cand) {
                const uint32_t my_cand = cand[base + lane];
                uint32_t lit_off, lit_cnt, eff_u, eff_v;
                if (LT) {
                    const uint2 hdr = l_rules[my_cand];
                    lit_off = hdr.x;
                    lit_cnt = hdr.y & 0xFFFFu;
                    eff_u = (hdr.y >> 16) & 0xFFu;
                    eff_v = hdr.y >> 24;
                } else {
                    const DevRule dr = a.rules[my_cand];
                    lit_off = dr.lit_off;
                    lit_cnt = dr.lit_cnt;
                    eff_u = dr.eff_unverified;
                    eff_v = dr.eff_verified;
                }
                unsigned long long acc_or = 0, acc_and = ~0ull;
                // literals four at a time: the four literal words are requested together, then the four column words — two
                // LDS round trips per four literals instead of eight dependent ones (a padding literal is column 0 = TRUE)
                for (uint32_t k = lit_off; k < lit_off + lit_cnt; k += 4) {
                    uint32_t lit[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) lit[q] = k + q < lit_off + lit_cnt ? (LT ? l_lits[k + q] : a.lits[k + q]) : 0u;
                    unsigned long long cw[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) cw[q] = col_word(lit[q] & LIT_ATOM_MASK);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        acc_and &= (lit[q] & LIT_NEG) ? ~cw[q] : cw[q];
                        if (lit[q] & LIT_TERM_END) {
                            acc_or |= acc_and;
                            acc_and = ~0ull;
                        }
                    }
                }
                // a match only decides when the rule's action list yields an effect for that client
                fire = acc_or & ((eff_u ? ~verified_mask : 0ull) | (eff_v ? verified_mask : 0ull));
            }
            if (dbg_skip & 256u) fire = 0;
            // first match wins: the candidates that fire for ANYBODY (a few per group: most candidates' terms stay false) are taken in
            // rule order, each word broadcast from its lane — a request's rule is the first word that holds its bit. (Round 4 parked
            // all 64 words in LDS and every request lane read them back, sixteen rounds of four broadcast reads per 64 candidates.)
            unsigned long long firing = (dbg_skip & 128u) ? 0ull : __ballot(fire != 0);
            uint32_t first = kNone;
            while (firing != 0 && pending != 0) {
                const int j = __builtin_ctzll(firing);
                firing &= firing - 1;
                const unsigned long long fj = (((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(fire >> 32), j) << 32) |
                                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)fire, j)) & pending;
                if (fj & mybit) first = (uint32_t)j;
                pending &= ~fj;
            }
            if (undecided && first != kNone) {
                undecided = false;
                const uint32_t jc = cand[base + first];
                uint32_t eff_u, eff_v;
                if (LT) {
                    const uint32_t y = l_rules[jc].y;
                    eff_u = (y >> 16) & 0xFFu;
                    eff_v = y >> 24;
                    const uint32_t pi = l_pub[jc];
                    my_rule = pi >= 0xFFF0u ? 0xFFFF0000u | pi : pi;
                } else {
                    const DevRule dr = a.rules[jc];
                    eff_u = dr.eff_unverified;
                    eff_v = dr.eff_verified;
                    my_rule = dr.public_idx;
                }
                my_action = (verified_mask & mybit) ? eff_v : eff_u;
            }
        }

        if (dbg_skip & 64u) my_rule = n_cand;  // profiling aid: report the candidate count instead of the deciding rule
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
        cur = nxt;
    }
    // Action counters: one atomic per counter and WORKGROUP. (One per wave was measured at 0.25 ms of a 0.6 ms kernel: at the end of
    // every round of workgroups thousands of same-address atomics queue up at the L2.)
    __syncthreads();  // every wave is done with its column file: the start of LDS can hold the tallies
    unsigned long long *tally = reinterpret_cast<unsigned long long *>(lds);
    if (lane == 0) {
        tally[wave * 4 + PWAF_ACTION_ALLOW] = cnt_allow;
        tally[wave * 4 + PWAF_ACTION_BLOCK] = cnt_block;
        tally[wave * 4 + PWAF_ACTION_CAPTCHA] = cnt_captcha;
        tally[wave * 4 + PWAF_ACTION_BYPASS] = cnt_bypass;
    }
    __syncthreads();
    if (a.counts != nullptr && tid < 4) {
        unsigned long long sum = 0;
        for (uint32_t wv = 0; wv < n_waves; wv++) sum += tally[wv * 4 + tid];
        if (sum) atomicAdd(&a.counts[tid], sum);
    }
}

// -------------------------------------------------------------------------------------------------
// verdict, ENTRY-LIST form (round 6; the default: VerdictArgs::sparse_mode 3, 4 = its test hook with 8 entry slots)
// -------------------------------------------------------------------------------------------------
// What a group of 64 requests knows about its columns arrives as a LIST already: the attribute kernel's (column, mask) pairs, one per
// atom that holds for somebody, and per scan pass a handful of distinct atoms named by the hit records. The sparse column file above
// turns that list into {dirty bits, rank, value} in two marking passes with a prefix sum between them and a third walk over the bitmap
// to list the dirty columns again for the trigger step — 0.14 ms of marking and 0.08 ms of listing / triggers in a 0.31 ms kernel
// (profiles/r6_verdict_sections.txt). Here an entry keeps the place it is appended at:
//     slot[column]  one BYTE per column: 0 = clean, 1 + entry index, 255 = the entry lies beyond e_cap: its word is spill[column]
//     vals[e], ecol[e]  the entry's 64-request word and its column (to clean slot[] after the group: only the entries are touched)
// One pass over the inputs, a whole batch of up to 64 entries per instruction: lane k of the batch registers writes slot, value and column
// of entry n + k and walks the column's trigger list. The hit records' atoms are first gathered into the batch registers with
// v_writelane (one entry per DISTINCT atom of a pass: lanes naming the same atom are folded by a ballot). Column 0 = TRUE is entry 0
// for the life of the wave. A column's word for the DNF step is two dependent LDS reads (slot, value), like the sparse file's.
// Duplicates: a column belongs to one pass; inside a pass only an overflow chain (a request with more than two atoms) can name a column that
// another lane's record already named — a pass with such a record appends one column at a time through a lookup of slot[] (rare).
__host__ __device__ static inline uint32_t verdict_wave_lds_el(uint32_t n_cols, uint32_t n_rules, uint32_t e_cap) {
    const uint32_t rulew = (n_rules + 31) / 32;
    // vals | rule bitmap | entry columns (u16) | candidates (u16) | slot bytes
    return (e_cap * 8 + rulew * 4 + ((e_cap * 2 + 3) & ~3u) + ((n_rules * 2 + 3) & ~3u) + ((n_cols + 7) & ~7u) + 15) & ~15u;
}

// one wave-uniform (column, mask) into lane l of three registers (see park64)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void park96(uint32_t &c, uint32_t &lo, uint32_t &hi, const uint32_t col, const unsigned long long m, const uint32_t l) {
    // (scalar registers whatever the register allocator did with the wave-uniform values: under scalar-register pressure it keeps some in
    // vector registers and hands THOSE to an "s" operand)
    const uint32_t s_col = (uint32_t)__builtin_amdgcn_readfirstlane((int)col), s_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)l);
    const uint32_t s_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m), s_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32));
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %5, m0\n\tv_writelane_b32 %2, %6, m0"
                 : "+v"(c), "+v"(lo), "+v"(hi)
                 : "s"(s_col), "s"(s_l), "s"(s_lo), "s"(s_hi)
                 : "m0");
}
#pragma clang diagnostic pop

template <bool LT, int BR>
__global__ __launch_bounds__(768, BR == 1 ? 6 : 1) void verdict2_kernel(VerdictArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t tid = threadIdx.x, wave = wave_index(), lane = tid & 63, n_waves = blockDim.x >> 6;
#ifdef PWAF_PROFILING
    const uint32_t dbg_skip = a.debug_skip;
#else
    constexpr uint32_t dbg_skip = 0;
#endif
    const uint32_t rulew = (a.n_rules + 31) / 32;
    const uint32_t e_cap = a.v_cap;  // entries held in LDS (entry 0 = TRUE included); later entries: the wave's spill array, by column
    const uint32_t wave_bytes = verdict_wave_lds_el(a.n_cols, a.n_rules, e_cap);
    unsigned char *mine = lds + (size_t)wave * wave_bytes;
    unsigned long long *vals = reinterpret_cast<unsigned long long *>(mine);
    uint32_t *rulebm = reinterpret_cast<uint32_t *>(vals + e_cap);
    uint16_t *ecol = reinterpret_cast<uint16_t *>(rulebm + rulew);
    uint16_t *cand = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(ecol) + ((e_cap * 2 + 3) & ~3u));
    uint8_t *slot = reinterpret_cast<unsigned char *>(cand) + ((a.n_rules * 2 + 3) & ~3u);
    const uint32_t slot_words = ((a.n_cols + 7) & ~7u) / 4;
    unsigned long long *spill = a.spill + (size_t)(blockIdx.x * n_waves + wave) * a.n_cols;
    unsigned char *tables = lds + (size_t)n_waves * wave_bytes;
    const VerdictTables vt = verdict_tables(a.n_cols, a.n_rules, a.n_trig, a.n_lits, LT);
    uint16_t *l_trig_off = reinterpret_cast<uint16_t *>(tables + vt.trig_off);
    uint16_t *l_trig_rules = reinterpret_cast<uint16_t *>(tables + vt.trig_rules);
    uint2 *l_rules = reinterpret_cast<uint2 *>(tables + vt.rules);
    uint32_t *l_lits = reinterpret_cast<uint32_t *>(tables + vt.lits);
    uint16_t *l_pub = reinterpret_cast<uint16_t *>(tables + vt.pub);
    if (LT) {
        for (uint32_t k = tid; k <= a.n_cols; k += blockDim.x) l_trig_off[k] = (uint16_t)a.trig_off[k];
        for (uint32_t k = tid; k < a.n_trig; k += blockDim.x) l_trig_rules[k] = a.trig_rules[k];
        for (uint32_t k = tid; k < a.n_rules; k += blockDim.x) {
            const DevRule dr = a.rules[k];
            l_rules[k] = make_uint2(dr.lit_off, dr.lit_cnt | ((uint32_t)dr.eff_unverified << 16) | ((uint32_t)dr.eff_verified << 24));
            l_pub[k] = (uint16_t)dr.public_idx;
        }
        for (uint32_t k = tid; k < a.n_lits; k += blockDim.x) l_lits[k] = a.lits[k];
    }
    // the wave's slot bytes start clean; entry 0 is column 0 = TRUE
    for (uint32_t k = lane; k < slot_words; k += 64) reinterpret_cast<uint32_t *>(slot)[k] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0) {
        slot[0] = 1;
        vals[0] = ~0ull;
        ecol[0] = 0;
    }
    __syncthreads();
    const unsigned long long mybit = 1ull << lane;
    const unsigned long long lt_mask = mybit - 1;
    unsigned long long cnt_block = 0, cnt_captcha = 0, cnt_bypass = 0, cnt_allow = 0;
    const uint32_t h_always = lane < rulew ? a.always_rules[lane] : 0u;
    const bool true_triggers = a.trig_off[1] != a.trig_off[0];  // rules whose chosen positive literal is the constant TRUE (`expression: None`)

    // inputs one group ahead, visited bits two groups ahead: see verdict_kernel
    // (<= 64 passes: a group of the 1k-rule set has one dense pass and two or three non-empty sparse ones; every slot costs a dozen instructions per group
    // whether it is used or not — the rest are fetched inside the group)
    constexpr int kPre = BR == 1 ? 4 : kVerdictPre;
    constexpr int kBitRegs = BR;
    struct Bits {
        unsigned long long w[kBitRegs];
    };
    struct Inputs {
        uint32_t rv[kPre];
        uint32_t base[kPre];
        Bits bits;
        unsigned long long rest[kBitRegs];
        uint32_t flags, n_pairs;
        uint4 pair0;
        uint32_t res0;
        // the lazy comparison variables (VerdictArgs::lazy_var, at most two), RAW — nothing here may wait for a load: a length as the request's END
        // offset (its start is the neighbour lane's end; lane 0's is fetched when an atom is evaluated: rare), the port as it is
        uint32_t lraw[2];
    };
    const uint32_t *my_bits[kBitRegs];
    uint32_t my_base[kBitRegs];
    bool my_live[kBitRegs];
#pragma unroll
    for (int r = 0; r < kBitRegs; r++) {
        const uint32_t ps = (uint32_t)r * 64 + lane;
        my_live[r] = ps < a.n_passes;
        my_bits[r] = nullptr;
        my_base[r] = 0;
        if (my_live[r]) {
            const PassInfo pi = a.passes[ps];
            const uint32_t kind = pi.kind_slot >> 24, sl = pi.kind_slot & 0xFFFFFFu;
            if (kind == 3) my_live[r] = false;
            my_bits[r] = kind == 1 ? a.cand_bits + (size_t)sl * a.bit_words : kind == 2 ? a.visit_bits + (size_t)sl * a.bit_words : nullptr;
            my_base[r] = pi.base;
        }
    }
    auto request_bits = [&](const uint32_t g, Bits &bt) {
        const uint32_t gg = min(g, a.n_groups - 1);
#pragma unroll
        for (int r = 0; r < kBitRegs; r++)
            bt.w[r] = !my_live[r] ? 0ull : my_bits[r] != nullptr ? *reinterpret_cast<const unsigned long long *>(my_bits[r] + 2 * (size_t)gg) : ~0ull;
    };
    auto request_inputs = [&](const uint32_t g, const Bits &bt, Inputs &in) {
        const uint32_t i = g * 64 + lane;
        const bool valid = g < a.n_groups && i < a.n;
        unsigned long long nzp[kBitRegs];
#pragma unroll
        for (int r = 0; r < kBitRegs; r++) nzp[r] = (uint32_t)r * 64 < a.n_passes ? __ballot(bt.w[r] != 0) : 0ull;
#pragma unroll
        for (int q = 0; q < kPre; q++) {
            uint32_t ps = kNone, base = kNone;
            unsigned long long word = 0;
#pragma unroll
            for (int r = 0; r < kBitRegs; r++) {
                if (ps == kNone && nzp[r] != 0) {
                    const int l = __builtin_ctzll(nzp[r]);
                    nzp[r] &= nzp[r] - 1;
                    ps = (uint32_t)r * 64 + (uint32_t)l;
                    word = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(bt.w[r] >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bt.w[r], l);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)my_base[r], l);
                }
            }
            in.base[q] = base;
            in.rv[q] = (ps != kNone && valid && ((word >> lane) & 1ull)) ? a.rec[(size_t)ps * a.n + i] : 0u;
        }
#pragma unroll
        for (int r = 0; r < kBitRegs; r++) in.rest[r] = nzp[r];
        in.bits = bt;
        in.flags = valid ? (uint32_t)a.flags[i] : 0u;
        const uint32_t gg = min(g, a.n_groups - 1);
        {
            // one register, three group-wide words: lane 0 = the attribute kernel's pair count, lane 1 + s = the group's FIRST offset of lazy length
            // variable s (lane 0's start: every other lane's start is its neighbour's end)
            const uint32_t *src = a.ghdr + gg;
#pragma unroll
            for (uint32_t sl = 0; sl < 2; sl++) {
                if (sl >= a.n_lazy_var) break;
                const uint32_t vi = a.lazy_var[sl];
                if (vi != 5u && lane == 1u + sl) src = (vi < 5u ? a.off[vi] : a.hoff[vi - 7u]) + gg * 64u;
            }
            in.n_pairs = *src;
        }
        in.pair0 = a.gpairs[(size_t)gg * a.pair_stride + lane];
        in.res0 = (a.res_words && valid) ? a.res_match[i] : 0u;
        in.lraw[0] = in.lraw[1] = 0u;
        const uint32_t ii = valid ? i : 0u;
#pragma unroll
        for (uint32_t sl = 0; sl < 2; sl++) {
            if (sl >= a.n_lazy_var || (dbg_skip & 2048u)) break;  // (wave-uniform)
            const uint32_t vi = a.lazy_var[sl];
            if (vi == 5u) {
                in.lraw[sl] = a.port[ii];
            } else {
                const uint32_t *o = vi < 5u ? a.off[vi] : a.hoff[vi - 7u];
                in.lraw[sl] = o[ii + 1];
            }
        }
    };
    const uint32_t g_stride = gridDim.x * n_waves;
    Inputs cur;
    Bits b_nxt;
    {
        Bits b_cur;
        request_bits(blockIdx.x * n_waves + wave, b_cur);
        request_inputs(blockIdx.x * n_waves + wave, b_cur, cur);
        request_bits(blockIdx.x * n_waves + wave + g_stride, b_nxt);
    }

    for (uint32_t g = blockIdx.x * n_waves + wave; g < a.n_groups; g += g_stride) {
        const uint32_t i = g * 64 + lane;
        const bool valid = i < a.n;
        const unsigned long long valid_mask = __ballot(valid);
        Inputs nxt;
        request_inputs(g + g_stride, b_nxt, nxt);
        request_bits(g + 2 * g_stride, b_nxt);

        // 1. the rule bitmap starts from the rules that need no positive column (a term made of negations only)
        for (uint32_t k = lane; k < rulew; k += 64) rulebm[k] = k < 64 ? h_always : a.always_rules[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        const uint32_t flags = cur.flags;
        const uint4 *pairs = a.gpairs + (size_t)g * a.pair_stride;
        const uint32_t n_pairs = (uint32_t)__builtin_amdgcn_readlane((int)cur.n_pairs, 0);
        const unsigned long long verified_mask = __ballot(valid && (flags & PWAF_FLAG_CAPTCHA_VERIFIED));

        uint32_t n_entries = 1;  // (entry 0 = TRUE)
        // a column's rules: the rule bitmap gets the trigger list of every appended column
        auto triggers = [&](const uint32_t c) {
            if (dbg_skip & 8u) return;
            const uint32_t kb = LT ? (uint32_t)l_trig_off[c] : a.trig_off[c], ke = LT ? (uint32_t)l_trig_off[c + 1] : a.trig_off[c + 1];
            for (uint32_t k = kb; k < ke; k++) {
                const uint32_t r = LT ? (uint32_t)l_trig_rules[k] : (uint32_t)a.trig_rules[k];
                atomicOr(&rulebm[r >> 5], 1u << (r & 31));
            }
        };
        // `cnt` entries, lane k < cnt holding entry n_entries + k (distinct columns, none of them appended before)
        if (true_triggers && lane == 0) triggers(0);  // (entry 0 = TRUE is never appended: a match-all rule is filed under column 0)
        auto append_batch = [&](const uint32_t c, const uint32_t lo, const uint32_t hi, const uint32_t cnt) {
            if (lane < cnt) {
                const uint32_t e = n_entries + lane;
                const unsigned long long m = ((unsigned long long)hi << 32) | lo;
                if (e < e_cap) {
                    slot[c] = (uint8_t)(e + 1u);
                    vals[e] = m;
                    ecol[e] = (uint16_t)c;
                } else {
                    slot[c] = 255;
                    atomicExch(spill + c, m);  // (a read-modify-write, like everything that touches a spilled word: performed at L2 in order)
                }
                triggers(c);
            }
            n_entries += cnt;
        };
        // the batch registers: the attribute kernel's first pairs are entries as they come (lane k = pair k); the scans' entries are gathered
        // behind them one (wave-uniform) column at a time — one batch, one append, in the common case
        const uint32_t n_first = (dbg_skip & 2u) ? 0u : min(n_pairs, 64u);
        uint32_t b_col = cur.pair0.x, b_lo = cur.pair0.z, b_hi = cur.pair0.w, bn = n_first;
        if (bn == 64) {
            append_batch(b_col, b_lo, b_hi, 64);
            bn = 0;
        }
        auto flush = [&]() {
            if (bn == 0) return;
            append_batch(b_col, b_lo, b_hi, bn);
            bn = 0;
        };
        auto append = [&](const uint32_t c, const unsigned long long m) {
            park96(b_col, b_lo, b_hi, c, m, bn);
            if (++bn == 64) flush();
        };
        // one wave-uniform column that MAY have an entry already (passes with overflow chains): through slot[]
        auto merge_or_append = [&](const uint32_t c, const unsigned long long m) {
            const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)slot[c]);
            if (s == 0) {
                const uint32_t e = n_entries++;
                if (lane == 0) {
                    if (e < e_cap) {
                        slot[c] = (uint8_t)(e + 1u);
                        vals[e] = m;
                        ecol[e] = (uint16_t)c;
                    } else {
                        slot[c] = 255;
                        atomicExch(spill + c, m);
                    }
                    triggers(c);
                }
            } else if (lane == 0) {
                if (s < 255u) vals[s - 1u] |= m;
                else atomicOr(spill + c, m);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        };
        // 2. scan results: per pass one entry per DISTINCT atom its records name (lanes naming the same atom are one ballot)
        auto mark_hits = [&](const uint32_t rv, const uint32_t base) {
            if (__ballot(rv != 0) == 0) return;
            const bool chained = (rv & REC_OVERFLOW) != 0;
            const bool any_chain = __ballot(chained) != 0;
            if (any_chain) {
                flush();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            const uint32_t x0 = chained ? 0u : rv & 0x7FFFu, x1 = chained ? 0u : (rv >> 15) & 0x7FFFu;
            unsigned long long t0 = __ballot(x0 != 0), t1 = __ballot(x1 != 0);
            while ((t0 | t1) != 0) {
                uint32_t xa;
                if (t0 != 0) xa = (uint32_t)__builtin_amdgcn_readlane((int)x0, __builtin_ctzll(t0));
                else xa = (uint32_t)__builtin_amdgcn_readlane((int)x1, __builtin_ctzll(t1));
                const unsigned long long s0 = __ballot(x0 == xa), s1 = __ballot(x1 == xa);
                t0 &= ~s0;
                t1 &= ~s1;
                if (any_chain) merge_or_append(base + xa - 1u, s0 | s1);
                else append(base + xa - 1u, s0 | s1);
            }
            if (any_chain) {
                // the chains, all lanes advancing together: per step the distinct atoms of the lanes that still have one
                uint32_t k = chained ? rv & ~REC_OVERFLOW : kNone;
                while (__ballot(k != kNone) != 0) {
                    uint32_t x = 0;
                    if (k != kNone) {
                        const PoolEntry pe = a.pool[k];
                        x = pe.atom + 1u;
                        k = pe.next;
                    }
                    unsigned long long todo = __ballot(x != 0);
                    while (todo != 0) {
                        const uint32_t xa = (uint32_t)__builtin_amdgcn_readlane((int)x, __builtin_ctzll(todo));
                        const unsigned long long same = __ballot(x == xa);
                        todo &= ~same;
                        merge_or_append(base + xa - 1u, same);
                    }
                }
            }
        };
        if (!(dbg_skip & 1u)) {
#pragma unroll
            for (int q = 0; q < kPre; q++) {
                if (cur.base[q] == kNone) break;
                mark_hits(cur.rv[q], cur.base[q]);
            }
#pragma unroll
            for (int r = 0; r < kBitRegs; r++) {
                unsigned long long m = cur.rest[r];
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint32_t ps = (uint32_t)r * 64 + (uint32_t)l;
                    const unsigned long long word = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(cur.bits.w[r] >> 32), l) << 32) |
                                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cur.bits.w[r], l);
                    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)my_base[r], l);
                    mark_hits((valid && ((word >> lane) & 1ull)) ? a.rec[(size_t)ps * a.n + i] : 0u, base);
                }
            }
        }
        // 2b. the specialized residual program's result words: one entry per rule that matched anybody in the group
        for (uint32_t w = 0; w < a.res_words; w++) {
            const uint32_t mw = w == 0 ? cur.res0 : (valid ? a.res_match[(size_t)w * a.n + i] : 0u);
            if (__ballot(mw != 0u) == 0ull) continue;
            for (uint32_t any = wave_or(mw); any; any &= any - 1u) {
                const uint32_t k = (uint32_t)__builtin_ctz(any);
                append(a.res_base + 32u * w + k, __ballot(((mw >> k) & 1u) != 0u));
            }
        }
        flush();
        // 3. the attribute kernel's pairs beyond the first 64 (rare)
        if (!(dbg_skip & 2u)) {
            for (uint32_t p0 = 64; p0 < n_pairs; p0 += 64) {
                const uint32_t cnt = min(64u, n_pairs - p0);
                uint4 pr = make_uint4(0u, 0u, 0u, 0u);
                if (lane < cnt) pr = pairs[p0 + lane];
                append_batch(pr.x, pr.z, pr.w, cnt);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // a column's word: two dependent LDS reads, no branch (entry 0 = TRUE stands in for a clean column's index) — unless the group
        // appended more entries than LDS holds (wave-uniform, rare): then a spilled word comes from L2, where the ORs were performed
        const bool spilled = n_entries > e_cap;
        auto col_word = [&](const uint32_t c) -> unsigned long long {
            const uint32_t s = slot[c];
            if (!spilled) {
                const unsigned long long v = vals[s ? s - 1u : 0u];
                return s ? v : 0ull;
            }
            if (s == 0) return 0ull;
            return s < 255u ? vals[s - 1u] : __hip_atomic_load(spill + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };

        // 4. ordered compaction of the rule bitmap into the candidate list (ascending rule index = evaluation order)
        uint32_t n_cand = 0;
        for (uint32_t wb = 0; wb < rulew && !(dbg_skip & 16u); wb += 64) {
            const uint32_t word = wb + lane < rulew ? rulebm[wb + lane] : 0u;
            const uint32_t pc = (uint32_t)__builtin_popcount(word), incl = wave_scan_add(pc);
            uint32_t pos = n_cand + incl - pc, wrd = word;
            while (wrd) {
                cand[pos++] = (uint16_t)((wb + lane) * 32 + (uint32_t)__builtin_ctz(wrd));
                wrd &= wrd - 1;
            }
            n_cand += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        // 5. evaluate candidates: one lane per rule, 64 requests per ALU op; first match wins (see verdict_kernel)
        unsigned long long pending = valid_mask;
        uint32_t n_exact = 0;
        bool undecided = valid;
        uint32_t my_action = PWAF_ACTION_ALLOW, my_rule = PWAF_RULE_NONE;
        for (uint32_t base = 0; base < n_cand && pending != 0 && !(dbg_skip & 32u); base += 64) {
            unsigned long long fire = 0, eff_mask = 0;
            uint32_t lit_off = 0, lit_cnt = 0;
            bool lazy_seen = false;
            if (base + lane < n_cand) {
                const uint32_t my_cand = cand[base + lane];
                uint32_t eff_u, eff_v;
                if (LT) {
                    const uint2 hdr = l_rules[my_cand];
                    lit_off = hdr.x;
                    lit_cnt = hdr.y & 0xFFFFu;
                    eff_u = (hdr.y >> 16) & 0xFFu;
                    eff_v = hdr.y >> 24;
                } else {
                    const DevRule dr = a.rules[my_cand];
                    lit_off = dr.lit_off;
                    lit_cnt = dr.lit_cnt;
                    eff_u = dr.eff_unverified;
                    eff_v = dr.eff_verified;
                }
                unsigned long long acc_or = 0, acc_and = ~0ull;
                for (uint32_t k = lit_off; k < lit_off + lit_cnt; k += 4) {
                    uint32_t lit[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) lit[q] = k + q < lit_off + lit_cnt ? (LT ? l_lits[k + q] : a.lits[k + q]) : 0u;
                    unsigned long long cw[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) cw[q] = col_word(lit[q] & LIT_ATOM_MASK);  // (a lazy literal's index is below n_cols too: the read is harmless)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        // a LAZY comparison atom reads as TRUE here, negated or not: what comes out is a superset of the rule's matches
                        const unsigned long long t = (lit[q] & LIT_NEG) ? ~cw[q] : cw[q];
                        acc_and &= (lit[q] & LIT_LAZY) ? ~0ull : t;
                        lazy_seen = lazy_seen || (lit[q] & LIT_LAZY) != 0u;
                        if (lit[q] & LIT_TERM_END) {
                            acc_or |= acc_and;
                            acc_and = ~0ull;
                        }
                    }
                }
                eff_mask = (eff_u ? ~verified_mask : 0ull) | (eff_v ? verified_mask : 0ull);
                fire = acc_or & eff_mask;
            }
            // Rules with lazy comparison atoms whose OTHER literals hold for somebody (rare): the rule again, exactly, one request per lane —
            // an eager literal's bit from its column word, a lazy one from the request's own value (fetched with the group's inputs)
            if (dbg_skip & 512u) n_exact += (uint32_t)__builtin_popcountll(__ballot(lazy_seen && fire != 0));
            for (unsigned long long need = (dbg_skip & 1024u) ? 0ull : __ballot(lazy_seen && fire != 0); need != 0; need &= need - 1) {
                const int j = __builtin_ctzll(need);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)lit_off, j), cn = (uint32_t)__builtin_amdgcn_readlane((int)lit_cnt, j);
                const unsigned long long em = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(eff_mask >> 32), j) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)eff_mask, j);
                bool r_or = false, r_and = true;
                for (uint32_t k = lo; k < lo + cn; k++) {
                    const uint32_t lit = (uint32_t)__builtin_amdgcn_readfirstlane((int)(LT ? l_lits[k] : a.lits[k]));
                    bool bit;
                    if (lit & LIT_LAZY) {
                        const uint32_t cc = lit & 0xFFFFu, sl = (lit >> 17) & 1u;
                        const uint32_t raw = sl ? cur.lraw[1] : cur.lraw[0];  // (no dynamic index: the inputs stay in registers)
                        uint32_t v = raw;
                        if (a.lazy_var[sl] != 5u) {  // length = end - the neighbour's end (lane 0: the group's first offset)
                            const uint32_t st0 = sl ? (uint32_t)__builtin_amdgcn_readlane((int)cur.n_pairs, 2) : (uint32_t)__builtin_amdgcn_readlane((int)cur.n_pairs, 1);
                            v = raw - (uint32_t)__builtin_amdgcn_update_dpp((int)st0, (int)raw, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
                        }
                        bit = (lit & 0x10000u) ? v <= cc : v == cc;
                        if (lit & 0x40000u) bit = !bit;  // (the atom's polarity on the device: engine.cpp, POLARITY)
                    } else {
                        bit = ((col_word(lit & LIT_ATOM_MASK) >> lane) & 1ull) != 0ull;
                    }
                    if (lit & LIT_NEG) bit = !bit;
                    r_and = r_and && bit;
                    if (lit & LIT_TERM_END) {
                        r_or = r_or || r_and;
                        r_and = true;
                    }
                }
                const unsigned long long exact = __ballot(r_or && valid) & em;
                uint32_t f_lo = (uint32_t)fire, f_hi = (uint32_t)(fire >> 32);
                park64(f_lo, f_hi, exact, (uint32_t)j);
                fire = ((unsigned long long)f_hi << 32) | f_lo;
            }
            unsigned long long firing = __ballot(fire != 0);
            uint32_t first = kNone;
            while (firing != 0 && pending != 0) {
                const int j = __builtin_ctzll(firing);
                firing &= firing - 1;
                const unsigned long long fj = (((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(fire >> 32), j) << 32) |
                                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)fire, j)) & pending;
                if (fj & mybit) first = (uint32_t)j;
                pending &= ~fj;
            }
            if (undecided && first != kNone) {
                undecided = false;
                const uint32_t jc = cand[base + first];
                uint32_t eff_u, eff_v;
                if (LT) {
                    const uint32_t y = l_rules[jc].y;
                    eff_u = (y >> 16) & 0xFFu;
                    eff_v = y >> 24;
                    const uint32_t pi = l_pub[jc];
                    my_rule = pi >= 0xFFF0u ? 0xFFFF0000u | pi : pi;
                } else {
                    const DevRule dr = a.rules[jc];
                    eff_u = dr.eff_unverified;
                    eff_v = dr.eff_verified;
                    my_rule = dr.public_idx;
                }
                my_action = (verified_mask & mybit) ? eff_v : eff_u;
            }
        }
        if (dbg_skip & 512u) my_rule = n_exact;  // profiling aid: rules evaluated exactly for their lazy comparison atoms
        if (dbg_skip & 64u) my_rule = n_cand | (n_entries << 16);  // profiling aid: candidates and entries of the group instead of the deciding rule

        // 6. outputs
        if (valid) {
            uint2 v;
            v.x = my_action;
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
            const unsigned long long hit = m_block | m_captcha | m_bypass;
            if (hit) {
                uint32_t basei = 0;
                if (lane == 0) basei = atomicAdd(a.n_matches, (uint32_t)__builtin_popcountll(hit));
                basei = __builtin_amdgcn_readfirstlane(basei);
                if (hit & mybit) a.match_idx[basei + (uint32_t)__builtin_popcountll(hit & lt_mask)] = i;
            }
        }
        // 7. the group cleans up after itself: the slot bytes of its entries (all of slot[] when entries went beyond e_cap: their columns
        //    are not listed)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (n_entries <= e_cap) {
            for (uint32_t e = 1u + lane; e < n_entries; e += 64) slot[ecol[e]] = 0;
        } else {
            for (uint32_t k = lane; k < slot_words; k += 64) reinterpret_cast<uint32_t *>(slot)[k] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane == 0) slot[0] = 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        cur = nxt;
    }
    __syncthreads();
    unsigned long long *tally = reinterpret_cast<unsigned long long *>(lds);
    if (lane == 0) {
        tally[wave * 4 + PWAF_ACTION_ALLOW] = cnt_allow;
        tally[wave * 4 + PWAF_ACTION_BLOCK] = cnt_block;
        tally[wave * 4 + PWAF_ACTION_CAPTCHA] = cnt_captcha;
        tally[wave * 4 + PWAF_ACTION_BYPASS] = cnt_bypass;
    }
    __syncthreads();
    if (a.counts != nullptr && tid < 4) {
        unsigned long long sum = 0;
        for (uint32_t wv = 0; wv < n_waves; wv++) sum += tally[wv * 4 + tid];
        if (sum) atomicAdd(&a.counts[tid], sum);
    }
}

// -------------------------------------------------------------------------------------------------
// attributes: everything about a request that is NOT a string scan — GeoIP record, ip-list membership, country / integer-set
// membership, length / port / asn comparisons — reduced per 64-request group to a list of (column, 64-request mask) pairs
// -------------------------------------------------------------------------------------------------
// The lookups are chains of dependent gathers (DIR-24 / trie levels, membership rows); the kernel needs no LDS and does not depend on
// the scans, so it runs BESIDE them on the engine's side stream at the lowest wave priority. One wave = one 64-request group; the
// verdict kernel only copies the group's pairs into its column file.
//
// What keeps it short (round 2: it had become the longest kernel, 2.8 GB fetched for 0.38 GB of input):
//   * everything that is a function of the GeoIP record — country-table bits, asn-set bits and the asn comparisons — is folded at
//     engine creation into a CLASS row, and records with equal rows share a class (a few hundred classes for 600k records: the
//     table is cache-resident, class 0 = "no predicate holds" needs no row at all);
//   * IPv4: ONE gather into a 2^24 x 4-byte table (DIR-24-8, 64 MiB) yields class | membership-set id << 16; prefixes longer than
//     /24 and oversized ids escape to an 8-byte side table and continue in the 8-bit trie nodes;
//   * a group's inputs are requested TWO groups ahead and its DIR-24 / root entries ONE group ahead, so the long-latency loads of the
//     next groups are in flight while the current group's rows are transposed (few waves per CU: nothing else hides them).
static constexpr uint32_t DIR_ESCAPE = 0x80000000u;

// ipres_kernel (round 3): the address lookups of the attribute path as a kernel of their own — one lane per request, ~20 registers,
// full occupancy. Round 2 did them inside attr_kernel, whose transposes need ~128 registers (4 waves per SIMD) and whose waves each
// walked one 64-request group at a time: every group waited for its own DIR-24 line (a miss to the Infinity Cache / HBM) and then
// for the trie levels of its few IPv6 lanes, with 16 waves per CU to hide it all — 0.44 ms alone for 10M requests, the longest
// kernel of the step. Here 32 waves per CU overlap those waits, and an IPv4 address behind the DIR-24 table costs ONE gather (round 2
// also fetched both 16-bit root entries for every lane and threw them away). Output: the GeoIP CLASS and the ip-list membership SET
// of every request, packed into one word (class | set << 16) when both fit 16 bits (else two words).
template <bool PACKED>
__global__ __launch_bounds__(256) void ipres_kernel(VerdictArgs a) {
    // A lookup is a chain of dependent accesses (address bytes -> first level -> run bitmap -> value; IPv6: root -> trie nodes), each an
    // L2 / Infinity-Cache round trip: with one request per lane at a time the kernel was latency-bound even at 32 waves per CU (0.24 ms
    // for 10M requests, 19 chains per lane one after the other). Every lane now walks U = 4 requests in lockstep: each phase issues
    // the loads of all four before any is used.
    constexpr uint32_t U = 4;
    const bool from_row = a.asn == nullptr;
#ifdef PWAF_PROFILING
    // timing experiments (wrong results): bit 16 = no table lookups for IPv4, bit 17 = no trie walks for IPv6
    const bool dir = a.dir_chunks != nullptr && !((a.debug_skip >> 16) & 1u);
    const bool skip_v6 = (a.debug_skip >> 17) & 1u, skip_v4 = (a.debug_skip >> 16) & 1u;
#else
    const bool dir = a.dir_chunks != nullptr;
    constexpr bool skip_v6 = false, skip_v4 = false;
#endif
    const uint32_t T = gridDim.x * 256u;
    for (uint32_t i0 = blockIdx.x * 256u + threadIdx.x; i0 < a.n; i0 += T * U) {
        uint32_t idx[U], ipw[U][4], eg[U], ei[U], k[U];
        bool live[U], v6[U], geo_walk[U], chunked[U];
        uint4 raw[U];
        uint32_t v6b[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            idx[u] = i0 + u * T;
            live[u] = idx[u] < a.n;
            const uint32_t j = live[u] ? idx[u] : i0;
            raw[u] = *reinterpret_cast<const uint4 *>(a.ip + (size_t)j * 16);
            v6b[u] = a.ip_is_v6[j];
        }
        uint32_t first[U], top16[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            ipw[u][0] = raw[u].x; ipw[u][1] = raw[u].y; ipw[u][2] = raw[u].z; ipw[u][3] = raw[u].w;
            v6[u] = v6b[u] != 0;
            // GeoipDB::lookup, pingoo/geoip.rs:73-91: loopback / multicast are "not found"
            geo_walk[u] = false;
            if (from_row && a.has_geo) {
                if (!v6[u]) {
                    const uint32_t b0 = ipw[u][0] & 0xFFu;
                    geo_walk[u] = !(b0 == 127u || (b0 & 0xF0u) == 0xE0u);
                } else {
                    const bool loopback = ipw[u][0] == 0 && ipw[u][1] == 0 && ipw[u][2] == 0 && ipw[u][3] == 0x01000000u;
                    geo_walk[u] = !(loopback || (ipw[u][0] & 0xFFu) == 0xFFu);
                }
            }
            top16[u] = (ip_byte(ipw[u], 0) << 8) | ip_byte(ipw[u], 1);
            // phase 1: the /16's line of the compressed DIR table (IPv4) — bitmap word, run counts, where the values live — or the two
            // 16-bit roots
            eg[u] = ei[u] = TRIE_LEAF;
            chunked[u] = !v6[u] && dir;
            first[u] = 0;
            if (!chunked[u] && !(v6[u] ? skip_v6 : skip_v4)) {
                if (geo_walk[u]) eg[u] = (v6[u] ? a.geo_root6 : a.geo_root4)[top16[u]];  // (the engine substitutes an all-leaf root for a family without prefixes)
                if (a.n_ip_lists) ei[u] = (v6[u] ? a.ip_root6 : a.ip_root4)[top16[u]];
            }
        }
        // phase 1b: the summary bit of the address's block of /24s (L2-resident bitmap): 0 = the table's most common entry, no gather
        uint32_t look[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            look[u] = chunked[u] ? 1u : 0u;
            if (chunked[u] && a.dir_summary != nullptr) {
                const uint32_t blk = ((top16[u] << 8) | ip_byte(ipw[u], 2)) >> a.dir_sum_shift;
                look[u] = (a.dir_summary[blk >> 5] >> (blk & 31u)) & 1u;
            }
        }
        uint4 rec4[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            rec4[u] = make_uint4(0u, 0u, 0u, 0u);
            // ONE 16-byte gather: the record of the address's group of 32 /24s
            if (look[u]) rec4[u] = *reinterpret_cast<const uint4 *>(a.dir_chunks + (size_t)top16[u] * kDirChunkWords + 4u * (ip_byte(ipw[u], 2) >> 5));
        }
        // phase 2: the run's entry — carried into the group, the first run inside it, or (rarely) a further run from dir_vals
        uint32_t e24[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            e24[u] = a.dir_common;
            if (look[u]) {
                const uint32_t rank = (uint32_t)__builtin_popcount(rec4[u].x & (0xFFFFFFFFu >> (31u - (ip_byte(ipw[u], 2) & 31u))));  // run starts at or before the /24, inside its group
                e24[u] = rank == 0 ? rec4[u].y : rank == 1 ? rec4[u].z : a.dir_vals[rec4[u].w + rank - 2u];
            }
        }
        // phase 4: escapes (a prefix longer than /24, or an id too large for the packed entry: rare)
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            k[u] = 2;
            if (!v6[u] && dir) {
                k[u] = 3;
                if (e24[u] & DIR_ESCAPE) {
                    const uint2 esc = a.dir_esc[e24[u] & ~DIR_ESCAPE];
                    eg[u] = esc.x;
                    ei[u] = esc.y;
                } else {
                    eg[u] = TRIE_LEAF | (e24[u] & 0xFFFFu);
                    ei[u] = TRIE_LEAF | (e24[u] >> 16);
                }
            }
            if (!geo_walk[u]) eg[u] = TRIE_LEAF | a.geo_default;  // no lookup: the default record's class
            if (a.n_ip_lists == 0) ei[u] = TRIE_LEAF;
        }
        // phase 5: the remaining trie levels (IPv6, escapes), all four walks advancing together
        for (;;) {
            bool more = false;
#pragma unroll
            for (uint32_t u = 0; u < U; u++) more = more || !((eg[u] & ei[u]) & TRIE_LEAF);
            if (!more) break;
            uint32_t ng[U], ni[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t byte = ip_byte(ipw[u], k[u] < 16 ? k[u] : 15u);
                ng[u] = (eg[u] & TRIE_LEAF) ? eg[u] : a.geo_nodes[(size_t)eg[u] * 256 + byte];
                ni[u] = (ei[u] & TRIE_LEAF) ? ei[u] : a.ip_nodes[(size_t)ei[u] * 256 + byte];
            }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                if (!((eg[u] & ei[u]) & TRIE_LEAF)) k[u]++;
                eg[u] = ng[u];
                ei[u] = ni[u];
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            if (!live[u]) continue;
            const uint32_t cls = eg[u] & ~TRIE_LEAF, set_id = ei[u] & ~TRIE_LEAF;
            if (PACKED) a.ipres[idx[u]] = cls | (set_id << 16);
            else reinterpret_cast<uint2 *>(a.ipres)[idx[u]] = make_uint2(cls, set_id);
        }
    }
}

struct AttrIn {
    uint32_t cls, set_id;
    uint32_t port, len[5], asn, country;
    uint32_t sstart, slen;  // the short-literal field's value: offset and length
};

// (__launch_bounds__(256, 7) — 72 registers, 7 waves per SIMD at the price of 3-5 spilled dwords — was measured: 0.229 ms alone against
// 0.227, nothing.)
// SMALL: the rule set's membership rows fit 4 ip-set words, 2 country words and one word each of port sets, asn sets and asn
// comparisons (a 1k-rule set with 124 CIDR lists does): 9 row registers instead of 36.
template <bool PACKED, bool SMALL>
__global__ __launch_bounds__(256) void attr_kernel(VerdictArgs a) {
    constexpr uint32_t SW = SMALL ? 4 : kSetWordsMax, CW = SMALL ? 2 : kCcWordsMax, IW = SMALL ? 1 : kIntWordsMax, QW = SMALL ? 1 : kAcmpWordsMax;
#ifdef PWAF_PROFILING
    if (a.debug_skip & 0x80000000u) __builtin_amdgcn_s_setprio(3);  // timing experiment
#endif
    const uint32_t lane = threadIdx.x & 63, wave = wave_index();
    const unsigned long long lt_mask = (1ull << lane) - 1;
    const bool from_row = a.asn == nullptr;  // asn / country come from the engine's own GeoIP record (or its default)
#ifdef PWAF_PROFILING
    // timing experiments (wrong results): which part of the kernel costs what
    const bool skip_rows = (a.debug_skip >> 18) & 1u, skip_transpose = (a.debug_skip >> 19) & 1u, skip_cmp = (a.debug_skip >> 20) & 1u;
#else
    constexpr bool skip_rows = false, skip_transpose = false, skip_cmp = false;
#endif
    // group-invariant: the first 64 comparison atoms, one per lane (col | code << 24, constant)
    uint32_t h_col = 0, h_c = 0;
    if (lane < a.n_cmp) {
        h_col = a.cmp[lane].col;
        h_c = a.cmp[lane].c;
    }
    const uint32_t g_stride = gridDim.x * 4;
    // stage A: the fixed-width inputs of a group (unconditional loads, clamped indices: no branch, so no wait is forced)
    auto load_in = [&](const uint32_t g, AttrIn &in) {
        const uint32_t i0 = g * 64 + lane;
        const uint32_t i = (g < a.n_groups && i0 < a.n) ? i0 : 0u;
        if (PACKED) {
            const uint32_t w = a.ipres[i];
            in.cls = w & 0xFFFFu;
            in.set_id = w >> 16;
        } else {
            const uint2 w = reinterpret_cast<const uint2 *>(a.ipres)[i];
            in.cls = w.x;
            in.set_id = w.y;
        }
        in.port = a.port[i];
#pragma unroll
        for (int f = 0; f < 5; f++) in.len[f] = ((a.cmp_vars >> f) & 1u) ? a.off[f][i + 1] - a.off[f][i] : 0u;  // (wave-uniform: only the lengths some comparison atom reads)
        in.asn = from_row ? 0u : a.asn[i];
        in.country = from_row ? 0u : (uint32_t)a.country[i];
        in.sstart = a.n_short ? a.short_off[i] : 0u;
        in.slen = a.n_short ? a.short_off[i + 1] - in.sstart : 0u;
    };
    // stage B: the short-literal field's first 8 bytes (arenas carry 16 readable slack bytes)
    auto load_short = [&](const AttrIn &in, uint32_t &s_lo, uint32_t &s_hi) {
        typedef uint2 __attribute__((aligned(1))) uint2_u;
        uint2 sv = make_uint2(0u, 0u);
        if (a.n_short) sv = *reinterpret_cast<const uint2_u *>(a.short_data + in.sstart);
        s_lo = sv.x;
        s_hi = sv.y;
    };
    AttrIn cur, nxt;
    uint32_t c_slo, c_shi;
    const uint32_t g0 = blockIdx.x * 4 + wave;
    load_in(g0, cur);
    load_in(g0 + g_stride, nxt);
    load_short(cur, c_slo, c_shi);
    // group-invariant: the first 64 short-literal atoms, one per lane
    ShortAtom h_short{0, 0, 0, 0};
    if (lane < a.n_short) h_short = a.short_atoms[lane];

    for (uint32_t g = g0; g < a.n_groups; g += g_stride) {
        const uint32_t i = g * 64 + lane;
        const bool valid = i < a.n;
        const unsigned long long valid_mask = __ballot(valid);
        uint4 *pairs = a.gpairs + (size_t)g * a.pair_stride;
        uint32_t n_pairs = 0;  // wave-uniform
        // lane-private (column, mask) -> the group's pair list, in lane order
        auto emit_pairs = [&](const bool has, const uint32_t c, const uint32_t lo, const uint32_t hi) {
            const unsigned long long em = __ballot(has);
            if (has) pairs[n_pairs + (uint32_t)__builtin_popcountll(em & lt_mask)] = make_uint4(c, 0u, lo, hi);
            n_pairs += (uint32_t)__builtin_popcountll(em);
        };
        const uint32_t port = cur.port;
        const uint32_t cls = cur.cls, set_id = cur.set_id;

        // ---- 1. membership rows: every row word of the request requested together, then ONE wait ----
        uint32_t asn = cur.asn, r_geo = 0;
        if (!from_row) {
            const uint32_t c0 = (cur.country & 0xFFu) - 'A', c1 = (cur.country >> 8) - 'A';
            r_geo = (c0 < 26u && c1 < 26u) ? c0 * 26u + c1 : 23u * 26u + 23u;  // invalid input is treated as "XX"
        }
        uint32_t r_int[2] = {0, 0};
#pragma unroll
        for (int var = 0; var < 2; var++) {
            // integer sets: ONE binary search per request over the union of every set tested against the variable
            if (!valid || a.iu_n[var] == 0 || (var == 1 && from_row)) continue;
            const long long v = var == VAR_PORT ? (long long)port : (long long)asn;
            uint32_t lo = 0, hi = a.iu_n[var];
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (a.iu_vals[var][mid] < v) lo = mid + 1;
                else hi = mid;
            }
            if (lo < a.iu_n[var] && a.iu_vals[var][lo] == v) r_int[var] = lo + 1;
        }
        // sources (kernels.h kSrc*): ip-set row, country words, port-set row, asn-set words, asn comparisons
        const uint32_t *srow = a.set_masks + (size_t)set_id * a.set_words;
        const uint32_t *crow = from_row ? a.class_rows + (size_t)cls * a.class_words : a.country_masks + (size_t)r_geo * a.cc_words;
        const uint32_t *prow = a.iu_masks[0] + (size_t)r_int[0] * a.iu_words[0];
        const uint32_t *arow = from_row ? crow + a.cc_words : a.iu_masks[1] + (size_t)r_int[1] * a.iu_words[1];
        const uint32_t *qrow = crow + a.cc_words + a.iu_words[1];
        const bool rows_on = valid && !skip_rows;
        const bool have_s = rows_on && a.n_ip_lists && set_id, have_c = rows_on && (from_row ? cls != 0 : true), have_p = rows_on && r_int[0],
                   have_a = rows_on && (from_row ? cls != 0 : r_int[1] != 0), have_q = rows_on && from_row && cls != 0;
        uint32_t rs[SW], rc[CW], rp[IW], ra[IW], rq[QW];
#pragma unroll
        for (uint32_t q = 0; q < SW; q++) rs[q] = (q < a.set_words && have_s) ? srow[q] : 0u;
#pragma unroll
        for (uint32_t q = 0; q < CW; q++) rc[q] = (q < a.cc_words && have_c) ? crow[q] : 0u;
#pragma unroll
        for (uint32_t q = 0; q < IW; q++) rp[q] = (q < a.iu_words[0] && have_p) ? prow[q] : 0u;
#pragma unroll
        for (uint32_t q = 0; q < IW; q++) ra[q] = (q < a.iu_words[1] && have_a) ? arow[q] : 0u;
#pragma unroll
        for (uint32_t q = 0; q < QW; q++) rq[q] = (q < a.acmp_words && have_q) ? qrow[q] : 0u;

        // ---- 2. prefetch: inputs of the group after next, the next group's short-literal bytes ----
        AttrIn nn;
        load_in(g + 2 * g_stride, nn);
        uint32_t n_slo, n_shi;
        load_short(nxt, n_slo, n_shi);

        // ---- 3. transposes: one ballot per bit that ANY of the 64 requests has set (wave-wide OR first, so absent bits cost
        //         nothing); lane b keeps bit b's request mask and owns that atom's pair ----
        auto transpose = [&](const uint32_t w, const uint32_t src_word) {
            const uint32_t orw0 = wave_or(w);
            if (orw0 == 0) return;
            // (the mask is PARKED in lane b with v_writelane: two vector instructions per bit — a compare of the lane id and two
            // selects, with the mask moved into vector registers first, were seven; the kernel is bound by vector-ALU issue)
            uint32_t mine_lo = 0, mine_hi = 0;
            for (uint32_t orw = orw0; orw; orw &= orw - 1) {
                const uint32_t b = (uint32_t)__builtin_ctz(orw);
                const unsigned long long m = __ballot((w & (1u << b)) != 0u);
                park64(mine_lo, mine_hi, m, b);
            }
            const bool owner = lane < 32 && ((orw0 >> lane) & 1u);
            const uint32_t c = owner ? a.bit_col[src_word * 32 + lane] : 0u;  // (source word, bit) -> column, 0 = no such atom
            emit_pairs(c != 0, c, mine_lo, mine_hi);
        };
        if (!skip_transpose) {
#pragma unroll
            for (uint32_t q = 0; q < SW; q++) if (q < a.set_words) transpose(rs[q], kSrcSet + q);  // (uniform conditions)
#pragma unroll
            for (uint32_t q = 0; q < CW; q++) if (q < a.cc_words) transpose(rc[q], kSrcCc + q);
#pragma unroll
            for (uint32_t q = 0; q < IW; q++) if (q < a.iu_words[0]) transpose(rp[q], kSrcPort + q);
#pragma unroll
            for (uint32_t q = 0; q < IW; q++) if (q < a.iu_words[1]) transpose(ra[q], kSrcAsn + q);
#pragma unroll
            for (uint32_t q = 0; q < QW; q++) if (q < a.acmp_words) transpose(rq[q], kSrcAcmp + q);
        }

        // Comparison atoms (lengths, port, asn against constants): the engine has reduced them to `v == c` / `v <= c` on 32-bit
        // values and tagged each with code = 2 * variable + operator; an atom is a scalar broadcast of its constant, one
        // vector compare and a ballot parked in the atom's lane. (asn comparisons of engine-resolved records are class-row bits.)
        for (uint32_t base = 0; base < a.n_cmp && !skip_cmp; base += 64) {
            uint32_t m_col = h_col, m_c = h_c;
            if (base != 0) {  // more than 64 comparison atoms: the later chunks are re-read per group
                m_col = m_c = 0;
                if (base + lane < a.n_cmp) {
                    m_col = a.cmp[base + lane].col;
                    m_c = a.cmp[base + lane].c;
                }
            }
            const uint32_t my_code = base + lane < a.n_cmp ? (m_col >> 24) & 0x7Fu : 0xFFu;
            const unsigned long long flip_atoms = __ballot(base + lane < a.n_cmp && ((m_col >> 31) & 1u) != 0u);  // atoms evaluated complemented (engine.cpp, POLARITY)
            uint32_t acc_lo = 0, acc_hi = 0;
            auto cmp_var = [&](const uint32_t v, const int vi) {
#pragma unroll
                for (int op = 0; op < 2; op++) {
                    unsigned long long todo = __ballot(my_code == (uint32_t)(2 * vi + op));
                    while (todo) {
                        const uint32_t j = (uint32_t)__builtin_ctzll(todo);
                        todo &= todo - 1;
                        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)m_c, (int)j);
                        unsigned long long m = __ballot(op == 0 ? v == c : v <= c);
                        if ((flip_atoms >> j) & 1ull) m = ~m;
                        park64(acc_lo, acc_hi, m & valid_mask, j);
                    }
                }
            };
#pragma unroll
            for (int f = 0; f < 5; f++)
                if ((a.cmp_vars >> f) & 1u) cmp_var(cur.len[f], f);
            if ((a.cmp_vars >> 5) & 1u) cmp_var(port, 5);
            if (!from_row && ((a.cmp_vars >> 6) & 1u)) cmp_var(asn, 6);
            for (uint32_t f = 0; f < a.n_hlen; f++) {  // EXTENSION: lengths of header columns (rare: fetched here, not prefetched)
                const uint32_t ii = valid ? i : 0u;
                cmp_var(a.hoff[f][ii + 1] - a.hoff[f][ii], 7 + (int)f);
            }
            emit_pairs((acc_lo | acc_hi) != 0, m_col & 0xFFFFFFu, acc_lo, acc_hi);
        }
        // Short-literal atoms (kernels.h: ShortAtom): the field's first 8 bytes against each literal under its length mask — a
        // scalar broadcast of the atom, two vector compares and a ballot parked in the atom's lane.
        for (uint32_t base = 0; base < a.n_short; base += 64) {
            ShortAtom m = h_short;
            if (base != 0) {
                m = ShortAtom{0, 0, 0, 0};
                if (base + lane < a.n_short) m = a.short_atoms[base + lane];
            }
            const uint32_t cnt = min(64u, a.n_short - base);
            uint32_t acc_lo = 0, acc_hi = 0;
            for (uint32_t j = 0; j < cnt; j++) {
                const uint32_t le = (uint32_t)__builtin_amdgcn_readlane((int)m.len_exact, (int)j), len = le & 0xFFu;
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)m.lit_lo, (int)j), hi = (uint32_t)__builtin_amdgcn_readlane((int)m.lit_hi, (int)j);
                const uint32_t mlo = len >= 4 ? 0xFFFFFFFFu : (1u << (8 * len)) - 1u, mhi = len >= 8 ? 0xFFFFFFFFu : len > 4 ? (1u << (8 * (len - 4))) - 1u : 0u;
                // (the length test is chosen by a SCALAR branch — `exact` is the atom's, not the request's — and the two masked compares
                // are one: xor, xor-and, and-or, compare)
                const unsigned long long len_ok = (le >> 8) ? __ballot(cur.slen == len) : __ballot(cur.slen >= len);
                const unsigned long long hit = __ballot((((c_slo ^ lo) & mlo) | ((c_shi ^ hi) & mhi)) == 0u) & len_ok & valid_mask;
                park64(acc_lo, acc_hi, hit, j);
            }
            emit_pairs((acc_lo | acc_hi) != 0, m.col, acc_lo, acc_hi);
        }
        if (lane == 0) a.ghdr[g] = n_pairs;
        cur = nxt;
        nxt = nn;
        c_slo = n_slo;
        c_shi = n_shi;
    }
}

// Flattens the first 24 bits of the GeoIP trie (leaves = class ids) and the ip-list trie (leaves = membership-set ids), IPv4 family,
// into one table of 4-byte entries: class | set << 16 when both walks end within 24 bits with ids that fit (class < 65536,
// set < 32768), otherwise DIR_ESCAPE | index of an 8-byte {geo entry, set entry} pair from which the attribute kernel continues.
// Two launches: `esc == nullptr` only counts the escapes.
__global__ __launch_bounds__(256) void dir24_kernel(VerdictArgs a, uint32_t *out, uint2 *esc, uint32_t *esc_count) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;  // the top 24 address bits
    if (x >= (1u << 24)) return;
    uint32_t eg = a.geo_root4[x >> 8];
    if (!(eg & TRIE_LEAF)) eg = a.geo_nodes[(size_t)eg * 256 + (x & 0xFFu)];
    uint32_t ei = a.ip_root4[x >> 8];
    if (!(ei & TRIE_LEAF)) ei = a.ip_nodes[(size_t)ei * 256 + (x & 0xFFu)];
    const uint32_t vg = eg & ~TRIE_LEAF, vi = ei & ~TRIE_LEAF;
    if ((eg & ei & TRIE_LEAF) && vg < 65536u && vi < 32768u) {
        if (out) out[x] = vg | (vi << 16);
        return;
    }
    const uint32_t idx = atomicAdd(esc_count, 1u);
    if (esc) {
        esc[idx] = make_uint2(eg, ei);
        out[x] = DIR_ESCAPE | idx;
    }
}

int launch_dir24(const VerdictArgs &a, void *out, void *esc, void *esc_count, void *stream) {
    hipLaunchKernelGGL(dir24_kernel, dim3((1u << 24) / 256), dim3(256), 0, (hipStream_t)stream, a, (uint32_t *)out, (uint2 *)esc, (uint32_t *)esc_count);
    return (int)hipGetLastError();
}

// address lookups: one lane per request at full occupancy (grid-stride; 8 workgroups of 256 per CU)
int launch_ipres(const VerdictArgs &a, void *stream) {
    if (a.n == 0) return 0;
#ifdef PWAF_PROFILING
    static const uint32_t forced = getenv("PWAF_IPRES_BLOCKS") ? (uint32_t)atoi(getenv("PWAF_IPRES_BLOCKS")) : 0u;
#else
    const uint32_t forced = 0;
#endif
    const uint32_t blocks = std::min<uint32_t>((a.n + 255) / 256, forced ? forced : 8 * std::max(1u, a.attr_blocks));
    if (a.ipres_packed) hipLaunchKernelGGL(ipres_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ipres_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

int launch_attr(const VerdictArgs &a, void *stream) {
    if (a.n == 0) return 0;
    // rows, transposes, comparisons (after launch_ipres on the same stream). A persistent grid of six workgroups per CU at the default wave priority 0, while the filter
    // waves raise theirs to 3: the attribute waves take the issue slots the filter leaves idle instead of competing for them
    // (measured on MI355X next to the stream filter: 512 / 768 / 1024 / 1536 / 2048 workgroups -> 2.00 / 2.00 / 1.98 / 1.94 / 1.94 ms per step).
#ifdef PWAF_PROFILING
    static const uint32_t forced = getenv("PWAF_ATTR_BLOCKS") ? (uint32_t)atoi(getenv("PWAF_ATTR_BLOCKS")) : 0u;
#else
    const uint32_t forced = 0;
#endif
    const uint32_t blocks = std::min<uint32_t>((a.n_groups + 3) / 4, forced ? forced : 6 * std::max(1u, a.attr_blocks));
    const bool small = a.set_words <= 4 && a.cc_words <= 2 && a.iu_words[0] <= 1 && a.iu_words[1] <= 1 && a.acmp_words <= 1;
    if (a.ipres_packed && small) hipLaunchKernelGGL((attr_kernel<true, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else if (a.ipres_packed) hipLaunchKernelGGL((attr_kernel<true, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else if (small) hipLaunchKernelGGL((attr_kernel<false, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((attr_kernel<false, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// Workgroup shape of the verdict kernel: as many waves as LDS (160 KiB) holds column files for, next to one copy of the program
// tables; if the tables do not leave room for at least 4 column files they stay in global memory (LT = false).
static constexpr uint32_t kLdsPerGroup = 160u * 1024u;
VerdictShape verdict_shape(uint32_t n_cols, uint32_t n_rules, uint32_t n_trig, uint32_t n_lits, bool force_global, int mode, uint32_t n_passes) {
    VerdictShape s{};
    const uint32_t t_lds = verdict_tables(n_cols, n_rules, n_trig, n_lits, true).end, t_glb = verdict_tables(n_cols, n_rules, n_trig, n_lits, false).end;
    if (mode >= 3) {
        // ENTRY LIST (mode 3; 4 = the test hook with 8 entry slots: every group takes the spill path): as many 12-wave workgroups per CU as
        // LDS holds, and the largest entry capacity that does not cost a wave (a group of the 1k-rule set appends 50-100 entries)
        s.sparse = 2;
        const uint32_t cu_waves = n_passes <= 64 ? 24u : 16u;
        auto fit = [&](uint32_t cap, uint32_t &bw, uint32_t &bk, uint32_t &lt) {
            const uint32_t wave = verdict_wave_lds_el(n_cols, n_rules, cap);
            const bool fits = !force_global && n_trig < 65536 && t_lds + 8 * wave <= kLdsPerGroup;
            const uint32_t tables = fits ? t_lds : t_glb;
            bw = 0; bk = 1; lt = fits ? 1u : 0u;
            for (uint32_t k = 1; k <= 3; k++) {
                const uint32_t share = kLdsPerGroup / k;
                if (share <= tables + wave) continue;
                const uint32_t w = std::min(std::min(12u, cu_waves / k), (share - tables) / wave);
                if (w * k > bw * bk) { bw = w; bk = k; }
            }
        };
        uint32_t cap = 8, bw = 0, bk = 1, lt = 0;
        if (mode == 4) fit(cap, bw, bk, lt);
        else {
            uint32_t w0, k0, l0;
            fit(64, w0, k0, l0);  // the occupancy the smallest capacity reaches ...
            for (uint32_t c : {254u, 224u, 192u, 160u, 128u, 96u, 64u}) {  // ... kept by the largest capacity that still reaches it
                fit(c, bw, bk, lt);
                cap = c;
                if (bw * bk >= w0 * k0) break;
            }
        }
        s.v_cap = cap;
        s.waves = bw;
        s.per_cu = bk;
        s.lds_tables = lt;
        s.lds_bytes = (lt ? t_lds : t_glb) + bw * verdict_wave_lds_el(n_cols, n_rules, cap);
        return s;
    }
    if (mode != 0) {
        // SPARSE column file (mode 1; 2 = the test hook with 8 value slots): as many workgroups of 12 waves per CU as LDS holds (two for
        // the 1k-rule set: 6 waves per SIMD), program tables in LDS when one such workgroup still fits beside them
        s.sparse = 1;
        s.v_cap = verdict_v_cap(n_cols, mode == 2);
        const uint32_t wave = verdict_wave_lds_sp(n_cols, n_rules, s.v_cap);
        const bool fits = !force_global && n_trig < 65536 && t_lds + 8 * wave <= kLdsPerGroup;
        s.lds_tables = fits ? 1 : 0;
        const uint32_t tables = fits ? t_lds : t_glb;
        // (registers: the variant for up to 64 passes is compiled for 6 waves per SIMD = 24 per CU, the general one takes 124 -> 16 per CU)
        const uint32_t cu_waves = n_passes <= 64 ? 24u : 16u;
        uint32_t best_w = 0, best_k = 1;
        for (uint32_t k = 1; k <= 3; k++) {  // k workgroups per CU
            const uint32_t share = kLdsPerGroup / k;
            if (share <= tables + wave) continue;
            const uint32_t w = std::min(std::min(12u, cu_waves / k), (share - tables) / wave);
            if (w * k > best_w * best_k) { best_w = w; best_k = k; }
        }
        s.waves = best_w;
        s.per_cu = best_k;
        s.lds_bytes = tables + s.waves * wave;
        return s;
    }
    const uint32_t wave = verdict_wave_lds(n_cols, n_rules);
    const bool fits = !force_global && n_trig < 65536 && t_lds + 4 * wave <= kLdsPerGroup;
    s.lds_tables = fits ? 1 : 0;
    const uint32_t tables = fits ? t_lds : t_glb;
    uint32_t w = tables + wave <= kLdsPerGroup ? (kLdsPerGroup - tables) / wave : 0;
    if (!fits) w = std::min(w, 4u);  // small workgroups: several share a CU
    s.waves = std::min(w, 16u);
    s.per_cu = 1;
    s.lds_bytes = tables + s.waves * wave;
    return s;
}

uint32_t verdict_blocks_sp(const VerdictShape &sh, uint32_t n_cus) { return std::max(1u, n_cus) * sh.per_cu; }  // the sparse kernel's persistent grid (sizes the spill array)

int launch_verdict(const VerdictArgs &a, void *stream) {
    VerdictShape sh = verdict_shape(a.n_cols, a.n_rules, a.n_trig, a.n_lits, a.force_global_tables != 0, (int)a.sparse_mode, a.n_passes);
    if (sh.waves == 0 || a.n_cols >= 65536u) return (int)hipErrorInvalidValue;  // (engine_create refuses such programs)
    if (sh.sparse == 1 && (a.v_cap != sh.v_cap || (sh.v_cap < a.n_cols && a.spill == nullptr))) return (int)hipErrorInvalidValue;
    if (sh.sparse == 2 && (a.v_cap != sh.v_cap || a.spill == nullptr)) return (int)hipErrorInvalidValue;
#ifdef PWAF_PROFILING
    static const uint32_t cap_waves = getenv("PWAF_VERDICT_WAVES") ? (uint32_t)atoi(getenv("PWAF_VERDICT_WAVES")) : 0u;  // timing experiment: fewer waves per CU (same results)
    if (cap_waves && cap_waves < sh.waves) sh.waves = cap_waves;
#endif
    constexpr int kBRmax = (kMaxPasses + 1 + 63) / 64;
    int variant = a.n_passes <= 64 ? 1 : 0;
#ifdef PWAF_PROFILING
    static const int forced_variant = getenv("PWAF_VERDICT_VARIANT") ? atoi(getenv("PWAF_VERDICT_VARIANT")) : -1;  // a MORE general variant may be forced (same results)
    if (forced_variant >= 0 && forced_variant < variant) variant = forced_variant;
#endif
    const void *fns[2][2][2] = {{{reinterpret_cast<const void *>(verdict_kernel<false, kBRmax, false>), reinterpret_cast<const void *>(verdict_kernel<false, 1, false>)},
                                 {reinterpret_cast<const void *>(verdict_kernel<true, kBRmax, false>), reinterpret_cast<const void *>(verdict_kernel<true, 1, false>)}},
                                {{reinterpret_cast<const void *>(verdict_kernel<false, kBRmax, true>), reinterpret_cast<const void *>(verdict_kernel<false, 1, true>)},
                                 {reinterpret_cast<const void *>(verdict_kernel<true, kBRmax, true>), reinterpret_cast<const void *>(verdict_kernel<true, 1, true>)}}};
    const void *fns2[2][2] = {{reinterpret_cast<const void *>(verdict2_kernel<false, kBRmax>), reinterpret_cast<const void *>(verdict2_kernel<false, 1>)},
                              {reinterpret_cast<const void *>(verdict2_kernel<true, kBRmax>), reinterpret_cast<const void *>(verdict2_kernel<true, 1>)}};
    const void *fn = sh.sparse == 2 ? fns2[sh.lds_tables ? 1 : 0][variant] : fns[sh.sparse ? 1 : 0][sh.lds_tables ? 1 : 0][variant];
    uint32_t blocks = (a.n_groups + sh.waves - 1) / sh.waves;
#ifdef PWAF_PROFILING
    static const uint32_t forced_cap = getenv("PWAF_VERDICT_BLOCKS") ? (uint32_t)atoi(getenv("PWAF_VERDICT_BLOCKS")) : 0u;
#else
    const uint32_t forced_cap = 0;
#endif
    // with LDS tables a workgroup fills a CU: one persistent workgroup per CU (measured: 0.331 ms vs 0.346 at 4 per CU — every
    // workgroup stages 25 KiB of tables and clears its column files once); small workgroups: a few rounds per CU
    const uint32_t cap = (forced_cap && !sh.sparse) ? forced_cap : sh.sparse ? verdict_blocks_sp(sh, a.attr_blocks) : sh.lds_tables ? std::max(1u, a.attr_blocks) : 2048u;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) return 0;
    void *args[] = {const_cast<VerdictArgs *>(&a)};
    hipError_t e = hipLaunchKernel(fn, dim3(blocks), dim3(sh.waves * 64), args, sh.lds_bytes, (hipStream_t)stream);
    return (int)(e != hipSuccess ? e : hipGetLastError());
}

// hipFuncSetAttribute applies to the CURRENT device: every device an engine is created on needs its own call (a per-thread
// or per-process "already configured" flag left the second device of a multi-GPU host at the 64 KiB default).
int configure_kernels(int device) {
    static std::mutex mu;
    static std::set<int> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(device)) return 0;
    const void *fns[] = {reinterpret_cast<const void *>(scan_kernel<1, false>), reinterpret_cast<const void *>(scan_kernel<2, false>),
                         reinterpret_cast<const void *>(scan_kernel<4, false>), reinterpret_cast<const void *>(scan_kernel<1, true>),
                         reinterpret_cast<const void *>(scan_kernel<2, true>), reinterpret_cast<const void *>(scan_kernel<4, true>),
                         reinterpret_cast<const void *>(verdict_kernel<true, (kMaxPasses + 1 + 63) / 64, false>), reinterpret_cast<const void *>(verdict_kernel<false, (kMaxPasses + 1 + 63) / 64, false>),
                         reinterpret_cast<const void *>(verdict_kernel<true, 1, false>), reinterpret_cast<const void *>(verdict_kernel<false, 1, false>),
                         reinterpret_cast<const void *>(verdict_kernel<true, (kMaxPasses + 1 + 63) / 64, true>), reinterpret_cast<const void *>(verdict_kernel<false, (kMaxPasses + 1 + 63) / 64, true>),
                         reinterpret_cast<const void *>(verdict_kernel<true, 1, true>), reinterpret_cast<const void *>(verdict_kernel<false, 1, true>),
                         reinterpret_cast<const void *>(verdict2_kernel<true, (kMaxPasses + 1 + 63) / 64>), reinterpret_cast<const void *>(verdict2_kernel<false, (kMaxPasses + 1 + 63) / 64>),
                         reinterpret_cast<const void *>(verdict2_kernel<true, 1>), reinterpret_cast<const void *>(verdict2_kernel<false, 1>),
                         reinterpret_cast<const void *>(filter_kernel<true>), reinterpret_cast<const void *>(filter_kernel<false>),
                         lscan_fn(false), lscan_fn(true)};
    for (const void *fn : fns) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerGroup);
        if (e != hipSuccess) return (int)e;
    }
    done.insert(device);
    return 0;
}

}  // namespace pwaf
