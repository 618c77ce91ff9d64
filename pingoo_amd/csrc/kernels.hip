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

static constexpr int kScanThreads = 512;
static constexpr int kScanWaves = kScanThreads / 64;
static constexpr int kVerdictThreads = 256;
static constexpr int kVerdictWaves = kVerdictThreads / 64;
static constexpr uint32_t kNone = 0xFFFFFFFFu;

uint32_t scan_lds_bytes(uint32_t n_hot, uint32_t stride) { return ((n_hot * stride * 2 + 15) & ~15u) + 256; }
uint32_t verdict_lds_bytes(uint32_t n_cols) { return kVerdictWaves * n_cols * 8; }

// -------------------------------------------------------------------------------------------------
// scan
// -------------------------------------------------------------------------------------------------
struct Hits {
    uint32_t a0, a1;  // local atom + 1, 0 = empty
    uint32_t ovf;     // head of the overflow chain, kNone = not overflowed
};

__device__ __noinline__ void pool_push(const ScanArgs &a, uint32_t atom, Hits &h) {
    const uint32_t idx = atomicAdd(a.pool_count, 1u);
    if (idx >= a.pool_cap) {
        atomicOr(a.status, 1u);
        return;
    }
    a.pool[idx].atom = atom;
    a.pool[idx].next = h.ovf;
    h.ovf = idx;
}

__device__ __noinline__ void record_atom(const ScanArgs &a, uint32_t atom, Hits &h) {
    if (h.ovf == kNone) {
        if (h.a0 == atom + 1 || h.a1 == atom + 1) return;
        if (h.a0 == 0) { h.a0 = atom + 1; return; }
        if (h.a1 == 0) { h.a1 = atom + 1; return; }
        pool_push(a, h.a0 - 1, h);
        pool_push(a, h.a1 - 1, h);
        pool_push(a, atom, h);
        return;
    }
    // overflowed: the chain holds every atom of this request; de-duplicate against it (this lane is its only writer)
    for (uint32_t i = h.ovf; i != kNone;) {
        const uint32_t at = __hip_atomic_load(&a.pool[i].atom, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (at == atom) return;
        i = __hip_atomic_load(&a.pool[i].next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    pool_push(a, atom, h);
}

__device__ __forceinline__ void emit_list(const ScanArgs &a, uint32_t id, Hits &h) {
    const uint32_t b = a.list_off[id], e = a.list_off[id + 1];
    for (uint32_t k = b; k < e; k++) record_atom(a, a.list[k], h);
}

__global__ __launch_bounds__(kScanThreads) void scan_kernel(ScanArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t hot_bytes = (a.n_hot * a.stride * 2 + 15) & ~15u;
    const uint16_t *ltab = reinterpret_cast<const uint16_t *>(lds);
    const uint8_t *cls = lds + hot_bytes;
    const uint16_t *gtab = a.tab;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    // stage the hot rows and the byte-class map into LDS (coalesced 16 B per lane; the global table is padded)
    for (uint32_t i = tid * 16; i < hot_bytes; i += kScanThreads * 16)
        *reinterpret_cast<uint4 *>(lds + i) = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(gtab) + i);
    if (tid < 64) reinterpret_cast<uint32_t *>(lds + hot_bytes)[tid] = reinterpret_cast<const uint32_t *>(a.classmap)[tid];
    __syncthreads();

    const uint32_t stride = a.stride, ncls = a.n_classes, n_hot = a.n_hot;
    // this wave's slab of requests: contiguous, 64-aligned so offset blocks are whole
    const uint32_t total_waves = gridDim.x * kScanWaves;
    const uint32_t per_wave = (((a.n + total_waves - 1) / total_waves) + 63) & ~63u;
    const uint32_t gw = blockIdx.x * kScanWaves + wave;
    const uint32_t w0 = min(a.n, gw * per_wave), w1 = min(a.n, w0 + per_wave);
    if (w0 >= w1) return;

    const unsigned long long lt_mask = (1ull << lane) - 1;
    const uint32_t start_emit = gtab[ncls + 1];  // emit-list id + 1 of state 0 (wave-uniform)

    uint32_t next = w0, blk = w0;
    auto load_off = [&](uint32_t base, uint32_t &lo, uint32_t &hi) {
        const uint32_t i = min(base + lane, a.n - 1);  // base + lane < n + 63; clamp keeps the load in bounds
        lo = a.off[i];
        hi = a.off[i + 1];
    };
    uint32_t o_lo, o_hi, n_lo = 0, n_hi = 0;
    load_off(blk, o_lo, o_hi);
    if (blk + 64 < w1) load_off(blk + 64, n_lo, n_hi);

    uint32_t r = kNone, p = 0, end = 0, st = 0;
    Hits h{0, 0, kNone};

    for (;;) {
        // ---- refill idle lanes from the slab (ballot + prefix popcount) ----
        const unsigned long long idle = __ballot(r == kNone);
        if (idle != 0 && next < w1) {
            const uint32_t avail = min(w1 - next, blk + 64 - next);
            const uint32_t rank = (uint32_t)__builtin_popcountll(idle & lt_mask);
            const bool take = r == kNone && rank < avail;
            const uint32_t j = take ? next + rank - blk : 0;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)o_lo);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)o_hi);
            if (take) {
                r = next + rank;
                p = lo;
                end = hi;
                st = 0;
                h = Hits{0, 0, kNone};
                if (start_emit) emit_list(a, start_emit - 1, h);
            }
            next += min((uint32_t)__builtin_popcountll(idle), avail);
            if (next == blk + 64 && next < w1) {
                blk += 64;
                o_lo = n_lo;
                o_hi = n_hi;
                if (blk + 64 < w1) load_off(blk + 64, n_lo, n_hi);
            }
        }
        if (__ballot(r != kNone) == 0) break;

        // ---- 16 bytes of every active lane's field ----
        const bool act = r != kNone && p < end;
        const uint32_t cnt = act ? min(16u, end - p) : 0u;
        uint32_t w[4] = {0, 0, 0, 0};
        if (act) __builtin_memcpy(w, a.data + p, 16);  // unaligned 16-byte load; arenas carry PWAF_ARENA_PAD slack
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t byte = (w[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            const bool use = (uint32_t)k < cnt;
            const uint32_t idx = st * stride + cls[byte];
            const bool hot = st < n_hot;
            uint32_t e = ltab[hot ? idx : 0u];
            if (use && !hot) e = gtab[idx];  // deep state: row comes from L2
            if (use) {
                st = e & 0x7FFFu;
                if (e & 0x8000u) {
                    const uint32_t id = st < n_hot ? ltab[st * stride + ncls + 1] : gtab[st * stride + ncls + 1];
                    emit_list(a, id - 1, h);
                }
            }
        }
        p += cnt;

        // ---- finished requests: end-of-field matches, then the hit record ----
        if (r != kNone && p >= end) {
            const uint32_t e = st < n_hot ? ltab[st * stride + ncls] : gtab[st * stride + ncls];
            if (e) emit_list(a, e - 1, h);
            a.rec[r] = h.ovf != kNone ? (REC_OVERFLOW | h.ovf) : (h.a0 | (h.a1 << 15));
            r = kNone;
        }
    }
}

int launch_scan(const ScanArgs &a, void *stream) {
    uint32_t lds = scan_lds_bytes(a.n_hot, a.stride);
    static thread_local uint32_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        configured = lds;
    }
    if (a.n == 0) return 0;
    // enough waves to fill 256 CUs several times over, but each with a few hundred requests so that work-pulling
    // has something to balance (a slab is >= 64 requests)
    uint32_t waves = (a.n + 255) / 256;
    uint32_t blocks = (waves + kScanWaves - 1) / kScanWaves;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(scan_kernel, dim3(blocks), dim3(kScanThreads), lds, (hipStream_t)stream, a);
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

__device__ __forceinline__ bool cmp_i64(long long v, uint32_t op, long long c) {
    return op == OP_EQ ? v == c : op == OP_LT ? v < c : v <= c;  // the compiler only emits EQ / LT / LE
}

__global__ __launch_bounds__(kVerdictThreads) void verdict_kernel(VerdictArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long *col = reinterpret_cast<unsigned long long *>(lds) + (size_t)wave * a.n_cols;
    const unsigned long long mybit = 1ull << lane;
    unsigned long long cnt_block = 0, cnt_captcha = 0, cnt_bypass = 0, cnt_allow = 0;  // wave-uniform tallies

    for (uint32_t g = blockIdx.x * kVerdictWaves + wave; g < a.n_groups; g += gridDim.x * kVerdictWaves) {
        const uint32_t i = g * 64 + lane;
        const bool valid = i < a.n;
        const unsigned long long valid_mask = __ballot(valid);

        // 1. clear the column file; column 0 is the constant TRUE
        for (uint32_t k = lane; k < a.n_cols; k += 64) col[k] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) col[0] = ~0ull;

        // 2. scan results: each lane ORs its request's bit into the columns its hit records name
        if (valid) {
            for (uint32_t ps = 0; ps < a.n_passes; ps++) {
                const uint32_t rv = a.rec[(size_t)ps * a.n + i];
                if (rv == 0) continue;
                const uint32_t base = a.pass_base[ps];
                if (rv & REC_OVERFLOW) {
                    for (uint32_t k = rv & ~REC_OVERFLOW; k != kNone;) {
                        const PoolEntry pe = a.pool[k];
                        atomicOr(&col[base + pe.atom], mybit);
                        k = pe.next;
                    }
                } else {
                    const uint32_t x0 = rv & 0x7FFFu, x1 = (rv >> 15) & 0x7FFFu;
                    if (x0) atomicOr(&col[base + x0 - 1], mybit);
                    if (x1) atomicOr(&col[base + x1 - 1], mybit);
                }
            }
        }

        // 3. this lane's request: lengths and numeric columns
        uint32_t len[PWAF_N_FIELDS] = {0, 0, 0, 0, 0};
        uint32_t ipw[4] = {0, 0, 0, 0};
        bool v6 = false;
        uint32_t port = 0, flags = 0, asn = 0, country = (uint32_t)'X' | ((uint32_t)'X' << 8);
        uint32_t set_id = 0;
        if (valid) {
#pragma unroll
            for (int f = 0; f < PWAF_N_FIELDS; f++) len[f] = a.off[f][i + 1] - a.off[f][i];
            const uint4 raw = *reinterpret_cast<const uint4 *>(a.ip + (size_t)i * 16);
            ipw[0] = raw.x; ipw[1] = raw.y; ipw[2] = raw.z; ipw[3] = raw.w;
            v6 = a.ip_is_v6[i] != 0;
            port = a.port[i];
            flags = a.flags[i];
            if (a.asn != nullptr) {
                asn = a.asn[i];
                country = a.country[i];
            } else if (a.has_geo) {
                // GeoipDB::lookup (pingoo/geoip.rs:73-91): loopback / multicast are "not found"
                bool skip;
                if (!v6) {
                    const uint32_t b0 = ipw[0] & 0xFFu;
                    skip = b0 == 127u || (b0 & 0xF0u) == 0xE0u;
                } else {
                    const bool loopback = ipw[0] == 0 && ipw[1] == 0 && ipw[2] == 0 && ipw[3] == 0x01000000u;
                    skip = loopback || (ipw[0] & 0xFFu) == 0xFFu;
                }
                if (!skip) {
                    const uint32_t rec = trie_lookup(a.geo_root4, a.geo_root6, a.geo_nodes, ipw, v6);
                    const GeoRec r = a.geo_recs[rec];
                    asn = r.asn;
                    country = r.country;
                }
            }
            if (a.n_ip_lists) set_id = trie_lookup(a.ip_root4, a.ip_root6, a.ip_nodes, ipw, v6);
        }
        uint32_t cidx;
        {
            const uint32_t c0 = (country & 0xFFu) - 'A', c1 = (country >> 8) - 'A';
            cidx = (c0 < 26u && c1 < 26u) ? c0 * 26u + c1 : 23u * 26u + 23u;  // invalid input is treated as "XX"
        }
        const unsigned long long verified_mask = __ballot(valid && (flags & PWAF_FLAG_CAPTCHA_VERIFIED));

        for (uint32_t k = 0; k < a.n_num_atoms; k++) {
            const NumAtomDev d = a.num_atoms[k];  // wave-uniform
            bool t = false;
            switch (d.kind) {
                case ATOM_LEN: t = cmp_i64((long long)len[d.var < PWAF_N_FIELDS ? d.var : 0], d.op, d.c); break;
                case ATOM_INT: t = cmp_i64(d.var == VAR_PORT ? (long long)port : (long long)asn, d.op, d.c); break;
                case ATOM_INTSET: {
                    const long long v = d.var == VAR_PORT ? (long long)port : (long long)asn;
                    uint32_t lo = d.ref, hi = d.ref2;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        const long long m = a.int_pool[mid];
                        if (m < v) lo = mid + 1;
                        else hi = mid;
                    }
                    t = lo < d.ref2 && a.int_pool[lo] == v;
                    break;
                }
                case ATOM_IPSET: t = (a.set_masks[(size_t)set_id * a.set_words + (d.ref >> 5)] >> (d.ref & 31)) & 1u; break;
                case ATOM_COUNTRY: t = (a.country_luts[(size_t)d.ref * 22 + (cidx >> 5)] >> (cidx & 31)) & 1u; break;
                default: break;
            }
            const unsigned long long m = __ballot(t && valid);
            if (lane == 0) col[d.col] = m;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        // 4. rules: one lane per rule, 64 requests per ALU op; first match (lowest rule index) wins
        unsigned long long pending = valid_mask;
        uint32_t my_action = PWAF_ACTION_ALLOW, my_rule = PWAF_RULE_NONE;
        for (uint32_t base = 0; base < a.n_rules && pending != 0; base += 64) {
            const uint32_t r = base + lane;
            unsigned long long fire = 0;
            uint32_t eff_u = 0, eff_v = 0, pub = 0;
            if (r < a.n_rules) {
                const DevRule dr = a.rules[r];
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

        // 5. outputs
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
                if (hit & mybit) a.match_idx[basei + (uint32_t)__builtin_popcountll(hit & (mybit - 1))] = i;
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

int launch_verdict(const VerdictArgs &a, void *stream) {
    uint32_t lds = verdict_lds_bytes(a.n_cols);
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
