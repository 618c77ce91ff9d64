// confirm.h — the confirm tier of a filtered pass (program.h: ConfirmTable): what a flagged chunk really holds.
//
// ONE implementation for the device kernel (kernels.hip: confirm_kernel), for the host model pwaf_engine_tune uses to rank the DFA
// rows that confirmed candidates visit, and for the CPU test hook that fuzzes the compiled tables against the oracle
// (pwaf_program_confirm_field): the code the GPU runs is the code the CPU suite checks. The reference evaluates every string
// predicate on every request (pingoo/rules.rs:37-51); here a literal predicate is evaluated exactly where the bigram filter pointed.
#pragma once
#include <cstdint>
#include <cstring>

#include "program.h"

#if defined(__HIPCC__)
#define PWAF_HD __host__ __device__ __forceinline__
#else
#define PWAF_HD inline
#endif

namespace pwaf {

struct ConfirmView {  // plain pointers: host tables or device tables
    const uint32_t *head;
    const ConfirmEntry *entries;
    const uint8_t *bytes;
    const uint32_t *classes;
    uint32_t mul, stride;
    uint32_t init;  // the filter's state of a stream with no history (GroupFilter::init)
};

PWAF_HD uint32_t confirm_load32(const uint8_t *p) {  // request TEXT: an arena in global memory
#if defined(__HIP_DEVICE_COMPILE__)
    // (a GLOBAL load, at any byte address — gfx950 needs no alignment — instead of the FLAT load a generic pointer gets)
    return *reinterpret_cast<const __attribute__((address_space(1))) uint32_t __attribute__((aligned(1))) *>((uintptr_t)p);
#else
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
#endif
}
// 16 / 8 bytes of request text at once: a scattered load instruction costs the texture addresser a cycle per lane whatever its width
// (DESIGN.md 6.1 found the streaming kernel bound by exactly that), so the chunk's 28 bytes are three instructions, not seven.
struct ConfirmText4 { uint32_t x, y, z, w; };
PWAF_HD ConfirmText4 confirm_load128(const uint8_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    typedef u4 u4_u __attribute__((aligned(1)));
    const u4 v = *reinterpret_cast<const __attribute__((address_space(1))) u4_u *>((uintptr_t)p);
    return ConfirmText4{v.x, v.y, v.z, v.w};
#else
    ConfirmText4 v;
    memcpy(&v, p, 16);
    return v;
#endif
}
PWAF_HD void confirm_load64(const uint8_t *p, uint32_t &lo, uint32_t &hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    typedef u2 u2_u __attribute__((aligned(1)));
    const u2 v = *reinterpret_cast<const __attribute__((address_space(1))) u2_u *>((uintptr_t)p);
    lo = v.x;
    hi = v.y;
#else
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
#endif
}

// The tier's TABLES (4-byte aligned words). SPACE says where they live for the device: 1 = global memory, 3 = the kernel's LDS copy (the
// generic pointer's low half is the LDS address: a ds_read instead of the FLAT load a generic pointer gets — a FLAT load that hits LDS
// still takes the texture path's latency); the host ignores it.
template <int SPACE>
PWAF_HD uint32_t confirm_table32(const uint8_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (SPACE == 3) return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uint32_t)(uintptr_t)p);
    return *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>((uintptr_t)p);
#else
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
#endif
}

// One entry against the text: does the factor occur with its window's last bigram at arena position i, inside the field [fs, fe)?
// 0 = no, 1 | atom << 8 = yes and it decides that literal atom, 2 = yes and it is a factor of a non-literal atom (walk).
// The chain of dependent accesses is what a comparison costs, so it is kept short: the entry's three words together, then value, mask
// and text of up to 16 bytes of the factor together.
template <int SPACE>
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t
    confirm_entry(const ConfirmEntry *entries, const uint8_t *bytes, const uint32_t *classes, const uint32_t index, const uint8_t *data, const uint32_t fs, const uint32_t fe,
                  const uint32_t i) {
    const uint8_t *ep = reinterpret_cast<const uint8_t *>(entries) + (size_t)index * sizeof(ConfirmEntry);
    const uint32_t e_off = confirm_table32<SPACE>(ep), e_ld = confirm_table32<SPACE>(ep + 4), e_af = confirm_table32<SPACE>(ep + 8);
    const uint32_t len = e_ld & 0xFFFFu, d = e_ld >> 16, atom = e_af & 0xFFFFu, flags = (e_af >> 16) & 0xFFu, n_cls = e_af >> 24;
    if (i < fs + d) return 0;
    const uint32_t q = i - d;
    if (q + len > fe) return 0;
    if ((flags & kConfirmAtStart) && q != fs) return 0;
    if ((flags & kConfirmAtEnd) && q + len != fe) return 0;
    const uint32_t l4 = (len + 3u) & ~3u;
    const uint8_t *val = bytes + e_off, *msk = val + l4;
    // four words (16 bytes: most factors whole) per round of loads — text, value and mask of all four are in flight together, so a
    // mismatch in the factor's last bytes (a near miss) costs one trip to the text, not one per word. (Reads up to 3 bytes past the
    // factor: arenas carry PWAF_ARENA_PAD slack, the masks there are zero.)
    ConfirmText4 first{0u, 0u, 0u, 0u};  // the factor's first 16 bytes of text: class positions inside them need no load of their own
    for (uint32_t w = 0; w < l4; w += 16) {
        const ConfirmText4 t4 = confirm_load128(data + q + w);  // (may read up to 15 bytes past the factor: PWAF_ARENA_PAD)
        if (w == 0u) first = t4;
        const uint32_t tw[4] = {t4.x, t4.y, t4.z, t4.w};
        uint32_t diff = 0;
#pragma unroll
        for (uint32_t u = 0; u < 4; u++)
            if (w + 4u * u < l4) diff |= (tw[u] ^ confirm_table32<SPACE>(val + w + 4u * u)) & confirm_table32<SPACE>(msk + w + 4u * u);
        if (diff) return 0;
    }
    const uint8_t *cls = msk + l4;
    const uint8_t *cw = reinterpret_cast<const uint8_t *>(classes);
    auto text_byte = [&](const uint32_t at) -> uint32_t {
        if (at < 16u) {
            const uint32_t word = at < 4u ? first.x : at < 8u ? first.y : at < 12u ? first.z : first.w;
            return (word >> (8u * (at & 3u))) & 0xFFu;
        }
        return confirm_load32(data + q + at) & 0xFFu;
    };
    for (uint32_t k = 0; k < n_cls; k += 2) {
        const uint32_t pc = confirm_table32<SPACE>(cls + 2u * k);  // two {position, class id} pairs (the pool is padded to whole dwords)
        const uint32_t t = text_byte(pc & 0xFFu);
        if (!((confirm_table32<SPACE>(cw + 4u * (((pc >> 8) & 0xFFu) * 8u + (t >> 5))) >> (t & 31u)) & 1u)) return 0;
        if (k + 1u < n_cls) {
            const uint32_t t2 = text_byte((pc >> 16) & 0xFFu);
            if (!((confirm_table32<SPACE>(cw + 4u * ((pc >> 24) * 8u + (t2 >> 5))) >> (t2 & 31u)) & 1u)) return 0;
        }
    }
    return atom == kConfirmWalk ? 2u : (1u | (atom << 8));
}

// A flagged 16-byte arena chunk as the confirm tier sees it: the positions where a window of the pass's filter really COMPLETED inside
// the field [fs, fe) (bit k = chunk byte k) and, for the first four of them in ascending order, the filter bin of the bigram there
// (12 bits each, 16 apart: a dynamic shift of one 64-bit register — selecting the bytes again by position made the compiler park the
// chunk in scratch memory).
struct ConfirmChunk {
    uint32_t mask;
    uint64_t bins;
};
// The filter's own shift-or automaton over the chunk, warmed up with the sampled bigrams of the 8 bytes before it (the state it had on
// the device: four pushes determine it): one or two of the sixteen positions survive. Registers and tab_at(bin) — the pass's filter
// table, which the device keeps in LDS — only. A match of any pattern holds a factor whose window completes at one of these
// positions, inside the field (what lies before the factor cannot matter: a bucket only tests its last kmin bigrams).
// bytes [16 c - 8, 16 c + 20) of the arena: the chunk, the 8 bytes before it (zeros for the arena's first chunk) and the 4 behind it
struct ConfirmBytes {
    uint32_t w[7];
};
PWAF_HD ConfirmBytes confirm_chunk_bytes(const uint8_t *data, const uint32_t c, const uint32_t readable /* arena bytes + PWAF_ARENA_PAD */) {
    const uint32_t base = c * 16u;
    ConfirmBytes t;
    if (base >= 8u && base + 24u <= readable) {  // (chunks are 16-aligned: every chunk but the arena's first has 8 bytes in front of it) two 16-byte loads: [base - 8, base + 24)
        const ConfirmText4 lo = confirm_load128(data + base - 8u), hi = confirm_load128(data + base + 8u);
        t.w[0] = lo.x; t.w[1] = lo.y; t.w[2] = lo.z; t.w[3] = lo.w;
        t.w[4] = hi.x; t.w[5] = hi.y; t.w[6] = hi.z;
    } else {
        const ConfirmText4 t4 = confirm_load128(data + base);
        t.w[0] = t.w[1] = 0u;
        if (base >= 8u) confirm_load64(data + base - 8u, t.w[0], t.w[1]);
        t.w[2] = t4.x; t.w[3] = t4.y; t.w[4] = t4.z; t.w[5] = t4.w;
        t.w[6] = confirm_load32(data + base + 16u);
    }
    return t;
}
template <class TabAt>
PWAF_HD ConfirmChunk confirm_windows_of(const ConfirmView &cv, const ConfirmBytes &tb, const uint32_t fs, const uint32_t fe, const uint32_t c, TabAt &&tab_at) {
    const uint32_t base = c * 16u;
    const bool pre = base >= 8u;
    // seven words, every position at a fixed place in them (the 24 steps are unrolled: no register is indexed by the position and
    // nothing moves between steps)
    const uint32_t *w = tb.w;
    uint32_t st = cv.init, mask = 0, found = 0;
    uint64_t bins = 0;
#pragma unroll
    for (uint32_t t = 0; t < 24; t++) {
        if (cv.stride == 2u && (t & 1u)) continue;  // (bigrams are sampled at the even bytes of the arena; chunks start at even addresses)
        const uint32_t lo = w[t >> 2] >> (8u * (t & 3u)), b0 = lo & 0xFFu, b1 = (t & 3u) == 3u ? w[(t >> 2) + 1] & 0xFFu : (lo >> 8) & 0xFFu;
        const uint32_t bin = filter_bin((uint8_t)b0, (uint8_t)b1, cv.mul);
        if (t < 8u) {
            if (pre) st = (st << 8) | tab_at(bin);  // (the arena's first chunk: the stream starts in the init state)
            continue;
        }
        st = (st << 8) | tab_at(bin);
        const uint32_t i = base + t - 8u;
        if (((~st) & 0xFF000000u) != 0u && i >= fs && i < fe) {  // a window completed here, its last bigram starting inside the field (its second byte may be the byte behind the field: the window of a short factor that ends with the field — filter.cpp, Model::best_window)
            mask |= 1u << (t - 8u);
            if (found < 4u) bins |= (uint64_t)bin << (16u * found);
            found++;
        }
    }
    ConfirmChunk r;
    r.mask = fe < fs + 2u ? 0u : mask;
    r.bins = bins;
    return r;
}
template <class TabAt>
PWAF_HD ConfirmChunk confirm_windows(const ConfirmView &cv, const uint8_t *data, const uint32_t fs, const uint32_t fe, const uint32_t c, TabAt &&tab_at) {
    return confirm_windows_of(cv, confirm_chunk_bytes(data, c, c * 16u + 20u), fs, fe, c, tab_at);  // (the host's arenas: only what the narrow form reads)
}
// filter bin of the bigram of the chunk's idx-th completed window (ascending positions), which sits at arena position pos
PWAF_HD uint32_t confirm_bin_of(const ConfirmChunk &ch, const uint32_t idx, const uint8_t *data, const uint32_t pos, const uint32_t mul) {
    if (idx < 4u) return (uint32_t)(ch.bins >> (16u * idx)) & 0xFFFu;
    const uint32_t two = confirm_load32(data + pos);  // (more than four windows completed in one chunk: rare)
    return filter_bin((uint8_t)(two & 0xFFu), (uint8_t)((two >> 8) & 0xFFu), mul);
}

// The flagged chunk c against the field [fs, fe) of one request, start to end: hit(atom) for every literal atom confirmed at a position
// of the chunk; returns true when a factor of a non-literal atom was confirmed (the request must be walked). head_at(bin) reads the
// confirm table's head word. (The host's form — tune, the CPU test hook. The device kernel runs the same three steps — windows, head
// word, entry comparison — as a per-lane state machine, so that the 64 requests of a wave each take THEIR next comparison per
// iteration instead of the wave taking the product of the nested loops' longest trips: kernels.hip, confirm_kernel.)
template <class TabAt, class HeadAt, class Hit>
PWAF_HD bool confirm_chunk(const ConfirmView &cv, const uint8_t *data, const uint32_t fs, const uint32_t fe, const uint32_t c, TabAt &&tab_at, HeadAt &&head_at, Hit &&hit) {
    const ConfirmChunk ch = confirm_windows(cv, data, fs, fe, c, tab_at);
    uint32_t mask = ch.mask, idx = 0;
    bool walk = false;
    while (mask) {
        const uint32_t k = (uint32_t)__builtin_ctz(mask);
        mask &= mask - 1u;
        const uint32_t hd = head_at(confirm_bin_of(ch, idx++, data, c * 16u + k, cv.mul));
        if (hd == 0u) continue;
        const uint32_t first = hd & 0xFFFFFu, cnt = hd >> 20;
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t res = confirm_entry<1>(cv.entries, cv.bytes, cv.classes, first + j, data, fs, fe, c * 16u + k);
            if (res == 2u) walk = true;
            else if (res & 1u) hit(res >> 8);
        }
    }
    return walk;
}

}  // namespace pwaf
