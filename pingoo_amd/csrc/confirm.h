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
};

PWAF_HD uint32_t confirm_load32(const uint8_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const uint32_t __attribute__((aligned(1))) *>(p);  // (gfx950 global loads need no alignment)
#else
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
#endif
}

// One entry against the text: does the factor occur with its window's last bigram at arena position i, inside the field [fs, fe)?
PWAF_HD bool confirm_entry(const ConfirmView &cv, const ConfirmEntry &e, const uint8_t *data, const uint32_t fs, const uint32_t fe, const uint32_t i) {
    const uint32_t len = e.len, d = e.d;
    if (i < fs + d) return false;
    const uint32_t q = i - d;
    if (q + len > fe) return false;
    if ((e.flags & kConfirmAtStart) && q != fs) return false;
    if ((e.flags & kConfirmAtEnd) && q + len != fe) return false;
    const uint32_t l4 = (len + 3u) & ~3u;
    const uint8_t *val = cv.bytes + e.bytes_off, *msk = val + l4;
    for (uint32_t w = 0; w < l4; w += 4)  // (reads up to 3 bytes past the factor: arenas carry PWAF_ARENA_PAD slack, the masks there are zero)
        if ((confirm_load32(data + q + w) ^ confirm_load32(val + w)) & confirm_load32(msk + w)) return false;
    const uint8_t *cls = msk + l4;
    for (uint32_t k = 0; k < e.n_cls; k++) {
        const uint32_t t = data[q + cls[2 * k]];
        if (!((cv.classes[(uint32_t)cls[2 * k + 1] * 8u + (t >> 5)] >> (t & 31u)) & 1u)) return false;
    }
    return true;
}

// The flagged 16-byte arena chunk c against the field [fs, fe) of one request: hit(atom) for every literal atom confirmed at a
// position of the chunk; returns true when a factor of a non-literal atom was confirmed (the request must be walked).
// head_at(bin) reads the table's head word (the device keeps the heads in LDS).
template <class HeadAt, class Hit>
PWAF_HD bool confirm_chunk(const ConfirmView &cv, const uint8_t *data, const uint32_t fs, const uint32_t fe, const uint32_t c, HeadAt &&head_at, Hit &&hit) {
    if (fe < fs + 2u) return false;
    const uint32_t base = c * 16u;
    // the chunk's bytes and the byte after it (second half of its last bigram)
    uint32_t w[5];
#pragma unroll
    for (uint32_t k = 0; k < 5; k++) w[k] = confirm_load32(data + base + 4u * k);
    bool walk = false;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        const uint32_t i = base + k;
        if (i < fs || i + 1u >= fe) continue;        // both bytes of the bigram inside the field
        if (cv.stride == 2u && (i & 1u)) continue;  // (bigrams are sampled at the even bytes of the arena)
        const uint32_t b0 = (w[k >> 2] >> (8u * (k & 3u))) & 0xFFu, b1 = (w[(k + 1u) >> 2] >> (8u * ((k + 1u) & 3u))) & 0xFFu;
        const uint32_t hd = head_at(filter_bin((uint8_t)b0, (uint8_t)b1, cv.mul));
        if (hd == 0u) continue;
        const uint32_t first = hd & 0xFFFFFu, cnt = hd >> 20;
        for (uint32_t j = 0; j < cnt; j++) {
            const ConfirmEntry e = cv.entries[first + j];
            if (!confirm_entry(cv, e, data, fs, fe, i)) continue;
            if (e.atom == kConfirmWalk) walk = true;
            else hit((uint32_t)e.atom);
        }
    }
    return walk;
}

}  // namespace pwaf
