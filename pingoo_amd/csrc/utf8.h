// utf8.h — the symbol a walker reads at a LEAD byte of a table in scalar mode (dfa.cpp): the class of the scalar value the bytes encode.
//
// ONE implementation for every walker: the device kernels (kernels.hip: lscan_kernel, scan_kernel; residual.h), the host walks of
// pwaf_engine_tune and the CPU test hooks. The reference's strings are Rust str — always well-formed (pingoo/rules.rs:16-25) — and the
// regex crate matches their SCALAR VALUES (regex 1.12.2, Cargo.lock:1694-1700); an ill-formed byte (no Rust str holds one) is a symbol
// of its own that no class matches (ScalarMap::ill_class).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define PWAF_U8_HD __host__ __device__ __forceinline__
#else
#define PWAF_U8_HD inline
#endif

namespace pwaf {

static constexpr uint32_t kUmapStage2 = (0x110000u >> 7) * 2u;  // byte offset of stage 2 in the device image: [stage1 u16 x 8704][stage2 u8 ...]

// b0 = the lead byte (>= 0xC0), next = the three bytes behind it (little-endian in one word: byte k at bits 8k), avail = bytes of the
// field from the lead byte on. True when the bytes are ONE well-formed sequence (strict: no overlongs, no surrogates, <= U+10FFFF); cp = its scalar value.
PWAF_U8_HD bool utf8_decode(const uint32_t b0, const uint32_t next, const uint32_t avail, uint32_t &cp) {
    const uint32_t len = b0 >= 0xF0u ? 4u : b0 >= 0xE0u ? 3u : 2u;
    if (b0 < 0xC2u || b0 > 0xF4u || avail < len) return false;
    const uint32_t c1 = next & 0xFFu, c2 = (next >> 8) & 0xFFu, c3 = (next >> 16) & 0xFFu;
    if ((c1 & 0xC0u) != 0x80u || (len > 2u && (c2 & 0xC0u) != 0x80u) || (len > 3u && (c3 & 0xC0u) != 0x80u)) return false;
    if (len == 2u) cp = ((b0 & 0x1Fu) << 6) | (c1 & 0x3Fu);
    else if (len == 3u) cp = ((b0 & 0x0Fu) << 12) | ((c1 & 0x3Fu) << 6) | (c2 & 0x3Fu);
    else cp = ((b0 & 0x07u) << 18) | ((c1 & 0x3Fu) << 12) | ((c2 & 0x3Fu) << 6) | (c3 & 0x3Fu);
    return !((len == 3u && (cp < 0x800u || (cp >= 0xD800u && cp <= 0xDFFFu))) || (len == 4u && (cp < 0x10000u || cp > 0x10FFFFu)));
}
// The class a walker reads at a lead byte.
PWAF_U8_HD uint32_t utf8_class(const uint8_t *umap, const uint32_t ill_class, const uint32_t b0, const uint32_t next, const uint32_t avail) {
    uint32_t cp = 0;
    if (!utf8_decode(b0, next, avail, cp)) return ill_class;
    const uint32_t block = reinterpret_cast<const uint16_t *>(umap)[cp >> 7];
    return umap[kUmapStage2 + block * 128u + (cp & 127u)];
}
// A CONTINUATION byte (0x80-0xBF): is it part of a well-formed sequence of its field? Then the walker stays (the sequence's symbol was read
// at its lead byte); else it is an ill-formed unit like any other (round 6: it used to be skipped, so that `a\x80b` held "ab" for a
// scalar-mode table — the oracle's decode_units never did). prev = the three bytes BEFORE it (the byte d positions before at bits 8(d-1)),
// self_next = the byte itself and the three after it (little-endian), back = min(3, bytes of the field before it), fwd = bytes of the field
// from it on (>= 1). The first lead byte met going backwards decides (a sequence holds nothing but continuation bytes behind its lead).
PWAF_U8_HD bool utf8_cont_covered(const uint32_t prev, const uint32_t self_next, const uint32_t back, const uint32_t fwd) {
    for (uint32_t d = 1; d <= 3u; d++) {
        if (d > back) return false;
        const uint32_t lead = (prev >> (8u * (d - 1u))) & 0xFFu;
        if (lead < 0x80u) return false;
        if (lead < 0xC0u) continue;
        // the three bytes behind the lead: the d - 1 bytes between it and this one, this one, what follows
        const uint32_t next = d == 1u ? self_next : d == 2u ? ((prev & 0xFFu) | (self_next << 8)) : (((prev >> 8) & 0xFFu) | ((prev & 0xFFu) << 8) | (self_next << 16));
        const uint32_t len = lead >= 0xF0u ? 4u : lead >= 0xE0u ? 3u : 2u;
        uint32_t cp = 0;
        return len > d && utf8_decode(lead, next, d + fwd, cp);
    }
    return false;
}

}  // namespace pwaf
