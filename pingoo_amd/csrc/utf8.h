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
// field from the lead byte on. Returns the class.
PWAF_U8_HD uint32_t utf8_class(const uint8_t *umap, const uint32_t ill_class, const uint32_t b0, const uint32_t next, const uint32_t avail) {
    const uint32_t len = b0 >= 0xF0u ? 4u : b0 >= 0xE0u ? 3u : 2u;
    if (b0 < 0xC2u || b0 > 0xF4u || avail < len) return ill_class;
    const uint32_t c1 = next & 0xFFu, c2 = (next >> 8) & 0xFFu, c3 = (next >> 16) & 0xFFu;
    if ((c1 & 0xC0u) != 0x80u || (len > 2u && (c2 & 0xC0u) != 0x80u) || (len > 3u && (c3 & 0xC0u) != 0x80u)) return ill_class;
    uint32_t cp;
    if (len == 2u) cp = ((b0 & 0x1Fu) << 6) | (c1 & 0x3Fu);
    else if (len == 3u) cp = ((b0 & 0x0Fu) << 12) | ((c1 & 0x3Fu) << 6) | (c2 & 0x3Fu);
    else cp = ((b0 & 0x07u) << 18) | ((c1 & 0x3Fu) << 12) | ((c2 & 0x3Fu) << 6) | (c3 & 0x3Fu);
    if ((len == 3u && (cp < 0x800u || (cp >= 0xD800u && cp <= 0xDFFFu))) || (len == 4u && (cp < 0x10000u || cp > 0x10FFFFu))) return ill_class;
    const uint32_t block = reinterpret_cast<const uint16_t *>(umap)[cp >> 7];
    return umap[kUmapStage2 + block * 128u + (cp & 127u)];
}

}  // namespace pwaf
