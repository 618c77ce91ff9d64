// Timing/semantics probe (not part of the product): which lane a gfx9 wave-shift DPP control reads from, and SDWA word selection.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out, unsigned carry) {
    unsigned lane = threadIdx.x;
    unsigned v = 100 + lane;
    unsigned m = 0x3FFCu, a1;
    unsigned packed = (lane * 4u + 3u) << 16 | 0xFFFFu;
    asm volatile("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(a1) : "v"(packed), "v"(m));
    unsigned shr = __builtin_amdgcn_update_dpp(carry, v, 0x138, 0xF, 0xF, false);  // wave_shr:1
    unsigned rol = __builtin_amdgcn_update_dpp(carry, v, 0x134, 0xF, 0xF, false);  // wave_rol:1
    unsigned shl = __builtin_amdgcn_update_dpp(carry, v, 0x130, 0xF, 0xF, false);  // wave_shl:1
    unsigned ror = __builtin_amdgcn_update_dpp(carry, v, 0x13C, 0xF, 0xF, false);  // wave_ror:1
    out[lane] = a1; out[64 + lane] = shr; out[128 + lane] = rol; out[192 + lane] = shl; out[256 + lane] = ror;
}
int main() {
    unsigned *d, h[320];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    probe<<<1, 64>>>(d, 7u);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const char *names[5] = {"sdwa_and_word1", "wave_shr1", "wave_rol1", "wave_shl1", "wave_ror1"};
    for (int k = 0; k < 5; k++) {
        printf("%s:", names[k]);
        for (int i = 0; i < 64; i++) if (i < 4 || i > 60 || (i >= 14 && i <= 17) || (i >= 30 && i <= 33)) printf(" [%d]=%u", i, h[64 * k + i]);
        printf("\n");
    }
    return 0;
}
