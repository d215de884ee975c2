#include <hip/hip_runtime.h>
#include <cstdio>
#define GG_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xf, 0xf, false))
__global__ void k(float* out) {
    int lane = threadIdx.x;
    float a = (float)lane, b = 100.f + lane;
    auto pr = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    out[lane] = __builtin_bit_cast(float, pr[0]);
    out[64 + lane] = __builtin_bit_cast(float, pr[1]);
    float s = (float)lane, s2 = s;
    asm volatile("" : "+v"(s2));
    auto p2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s), __builtin_bit_cast(unsigned, s2), false, false);
    out[128 + lane] = __builtin_bit_cast(float, p2[0]);
    out[192 + lane] = __builtin_bit_cast(float, p2[1]);
    {
        float x = (float)lane, y = 100.f + lane;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
        out[512 + lane] = x; out[576 + lane] = y;
        float u = (float)lane, w = (float)lane;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(w));
        out[640 + lane] = u + w;
    }
    out[256 + lane] = GG_DPP((float)lane, 0x128);
    out[320 + lane] = GG_DPP((float)lane, 0x141);
    out[384 + lane] = GG_DPP((float)lane, 0xB1);
    out[448 + lane] = GG_DPP((float)lane, 0x4E);
}
int main() {
    float* d; hipMalloc(&d, 704 * 4);
    k<<<1, 64>>>(d);
    float h[704]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[11] = {"p16 vdst'", "p16 src0'", "p32 vdst'", "p32 src0'", "row_ror8", "half_mirror", "qp1032", "qp2301", "asm16 x", "asm16 y", "asm32 sum"};
    for (int r = 0; r < 11; ++r) { printf("%-12s", nm[r]); for (int i = 0; i < 64; ++i) printf(" %g", h[r * 64 + i]); printf("\n"); }
    return 0;
}
