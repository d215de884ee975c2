// Standalone timing harness for the weight-gradient kernel (tools only; includes the kernel source directly).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gemm_gan_amd/csrc [-DGG_WG_WT=2 -DGG_WG_DEPTH=2] tools/wgrad_probe.hip -o tools/wgrad_probe
//   ./tools/wgrad_probe <case> [M]     cases: qkv ffn1 outproj ffn2
#include "../gemm_gan_amd/csrc/wgrad.hip"
#include <stdio.h>
#include <string.h>
namespace gg { void set_error(const std::string& s) { fprintf(stderr, "gg error: %s\n", s.c_str()); } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#ifdef GG_WG_STORE
// second pass of the store + reduce probe: out[i] += sum over slices of part[s][i]  (float4 per thread)
__global__ void reduce_slices(const float* __restrict__ part, float* __restrict__ out, long n4, int slices) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = reinterpret_cast<const f32x4*>(out)[i];
    for (int s = 0; s < slices; ++s) a += reinterpret_cast<const f32x4*>(part)[(long)s * n4 + i];
    reinterpret_cast<f32x4*>(out)[i] = a;
}
#endif
int main(int argc, char** argv) {
    const char* cs = argc > 1 ? argv[1] : "qkv";
    const long M = argc > 2 ? atol(argv[2]) : 131584;
    void *X, *Y; float* dW;
    CK(hipMalloc(&X, M * 512 * 4)); CK(hipMalloc(&Y, M * 768 * 4)); CK(hipMalloc(&dW, 256L * 1024 * 1024));
    CK(hipMemset(X, 0, M * 512 * 4)); CK(hipMemset(Y, 0, M * 768 * 4)); CK(hipMemset(dW, 0, 256L * 1024 * 1024));
    int N, K, yb = 1, xb; double bytes;
    long Mx = M;
    gg::WgradFilm film; gg::WgradFilmGrad fg; const gg::WgradFilm* pf = nullptr; const gg::WgradFilmGrad* pg = nullptr;
    float *gb, *dgb;
    CK(hipMalloc(&gb, 1024 * 2048 * 4)); CK(hipMalloc(&dgb, 1024 * 2048 * 4)); CK(hipMemset(gb, 0, 1024 * 2048 * 4)); CK(hipMemset(dgb, 0, 1024 * 2048 * 4));
    if (!strcmp(cs, "qkv")) { N = 768; K = 256; xb = 0; }
    else if (!strcmp(cs, "ffn1")) { N = 512; K = 256; xb = 0; }
    else if (!strcmp(cs, "outproj")) { N = 256; K = 256; xb = 1; }
    else if (!strcmp(cs, "ffn2")) { N = 256; K = 512; xb = 1; }
    else if (!strcmp(cs, "qkvb")) { N = 768; K = 256; xb = 1; }
    else if (!strcmp(cs, "film")) { N = 256; K = 1024; xb = 0; yb = 0; Mx = 65536; film.g = gb; film.b = gb + 1024; film.ld = 2048; film.group = 256; pf = &film; }
    else if (!strcmp(cs, "fgrad")) { N = 256; K = 1024; xb = 0; yb = 0; Mx = 65536; fg.W = (const float*)Y; fg.ldw = 1024; fg.dgamma = dgb; fg.dbeta = dgb + 1024; fg.ld = 2048; fg.tokens = 256; pg = &fg; }
    else { printf("unknown case\n"); return 1; }
    const long M2 = Mx; bytes = (double)M2 * (N * (yb ? 2.0 : 4.0) + K * (xb ? 2.0 : 4.0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (gg::wgrad(Y, N, yb, X, K, xb, pg ? nullptr : dW, K, M2, N, K, 0, pf, pg, nullptr, 0)) return 1;
    CK(hipDeviceSynchronize());
#ifdef GG_WG_STORE
    const int panels = ((N + 127) / 128) * ((K + 255) / 256);
    int splits = (256 + panels - 1) / panels; if (splits >= 8) splits = splits / 8 * 8;
    float* outW; CK(hipMalloc(&outW, (long)N * K * 4)); CK(hipMemset(outW, 0, (long)N * K * 4));
    const long n4 = (long)N * K / 4;
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) {
        gg::wgrad(Y, N, yb, X, K, xb, dW, K, M2, N, K, 0, pf, pg, nullptr, 0);
        reduce_slices<<<(unsigned)((n4 + 255) / 256), 256>>>(dW, outW, n4, splits);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
#else
    CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) gg::wgrad(Y, N, yb, X, K, xb, pg ? nullptr : dW, K, M2, N, K, 0, pf, pg, nullptr, 0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
#endif
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-8s M=%ld N=%d K=%d: %7.1f us  %.2f TB/s algorithmic  %.0f TFLOP/s\n", cs, M2, N, K, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12,
           2.0 * M2 * N * K / (ms / 20 * 1e-3) / 1e12);
    return 0;
}
