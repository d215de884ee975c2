// Standalone timing harness for the token-on-lane Linear kernels (tools only; includes the kernel source directly).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGG_TLIN_DBG_RT -I gemm_gan_amd/csrc tools/tlin_probe.hip -o tools/tlin_probe
//   ./tools/tlin_probe <case> [M]     cases: qkv ffn1 outproj ffn2 dmask
#include "../gemm_gan_amd/csrc/tlin.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
namespace gg { void set_error(const std::string& s) { fprintf(stderr, "gg error: %s\n", s.c_str()); } 
DropKey make_drop_key(float p, uint64_t, uint32_t, uint32_t) { DropKey k; k.p = p; k.k0 = 12345u; k.thr = (uint32_t)(p * 65536.f + 0.5f); return k; } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const char* cs = argc > 1 ? argv[1] : "qkv";
    const long M = argc > 2 ? atol(argv[2]) : 197376;
    const int E = 256, F = 512;
    void *X, *Y, *Y2, *W, *R; float *bias, *g, *b, *stats; unsigned long long* stamps;
    CK(hipMalloc(&X, M * 1024 * 4)); CK(hipMalloc(&Y, M * 1024 * 4)); CK(hipMalloc(&Y2, M * 256 * 4)); CK(hipMalloc(&R, M * 512 * 4));
    CK(hipMalloc(&W, 1024 * 1024 * 2)); CK(hipMalloc(&bias, 4096)); CK(hipMalloc(&g, 4096)); CK(hipMalloc(&b, 4096)); CK(hipMalloc(&stats, M * 8));
    CK(hipMalloc(&stamps, (M / 64 + 8) * 32));
    CK(hipMemset(X, 0, M * 1024 * 4)); CK(hipMemset(W, 0, 1024 * 1024 * 2)); CK(hipMemset(R, 0, M * 512 * 4)); CK(hipMemset(bias, 0, 4096)); CK(hipMemset(g, 0, 4096)); CK(hipMemset(b, 0, 4096));
    gg::TlinP t; t.M = M; t.W = W; t.bias = bias;
    double bytes = 0;
    if (!strcmp(cs, "qkv")) { t.X = X; t.ldx = E; t.K = E; t.ldw = E; t.Y = Y; t.ldy = 3 * E; t.N = 3 * E; t.y_bf16 = 1; bytes = M * (1024.0 + 1536); }
    else if (!strcmp(cs, "ffn1")) { t.X = X; t.ldx = E; t.K = E; t.ldw = E; t.Y = Y; t.ldy = F; t.N = F; t.y_bf16 = 1; t.act_relu = 1; t.drop = gg::make_drop_key(0.1f, 1, 1, 1); t.drop_ld = F; bytes = M * (1024.0 + 1024); }
    else if (!strcmp(cs, "dmask")) { t.X = X; t.x_bf16 = 1; t.ldx = E; t.K = E; t.ldw = E; t.Y = Y; t.ldy = F; t.N = F; t.y_bf16 = 1; t.mask_ref = R; t.ldref = F; t.mask_bf16 = 1; t.mask_scale = 1.1f; t.bias = nullptr; bytes = M * (512.0 + 1024 + 1024); }
    else if (!strcmp(cs, "outproj")) { t.X = X; t.x_bf16 = 1; t.ldx = E; t.K = E; t.ldw = E; t.Y = Y; t.ldy = E; t.N = E; t.drop = gg::make_drop_key(0.1f, 1, 1, 1); t.drop_ld = E; t.res = (const float*)R; t.ldres = E; t.res_rows = M; t.ln_g = g; t.ln_b = b; t.ln_y = (float*)Y2; t.ln_stats = stats; bytes = M * (512.0 + 1024 + 2048); }
    else if (!strcmp(cs, "ffn2")) { t.X = X; t.x_bf16 = 1; t.ldx = F; t.K = F; t.ldw = F; t.Y = Y; t.ldy = E; t.N = E; t.drop = gg::make_drop_key(0.1f, 1, 1, 1); t.drop_ld = E; t.res = (const float*)R; t.ldres = E; t.res_rows = M; t.ln_g = g; t.ln_b = b; t.ln_y = (float*)Y2; t.ln_stats = stats; bytes = M * (1024.0 + 1024 + 2048); }
    else { printf("unknown case\n"); return 1; }
    if (!gg::tlin_supported(t)) { printf("unsupported\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (gg::tlin(t, 0)) return 1;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) gg::tlin(t, 0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s M=%ld: %.1f us  %.2f TB/s algorithmic\n", cs, M, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
    // stamps (stream kernel only)
    t.stamps = stamps; CK(hipMemset(stamps, 0, (M / 64 + 8) * 32));
    gg::tlin(t, 0); CK(hipDeviceSynchronize());
    const long nb = (M + 127) / 128;
    std::vector<unsigned long long> h(4 * nb); CK(hipMemcpy(h.data(), stamps, 32 * nb, hipMemcpyDeviceToHost));
    if (h[3] != 0) {
        unsigned long long t0 = ~0ull, t1 = 0; std::vector<double> dx, dit, dtot;
        for (long i = 0; i < nb; ++i) { t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 3]); dx.push_back(double(h[4 * i + 1] - h[4 * i])); dit.push_back(double(h[4 * i + 3] - h[4 * i + 2])); dtot.push_back(double(h[4 * i + 3] - h[4 * i])); }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("stamps (ticks): kernel span %.0f; per block median: X staging %.0f, iterations after the first pair %.0f, total %.0f; blocks %ld\n", double(t1 - t0), med(dx), med(dit), med(dtot), nb);
        // start-time distribution: how many blocks started in each tenth of the span
        int hist[10] = {0}; for (long i = 0; i < nb; ++i) hist[std::min(9, int(10.0 * (h[4 * i] - t0) / double(t1 - t0)))]++;
        printf("block starts per tenth of the span:"); for (int i = 0; i < 10; ++i) printf(" %d", hist[i]); printf("\n");
    }
    return 0;
}
