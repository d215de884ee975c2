// Access-pattern bandwidth probe for MI355X (tools only).  hipcc --offload-arch=gfx950 -O3 bw_probe.hip -o bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (a) a wave reads 64 consecutive rows of a [M][K] fp32 matrix, one full 1 KiB row (K=256) per instruction
__global__ void rd_fullrow(const float* X, float* out, long M, int K) {
    const int lane = threadIdx.x & 63; const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    f4 acc = {0, 0, 0, 0};
    for (int i = 0; i < 64; ++i) { long r = w * 64 + i; if (r < M) for (int c = lane; c < K / 4; c += 64) acc += *(const f4*)(X + r * K + 4 * c); }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = acc[0];
}
// (b) same rows, but in 256-byte column windows (4 rows x 256 B per instruction), window after window
__global__ void rd_window(const float* X, float* out, long M, int K) {
    const int lane = threadIdx.x & 63; const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    f4 acc = {0, 0, 0, 0};
    for (int q = 0; q < K / 64; ++q)
        for (int i = 0; i < 16; ++i) { long r = w * 64 + i * 4 + (lane >> 4); if (r < M) acc += *(const f4*)(X + r * K + q * 64 + 4 * (lane & 15)); }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = acc[0];
}
// (c) a lane owns a row (token) and writes 16-byte pieces: per instruction 32 rows x 32 B (tlin epilogue shape)
__global__ void wr_lane_rows(float* Y, long M, int N) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5; const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    for (int t = 0; t < 2; ++t) { long r = w * 64 + t * 32 + c; if (r >= M) continue;
        for (int nt = 0; nt < N / 32; ++nt) for (int g = 0; g < 4; ++g) { f4 v = {1.f * nt, 2, 3, 4}; *(f4*)(Y + r * N + nt * 32 + 8 * g + 4 * h) = v; } }
}
// (d) full-row contiguous writes (1 KiB per instruction)
__global__ void wr_fullrow(float* Y, long M, int N) {
    const int lane = threadIdx.x & 63; const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    for (int i = 0; i < 64; ++i) { long r = w * 64 + i; if (r < M) for (int c = lane; c < N / 4; c += 64) { f4 v = {1.f * i, 2, 3, 4}; *(f4*)(Y + r * N + 4 * c) = v; } }
}
// (e) tile-GEMM style epilogue: per instruction 2 x 128 B row segments (32 lanes x 4 B), 64 instructions
__global__ void wr_tile4B(float* Y, long M, int N) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5; const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long tiles_n = N / 32; const long tm = w / tiles_n, tn = w % tiles_n;   // a wave owns a 32x32 tile
    for (int i = 0; i < 16; ++i) { long r = tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h; if (r < M) Y[r * N + tn * 32 + c] = 1.f * i; }
}

// (f) tlin resident out_proj traffic shape with every load of a 32-token wave tile issued up front:
//     X bf16 [M][256] (row windows), residual fp32 [M][256] lane-owns-row, two fp32 outputs lane-owns-row.
//     lds bytes force the occupancy (1 block / CU at > 80 KB).
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mix_probe(const unsigned short* X, const float* R, float* Y0, float* Y1, long M) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5; const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long t0 = w * 32; if (t0 >= M) return;
    u4 xv[16]; f4 rr[32];
    for (int i = 0; i < 16; ++i) { long r = t0 + 2 * i + (lane >> 5); xv[i] = *(const u4*)(X + r * 256 + 8 * (lane & 31)); }
    const long tok = t0 + c;
    for (int i = 0; i < 32; ++i) rr[i] = *(const f4*)(R + tok * 256 + (i >> 2) * 32 + 8 * (i & 3) + 4 * h);
    unsigned s = 0; for (int i = 0; i < 16; ++i) s += xv[i][0] ^ xv[i][1] ^ xv[i][2] ^ xv[i][3];
    if (smem && s == 0x12345u) smem[lane] = 1;
    const float sf = (float)(s & 1);
    for (int i = 0; i < 32; ++i) { f4 v = rr[i] + sf; *(f4*)(Y0 + tok * 256 + (i >> 2) * 32 + 8 * (i & 3) + 4 * h) = v; }
    for (int i = 0; i < 32; ++i) { f4 v = rr[i] * 2.f + sf; *(f4*)(Y1 + tok * 256 + (i >> 2) * 32 + 8 * (i & 3) + 4 * h) = v; }
}
// (g) plain grid-stride copy, 16 B per lane
__global__ void copy_probe(const f4* a, f4* b, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

// (h) bf16 output, lane owns a row and writes 8-byte pieces (stream-kernel epilogue, 32x32 MFMA layout): 32 rows x 2 x 8 B
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void wr_lane_rows_bf16(unsigned short* Y, long M, int N) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5; const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    long r = w * 32 + c; if (r >= M) return;
    for (int nt = 0; nt < N / 32; ++nt) for (int g = 0; g < 4; ++g) { u2 v = {(unsigned)nt, 2u}; *(u2*)(Y + r * N + nt * 32 + 8 * g + 4 * h) = v; }
}
// (i) same bytes, each instruction writes 16 rows x 64 contiguous bytes (4 lanes x 16 B per row)
__global__ void wr_rows64_bf16(unsigned short* Y, long M, int N) {
    const int lane = threadIdx.x & 63; const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    for (int nt = 0; nt < N / 32; ++nt) for (int i = 0; i < 2; ++i) { long r = w * 32 + i * 16 + (lane >> 2); if (r < M) { u4 v = {(unsigned)nt, 2u, 3u, 4u}; *(u4*)(Y + r * N + nt * 32 + 8 * (lane & 3)) = v; } }
}
// (j) same bytes, 32 B per lane (16 features): 32 rows x 2 x 32 B via two 16-B stores
__global__ void wr_lane32_bf16(unsigned short* Y, long M, int N) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5; const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    long r = w * 32 + c; if (r >= M) return;
    for (int nt = 0; nt < N / 32; ++nt) for (int g = 0; g < 2; ++g) { u4 v = {(unsigned)nt, 2u, 3u, 4u}; *(u4*)(Y + r * N + nt * 32 + 16 * h + 8 * g) = v; }
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 10; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10 * 1e3f;
}
int main() {
    const long M = 197376; float *X, *Y, *out;
    CK(hipMalloc(&X, M * 1024 * 4)); CK(hipMalloc(&Y, M * 1024 * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(X, 0, M * 1024 * 4));
    const long waves = (M + 63) / 64; const int blocks = (int)((waves + 3) / 4);
    for (int K : {256, 512, 1024}) {
        float t = timeit([&] { rd_fullrow<<<blocks, 256>>>(X, out, M, K); });
        printf("read  full rows      K=%4d: %7.1f us %6.2f TB/s\n", K, t, M * K * 4.0 / t / 1e6);
        t = timeit([&] { rd_window<<<blocks, 256>>>(X, out, M, K); });
        printf("read  256B windows   K=%4d: %7.1f us %6.2f TB/s\n", K, t, M * K * 4.0 / t / 1e6);
    }
    for (int N : {256, 768}) {
        float t = timeit([&] { wr_lane_rows<<<blocks, 256>>>(Y, M, N); });
        printf("write lane-owns-row  N=%4d: %7.1f us %6.2f TB/s\n", N, t, M * N * 4.0 / t / 1e6);
        t = timeit([&] { wr_fullrow<<<blocks, 256>>>(Y, M, N); });
        printf("write full rows      N=%4d: %7.1f us %6.2f TB/s\n", N, t, M * N * 4.0 / t / 1e6);
        const long w2 = (M / 32) * (N / 32); const int b2 = (int)((w2 + 3) / 4);
        t = timeit([&] { wr_tile4B<<<b2, 256>>>(Y, M, N); });
        printf("write 32x32 tile 4B  N=%4d: %7.1f us %6.2f TB/s\n", N, t, M * N * 4.0 / t / 1e6);
    }
    {
        unsigned short* Xb = (unsigned short*)X; float* R = X + M * 128; float* Y1 = Y + M * 256;
        const long wv = M / 32; const int bl = (int)((wv + 3) / 4);
        for (int lds : {0, 40000, 60000, 100000}) {
            hipFuncSetAttribute((const void*)mix_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            float t = timeit([&] { mix_probe<<<bl, 256, lds>>>(Xb, R, Y, Y1, M); });
            printf("mix out_proj shape lds=%6d: %7.1f us %6.2f TB/s\n", lds, t, M * 3584.0 / t / 1e6);
        }
        for (long MM : {65792L, 131584L}) {
            const int b2 = (int)((MM / 32 + 3) / 4);
            hipFuncSetAttribute((const void*)mix_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100000);
            float t = timeit([&] { mix_probe<<<b2, 256, 100000>>>(Xb, R, Y, Y1, MM); });
            printf("mix out_proj shape M=%ld lds=100000: %7.1f us %6.2f TB/s\n", MM, t, MM * 3584.0 / t / 1e6);
        }
        const long n = M * 256;   // 16-B elements: 808 MB each way
        float t = timeit([&] { copy_probe<<<256 * 8, 256>>>((const f4*)X, (f4*)Y, n); });
        printf("copy 2x%.0f MB: %7.1f us %6.2f TB/s (read+write)\n", n * 16.0 / 1e6, t, 2.0 * n * 16 / t / 1e6);
    }
    for (int N : {512, 768}) {
        const long wv = (M + 31) / 32; const int bl = (int)((wv + 3) / 4);
        float t = timeit([&] { wr_lane_rows_bf16<<<bl, 256>>>((unsigned short*)Y, M, N); });
        printf("write bf16 lane-row 8B   N=%4d: %7.1f us %6.2f TB/s\n", N, t, M * N * 2.0 / t / 1e6);
        t = timeit([&] { wr_rows64_bf16<<<bl, 256>>>((unsigned short*)Y, M, N); });
        printf("write bf16 16rows x 64B  N=%4d: %7.1f us %6.2f TB/s\n", N, t, M * N * 2.0 / t / 1e6);
        t = timeit([&] { wr_lane32_bf16<<<bl, 256>>>((unsigned short*)Y, M, N); });
        printf("write bf16 lane 2x16B    N=%4d: %7.1f us %6.2f TB/s\n", N, t, M * N * 2.0 / t / 1e6);
    }
    return 0;
}
