// Store-pattern probe: bf16 output tiles [32 tokens x 512 B rows] written (a) as the weight-stationary epilogues do - a lane owns one
// token row and writes 16-byte pieces, so one store instruction touches 32 rows - or (b) row-contiguous (64 lanes x 16 B = two full
// 512-byte rows per instruction).   build: hipcc --offload-arch=gfx950 -O3 -o tools/store_probe tools/store_probe.hip ; run: ./tools/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// rows of 256 bf16 (512 B).  persistent grid, 8 waves per workgroup, tile = 32 rows
template <int MODE>
__global__ __launch_bounds__(512, 2) void store_k(unsigned char* out, long ntiles) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        unsigned char* tile = out + t * 32 * 512;
        if (MODE == 0) {           // lane (c, h) of wave w: row c, bytes [64 w + 32 h, +32) as two 16-byte stores
            unsigned char* p = tile + c * 512 + 64 * wave + 32 * h;
            *reinterpret_cast<u32x4*>(p) = v;
            *reinterpret_cast<u32x4*>(p + 16) = v;
        } else {                   // thread i: 16-byte piece i of the tile in row-major order (32 pieces per row), two passes
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int piece = tid + 512 * k;
                *reinterpret_cast<u32x4*>(tile + piece * 16) = v;
            }
        }
    }
}
int main() {
    const long M = 197376, ntiles = M / 32;
    unsigned char* buf;
    CK(hipMalloc(&buf, M * 512));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 2; ++mode)
        for (int grid : {233, 466}) {
            for (int i = 0; i < 3; ++i) { if (mode == 0) store_k<0><<<grid, 512>>>(buf, ntiles); else store_k<1><<<grid, 512>>>(buf, ntiles); }
            CK(hipEventRecord(a));
            const int reps = 20;
            for (int i = 0; i < reps; ++i) { if (mode == 0) store_k<0><<<grid, 512>>>(buf, ntiles); else store_k<1><<<grid, 512>>>(buf, ntiles); }
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            printf("%s grid %d: %.1f us per pass, %.2f TB/s\n", mode == 0 ? "lane-per-row 16 B pieces" : "row-contiguous        ", grid, ms * 1e3 / reps, M * 512.0 / (ms * 1e-3 / reps) / 1e12);
        }
    return 0;
}
