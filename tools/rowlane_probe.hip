// Row-per-lane vs line-coalesced global access for token tiles [32 rows x 512 B] at 8 waves per CU (the fused encoder kernels of
// csrc/enc.hip): (0) lane (c, h) reads row c, bytes [64 t + 32 h, +32) as two 16-byte loads - one wave instruction touches 32 rows;
// (1) lane L reads row L/8 + 8 i, bytes [128 s + 16 (L % 8), +16) - one wave instruction = eight full 128-byte lines.  Same bytes.
// Stores likewise (modes 2 / 3).   build: hipcc --offload-arch=gfx950 -O3 -o tools/rowlane_probe tools/rowlane_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(unsigned char* buf, long nunits, unsigned* sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    u32x4 acc = {0, 0, 0, 0};
    for (long u = blockIdx.x + (long)gridDim.x * wave; u < nunits; u += (long)gridDim.x * 8) {
        unsigned char* tile = buf + u * 32 * 512;
        if (MODE == 0 || MODE == 2) {
            unsigned char* p = tile + c * 512 + 32 * h;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (MODE == 0) { acc += *reinterpret_cast<const u32x4*>(p + 64 * t); acc += *reinterpret_cast<const u32x4*>(p + 64 * t + 16); }
                else { *reinterpret_cast<u32x4*>(p + 64 * t) = acc; *reinterpret_cast<u32x4*>(p + 64 * t + 16) = acc; }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned char* p = tile + ((lane >> 3) + 8 * i) * 512 + 128 * s + 16 * (lane & 7);
                    if (MODE == 1) acc += *reinterpret_cast<const u32x4*>(p);
                    else *reinterpret_cast<u32x4*>(p) = acc;
                }
        }
    }
    if (acc[0] == 0x12345678u) sink[0] = acc[1] + acc[2] + acc[3];
}
int main() {
    const long M = 197376, nunits = M / 32;
    unsigned char* buf; unsigned* sink;
    CK(hipMalloc(&buf, M * 512)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, M * 512));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* nm[4] = {"row-per-lane loads ", "line-coalesced loads", "row-per-lane stores", "line-coalesced stores"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
        auto go = [&]() {
            if (mode == 0) k<0><<<256, 512>>>(buf, nunits, sink); else if (mode == 1) k<1><<<256, 512>>>(buf, nunits, sink);
            else if (mode == 2) k<2><<<256, 512>>>(buf, nunits, sink); else k<3><<<256, 512>>>(buf, nunits, sink);
        };
        for (int i = 0; i < 3; ++i) go();
        CK(hipEventRecord(a));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) go();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("%s: %.1f us per pass of %ld MB, %.2f TB/s\n", nm[mode], ms * 1e3 / reps, M * 512 / 1000000, M * 512.0 / (ms * 1e-3 / reps) / 1e12);
    }
    return 0;
}
