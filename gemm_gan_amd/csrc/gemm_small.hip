// Latency-optimised bf16-MFMA GEMM for the SMALL products of the step (MLP heads, gradient-penalty chain,
// cross-attention projections: M <= ~1000 rows).  Same GemmP contract / epilogue as gemm_bf16.hip, different shape:
// 64x64 output tiles (4x more workgroups than 128x128) and a 256-deep K slab staged in ONE memory round trip
// (every thread has its 32 16-byte loads in flight at once), so a K <= 256 product costs one latency instead of
// four; deeper K either loops over slabs or is split over workgroups (fp32 atomics) by the caller.
#include "gg_common.h"

namespace gg {
namespace {
constexpr int TM = 64, TN = 64, TK = 256, NT = 256;
constexpr int LDS_LD = TK + 8;                         // bf16 per LDS row (528 B)
constexpr int SMEM = 2 * TM * LDS_LD * 2;              // A + B slabs = 67,584 B

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// stage one 64 x 256 operand slab into LDS as [row][k] bf16; rows_total / kend bound the valid region
template <int LAY>
__device__ __forceinline__ void stage(__bf16* tile, const float* __restrict__ base, long ld, int rows_total, int row0, int k0,
                                      int kend, bool vec, int tid) {
    if (LAY == LAY_KC) {
        // thread: row = tid >> 2, 64 consecutive k starting at 64 * (tid & 3): 16 float4 loads, 8 x 16-byte LDS stores
        const int row = tid >> 2, kq = (tid & 3) * 64;
        const int gr = row0 + row;
        f32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int gk = k0 + kq + 4 * i;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gr < rows_total) {
                const float* src = base + (long)gr * ld + gk;
                if (vec) {
                    if (gk < kend) v[i] = *reinterpret_cast<const f32x4*>(src);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (gk + j < kend) v[i][j] = src[j];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u32x4 w = {pack2(v[2 * i][0], v[2 * i][1]), pack2(v[2 * i][2], v[2 * i][3]), pack2(v[2 * i + 1][0], v[2 * i + 1][1]),
                       pack2(v[2 * i + 1][2], v[2 * i + 1][3])};
            *reinterpret_cast<u32x4*>(tile + row * LDS_LD + kq + 8 * i) = w;
        }
    } else {
        // memory is [k][rows]: thread owns 4 adjacent rows (mc) and 8 consecutive k per group (kg, kg+16):
        // 16 float4 loads, register transpose, 8 x 16-byte LDS stores
        const int mc = tid & 15, kg = tid >> 4;
        const int gr = row0 + 4 * mc;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int gk = k0 + 8 * (kg + 16 * half) + j;
                v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (gk < kend) {
                    const float* src = base + (long)gk * ld + gr;
                    if (vec) {
                        if (gr < rows_total) v[j] = *reinterpret_cast<const f32x4*>(src);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (gr + q < rows_total) v[j][q] = src[q];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x4 w = {pack2(v[0][q], v[1][q]), pack2(v[2][q], v[3][q]), pack2(v[4][q], v[5][q]), pack2(v[6][q], v[7][q])};
                *reinterpret_cast<u32x4*>(tile + (4 * mc + q) * LDS_LD + 8 * (kg + 16 * half)) = w;
            }
        }
    }
}

template <int LA, int LB>
__global__ __launch_bounds__(NT) void gemm_small_kernel(const GemmP p, int vecA, int vecB, int kchunk) {
    extern __shared__ __attribute__((aligned(16))) __bf16 sm_small[];
    __bf16* As = sm_small;
    __bf16* Bs = sm_small + TM * LDS_LD;
    const int tid = threadIdx.x;
    const int tiles_n = (p.N + TN - 1) / TN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    const int z = blockIdx.z;
    const int bz = z / p.splitk, sk = z % p.splitk;
    const int zo = bz / p.batch_inner, zi = bz % p.batch_inner;
    const float* A = p.A + zo * p.sAo + zi * p.sAi;
    const float* B = p.B + zo * p.sBo + zi * p.sBi;
    float* C = p.C + zo * p.sCo + zi * p.sCi;
    const int kbeg = sk * kchunk, kend = min(p.K, kbeg + kchunk);

    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        if (k0 > kbeg) __syncthreads();
        stage<LA>(As, A, p.lda, p.M, m0, k0, kend, vecA != 0, tid);
        stage<LB>(Bs, B, p.ldb, p.N, n0, k0, kend, vecB != 0, tid);
        __syncthreads();
        const int steps = (min(TK, kend - k0) + 15) / 16;
        for (int s = 0; s < steps; ++s) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + (wm * 32 + r) * LDS_LD + 16 * s + 8 * h);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + (wn * 32 + r) * LDS_LD + 16 * s + 8 * h);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    const bool atomic = p.splitk > 1;
    if (atomic && kend <= kbeg) return;
    const int col = n0 + wn * 32 + r;
    if (col >= p.N) return;
    const uint8_t* cmask = p.colmask ? p.colmask + (long)(p.colmask_mod > 0 ? zo % p.colmask_mod : zo) * p.colmask_stride : nullptr;
    const float bias = (p.bias && sk == 0) ? p.bias[col] : 0.f;
    const bool masked = cmask && cmask[col];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row >= p.M) continue;
        const long crow = p.c_row_group ? (long)row + row / p.c_row_group + 1 : (long)row;
        float* cp = C + crow * p.ldc + col;
        float v = p.alpha * acc[i] + bias;
        if (atomic) {
            atomicAdd(cp, v);
        } else {
            if (p.accumulate) v += *cp;
            if (p.act == ACT_LRELU) v = v > 0.f ? v : p.slope * v;
            if (masked) v = -INFINITY;
            *cp = v;
        }
    }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
template <int LA, int LB>
int launch(const GemmP& p, int vecA, int vecB, int kchunk, dim3 grid, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_small_kernel<LA, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr = true;
    }
    hipLaunchKernelGGL((gemm_small_kernel<LA, LB>), grid, dim3(NT), SMEM, st, p, vecA, vecB, kchunk);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace

bool gemm_small_wanted(const GemmP& p) {
    if (p.film_gamma || p.a_bf16 || p.b_bf16) return false;
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.batch;
    return tiles128 * p.splitk <= 128;                 // would not fill the chip with 128x128 tiles anyway
}

int gemm_small(const GemmP& p, hipStream_t st) {
    GG_REQUIRE(p.A && p.B && p.C && p.M > 0 && p.N > 0 && p.K >= 0, "bad operands");
    GG_REQUIRE(!(p.splitk > 1 && (p.act != ACT_NONE || p.colmask)), "split-K epilogue must be linear");
    const bool strA = (p.sAo % 4 == 0) && (p.sAi % 4 == 0), strB = (p.sBo % 4 == 0) && (p.sBi % 4 == 0);
    const int vecA = al16(p.A) && p.lda % 4 == 0 && strA && ((p.layA == LAY_KC) ? p.K % 4 == 0 : p.M % 4 == 0);
    const int vecB = al16(p.B) && p.ldb % 4 == 0 && strB && ((p.layB == LAY_KC) ? p.K % 4 == 0 : p.N % 4 == 0);
    int kchunk = (p.K + p.splitk - 1) / p.splitk;
    kchunk = (kchunk + 15) / 16 * 16;
    if (kchunk <= 0) kchunk = 16;
    const long tiles = (long)((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN);
    dim3 grid((unsigned)tiles, 1, (unsigned)(p.batch * p.splitk));
    if (p.layA == LAY_KC && p.layB == LAY_KC) return launch<LAY_KC, LAY_KC>(p, vecA, vecB, kchunk, grid, st);
    if (p.layA == LAY_KC && p.layB == LAY_KS) return launch<LAY_KC, LAY_KS>(p, vecA, vecB, kchunk, grid, st);
    if (p.layA == LAY_KS && p.layB == LAY_KC) return launch<LAY_KS, LAY_KC>(p, vecA, vecB, kchunk, grid, st);
    return launch<LAY_KS, LAY_KS>(p, vecA, vecB, kchunk, grid, st);
}
}  // namespace gg
