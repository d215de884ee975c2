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

// One 64 x 256 operand slab on its way into LDS as [row][k] bf16; rows_total / kend bound the valid region.
// load() and store() are separate so that BOTH operands' loads are in flight before either is written to LDS, and the
// aligned path (vec: every 16-byte piece is wholly inside or wholly outside the operand) issues its 16 loads
// unconditionally from clamped addresses and blanks the outside pieces afterwards with selects - no branch, no wait
// between loads.  (With a guard around each load the compiler waited for every load before issuing the next: ~70
// serialised round trips per slab, 10-16 us for a product whose arithmetic takes one.)
template <int LAY>
struct SlabStager {
    f32x4 v[16];
    unsigned blank = 0;       // aligned path: pieces that lie outside the operand (zeroed when written to LDS, not before:
                              // a select right behind the loads would make the other operand's loads wait for these)
    __device__ __forceinline__ void load(const float* __restrict__ base, long ld, int rows_total, int row0, int k0, int kend, bool vec, int tid) {
        if (LAY == LAY_KC) {
            // thread: row = tid >> 2, 64 consecutive k starting at 64 * (tid & 3): 16 float4 loads
            const int row = tid >> 2, kq = (tid & 3) * 64;
            const int gr = row0 + row;
            if (vec) {
                const float* src = base + (long)min(gr, rows_total - 1) * ld;
                const bool rok = gr < rows_total;
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + min(k0 + kq + 4 * i, kend - 4));
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (!(rok && k0 + kq + 4 * i < kend)) blank |= 1u << i;
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int gk = k0 + kq + 4 * i;
                    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (gr < rows_total) {
                        const float* src = base + (long)gr * ld + gk;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (gk + j < kend) v[i][j] = src[j];
                    }
                }
            }
        } else {
            // memory is [k][rows]: thread owns 4 adjacent rows (mc) and 8 consecutive k per group (kg, kg+16): 16 float4 loads
            const int mc = tid & 15, kg = tid >> 4;
            const int gr = row0 + 4 * mc;
            if (vec) {
                const float* src = base + min(gr, rows_total - 4);
                const bool rok = gr < rows_total;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int gk = k0 + 8 * (kg + 16 * (i >> 3)) + (i & 7);
                    v[i] = *reinterpret_cast<const f32x4*>(src + (long)min(gk, kend - 1) * ld);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (!(rok && k0 + 8 * (kg + 16 * (i >> 3)) + (i & 7) < kend)) blank |= 1u << i;
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int gk = k0 + 8 * (kg + 16 * (i >> 3)) + (i & 7);
                    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (gk < kend) {
                        const float* src = base + (long)gk * ld + gr;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (gr + q < rows_total) v[i][q] = src[q];
                    }
                }
            }
        }
    }
    __device__ __forceinline__ void store(__bf16* tile, int tid) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if ((blank >> i) & 1u) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (LAY == LAY_KC) {
            const int row = tid >> 2, kq = (tid & 3) * 64;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u32x4 w = {pack2(v[2 * i][0], v[2 * i][1]), pack2(v[2 * i][2], v[2 * i][3]), pack2(v[2 * i + 1][0], v[2 * i + 1][1]),
                           pack2(v[2 * i + 1][2], v[2 * i + 1][3])};
                *reinterpret_cast<u32x4*>(tile + row * LDS_LD + kq + 8 * i) = w;
            }
        } else {
            const int mc = tid & 15, kg = tid >> 4;       // register transpose: 8 consecutive k of one row per 16-byte store
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4* u = v + 8 * half;
                    u32x4 w = {pack2(u[0][q], u[1][q]), pack2(u[2][q], u[3][q]), pack2(u[4][q], u[5][q]), pack2(u[6][q], u[7][q])};
                    *reinterpret_cast<u32x4*>(tile + (4 * mc + q) * LDS_LD + 8 * (kg + 16 * half)) = w;
                }
        }
    }
};

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
        SlabStager<LA> sa;
        SlabStager<LB> sb;
        sa.load(A, p.lda, p.M, m0, k0, kend, vecA != 0, tid);
        sb.load(B, p.ldb, p.N, n0, k0, kend, vecB != 0, tid);
        sa.store(As, tid);
        sb.store(Bs, tid);
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

// ------------------------------------------------------------------------------------------------------------------
// "Tiny" variant for the aligned case: 32 x 32 output tiles (4x the workgroups of the 64 x 64 kernel - the head products
// are 768 x 256: 48 tiles there, 192 here, of 256 CUs), the four waves of a workgroup split K between them, and every
// wave takes its MFMA fragments STRAIGHT FROM GLOBAL MEMORY into registers: a lane's A fragment is 8 consecutive k of
// its row, i.e. two float4 loads from a K-contiguous operand (eight coalesced dword loads from a K-strided one), so
// there is no LDS staging, no barrier before the first MFMA and a dependent chain of K/64 MFMAs per wave instead of K/16.
// One LDS exchange adds the four partial tiles; wave w finishes rows 8w .. 8w+7 of the tile.  Same GemmP epilogue.
constexpr int QT = 32, TG = 4;         // tile edge; 16-wide k steps whose loads are in flight together
template <int LAY>
struct FragLoader {
    float x[TG][8];
    // step s of the group starting at k = kk0 (this lane: + 8h); clamped addresses, validity applied in get()
    __device__ __forceinline__ void load(const float* __restrict__ base, long ld, int row_c, int kk0, int kmax8) {
#pragma unroll
        for (int s = 0; s < TG; ++s) {
            const int kk = min(kk0 + 16 * s, kmax8);
            if (LAY == LAY_KC) {
                const float* q = base + (long)row_c * ld + kk;
                const f32x4 a = *reinterpret_cast<const f32x4*>(q), b = *reinterpret_cast<const f32x4*>(q + 4);
                x[s][0] = a[0]; x[s][1] = a[1]; x[s][2] = a[2]; x[s][3] = a[3];
                x[s][4] = b[0]; x[s][5] = b[1]; x[s][6] = b[2]; x[s][7] = b[3];
            } else {
                const float* q = base + (long)kk * ld + row_c;
#pragma unroll
                for (int j = 0; j < 8; ++j) x[s][j] = q[(long)j * ld];
            }
        }
    }
    __device__ __forceinline__ bf16x8 get(int s, bool valid) const {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = valid ? x[s][j] : 0.f;
        const u32x4 w = {pack2(y[0], y[1]), pack2(y[2], y[3]), pack2(y[4], y[5]), pack2(y[6], y[7])};
        return __builtin_bit_cast(bf16x8, w);
    }
    // the fragment as NS bf16 parts (hi, then the roundings of the successive remainders): the split-operand form of GG_PREC_BF16X3
    template <int NS>
    __device__ __forceinline__ void get_parts(int s, bool valid, bf16x8 (&out)[NS]) const {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = valid ? x[s][j] : 0.f;
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
            __bf16 b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = (__bf16)y[j];
            const bf16x2_t w0 = {b[0], b[1]}, w1 = {b[2], b[3]}, w2 = {b[4], b[5]}, w3 = {b[6], b[7]};
            const u32x4 w = {__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w1), __builtin_bit_cast(unsigned, w2),
                             __builtin_bit_cast(unsigned, w3)};
            out[sp] = __builtin_bit_cast(bf16x8, w);
            if (sp + 1 < NS) {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] -= (float)b[j];
            }
        }
    }
};

// NS = 1: bf16 operands.  NS = 3: both operands as three bf16 parts, six part products per step, fp32 accumulate - fp32-grade results
// for the few-tile products of the bf16x3 parity mode (MLP heads, cross-attention projections), which otherwise run on the
// 128 x 128 fp32-input tile kernel at ~50 us per launch
template <int LA, int LB, int NS = 1>
__global__ __launch_bounds__(NT) void gemm_tiny_kernel(const GemmP p, int kchunk) {
    __shared__ float red[4][16][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_n = (p.N + QT - 1) / QT;
    const int m0 = (blockIdx.x / tiles_n) * QT, n0 = (blockIdx.x % tiles_n) * QT;
    const int z = blockIdx.z;
    const int bz = z / p.splitk, sk = z % p.splitk;
    const int zo = bz / p.batch_inner, zi = bz % p.batch_inner;
    const float* A = p.A + zo * p.sAo + zi * p.sAi;
    const float* B = p.B + zo * p.sBo + zi * p.sBi;
    float* C = p.C + zo * p.sCo + zi * p.sCi;
    const int kbeg = sk * kchunk, kend = min(p.K, kbeg + kchunk);
    // this wave's share of [kbeg, kend), in whole 16-wide steps
    const int kw = ((kend - kbeg + 3) / 4 + 15) / 16 * 16;
    const int kw0 = kbeg + wave * kw, kw1 = min(kend, kw0 + kw);
    const int ra = min(m0 + r, p.M - 1), rb = min(n0 + r, p.N - 1);
    const int kmax8 = kend - 8;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (kw0 < kw1) {
        FragLoader<LA> fa[2];
        FragLoader<LB> fb[2];
        const int ngroups = (kw1 - kw0 + 16 * TG - 1) / (16 * TG);
        fa[0].load(A, p.lda, ra, kw0 + 8 * h, kmax8);
        fb[0].load(B, p.ldb, rb, kw0 + 8 * h, kmax8);
        for (int g0 = 0; g0 < ngroups; g0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int g = g0 + u;
                if (g >= ngroups) break;
                if (g + 1 < ngroups) {
                    fa[u ^ 1].load(A, p.lda, ra, kw0 + 16 * TG * (g + 1) + 8 * h, kmax8);
                    fb[u ^ 1].load(B, p.ldb, rb, kw0 + 16 * TG * (g + 1) + 8 * h, kmax8);
                }
#pragma unroll
                for (int s = 0; s < TG; ++s) {
                    const bool valid = kw0 + 16 * TG * g + 16 * s + 8 * h < kw1;
                    if constexpr (NS == 1) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u].get(s, valid), fb[u].get(s, valid), acc, 0, 0, 0);
                    } else {
                        bf16x8 a3[NS], b3[NS];
                        fa[u].template get_parts<NS>(s, valid, a3);
                        fb[u].template get_parts<NS>(s, valid, b3);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[2], b3[0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[0], acc, 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    const bool atomic = p.splitk > 1;
    if (atomic && kend <= kbeg) return;
    const int col = n0 + r;
    if (col >= p.N) return;
    const uint8_t* cmask = p.colmask ? p.colmask + (long)(p.colmask_mod > 0 ? zo % p.colmask_mod : zo) * p.colmask_stride : nullptr;
    const float bias = (p.bias && sk == 0) ? p.bias[col] : 0.f;
    const bool masked = cmask && cmask[col];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * wave + q;                                   // accumulator register this wave finishes
        const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row >= p.M) continue;
        const float sum = red[0][i][lane] + red[1][i][lane] + red[2][i][lane] + red[3][i][lane];
        const long crow = p.c_row_group ? (long)row + row / p.c_row_group + 1 : (long)row;
        float* cp = C + crow * p.ldc + col;
        float v = p.alpha * sum + bias;
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

namespace {
bool tiny_ok(const GemmP& p) {
    const bool strA = (p.sAo % 4 == 0) && (p.sAi % 4 == 0), strB = (p.sBo % 4 == 0) && (p.sBi % 4 == 0);
    const bool kcA = p.layA == LAY_KC, kcB = p.layB == LAY_KC;
    const bool okA = kcA ? (al16(p.A) && p.lda % 4 == 0 && strA) : true, okB = kcB ? (al16(p.B) && p.ldb % 4 == 0 && strB) : true;
    const long tiles32 = (long)((p.M + QT - 1) / QT) * ((p.N + QT - 1) / QT);
    return okA && okB && p.K % 8 == 0 && p.K >= 8 && tiles32 * p.batch * p.splitk <= 4096;
}
}  // namespace
// split-operand (six bf16 part products, fp32-grade) form of the register-direct 32 x 32 kernel; false: the caller keeps its fp32 kernel
bool gemm_small_x3_ok(const GemmP& p) {
    if (p.film_gamma || p.a_bf16 || p.b_bf16 || !p.A || !p.B || !p.C || p.M <= 0 || p.N <= 0) return false;
    if (p.splitk > 1 && (p.act != ACT_NONE || p.colmask)) return false;
    return tiny_ok(p);
}
int gemm_small_x3(const GemmP& p, hipStream_t st) {
    GG_REQUIRE(gemm_small_x3_ok(p), "gemm_small_x3: operands not aligned for the register-direct kernel");
    int kchunk = (p.K + p.splitk - 1) / p.splitk;
    kchunk = (kchunk + 15) / 16 * 16;
    const long tiles32 = (long)((p.M + QT - 1) / QT) * ((p.N + QT - 1) / QT);
    dim3 grid32((unsigned)tiles32, 1, (unsigned)(p.batch * p.splitk));
    const bool kcA = p.layA == LAY_KC, kcB = p.layB == LAY_KC;
#define GG_TINY3(LA_, LB_) hipLaunchKernelGGL((gemm_tiny_kernel<LA_, LB_, 3>), grid32, dim3(NT), 0, st, p, kchunk)
    if (kcA && kcB) GG_TINY3(LAY_KC, LAY_KC);
    else if (kcA) GG_TINY3(LAY_KC, LAY_KS);
    else if (kcB) GG_TINY3(LAY_KS, LAY_KC);
    else GG_TINY3(LAY_KS, LAY_KS);
#undef GG_TINY3
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

bool gemm_small_wanted(const GemmP& p) {
    if (p.film_gamma || p.a_bf16 || p.b_bf16) return false;
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.batch;
    static const long cap = getenv("GG_SMALL_MAX") ? atol(getenv("GG_SMALL_MAX")) : 256;      // (256: the gene-dimension products of the critic head, 36 -> 31 and 46 -> 31 us)
    return tiles128 * p.splitk <= cap;                 // would not fill the chip with 128x128 tiles anyway
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
    // aligned operands whose K is a whole number of 8-element fragment halves: the register-direct 32 x 32 kernel
    static const bool no_tiny = getenv("GG_NO_GEMM_TINY") != nullptr;
    const bool kcA = p.layA == LAY_KC, kcB = p.layB == LAY_KC;
    const bool okA = kcA ? (al16(p.A) && p.lda % 4 == 0 && strA) : true, okB = kcB ? (al16(p.B) && p.ldb % 4 == 0 && strB) : true;
    const long tiles32 = (long)((p.M + QT - 1) / QT) * ((p.N + QT - 1) / QT);
    if (!no_tiny && okA && okB && p.K % 8 == 0 && p.K >= 8 && tiles32 * p.batch * p.splitk <= 4096) {
        dim3 grid32((unsigned)tiles32, 1, (unsigned)(p.batch * p.splitk));
#define GG_TINY(LA_, LB_) hipLaunchKernelGGL((gemm_tiny_kernel<LA_, LB_>), grid32, dim3(NT), 0, st, p, kchunk)
        if (kcA && kcB) GG_TINY(LAY_KC, LAY_KC);
        else if (kcA) GG_TINY(LAY_KC, LAY_KS);
        else if (kcB) GG_TINY(LAY_KS, LAY_KC);
        else GG_TINY(LAY_KS, LAY_KS);
#undef GG_TINY
        GG_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const long tiles = (long)((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN);
    dim3 grid((unsigned)tiles, 1, (unsigned)(p.batch * p.splitk));
    if (p.layA == LAY_KC && p.layB == LAY_KC) return launch<LAY_KC, LAY_KC>(p, vecA, vecB, kchunk, grid, st);
    if (p.layA == LAY_KC && p.layB == LAY_KS) return launch<LAY_KC, LAY_KS>(p, vecA, vecB, kchunk, grid, st);
    if (p.layA == LAY_KS && p.layB == LAY_KC) return launch<LAY_KS, LAY_KC>(p, vecA, vecB, kchunk, grid, st);
    return launch<LAY_KS, LAY_KS>(p, vecA, vecB, kchunk, grid, st);
}
}  // namespace gg
