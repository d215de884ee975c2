// Generic batched / split-K GEMM on the CDNA4 fp32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
// Exact-fp32 path of the engine ("parity mode"): every Linear / attention product of the hot path
//  and its two backward products go through this one kernel, so one
// set of fragment maps has to be right.  Tiling is wave64-native: a 256-thread workgroup (4 waves,
// one per SIMD) owns a 128x128 C tile, each wave a 64x64 quadrant = 2x2 MFMA 32x32 accumulators
// (64 accumulator VGPRs).  K is consumed 32 at a time through double-buffered LDS tiles that are
// register-prefetched one tile ahead.
//
// Operand layouts: an operand is either K-contiguous ([rows][K], LAY_KC) or K-strided ([K][rows],
// LAY_KS).  Because the f32 MFMA takes ONE float per lane per operand, no transpose is ever needed:
//   LAY_KC tile lives in LDS as [128][32+4] and lane (r,h) reads float4 [row r][kk+4h .. kk+4h+3],
//   LAY_KS tile lives in LDS as [32][128+4] and lane (r,h) reads 4 scalars [kk+4h+e][row r].
// Both give MFMA step e of an 8-deep group the reduction indices { kk+e (lanes 0-31), kk+4+e (lanes
// 32-63) } - any permutation of k is fine as long as A and B agree.
//
// C/D fragment map (guide section 3): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include "gg_common.h"

namespace gg {

namespace {
constexpr int BM = 128, BN = 128, BK = 32, NTHREADS = 256;
constexpr int LDKC = BK + 4;    // floats per LDS row of a K-contiguous tile  (144 B, 16-B aligned)
constexpr int LDKS = BM + 4;    // floats per LDS row of a K-strided tile     (528 B, 16-B aligned)
constexpr int TILE_FLOATS = BM * LDKC;   // 4608 >= BK*LDKS = 4224
constexpr int SMEM_BYTES = 4 * TILE_FLOATS * (int)sizeof(float);   // A,B x 2 buffers = 73,728 B

struct Flags {
    int vecA, vecB, vecFilm;
    int kchunk;
};

// ---- global -> register staging of one 128 x 32 operand tile (4 float4 per thread) ---------------
// Returns a bit mask of the slots outside the operand.  The aligned path loads UNCONDITIONALLY from clamped addresses and
// the caller blanks those slots (blank_tile) after the MFMA block of the current tile - see gemm_bf16.hip: a guard
// around each load made the compiler wait for every load where it was issued.
template <int LAY>
__device__ __forceinline__ unsigned load_tile(const float* __restrict__ base, long ld, int rows_total, int row0,
                                              int k0, int kend, bool vec, f32x4 (&r)[4], int tid) {
    if (vec) {
        unsigned blank = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + NTHREADS * i;
            if (LAY == LAY_KC) {
                const int row = f >> 3, c4 = f & 7;
                const int gr = row0 + row, gk = k0 + 4 * c4;
                r[i] = *reinterpret_cast<const f32x4*>(base + (long)min(gr, rows_total - 1) * ld + min(gk, kend - 4));
                if (!(gr < rows_total && gk < kend)) blank |= 1u << i;
            } else {
                const int k = f >> 5, c4 = f & 31;
                const int gk = k0 + k, gr = row0 + 4 * c4;
                r[i] = *reinterpret_cast<const f32x4*>(base + (long)min(gk, kend - 1) * ld + min(gr, rows_total - 4));
                if (!(gk < kend && gr < rows_total)) blank |= 1u << i;
            }
        }
        return blank;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + NTHREADS * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (LAY == LAY_KC) {
            const int row = f >> 3, c4 = f & 7;
            const int gr = row0 + row, gk = k0 + 4 * c4;
            if (gr < rows_total) {
                const float* src = base + (long)gr * ld + gk;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (gk + j < kend) v[j] = src[j];
            }
        } else {
            const int k = f >> 5, c4 = f & 31;
            const int gk = k0 + k, gr = row0 + 4 * c4;
            if (gk < kend) {
                const float* src = base + (long)gk * ld + gr;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (gr + j < rows_total) v[j] = src[j];
            }
        }
        r[i] = v;
    }
    return 0;
}
__device__ __forceinline__ void blank_tile(f32x4 (&r)[4], unsigned blank) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if ((blank >> i) & 1u) r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// FiLM on the A operand (LAY_KC):  a' = gamma[g][k] * a + beta[g][k],  g = m / group
__device__ __forceinline__ void film_tile(const GemmP& p, int row0, int k0, int kend, bool vec, f32x4 (&r)[4], int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + NTHREADS * i;
        const int row = f >> 3, c4 = f & 7;
        const int gr = row0 + row, gk = k0 + 4 * c4;
        if (gr < p.M && gk < kend) {
            const long off = (long)(gr / p.film_group) * p.film_ld + gk;
            if (vec) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(p.film_gamma + off);
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.film_beta + off);
                r[i] = g * r[i] + b;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (gk + j < kend) r[i][j] = p.film_gamma[off + j] * r[i][j] + p.film_beta[off + j];
            }
        }
    }
}

template <int LAY>
__device__ __forceinline__ void store_tile(float* tile, const f32x4 (&r)[4], int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + NTHREADS * i;
        if (LAY == LAY_KC) {
            const int row = f >> 3, c4 = f & 7;
            *reinterpret_cast<f32x4*>(tile + row * LDKC + 4 * c4) = r[i];
        } else {
            const int k = f >> 5, c4 = f & 31;
            *reinterpret_cast<f32x4*>(tile + k * LDKS + 4 * c4) = r[i];
        }
    }
}

// fragment for the 8-deep k group starting at kk: out[e] feeds MFMA step e (k = kk + 4h + e)
template <int LAY>
__device__ __forceinline__ void read_frag(const float* tile, int row, int kk, int h, float (&out)[4]) {
    if (LAY == LAY_KC) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * LDKC + kk + 4 * h);
        out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = tile[(kk + 4 * h + e) * LDKS + row];
    }
}

template <int LA, int LB>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(const GemmP p, const Flags fl) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][TILE_FLOATS]
    float* Bs = smem + 2 * TILE_FLOATS;     // [2][TILE_FLOATS]

    const int tid = threadIdx.x;
    const int tiles_n = (p.N + BN - 1) / BN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so block
    // b and b+8 share an L2.  Give one XCD ALL N-tiles of a row panel back to back: the A panel is
    // then fetched from HBM once and re-read from that XCD's L2 (speed only; any placement is correct).
    int tile_m, tile_n;
    {
        const int tiles_m = (p.M + BM - 1) / BM;
        const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3;
        const int full = (tiles_m / 8) * 8;              // row panels covered by the 8-way interleave
        const int pm = (local / tiles_n) * 8 + xcd;
        if (pm < full) {
            tile_m = pm;
            tile_n = local % tiles_n;
        } else {                                         // tail panels (tiles_m % 8): plain order
            const int t = bid - full * tiles_n;
            tile_m = full + t / tiles_n;
            tile_n = t % tiles_n;
        }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int z = blockIdx.z;
    const int bz = z / p.splitk, sk = z % p.splitk;
    const int zo = bz / p.batch_inner, zi = bz % p.batch_inner;
    const float* A = p.A + zo * p.sAo + zi * p.sAi;
    const float* B = p.B + zo * p.sBo + zi * p.sBi;
    float* C = p.C + zo * p.sCo + zi * p.sCi;

    const int kbeg = sk * fl.kchunk;
    const int kend = min(p.K, kbeg + fl.kchunk);
    const int nkt = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;

    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    f32x4 ra[4], rb[4];
    unsigned ba = 0, bb = 0;
    if (nkt > 0) {
        ba = load_tile<LA>(A, p.lda, p.M, m0, kbeg, kend, fl.vecA, ra, tid);
        bb = load_tile<LB>(B, p.ldb, p.N, n0, kbeg, kend, fl.vecB, rb, tid);
        blank_tile(ra, ba);
        if (LA == LAY_KC && p.film_gamma) film_tile(p, m0, kbeg, kend, fl.vecFilm, ra, tid);
        blank_tile(rb, bb);
        store_tile<LA>(As, ra, tid);
        store_tile<LB>(Bs, rb, tid);
    }
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1 < nkt);
        if (more) {
            const int k0 = kbeg + (kt + 1) * BK;
            ba = load_tile<LA>(A, p.lda, p.M, m0, k0, kend, fl.vecA, ra, tid);
            bb = load_tile<LB>(B, p.ldb, p.N, n0, k0, kend, fl.vecB, rb, tid);
        }
        const float* at = As + cur * TILE_FLOATS;
        const float* bt = Bs + cur * TILE_FLOATS;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 8) {
            float a[2][4], b[2][4];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) read_frag<LA>(at, wm * 64 + mi * 32 + r, kk, h, a[mi]);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) read_frag<LB>(bt, wn * 64 + ni * 32 + r, kk, h, b[ni]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][e], b[ni][e], acc[mi][ni], 0, 0, 0);
        }
        if (more) {
            blank_tile(ra, ba);
            if (LA == LAY_KC && p.film_gamma) film_tile(p, m0, kbeg + (kt + 1) * BK, kend, fl.vecFilm, ra, tid);
            blank_tile(rb, bb);
            store_tile<LA>(As + (cur ^ 1) * TILE_FLOATS, ra, tid);
            store_tile<LB>(Bs + (cur ^ 1) * TILE_FLOATS, rb, tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue -------------------------------------------------------------------------------
    const bool atomic = p.splitk > 1;
    if (atomic && nkt == 0) return;
    const uint8_t* cmask = p.colmask ? p.colmask + (long)(p.colmask_mod > 0 ? zo % p.colmask_mod : zo) * p.colmask_stride : nullptr;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + r;
            if (col >= p.N) continue;
            const float bias = (p.bias && sk == 0) ? p.bias[col] : 0.f;
            const bool masked = cmask && cmask[col];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = m0 + wm * 64 + mi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row >= p.M) continue;
                const long crow = p.c_row_group ? (long)row + row / p.c_row_group + 1 : (long)row;
                float* cp = C + crow * p.ldc + col;
                float v = p.alpha * acc[mi][ni][i] + bias;
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
    }
}

inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int LA, int LB>
int launch(const GemmP& p, const Flags& fl, dim3 grid, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<LA, LB>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<LA, LB>), grid, dim3(NTHREADS), SMEM_BYTES, st, p, fl);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace

int gemm_f32(const GemmP& p, hipStream_t st) {
    GG_REQUIRE(p.A && p.B && p.C, "null operand");
    GG_REQUIRE(p.M > 0 && p.N > 0 && p.K >= 0, "bad dims");
    GG_REQUIRE(p.batch >= 1 && p.batch_inner >= 1 && p.batch % p.batch_inner == 0, "bad batch");
    GG_REQUIRE(p.splitk >= 1, "bad splitk");
    GG_REQUIRE(!(p.splitk > 1 && (p.act != ACT_NONE || p.colmask)), "split-K epilogue must be linear");
    GG_REQUIRE(!(p.film_gamma && p.layA != LAY_KC), "FiLM transform needs a K-contiguous A");
    GG_REQUIRE((long)p.batch * p.splitk <= 65535, "grid.z overflow");
    Flags fl;
    const bool strA = (p.sAo % 4 == 0) && (p.sAi % 4 == 0), strB = (p.sBo % 4 == 0) && (p.sBi % 4 == 0);
    fl.vecA = aligned16(p.A) && p.lda % 4 == 0 && strA && ((p.layA == LAY_KC) ? p.K % 4 == 0 : p.M % 4 == 0);
    fl.vecB = aligned16(p.B) && p.ldb % 4 == 0 && strB && ((p.layB == LAY_KC) ? p.K % 4 == 0 : p.N % 4 == 0);
    fl.vecFilm = p.film_gamma && aligned16(p.film_gamma) && aligned16(p.film_beta) && p.film_ld % 4 == 0 && p.K % 4 == 0;
    int kchunk = (p.K + p.splitk - 1) / p.splitk;
    kchunk = (kchunk + BK - 1) / BK * BK;
    fl.kchunk = kchunk > 0 ? kchunk : BK;
    const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    GG_REQUIRE(tiles <= 2147483647L, "grid.x overflow");
    dim3 grid((unsigned)tiles, 1, (unsigned)(p.batch * p.splitk));
    if (p.layA == LAY_KC && p.layB == LAY_KC) return launch<LAY_KC, LAY_KC>(p, fl, grid, st);
    if (p.layA == LAY_KC && p.layB == LAY_KS) return launch<LAY_KC, LAY_KS>(p, fl, grid, st);
    if (p.layA == LAY_KS && p.layB == LAY_KC) return launch<LAY_KS, LAY_KC>(p, fl, grid, st);
    return launch<LAY_KS, LAY_KS>(p, fl, grid, st);
}

}  // namespace gg
