// Generic batched / split-K GEMM on the CDNA4 bf16 matrix cores (v_mfma_f32_32x32x16_bf16), fp32
// accumulation, fp32 operands in HBM converted to bf16 (round-to-nearest-even) while they are staged
// into LDS.  "Perf mode" twin of gemm_f32.hip: same GemmP contract, same 128x128 tile / 4-wave
// decomposition, same C/D fragment map and epilogue; what changes is the K step (64), the LDS image
// (bf16, [128 rows][64+8] so a lane's 16-byte fragment read is conflict-free) and the staging:
//   LAY_KC operand ([rows][K]): a thread loads 8 consecutive k (2 x float4) of one row, packs 8 bf16
//                               and writes one 16-byte LDS chunk;
//   LAY_KS operand ([K][rows]): a thread loads the same 4 rows at 8 consecutive k (8 x float4, each a
//                               coalesced 1 KiB wave access), transposes in registers and writes four
//                               16-byte chunks - the LDS image is always K-contiguous, so the MFMA
//                               fragment (8 consecutive k per lane: A[r][8h+j] / B[8h+j][c]) is one
//                               ds_read_b128 for either layout.
#include "gg_common.h"

namespace gg {

namespace {
constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int LDT = BK + 8;                 // bf16 elements per LDS row (144 B)
constexpr int TILE_ELEMS = BM * LDT;        // 9216 bf16 = 18,432 B
constexpr int SMEM_BYTES = 4 * TILE_ELEMS * 2;   // A,B x 2 buffers = 73,728 B

struct Flags {
    int vecA, vecB, vecFilm;
    int kchunk;
};

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// ---- global -> registers: 8 float4 per thread per operand ------------------------------------------
// LAY_KC: slot i = chunk (row = f>>3, c8 = f&7), f = tid + 256*(i>>1), half (i&1): k = k0 + 8*c8 + 4*(i&1)
// LAY_KS: slot j = k row  k0 + 8*(tid>>5) + j, rows row0 + 4*(tid&31) .. +3
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// bf16-stored K-strided operand (weight-gradient products of bf16 branch gradients / activations): 8-byte loads of
// 4 adjacent rows at one k, expanded exactly to fp32 so the staging below is shared with the fp32 path
// All loaders return a bit mask of the slots that lie outside the operand.  The aligned paths issue their 8 loads
// UNCONDITIONALLY from clamped addresses and leave blanking (and the bf16 expansion) to finish_tile(), which runs after the
// MFMA block of the current tile: a guard around each load made the compiler wait for every load where it was issued,
// which serialised the loads and put them in front of the MFMAs they were meant to hide behind.
__device__ __forceinline__ unsigned load_tile_ks_bf16(const __bf16* __restrict__ base, long ld, int rows_total, int row0, int k0,
                                                       int kend, f32x4 (&r)[8], int tid) {
    unsigned blank = 0;
    const int gr = row0 + 4 * (tid & 31);
    const __bf16* src = base + (gr < rows_total ? gr : 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gk = k0 + 8 * (tid >> 5) + i;
        const u32x2 w = *reinterpret_cast<const u32x2*>(src + (long)min(gk, kend - 1) * ld);
        r[i] = __builtin_bit_cast(f32x4, u32x4{w[0], w[1], 0u, 0u});      // raw bits: expanded by finish_tile
        if (!(gk < kend && gr < rows_total)) blank |= 1u << i;
    }
    return blank;
}

template <int LAY>
__device__ __forceinline__ unsigned load_tile(const float* __restrict__ base, long ld, int rows_total, int row0, int k0,
                                              int kend, bool vec, f32x4 (&r)[8], int tid) {
    unsigned blank = 0;
    if (vec) {
        if (LAY == LAY_KC) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = tid + NTHREADS * (i >> 1);
                const int row = f >> 3, c8 = f & 7;
                const int gr = row0 + row, gk = k0 + 8 * c8 + 4 * (i & 1);
                r[i] = *reinterpret_cast<const f32x4*>(base + (long)min(gr, rows_total - 1) * ld + min(gk, kend - 4));
                if (!(gr < rows_total && gk < kend)) blank |= 1u << i;
            }
        } else {
            const int gr = row0 + 4 * (tid & 31);
            const float* src = base + min(gr, rows_total - 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gk = k0 + 8 * (tid >> 5) + i;
                r[i] = *reinterpret_cast<const f32x4*>(src + (long)min(gk, kend - 1) * ld);
                if (!(gk < kend && gr < rows_total)) blank |= 1u << i;
            }
        }
        return blank;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (LAY == LAY_KC) {
            const int f = tid + NTHREADS * (i >> 1);
            const int row = f >> 3, c8 = f & 7;
            const int gr = row0 + row, gk = k0 + 8 * c8 + 4 * (i & 1);
            if (gr < rows_total) {
                const float* src = base + (long)gr * ld + gk;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (gk + j < kend) v[j] = src[j];
            }
        } else {
            const int gk = k0 + 8 * (tid >> 5) + i, gr = row0 + 4 * (tid & 31);
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
// blank the out-of-range slots; RAW16: expand the two dwords of 4 bf16 values to fp32 first
template <bool RAW16>
__device__ __forceinline__ void finish_tile(f32x4 (&r)[8], unsigned blank) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (RAW16) {
            const u32x4 u = __builtin_bit_cast(u32x4, r[i]);
            const unsigned w0 = u[0], w1 = u[1];
            r[i] = f32x4{__builtin_bit_cast(float, w0 << 16), __builtin_bit_cast(float, w0 & 0xffff0000u),
                         __builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)};
        }
        if ((blank >> i) & 1u) r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// FiLM on the A operand (LAY_KC): a' = gamma[g][k]*a + beta[g][k], g = m / group
__device__ __forceinline__ void film_tile(const GemmP& p, int row0, int k0, int kend, bool vec, f32x4 (&r)[8], int tid) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = tid + NTHREADS * (i >> 1);
        const int row = f >> 3, c8 = f & 7;
        const int gr = row0 + row, gk = k0 + 8 * c8 + 4 * (i & 1);
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
__device__ __forceinline__ void store_tile(__bf16* tile, const f32x4 (&r)[8], int tid) {
    if (LAY == LAY_KC) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int f = tid + NTHREADS * c;
            const int row = f >> 3, c8 = f & 7;
            const f32x4 lo = r[2 * c], hi = r[2 * c + 1];
            u32x4 w = {pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
            *reinterpret_cast<u32x4*>(tile + row * LDT + 8 * c8) = w;
        }
    } else {
        const int kg = tid >> 5, c4 = tid & 31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 w = {pack2(r[0][q], r[1][q]), pack2(r[2][q], r[3][q]), pack2(r[4][q], r[5][q]), pack2(r[6][q], r[7][q])};
            *reinterpret_cast<u32x4*>(tile + (4 * c4 + q) * LDT + 8 * kg) = w;
        }
    }
}

template <int LA, int LB, bool AB = false, bool BB = false>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(const GemmP p, const Flags fl) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem_bf16[];
    __bf16* As = smem_bf16;
    __bf16* Bs = smem_bf16 + 2 * TILE_ELEMS;

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
    // bf16-stored operands: p.A / p.B point at bf16 data, offsets are in elements
    const __bf16* Ab = reinterpret_cast<const __bf16*>(p.A) + zo * p.sAo + zi * p.sAi;
    const __bf16* Bb = reinterpret_cast<const __bf16*>(p.B) + zo * p.sBo + zi * p.sBi;

    const int kbeg = sk * fl.kchunk;
    const int kend = min(p.K, kbeg + fl.kchunk);
    const int nkt = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;

    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    auto loadA = [&](int k0, f32x4 (&ra)[8]) -> unsigned {
        if constexpr (AB) return load_tile_ks_bf16(Ab, p.lda, p.M, m0, k0, kend, ra, tid);
        else return load_tile<LA>(A, p.lda, p.M, m0, k0, kend, fl.vecA, ra, tid);
    };
    auto loadB = [&](int k0, f32x4 (&rb)[8]) -> unsigned {
        if constexpr (BB) return load_tile_ks_bf16(Bb, p.ldb, p.N, n0, k0, kend, rb, tid);
        else return load_tile<LB>(B, p.ldb, p.N, n0, k0, kend, fl.vecB, rb, tid);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    f32x4 ra[8], rb[8];
    unsigned ba = 0, bb = 0;
    if (nkt > 0) {
        ba = loadA(kbeg, ra);
        bb = loadB(kbeg, rb);
        finish_tile<AB>(ra, ba);
        if (LA == LAY_KC && p.film_gamma) film_tile(p, m0, kbeg, kend, fl.vecFilm, ra, tid);
        finish_tile<BB>(rb, bb);
        store_tile<LA>(As, ra, tid);
        store_tile<LB>(Bs, rb, tid);
    }
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1 < nkt);
        if (more) {
            const int k0 = kbeg + (kt + 1) * BK;
            ba = loadA(k0, ra);
            bb = loadB(k0, rb);
        }
        const __bf16* at = As + cur * TILE_ELEMS;
        const __bf16* bt = Bs + cur * TILE_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                a[mi] = *reinterpret_cast<const bf16x8*>(at + (wm * 64 + mi * 32 + r) * LDT + kk + 8 * h);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                b[ni] = *reinterpret_cast<const bf16x8*>(bt + (wn * 64 + ni * 32 + r) * LDT + kk + 8 * h);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (more) {
            finish_tile<AB>(ra, ba);
            if (LA == LAY_KC && p.film_gamma) film_tile(p, m0, kbeg + (kt + 1) * BK, kend, fl.vecFilm, ra, tid);
            finish_tile<BB>(rb, bb);
            store_tile<LA>(As + (cur ^ 1) * TILE_ELEMS, ra, tid);
            store_tile<LB>(Bs + (cur ^ 1) * TILE_ELEMS, rb, tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue (identical contract to gemm_f32.hip) ----------------------------------------------
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

template <int LA, int LB, bool AB = false, bool BB = false>
int launch(const GemmP& p, const Flags& fl, dim3 grid, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<LA, LB, AB, BB>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<LA, LB, AB, BB>), grid, dim3(NTHREADS), SMEM_BYTES, st, p, fl);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace

int gemm_bf16(const GemmP& p, hipStream_t st) {
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
    if (p.a_bf16 || p.b_bf16) {
        GG_REQUIRE(p.layA == LAY_KS && p.layB == LAY_KS, "bf16-stored operands are supported for (K-strided, K-strided) products only");
        GG_REQUIRE(p.lda % 4 == 0 && p.ldb % 4 == 0 && p.M % 4 == 0 && p.N % 4 == 0 && (reinterpret_cast<uintptr_t>(p.A) & 7) == 0 &&
                       (reinterpret_cast<uintptr_t>(p.B) & 7) == 0 && strA && strB, "bf16-stored operands need 4-element alignment");
        if (p.a_bf16 && p.b_bf16) return launch<LAY_KS, LAY_KS, true, true>(p, fl, grid, st);
        if (p.a_bf16) return launch<LAY_KS, LAY_KS, true, false>(p, fl, grid, st);
        return launch<LAY_KS, LAY_KS, false, true>(p, fl, grid, st);
    }
    if (p.layA == LAY_KC && p.layB == LAY_KC) return launch<LAY_KC, LAY_KC>(p, fl, grid, st);
    if (p.layA == LAY_KC && p.layB == LAY_KS) return launch<LAY_KC, LAY_KS>(p, fl, grid, st);
    if (p.layA == LAY_KS && p.layB == LAY_KC) return launch<LAY_KS, LAY_KC>(p, fl, grid, st);
    return launch<LAY_KS, LAY_KS>(p, fl, grid, st);
}

}  // namespace gg
