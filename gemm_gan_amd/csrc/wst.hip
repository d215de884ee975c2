// Weight-stationary Linear + dropout + residual + LayerNorm for the encoder width (N = 256 output features).
//
//   r = res + drop(X W^T + bias) ;  y = LayerNorm(r) * gamma + beta         (out-proj / FFN2 of an encoder layer, forward)
//
// The token-on-lane kernels of tlin.hip keep the ACTIVATIONS of a wave's tokens in registers and stream the weights through
// LDS: 128 KB (K = 256) or 256 KB (K = 512) of weights per 64 tokens, one barrier and one L2 round trip per 32-feature chunk
// - the waves spend 60 % of their time waiting for those chunks (profiles/r02_pmc_mfma.json) and the kernel sits at 0.4 of
// the HBM roofline although it moves only activations through HBM.  Here the roles are swapped:
//   * a PERSISTENT workgroup loads its share of W once, as MFMA A fragments that stay in registers for the whole launch
//     (wave w owns output features [N/NW * w, N/NW * (w+1)): RT row tiles of 32, all K);
//   * token tiles of 32 rows stream through a double-buffered LDS image of X (the only LDS traffic besides two floats per
//     token and wave), every wave multiplies the whole tile against its resident rows: RT * K/16 back-to-back MFMAs;
//   * accumulators have features in registers and tokens on lanes (rows fed in the permuted order of tlin.hip, so a lane
//     owns 16 consecutive features: 64-byte runs for residual loads and stores); LayerNorm needs all 256 features of a
//     token, which live in NW waves: each wave contributes (sum, sum of squares) of its slice through LDS.
// K = 256: 4 waves x 2 row tiles (128 weight registers per lane), two workgroups per CU; K = 512: 8 waves x 1 row tile.
// Same arithmetic, same dropout stream (element index = token * drop_ld + feature) as tlin_res16_kernel.
//
// The same structure with other epilogues (EPI): ACC  y += X W^T (fp32; the data-gradient Linears dx1 += dh W1, N = 256,
// K = 512), ACT  y = drop(relu(X W^T + b)) -> bf16 (FFN1, N = 512, K = 256, fp32 X converted while staging), MASK
// y = (X W^T) * [ref > 0] * scale -> bf16 (dh = (dres W2) gated by the stored hidden activations).
#include "kernels.h"
#include "drop_rng.h"
#include "fp8_util.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace gg {
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr float LN_EPS = 1e-5f;
constexpr int TT = 32;                                  // tokens per tile

__device__ __forceinline__ int a_row_of_lane(int r) { return 16 * ((r >> 2) & 1) + (r & 3) + 4 * (r >> 3); }

enum { EPI_LN = 0, EPI_ACC = 1, EPI_ACT = 2, EPI_MASK = 3, EPI_LNB = 4 };

// Column sums over the 32 token lanes of a half-wave for 16 per-lane values (one feature each): a reduce-scatter butterfly -
// every step halves the number of values a lane carries - so 16 shuffles instead of 80.  Returns, in every lane, the total of value
// index 8*b4 + 4*b3 + 2*b2 + b1 (b_k = bit k of the lane's token index c) over the 32 lanes of its half.
__device__ __forceinline__ float colsum16(float (&v)[16], int c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool up = c & 16;
        const float send = up ? v[i] : v[i + 8], keep = up ? v[i + 8] : v[i];
        v[i] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool up = c & 8;
        const float send = up ? v[i] : v[i + 4], keep = up ? v[i + 4] : v[i];
        v[i] = keep + __shfl_xor(send, 8, 64);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bool up = c & 4;
        const float send = up ? v[i] : v[i + 2], keep = up ? v[i + 2] : v[i];
        v[i] = keep + __shfl_xor(send, 4, 64);
    }
    {
        const bool up = c & 2;
        const float send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
        v[0] = keep + __shfl_xor(send, 2, 64);
    }
    return v[0] + __shfl_xor(v[0], 1, 64);
}
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// F8: e4m3 operands on v_mfma_scale_f32_32x32x64_f8f6f4 (see tlin.hip "fp8 operand path"): W from the e4m3 shadow (half the
// weight registers: N = 768 fits), X quantised to e4m3 while staging, a lane's fragment = 32 consecutive k (32 bytes).
// GROUPS > 1: the output has GROUPS * N columns (the packed QKV projection: 3 x 256) and the grid holds GROUPS workgroups per
// token-tile owner, each stationary on its own N rows of W and streaming the SAME tiles; the block index is decoded so that
// the groups of one owner sit on one XCD (ids that differ by multiples of 8) and re-read the tile from that XCD's L2.
// PF: the epilogue's per-token operands (EPI_LN: the bf16 residual rows; EPI_MASK: the gate reference rows) are requested ONE TILE
// AHEAD into a second register set.  Requested at the start of their own tile they have only the MFMA phase (~0.5 - 0.9 us) to
// arrive, and the waves then sit out the rest of an HBM round trip in the epilogue: 43 - 57 % of the wave time of these kernels
// was s_waitcnt (profiles/r03_pmc_mfma.json).  bf16 operands only (8 registers per row tile and set).
template <int NW, int RT, int KS, bool XB = true, int EPI = EPI_LN, bool F8 = false, int GROUPS = 1, bool PF = false>
__global__ __launch_bounds__(64 * NW, (NW == 4 && KS >= 48) ? 1 : 2) void wst_ln_kernel(const TlinP p) {
    const DropKey dkey = drop_live(p.drop);
    const int grp = GROUPS > 1 ? (int)(blockIdx.x >> 3) % GROUPS : 0;
    const long owner = GROUPS > 1 ? (long)(blockIdx.x / (8 * GROUPS)) * 8 + (blockIdx.x & 7) : (long)blockIdx.x;
    const long nown = GROUPS > 1 ? (long)gridDim.x / GROUPS : (long)gridDim.x;
    constexpr int K = 16 * KS, N = 32 * NW * RT, NTH = 64 * NW;
    constexpr int ESZ = F8 ? 1 : 2;                     // bytes per operand element in LDS
    constexpr int XLD = F8 ? K + 16 : K + 8;            // elements per LDS row: 4 banks per row step, conflict-free 16-byte reads
    constexpr int KS8 = K / 64;                         // k-steps of the fp8 instruction
    constexpr int PPR = XB ? K / 8 : K / 4;             // 16-byte pieces per row of X as stored in memory
    constexpr int XP = TT * PPR / NTH;                  // pieces per thread and tile
    static_assert(TT * PPR % NTH == 0, "tile staging must divide evenly");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* const Xs = reinterpret_cast<__bf16*>(smem_raw);                        // [2][TT * XLD]
    unsigned char* const Xs8 = smem_raw;                                           // the same image in fp8 mode (bytes)
    float* const Red = reinterpret_cast<float*>(smem_raw + 2 * TT * XLD * ESZ);    // [2][NW][TT][2]  (sum, sum of squares)
    float* const Ps = Red + 2 * NW * TT * 2;                                       // bias | gamma | beta  [3][N]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    const int f0 = 32 * RT * wave;                      // first output feature of this wave (inside its column group)
    const int gcol = grp * (32 * NW * RT);              // first output column of this workgroup's group
    const long ntiles = (p.M + TT - 1) / TT;
    const int last_tok = (int)p.M - 1;

    // resident weights: A fragment of k-step s for row tile rt = W[f0 + 32 rt + a_row_of_lane(c)][16 s + 8 h .. + 7]
    bf16x8 wA[F8 ? 1 : RT][F8 ? 1 : KS];
    i32x8 wA8[F8 ? RT : 1][F8 ? KS8 : 1];
    if constexpr (F8) {
        const unsigned char* Wp = reinterpret_cast<const unsigned char*>(p.W);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned char* wr = Wp + (long)(f0 + 32 * rt + a_row_of_lane(c)) * p.ldw + 32 * h;
#pragma unroll
            for (int s = 0; s < KS8; ++s) {
                const u32x4 lo = *reinterpret_cast<const u32x4*>(wr + 64 * s), hi = *reinterpret_cast<const u32x4*>(wr + 64 * s + 16);
                wA8[rt][s] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            }
        }
    } else {
        const __bf16* Wp = reinterpret_cast<const __bf16*>(p.W);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const __bf16* wr = Wp + (long)(gcol + f0 + 32 * rt + a_row_of_lane(c)) * p.ldw + 8 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) wA[rt][s] = *reinterpret_cast<const bf16x8*>(wr + 16 * s);
        }
    }
    const int sc_w = F8 ? 127 - *p.w_exp : 0, sc_x = F8 ? 127 - p.x_exp : 0;
    const float xscale = F8 ? exp2i(p.x_exp) : 1.f;
    for (int i = tid; i < N; i += NTH) {
        Ps[i] = p.bias ? p.bias[gcol + i] : 0.f;
        if constexpr (EPI == EPI_LN) {
            Ps[N + i] = p.ln_g[i];
            Ps[2 * N + i] = p.ln_b[i];
        }
        if constexpr (EPI == EPI_LNB) Ps[N + i] = p.ln_g[i];
    }

    // staging of one X tile: thread -> XP pieces (row = f / PPR, piece = f % PPR), rows past the end clamped
    const unsigned char* Xp = reinterpret_cast<const unsigned char*>(p.X);
    u32x4 xr[XP];
    auto load_x = [&](long tile) {
        const int tok0 = (int)(tile * TT);
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int f = tid + NTH * i, row = f / PPR, pc = f % PPR;
            xr[i] = *reinterpret_cast<const u32x4*>(Xp + (long)min(tok0 + row, last_tok) * p.ldx * (XB ? 2 : 4) + 16 * pc);
        }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int f = tid + NTH * i, row = f / PPR, pc = f % PPR;
            if constexpr (F8 && XB) {          // 8 bf16 -> 8 e4m3
                float f8[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f8[2 * j] = __builtin_bit_cast(float, xr[i][j] << 16);
                    f8[2 * j + 1] = __builtin_bit_cast(float, xr[i][j] & 0xffff0000u);
                }
                *reinterpret_cast<u32x2*>(&Xs8[buf * TT * XLD + row * XLD + 8 * pc]) =
                    u32x2{cvt4_fp8(f8[0], f8[1], f8[2], f8[3], xscale), cvt4_fp8(f8[4], f8[5], f8[6], f8[7], xscale)};
            } else if constexpr (F8) {         // 4 fp32 -> 4 e4m3
                const f32x4 v = __builtin_bit_cast(f32x4, xr[i]);
                *reinterpret_cast<unsigned*>(&Xs8[buf * TT * XLD + row * XLD + 4 * pc]) = cvt4_fp8(v[0], v[1], v[2], v[3], xscale);
            } else if constexpr (XB) {
                *reinterpret_cast<u32x4*>(&Xs[buf * TT * XLD + row * XLD + 8 * pc]) = xr[i];
            } else {        // fp32 in memory, bf16 in LDS
                const f32x4 v = __builtin_bit_cast(f32x4, xr[i]);
                *reinterpret_cast<u32x2*>(&Xs[buf * TT * XLD + row * XLD + 4 * pc]) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
            }
        }
    };

    long tile = owner;
    if (tile >= ntiles) return;
    // Every load of the tile pipeline is UNCONDITIONAL (tile indices clamped to the last tile; rows are clamped inside load_x):
    // with `if (next tile exists) load` the compiler's wait-count pass sees paths with and without loads in flight and falls back
    // to s_waitcnt vmcnt(0) before the epilogue - which waits for the rows of the tile AFTER next that were requested a moment
    // before, i.e. the register prefetch collapses to nothing (found in the ISA; 43 - 57 % of these kernels' wave time was waits).
    load_x(tile);
    store_x(0);
    load_x(min(tile + nown, ntiles - 1));
    const float ksd = p.drop.p > 0.f ? 1.f / (1.f - p.drop.p) : 1.f;
    const bool drop_on = p.drop.p > 0.f;

    float col_g = 0.f, col_b = 0.f, col_c = 0.f;        // EPI_LNB: this lane's column sums (feature f0 + 16 h + colsum16's index)
    static_assert(!PF || EPI == EPI_LN || EPI == EPI_MASK, "one-tile-ahead epilogue operands: LayerNorm residual / gate reference");
    u32x4 ecur[PF ? RT : 1][2], enxt[PF ? RT : 1][2];
    auto request_epi = [&](long tl, u32x4 (&dst)[PF ? RT : 1][2]) {
        const int tk = min((int)(tl * TT) + c, last_tok);
        const __bf16* src = EPI == EPI_LN ? reinterpret_cast<const __bf16*>(p.res) + (long)(tk % (int)p.res_rows) * p.ldres + f0 + 16 * h
                                          : reinterpret_cast<const __bf16*>(p.mask_ref) + (long)tk * p.ldref + f0 + 16 * h;
#pragma unroll
        for (int rt = 0; rt < (PF ? RT : 1); ++rt) {
            dst[rt][0] = *reinterpret_cast<const u32x4*>(src + 32 * rt);
            dst[rt][1] = *reinterpret_cast<const u32x4*>(src + 32 * rt + 8);
        }
    };
    if constexpr (PF) request_epi(tile, ecur);
    int buf = 0;
    for (; tile < ntiles; tile += nown, buf ^= 1) {
        const int tok = (int)(tile * TT) + c;
        const bool valid = tok <= last_tok;
        const int tokc = valid ? tok : last_tok;        // clamped lanes recompute the last row; their stores are predicated off
        const bool keep_y = p.y_rows < 0 || tokc < p.y_rows;
        __syncthreads();                                // X(tile) is in Xs[buf]; Red[buf] of two tiles ago has been consumed
        // epilogue operands of this tile (residual rows / previous output / gate reference): requested before the products
        f32x4 res[EPI == EPI_LN || EPI == EPI_ACC || EPI == EPI_LNB ? RT : 1][4];
        f32x4 rpre[EPI == EPI_LNB ? RT : 1][4];          // EPI_LNB: the pre-LayerNorm sums of the token (bf16 form: raw words in [0], [1],
                                                         // expanded only in the epilogue - a conversion here would wait for the load at once)
        float2 lstat = float2{0.f, 1.f};
        u32x4 mref[EPI == EPI_MASK ? RT : 1][2];
        if constexpr (PF) {
            request_epi(min(tile + nown, ntiles - 1), enxt);      // the NEXT tile's rows (the last tile re-requests its own)
        } else if constexpr (EPI == EPI_LN) {
            if (p.res_bf16) {
                const __bf16* resp = reinterpret_cast<const __bf16*>(p.res) + (long)(tokc % (int)p.res_rows) * p.ldres + f0 + 16 * h;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {          // 16 bf16 values in the first two of the four fp32 quads' registers
                    res[rt][0] = __builtin_bit_cast(f32x4, *reinterpret_cast<const u32x4*>(resp + 32 * rt));
                    res[rt][1] = __builtin_bit_cast(f32x4, *reinterpret_cast<const u32x4*>(resp + 32 * rt + 8));
                }
            } else {
                const float* resp = p.res + (long)(tokc % (int)p.res_rows) * p.ldres + f0 + 16 * h;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) res[rt][g] = *reinterpret_cast<const f32x4*>(resp + 32 * rt + 4 * g);
            }
        }
        if constexpr (EPI == EPI_LNB) {
            static_assert(EPI != EPI_LNB || (RT == 1 && GROUPS == 1 && !F8), "LayerNorm-backward epilogue: 8 waves x 32 features");
            if (p.res_bf16) {
                const __bf16* rp = reinterpret_cast<const __bf16*>(p.res) + (long)tokc * p.ldres + f0 + 16 * h;
                rpre[0][0] = __builtin_bit_cast(f32x4, *reinterpret_cast<const u32x4*>(rp));
                rpre[0][1] = __builtin_bit_cast(f32x4, *reinterpret_cast<const u32x4*>(rp + 8));
            } else {
                const float* rp = p.res + (long)tokc * p.ldres + f0 + 16 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) rpre[0][g] = *reinterpret_cast<const f32x4*>(rp + 4 * g);
            }
            lstat = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (long)tokc);
        }
        if constexpr (EPI == EPI_ACC || EPI == EPI_LNB) {
            const float* yo = reinterpret_cast<const float*>(p.Y) + (long)tokc * p.ldy + gcol + f0 + 16 * h;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int g = 0; g < 4; ++g) res[rt][g] = *reinterpret_cast<const f32x4*>(yo + 32 * rt + 4 * g);
        }
        if constexpr (EPI == EPI_MASK && !PF) {
            const __bf16* mr = reinterpret_cast<const __bf16*>(p.mask_ref) + (long)tokc * p.ldref + f0 + 16 * h;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                mref[rt][0] = *reinterpret_cast<const u32x4*>(mr + 32 * rt);
                mref[rt][1] = *reinterpret_cast<const u32x4*>(mr + 32 * rt + 8);
            }
        }
        // keep the epilogue-operand requests above the MFMA block: left to itself the scheduler sinks them below it (register
        // pressure), where they sit behind the NEXT tile's row requests and the epilogue waits for everything (vmcnt(0))
        if constexpr (EPI == EPI_ACC || PF) __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][i] = 0.f;
        if constexpr (F8) {
            const unsigned char* xb8 = Xs8 + buf * TT * XLD + c * XLD + 32 * h;
#pragma unroll
            for (int s = 0; s < KS8; ++s) {
                const i32x8 xf = lds_frag32(xb8 + 64 * s);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wA8[rt][s], xf, acc[rt], 0, 0, 0, sc_w, 0, sc_x);
            }
        } else {
            const __bf16* xb = Xs + buf * TT * XLD + c * XLD + 8 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xb + 16 * s);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA[rt][s], xf, acc[rt], 0, 0, 0);
            }
        }
        // the next tile's rows (already in registers) go to the other buffer, the tile after that is requested
        store_x(buf ^ 1);          // (after the last tile: rows nobody reads)
        load_x(min(tile + 2L * nown, ntiles - 1));
        if constexpr (EPI == EPI_LNB) {
            // ---- += then LayerNorm backward: dy = Y + acc; dr = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) -------------------
            const float mean = lstat.x, rstd = lstat.y;
            const float nmr = -mean * rstd;              // xhat = r * rstd + nmr, recomputed where it is needed (no second 16-register copy)
            if (p.res_bf16) {                            // expand the raw bf16 words (high groups first: [0], [1] are overwritten last)
                const u32x4 w0 = __builtin_bit_cast(u32x4, rpre[0][0]), w1 = __builtin_bit_cast(u32x4, rpre[0][1]);
#pragma unroll
                for (int g = 3; g >= 0; --g) {
                    const unsigned a = g < 2 ? w0[2 * (g & 1)] : w1[2 * (g & 1)], b = g < 2 ? w0[2 * (g & 1) + 1] : w1[2 * (g & 1) + 1];
                    rpre[0][g] = f32x4{__builtin_bit_cast(float, a << 16), __builtin_bit_cast(float, a & 0xffff0000u),
                                       __builtin_bit_cast(float, b << 16), __builtin_bit_cast(float, b & 0xffff0000u)};
                }
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 gw = *reinterpret_cast<const f32x4*>(&Ps[N + f0 + 16 * h + 4 * g]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * g + j;
                    acc[0][i] += res[0][g][j];              // dy
                    const float gy = acc[0][i] * gw[j];
                    s1 += gy;
                    s2 += gy * (rpre[0][g][j] * rstd + nmr);
                }
            }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            float* const red = Red + buf * NW * TT * 2;
            if (h == 0) *reinterpret_cast<float2*>(&red[(wave * TT + c) * 2]) = float2{s1, s2};
            __syncthreads();
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float2 v = *reinterpret_cast<const float2*>(&red[(w * TT + c) * 2]);
                t1 += v.x;
                t2 += v.y;
            }
            t1 *= 1.f / N;
            t2 *= 1.f / N;
            float* const drp = p.ln_y + (long)tokc * p.ldy + f0 + 16 * h;
            __bf16* const dbp = reinterpret_cast<__bf16*>(p.lnb_dres) + (long)tokc * p.ldy + f0 + 16 * h;
            const uint64_t dbase3 = (uint64_t)tokc * p.drop_ld + f0 + 16 * h;
            float cc[16];
            unsigned packed[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 gw = *reinterpret_cast<const f32x4*>(&Ps[N + f0 + 16 * h + 4 * g]);
                f32x4 v, vb;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (acc[0][4 * g + j] * gw[j] - t1 - (rpre[0][g][j] * rstd + nmr) * t2) * rstd;
                vb = v;
                if (drop_on) {
                    float f[4];
                    drop_factor4(dkey, dbase3 + 4 * g, ksd, f);
#pragma unroll
                    for (int j = 0; j < 4; ++j) vb[j] = v[j] * f[j];
                }
                if (valid) *reinterpret_cast<f32x4*>(drp + 4 * g) = v;
                packed[2 * g] = pack2(vb[0], vb[1]);
                packed[2 * g + 1] = pack2(vb[2], vb[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) cc[4 * g + j] = valid ? vb[j] : 0.f;
            }
            if (valid) {
                *reinterpret_cast<u32x4*>(dbp) = u32x4{packed[0], packed[1], packed[2], packed[3]};
                *reinterpret_cast<u32x4*>(dbp + 8) = u32x4{packed[4], packed[5], packed[6], packed[7]};
            }
            col_c += colsum16(cc, c);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) cc[4 * g + j] = valid ? acc[0][4 * g + j] * (rpre[0][g][j] * rstd + nmr) : 0.f;
            col_g += colsum16(cc, c);
#pragma unroll
            for (int i = 0; i < 16; ++i) cc[i] = valid ? acc[0][i] : 0.f;
            col_b += colsum16(cc, c);
        } else if constexpr (EPI != EPI_LN) {
            // ---- plain epilogues: no cross-wave step, one barrier per tile ------------------------------------------
            const uint64_t dbase2 = (uint64_t)tokc * p.drop_ld + gcol + f0 + 16 * h;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                unsigned packed[8];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[f0 + 32 * rt + 16 * h + 4 * g]);
                    f32x4 v = {acc[rt][4 * g], acc[rt][4 * g + 1], acc[rt][4 * g + 2], acc[rt][4 * g + 3]};
                    v += bb;
                    if constexpr (EPI == EPI_ACC) {
                        v += res[rt][g];
                        if (valid) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.Y) + (long)tokc * p.ldy + gcol + f0 + 16 * h + 32 * rt + 4 * g) = v;
                    }
                    if constexpr (EPI == EPI_ACT) {
                        if (p.act_relu) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                        }
                        if (drop_on) {
                            float f[4];
                            drop_factor4(dkey, dbase2 + 32 * rt + 4 * g, ksd, f);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] *= f[j];
                        }
                    }
                    if constexpr (EPI == EPI_MASK) {
                        const u32x4 mw = PF ? ecur[PF ? rt : 0][g >> 1] : mref[rt][g >> 1];
                        const unsigned w0 = mw[2 * (g & 1)], w1 = mw[2 * (g & 1) + 1];
                        const float m[4] = {__builtin_bit_cast(float, w0 << 16), __builtin_bit_cast(float, w0 & 0xffff0000u),
                                            __builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)};
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = m[j] > 0.f ? v[j] * p.mask_scale : 0.f;
                    }
                    packed[2 * g] = pack2(v[0], v[1]);
                    packed[2 * g + 1] = pack2(v[2], v[3]);
                }
                if constexpr (EPI == EPI_ACT || EPI == EPI_MASK) {
                    if (valid) {
                        __bf16* yo = reinterpret_cast<__bf16*>(p.Y) + (long)tokc * p.ldy + gcol + f0 + 16 * h + 32 * rt;
                        *reinterpret_cast<u32x4*>(yo) = u32x4{packed[0], packed[1], packed[2], packed[3]};
                        *reinterpret_cast<u32x4*>(yo + 8) = u32x4{packed[4], packed[5], packed[6], packed[7]};
                    }
                }
            }
        } else {
        // ---- epilogue 1: bias, dropout, residual; r stored; this wave's share of the row sums ------------------------
        float s1 = 0.f, s2 = 0.f;
        float* const yb = reinterpret_cast<float*>(p.Y) + (long)tokc * p.ldy + f0 + 16 * h;
        __bf16* const ybh = reinterpret_cast<__bf16*>(p.Y) + (long)tokc * p.ldy + f0 + 16 * h;
        const uint64_t dbase = (uint64_t)tokc * p.drop_ld + f0 + 16 * h;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            unsigned rpk[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[f0 + 32 * rt + 16 * h + 4 * g]);
                f32x4 v = {acc[rt][4 * g], acc[rt][4 * g + 1], acc[rt][4 * g + 2], acc[rt][4 * g + 3]};
                v += bb;
                if (drop_on) {
                    float f[4];
                    drop_factor4(dkey, dbase + 32 * rt + 4 * g, ksd, f);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= f[j];
                }
                if (PF || p.res_bf16) {
                    const u32x4 rw = PF ? ecur[PF ? rt : 0][g >> 1] : __builtin_bit_cast(u32x4, res[rt][g >> 1]);
                    const unsigned w0 = rw[2 * (g & 1)], w1 = rw[2 * (g & 1) + 1];
                    v += f32x4{__builtin_bit_cast(float, w0 << 16), __builtin_bit_cast(float, w0 & 0xffff0000u),
                               __builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)};
                } else {
                    v += res[rt][g];
                }
                if (!p.y_bf16 && valid && keep_y) *reinterpret_cast<f32x4*>(yb + 32 * rt + 4 * g) = v;
                rpk[2 * g] = pack2(v[0], v[1]);
                rpk[2 * g + 1] = pack2(v[2], v[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[rt][4 * g + j] = v[j];
                    s1 += v[j];
                    s2 += v[j] * v[j];
                }
            }
            if (p.y_bf16 && valid && keep_y) {          // the pre-LayerNorm sum kept for the backward pass in bf16 (statistics from the fp32 values)
                *reinterpret_cast<u32x4*>(ybh + 32 * rt) = u32x4{rpk[0], rpk[1], rpk[2], rpk[3]};
                *reinterpret_cast<u32x4*>(ybh + 32 * rt + 8) = u32x4{rpk[4], rpk[5], rpk[6], rpk[7]};
            }
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        float* const red = Red + buf * NW * TT * 2;
        if (h == 0) *reinterpret_cast<float2*>(&red[(wave * TT + c) * 2]) = float2{s1, s2};
        __syncthreads();
        // ---- epilogue 2: LayerNorm over the N features of the token (NW partial sums), y stored ------------------------
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float2 v = *reinterpret_cast<const float2*>(&red[(w * TT + c) * 2]);
            t1 += v.x;
            t2 += v.y;
        }
        const float mean = t1 * (1.f / N);
        const float var = fmaxf(t2 * (1.f / N) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + LN_EPS);
        float* const lb = p.ln_y + (long)tokc * p.ldy + f0 + 16 * h;
        __bf16* const lbh = reinterpret_cast<__bf16*>(p.ln_y) + (long)tokc * p.ldy + f0 + 16 * h;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            unsigned packed[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = f0 + 32 * rt + 16 * h + 4 * g;
                const f32x4 gg_ = *reinterpret_cast<const f32x4*>(&Ps[N + n]);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[2 * N + n]);
                f32x4 y;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = (acc[rt][4 * g + j] - mean) * rstd * gg_[j] + bb[j];
                if (!p.ln_y_bf16 && valid) *reinterpret_cast<f32x4*>(lb + 32 * rt + 4 * g) = y;
                packed[2 * g] = pack2(y[0], y[1]);
                packed[2 * g + 1] = pack2(y[2], y[3]);
            }
            if (p.ln_y_bf16 && valid) {
                *reinterpret_cast<u32x4*>(lbh + 32 * rt) = u32x4{packed[0], packed[1], packed[2], packed[3]};
                *reinterpret_cast<u32x4*>(lbh + 32 * rt + 8) = u32x4{packed[4], packed[5], packed[6], packed[7]};
            }
        }
        if (wave == 0 && h == 0 && valid && keep_y) {
            p.ln_stats[2 * (long)tok] = mean;
            p.ln_stats[2 * (long)tok + 1] = rstd;
        }
        }       // EPI_LN
        if constexpr (PF) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { ecur[rt][0] = enxt[rt][0]; ecur[rt][1] = enxt[rt][1]; }
        }
    }
    if constexpr (EPI == EPI_LNB) {
        if (!(c & 1)) {         // odd lanes hold the same totals
            const int n = f0 + 16 * h + 8 * ((c >> 4) & 1) + 4 * ((c >> 3) & 1) + 2 * ((c >> 2) & 1) + ((c >> 1) & 1);
            atomicAdd(p.lnb_dgamma + n, col_g);
            atomicAdd(p.lnb_dbeta + n, col_b);
            if (p.lnb_dbias) atomicAdd(p.lnb_dbias + n, col_c);
        }
    }
}

template <int NW, int RT, int KS, bool XB = true, int EPI = EPI_LN, bool F8 = false, int GROUPS = 1, bool PF = false>
int launch(const TlinP& p, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    static_assert(GROUPS == 1 || ((EPI == EPI_ACT || EPI == EPI_ACC) && !F8), "column groups: the plain epilogues only");
    constexpr int K = 16 * KS, N = 32 * NW * RT;
    constexpr size_t smem = (size_t)2 * TT * (F8 ? K + 16 : (K + 8) * 2) + (size_t)2 * NW * TT * 2 * 4 + (size_t)3 * N * 4;
    static bool attr_set = false;
    static int n_cu = 0;
    if (!attr_set) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wst_ln_kernel<NW, RT, KS, XB, EPI, F8, GROUPS, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        int dev = 0;
        GG_CHECK_HIP(hipGetDevice(&dev));
        GG_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_set = true;
    }
    const long ntiles = (p.M + TT - 1) / TT;
    // The persistent grid covers 91 % of the CUs (29 of the 32 per XCD), not all of them: these kernels hold their CUs for the whole
    // launch, and the weight-gradient kernels of the side streams - whose atomic panel adds do not need HBM - then find free CUs
    // beside every Linear instead of waiting for one to end (ms per step, interleaved: 100 % 28.96, 97 % 28.94, 94 % 28.66,
    // 91 % 28.61, 88 % 28.82, 75 % 29.45).  GG_WST_CU_PCT overrides.
    static const int cu_pct = getenv("GG_WST_CU_PCT") ? atoi(getenv("GG_WST_CU_PCT")) : 91;
    // p.grid_pct: the engine asks for the whole chip where it knows nothing runs beside the launch (forward passes on the caller's stream)
    const long cus = std::max<long>(8, (long)n_cu * (p.grid_pct > 0 ? std::min(p.grid_pct, 100) : cu_pct) / 100 / 8 * 8);
    const long slots = cus * ((NW == 4 && KS < 48) ? 2 : 1);          // resident workgroups: two per CU with 4 waves, one with 8 (or with the K = 768 tile)
    unsigned grid = (unsigned)std::min<long>(ntiles, slots);
    if (GROUPS > 1) {       // owners in whole groups of 8 (one per XCD), GROUPS workgroups each
        const long owners = std::max<long>(8, std::min<long>((ntiles + 7) / 8 * 8, slots / GROUPS / 8 * 8));
        grid = (unsigned)(owners * GROUPS);
    }
    if (ev0) hipExtLaunchKernelGGL((wst_ln_kernel<NW, RT, KS, XB, EPI, F8, GROUPS, PF>), dim3(grid), dim3(64 * NW), (unsigned)smem, st, ev0, ev1, 0, p);
    else hipLaunchKernelGGL((wst_ln_kernel<NW, RT, KS, XB, EPI, F8, GROUPS, PF>), dim3(grid), dim3(64 * NW), smem, st, p);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
}  // namespace

// out-proj / FFN2 of the production width, forward: bf16 X, fp32 residual + LayerNorm outputs
bool wst_ln_supported(const TlinP& p) {
    if (!p.ln_g || !p.ln_b || !p.ln_y || !p.ln_stats || !p.res || !p.x_bf16 || p.fp8) return false;
    if (p.N != 256 || (p.K != 256 && p.K != 512) || p.M < 1) return false;
    if (p.accumulate || p.mask_ref || p.act_relu || p.film_g || p.y_row_group) return false;
    if (!al16(p.X) || !al16(p.W) || !al16(p.Y) || !al16(p.res) || !al16(p.ln_y) || p.ldx % 8 || p.ldw % 8 || p.ldy % 4 || p.ldres % 4) return false;
    if ((p.res_bf16 && p.ldres % 8) || ((p.ln_y_bf16 || p.y_bf16) && p.ldy % 8)) return false;
    if (p.drop.p > 0.f && p.drop_ld % 2) return false;
    return true;
}

int wst_ln(const TlinP& p, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    GG_REQUIRE(wst_ln_supported(p), "wst_ln: unsupported shape / alignment");
    static const bool no_pf = getenv("GG_NO_WST_PF") != nullptr;
    if (p.res_bf16 && !no_pf) {       // bf16 residual rows: requested one tile ahead
        if (p.K == 256) return launch<4, 2, 16, true, EPI_LN, false, 1, true>(p, st, ev0, ev1);
        return launch<8, 1, 32, true, EPI_LN, false, 1, true>(p, st, ev0, ev1);
    }
    if (p.K == 256) return launch<4, 2, 16>(p, st, ev0, ev1);
    return launch<8, 1, 32>(p, st, ev0, ev1);
}

// the other token-tall Linears of the production width that have a weight-stationary instantiation:
//   1  y (fp32) += X W^T, N = 256, K = 512, bf16 X                       (dx1 += dh W1)
//   2  y (bf16) = drop(relu(X W^T + b)), N = 512, K = 256, fp32 X        (FFN1 forward)
//   3  y (bf16) = (X W^T) gated by ref, N = 512, K = 256, bf16 X          (dh = dres W2 * [h > 0])
//   4  y (bf16) = act(X W^T + b), N = 256, K = 256, bf16 X              (dctx = dres Wo)
//   5  as 1 with K = 768                                                 (dx += dqkv Win)
//   6  y (bf16) = X W^T + b, N = 768 = 3 column groups of 256, K = 256, fp32 X   (packed QKV projection, forward)
//   7  as 5 (K = 768) as 2 column groups of 128                          (dx += dqkv Win)
//   8 / 9  as 2 / 6 with bf16 X (the LayerNorm outputs stored in bf16: engine.hip "xst")
//   10 as 1 followed by the LayerNorm backward of the sum (EPI_LNB: dr, masked bf16 branch gradient, gamma / beta / bias column sums)
int wst_kind(const TlinP& p) {
    if (p.lnb_dres) {       // += then LayerNorm backward (dx1 = dr2 + dh W1 followed by LN1 backward)
        static const bool off = getenv("GG_NO_WST_LNB") != nullptr;
        if (off || p.fp8 || p.film_g || p.y_row_group || p.M < 1 || !p.accumulate || p.mask_ref || p.act_relu || p.y_bf16 || !p.x_bf16 || p.bias ||
            p.ln_y_bf16 || (p.res_bf16 && p.ldres % 8))
            return 0;
        if (p.N != 256 || p.K != 512 || !p.res || !p.ln_g || !p.ln_y || !p.ln_stats || !p.lnb_dgamma || !p.lnb_dbeta) return 0;
        if (!al16(p.X) || !al16(p.W) || !al16(p.Y) || !al16(p.res) || !al16(p.ln_y) || !al16(p.lnb_dres) || p.ldw % 8 || p.ldx % 8 || p.ldy % 8 ||
            p.ldres % 4 || p.res_rows < p.M)
            return 0;
        if (p.drop.p > 0.f && (p.drop_ld % 2)) return 0;
        return 10;
    }
    if (p.fp8 || p.ln_g || p.res || p.film_g || p.y_row_group || p.M < 1) return 0;
    if (!al16(p.X) || !al16(p.W) || !al16(p.Y) || p.ldw % 8 || p.ldx % (p.x_bf16 ? 8 : 4)) return 0;
    if (p.drop.p > 0.f && p.drop_ld % 2) return 0;
    // K = 768 (dx += dqkv Win) has an instantiation (case 5) but its 192 weight registers leave no room: 172 B of spills per
    // lane, 1.96 ms against 1.67 ms for the token-on-lane kernel at cfg3 - it stays there
    static const bool k768 = getenv("GG_WST_K768") != nullptr;
    // K = 768 (dx += dqkv Win): two column groups of 128 per token-tile owner, 4 waves with 192 weight registers each at one
    // wave per SIMD (no spills: 256 VGPRs + 103 AGPRs) - 1.66 -> 1.22 ms per step against the token-on-lane kernel, 4.25 TB/s
    static const bool k768g = getenv("GG_NO_WST_K768G") == nullptr;
    if (k768g && p.accumulate && !p.mask_ref && !p.act_relu && !p.y_bf16 && p.x_bf16 && p.N == 256 && p.K == 768 && p.drop.p == 0.f && !p.bias &&
        p.ldy % 4 == 0)
        return 7;
    if (p.accumulate && !p.mask_ref && !p.act_relu && !p.y_bf16 && p.x_bf16 && p.N == 256 && (p.K == 512 || (k768 && p.K == 768)) &&
        p.drop.p == 0.f && !p.bias && p.ldy % 4 == 0)
        return p.K == 512 ? 1 : 5;
    if (!p.accumulate && !p.mask_ref && p.y_bf16 && p.x_bf16 && p.N == 256 && p.K == 256 && p.ldy % 8 == 0) return 4;
    if (!p.accumulate && !p.mask_ref && p.y_bf16 && p.N == 512 && p.K == 256 && p.ldy % 8 == 0) return p.x_bf16 ? 8 : 2;
    static const bool no_qkv = getenv("GG_NO_WST_QKV") != nullptr;
    if (!no_qkv && !p.accumulate && !p.mask_ref && p.y_bf16 && p.N == 768 && p.K == 256 && p.ldy % 8 == 0 && !p.act_relu &&
        p.drop.p == 0.f)
        return p.x_bf16 ? 9 : 6;
    if (!p.accumulate && p.mask_ref && p.mask_bf16 && p.y_bf16 && p.x_bf16 && p.N == 512 && p.K == 256 && !p.act_relu &&
        p.drop.p == 0.f && !p.bias && p.ldy % 8 == 0 && p.ldref % 8 == 0 && al16(p.mask_ref))
        return 3;
    return 0;
}
// fp8 mode (p.fp8, e4m3 shadow weights): 1 / 2 Linear + dropout + residual + LayerNorm (K = 256 / 512), 3 FFN1 (N = 512, fp32 X),
// 4 the packed QKV projection (N = 768, fp32 X: 96 weight registers per lane as e4m3, out of reach in bf16)
int wst_fp8_kind(const TlinP& p) {
    if (!p.fp8 || !p.w_exp || p.film_g || p.y_row_group || p.mask_ref || p.accumulate || p.M < 1) return 0;
    if (!al16(p.X) || !al16(p.W) || !al16(p.Y) || p.ldw % 16 || p.ldx % (p.x_bf16 ? 8 : 4)) return 0;
    if (p.drop.p > 0.f && p.drop_ld % 2) return 0;
    if (p.ln_g) {
        if (!(p.ln_b && p.ln_y && p.ln_stats && p.res && p.x_bf16 && !p.act_relu && p.N == 256)) return 0;
        if (!al16(p.res) || !al16(p.ln_y) || p.ldy % 4 || p.ldres % 4) return 0;
        if ((p.res_bf16 && p.ldres % 8) || ((p.ln_y_bf16 || p.y_bf16) && p.ldy % 8)) return 0;
        return p.K == 256 ? 1 : (p.K == 512 ? 2 : 0);
    }
    if (p.res || !p.y_bf16 || p.K != 256 || p.ldy % 8) return 0;
    if (p.x_bf16) return p.N == 512 ? 5 : (p.N == 768 ? 6 : 0);           // bf16-stored LayerNorm outputs as X
    return p.N == 512 ? 3 : (p.N == 768 ? 4 : 0);
}
int wst_fp8(const TlinP& p, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    switch (wst_fp8_kind(p)) {
        case 1: return p.res_bf16 ? launch<4, 2, 16, true, EPI_LN, true, 1, true>(p, st, ev0, ev1) : launch<4, 2, 16, true, EPI_LN, true>(p, st, ev0, ev1);
        case 2: return p.res_bf16 ? launch<8, 1, 32, true, EPI_LN, true, 1, true>(p, st, ev0, ev1) : launch<8, 1, 32, true, EPI_LN, true>(p, st, ev0, ev1);
        case 3: return launch<8, 2, 16, false, EPI_ACT, true>(p, st, ev0, ev1);
        case 4: return launch<8, 3, 16, false, EPI_ACT, true>(p, st, ev0, ev1);
        case 5: return launch<8, 2, 16, true, EPI_ACT, true>(p, st, ev0, ev1);
        case 6: return launch<8, 3, 16, true, EPI_ACT, true>(p, st, ev0, ev1);
    }
    set_error("wst_fp8: no instantiation for this call");
    return -2;
}
int wst_other(const TlinP& p, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    switch (wst_kind(p)) {
        case 1: return launch<8, 1, 32, true, EPI_ACC>(p, st, ev0, ev1);
        case 2: return launch<8, 2, 16, false, EPI_ACT>(p, st, ev0, ev1);
        case 3: {
            static const bool no_pf = getenv("GG_NO_WST_PF") != nullptr;
            return no_pf ? launch<8, 2, 16, true, EPI_MASK>(p, st, ev0, ev1) : launch<8, 2, 16, true, EPI_MASK, false, 1, true>(p, st, ev0, ev1);
        }
        case 4: return launch<4, 2, 16, true, EPI_ACT>(p, st, ev0, ev1);
        case 5: return launch<8, 1, 48, true, EPI_ACC>(p, st, ev0, ev1);
        case 6: return launch<4, 2, 16, false, EPI_ACT, false, 3>(p, st, ev0, ev1);
        case 7: return launch<4, 1, 48, true, EPI_ACC, false, 2>(p, st, ev0, ev1);
        case 8: return launch<8, 2, 16, true, EPI_ACT>(p, st, ev0, ev1);
        case 9: return launch<4, 2, 16, true, EPI_ACT, false, 3>(p, st, ev0, ev1);
        case 10: return launch<8, 1, 32, true, EPI_LNB>(p, st, ev0, ev1);
    }
    set_error("wst_other: no instantiation for this call");
    return -2;
}

}  // namespace gg
