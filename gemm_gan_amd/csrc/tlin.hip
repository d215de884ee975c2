// Token-on-lane Linear kernels for the tall-skinny products of the encoder stack (bf16 MFMA).
//
//   Y^T[N x tokens] = W[N x K] . X^T[K x tokens]        (v_mfma_f32_32x32x16_bf16, fp32 accumulate)
//
// Every Linear of the conditioning stack has a huge token count M (65k - 200k rows) and tiny N, K
// (256 - 1024): the weights fit in L2, the activations do not fit anywhere.  A tile GEMM re-reads the
// activation panel once per N tile and was measured at 1.6 TB/s of algorithmic traffic; here
//   * a wave owns TOK (32 or 64) tokens and keeps their activations REGISTER-RESIDENT as MFMA B
//     fragments (lane = token, 8 consecutive k per lane), read from HBM exactly once through a small
//     wave-private LDS slab (full-line coalesced loads, fp32 -> bf16 on the way, FiLM optionally fused);
//   * the weights (pre-converted bf16 shadow copy, [N][K] row-major) stream through a double-buffered
//     LDS chunk of 32 output features as the MFMA A operand, shared by the 4 waves of the workgroup;
//   * the accumulator tile then has OUTPUT FEATURES in registers and TOKENS on lanes, so the epilogue
//     (bias, ReLU, dropout, activation-mask, += , residual add and LayerNorm over the features) is
//     in-lane arithmetic plus one cross-half shuffle, and each lane stores 16-byte pieces of its own row.
// Two instantiation families:
//   stream   (NT_RES = 0): K <= 256 resident, any N streamed 32 features at a time, TOK = 32.
//   resident (NT_RES = N/32 <= 8): all N accumulators resident, K streamed in slices of 256, TOK = 32;
//                                  enables the fused residual + LayerNorm epilogue (N = E).
#include "kernels.h"

namespace gg {

namespace {

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float drop_factor(const DropKey& k, uint64_t i, float keep_scale) {
    uint32_t lo = (uint32_t)i, hi = (uint32_t)(i >> 32);
    uint32_t h = fmix32(lo * 0x9E3779B1u + k.k0);
    h = fmix32(h ^ k.k1 ^ (hi * 0x7F4A7C15u));
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    return u >= k.p ? keep_scale : 0.f;
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

constexpr int XLD = 64 + 8;         // bf16 per LDS row of the X staging slab (144 B)
constexpr float LN_EPS = 1e-5f;

// KSL: length of the K slice kept register-resident (64 | 128 | 256); K must be a multiple of KSL so that
// every MFMA chain below is straight-line code with compile-time fragment indices.
// XB / YB: activations read / written as bf16 (tensors that only ever feed bf16 MFMA operands are stored in bf16)
template <int TOK, int NT_RES, int KSL, bool XB, bool YB>
__global__ __launch_bounds__(256, (NT_RES == 0 ? 2 : 1)) void tlin_kernel(const TlinP p) {
    constexpr int TT = TOK / 32;                       // 32-token fragment sets per wave
    constexpr int NACC = NT_RES > 0 ? NT_RES : 1;
    constexpr int WLD = KSL + 8;                       // bf16 per LDS row of a weight chunk
    constexpr int PIECES = KSL / 8;                    // 16-byte pieces per weight row
    constexpr int WLOADS = (32 * PIECES + 255) / 256;  // pieces per thread per chunk
    __shared__ __attribute__((aligned(16))) __bf16 Ws[2][32 * (KSL + 8)];
    __shared__ __attribute__((aligned(16))) __bf16 Xs[4][TOK * XLD];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    const long tok0 = (long)blockIdx.x * (4 * TOK) + wave * TOK;
    const int ntiles = p.N / 32;
    const int nks = p.K / KSL;
    const int nchunks = nks * ntiles;
    const long last_tok = p.M - 1;
    const float ksd = p.drop.p > 0.f ? 1.f / (1.f - p.drop.p) : 1.f;

    // ---- weight chunk pipeline: chunk (ks, nt) = 32 output features x KSL reduction elements ----------------
    u32x4 wreg[WLOADS];
    const __bf16* Wp = reinterpret_cast<const __bf16*>(p.W);
    auto load_chunk = [&](int ks, int nt) {
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            const int f = tid + 256 * i;
            const int row = f / PIECES, piece = f % PIECES;
            wreg[i] = u32x4{0u, 0u, 0u, 0u};
            if (row < 32) wreg[i] = *reinterpret_cast<const u32x4*>(Wp + (long)(nt * 32 + row) * p.ldw + ks * KSL + 8 * piece);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            const int f = tid + 256 * i;
            const int row = f / PIECES, piece = f % PIECES;
            if (row < 32) *reinterpret_cast<u32x4*>(&Ws[buf][row * WLD + 8 * piece]) = wreg[i];
        }
    };
    // next chunk after (ks, nt) in (ks outer, nt inner) order
    auto next_of = [&](int ks, int nt, int& nks_, int& nnt_) {
        nnt_ = nt + 1; nks_ = ks;
        if (nnt_ == ntiles) { nnt_ = 0; nks_ = ks + 1; }
    };

    f32x16 acc[NACC][TT];
    bf16x8 xf[TT][KSL / 16];

    // staged rows of this lane (rows past the end re-read the last valid token: never stored) and their
    // FiLM group, in 32-bit arithmetic, computed once.  One load instruction covers 64 reduction elements of
    // RPI rows: fp32 -> 16 lanes x 16 B per row (4 rows), bf16 -> 8 lanes x 16 B per row (8 rows).
    constexpr int RPI = XB ? 8 : 4;
    constexpr int NLD = TOK / RPI;
    const int lrow = XB ? (lane >> 3) : (lane >> 4);
    const int lcol = XB ? 8 * (lane & 7) : 4 * (lane & 15);      // element offset inside the 64-wide window
    int rtok[NLD], rgrp[NLD];
    {
        const int t0 = (int)tok0 + lrow, lt = (int)last_tok;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            rtok[i] = min(t0 + RPI * i, lt);
            rgrp[i] = (!XB && p.film_g) ? rtok[i] / p.film_group : 0;
        }
    }
    load_chunk(0, 0);
    store_chunk(0);
    int chunk = 0;
    for (int ks = 0; ks < nks; ++ks) {
        // ---- stage this wave's X slice: HBM -> (fp32 -> bf16, FiLM) -> LDS slab -> B fragments --------
#pragma unroll
        for (int q4 = 0; q4 < KSL / 64; ++q4) {
            const int kbase = ks * KSL + q4 * 64 + lcol;
            if constexpr (XB) {
                const __bf16* Xb = reinterpret_cast<const __bf16*>(p.X);
                u32x4 v[NLD];
#pragma unroll
                for (int i = 0; i < NLD; ++i) v[i] = *reinterpret_cast<const u32x4*>(Xb + (long)rtok[i] * p.ldx + kbase);
#pragma unroll
                for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(&Xs[wave][(i * RPI + lrow) * XLD + lcol]) = v[i];
            } else {
                const float* Xf = reinterpret_cast<const float*>(p.X);
                f32x4 v[NLD];
#pragma unroll
                for (int i = 0; i < NLD; ++i) v[i] = *reinterpret_cast<const f32x4*>(Xf + (long)rtok[i] * p.ldx + kbase);
                if (p.film_g) {
#pragma unroll
                    for (int i = 0; i < NLD; ++i) {
                        const long fo = (long)rgrp[i] * p.film_ld + kbase;
                        const f32x4 g = *reinterpret_cast<const f32x4*>(p.film_g + fo);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(p.film_b + fo);
                        v[i] = g * v[i] + b;
                    }
                }
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    u32x2 w = {pack2(v[i][0], v[i][1]), pack2(v[i][2], v[i][3])};
                    *reinterpret_cast<u32x2*>(&Xs[wave][(i * RPI + lrow) * XLD + lcol]) = w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    xf[t][q4 * 4 + s] = *reinterpret_cast<const bf16x8*>(&Xs[wave][(t * 32 + c) * XLD + 16 * s + 8 * h]);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();     // chunk `chunk` is in Ws[chunk & 1]

        if constexpr (NT_RES > 0) {
#pragma unroll
            for (int nt = 0; nt < NT_RES; ++nt) {
                const int buf = chunk & 1;
                const bool more = chunk + 1 < nchunks;
                if (more) { int a_, b_; next_of(ks, nt, a_, b_); load_chunk(a_, b_); }
                if (ks == 0) {
#pragma unroll
                    for (int t = 0; t < TT; ++t)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[nt][t][i] = 0.f;
                }
                // weight fragments are fetched four k-steps at a time so the LDS latency is paid once per group
#pragma unroll
                for (int s4 = 0; s4 < KSL / 64; ++s4) {
                    bf16x8 wf[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) wf[u] = *reinterpret_cast<const bf16x8*>(&Ws[buf][c * WLD + 16 * (4 * s4 + u) + 8 * h]);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int t = 0; t < TT; ++t)
                            acc[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u], xf[t][4 * s4 + u], acc[nt][t], 0, 0, 0);
                }
                if (more) store_chunk(buf ^ 1);
                __syncthreads();
                ++chunk;
            }
        } else {
            for (int nt = 0; nt < ntiles; ++nt) {
                const int buf = chunk & 1;
                const bool more = chunk + 1 < nchunks;
                if (more) load_chunk(0, nt + 1);
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[0][t][i] = 0.f;
#pragma unroll
                for (int s4 = 0; s4 < KSL / 64; ++s4) {
                    bf16x8 wf[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) wf[u] = *reinterpret_cast<const bf16x8*>(&Ws[buf][c * WLD + 16 * (4 * s4 + u) + 8 * h]);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int t = 0; t < TT; ++t)
                            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u], xf[t][4 * s4 + u], acc[0][t], 0, 0, 0);
                }
                // ---- streamed epilogue: 32 features x TOK tokens -------------------------------------------
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    const long tok = tok0 + t * 32 + c;
                    const bool valid = tok < p.M;
                    const long tokc = valid ? tok : last_tok;
                    const long yrow = p.y_row_group ? tokc + tokc / p.y_row_group + 1 : tokc;
                    const long yoff = yrow * p.ldy + nt * 32 + 4 * h;
                    const long moff = tokc * p.ldref + nt * 32 + 4 * h;
                    f32x4 mm[4], yy[4], bb[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        mm[g] = f32x4{1.f, 1.f, 1.f, 1.f};
                        if (p.mask_ref) {
                            if (p.mask_bf16) {     // only the sign matters: expand the four bf16 to fp32 bit patterns
                                const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.mask_ref) + moff + 8 * g);
                                mm[g] = f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xffff0000u),
                                              __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xffff0000u)};
                            } else {
                                mm[g] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.mask_ref) + moff + 8 * g);
                            }
                        }
                        yy[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (!YB) {
                            if (p.accumulate) yy[g] = *reinterpret_cast<const f32x4*>(reinterpret_cast<float*>(p.Y) + yoff + 8 * g);
                        }
                        bb[g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nt * 32 + 8 * g + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = nt * 32 + 8 * g + 4 * h;
                        f32x4 v = {acc[0][t][4 * g], acc[0][t][4 * g + 1], acc[0][t][4 * g + 2], acc[0][t][4 * g + 3]};
                        v += bb[g];
                        if (p.act_relu) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                        }
                        if (p.drop.p > 0.f) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] *= drop_factor(p.drop, (uint64_t)tokc * p.drop_ld + n + j, ksd);
                        }
                        if (p.mask_ref) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = mm[g][j] > 0.f ? v[j] * p.mask_scale : 0.f;
                        }
                        v += yy[g];
                        if (valid) {
                            if constexpr (YB) {
                                u32x2 w = {pack2(v[0], v[1]), pack2(v[2], v[3])};
                                *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(p.Y) + yoff + 8 * g) = w;
                            } else {
                                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.Y) + yoff + 8 * g) = v;
                            }
                        }
                    }
                }
                if (more) store_chunk(buf ^ 1);
                __syncthreads();
                ++chunk;
            }
        }
    }

    if constexpr (NT_RES > 0) {
        // ---- resident epilogue: bias, dropout, residual, LayerNorm over the N = 32*NT_RES features ----------
        constexpr int t = 0;                         // TOK == 32 in resident mode
        const long tok = tok0 + c;
        const bool valid = tok < p.M;
        const long tokc = valid ? tok : last_tok;        // clamped row: loads are unconditional (batched, no exec branches)
        const long yrow = p.y_row_group ? tokc + tokc / p.y_row_group + 1 : tokc;
        const float* resp = p.res ? p.res + (tokc % p.res_rows) * p.ldres : nullptr;
        float* yb = reinterpret_cast<float*>(p.Y) + yrow * p.ldy;
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT_RES; ++nt) {
            f32x4 rr[4], yy[4], bb[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {                // issue the tile's loads together
                const int n = nt * 32 + 8 * g + 4 * h;
                rr[g] = resp ? *reinterpret_cast<const f32x4*>(resp + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                yy[g] = p.accumulate ? *reinterpret_cast<const f32x4*>(yb + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                bb[g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nt * 32 + 8 * g + 4 * h;
                f32x4 v = {acc[nt][t][4 * g], acc[nt][t][4 * g + 1], acc[nt][t][4 * g + 2], acc[nt][t][4 * g + 3]};
                v += bb[g];
                if (p.drop.p > 0.f) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= drop_factor(p.drop, (uint64_t)tokc * p.drop_ld + n + j, ksd);
                }
                v += rr[g] + yy[g];
                if (valid) *reinterpret_cast<f32x4*>(yb + n) = v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[nt][t][4 * g + j] = v[j];
                    sum += v[j];
                }
            }
        }
        if (p.ln_g) {
            const float invn = 1.f / (float)(32 * NT_RES);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * invn;
            float var = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT_RES; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float d = acc[nt][t][i] - mean;
                    var += d * d;
                }
            var += __shfl_xor(var, 32, 64);
            const float rstd = rsqrtf(var * invn + LN_EPS);
            if (valid) {
#pragma unroll
                for (int nt = 0; nt < NT_RES; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = nt * 32 + 8 * g + 4 * h;
                        const f32x4 gg_ = *reinterpret_cast<const f32x4*>(p.ln_g + n);
                        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ln_b + n);
                        f32x4 y;
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[j] = (acc[nt][t][4 * g + j] - mean) * rstd * gg_[j] + bb[j];
                        *reinterpret_cast<f32x4*>(p.ln_y + tok * p.ldy + n) = y;
                    }
                if (h == 0) {
                    p.ln_stats[2 * tok] = mean;
                    p.ln_stats[2 * tok + 1] = rstd;
                }
            }
        }
    }
}

// one workgroup column per 2-D parameter: wb = bf16(W) [rows][cols], wtb = bf16(W^T) [cols][rows]
__global__ void shadow_kernel(const float* __restrict__ w, __bf16* __restrict__ wb, __bf16* __restrict__ wtb,
                              const ShadowEntry* __restrict__ tab) {
    const ShadowEntry e = tab[blockIdx.y];
    const long n = (long)e.rows * e.cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / e.cols), c = (int)(i % e.cols);
        const __bf16 v = (__bf16)w[e.off + i];
        wb[e.off + i] = v;
        wtb[e.off + (long)c * e.rows + r] = v;
    }
}

template <int TOK, int NT_RES, int KSL, bool XB, bool YB>
int launch(const TlinP& p, hipStream_t st) {
    const long blocks = (p.M + 4 * TOK - 1) / (4 * TOK);
    hipLaunchKernelGGL((tlin_kernel<TOK, NT_RES, KSL, XB, YB>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int TOK, int NT_RES, int KSL>
int launch_t(const TlinP& p, hipStream_t st) {
    if (NT_RES == 0) {
        if (p.x_bf16) return p.y_bf16 ? launch<TOK, NT_RES, KSL, true, true>(p, st) : launch<TOK, NT_RES, KSL, true, false>(p, st);
        return p.y_bf16 ? launch<TOK, NT_RES, KSL, false, true>(p, st) : launch<TOK, NT_RES, KSL, false, false>(p, st);
    }
    return p.x_bf16 ? launch<TOK, NT_RES, KSL, true, false>(p, st) : launch<TOK, NT_RES, KSL, false, false>(p, st);
}
inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
inline bool needs_resident(const TlinP& p) { return p.ln_g || p.res || !(p.K == 64 || p.K == 128 || p.K == 256); }
}  // namespace

// supported instantiations: stream K in {64,128,256} (any N % 32 == 0);
// resident (N, K-slice) in {(256, 256), (128, 128), (64, 64)} with K a multiple of the slice
bool tlin_supported(const TlinP& p) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return false;
    if (p.N % 32 || p.K % 64) return false;
    if (!al16(p.X) || !al16(p.W) || !al16(p.Y) || p.ldx % (p.x_bf16 ? 8 : 4) || p.ldy % 4 || p.ldw % 8) return false;
    if (p.bias && !al16(p.bias)) return false;
    if (p.film_g && (p.x_bf16 || !al16(p.film_g) || !al16(p.film_b) || p.film_ld % 4 || p.film_group <= 0)) return false;
    if (p.mask_ref && (!al16(p.mask_ref) || p.ldref % 4)) return false;
    if (p.res && (!al16(p.res) || p.ldres % 4)) return false;
    if (p.ln_g && (!al16(p.ln_g) || !al16(p.ln_b) || !al16(p.ln_y))) return false;
    if (p.y_bf16 && p.accumulate) return false;
    if (!needs_resident(p)) return true;
    if (p.mask_ref || p.act_relu || p.y_bf16) return false;   // not implemented in the resident epilogue
    if (p.N == 256) return p.K % 256 == 0;
    if (p.N == 128) return p.K % 128 == 0;
    if (p.N == 64) return p.K % 64 == 0;
    return false;
}

int tlin(const TlinP& p, hipStream_t st) {
    GG_REQUIRE(tlin_supported(p), "tlin: unsupported shape / alignment");
    if (!needs_resident(p)) {
        if (p.K == 256) return launch_t<32, 0, 256>(p, st);
        if (p.K == 128) return launch_t<32, 0, 128>(p, st);
        return launch_t<32, 0, 64>(p, st);
    }
    if (p.N == 256) return launch_t<32, 8, 256>(p, st);
    if (p.N == 128) return launch_t<32, 4, 128>(p, st);
    return launch_t<32, 2, 64>(p, st);
}

int k_shadow_weights(const float* w, void* wb, void* wtb, const ShadowEntry* tab_dev, int n_entries, hipStream_t st) {
    if (n_entries <= 0) return 0;
    shadow_kernel<<<dim3(64, n_entries), 256, 0, st>>>(w, reinterpret_cast<__bf16*>(wb), reinterpret_cast<__bf16*>(wtb), tab_dev);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gg
