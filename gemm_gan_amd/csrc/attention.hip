// Fused multi-head self-attention for the encoder layers (bf16 MFMA, fp32 accumulate / softmax):
// forward + backward without ever writing the [S,S] probability tensor to HBM.
//
// Reference semantics: torch F.multi_head_attention_forward (torch/nn/functional.py:6206-6660) as used
// by nn.TransformerEncoderLayer: softmax(q k^T / sqrt(dh) + key_padding(-inf)) -> dropout -> @ v, on
// the packed projection qkv [N, S, 3E] (heads addressed by column offset).
//
// MFMA orientation (v_mfma_f32_32x32x16_bf16, C/D map col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)):
//   forward and dQ kernels compute the TRANSPOSED score tile  S^T[key, q] = K Q^T  so that one lane
//   owns one query column: row max / row sum / online rescale are in-register plus ONE cross-half
//   shuffle, and the probability tile is already the B operand of the next product
//   (O^T = V^T P^T, dQ^T = K^T dS^T: "accumulator tile as the next MFMA's operand", guide section 3).
//   the dK/dV kernel computes S[q, key] = Q K^T with the KEY on the lane: P and dS are then the B
//   operands of dV^T = dO^T P and dK^T = Q^T dS, accumulated over all query tiles in registers.
// Outputs are written as 16-byte stores (a lane holds 4 consecutive head-dim elements per register
// group), two half-waves completing each 32-byte sector.
//
// Dropout uses the same counter hash and the same element index ((n*nh+h)*S+q)*S+key as the unfused
// path (kernels.hip), so both paths draw identical masks.
#include "kernels.h"
#include <cstdlib>
#include "drop_rng.h"

namespace gg {

namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ bf16x8 frag_from_f32(const float* v) {     // 8 floats -> 8 bf16
    u32x4 w = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ bf16x8 frag_from_acc(const f32x16& a, int s2) {   // regs 8*s2 .. 8*s2+7
    u32x4 w = {pack2(a[8 * s2 + 0], a[8 * s2 + 1]), pack2(a[8 * s2 + 2], a[8 * s2 + 3]),
               pack2(a[8 * s2 + 4], a[8 * s2 + 5]), pack2(a[8 * s2 + 6], a[8 * s2 + 7])};
    return __builtin_bit_cast(bf16x8, w);
}
// row index inside a 32x32 accumulator tile held in register i by lane-half h
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

constexpr float LOG2E = 1.4426950408889634f;
// Query tiles are dealt to the 4 waves of a workgroup round-robin.  When one tile is left over (S = 257: eight full
// tiles + the CLS row) and it holds at most CO_MAXQ queries, all four waves share it - each takes every 4th key tile -
// and merge their partial results through LDS, instead of one wave walking a third pass alone (27 -> 21 tile steps).
constexpr int CO_MAXQ = 2;       // 3 partial sets (waves 1..3) x CO_MAXQ x (DH + 2) floats must fit beside K / V^T below 80 KB (two workgroups per CU)
typedef short s16x4_t __attribute__((ext_vector_type(4)));
// 8 consecutive elements at element offset `off` of a fp32 (IOB = false) or bf16 (IOB = true) tensor, as an MFMA fragment
template <bool IOB>
__device__ __forceinline__ bf16x8 load_frag8(const void* base, long off) {
    if constexpr (IOB) {
        return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(base) + off);
    } else {
        const float* p = reinterpret_cast<const float*>(base) + off;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
        u32x4 w = {pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
        return __builtin_bit_cast(bf16x8, w);
    }
}
// the same 8 elements as fp32 values
template <bool IOB>
__device__ __forceinline__ void load_f32x8(const void* base, long off, float (&v)[8]) {
    if constexpr (IOB) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(base) + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[2 * j] = __builtin_bit_cast(float, w[j] << 16);
            v[2 * j + 1] = __builtin_bit_cast(float, w[j] & 0xffff0000u);
        }
    } else {
        const float* p = reinterpret_cast<const float*>(base) + off;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
    }
}
// 4 consecutive fp32 results to a fp32 / bf16 tensor
template <bool IOB>
__device__ __forceinline__ void store4(void* base, long off, f32x4 v) {
    if constexpr (IOB) {
        u32x2 w = {pack2(v[0], v[1]), pack2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(base) + off) = w;
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = v;
    }
}
// raw v_exp_f32 (2^x): inputs here are <= 0 or -inf; flush-to-zero of tiny results is harmless for softmax
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- LDS staging -------------------------------------------------------------------------------------
// Row padding of the TRANSPOSED images img[d][row] (Vt of the forward, Kt of the dQ kernels).  A lane reads 8 bytes of row
// d = its index c: with a row stride of (rows + 8) bf16 = 4 * odd dwords, lanes c and c + 16 of a 32-lane group hit the same
// banks (half of the LDS cycles of these kernels were conflict cycles, SQ_LDS_BANK_CONFLICT); (rows + 12) bf16 = 2 * odd
// dwords spreads 32 lanes over all 64 banks.  The rows are then only 8-byte aligned: the staging writes go out as two
// 8-byte stores.
constexpr int TPAD = 12;
// row-major image  img[row][d] (ld = DH+8 bf16) of X[row, col0 + d], rows >= S zero-filled.
// Loads are unconditional from clamped rows, four per thread in flight, zero-fill by select at the LDS write: a guard
// around the load made every iteration of the loop one serialised memory round trip (9 of them for S = 257).
template <int DH, bool IOB>
__device__ __forceinline__ void stage_rows(__bf16* img, const void* X, long xoff, long ldx, int S, int Sp, int tid, int nthreads) {
    constexpr int LD = DH + 8, U = 4;
    const int nchunk = Sp * (DH / 8);
    if (S <= 0) {                                       // nothing valid to read (never the case for the callers' chunking)
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = tid; c < nchunk; c += nthreads) *reinterpret_cast<bf16x8*>(img + (c / (DH / 8)) * LD + 8 * (c % (DH / 8))) = z;
        return;
    }
    const int rmax = S - 1;
    for (int c0 = tid; c0 < nchunk; c0 += U * nthreads) {
        bf16x8 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = min(c0 + u * nthreads, nchunk - 1);
            const int row = c / (DH / 8), c8 = c % (DH / 8);
            w[u] = load_frag8<IOB>(X, xoff + (long)min(row, rmax) * ldx + 8 * c8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * nthreads;
            if (c < nchunk) {
                const int row = c / (DH / 8), c8 = c % (DH / 8);
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<bf16x8*>(img + row * LD + 8 * c8) = row < S ? w[u] : z;
            }
        }
    }
}
// transposed image  img[d][row] (ld = Sp+8 bf16) of X[row, col0 + d]  (same loading discipline)
template <int DH, bool IOB>
__device__ __forceinline__ void stage_transposed(__bf16* img, const void* X, long xoff, long ldx, int S, int Sp, int tid, int nthreads) {
    const int LD = Sp + TPAD;
    const int nunit = (Sp / 8) * (DH / 4);
    if (S <= 0) {
        for (int u = tid; u < nunit; u += nthreads)
            for (int q = 0; q < 4; ++q)
                for (int hh = 0; hh < 2; ++hh)
                    *reinterpret_cast<s16x4_t*>(img + (4 * (u % (DH / 4)) + q) * LD + 8 * (u / (DH / 4)) + 4 * hh) = s16x4_t{0, 0, 0, 0};
        return;
    }
    const int rmax = S - 1;
    for (int u = tid; u < nunit; u += nthreads) {
        const int r8 = u / (DH / 4), d4 = u % (DH / 4);
        if constexpr (IOB) {
            const __bf16* Xb = reinterpret_cast<const __bf16*>(X) + xoff;
            s16x4_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const s16x4_t*>(Xb + (long)min(8 * r8 + j, rmax) * ldx + 4 * d4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (8 * r8 + j >= S) v[j] = s16x4_t{0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s16x4_t w0, w1;
#pragma unroll
                for (int j = 0; j < 4; ++j) { w0[j] = v[j][q]; w1[j] = v[4 + j][q]; }
                *reinterpret_cast<s16x4_t*>(img + (4 * d4 + q) * LD + 8 * r8) = w0;
                *reinterpret_cast<s16x4_t*>(img + (4 * d4 + q) * LD + 8 * r8 + 4) = w1;
            }
        } else {
            const float* Xf = reinterpret_cast<const float*>(X) + xoff;
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(Xf + (long)min(8 * r8 + j, rmax) * ldx + 4 * d4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (8 * r8 + j >= S) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<u32x2*>(img + (4 * d4 + q) * LD + 8 * r8) = u32x2{pack2(v[0][q], v[1][q]), pack2(v[2][q], v[3][q])};
                *reinterpret_cast<u32x2*>(img + (4 * d4 + q) * LD + 8 * r8 + 4) = u32x2{pack2(v[4][q], v[5][q]), pack2(v[6][q], v[7][q])};
            }
        }
    }
}
// A-operand fragment of a TRANSPOSED image for the "accumulator as B operand" product: element j of
// lane-half h must be the source row 16*s2 + 8*(j>>2) + 4*h + (j&3) (guide section 3) of column d.
__device__ __forceinline__ bf16x8 frag_transposed(const __bf16* img, int ld, int d, int row0, int s2, int h) {
    const __bf16* p = img + d * ld + row0 + 16 * s2 + 4 * h;
    const u32x2 a = *reinterpret_cast<const u32x2*>(p);
    const u32x2 b = *reinterpret_cast<const u32x2*>(p + 8);
    u32x4 w = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8, w);
}

// ------------------------------------------------------------------------------------------------------
// forward: one workgroup (4 waves) per (sample, head); K row-major + V transposed resident in LDS; each
// wave walks query tiles of 32 rows with an online softmax over key tiles of 32.
// ------------------------------------------------------------------------------------------------------
template <int DH, bool IOB>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const void* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                       int mask_B, void* __restrict__ ctx, float* __restrict__ lse2,
                                                       int S, int E, int nh, DropKey drop_in, int qkv_B) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Sp = (S + 31) / 32 * 32;
    constexpr int LDK = DH + 8;
    const int LDV = Sp + TPAD;
    __bf16* Ks = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Vt = Ks + Sp * LDK;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vt + DH * LDV);
    constexpr int DT = (DH + 31) / 32;          // head-dim tiles of the O^T accumulator
    constexpr int KS = DH / 16;                 // k-steps of the score product

    const int n = blockIdx.x / nh, hd = blockIdx.x % nh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;      // this (sample, head) inside qkv; replicas may share one projection
    stage_rows<DH, IOB>(Ks, qkv, base + E, ld, S, Sp, tid, 256);
    stage_transposed<DH, IOB>(Vt, qkv, base + 2 * E, ld, S, Sp, tid, 256);
    for (int i = tid; i < Sp; i += 256) Ms[i] = (i >= S) ? 1 : (mask ? mask[(long)(n % mask_B) * S + i] : 0);
    __syncthreads();
    uint8_t* Mt = Ms + Sp;                                 // per key tile: does it hold any masked / padded key?
    float* Co = reinterpret_cast<float*>(Mt + 64);        // [waves 1..3][CO_MAXQ][DH + 2] partials of the shared query tile
    for (int t = tid; t < Sp / 32; t += 256) {
        uint8_t any = 0;
        for (int j = 0; j < 32; ++j) any |= Ms[t * 32 + j];
        Mt[t] = any;
    }
    __syncthreads();

    const float sc = rsqrtf((float)DH) * LOG2E;     // applied to the fp32 scores after the MFMA
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    const int nqt = Sp / 32, nkt = Sp / 32;
    const bool coop = (nqt % 4 == 1) && nqt > 1 && (S - 32 * (nqt - 1)) <= CO_MAXQ;      // workgroup-uniform
    const int nq_main = coop ? nqt - 1 : nqt;
    for (int it = wave; it < nq_main + (coop ? 4 : 0); it += 4) {
        const bool shared = it >= nq_main;             // the left-over tile: every wave passes here exactly once
        const int qt = shared ? nqt - 1 : it;
        const int kt0 = shared ? wave : 0, kstep = shared ? 4 : 1;
        const int q = qt * 32 + c;
        bf16x8 qf[KS];
        const int qc = min(q, S - 1);                  // rows past the end re-read the last query: never stored
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = load_frag8<IOB>(qkv, base + (long)qc * ld + 16 * s + 8 * h);
        float m = -INFINITY, l = 0.f;
        f32x16 O[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
        // dropout stream state of this lane's row at its first key pair (drop_rng.h: linear in the pair index)
        const uint32_t srow = drop_state(drop, (((uint64_t)blockIdx.x * S + (uint64_t)q) * (uint64_t)drop_attn_ld(S)) >> 1) + (uint32_t)(2 * h) * DROP_PHI;
        for (int kt = kt0; kt < nkt; kt += kstep) {
            f32x16 s16;
#pragma unroll
            for (int i = 0; i < 16; ++i) s16[i] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + c) * LDK + 16 * s + 8 * h);
                s16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s16, 0, 0, 0);
            }
            float mt = -INFINITY;
            if (__builtin_amdgcn_readfirstlane((int)Mt[kt])) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 32 + acc_row(i, h);
                    s16[i] = Ms[key] ? -INFINITY : s16[i] * sc;
                    mt = fmaxf(mt, s16[i]);
                }
            } else {                    // tile without masked keys: no per-element mask lookups
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s16[i] *= sc;
                    mt = fmaxf(mt, s16[i]);
                }
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            // Lazy reference update: the softmax reference m only moves when a tile's maximum exceeds it by more than
            // 2^8 (scores are in the log2 domain), so p <= 256 in between - exact after the final O / l - and the
            // O *= alpha pass over the 32 output accumulators is skipped for almost every tile.
            const bool move = mt > m + 8.f;
            if (__builtin_amdgcn_ballot_w64(move) != 0) {
                const float mn = move ? mt : m;
                const float alpha = (m == -INFINITY) ? 0.f : fast_exp2(m - mn);      // mn is finite wherever move is set
                l *= alpha;
                m = mn;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
            }
            const float mref = (m == -INFINITY) ? 0.f : m;
            float lt = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = fast_exp2(s16[i] - mref);
                lt += p;
                s16[i] = p;
            }
            lt += __shfl_xor(lt, 32, 64);
            l += lt;
            if (drop.p > 0.f) {     // registers 4g..4g+3 hold 4 consecutive keys: two pair hashes
                const uint32_t skt = srow + (uint32_t)(kt * 16) * DROP_PHI;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t b0 = drop_bits(skt + (uint32_t)(4 * g) * DROP_PHI), b1 = drop_bits(skt + (uint32_t)(4 * g + 1) * DROP_PHI);
                    s16[4 * g + 0] = drop_keep_even(b0, drop.thr) ? s16[4 * g + 0] : 0.f;
                    s16[4 * g + 1] = drop_keep_odd(b0, drop.thr) ? s16[4 * g + 1] : 0.f;
                    s16[4 * g + 2] = drop_keep_even(b1, drop.thr) ? s16[4 * g + 2] : 0.f;
                    s16[4 * g + 3] = drop_keep_odd(b1, drop.thr) ? s16[4 * g + 3] : 0.f;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = frag_from_acc(s16, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 vf = frag_transposed(Vt, LDV, min(dt * 32 + c, DH - 1), kt * 32, s2, h);
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, O[dt], 0, 0, 0);
                }
            }
        }
        if (shared) {      // merge the four key-range partials of the shared tile (softmax-weighted) in wave 0
            if (wave != 0 && q < S) {
                float* part = Co + ((wave - 1) * CO_MAXQ + min(c, CO_MAXQ - 1)) * (DH + 2);
                if (h == 0) { part[0] = m; part[1] = l; }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int d = dt * 32 + acc_row(i, h);
                        if (d < DH) part[2 + d] = O[dt][i];
                    }
            }
            __syncthreads();
            if (wave != 0) continue;
            if (q < S) {
                const float* p0 = Co + min(c, CO_MAXQ - 1) * (DH + 2);
                constexpr int WS = CO_MAXQ * (DH + 2);
                const float M = fmaxf(fmaxf(m, p0[0]), fmaxf(p0[WS], p0[2 * WS]));
                const float mref = (M == -INFINITY) ? 0.f : M;
                const float w0 = fast_exp2(m - mref);
                float wg[3];
                l *= w0;
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    wg[w] = fast_exp2(p0[w * WS] - mref);
                    l += p0[w * WS + 1] * wg[w];
                }
                m = M;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int d = min(dt * 32 + acc_row(i, h), DH - 1);
                        float v = O[dt][i] * w0;
#pragma unroll
                        for (int w = 0; w < 3; ++w) v += p0[w * WS + 2 + d] * wg[w];
                        O[dt][i] = v;
                    }
            }
        }
        if (q < S) {
            const float inv = ks / l;
            const long out = ((long)n * S + q) * E + hd * DH;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * h;
                    if (d < DH) {
                        f32x4 v = {O[dt][4 * g] * inv, O[dt][4 * g + 1] * inv, O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv};
                        store4<IOB>(ctx, out + d, v);
                    }
                }
            if (h == 0) lse2[(long)blockIdx.x * S + q] = m + log2f(l);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Row-major-only variants (round 2).  The kernels above keep a TRANSPOSED copy of V (forward) / K (dQ) in LDS for the
// "accumulator tile as B operand" product; gfx950's transposing LDS read (frag_tr_rows below, the dK/dV kernel's idiom) takes
// that fragment out of the ROW-MAJOR image instead.  What that buys:
//   * one staging pass less, all of it 16-byte coalesced loads, and EVERY load of the workgroup - K, V, the mask and the first
//     query tile - is in flight before the first LDS write: one global round trip per workgroup where the kernels above made
//     nine (forward) to twenty (dQ) serialised ones (43 % / 62 % of their wave time was spent waiting on memory);
//   * only the rows up to S rounded to 8 are staged (reads of the last key tile are clamped onto a zero row), so K + V of
//     S = 257 / dh = 64 take 76 KB: two workgroups per CU without chunking the keys in the dQ kernel;
//   * NW = 8 waves per workgroup: four waves per SIMD hide the LDS / MFMA / exp latencies of each other.
// Arithmetic, accumulation order inside a (query tile, key tile) pair and the dropout stream are those of the kernels above.
// ------------------------------------------------------------------------------------------------------
typedef short s16x4r __attribute__((ext_vector_type(4)));
// rows staged by the row-major kernels: S rounded up so that - unless S is a multiple of 32 - at least ONE zero row follows the
// data (row R - 1 >= S).  Reads of the last 32-row tile are clamped onto that row: finite for the operands that meet a zero
// probability anyway (K, V rows past the end) and exactly zero where the dK/dV kernel relies on it (Q, dO rows past the end).
__host__ __device__ __forceinline__ int rm_rows(int S) {
    const int Sp = (S + 31) / 32 * 32, r = (S + 8) / 8 * 8;
    return r < Sp ? r : Sp;
}
// Per-lane LDS element offsets of one 32-row tile's reads, computed once per kernel: the four transposing reads
// (element j of lane (c, h) of fragment s2 = img[row0 + 16*s2 + 8*(j>>2) + 4*h + (j&3)][col0 + c]) and the row-major fragment row
// (row0 + c, column 8*h).  Per tile they cost one add each; only a tile that reaches past the staged rows takes the clamping path
// (wave-uniform branch).  The address arithmetic used to be a fifth of the VALU work of a tile step.
struct TileOff {
    int tr[4], rm;             // [2*s2 + b]: rows 16*s2 + 8*b + 4*hh + (i>>2); rm: row c
    int ld, rlim;              // uniform
    __device__ __forceinline__ void init(int ld_, int rlim_, int lane) {
        const int i = lane & 15, grp = lane >> 4;
        ld = ld_; rlim = rlim_;
        const int rr = 4 * (grp >> 1) + (i >> 2), co = 16 * (grp & 1) + 4 * (i & 3);
#pragma unroll
        for (int k = 0; k < 4; ++k) tr[k] = (rr + 8 * k) * ld + co;
        rm = (lane & 31) * ld + 8 * (lane >> 5);
    }
    __device__ __forceinline__ void tile(int row0, int lane, int (&t)[4], int& r) const {
        if (row0 + 31 > rlim) {        // the lane-derived pieces are recomputed here rather than kept in registers
            const int i = lane & 15, grp = lane >> 4;
            const int rr = 4 * (grp >> 1) + (i >> 2), co = 16 * (grp & 1) + 4 * (i & 3);
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = min(row0 + rr + 8 * k, rlim) * ld + co;
            r = min(row0 + (lane & 31), rlim) * ld + 8 * (lane >> 5);
        } else {
            const int o = row0 * ld;
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = tr[k] + o;
            r = rm + o;
        }
    }
};
__device__ __forceinline__ bf16x8 frag_tr_at(const __bf16* img, const int (&t)[4], int s2, int col0) {
    typedef __attribute__((address_space(3))) s16x4r lds_s16x4;
    const s16x4r a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + t[2 * s2] + col0));
    const s16x4r b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + t[2 * s2 + 1] + col0));
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
// the same fragment with the addresses computed in place (no registers held across tiles: the 8-wave forward kernel has none to spare)
__device__ __forceinline__ bf16x8 frag_tr_rows(const __bf16* img, int ld, int row0, int rlim, int col0, int s2, int lane) {
    const int i = lane & 15, grp = lane >> 4;
    const int ra = row0 + 16 * s2 + 4 * (grp >> 1) + (i >> 2);
    const int co = col0 + 16 * (grp & 1) + 4 * (i & 3);
    typedef __attribute__((address_space(3))) s16x4r lds_s16x4;
    const s16x4r a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + min(ra, rlim) * ld + co));
    const s16x4r b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + min(ra + 8, rlim) * ld + co));
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
// K and V row-major images of R = ceil8(S) rows (rows >= S zero): all loads of a pass in flight before its first LDS write.
// `mid` runs once, uniformly in every thread, between the first pass's loads and its LDS writes: work that consumes OTHER loads
// issued earlier (the key flags) goes there, so that it does not put a wait in front of these loads.
template <int DH, bool IOB, int NT, int U, typename F>
__device__ __forceinline__ void stage_two_rows(__bf16* As, __bf16* Bs, const void* XA, long aoff, long lda, const void* XB, long boff, long ldb,
                                               int S, int R, int tid, F&& mid) {
    constexpr int LD = DH + 8, CPR = DH / 8;
    const int nchunk = R * CPR;
    const int rmax = max(S - 1, 0);
    for (int c0 = 0; c0 < nchunk; c0 += U * NT) {          // uniform trip count
        bf16x8 aw[U], bw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = min(c0 + tid + u * NT, nchunk - 1);
            const long row = min(cc / CPR, rmax);
            aw[u] = load_frag8<IOB>(XA, aoff + row * lda + 8 * (cc % CPR));
            bw[u] = load_frag8<IOB>(XB, boff + row * ldb + 8 * (cc % CPR));
        }
        if (c0 == 0) mid();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c0 + tid + u * NT;
            if (cc < nchunk) {
                const int row = cc / CPR, c8 = cc % CPR;
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<bf16x8*>(As + row * LD + 8 * c8) = row < S ? aw[u] : z;
                *reinterpret_cast<bf16x8*>(Bs + row * LD + 8 * c8) = row < S ? bw[u] : z;
            }
        }
    }
}
template <int DH, bool IOB, int NT, int U, typename F>
__device__ __forceinline__ void stage_kv_rows(__bf16* Ks, __bf16* Vs, const void* X, long koff, long voff, long ldx, int S, int R, int tid, F&& mid) {
    stage_two_rows<DH, IOB, NT, U>(Ks, Vs, X, koff, ldx, X, voff, ldx, S, R, tid, mid);
}
// Key flags Ms[Sp] (1 = masked or past the end) and per-tile "any flag" Mt (one ballot per 64 keys).  The mask bytes of the
// first MASK_U * NT keys are loaded by mask_request (unconditionally, clamped) and consumed by mask_flags later.
constexpr int MASK_U = 2;
template <int NT>
__device__ __forceinline__ void mask_request(uint8_t (&mf)[MASK_U], const uint8_t* mask, long moff, int S, int tid) {
#pragma unroll
    for (int u = 0; u < MASK_U; ++u) mf[u] = mask ? mask[moff + min(tid + u * NT, max(S - 1, 0))] : (uint8_t)0;
}
template <int NT>
__device__ __forceinline__ void mask_flags(uint8_t* Ms, uint8_t* Mt, const uint8_t (&mf)[MASK_U], const uint8_t* mask, long moff, int S, int Sp, int tid) {
    for (int u = 0; u * NT < Sp; ++u) {                    // uniform trip count
        const int i = u * NT + tid;
        uint8_t f = 0;
        if (u < MASK_U) f = u == 0 ? mf[0] : mf[1];
        else if (i < S && mask) f = mask[moff + i];
        if (i >= S) f = i < Sp ? 1 : 0;
        if (i < Sp) Ms[i] = f;
        const uint64_t b = __builtin_amdgcn_ballot_w64(f != 0);
        if ((tid & 31) == 0 && i < Sp) Mt[i >> 5] = (tid & 32) ? ((b >> 32) != 0) : ((uint32_t)b != 0);
    }
}

template <int DH, bool IOB, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void attn_fwd_rm_kernel(const void* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                                  int mask_B, void* __restrict__ ctx, float* __restrict__ lse2,
                                                                  int S, int E, int nh, DropKey drop_in, int qkv_B) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = 64 * NW;
    const int Sp = (S + 31) / 32 * 32, R = rm_rows(S);
    constexpr int LDK = DH + 8;
    __bf16* Ks = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Vs = Ks + R * LDK;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vs + R * LDK);
    uint8_t* Mt = Ms + Sp;
    float* Co = reinterpret_cast<float*>(Mt + 64);        // [waves 1..NW-1][CO_MAXQ][DH + 2] partials of the shared query tile
    constexpr int DT = (DH + 31) / 32;
    constexpr int KS = DH / 16;

    const int n = blockIdx.x / nh, hd = blockIdx.x % nh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the tile loops and their branches run on the SALU
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;
    const int nqt = Sp / 32, nkt = Sp / 32;
    const bool coop = (nqt % NW == 1) && nqt > 1 && (S - 32 * (nqt - 1)) <= CO_MAXQ;      // workgroup-uniform
    const int nq_main = coop ? nqt - 1 : nqt;
    const int n_items = nq_main + (coop ? NW : 0);

    // the query fragments of a wave's NEXT tile are requested before the current one is multiplied
    bf16x8 qn[KS];
    auto request = [&](int it) {
        const int qt = it >= nq_main ? nqt - 1 : it;
        const int qc = min(qt * 32 + c, S - 1);        // rows past the end re-read the last query: never stored
#pragma unroll
        for (int s = 0; s < KS; ++s) qn[s] = load_frag8<IOB>(qkv, base + (long)qc * ld + 16 * s + 8 * h);
    };
    if (wave < n_items) request(wave);
    const long moff = (long)(n % mask_B) * S;
    uint8_t mf[MASK_U];
    mask_request<NT>(mf, mask, moff, S, tid);
    stage_kv_rows<DH, IOB, NT, (NW == 8 ? 5 : 9)>(Ks, Vs, qkv, base + E, base + 2 * E, ld, S, R, tid,
                                                   [&]() { mask_flags<NT>(Ms, Mt, mf, mask, moff, S, Sp, tid); });
    __syncthreads();

    const float sc = rsqrtf((float)DH) * LOG2E;     // folded into the exponent's multiply-add
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    constexpr bool TOFF = NW < 8;                   // precomputed tile offsets cost five registers
    TileOff toff;
    if constexpr (TOFF) toff.init(LDK, R - 1, lane);
    for (int it = wave; it < n_items; it += NW) {
        const bool shared = it >= nq_main;             // the left-over tile: every wave passes here exactly once
        const int qt = shared ? nqt - 1 : it;
        const int kt0 = shared ? wave : 0, kstep = shared ? NW : 1;
        const int q = qt * 32 + c;
        bf16x8 qf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = qn[s];
        if (it + NW < n_items) request(it + NW);
        float m = -INFINITY, l = 0.f;
        f32x16 O[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
        const uint32_t srow = drop_state(drop, (((uint64_t)blockIdx.x * S + (uint64_t)q) * (uint64_t)drop_attn_ld(S)) >> 1) + (uint32_t)(2 * h) * DROP_PHI;
        for (int kt = kt0; kt < nkt; kt += kstep) {
            f32x16 s16;
#pragma unroll
            for (int i = 0; i < 16; ++i) s16[i] = 0.f;
            int tro[4], kro;
            if constexpr (TOFF) toff.tile(kt * 32, lane, tro, kro);
            else kro = min(kt * 32 + c, R - 1) * LDK + 8 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + kro + 16 * s);
                s16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s16, 0, 0, 0);
            }
            // TOFF variants: maximum of the RAW scores (the scale is positive), the scale itself rides in the exponent's multiply-add
            float mt = -INFINITY;
            if (__builtin_amdgcn_readfirstlane((int)Mt[kt])) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 32 + acc_row(i, h);
                    s16[i] = Ms[key] ? -INFINITY : (TOFF ? s16[i] : s16[i] * sc);
                    mt = fmaxf(mt, s16[i]);
                }
            } else {                    // tile without masked keys: no per-element mask lookups
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if constexpr (!TOFF) s16[i] *= sc;
                    mt = fmaxf(mt, s16[i]);
                }
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            if constexpr (TOFF) mt *= sc;
            const bool move = mt > m + 8.f;            // lazy reference update, as in attn_fwd_kernel
            if (__builtin_amdgcn_ballot_w64(move) != 0) {
                const float mn = move ? mt : m;
                const float alpha = (m == -INFINITY) ? 0.f : fast_exp2(m - mn);
                l *= alpha;
                m = mn;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
            }
            const float nmref = (m == -INFINITY) ? 0.f : -m;
            float lt = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = TOFF ? fast_exp2(__builtin_fmaf(s16[i], sc, nmref)) : fast_exp2(s16[i] + nmref);
                lt += p;
                s16[i] = p;
            }
            lt += __shfl_xor(lt, 32, 64);
            l += lt;
            if (drop.p > 0.f) {
                const uint32_t skt = srow + (uint32_t)(kt * 16) * DROP_PHI;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t b0 = drop_bits(skt + (uint32_t)(4 * g) * DROP_PHI), b1 = drop_bits(skt + (uint32_t)(4 * g + 1) * DROP_PHI);
                    s16[4 * g + 0] = drop_keep_even(b0, drop.thr) ? s16[4 * g + 0] : 0.f;
                    s16[4 * g + 1] = drop_keep_odd(b0, drop.thr) ? s16[4 * g + 1] : 0.f;
                    s16[4 * g + 2] = drop_keep_even(b1, drop.thr) ? s16[4 * g + 2] : 0.f;
                    s16[4 * g + 3] = drop_keep_odd(b1, drop.thr) ? s16[4 * g + 3] : 0.f;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = frag_from_acc(s16, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 vf = TOFF ? frag_tr_at(Vs, tro, s2, dt * 32) : frag_tr_rows(Vs, LDK, kt * 32, R - 1, dt * 32, s2, lane);
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, O[dt], 0, 0, 0);
                }
            }
        }
        if (shared) {      // merge the NW key-range partials of the shared tile (softmax-weighted) in wave 0
            if (wave != 0 && q < S) {
                float* part = Co + ((wave - 1) * CO_MAXQ + min(c, CO_MAXQ - 1)) * (DH + 2);
                if (h == 0) { part[0] = m; part[1] = l; }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int d = dt * 32 + acc_row(i, h);
                        if (d < DH) part[2 + d] = O[dt][i];
                    }
            }
            __syncthreads();
            if (wave != 0) continue;
            if (q < S) {
                const float* p0 = Co + min(c, CO_MAXQ - 1) * (DH + 2);
                constexpr int WS = CO_MAXQ * (DH + 2);
                float M = m;
#pragma unroll
                for (int w = 0; w < NW - 1; ++w) M = fmaxf(M, p0[w * WS]);
                const float mref = (M == -INFINITY) ? 0.f : M;
                const float w0 = fast_exp2(m - mref);
                l *= w0;
                m = M;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) O[dt][i] *= w0;
#pragma unroll 1
                for (int w = 0; w < NW - 1; ++w) {          // one partial at a time: 32 LDS values in flight, not 32 * (NW - 1)
                    const float wgt = fast_exp2(p0[w * WS] - mref);
                    l += p0[w * WS + 1] * wgt;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) O[dt][i] += p0[w * WS + 2 + min(dt * 32 + acc_row(i, h), DH - 1)] * wgt;
                }
            }
        }
        if (q < S) {
            const float inv = ks / l;
            const long out = ((long)n * S + q) * E + hd * DH;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * h;
                    if (d < DH) {
                        f32x4 v = {O[dt][4 * g] * inv, O[dt][4 * g + 1] * inv, O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv};
                        store4<IOB>(ctx, out + d, v);
                    }
                }
            if (h == 0) lse2[(long)blockIdx.x * S + q] = m + log2f(l);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// forward, long sequences (S above the LDS-resident range of attn_fwd_kernel, e.g. 1 025 patch tokens): grid.y splits
// the query tiles four at a time (one per wave) and the keys stream through LDS in chunks of CK; the online softmax
// state of a wave's query tile lives in registers across the chunks.  Same arithmetic and dropout stream as above.
// ------------------------------------------------------------------------------------------------------
template <int DH, bool IOB>
__global__ __launch_bounds__(256, 2) void attn_fwd_long_kernel(const void* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                       int mask_B, void* __restrict__ ctx, float* __restrict__ lse2,
                                                       int S, int E, int nh, DropKey drop_in, int qkv_B, int CK, int npairs, int nqg) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Sp = (S + 31) / 32 * 32;
    constexpr int LDK = DH + 8;
    const int LDV = CK + TPAD;                     // one chunk of CK keys (multiple of 32) is resident at a time
    __bf16* Ks = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Vt = Ks + CK * LDK;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vt + DH * LDV);
    constexpr int DT = (DH + 31) / 32;          // head-dim tiles of the O^T accumulator
    constexpr int KS = DH / 16;                 // k-steps of the score product

    // XCD-aware order: workgroup b runs on XCD b % 8; the nqg query groups of one (sample, head) pair are consecutive
    // slots of ONE XCD, so the pair's K / V chunks are re-read from that XCD's L2, not from HBM
    const int pair = ((int)(blockIdx.x >> 3) / nqg) * 8 + (int)(blockIdx.x & 7);
    const int qg = (int)(blockIdx.x >> 3) % nqg;
    if (pair >= npairs) return;
    const int n = pair / nh, hd = pair % nh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;      // this (sample, head) inside qkv; replicas may share one projection
    for (int i = tid; i < Sp; i += 256) Ms[i] = (i >= S) ? 1 : (mask ? mask[(long)(n % mask_B) * S + i] : 0);
    __syncthreads();
    uint8_t* Mt = Ms + Sp;                                 // per key tile: does it hold any masked / padded key?
    for (int t = tid; t < Sp / 32; t += 256) {
        uint8_t any = 0;
        for (int j = 0; j < 32; ++j) any |= Ms[t * 32 + j];
        Mt[t] = any;
    }
    __syncthreads();

    const float sc = rsqrtf((float)DH) * LOG2E;     // applied to the fp32 scores after the MFMA
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    const int nqt = Sp / 32;
    // one query tile per wave: tile 4 * (query group) + wave; waves past the end still take part in the staging barriers
    const int qt = 4 * qg + wave;
    const bool active = qt < nqt;
    {
        const int q = qt * 32 + c;
        bf16x8 qf[KS];
        const int qc = min(max(q, 0), S - 1);          // rows past the end re-read the last query: never stored
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = load_frag8<IOB>(qkv, base + (long)qc * ld + 16 * s + 8 * h);
        float m = -INFINITY, l = 0.f;
        f32x16 O[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
        // dropout stream state of this lane's row at its first key pair (drop_rng.h: linear in the pair index)
        const uint32_t srow = drop_state(drop, (((uint64_t)pair * S + (uint64_t)q) * (uint64_t)drop_attn_ld(S)) >> 1) + (uint32_t)(2 * h) * DROP_PHI;
        for (int kbase = 0; kbase < Sp; kbase += CK) {
        const int krows = min(CK, Sp - kbase);             // staged rows of this chunk (multiple of 32)
        __syncthreads();                                   // the previous chunk's readers are done (first pass: Mt is written)
        stage_rows<DH, IOB>(Ks, qkv, base + E + (long)kbase * ld, ld, max(0, min(S - kbase, krows)), krows, tid, 256);
        stage_transposed<DH, IOB>(Vt, qkv, base + 2 * E + (long)kbase * ld, ld, max(0, min(S - kbase, krows)), CK, tid, 256);
        __syncthreads();
        if (active)
        for (int kl = 0; kl < krows / 32; ++kl) {
            const int kt = kbase / 32 + kl;
            f32x16 s16;
#pragma unroll
            for (int i = 0; i < 16; ++i) s16[i] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kl * 32 + c) * LDK + 16 * s + 8 * h);
                s16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s16, 0, 0, 0);
            }
            float mt = -INFINITY;
            if (__builtin_amdgcn_readfirstlane((int)Mt[kt])) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 32 + acc_row(i, h);
                    s16[i] = Ms[key] ? -INFINITY : s16[i] * sc;
                    mt = fmaxf(mt, s16[i]);
                }
            } else {                    // tile without masked keys: no per-element mask lookups
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s16[i] *= sc;
                    mt = fmaxf(mt, s16[i]);
                }
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            // Lazy reference update: the softmax reference m only moves when a tile's maximum exceeds it by more than
            // 2^8 (scores are in the log2 domain), so p <= 256 in between - exact after the final O / l - and the
            // O *= alpha pass over the 32 output accumulators is skipped for almost every tile.
            const bool move = mt > m + 8.f;
            if (__builtin_amdgcn_ballot_w64(move) != 0) {
                const float mn = move ? mt : m;
                const float alpha = (m == -INFINITY) ? 0.f : fast_exp2(m - mn);      // mn is finite wherever move is set
                l *= alpha;
                m = mn;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
            }
            const float mref = (m == -INFINITY) ? 0.f : m;
            float lt = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = fast_exp2(s16[i] - mref);
                lt += p;
                s16[i] = p;
            }
            lt += __shfl_xor(lt, 32, 64);
            l += lt;
            if (drop.p > 0.f) {     // registers 4g..4g+3 hold 4 consecutive keys: two pair hashes
                const uint32_t skt = srow + (uint32_t)(kt * 16) * DROP_PHI;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t b0 = drop_bits(skt + (uint32_t)(4 * g) * DROP_PHI), b1 = drop_bits(skt + (uint32_t)(4 * g + 1) * DROP_PHI);
                    s16[4 * g + 0] = drop_keep_even(b0, drop.thr) ? s16[4 * g + 0] : 0.f;
                    s16[4 * g + 1] = drop_keep_odd(b0, drop.thr) ? s16[4 * g + 1] : 0.f;
                    s16[4 * g + 2] = drop_keep_even(b1, drop.thr) ? s16[4 * g + 2] : 0.f;
                    s16[4 * g + 3] = drop_keep_odd(b1, drop.thr) ? s16[4 * g + 3] : 0.f;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = frag_from_acc(s16, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 vf = frag_transposed(Vt, LDV, min(dt * 32 + c, DH - 1), kl * 32, s2, h);
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, O[dt], 0, 0, 0);
                }
            }
        }
        }   // key chunks
        if (active && q < S) {
            const float inv = ks / l;
            const long out = ((long)n * S + q) * E + hd * DH;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * h;
                    if (d < DH) {
                        f32x4 v = {O[dt][4 * g] * inv, O[dt][4 * g + 1] * inv, O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv};
                        store4<IOB>(ctx, out + d, v);
                    }
                }
            if (h == 0) lse2[(long)pair * S + q] = m + log2f(l);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Long sequences, row-major streaming variants (round 2).  The *_long kernels above stage every key chunk with several
// serialised global round trips between two barriers and give a wave ONE query tile: at S = 1 025 they ran 4 - 5x above the
// VALU time of their tile steps (image-transformer config: 45 of 84 ms per step).  Here
//   * chunks of CK = 128 rows of two row-major operands (K and V; Q and dO in the dK/dV kernel) travel global -> registers ->
//     one of TWO LDS buffers: chunk c + 2 is requested and chunk c + 1 written to the other buffer BEFORE chunk c is multiplied,
//     so a chunk's loads have a whole chunk of MFMA / softmax work to land and there is one barrier per chunk;
//   * a wave owns QT = 2 tiles of the resident side (query tiles in forward / dQ, key tiles in dK/dV): 8 tiles per workgroup
//     halve the number of passes over the streamed operands, and every LDS fragment feeds two MFMAs;
//   * transposed operands through ds_read_b64_tr_b16, tile offsets precomputed, wave index scalar (see the *_rm kernels).
// Same arithmetic and dropout stream as the kernels above.
// ------------------------------------------------------------------------------------------------------
constexpr int SCK = 128;                                    // streamed rows per chunk
constexpr int SQT = 2;                                      // resident tiles per wave
// registers of one chunk in flight: SCK rows x DH columns of two operands over 256 threads
template <int DH, bool IOB, int CK = SCK, int NT = 256>
struct ChunkRegs {
    static constexpr int CPR = DH / 8, NP = CK * CPR / NT > 0 ? CK * CPR / NT : 1;
    bf16x8 a[NP], b[NP];
    // rows [row0, row0 + SCK) of A (XA + aoff, stride lda) and B; rows >= S are clamped for the load and zero-filled by store()
    __device__ __forceinline__ void load(const void* XA, long aoff, long lda, const void* XB, long boff, long ldb, int row0, int S, int tid) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int p = min(tid + NT * u, CK * CPR - 1);
            const long row = min(row0 + p / CPR, S - 1);
            a[u] = load_frag8<IOB>(XA, aoff + row * lda + 8 * (p % CPR));
            b[u] = load_frag8<IOB>(XB, boff + row * ldb + 8 * (p % CPR));
        }
    }
    __device__ __forceinline__ void store(__bf16* As, __bf16* Bs, int row0, int S, int tid) const {
        constexpr int LD = DH + 8;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int p = tid + NT * u;
            if (p < CK * CPR) {
                const int row = p / CPR, c8 = p % CPR;
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<bf16x8*>(As + row * LD + 8 * c8) = row0 + row < S ? a[u] : z;
                *reinterpret_cast<bf16x8*>(Bs + row * LD + 8 * c8) = row0 + row < S ? b[u] : z;
            }
        }
    }
};
size_t stream_smem(int S, int DH) {                         // two buffers x two operands + key flags (forward / dQ)
    const int Sp = (S + 31) / 32 * 32;
    return (size_t)4 * SCK * (DH + 8) * 2 + Sp + 64 + 16;
}

template <int DH, bool IOB, int NW, int QT>
__global__ __launch_bounds__(64 * NW, NW / 2) void attn_fwd_stream_kernel(const void* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                                 int mask_B, void* __restrict__ ctx, float* __restrict__ lse2,
                                                                 int S, int E, int nh, DropKey drop_in, int qkv_B, int npairs, int nqg) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Sp = (S + 31) / 32 * 32, R = rm_rows(S);
    constexpr int LDK = DH + 8, IMG = SCK * LDK;
    __bf16* Kb = reinterpret_cast<__bf16*>(smem_raw);           // [2][IMG]
    __bf16* Vb = Kb + 2 * IMG;                                  // [2][IMG]
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vb + 2 * IMG);
    uint8_t* Mt = Ms + Sp;
    constexpr int DT = (DH + 31) / 32;
    constexpr int KS = DH / 16;

    // XCD-aware order (see attn_fwd_long_kernel): the query groups of one (sample, head) pair share an XCD's L2
    const int pair = ((int)(blockIdx.x >> 3) / nqg) * 8 + (int)(blockIdx.x & 7);
    const int qg = (int)(blockIdx.x >> 3) % nqg;
    if (pair >= npairs) return;
    const int n = pair / nh, hd = pair % nh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;
    const long moff = (long)(n % mask_B) * S;
    const int nkt = Sp / 32;
    const int nchunks = (R + SCK - 1) / SCK;

    // this wave's query tiles: NW * QT * qg + wave + NW * j (tiles past the end recompute the last row and store nothing)
    bf16x8 qf[QT][KS];
    int q[QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
        q[j] = (NW * QT * qg + wave + NW * j) * 32 + c;
        const int qc = min(q[j], S - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[j][s] = load_frag8<IOB>(qkv, base + (long)qc * ld + 16 * s + 8 * h);
    }
    uint8_t mf[MASK_U];
    mask_request<64 * NW>(mf, mask, moff, S, tid);
    ChunkRegs<DH, IOB, SCK, 64 * NW> cr;
    cr.load(qkv, base + E, ld, qkv, base + 2 * E, ld, 0, S, tid);
    mask_flags<64 * NW>(Ms, Mt, mf, mask, moff, S, Sp, tid);
    cr.store(Kb, Vb, 0, S, tid);
    if (nchunks > 1) cr.load(qkv, base + E, ld, qkv, base + 2 * E, ld, SCK, S, tid);
    __syncthreads();

    const float sc = rsqrtf((float)DH) * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    float m[QT], l[QT];
    f32x16 O[QT][DT];
    uint32_t srow[QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
        m[j] = -INFINITY; l[j] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[j][dt][i] = 0.f;
        srow[j] = drop_state(drop, (((uint64_t)pair * S + (uint64_t)q[j]) * (uint64_t)drop_attn_ld(S)) >> 1) + (uint32_t)(2 * h) * DROP_PHI;
    }
    TileOff toff;
    toff.init(LDK, SCK - 1, lane);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) cr.store(Kb + (buf ^ 1) * IMG, Vb + (buf ^ 1) * IMG, (ch + 1) * SCK, S, tid);
        if (ch + 2 < nchunks) cr.load(qkv, base + E, ld, qkv, base + 2 * E, ld, (ch + 2) * SCK, S, tid);
        const __bf16* Ks = Kb + buf * IMG;
        const __bf16* Vs = Vb + buf * IMG;
        toff.rlim = min(SCK, R - ch * SCK) - 1;
        const int kt_end = min(nkt, (ch + 1) * (SCK / 32));
        for (int kt = ch * (SCK / 32); kt < kt_end; ++kt) {
            int tro[4], kro;
            toff.tile((kt - ch * (SCK / 32)) * 32, lane, tro, kro);
            f32x16 s16[QT];
#pragma unroll
            for (int j = 0; j < QT; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) s16[j][i] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + kro + 16 * s);
#pragma unroll
                for (int j = 0; j < QT; ++j) s16[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[j][s], s16[j], 0, 0, 0);
            }
            const bool masked = __builtin_amdgcn_readfirstlane((int)Mt[kt]) != 0;
#pragma unroll
            for (int j = 0; j < QT; ++j) {
                float mt = -INFINITY;
                if (masked) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kt * 32 + acc_row(i, h);
                        s16[j][i] = Ms[key] ? -INFINITY : s16[j][i] * sc;
                        mt = fmaxf(mt, s16[j][i]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        s16[j][i] *= sc;
                        mt = fmaxf(mt, s16[j][i]);
                    }
                }
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const bool move = mt > m[j] + 8.f;         // lazy reference update, as in attn_fwd_kernel
                if (__builtin_amdgcn_ballot_w64(move) != 0) {
                    const float mn = move ? mt : m[j];
                    const float alpha = (m[j] == -INFINITY) ? 0.f : fast_exp2(m[j] - mn);
                    l[j] *= alpha;
                    m[j] = mn;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) O[j][dt][i] *= alpha;
                }
                const float mref = (m[j] == -INFINITY) ? 0.f : m[j];
                float lt = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = fast_exp2(s16[j][i] - mref);
                    lt += p;
                    s16[j][i] = p;
                }
                lt += __shfl_xor(lt, 32, 64);
                l[j] += lt;
                if (drop.p > 0.f) {
                    const uint32_t skt = srow[j] + (uint32_t)(kt * 16) * DROP_PHI;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint32_t b0 = drop_bits(skt + (uint32_t)(4 * g) * DROP_PHI), b1 = drop_bits(skt + (uint32_t)(4 * g + 1) * DROP_PHI);
                        s16[j][4 * g + 0] = drop_keep_even(b0, drop.thr) ? s16[j][4 * g + 0] : 0.f;
                        s16[j][4 * g + 1] = drop_keep_odd(b0, drop.thr) ? s16[j][4 * g + 1] : 0.f;
                        s16[j][4 * g + 2] = drop_keep_even(b1, drop.thr) ? s16[j][4 * g + 2] : 0.f;
                        s16[j][4 * g + 3] = drop_keep_odd(b1, drop.thr) ? s16[j][4 * g + 3] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf[QT];
#pragma unroll
                for (int j = 0; j < QT; ++j) pf[j] = frag_from_acc(s16[j], s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 vf = frag_tr_at(Vs, tro, s2, dt * 32);
#pragma unroll
                    for (int j = 0; j < QT; ++j) O[j][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[j], O[j][dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < QT; ++j) {
        if (q[j] < S) {
            const float inv = ks / l[j];
            const long out = ((long)n * S + q[j]) * E + hd * DH;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * h;
                    if (d < DH) {
                        f32x4 v = {O[j][dt][4 * g] * inv, O[j][dt][4 * g + 1] * inv, O[j][dt][4 * g + 2] * inv, O[j][dt][4 * g + 3] * inv};
                        store4<IOB>(ctx, out + d, v);
                    }
                }
            if (h == 0) lse2[(long)pair * S + q[j]] = m[j] + log2f(l[j]);
        }
    }
}

// delta[n,h,q] = sum_d dO[n,q,h*DH+d] * O[n,q,h*DH+d]
__global__ void attn_delta_kernel(const float* __restrict__ dO, const float* __restrict__ O, float* __restrict__ delta,
                                  long rows, int S, int E, int nh) {
    const int dh = E / nh;
    const long total = rows * nh;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / nh;
        const int hd = (int)(i % nh);
        const float* a = dO + row * E + hd * dh;
        const float* b = O + row * E + hd * dh;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s += a[d] * b[d];
        const long n = row / S, q = row % S;
        delta[(n * nh + hd) * S + q] = s;
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, dQ: same orientation as forward.  LDS: K row-major, V row-major, K transposed.
// ------------------------------------------------------------------------------------------------------
template <int DH, bool IOB>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const void* __restrict__ qkv, const void* __restrict__ ctx,
                                                          const void* __restrict__ dctx,
                                                          const float* __restrict__ lse2, float* __restrict__ delta,
                                                          const uint8_t* __restrict__ mask, int mask_B,
                                                          void* __restrict__ dqkv, int S, int E, int nh, DropKey drop_in, int qkv_B) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Sp = (S + 31) / 32 * 32;
    constexpr int LDK = DH + 8;
    const int LDT = Sp + TPAD;
    __bf16* Ks = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Vs = Ks + Sp * LDK;
    __bf16* Kt = Vs + Sp * LDK;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Kt + DH * LDT);
    constexpr int DT = (DH + 31) / 32;
    constexpr int KS = DH / 16;

    const int n = blockIdx.x / nh, hd = blockIdx.x % nh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;
    stage_rows<DH, IOB>(Ks, qkv, base + E, ld, S, Sp, tid, 256);
    stage_rows<DH, IOB>(Vs, qkv, base + 2 * E, ld, S, Sp, tid, 256);
    stage_transposed<DH, IOB>(Kt, qkv, base + E, ld, S, Sp, tid, 256);
    for (int i = tid; i < Sp; i += 256) Ms[i] = (i >= S) ? 1 : (mask ? mask[(long)(n % mask_B) * S + i] : 0);
    __syncthreads();
    uint8_t* Mt = Ms + Sp;                                 // per key tile: does it hold any masked / padded key?
    float* Co = reinterpret_cast<float*>(Mt + 64);        // [waves 1..3][CO_MAXQ][DH + 2] partials of the shared query tile
    for (int t = tid; t < Sp / 32; t += 256) {
        uint8_t any = 0;
        for (int j = 0; j < 32; ++j) any |= Ms[t * 32 + j];
        Mt[t] = any;
    }
    __syncthreads();

    const float scale = rsqrtf((float)DH);
    const float sc = scale * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    const int nqt = Sp / 32, nkt = Sp / 32;
    const bool coop = (nqt % 4 == 1) && nqt > 1 && (S - 32 * (nqt - 1)) <= CO_MAXQ;      // see CO_MAXQ
    const int nq_main = coop ? nqt - 1 : nqt;
    for (int it = wave; it < nq_main + (coop ? 4 : 0); it += 4) {
        const bool shared = it >= nq_main;
        const int qt = shared ? nqt - 1 : it;
        const int kt0 = shared ? wave : 0, kstep = shared ? 4 : 1;
        const int q = qt * 32 + c;
        bf16x8 qf[KS], df[KS];
        float L2 = 0.f, dl = 0.f;
        if (q < S) L2 = lse2[(long)blockIdx.x * S + q];
        const int qc = min(q, S - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const long off = ((long)n * S + qc) * E + hd * DH + 16 * s + 8 * h;
            qf[s] = load_frag8<IOB>(qkv, base + (long)qc * ld + 16 * s + 8 * h);
            df[s] = load_frag8<IOB>(dctx, off);
            float dv[8], ov[8];
            load_f32x8<IOB>(dctx, off, dv);
            load_f32x8<IOB>(ctx, off, ov);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += dv[j] * ov[j];
        }
        // delta[q] = sum_d dO*O : this lane-half covered half of the head dim, the other half the rest
        dl += __shfl_xor(dl, 32, 64);
        if (q < S && h == 0 && (!shared || wave == 0)) delta[(long)blockIdx.x * S + q] = dl;      // consumed by the dK/dV kernel
        f32x16 dQ[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) dQ[dt][i] = 0.f;
        const uint32_t srow = drop_state(drop, (((uint64_t)blockIdx.x * S + (uint64_t)q) * (uint64_t)drop_attn_ld(S)) >> 1) + (uint32_t)(2 * h) * DROP_PHI;
        for (int kt = kt0; kt < nkt; kt += kstep) {
            f32x16 s16, dp16;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s16[i] = 0.f; dp16[i] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + c) * LDK + 16 * s + 8 * h);
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + (kt * 32 + c) * LDK + 16 * s + 8 * h);
                s16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s16, 0, 0, 0);
                dp16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[s], dp16, 0, 0, 0);
            }
            if (drop.p > 0.f) {
                const uint32_t skt = srow + (uint32_t)(kt * 16) * DROP_PHI;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t b0 = drop_bits(skt + (uint32_t)(4 * g) * DROP_PHI), b1 = drop_bits(skt + (uint32_t)(4 * g + 1) * DROP_PHI);
                    dp16[4 * g + 0] = drop_keep_even(b0, drop.thr) ? dp16[4 * g + 0] * ks : 0.f;
                    dp16[4 * g + 1] = drop_keep_odd(b0, drop.thr) ? dp16[4 * g + 1] * ks : 0.f;
                    dp16[4 * g + 2] = drop_keep_even(b1, drop.thr) ? dp16[4 * g + 2] * ks : 0.f;
                    dp16[4 * g + 3] = drop_keep_odd(b1, drop.thr) ? dp16[4 * g + 3] * ks : 0.f;
                }
            }
            if (__builtin_amdgcn_readfirstlane((int)Mt[kt])) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 32 + acc_row(i, h);
                    const float p = Ms[key] ? 0.f : fast_exp2(s16[i] * sc - L2);
                    s16[i] = p * (dp16[i] - dl) * scale;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) s16[i] = fast_exp2(s16[i] * sc - L2) * (dp16[i] - dl) * scale;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 sf = frag_from_acc(s16, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 kf = frag_transposed(Kt, LDT, min(dt * 32 + c, DH - 1), kt * 32, s2, h);
                    dQ[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, sf, dQ[dt], 0, 0, 0);
                }
            }
        }
        if (shared) {      // sum the four key-range partials of the shared tile in wave 0
            if (wave != 0 && q < S) {
                float* part = Co + ((wave - 1) * CO_MAXQ + min(c, CO_MAXQ - 1)) * (DH + 2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int d = dt * 32 + acc_row(i, h);
                        if (d < DH) part[d] = dQ[dt][i];
                    }
            }
            __syncthreads();
            if (wave != 0) continue;
            if (q < S) {
                const float* p0 = Co + min(c, CO_MAXQ - 1) * (DH + 2);
                constexpr int WS = CO_MAXQ * (DH + 2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int d = min(dt * 32 + acc_row(i, h), DH - 1);
                        dQ[dt][i] += p0[d] + p0[WS + d] + p0[2 * WS + d];
                    }
            }
        }
        if (q < S) {
            const long out = ((long)n * S + q) * ld + hd * DH;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * h;
                    if (d < DH) {
                        f32x4 v = {dQ[dt][4 * g], dQ[dt][4 * g + 1], dQ[dt][4 * g + 2], dQ[dt][4 * g + 3]};
                        store4<IOB>(dqkv, out + d, v);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, dQ, key-chunked variant: the keys are staged in NCH = 2 chunks, so the LDS images are half as large and
// TWO workgroups fit a CU (the kernel is VALU-issue bound: a second wave per SIMD is worth ~1.4x).  A wave keeps
// the dQ accumulators of all its query tiles (at most DQ_SLOTS) across the chunks and reloads their Q / dO
// fragments per chunk (L2 hits).  Same arithmetic, same dropout stream, same shared left-over tile as above.
// ------------------------------------------------------------------------------------------------------
constexpr int DQ_SLOTS = 3;
template <int DH, bool IOB>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_kernel(const void* __restrict__ qkv, const void* __restrict__ ctx,
                                                              const void* __restrict__ dctx,
                                                              const float* __restrict__ lse2, float* __restrict__ delta,
                                                              const uint8_t* __restrict__ mask, int mask_B,
                                                              void* __restrict__ dqkv, int S, int E, int nh, DropKey drop_in, int qkv_B) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Sp = (S + 31) / 32 * 32;
    const int nqt = Sp / 32, nkt = Sp / 32;
    const int ckt = (nkt + 1) / 2;                  // key tiles per chunk
    const int CK = ckt * 32;                        // keys per chunk
    constexpr int LDK = DH + 8;
    const int LDT = CK + TPAD;
    __bf16* Ks = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Vs = Ks + CK * LDK;
    __bf16* Kt = Vs + CK * LDK;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Kt + DH * LDT);
    uint8_t* Mt = Ms + Sp;
    float* Co = reinterpret_cast<float*>(Mt + 64);
    constexpr int DT = (DH + 31) / 32;
    constexpr int KS = DH / 16;

    const int n = blockIdx.x / nh, hd = blockIdx.x % nh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;
    for (int i = tid; i < Sp; i += 256) Ms[i] = (i >= S) ? 1 : (mask ? mask[(long)(n % mask_B) * S + i] : 0);
    __syncthreads();
    for (int t = tid; t < nkt; t += 256) {
        uint8_t any = 0;
        for (int j = 0; j < 32; ++j) any |= Ms[t * 32 + j];
        Mt[t] = any;
    }

    const float scale = rsqrtf((float)DH);
    const float sc = scale * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    const bool coop = (nqt % 4 == 1) && nqt > 1 && (S - 32 * (nqt - 1)) <= CO_MAXQ;
    const int nq_main = coop ? nqt - 1 : nqt;
    const int n_items = nq_main + (coop ? 4 : 0);

    f32x16 dQ[DQ_SLOTS][DT];
    float L2s[DQ_SLOTS], dls[DQ_SLOTS];
#pragma unroll
    for (int sl = 0; sl < DQ_SLOTS; ++sl) {
        L2s[sl] = 0.f; dls[sl] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) dQ[sl][dt][i] = 0.f;
    }

    for (int ch = 0; ch < 2; ++ch) {
        const int kbase = ch * CK;                               // first key of the chunk
        const int krows = min(CK, Sp - kbase);                  // staged rows of the chunk (multiple of 32)
        __syncthreads();                                         // the previous chunk's readers are done (and Mt is written)
        // rows beyond S are zero-filled by the staging helpers (their S / Sp arguments are chunk-relative)
        stage_rows<DH, IOB>(Ks, qkv, base + E + (long)kbase * ld, ld, max(0, min(S - kbase, krows)), krows, tid, 256);
        stage_rows<DH, IOB>(Vs, qkv, base + 2 * E + (long)kbase * ld, ld, max(0, min(S - kbase, krows)), krows, tid, 256);
        stage_transposed<DH, IOB>(Kt, qkv, base + E + (long)kbase * ld, ld, max(0, min(S - kbase, krows)), CK, tid, 256);
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < DQ_SLOTS; ++sl) {
            const int it = wave + 4 * sl;
            if (it >= n_items) continue;
            const bool shared = it >= nq_main;
            const int qt = shared ? nqt - 1 : it;
            const int q = qt * 32 + c;
            const int qc = min(q, S - 1);
            bf16x8 qf[KS], df[KS];
            float dl = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const long off = ((long)n * S + qc) * E + hd * DH + 16 * s + 8 * h;
                qf[s] = load_frag8<IOB>(qkv, base + (long)qc * ld + 16 * s + 8 * h);
                df[s] = load_frag8<IOB>(dctx, off);
                if (ch == 0) {
                    float dv[8], ov[8];
                    load_f32x8<IOB>(dctx, off, dv);
                    load_f32x8<IOB>(ctx, off, ov);
#pragma unroll
                    for (int j = 0; j < 8; ++j) dl += dv[j] * ov[j];
                }
            }
            if (ch == 0) {
                dl += __shfl_xor(dl, 32, 64);
                dls[sl] = dl;
                L2s[sl] = q < S ? lse2[(long)blockIdx.x * S + q] : 0.f;
                if (q < S && h == 0 && (!shared || wave == 0)) delta[(long)blockIdx.x * S + q] = dl;
            }
            dl = dls[sl];
            const float L2 = L2s[sl];
            const uint32_t srow = drop_state(drop, (((uint64_t)blockIdx.x * S + (uint64_t)q) * (uint64_t)drop_attn_ld(S)) >> 1) + (uint32_t)(2 * h) * DROP_PHI;
            // key tiles of this chunk; the shared tile takes every 4th GLOBAL tile index
            const int kt_lo = ch * ckt, kt_hi = min(nkt, kt_lo + ckt);
            int kt = kt_lo;
            int kstep = 1;
            if (shared) {
                kstep = 4;
                kt = kt_lo + ((wave - kt_lo) % 4 + 4) % 4;
            }
            for (; kt < kt_hi; kt += kstep) {
                const int lr = (kt - kt_lo) * 32;                 // first LDS row of the tile
                f32x16 s16, dp16;
#pragma unroll
                for (int i = 0; i < 16; ++i) { s16[i] = 0.f; dp16[i] = 0.f; }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (lr + c) * LDK + 16 * s + 8 * h);
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + (lr + c) * LDK + 16 * s + 8 * h);
                    s16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s16, 0, 0, 0);
                    dp16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[s], dp16, 0, 0, 0);
                }
                if (drop.p > 0.f) {
                    const uint32_t skt = srow + (uint32_t)(kt * 16) * DROP_PHI;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint32_t b0 = drop_bits(skt + (uint32_t)(4 * g) * DROP_PHI), b1 = drop_bits(skt + (uint32_t)(4 * g + 1) * DROP_PHI);
                        dp16[4 * g + 0] = drop_keep_even(b0, drop.thr) ? dp16[4 * g + 0] * ks : 0.f;
                        dp16[4 * g + 1] = drop_keep_odd(b0, drop.thr) ? dp16[4 * g + 1] * ks : 0.f;
                        dp16[4 * g + 2] = drop_keep_even(b1, drop.thr) ? dp16[4 * g + 2] * ks : 0.f;
                        dp16[4 * g + 3] = drop_keep_odd(b1, drop.thr) ? dp16[4 * g + 3] * ks : 0.f;
                    }
                }
                if (__builtin_amdgcn_readfirstlane((int)Mt[kt])) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kt * 32 + acc_row(i, h);
                        const float p = Ms[key] ? 0.f : fast_exp2(s16[i] * sc - L2);
                        s16[i] = p * (dp16[i] - dl) * scale;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) s16[i] = fast_exp2(s16[i] * sc - L2) * (dp16[i] - dl) * scale;
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 sf = frag_from_acc(s16, s2);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const bf16x8 kf = frag_transposed(Kt, LDT, min(dt * 32 + c, DH - 1), lr, s2, h);
                        dQ[sl][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, sf, dQ[sl][dt], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- outputs -------------------------------------------------------------------------------------------
#pragma unroll
    for (int sl = 0; sl < DQ_SLOTS; ++sl) {
        const int it = wave + 4 * sl;
        const bool live = it < n_items;
        const bool shared = live && it >= nq_main;
        const int qt = shared ? nqt - 1 : it;
        const int q = qt * 32 + c;
        // the shared tile sits in the same slot of every wave (nq_main % 4 == 0), so this barrier is uniform
        if (coop && 4 * sl + 0 >= nq_main && 4 * sl < nq_main + 4) {
            if (wave != 0 && q < S) {
                float* part = Co + ((wave - 1) * CO_MAXQ + min(c, CO_MAXQ - 1)) * (DH + 2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int d = dt * 32 + acc_row(i, h);
                        if (d < DH) part[d] = dQ[sl][dt][i];
                    }
            }
            __syncthreads();
            if (wave != 0) continue;
            if (q < S) {
                const float* p0 = Co + min(c, CO_MAXQ - 1) * (DH + 2);
                constexpr int WS = CO_MAXQ * (DH + 2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int d = min(dt * 32 + acc_row(i, h), DH - 1);
                        dQ[sl][dt][i] += p0[d] + p0[WS + d] + p0[2 * WS + d];
                    }
            }
        }
        if (live && q < S) {
            const long out = ((long)n * S + q) * ld + hd * DH;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * h;
                    if (d < DH) {
                        f32x4 v = {dQ[sl][dt][4 * g], dQ[sl][dt][4 * g + 1], dQ[sl][dt][4 * g + 2], dQ[sl][dt][4 * g + 3]};
                        store4<IOB>(dqkv, out + d, v);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, dQ, row-major-only variant (see attn_fwd_rm_kernel): K and V row-major, the K^T operand of dQ^T = K^T dS^T through
// the transposing LDS read.  All keys resident (two workgroups per CU at S = 257 / dh = 64), so a query tile's Q / dO / O
// rows are read once; with PF the next tile's rows are requested while the current one is multiplied.
// ------------------------------------------------------------------------------------------------------
// DROP: dropout compiled in or out - a run-time test of drop.p inside the tile loop splits its body into basic blocks the
// scheduler cannot move LDS reads and MFMAs across
template <int DH, bool IOB, int NW, bool PF, bool DROP>
__global__ __launch_bounds__(64 * NW, NW / 2) void attn_bwd_dq_rm_kernel(const void* __restrict__ qkv, const void* __restrict__ ctx,
                                                                     const void* __restrict__ dctx,
                                                                     const float* __restrict__ lse2, float* __restrict__ delta,
                                                                     const uint8_t* __restrict__ mask, int mask_B,
                                                                     void* __restrict__ dqkv, int S, int E, int nh, DropKey drop_in, int qkv_B) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = 64 * NW;
    const int Sp = (S + 31) / 32 * 32, R = rm_rows(S);
    constexpr int LDK = DH + 8;
    __bf16* Ks = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Vs = Ks + R * LDK;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vs + R * LDK);
    uint8_t* Mt = Ms + Sp;
    float* Co = reinterpret_cast<float*>(Mt + 64);
    constexpr int DT = (DH + 31) / 32;
    constexpr int KS = DH / 16;

    const int n = blockIdx.x / nh, hd = blockIdx.x % nh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the tile loops and their branches run on the SALU
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;
    const int nqt = Sp / 32, nkt = Sp / 32;
    const bool coop = (nqt % NW == 1) && nqt > 1 && (S - 32 * (nqt - 1)) <= CO_MAXQ;
    const int nq_main = coop ? nqt - 1 : nqt;
    const int n_items = nq_main + (coop ? NW : 0);

    // one query tile's operands: Q and dO fragments, the O row pieces for delta = sum_d dO * O, the row's log-sum-exp
    bf16x8 qn[KS], dn[KS];
    float on[KS][8], dvn[KS][8];                       // fp32 I/O only: the bf16 route recovers dO from dn and keeps O packed
    bf16x8 opk[KS];
    float l2n = 0.f;
    auto request = [&](int it) {
        const int qt = it >= nq_main ? nqt - 1 : it;
        const int qc = min(qt * 32 + c, S - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const long off = ((long)n * S + qc) * E + hd * DH + 16 * s + 8 * h;
            qn[s] = load_frag8<IOB>(qkv, base + (long)qc * ld + 16 * s + 8 * h);
            dn[s] = load_frag8<IOB>(dctx, off);
            if constexpr (IOB) {
                opk[s] = load_frag8<true>(ctx, off);
            } else {
                load_f32x8<false>(dctx, off, dvn[s]);
                load_f32x8<false>(ctx, off, on[s]);
            }
        }
        l2n = lse2[(long)blockIdx.x * S + qc];
    };
    auto delta_of_request = [&]() -> float {            // this lane-half's part of delta for the requested rows
        float dl = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if constexpr (IOB) {
                const u32x4 dw = __builtin_bit_cast(u32x4, dn[s]), ow = __builtin_bit_cast(u32x4, opk[s]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dl += __builtin_bit_cast(float, dw[j] << 16) * __builtin_bit_cast(float, ow[j] << 16);
                    dl += __builtin_bit_cast(float, dw[j] & 0xffff0000u) * __builtin_bit_cast(float, ow[j] & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += dvn[s][j] * on[s][j];
            }
        }
        return dl;
    };
    if (wave < n_items) request(wave);
    const long moff = (long)(n % mask_B) * S;
    uint8_t mf[MASK_U];
    mask_request<NT>(mf, mask, moff, S, tid);
    stage_kv_rows<DH, IOB, NT, (NW == 8 ? 5 : 9)>(Ks, Vs, qkv, base + E, base + 2 * E, ld, S, R, tid,
                                                   [&]() { mask_flags<NT>(Ms, Mt, mf, mask, moff, S, Sp, tid); });
    __syncthreads();

    const float scale = rsqrtf((float)DH);
    const float sc = scale * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    TileOff toff;
    toff.init(LDK, R - 1, lane);
    for (int it = wave; it < n_items; it += NW) {
        const bool shared = it >= nq_main;
        const int qt = shared ? nqt - 1 : it;
        const int kt0 = shared ? wave : 0, kstep = shared ? NW : 1;
        const int q = qt * 32 + c;
        if (!PF && it != wave) request(it);
        bf16x8 qf[KS], df[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { qf[s] = qn[s]; df[s] = dn[s]; }
        float dl = delta_of_request();
        const float L2 = q < S ? l2n : 0.f;
        if (PF && it + NW < n_items) request(it + NW);
        dl += __shfl_xor(dl, 32, 64);
        if (q < S && h == 0 && (!shared || wave == 0)) delta[(long)blockIdx.x * S + q] = dl;      // consumed by the dK/dV kernel
        f32x16 dQ[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) dQ[dt][i] = 0.f;
        const uint32_t srow = drop_state(drop, (((uint64_t)blockIdx.x * S + (uint64_t)q) * (uint64_t)drop_attn_ld(S)) >> 1) + (uint32_t)(2 * h) * DROP_PHI;
        for (int kt = kt0; kt < nkt; kt += kstep) {
            f32x16 s16, dp16;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s16[i] = 0.f; dp16[i] = 0.f; }
            int tro[4], krow;
            toff.tile(kt * 32, lane, tro, krow);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + krow + 16 * s);
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + krow + 16 * s);
                s16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s16, 0, 0, 0);
                dp16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[s], dp16, 0, 0, 0);
            }
            if constexpr (DROP) {
                const uint32_t skt = srow + (uint32_t)(kt * 16) * DROP_PHI;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t b0 = drop_bits(skt + (uint32_t)(4 * g) * DROP_PHI), b1 = drop_bits(skt + (uint32_t)(4 * g + 1) * DROP_PHI);
                    dp16[4 * g + 0] = drop_keep_even(b0, drop.thr) ? dp16[4 * g + 0] * ks : 0.f;
                    dp16[4 * g + 1] = drop_keep_odd(b0, drop.thr) ? dp16[4 * g + 1] * ks : 0.f;
                    dp16[4 * g + 2] = drop_keep_even(b1, drop.thr) ? dp16[4 * g + 2] * ks : 0.f;
                    dp16[4 * g + 3] = drop_keep_odd(b1, drop.thr) ? dp16[4 * g + 3] * ks : 0.f;
                }
            }
            // dS without the 1 / sqrt(dh) factor: it multiplies the finished dQ tile once instead of every element of every tile
            if (__builtin_amdgcn_readfirstlane((int)Mt[kt])) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 32 + acc_row(i, h);
                    const float p = Ms[key] ? 0.f : fast_exp2(__builtin_fmaf(s16[i], sc, -L2));
                    s16[i] = p * (dp16[i] - dl);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) s16[i] = fast_exp2(__builtin_fmaf(s16[i], sc, -L2)) * (dp16[i] - dl);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 sf = frag_from_acc(s16, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 kf = frag_tr_at(Ks, tro, s2, dt * 32);
                    dQ[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, sf, dQ[dt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) dQ[dt][i] *= scale;
        if (shared) {      // sum the NW key-range partials of the shared tile in wave 0
            if (wave != 0 && q < S) {
                float* part = Co + ((wave - 1) * CO_MAXQ + min(c, CO_MAXQ - 1)) * (DH + 2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int d = dt * 32 + acc_row(i, h);
                        if (d < DH) part[d] = dQ[dt][i];
                    }
            }
            __syncthreads();
            if (wave != 0) continue;
            if (q < S) {
                const float* p0 = Co + min(c, CO_MAXQ - 1) * (DH + 2);
                constexpr int WS = CO_MAXQ * (DH + 2);
#pragma unroll 1
                for (int w = 0; w < NW - 1; ++w)             // one partial at a time (register pressure)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) dQ[dt][i] += p0[w * WS + min(dt * 32 + acc_row(i, h), DH - 1)];
            }
        }
        if (q < S) {
            const long out = ((long)n * S + q) * ld + hd * DH;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * h;
                    if (d < DH) {
                        f32x4 v = {dQ[dt][4 * g], dQ[dt][4 * g + 1], dQ[dt][4 * g + 2], dQ[dt][4 * g + 3]};
                        store4<IOB>(dqkv, out + d, v);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, dQ, long sequences: attn_bwd_dq2_kernel's scheme with a run-time number of key chunks of CK keys and grid.y
// over groups of 4 * DQ_SLOTS query tiles (no shared left-over tile: every tile belongs to one wave).
// ------------------------------------------------------------------------------------------------------
template <int DH, bool IOB>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_long_kernel(const void* __restrict__ qkv, const void* __restrict__ ctx,
                                                              const void* __restrict__ dctx,
                                                              const float* __restrict__ lse2, float* __restrict__ delta,
                                                              const uint8_t* __restrict__ mask, int mask_B,
                                                              void* __restrict__ dqkv, int S, int E, int nh, DropKey drop_in, int qkv_B, int CK, int npairs, int nqg) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Sp = (S + 31) / 32 * 32;
    const int nqt = Sp / 32, nkt = Sp / 32;
    const int ckt = CK / 32;                        // key tiles per chunk (CK keys, a multiple of 32, resident at a time)
    const int nch = (nkt + ckt - 1) / ckt;
    constexpr int LDK = DH + 8;
    const int LDT = CK + TPAD;
    __bf16* Ks = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Vs = Ks + CK * LDK;
    __bf16* Kt = Vs + CK * LDK;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Kt + DH * LDT);
    uint8_t* Mt = Ms + Sp;
    constexpr int DT = (DH + 31) / 32;
    constexpr int KS = DH / 16;

    // XCD-aware order (see attn_fwd_long_kernel): the query groups of one (sample, head) pair share an XCD's L2
    const int pair = ((int)(blockIdx.x >> 3) / nqg) * 8 + (int)(blockIdx.x & 7);
    const int qg = (int)(blockIdx.x >> 3) % nqg;
    if (pair >= npairs) return;
    const int n = pair / nh, hd = pair % nh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;
    for (int i = tid; i < Sp; i += 256) Ms[i] = (i >= S) ? 1 : (mask ? mask[(long)(n % mask_B) * S + i] : 0);
    __syncthreads();
    for (int t = tid; t < nkt; t += 256) {
        uint8_t any = 0;
        for (int j = 0; j < 32; ++j) any |= Ms[t * 32 + j];
        Mt[t] = any;
    }

    const float scale = rsqrtf((float)DH);
    const float sc = scale * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    const int it0 = 4 * DQ_SLOTS * qg;           // first query tile of this workgroup (grid.y splits the query tiles)

    f32x16 dQ[DQ_SLOTS][DT];
    float L2s[DQ_SLOTS], dls[DQ_SLOTS];
#pragma unroll
    for (int sl = 0; sl < DQ_SLOTS; ++sl) {
        L2s[sl] = 0.f; dls[sl] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) dQ[sl][dt][i] = 0.f;
    }

    for (int ch = 0; ch < nch; ++ch) {
        const int kbase = ch * CK;                               // first key of the chunk
        const int krows = min(CK, Sp - kbase);                  // staged rows of the chunk (multiple of 32)
        __syncthreads();                                         // the previous chunk's readers are done (and Mt is written)
        // rows beyond S are zero-filled by the staging helpers (their S / Sp arguments are chunk-relative)
        stage_rows<DH, IOB>(Ks, qkv, base + E + (long)kbase * ld, ld, max(0, min(S - kbase, krows)), krows, tid, 256);
        stage_rows<DH, IOB>(Vs, qkv, base + 2 * E + (long)kbase * ld, ld, max(0, min(S - kbase, krows)), krows, tid, 256);
        stage_transposed<DH, IOB>(Kt, qkv, base + E + (long)kbase * ld, ld, max(0, min(S - kbase, krows)), CK, tid, 256);
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < DQ_SLOTS; ++sl) {
            const int qt = it0 + wave + 4 * sl;
            if (qt >= nqt) continue;
            const int q = qt * 32 + c;
            const int qc = min(q, S - 1);
            bf16x8 qf[KS], df[KS];
            float dl = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const long off = ((long)n * S + qc) * E + hd * DH + 16 * s + 8 * h;
                qf[s] = load_frag8<IOB>(qkv, base + (long)qc * ld + 16 * s + 8 * h);
                df[s] = load_frag8<IOB>(dctx, off);
                if (ch == 0) {
                    float dv[8], ov[8];
                    load_f32x8<IOB>(dctx, off, dv);
                    load_f32x8<IOB>(ctx, off, ov);
#pragma unroll
                    for (int j = 0; j < 8; ++j) dl += dv[j] * ov[j];
                }
            }
            if (ch == 0) {
                dl += __shfl_xor(dl, 32, 64);
                dls[sl] = dl;
                L2s[sl] = q < S ? lse2[(long)pair * S + q] : 0.f;
                if (q < S && h == 0) delta[(long)pair * S + q] = dl;
            }
            dl = dls[sl];
            const float L2 = L2s[sl];
            const uint32_t srow = drop_state(drop, (((uint64_t)pair * S + (uint64_t)q) * (uint64_t)drop_attn_ld(S)) >> 1) + (uint32_t)(2 * h) * DROP_PHI;
            const int kt_lo = ch * ckt, kt_hi = min(nkt, kt_lo + ckt);      // key tiles of this chunk
            for (int kt = kt_lo; kt < kt_hi; ++kt) {
                const int lr = (kt - kt_lo) * 32;                 // first LDS row of the tile
                f32x16 s16, dp16;
#pragma unroll
                for (int i = 0; i < 16; ++i) { s16[i] = 0.f; dp16[i] = 0.f; }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (lr + c) * LDK + 16 * s + 8 * h);
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + (lr + c) * LDK + 16 * s + 8 * h);
                    s16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s16, 0, 0, 0);
                    dp16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[s], dp16, 0, 0, 0);
                }
                if (drop.p > 0.f) {
                    const uint32_t skt = srow + (uint32_t)(kt * 16) * DROP_PHI;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint32_t b0 = drop_bits(skt + (uint32_t)(4 * g) * DROP_PHI), b1 = drop_bits(skt + (uint32_t)(4 * g + 1) * DROP_PHI);
                        dp16[4 * g + 0] = drop_keep_even(b0, drop.thr) ? dp16[4 * g + 0] * ks : 0.f;
                        dp16[4 * g + 1] = drop_keep_odd(b0, drop.thr) ? dp16[4 * g + 1] * ks : 0.f;
                        dp16[4 * g + 2] = drop_keep_even(b1, drop.thr) ? dp16[4 * g + 2] * ks : 0.f;
                        dp16[4 * g + 3] = drop_keep_odd(b1, drop.thr) ? dp16[4 * g + 3] * ks : 0.f;
                    }
                }
                if (__builtin_amdgcn_readfirstlane((int)Mt[kt])) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kt * 32 + acc_row(i, h);
                        const float p = Ms[key] ? 0.f : fast_exp2(s16[i] * sc - L2);
                        s16[i] = p * (dp16[i] - dl) * scale;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) s16[i] = fast_exp2(s16[i] * sc - L2) * (dp16[i] - dl) * scale;
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 sf = frag_from_acc(s16, s2);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const bf16x8 kf = frag_transposed(Kt, LDT, min(dt * 32 + c, DH - 1), lr, s2, h);
                        dQ[sl][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, sf, dQ[sl][dt], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- outputs -------------------------------------------------------------------------------------------
#pragma unroll
    for (int sl = 0; sl < DQ_SLOTS; ++sl) {
        const int qt = it0 + wave + 4 * sl;
        const bool live = qt < nqt;
        const int q = qt * 32 + c;
        if (live && q < S) {
            const long out = ((long)n * S + q) * ld + hd * DH;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * h;
                    if (d < DH) {
                        f32x4 v = {dQ[sl][dt][4 * g], dQ[sl][dt][4 * g + 1], dQ[sl][dt][4 * g + 2], dQ[sl][dt][4 * g + 3]};
                        store4<IOB>(dqkv, out + d, v);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, dK / dV: keys on the lanes.  One WAVE per (sample, head, 32-key tile) - no workgroup barriers.
// The wave keeps K, V fragments of its key tile and the dK^T, dV^T [dh x 32 keys] accumulators in
// registers and walks all query tiles.  Per query tile it loads its 32 rows of Q and dO once: the packed
// bf16 registers ARE the row-major A fragments of S = Q K^T and dP = dO V^T; the same 16-byte pieces go to a
// wave-private LDS slab from which the TRANSPOSED A fragments of dV^T += dO^T P and dK^T += Q^T dS come
// back through ds_read_b64_tr_b16 (hardware transpose read: lane 4q+p of a 16-lane group supplies row q,
// columns 4p..4p+3 of a 4x16 block and receives column (lane & 15), rows 0..3).
// ------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 frag_tr(const __bf16* img, int ld, int col0, int s2, int lane) {
    const int i = lane & 15, grp = lane >> 4;
    const int hh = grp >> 1, colhalf = grp & 1;
    const __bf16* p0 = img + (16 * s2 + 4 * hh + (i >> 2)) * ld + col0 + 16 * colhalf + 4 * (i & 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * ld));
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

template <int DH, bool IOB>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const void* __restrict__ qkv, const void* __restrict__ dctx,
                                                           const float* __restrict__ lse2, const float* __restrict__ delta,
                                                           const uint8_t* __restrict__ mask, int mask_B,
                                                           void* __restrict__ dqkv, int S, int E, int nh, DropKey drop_in,
                                                           long total_items, int qkv_B) {
    const DropKey drop = drop_live(drop_in);
    constexpr int LDR = DH + 8;      // row-major tile [32][DH+8] bf16
    constexpr int DT = (DH + 31) / 32;
    constexpr int KS = DH / 16;
    __shared__ __attribute__((aligned(16))) __bf16 Qr[4][32 * LDR];
    __shared__ __attribute__((aligned(16))) __bf16 Dr[4][32 * LDR];
    __shared__ __attribute__((aligned(16))) float Ls[4][32], Dl[4][32];

    const int Sp = (S + 31) / 32 * 32;
    const int nqt = Sp / 32, nkt = Sp / 32;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= total_items) return;                   // whole wave leaves: nothing below synchronises across waves
    const int kt = (int)(item % nkt);
    const long nhid = item / nkt;                      // n * nh + head
    const int n = (int)(nhid / nh), hd = (int)(nhid % nh);
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;
    const long dbase = (long)n * S * E + hd * DH;
    const float scale = rsqrtf((float)DH);
    const float sc = scale * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;

    const int key = kt * 32 + c;
    const bool kvalid = key < S && !(mask && mask[(long)(n % mask_B) * S + key]);
    bf16x8 kf[KS], vf[KS];
    const int keyc = min(key, S - 1);                  // keys past the end re-read the last key: masked below, never stored
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        kf[s] = load_frag8<IOB>(qkv, base + (long)keyc * ld + E + 16 * s + 8 * h);
        vf[s] = load_frag8<IOB>(qkv, base + (long)keyc * ld + 2 * E + 16 * s + 8 * h);
    }
    f32x16 dK[DT], dV[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dK[dt][i] = 0.f; dV[dt][i] = 0.f; }

    __bf16* qimg = Qr[wave];
    __bf16* dimg = Dr[wave];
    // dropout stream constants of this lane (drop_rng.h): pair state of (row 0 of tile 0, this key), row step, parity
    const int par = c & 1;
    const uint32_t Sd = (uint32_t)drop_attn_ld(S);
    const uint32_t rowmul = (Sd / 2) * DROP_PHI;
    const uint32_t stile = drop_state(drop, (((uint64_t)nhid * S) * Sd) / 2 + (uint64_t)(key >> 1)) + (uint32_t)(par + 4 * h) * rowmul;
    const int shl = par ? 0 : 16;                      // odd element -> high half, even element -> low half
    const uint32_t thr_hi = drop.thr << 16;
    // The query tile of the NEXT iteration is requested while this one is multiplied (register double buffer): every one
    // of the nqt iterations used to begin with a global round trip that the two resident waves per SIMD do not cover.
    bf16x8 qn[KS], dn[KS];
    float lsn = 0.f, dln = 0.f;
    auto request = [&](int qt) {
        const int qc = min(qt * 32 + c, S - 1);        // rows past the end: finite duplicates, zeroed by (qq < S) below
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qn[s] = load_frag8<IOB>(qkv, base + (long)qc * ld + 16 * s + 8 * h);
            dn[s] = load_frag8<IOB>(dctx, dbase + (long)qc * E + 16 * s + 8 * h);
        }
        lsn = lse2[nhid * S + qc];
        dln = delta[nhid * S + qc];
    };
    request(0);
    for (int qt = 0; qt < nqt; ++qt) {
        const int q = qt * 32 + c;
        bf16x8 qa[KS], da[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qa[s] = qn[s];
            da[s] = dn[s];
            *reinterpret_cast<bf16x8*>(qimg + c * LDR + 16 * s + 8 * h) = qa[s];
            *reinterpret_cast<bf16x8*>(dimg + c * LDR + 16 * s + 8 * h) = da[s];
        }
        if (h == 0) {
            Ls[wave][c] = q < S ? lsn : 0.f;
            Dl[wave][c] = q < S ? dln : 0.f;
        }
        if (qt + 1 < nqt) request(qt + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        f32x16 s16, dp16;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s16[i] = 0.f; dp16[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            s16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[s], kf[s], s16, 0, 0, 0);
            dp16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[s], vf[s], dp16, 0, 0, 0);
        }
        // rows 8g + 4h .. +3 of the tile (registers 4g .. 4g+3): log-sum-exp and delta as two 16-byte LDS reads per group
        f32x4 lsv[4], dlv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            lsv[g] = *reinterpret_cast<const f32x4*>(&Ls[wave][8 * g + 4 * h]);
            dlv[g] = *reinterpret_cast<const f32x4*>(&Dl[wave][8 * g + 4 * h]);
        }
        // Dropout: element (row qq, key) pairs with key ^ 1, i.e. with the NEIGHBOUR LANE.  Each lane hashes the rows of
        // its own parity (8 of its 16 registers) and fetches the other 8 hashes from its partner with one DPP move:
        // half a hash per element.  state(row) is linear in the row (drop_rng.h).
        uint32_t bits[16];
        if (drop.p > 0.f) {
            const uint32_t st = stile + (uint32_t)(qt * 32) * rowmul;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t mine = drop_bits(st + (uint32_t)(2 * (k & 1) + 8 * (k >> 1)) * rowmul);      // row of register 2k + par
                const uint32_t theirs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                bits[2 * k] = par ? theirs : mine;
                bits[2 * k + 1] = par ? mine : theirs;
            }
        }
        const bool rows_full = qt * 32 + 32 <= S;
        f32x16 pd16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = fast_exp2(s16[i] * sc - lsv[i >> 2][i & 3]);
            float pk = p, dpk = dp16[i];
            if (drop.p > 0.f) {
                const bool keep = (bits[i] << shl) >= thr_hi;
                pk = keep ? p * ks : 0.f;
                dpk = keep ? dpk * ks : 0.f;
            }
            float ds = p * (dpk - dlv[i >> 2][i & 3]) * scale;
            const bool ok = kvalid && (rows_full || qt * 32 + acc_row(i, h) < S);
            pd16[i] = ok ? pk : 0.f;
            s16[i] = ok ? ds : 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = frag_from_acc(pd16, s2);
            const bf16x8 sf = frag_from_acc(s16, s2);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 dof = frag_tr(dimg, LDR, dt * 32, s2, lane);
                const bf16x8 qtf = frag_tr(qimg, LDR, dt * 32, s2, lane);
                dV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, pf, dV[dt], 0, 0, 0);
                dK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, sf, dK[dt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (key < S) {
        const long outk = ((long)n * S + key) * ld + E + hd * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * h;
                if (d < DH) {
                    f32x4 a = {dK[dt][4 * g], dK[dt][4 * g + 1], dK[dt][4 * g + 2], dK[dt][4 * g + 3]};
                    f32x4 b = {dV[dt][4 * g], dV[dt][4 * g + 1], dV[dt][4 * g + 2], dV[dt][4 * g + 3]};
                    store4<IOB>(dqkv, outk + d, a);
                    store4<IOB>(dqkv, outk + E + d, b);
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, dK / dV, workgroup-resident variant (see attn_fwd_rm_kernel).  The kernel above gives every WAVE its own key tile
// and has it fetch all nqt query tiles (Q, dO, lse, delta) from global memory into a private LDS slab: nkt x the traffic, one
// global round trip and two LDS writes per tile step.  Here one workgroup owns a (sample, head): Q and dO row-major, lse and
// delta are staged ONCE (one round trip, ceil8(S) rows), the waves take key tiles round-robin with K / V fragments in
// registers (the next tile's requested while the current one is multiplied), and the row-major A fragments of S = Q K^T,
// dP = dO V^T and the transposed ones of dV^T += dO^T P, dK^T += Q^T dS all come out of the same two images: no LDS writes,
// no barriers inside the loop.  A left-over key tile of at most CO_MAXQ keys (S = 257) is shared: every wave takes every
// NW-th query tile for it and the partial dK / dV rows are summed through LDS.
// Arithmetic and dropout stream as above.
// ------------------------------------------------------------------------------------------------------
template <int DH, bool IOB, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void attn_bwd_dkv_rm_kernel(const void* __restrict__ qkv, const void* __restrict__ dctx,
                                                                          const float* __restrict__ lse2, const float* __restrict__ delta,
                                                                          const uint8_t* __restrict__ mask, int mask_B,
                                                                          void* __restrict__ dqkv, int S, int E, int nh, DropKey drop_in, int qkv_B) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = 64 * NW;
    const int Sp = (S + 31) / 32 * 32, R = rm_rows(S);
    constexpr int LDK = DH + 8;
    __bf16* Qs = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Ds = Qs + R * LDK;
    float* Ls = reinterpret_cast<float*>(Ds + R * LDK);
    float* Dl = Ls + Sp;
    float* Co = Dl + Sp;                                   // [waves 1..NW-1][CO_MAXQ][2 * DH]
    constexpr int DT = (DH + 31) / 32;
    constexpr int KS = DH / 16;

    const long nhid = blockIdx.x;
    const int n = blockIdx.x / nh, hd = blockIdx.x % nh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the tile loops and their branches run on the SALU
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;
    const long dbase = (long)n * S * E + hd * DH;
    const long moff = (long)(n % mask_B) * S;
    const int nqt = Sp / 32, nkt = Sp / 32;
    const bool coop = (nkt % NW == 1) && nkt > 1 && (S - 32 * (nkt - 1)) <= CO_MAXQ;
    const int nk_main = coop ? nkt - 1 : nkt;
    const int n_items = nk_main + (coop ? NW : 0);

    bf16x8 kn[KS], vn[KS];
    uint8_t mkn = 0;
    auto request_kv = [&](int it) {
        const int kt = it >= nk_main ? nkt - 1 : it;
        const int keyc = min(kt * 32 + c, S - 1);          // keys past the end re-read the last key: masked below, never stored
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kn[s] = load_frag8<IOB>(qkv, base + (long)keyc * ld + E + 16 * s + 8 * h);
            vn[s] = load_frag8<IOB>(qkv, base + (long)keyc * ld + 2 * E + 16 * s + 8 * h);
        }
        mkn = mask ? mask[moff + keyc] : (uint8_t)0;
    };
    if (wave < n_items) request_kv(wave);
    stage_two_rows<DH, IOB, NT, (NW == 8 ? 5 : 9)>(Qs, Ds, qkv, base, ld, dctx, dbase, (long)E, S, R, tid, [&]() {
        for (int u = 0; u * NT < Sp; ++u) {
            const int i = u * NT + tid;
            const int ic = min(i, S - 1);
            const float a = lse2[nhid * S + ic], b = delta[nhid * S + ic];
            if (i < Sp) {
                Ls[i] = i < S ? a : 0.f;
                Dl[i] = i < S ? b : 0.f;
            }
        }
    });
    __syncthreads();

    const float scale = rsqrtf((float)DH);
    const float sc = scale * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    const int par = c & 1;
    const uint32_t Sd = (uint32_t)drop_attn_ld(S);
    const uint32_t rowmul = (Sd / 2) * DROP_PHI;
    const int shl = par ? 0 : 16;                      // odd element -> high half, even element -> low half
    const uint32_t thr_hi = drop.thr << 16;
    TileOff toff;
    toff.init(LDK, R - 1, lane);
    for (int it = wave; it < n_items; it += NW) {
        const bool shared = it >= nk_main;
        const int kt = shared ? nkt - 1 : it;
        const int qt0 = shared ? wave : 0, qstep = shared ? NW : 1;
        const int key = kt * 32 + c;
        const bool kvalid = key < S && !mkn;
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { kf[s] = kn[s]; vf[s] = vn[s]; }
        if (it + NW < n_items) request_kv(it + NW);
        f32x16 dK[DT], dV[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) { dK[dt][i] = 0.f; dV[dt][i] = 0.f; }
        // dropout stream constants of this lane (drop_rng.h): pair state of (row 0 of tile 0, this key), row step, parity
        const uint32_t stile = drop_state(drop, (((uint64_t)nhid * S) * Sd) / 2 + (uint64_t)(key >> 1)) + (uint32_t)(par + 4 * h) * rowmul;
        // No validity selects inside the loop:
        //  * query rows past the end are ZERO rows of Q and dO (rm_rows) with lse = delta = 0: p = 1, dP = 0, dS = p (0 - 0) = 0, and
        //    dV^T += dO^T P sees a zero row of dO - both contributions vanish by themselves;
        //  * a masked / padded key is one LANE's column of dK^T, dV^T: it is zeroed once, after the loop;
        //  * 1 / sqrt(dh) multiplies the finished dK tile once.
        for (int qt = qt0; qt < nqt; qt += qstep) {
            f32x16 s16, dp16;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s16[i] = 0.f; dp16[i] = 0.f; }
            int tro[4], qrow;
            toff.tile(qt * 32, lane, tro, qrow);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qs + qrow + 16 * s);
                const bf16x8 da = *reinterpret_cast<const bf16x8*>(Ds + qrow + 16 * s);
                s16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[s], s16, 0, 0, 0);
                dp16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[s], dp16, 0, 0, 0);
            }
            // rows 8g + 4h .. +3 of the tile (registers 4g .. 4g+3): log-sum-exp and delta as two 16-byte LDS reads per group
            f32x4 lsv[4], dlv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                lsv[g] = *reinterpret_cast<const f32x4*>(&Ls[qt * 32 + 8 * g + 4 * h]);
                dlv[g] = *reinterpret_cast<const f32x4*>(&Dl[qt * 32 + 8 * g + 4 * h]);
            }
            f32x16 pd16;
            if (drop.p > 0.f) {
                // Each lane hashes the rows of its own parity (register 2k + par); the even lane's word then serves register 2k of
                // BOTH lanes of the pair and the odd lane's word register 2k + 1: two DPP broadcasts, no per-lane selects.
                const uint32_t st = stile + (uint32_t)(qt * 32) * rowmul;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t mine = drop_bits(st + (uint32_t)(2 * (k & 1) + 8 * (k >> 1)) * rowmul);
                    const uint32_t be = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xA0, 0xf, 0xf, true);   // quad_perm [0,0,2,2]
                    const uint32_t bo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xF5, 0xf, 0xf, true);   // quad_perm [1,1,3,3]
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int i = 2 * k + e;
                        const float km = ((e ? bo : be) << shl) >= thr_hi ? ks : 0.f;
                        const float p = fast_exp2(__builtin_fmaf(s16[i], sc, -lsv[i >> 2][i & 3]));
                        pd16[i] = p * km;
                        s16[i] = p * __builtin_fmaf(dp16[i], km, -dlv[i >> 2][i & 3]);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = fast_exp2(__builtin_fmaf(s16[i], sc, -lsv[i >> 2][i & 3]));
                    pd16[i] = p;
                    s16[i] = p * (dp16[i] - dlv[i >> 2][i & 3]);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = frag_from_acc(pd16, s2);
                const bf16x8 sf = frag_from_acc(s16, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 dof = frag_tr_at(Ds, tro, s2, dt * 32);
                    const bf16x8 qtf = frag_tr_at(Qs, tro, s2, dt * 32);
                    dV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, pf, dV[dt], 0, 0, 0);
                    dK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, sf, dK[dt], 0, 0, 0);
                }
            }
        }
        {
            const float kz = kvalid ? scale : 0.f, vz = kvalid ? 1.f : 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) { dK[dt][i] *= kz; dV[dt][i] *= vz; }
        }
        if (shared) {      // sum the NW query-range partials of the shared key tile in wave 0
            constexpr int WS = CO_MAXQ * 2 * DH;
            if (wave != 0 && key < S) {
                float* part = Co + ((wave - 1) * CO_MAXQ + min(c, CO_MAXQ - 1)) * (2 * DH);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int d = dt * 32 + acc_row(i, h);
                        if (d < DH) { part[d] = dK[dt][i]; part[DH + d] = dV[dt][i]; }
                    }
            }
            __syncthreads();
            if (wave != 0) continue;
            if (key < S) {
                const float* p0 = Co + min(c, CO_MAXQ - 1) * (2 * DH);
#pragma unroll 1
                for (int w = 0; w < NW - 1; ++w)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int d = min(dt * 32 + acc_row(i, h), DH - 1);
                            dK[dt][i] += p0[w * WS + d];
                            dV[dt][i] += p0[w * WS + DH + d];
                        }
            }
        }
        if (key < S) {
            const long outk = ((long)n * S + key) * ld + E + hd * DH;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * h;
                    if (d < DH) {
                        f32x4 a = {dK[dt][4 * g], dK[dt][4 * g + 1], dK[dt][4 * g + 2], dK[dt][4 * g + 3]};
                        f32x4 b = {dV[dt][4 * g], dV[dt][4 * g + 1], dV[dt][4 * g + 2], dV[dt][4 * g + 3]};
                        store4<IOB>(dqkv, outk + d, a);
                        store4<IOB>(dqkv, outk + E + d, b);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, long sequences, streaming row-major variants (see attn_fwd_stream_kernel).  One resident tile per wave (the dQ /
// dK + dV accumulators and both operand fragment sets leave no room for two), four per workgroup.
// dQ: the wave's query tile with Q, dO fragments, lse and delta in registers; K and V chunks stream.
// ------------------------------------------------------------------------------------------------------
template <int DH, bool IOB, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_stream_kernel(const void* __restrict__ qkv, const void* __restrict__ ctx,
                                                                    const void* __restrict__ dctx,
                                                                    const float* __restrict__ lse2, float* __restrict__ delta,
                                                                    const uint8_t* __restrict__ mask, int mask_B,
                                                                    void* __restrict__ dqkv, int S, int E, int nh, DropKey drop_in, int qkv_B,
                                                                    int npairs, int nqg) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Sp = (S + 31) / 32 * 32, R = rm_rows(S);
    constexpr int LDK = DH + 8, IMG = SCK * LDK;
    __bf16* Kb = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* Vb = Kb + 2 * IMG;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vb + 2 * IMG);
    uint8_t* Mt = Ms + Sp;
    constexpr int DT = (DH + 31) / 32;
    constexpr int KS = DH / 16;

    const int pair = ((int)(blockIdx.x >> 3) / nqg) * 8 + (int)(blockIdx.x & 7);
    const int qg = (int)(blockIdx.x >> 3) % nqg;
    if (pair >= npairs) return;
    const int n = pair / nh, hd = pair % nh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;
    const long moff = (long)(n % mask_B) * S;
    const int nkt = Sp / 32;
    const int nchunks = (R + SCK - 1) / SCK;

    const int q = (4 * qg + wave) * 32 + c;
    const int qc = min(q, S - 1);
    bf16x8 qf[KS], df[KS];
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const long off = ((long)n * S + qc) * E + hd * DH + 16 * s + 8 * h;
        qf[s] = load_frag8<IOB>(qkv, base + (long)qc * ld + 16 * s + 8 * h);
        df[s] = load_frag8<IOB>(dctx, off);
        float dv[8], ov[8];
        load_f32x8<IOB>(dctx, off, dv);
        load_f32x8<IOB>(ctx, off, ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += dv[j] * ov[j];
    }
    const float L2raw = lse2[(long)pair * S + qc];
    uint8_t mf[MASK_U];
    mask_request<256>(mf, mask, moff, S, tid);
    ChunkRegs<DH, IOB> cr;
    cr.load(qkv, base + E, ld, qkv, base + 2 * E, ld, 0, S, tid);
    mask_flags<256>(Ms, Mt, mf, mask, moff, S, Sp, tid);
    cr.store(Kb, Vb, 0, S, tid);
    if (nchunks > 1) cr.load(qkv, base + E, ld, qkv, base + 2 * E, ld, SCK, S, tid);
    dl += __shfl_xor(dl, 32, 64);
    if (q < S && h == 0) delta[(long)pair * S + q] = dl;      // consumed by the dK/dV kernel
    const float L2 = q < S ? L2raw : 0.f;
    __syncthreads();

    const float scale = rsqrtf((float)DH);
    const float sc = scale * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    f32x16 dQ[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dQ[dt][i] = 0.f;
    const uint32_t srow = drop_state(drop, (((uint64_t)pair * S + (uint64_t)q) * (uint64_t)drop_attn_ld(S)) >> 1) + (uint32_t)(2 * h) * DROP_PHI;
    TileOff toff;
    toff.init(LDK, SCK - 1, lane);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) cr.store(Kb + (buf ^ 1) * IMG, Vb + (buf ^ 1) * IMG, (ch + 1) * SCK, S, tid);
        if (ch + 2 < nchunks) cr.load(qkv, base + E, ld, qkv, base + 2 * E, ld, (ch + 2) * SCK, S, tid);
        const __bf16* Ks = Kb + buf * IMG;
        const __bf16* Vs = Vb + buf * IMG;
        toff.rlim = min(SCK, R - ch * SCK) - 1;
        const int kt_end = min(nkt, (ch + 1) * (SCK / 32));
        for (int kt = ch * (SCK / 32); kt < kt_end; ++kt) {
            int tro[4], krow;
            toff.tile((kt - ch * (SCK / 32)) * 32, lane, tro, krow);
            f32x16 s16, dp16;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s16[i] = 0.f; dp16[i] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + krow + 16 * s);
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + krow + 16 * s);
                s16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s16, 0, 0, 0);
                dp16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, df[s], dp16, 0, 0, 0);
            }
            if constexpr (DROP) {
                const uint32_t skt = srow + (uint32_t)(kt * 16) * DROP_PHI;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t b0 = drop_bits(skt + (uint32_t)(4 * g) * DROP_PHI), b1 = drop_bits(skt + (uint32_t)(4 * g + 1) * DROP_PHI);
                    dp16[4 * g + 0] = drop_keep_even(b0, drop.thr) ? dp16[4 * g + 0] * ks : 0.f;
                    dp16[4 * g + 1] = drop_keep_odd(b0, drop.thr) ? dp16[4 * g + 1] * ks : 0.f;
                    dp16[4 * g + 2] = drop_keep_even(b1, drop.thr) ? dp16[4 * g + 2] * ks : 0.f;
                    dp16[4 * g + 3] = drop_keep_odd(b1, drop.thr) ? dp16[4 * g + 3] * ks : 0.f;
                }
            }
            if (__builtin_amdgcn_readfirstlane((int)Mt[kt])) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 32 + acc_row(i, h);
                    const float p = Ms[key] ? 0.f : fast_exp2(__builtin_fmaf(s16[i], sc, -L2));
                    s16[i] = p * (dp16[i] - dl);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) s16[i] = fast_exp2(__builtin_fmaf(s16[i], sc, -L2)) * (dp16[i] - dl);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 sf = frag_from_acc(s16, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 kf = frag_tr_at(Ks, tro, s2, dt * 32);
                    dQ[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, sf, dQ[dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (q < S) {
        const long out = ((long)n * S + q) * ld + hd * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * h;
                if (d < DH) {
                    f32x4 v = {dQ[dt][4 * g] * scale, dQ[dt][4 * g + 1] * scale, dQ[dt][4 * g + 2] * scale, dQ[dt][4 * g + 3] * scale};
                    store4<IOB>(dqkv, out + d, v);
                }
            }
    }
}

// dK / dV: the wave's key tile with K, V fragments in registers; Q and dO chunks (with their lse / delta rows) stream.
constexpr int DCK = 64;                                     // dK/dV: 64-row chunks (16 registers in flight: the kernel has no more to spare)
size_t dkv_stream_smem(int DH) { return (size_t)4 * DCK * (DH + 8) * 2 + (size_t)4 * DCK * 4; }
template <int DH, bool IOB, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_stream_kernel(const void* __restrict__ qkv, const void* __restrict__ dctx,
                                                                     const float* __restrict__ lse2, const float* __restrict__ delta,
                                                                     const uint8_t* __restrict__ mask, int mask_B,
                                                                     void* __restrict__ dqkv, int S, int E, int nh, DropKey drop_in, int qkv_B,
                                                                     int npairs, int nkg) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Sp = (S + 31) / 32 * 32, R = rm_rows(S);
    constexpr int LDK = DH + 8, IMG = DCK * LDK;
    __bf16* Qb = reinterpret_cast<__bf16*>(smem_raw);           // [2][IMG]
    __bf16* Db = Qb + 2 * IMG;                                  // [2][IMG]
    float* Lb = reinterpret_cast<float*>(Db + 2 * IMG);         // [2][DCK] log-sum-exp rows of the chunk
    float* Eb = Lb + 2 * DCK;                                   // [2][DCK] delta rows
    constexpr int DT = (DH + 31) / 32;
    constexpr int KS = DH / 16;

    const int pair = ((int)(blockIdx.x >> 3) / nkg) * 8 + (int)(blockIdx.x & 7);
    const int kg = (int)(blockIdx.x >> 3) % nkg;
    if (pair >= npairs) return;
    const long nhid = pair;
    const int n = pair / nh, hd = pair % nh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const long base = (long)(n % qkv_B) * S * ld + hd * DH;
    const long dbase = (long)n * S * E + hd * DH;
    const int nqt = Sp / 32;
    const int nchunks = (R + DCK - 1) / DCK;

    const int key = (4 * kg + wave) * 32 + c;
    const int keyc = min(key, S - 1);                  // keys past the end re-read the last key: zeroed below, never stored
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        kf[s] = load_frag8<IOB>(qkv, base + (long)keyc * ld + E + 16 * s + 8 * h);
        vf[s] = load_frag8<IOB>(qkv, base + (long)keyc * ld + 2 * E + 16 * s + 8 * h);
    }
    const uint8_t mk = mask ? mask[(long)(n % mask_B) * S + keyc] : (uint8_t)0;
    ChunkRegs<DH, IOB, DCK> cr;
    float lr = 0.f, er = 0.f;                          // this thread's lse / delta row of the chunk in flight (threads 0 .. DCK-1)
    auto load_chunk = [&](int row0) {
        cr.load(qkv, base, ld, dctx, dbase, (long)E, row0, S, tid);
        const long r = nhid * S + min(row0 + (tid & (DCK - 1)), S - 1);
        lr = lse2[r];
        er = delta[r];
    };
    auto store_chunk = [&](int b, int row0) {
        cr.store(Qb + b * IMG, Db + b * IMG, row0, S, tid);
        if (tid < DCK) {
            Lb[b * DCK + tid] = row0 + tid < S ? lr : 0.f;
            Eb[b * DCK + tid] = row0 + tid < S ? er : 0.f;
        }
    };
    load_chunk(0);
    store_chunk(0, 0);
    if (nchunks > 1) load_chunk(DCK);
    __syncthreads();

    const float scale = rsqrtf((float)DH);
    const float sc = scale * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    const int par = c & 1;
    const uint32_t Sd = (uint32_t)drop_attn_ld(S);
    const uint32_t rowmul = (Sd / 2) * DROP_PHI;
    const int shl = par ? 0 : 16;
    const uint32_t thr_hi = drop.thr << 16;
    const uint32_t stile = drop_state(drop, (((uint64_t)nhid * S) * Sd) / 2 + (uint64_t)(key >> 1)) + (uint32_t)(par + 4 * h) * rowmul;
    f32x16 dK[DT], dV[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dK[dt][i] = 0.f; dV[dt][i] = 0.f; }
    TileOff toff;
    toff.init(LDK, DCK - 1, lane);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) store_chunk(buf ^ 1, (ch + 1) * DCK);
        if (ch + 2 < nchunks) load_chunk((ch + 2) * DCK);
        const __bf16* Qs = Qb + buf * IMG;
        const __bf16* Ds = Db + buf * IMG;
        const float* Ls = Lb + buf * DCK;
        const float* Dl = Eb + buf * DCK;
        toff.rlim = min(DCK, R - ch * DCK) - 1;
        const int qt_end = min(nqt, (ch + 1) * (DCK / 32));
        for (int qt = ch * (DCK / 32); qt < qt_end; ++qt) {       // see attn_bwd_dkv_rm_kernel: no validity selects in here
            const int ql = (qt - ch * (DCK / 32)) * 32;
            int tro[4], qrow;
            toff.tile(ql, lane, tro, qrow);
            f32x16 s16, dp16;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s16[i] = 0.f; dp16[i] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qs + qrow + 16 * s);
                const bf16x8 da = *reinterpret_cast<const bf16x8*>(Ds + qrow + 16 * s);
                s16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[s], s16, 0, 0, 0);
                dp16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[s], dp16, 0, 0, 0);
            }
            f32x4 lsv[4], dlv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                lsv[g] = *reinterpret_cast<const f32x4*>(&Ls[ql + 8 * g + 4 * h]);
                dlv[g] = *reinterpret_cast<const f32x4*>(&Dl[ql + 8 * g + 4 * h]);
            }
            f32x16 pd16;
            if constexpr (DROP) {
                const uint32_t st = stile + (uint32_t)(qt * 32) * rowmul;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t mine = drop_bits(st + (uint32_t)(2 * (k & 1) + 8 * (k >> 1)) * rowmul);
                    const uint32_t be = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xA0, 0xf, 0xf, true);   // quad_perm [0,0,2,2]
                    const uint32_t bo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xF5, 0xf, 0xf, true);   // quad_perm [1,1,3,3]
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int i = 2 * k + e;
                        const float km = ((e ? bo : be) << shl) >= thr_hi ? ks : 0.f;
                        const float p = fast_exp2(__builtin_fmaf(s16[i], sc, -lsv[i >> 2][i & 3]));
                        pd16[i] = p * km;
                        s16[i] = p * __builtin_fmaf(dp16[i], km, -dlv[i >> 2][i & 3]);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = fast_exp2(__builtin_fmaf(s16[i], sc, -lsv[i >> 2][i & 3]));
                    pd16[i] = p;
                    s16[i] = p * (dp16[i] - dlv[i >> 2][i & 3]);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = frag_from_acc(pd16, s2);
                const bf16x8 sf = frag_from_acc(s16, s2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 dof = frag_tr_at(Ds, tro, s2, dt * 32);
                    const bf16x8 qtf = frag_tr_at(Qs, tro, s2, dt * 32);
                    dV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, pf, dV[dt], 0, 0, 0);
                    dK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, sf, dK[dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (key < S) {
        const bool kvalid = !mk;
        const float kz = kvalid ? scale : 0.f, vz = kvalid ? 1.f : 0.f;
        const long outk = ((long)n * S + key) * ld + E + hd * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * h;
                if (d < DH) {
                    f32x4 a = {dK[dt][4 * g] * kz, dK[dt][4 * g + 1] * kz, dK[dt][4 * g + 2] * kz, dK[dt][4 * g + 3] * kz};
                    f32x4 b = {dV[dt][4 * g] * vz, dV[dt][4 * g + 1] * vz, dV[dt][4 * g + 2] * vz, dV[dt][4 * g + 3] * vz};
                    store4<IOB>(dqkv, outk + d, a);
                    store4<IOB>(dqkv, outk + E + d, b);
                }
            }
    }
}

// ======================================================================================================
// Split-operand ("bf16x3") attention for GG_PREC_BF16X3: fp32 tensors in memory, every MFMA operand as NS bf16 parts
// (v = hi + lo (+ lo2)), each product tile as the 3 (NS = 2) or 6 (NS = 3) part products down to 2^-16 of the leading one,
// fp32 accumulate.  Structure = the streaming row-major kernels above (resident tile per wave in registers, the other side
// through double-buffered LDS chunks, transposed operands through ds_read_b64_tr_b16, same masks, same dropout stream), with
// one LDS image per operand PART.  NS = 3 in the forward pass (fp32-grade context rows: they decide ReLU gates downstream),
// NS = 2 in the backward kernels (see tlin3.hip).  A wave whose tile lies wholly past the end skips the arithmetic but
// keeps staging (S = 257: the ninth query tile is one row).
// ======================================================================================================
template <int NS>
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&out)[NS]) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j];
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
        __bf16 b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = (__bf16)r[j];
        const bf16x2_t w0 = {b[0], b[1]}, w1 = {b[2], b[3]}, w2 = {b[4], b[5]}, w3 = {b[6], b[7]};
        const u32x4 w = {__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w1), __builtin_bit_cast(unsigned, w2),
                         __builtin_bit_cast(unsigned, w3)};
        out[sp] = __builtin_bit_cast(bf16x8, w);
        if (sp + 1 < NS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] -= (float)b[j];
        }
    }
}
template <int NS>
__device__ __forceinline__ void split_acc(const f32x16& a, int s2, bf16x8 (&out)[NS]) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a[8 * s2 + j];
    split8<NS>(v, out);
}
// acc += a b over the part products, smallest terms first
template <int NS>
__device__ __forceinline__ void mma_parts(f32x16& acc, const bf16x8 (&a)[NS], const bf16x8 (&b)[NS]) {
    if constexpr (NS == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}
// 8 consecutive fp32 values of a tensor as NS fragments
template <int NS>
__device__ __forceinline__ void load_parts(const float* p, bf16x8 (&out)[NS]) {
    float v[8];
    const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
    split8<NS>(v, out);
}
// one chunk (CK rows x DH columns of two fp32 operands) in flight: registers -> NS LDS images per operand
template <int DH, int CK, int NS>
struct ChunkRegs3 {
    static constexpr int CPR = DH / 8, NP = CK * CPR / 256 > 0 ? CK * CPR / 256 : 1, LD = DH + 8, IMG = CK * LD;
    f32x4 a[NP][2], b[NP][2];
    __device__ __forceinline__ void load(const float* XA, long lda, const float* XB, long ldb, int row0, int S, int tid) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int p = min(tid + 256 * u, CK * CPR - 1);
            const long row = min(row0 + p / CPR, S - 1);
            const float* pa = XA + row * lda + 8 * (p % CPR);
            const float* pb = XB + row * ldb + 8 * (p % CPR);
            a[u][0] = *reinterpret_cast<const f32x4*>(pa); a[u][1] = *reinterpret_cast<const f32x4*>(pa + 4);
            b[u][0] = *reinterpret_cast<const f32x4*>(pb); b[u][1] = *reinterpret_cast<const f32x4*>(pb + 4);
        }
    }
    // images of part sp: A at As + sp * part_stride, B at Bs + sp * part_stride; rows >= S zero
    __device__ __forceinline__ void store(__bf16* As, __bf16* Bs, int part_stride, int row0, int S, int tid) const {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int p = tid + 256 * u;
            if (p < CK * CPR) {
                const int row = p / CPR, c8 = p % CPR;
                const bool in = row0 + row < S;
                float va[8], vb[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    va[j] = in ? a[u][0][j] : 0.f; va[4 + j] = in ? a[u][1][j] : 0.f;
                    vb[j] = in ? b[u][0][j] : 0.f; vb[4 + j] = in ? b[u][1][j] : 0.f;
                }
                bf16x8 pa[NS], pb[NS];
                split8<NS>(va, pa);
                split8<NS>(vb, pb);
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) {
                    *reinterpret_cast<bf16x8*>(As + sp * part_stride + row * LD + 8 * c8) = pa[sp];
                    *reinterpret_cast<bf16x8*>(Bs + sp * part_stride + row * LD + 8 * c8) = pb[sp];
                }
            }
        }
    }
};
template <int NS> struct X3Chunk { static constexpr int CK = NS == 3 ? 64 : 128; };
// LDS: per part [2 buffers][2 operands][CK * (DH + 8)] bf16, then the key flags
template <int NS>
size_t x3_smem(int S, int DH, int ck = X3Chunk<NS>::CK) {
    const int Sp = (S + 31) / 32 * 32;
    return (size_t)NS * 4 * ck * (DH + 8) * 2 + Sp + 64 + 16;
}

// CK: keys per LDS chunk.  With CK = 32 the three-part forward kernel's images are 55 KB and its 229 registers leave room for a second
// workgroup per CU (two waves per SIMD instead of one); launch_bounds' second argument only caps the register budget
template <int DH, int NS, int CK = X3Chunk<NS>::CK>
__global__ __launch_bounds__(256, CK == 32 ? 2 : 1) void attn_fwd_x3_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ mask, int mask_B,
                                                            float* __restrict__ ctx, float* __restrict__ lse2, int S, int E, int nh,
                                                            DropKey drop_in, int qkv_B, int npairs, int nqg) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Sp = (S + 31) / 32 * 32, R = rm_rows(S);
    constexpr int LDK = DH + 8, IMG = CK * LDK, PART = 4 * IMG;         // part sp: [buf][K | V][IMG]
    __bf16* L0 = reinterpret_cast<__bf16*>(smem_raw);
    uint8_t* Ms = reinterpret_cast<uint8_t*>(L0 + NS * PART);
    uint8_t* Mt = Ms + Sp;
    constexpr int DT = (DH + 31) / 32, KS = DH / 16;

    const int pair = ((int)(blockIdx.x >> 3) / nqg) * 8 + (int)(blockIdx.x & 7);
    const int qg = (int)(blockIdx.x >> 3) % nqg;
    if (pair >= npairs) return;
    const int n = pair / nh, hd = pair % nh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const float* qb = qkv + (long)(n % qkv_B) * S * ld + hd * DH;
    const long moff = (long)(n % mask_B) * S;
    const int nkt = Sp / 32;
    const int nchunks = (R + CK - 1) / CK;
    const int qt = 4 * qg + wave;
    const bool live = qt * 32 < S;                      // wave-uniform
    const int q = qt * 32 + c;
    const int qc = min(q, S - 1);

    bf16x8 qf[KS][NS];
#pragma unroll
    for (int s = 0; s < KS; ++s) load_parts<NS>(qb + (long)qc * ld + 16 * s + 8 * h, qf[s]);
    uint8_t mf[MASK_U];
    mask_request<256>(mf, mask, moff, S, tid);
    ChunkRegs3<DH, CK, NS> cr;
    cr.load(qb + E, ld, qb + 2 * E, ld, 0, S, tid);
    mask_flags<256>(Ms, Mt, mf, mask, moff, S, Sp, tid);
    cr.store(L0, L0 + IMG, PART, 0, S, tid);
    if (nchunks > 1) cr.load(qb + E, ld, qb + 2 * E, ld, CK, S, tid);
    __syncthreads();

    const float sc = rsqrtf((float)DH) * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    float m = -INFINITY, l = 0.f;
    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[dt][i] = 0.f;
    const uint32_t srow = drop_state(drop, (((uint64_t)pair * S + (uint64_t)q) * (uint64_t)drop_attn_ld(S)) >> 1) + (uint32_t)(2 * h) * DROP_PHI;
    TileOff toff;
    toff.init(LDK, CK - 1, lane);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) cr.store(L0 + (buf ^ 1) * 2 * IMG, L0 + (buf ^ 1) * 2 * IMG + IMG, PART, (ch + 1) * CK, S, tid);
        if (ch + 2 < nchunks) cr.load(qb + E, ld, qb + 2 * E, ld, (ch + 2) * CK, S, tid);
        const __bf16* Ks = L0 + buf * 2 * IMG;          // part sp at + sp * PART
        const __bf16* Vs = Ks + IMG;
        toff.rlim = min(CK, R - ch * CK) - 1;
        const int kt_end = min(nkt, (ch + 1) * (CK / 32));
        if (live)
        for (int kt = ch * (CK / 32); kt < kt_end; ++kt) {
            int tro[4], kro;
            toff.tile((kt - ch * (CK / 32)) * 32, lane, tro, kro);
            f32x16 s16;
#pragma unroll
            for (int i = 0; i < 16; ++i) s16[i] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 kf[NS];
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) kf[sp] = *reinterpret_cast<const bf16x8*>(Ks + sp * PART + kro + 16 * s);
                mma_parts<NS>(s16, kf, qf[s]);
            }
            float mt = -INFINITY;
            if (__builtin_amdgcn_readfirstlane((int)Mt[kt])) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 32 + acc_row(i, h);
                    s16[i] = Ms[key] ? -INFINITY : s16[i] * sc;
                    mt = fmaxf(mt, s16[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s16[i] *= sc;
                    mt = fmaxf(mt, s16[i]);
                }
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const bool move = mt > m + 8.f;            // lazy reference update, as in attn_fwd_kernel
            if (__builtin_amdgcn_ballot_w64(move) != 0) {
                const float mn = move ? mt : m;
                const float alpha = (m == -INFINITY) ? 0.f : fast_exp2(m - mn);
                l *= alpha;
                m = mn;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
            }
            const float mref = (m == -INFINITY) ? 0.f : m;
            float lt = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = exp2f(s16[i] - mref);      // the accurate exponential: this kernel's results are held to fp32 grade
                lt += p;
                s16[i] = p;
            }
            lt += __shfl_xor(lt, 32, 64);
            l += lt;
            if (drop.p > 0.f) {
                const uint32_t skt = srow + (uint32_t)(kt * 16) * DROP_PHI;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t b0 = drop_bits(skt + (uint32_t)(4 * g) * DROP_PHI), b1 = drop_bits(skt + (uint32_t)(4 * g + 1) * DROP_PHI);
                    s16[4 * g + 0] = drop_keep_even(b0, drop.thr) ? s16[4 * g + 0] : 0.f;
                    s16[4 * g + 1] = drop_keep_odd(b0, drop.thr) ? s16[4 * g + 1] : 0.f;
                    s16[4 * g + 2] = drop_keep_even(b1, drop.thr) ? s16[4 * g + 2] : 0.f;
                    s16[4 * g + 3] = drop_keep_odd(b1, drop.thr) ? s16[4 * g + 3] : 0.f;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf[NS];
                split_acc<NS>(s16, s2, pf);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    bf16x8 vf[NS];
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) vf[sp] = frag_tr_at(Vs + sp * PART, tro, s2, dt * 32);
                    mma_parts<NS>(O[dt], vf, pf);
                }
            }
        }
        __syncthreads();
    }
    if (q < S) {
        const float inv = ks / l;
        const long out = ((long)n * S + q) * E + hd * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * h;
                if (d < DH) {
                    f32x4 v = {O[dt][4 * g] * inv, O[dt][4 * g + 1] * inv, O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv};
                    *reinterpret_cast<f32x4*>(ctx + out + d) = v;
                }
            }
        if (h == 0) lse2[(long)pair * S + q] = m + log2f(l);
    }
}

// dQ: the wave's query tile (Q, dO parts, lse, delta in registers); K and V chunks stream
// CK / OCC: keys per LDS chunk and workgroups per CU the register budget is cut for (production width, two parts: 64 keys = 74 KB and
// <= 256 registers put two workgroups on a CU; the default is the one-workgroup form)
template <int DH, int NS, int CK = X3Chunk<NS>::CK, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void attn_bwd_dq_x3_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                               const float* __restrict__ dctx, const float* __restrict__ lse2,
                                                               float* __restrict__ delta, const uint8_t* __restrict__ mask, int mask_B,
                                                               float* __restrict__ dqkv, int S, int E, int nh, DropKey drop_in, int qkv_B,
                                                               int npairs, int nqg) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Sp = (S + 31) / 32 * 32, R = rm_rows(S);
    constexpr int LDK = DH + 8, IMG = CK * LDK, PART = 4 * IMG;
    __bf16* L0 = reinterpret_cast<__bf16*>(smem_raw);
    uint8_t* Ms = reinterpret_cast<uint8_t*>(L0 + NS * PART);
    uint8_t* Mt = Ms + Sp;
    constexpr int DT = (DH + 31) / 32, KS = DH / 16;

    const int pair = ((int)(blockIdx.x >> 3) / nqg) * 8 + (int)(blockIdx.x & 7);
    const int qg = (int)(blockIdx.x >> 3) % nqg;
    if (pair >= npairs) return;
    const int n = pair / nh, hd = pair % nh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const float* qb = qkv + (long)(n % qkv_B) * S * ld + hd * DH;
    const long moff = (long)(n % mask_B) * S;
    const int nkt = Sp / 32;
    const int nchunks = (R + CK - 1) / CK;
    const int qt = 4 * qg + wave;
    const bool live = qt * 32 < S;
    const int q = qt * 32 + c;
    const int qc = min(q, S - 1);

    bf16x8 qf[KS][NS], df[KS][NS];
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const long off = ((long)n * S + qc) * E + hd * DH + 16 * s + 8 * h;
        load_parts<NS>(qb + (long)qc * ld + 16 * s + 8 * h, qf[s]);
        float dv[8], ov[8];
        load_f32x8<false>(dctx, off, dv);
        load_f32x8<false>(ctx, off, ov);
        split8<NS>(dv, df[s]);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += dv[j] * ov[j];
    }
    const float L2raw = lse2[(long)pair * S + qc];
    uint8_t mf[MASK_U];
    mask_request<256>(mf, mask, moff, S, tid);
    ChunkRegs3<DH, CK, NS> cr;
    cr.load(qb + E, ld, qb + 2 * E, ld, 0, S, tid);
    mask_flags<256>(Ms, Mt, mf, mask, moff, S, Sp, tid);
    cr.store(L0, L0 + IMG, PART, 0, S, tid);
    if (nchunks > 1) cr.load(qb + E, ld, qb + 2 * E, ld, CK, S, tid);
    dl += __shfl_xor(dl, 32, 64);
    if (q < S && h == 0) delta[(long)pair * S + q] = dl;      // consumed by the dK/dV kernel
    const float L2 = q < S ? L2raw : 0.f;
    __syncthreads();

    const float scale = rsqrtf((float)DH);
    const float sc = scale * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    f32x16 dQ[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dQ[dt][i] = 0.f;
    const uint32_t srow = drop_state(drop, (((uint64_t)pair * S + (uint64_t)q) * (uint64_t)drop_attn_ld(S)) >> 1) + (uint32_t)(2 * h) * DROP_PHI;
    TileOff toff;
    toff.init(LDK, CK - 1, lane);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) cr.store(L0 + (buf ^ 1) * 2 * IMG, L0 + (buf ^ 1) * 2 * IMG + IMG, PART, (ch + 1) * CK, S, tid);
        if (ch + 2 < nchunks) cr.load(qb + E, ld, qb + 2 * E, ld, (ch + 2) * CK, S, tid);
        const __bf16* Ks = L0 + buf * 2 * IMG;
        const __bf16* Vs = Ks + IMG;
        toff.rlim = min(CK, R - ch * CK) - 1;
        const int kt_end = min(nkt, (ch + 1) * (CK / 32));
        if (live)
        for (int kt = ch * (CK / 32); kt < kt_end; ++kt) {
            int tro[4], krow;
            toff.tile((kt - ch * (CK / 32)) * 32, lane, tro, krow);
            f32x16 s16, dp16;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s16[i] = 0.f; dp16[i] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 kf[NS], vf[NS];
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) {
                    kf[sp] = *reinterpret_cast<const bf16x8*>(Ks + sp * PART + krow + 16 * s);
                    vf[sp] = *reinterpret_cast<const bf16x8*>(Vs + sp * PART + krow + 16 * s);
                }
                mma_parts<NS>(s16, kf, qf[s]);
                mma_parts<NS>(dp16, vf, df[s]);
            }
            if (drop.p > 0.f) {
                const uint32_t skt = srow + (uint32_t)(kt * 16) * DROP_PHI;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t b0 = drop_bits(skt + (uint32_t)(4 * g) * DROP_PHI), b1 = drop_bits(skt + (uint32_t)(4 * g + 1) * DROP_PHI);
                    dp16[4 * g + 0] = drop_keep_even(b0, drop.thr) ? dp16[4 * g + 0] * ks : 0.f;
                    dp16[4 * g + 1] = drop_keep_odd(b0, drop.thr) ? dp16[4 * g + 1] * ks : 0.f;
                    dp16[4 * g + 2] = drop_keep_even(b1, drop.thr) ? dp16[4 * g + 2] * ks : 0.f;
                    dp16[4 * g + 3] = drop_keep_odd(b1, drop.thr) ? dp16[4 * g + 3] * ks : 0.f;
                }
            }
            if (__builtin_amdgcn_readfirstlane((int)Mt[kt])) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 32 + acc_row(i, h);
                    const float p = Ms[key] ? 0.f : exp2f(__builtin_fmaf(s16[i], sc, -L2));
                    s16[i] = p * (dp16[i] - dl);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) s16[i] = exp2f(__builtin_fmaf(s16[i], sc, -L2)) * (dp16[i] - dl);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 sf[NS];
                split_acc<NS>(s16, s2, sf);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    bf16x8 kf[NS];
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) kf[sp] = frag_tr_at(Ks + sp * PART, tro, s2, dt * 32);
                    mma_parts<NS>(dQ[dt], kf, sf);
                }
            }
        }
        __syncthreads();
    }
    if (q < S) {
        const long out = ((long)n * S + q) * ld + hd * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * h;
                if (d < DH) {
                    f32x4 v = {dQ[dt][4 * g] * scale, dQ[dt][4 * g + 1] * scale, dQ[dt][4 * g + 2] * scale, dQ[dt][4 * g + 3] * scale};
                    *reinterpret_cast<f32x4*>(dqkv + out + d) = v;
                }
            }
    }
}

// dK / dV: the wave's key tile (K, V parts in registers); Q and dO chunks with their lse / delta rows stream (64-row chunks)
template <int NS>
size_t x3_dkv_smem(int DH) { return (size_t)NS * 4 * 64 * (DH + 8) * 2 + (size_t)4 * 64 * 4; }
template <int DH, int NS, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void attn_bwd_dkv_x3_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                                const float* __restrict__ lse2, const float* __restrict__ delta,
                                                                const uint8_t* __restrict__ mask, int mask_B, float* __restrict__ dqkv,
                                                                int S, int E, int nh, DropKey drop_in, int qkv_B, int npairs, int nkg) {
    const DropKey drop = drop_live(drop_in);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int CK = 64;
    const int Sp = (S + 31) / 32 * 32, R = rm_rows(S);
    constexpr int LDK = DH + 8, IMG = CK * LDK, PART = 4 * IMG;         // part sp: [buf][Q | dO][IMG]
    __bf16* L0 = reinterpret_cast<__bf16*>(smem_raw);
    float* Lb = reinterpret_cast<float*>(L0 + NS * PART);       // [2][CK] log-sum-exp rows of the chunk
    float* Eb = Lb + 2 * CK;                                    // [2][CK] delta rows
    constexpr int DT = (DH + 31) / 32, KS = DH / 16;

    const int pair = ((int)(blockIdx.x >> 3) / nkg) * 8 + (int)(blockIdx.x & 7);
    const int kg = (int)(blockIdx.x >> 3) % nkg;
    if (pair >= npairs) return;
    const long nhid = pair;
    const int n = pair / nh, hd = pair % nh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const long ld = 3L * E;
    const float* qb = qkv + (long)(n % qkv_B) * S * ld + hd * DH;
    const float* db = dctx + (long)n * S * E + hd * DH;
    const int nqt = Sp / 32;
    const int nchunks = (R + CK - 1) / CK;
    const int ktile = 4 * kg + wave;
    const bool live = ktile * 32 < S;
    const int key = ktile * 32 + c;
    const int keyc = min(key, S - 1);                  // keys past the end re-read the last key: zeroed below, never stored
    bf16x8 kf[KS][NS], vf[KS][NS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        load_parts<NS>(qb + (long)keyc * ld + E + 16 * s + 8 * h, kf[s]);
        load_parts<NS>(qb + (long)keyc * ld + 2 * E + 16 * s + 8 * h, vf[s]);
    }
    const uint8_t mk = mask ? mask[(long)(n % mask_B) * S + keyc] : (uint8_t)0;
    ChunkRegs3<DH, CK, NS> cr;
    float lr = 0.f, er = 0.f;                          // this thread's lse / delta row of the chunk in flight (threads 0 .. CK-1)
    auto load_chunk = [&](int row0) {
        cr.load(qb, ld, db, (long)E, row0, S, tid);
        const long r = nhid * S + min(row0 + (tid & (CK - 1)), S - 1);
        lr = lse2[r];
        er = delta[r];
    };
    auto store_chunk = [&](int b, int row0) {
        cr.store(L0 + b * 2 * IMG, L0 + b * 2 * IMG + IMG, PART, row0, S, tid);
        if (tid < CK) {
            Lb[b * CK + tid] = row0 + tid < S ? lr : 0.f;
            Eb[b * CK + tid] = row0 + tid < S ? er : 0.f;
        }
    };
    load_chunk(0);
    store_chunk(0, 0);
    if (nchunks > 1) load_chunk(CK);
    __syncthreads();

    const float scale = rsqrtf((float)DH);
    const float sc = scale * LOG2E;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    const uint64_t Sd = (uint64_t)drop_attn_ld(S);
    f32x16 dK[DT], dV[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dK[dt][i] = 0.f; dV[dt][i] = 0.f; }
    TileOff toff;
    toff.init(LDK, CK - 1, lane);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) store_chunk(buf ^ 1, (ch + 1) * CK);
        if (ch + 2 < nchunks) load_chunk((ch + 2) * CK);
        const __bf16* Qs = L0 + buf * 2 * IMG;
        const __bf16* Ds = Qs + IMG;
        const float* Ls = Lb + buf * CK;
        const float* Dl = Eb + buf * CK;
        toff.rlim = min(CK, R - ch * CK) - 1;
        const int qt_end = min(nqt, (ch + 1) * (CK / 32));
        if (live)
        for (int qt = ch * (CK / 32); qt < qt_end; ++qt) {       // rows past the end are zero rows of Q and dO with lse = delta = 0
            const int ql = (qt - ch * (CK / 32)) * 32;
            int tro[4], qrow;
            toff.tile(ql, lane, tro, qrow);
            f32x16 s16, dp16;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s16[i] = 0.f; dp16[i] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 qa[NS], da[NS];
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) {
                    qa[sp] = *reinterpret_cast<const bf16x8*>(Qs + sp * PART + qrow + 16 * s);
                    da[sp] = *reinterpret_cast<const bf16x8*>(Ds + sp * PART + qrow + 16 * s);
                }
                mma_parts<NS>(s16, qa, kf[s]);          // S[q, key]: query rows in registers, the key on the lane
                mma_parts<NS>(dp16, da, vf[s]);
            }
            f32x16 pd16;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qr = ql + acc_row(i, h);             // row inside the chunk
                const float p = exp2f(__builtin_fmaf(s16[i], sc, -Ls[qr]));
                float km = 1.f;
                if (drop.p > 0.f) km = drop_factor(drop, (nhid * (uint64_t)S + (uint64_t)(ch * CK + qr)) * Sd + (uint64_t)key, ks);
                pd16[i] = p * km;
                s16[i] = p * __builtin_fmaf(dp16[i], km, -Dl[qr]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf[NS], sf[NS];
                split_acc<NS>(pd16, s2, pf);
                split_acc<NS>(s16, s2, sf);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    bf16x8 dof[NS], qtf[NS];
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) {
                        dof[sp] = frag_tr_at(Ds + sp * PART, tro, s2, dt * 32);
                        qtf[sp] = frag_tr_at(Qs + sp * PART, tro, s2, dt * 32);
                    }
                    mma_parts<NS>(dV[dt], dof, pf);
                    mma_parts<NS>(dK[dt], qtf, sf);
                }
            }
        }
        __syncthreads();
    }
    if (key < S) {
        const bool kvalid = !mk;
        const float kz = kvalid ? scale : 0.f, vz = kvalid ? 1.f : 0.f;
        const long outk = ((long)n * S + key) * ld + E + hd * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * h;
                if (d < DH) {
                    f32x4 a = {dK[dt][4 * g] * kz, dK[dt][4 * g + 1] * kz, dK[dt][4 * g + 2] * kz, dK[dt][4 * g + 3] * kz};
                    f32x4 b = {dV[dt][4 * g] * vz, dV[dt][4 * g + 1] * vz, dV[dt][4 * g + 2] * vz, dV[dt][4 * g + 3] * vz};
                    *reinterpret_cast<f32x4*>(dqkv + outk + d) = a;
                    *reinterpret_cast<f32x4*>(dqkv + outk + E + d) = b;
                }
            }
    }
}

size_t fwd_smem(int S, int DH) {
    const int Sp = (S + 31) / 32 * 32;
    return (size_t)Sp * (DH + 8) * 2 + (size_t)DH * (Sp + TPAD) * 2 + Sp + 64 + 16 + (size_t)3 * CO_MAXQ * (DH + 2) * 4;
}
size_t dq_smem(int S, int DH) {
    const int Sp = (S + 31) / 32 * 32;
    return (size_t)2 * Sp * (DH + 8) * 2 + (size_t)DH * (Sp + TPAD) * 2 + Sp + 64 + 16 + (size_t)3 * CO_MAXQ * (DH + 2) * 4;
}

size_t rm_smem(int S, int DH, int NW) {          // attn_fwd_rm_kernel / attn_bwd_dq_rm_kernel: K and V row-major, ceil8(S) rows
    const int Sp = (S + 31) / 32 * 32, R = rm_rows(S);
    return (size_t)2 * R * (DH + 8) * 2 + Sp + 64 + 16 + (size_t)(NW - 1) * CO_MAXQ * (DH + 2) * 4;
}
size_t dkv_rm_smem(int S, int DH, int NW) {
    const int Sp = (S + 31) / 32 * 32, R = rm_rows(S);
    return (size_t)2 * R * (DH + 8) * 2 + (size_t)2 * Sp * 4 + (size_t)(NW - 1) * CO_MAXQ * 2 * DH * 4;
}
// waves per workgroup of the row-major kernels, 0 = off (GG_ATTN_V1).  Measured at S = 257 / dh = 64 / B = 256 (tools/attn_ab.sh):
// forward 155 us (kernels above) -> 146 us (4 waves) -> 132 us (8 waves, 128 registers); dQ 211 us -> 135 us (4 waves with the
// next-tile request) - the 8-wave dQ kernel needs more than its 128 registers and spills (208 us).
int rm_waves(bool fwd) {
    static const bool off = getenv("GG_ATTN_V1") != nullptr;
    static const int nwf = [] { const char* e = getenv("GG_ATTN_FWD_NW"); return e && atoi(e) == 4 ? 4 : 8; }();
    static const int nwd = [] { const char* e = getenv("GG_ATTN_DQ_NW"); return e && atoi(e) == 8 ? 8 : 4; }();
    return off ? 0 : (fwd ? nwf : nwd);
}

size_t dq2_smem(int S, int DH) {
    const int Sp = (S + 31) / 32 * 32;
    const int CK = ((Sp / 32 + 1) / 2) * 32;
    return (size_t)2 * CK * (DH + 8) * 2 + (size_t)DH * (CK + TPAD) * 2 + Sp + 64 + 16 + (size_t)3 * CO_MAXQ * (DH + 2) * 4;
}
// long-sequence kernels: keys stream through LDS in chunks of CK (a multiple of 32) chosen so that two workgroups share a CU
constexpr int LONG_MAX_S = 2048;                 // 64 key-tile flags
size_t fwd_long_smem(int S, int DH, int CK) {
    const int Sp = (S + 31) / 32 * 32;
    return (size_t)CK * (DH + 8) * 2 + (size_t)DH * (CK + TPAD) * 2 + Sp + 64 + 16;
}
size_t dq_long_smem(int S, int DH, int CK) {
    const int Sp = (S + 31) / 32 * 32;
    return (size_t)2 * CK * (DH + 8) * 2 + (size_t)DH * (CK + TPAD) * 2 + Sp + 64 + 16;
}
template <typename F>
int long_chunk(int S, int DH, F smem) {         // largest chunk with 2 * smem <= 160 KB (at least one key tile)
    int ck = 32;
    while (ck + 32 <= 512 && 2 * smem(S, DH, ck + 32) <= 160 * 1024) ck += 32;
    return ck;
}
bool short_ok(int S, int dh) { return dq_smem(S, dh) <= 160 * 1024 && fwd_smem(S, dh) <= 160 * 1024; }
template <typename K>
int set_smem(K kernel, size_t bytes) {
    GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
}  // namespace

bool flash_attn_supported(int S, int E, int nh) {
    if (nh <= 0 || E % nh) return false;
    const int dh = E / nh;
    if (!(dh == 16 || dh == 32 || dh == 64)) return false;
    if (E % 4) return false;
    return short_ok(S, dh) || S <= LONG_MAX_S;      // beyond the LDS-resident range: the key-streaming kernels
}

// Name of the kernel flash_attn_fwd (which = 0) / the dQ (1) / the dK/dV (2) half of flash_attn_bwd launches for this shape:
// the profile classes of the engine and the PMC summaries are keyed by the kernels' own names.  Mirrors the dispatch below.
const char* flash_attn_kernel_name(int which, int S, int E, int nh) {
    if (!flash_attn_supported(S, E, nh)) return "";
    const int dh = E / nh;
    // A/B switches: read once per process (this function runs on every attention enqueue when the profiler is on)
    static const bool env_long = getenv("GG_ATTN_LONG") != nullptr, env_long_v1 = getenv("GG_ATTN_LONG_V1") != nullptr,
                      env_dq_long_v1 = getenv("GG_ATTN_DQ_LONG_V1") != nullptr, env_dq1 = getenv("GG_ATTN_DQ1") != nullptr,
                      env_dkv_v1 = getenv("GG_ATTN_DKV_V1") != nullptr;
    const bool lng = env_long || !short_ok(S, dh);
    const bool strm = lng && !env_long_v1;
    const int nqt = (S + 31) / 32;
    if (which == 0) {
        const int nw = rm_waves(true);
        if (!lng && nw && rm_smem(S, dh, nw) <= 160 * 1024) return "attn_fwd_rm_kernel";
        return strm ? "attn_fwd_stream_kernel" : lng ? "attn_fwd_long_kernel" : "attn_fwd_kernel";
    }
    const int nw = rm_waves(false);
    if (which == 1) {
        if (!lng && nw && rm_smem(S, dh, nw) <= 160 * 1024) return "attn_bwd_dq_rm_kernel";
        if (strm && !env_dq_long_v1) return "attn_bwd_dq_stream_kernel";
        if (lng) return "attn_bwd_dq_long_kernel";
        const bool coop = (nqt % 4 == 1) && nqt > 1 && (S - 32 * (nqt - 1)) <= CO_MAXQ;
        const int items_q = (coop ? nqt - 1 : nqt) + (coop ? 4 : 0);
        const bool dq2 = !env_dq1 && nqt >= 2 && items_q <= 4 * DQ_SLOTS && 2 * dq2_smem(S, dh) <= 160 * 1024;
        return dq2 ? "attn_bwd_dq2_kernel" : "attn_bwd_dq_kernel";
    }
    if (strm) return "attn_bwd_dkv_stream_kernel";
    if (!lng && nw && !env_dkv_v1 && 2 * dkv_rm_smem(S, dh, 4) <= 160 * 1024) return "attn_bwd_dkv_rm_kernel";
    return "attn_bwd_dkv_kernel";
}

// dropout compiled in or out (DROPV inside the argument): the dQ kernels and the streaming dK/dV kernel gain 5 % from a tile
// loop without the run-time test of drop.p (one basic block to schedule); forward and the resident dK/dV kernel LOSE 2 - 4 %
// (the merged block raises their register pressure) and keep the run-time test
#define GG_DROPSW(...)                                                                 \
    do {                                                                               \
        if (drop.p > 0.f) { constexpr bool DROPV = true; __VA_ARGS__; }                \
        else { constexpr bool DROPV = false; __VA_ARGS__; }                            \
    } while (0)

int flash_attn_fwd(const void* qkv, const uint8_t* mask, int mask_B, void* ctx, float* lse2, long N, int S, int E, int nh,
                   DropKey drop, int io_bf16, hipStream_t st, long qkv_B) {
    const int qB = (int)(qkv_B > 0 ? qkv_B : N);
    GG_REQUIRE(flash_attn_supported(S, E, nh), "flash attention: unsupported shape");
    const int dh = E / nh;
    static const bool force_long = getenv("GG_ATTN_LONG") != nullptr;      // A/B and tests: the key-streaming kernels at any S
    const bool lng = force_long || !short_ok(S, dh);
    const int ck = lng ? long_chunk(S, dh, fwd_long_smem) : 0;
    const size_t sm = lng ? fwd_long_smem(S, dh, ck) : fwd_smem(S, dh);
    const int nqg = ((S + 31) / 32 + 3) / 4;                  // long kernels: query groups of four tiles
    const dim3 grid(lng ? (unsigned)(((N * nh + 7) / 8) * 8 * nqg) : (unsigned)(N * nh));
    static const bool strm_off = getenv("GG_ATTN_LONG_V1") != nullptr;
    const bool strm = lng && !strm_off;                         // streaming row-major kernel: 8 query tiles per workgroup (8 waves x 1 or 4 x 2)
    static const bool strm8 = getenv("GG_ATTN_STREAM_NW4") == nullptr;
    const size_t sms = stream_smem(S, dh);
    const int nqgs = ((S + 31) / 32 + 4 * SQT - 1) / (4 * SQT);
    const dim3 grids((unsigned)(((N * nh + 7) / 8) * 8 * nqgs));
    const int nw = rm_waves(true);
    const bool rm = !lng && nw && rm_smem(S, dh, nw) <= 160 * 1024;
    const size_t smr = rm ? rm_smem(S, dh, nw) : 0;
#define GG_FWD(D, B)                                                                                          \
    do {                                                                                                      \
        if (rm && nw == 8) {                                                                                  \
            GG_TRY(set_smem(&attn_fwd_rm_kernel<D, B, 8>, smr)); hipLaunchKernelGGL((attn_fwd_rm_kernel<D, B, 8>), grid, dim3(512), smr, st, qkv, mask, mask_B, ctx, lse2, S, E, nh, drop, qB); \
        } else if (rm) {                                                                                      \
            GG_TRY(set_smem(&attn_fwd_rm_kernel<D, B, 4>, smr)); hipLaunchKernelGGL((attn_fwd_rm_kernel<D, B, 4>), grid, dim3(256), smr, st, qkv, mask, mask_B, ctx, lse2, S, E, nh, drop, qB); \
        } else if (lng && strm) {                                                                                    \
            if (strm8) {                                                                                      \
                GG_TRY(set_smem(&attn_fwd_stream_kernel<D, B, 8, 1>, sms)); hipLaunchKernelGGL((attn_fwd_stream_kernel<D, B, 8, 1>), grids, dim3(512), sms, st, qkv, mask, mask_B, ctx, lse2, S, E, nh, drop, qB, (int)(N * nh), nqgs); \
            } else {                                                                                          \
                GG_TRY(set_smem(&attn_fwd_stream_kernel<D, B, 4, 2>, sms)); hipLaunchKernelGGL((attn_fwd_stream_kernel<D, B, 4, 2>), grids, dim3(256), sms, st, qkv, mask, mask_B, ctx, lse2, S, E, nh, drop, qB, (int)(N * nh), nqgs); \
            }                                                                                                 \
        } else if (lng) {                                                                                            \
            GG_TRY(set_smem(&attn_fwd_long_kernel<D, B>, sm));                                                \
            hipLaunchKernelGGL((attn_fwd_long_kernel<D, B>), grid, dim3(256), sm, st, qkv, mask, mask_B, ctx, lse2, S, E, nh, drop, qB, ck, (int)(N * nh), nqg); \
        } else {                                                                                              \
            GG_TRY(set_smem(&attn_fwd_kernel<D, B>, sm));                                                     \
            hipLaunchKernelGGL((attn_fwd_kernel<D, B>), grid, dim3(256), sm, st, qkv, mask, mask_B, ctx, lse2, S, E, nh, drop, qB); \
        }                                                                                                     \
    } while (0)
    if (io_bf16) {
        if (dh == 64) GG_FWD(64, true);
        else if (dh == 32) GG_FWD(32, true);
        else GG_FWD(16, true);
    } else {
        if (dh == 64) GG_FWD(64, false);
        else if (dh == 32) GG_FWD(32, false);
        else GG_FWD(16, false);
    }
#undef GG_FWD
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

int flash_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse2, float* delta, const uint8_t* mask,
                   int mask_B, void* dqkv, long N, int S, int E, int nh, DropKey drop, int io_bf16, hipStream_t st, long qkv_B, hipEvent_t ev_mid) {
    const int qB = (int)(qkv_B > 0 ? qkv_B : N);
    GG_REQUIRE(flash_attn_supported(S, E, nh), "flash attention: unsupported shape");
    const int dh = E / nh;
    const size_t sm = dq_smem(S, dh);
    const dim3 grid((unsigned)(N * nh));
    const long items = N * nh * ((S + 31) / 32);       // one wave per (sample, head, key tile)
    // key-chunked dQ variant: two workgroups per CU when a wave has at most DQ_SLOTS query tiles and the halves fit
    const int nqt_ = (S + 31) / 32;
    const bool coop_ = (nqt_ % 4 == 1) && nqt_ > 1 && (S - 32 * (nqt_ - 1)) <= CO_MAXQ;
    const int items_q = (coop_ ? nqt_ - 1 : nqt_) + (coop_ ? 4 : 0);
    const size_t sm2 = dq2_smem(S, dh);
    static const bool dq2_off = getenv("GG_ATTN_DQ1") != nullptr;
    const bool use_dq2 = !dq2_off && nqt_ >= 2 && items_q <= 4 * DQ_SLOTS && 2 * sm2 <= 160 * 1024;
    static const bool force_long = getenv("GG_ATTN_LONG") != nullptr;
    const bool lng = force_long || !short_ok(S, dh);
    const int ck = lng ? long_chunk(S, dh, dq_long_smem) : 0;
    const size_t sml = lng ? dq_long_smem(S, dh, ck) : 0;
    const int nqgl = (nqt_ + 4 * DQ_SLOTS - 1) / (4 * DQ_SLOTS);
    const dim3 gridl((unsigned)(((N * nh + 7) / 8) * 8 * nqgl));
    static const bool strm_off = getenv("GG_ATTN_LONG_V1") != nullptr;
    static const bool strm_dq = getenv("GG_ATTN_DQ_LONG_V1") == nullptr;
    const bool strm = lng && !strm_off;                         // streaming row-major kernels: one query / key tile per wave, four per workgroup
    const size_t sms = stream_smem(S, dh), smks = dkv_stream_smem(dh);
    const int ngs = (nqt_ + 3) / 4;
    const dim3 grids((unsigned)(((N * nh + 7) / 8) * 8 * ngs));
    const int nw = rm_waves(false);
    const bool rm = !lng && nw && rm_smem(S, dh, nw) <= 160 * 1024;
    const size_t smr = rm ? rm_smem(S, dh, nw) : 0;
    static const bool dkv_v1 = getenv("GG_ATTN_DKV_V1") != nullptr;
    const size_t smk = dkv_rm_smem(S, dh, 4);
    const bool rmk = !lng && nw && !dkv_v1 && 2 * smk <= 160 * 1024;      // worth it only with two workgroups per CU
#define GG_BWD_RM(D, B, NW_, PF_)                                                                                           \
    do {                                                                                                                    \
        if (drop.p > 0.f) {                                                                                                 \
            GG_TRY(set_smem(&attn_bwd_dq_rm_kernel<D, B, NW_, PF_, true>, smr));                                            \
            hipLaunchKernelGGL((attn_bwd_dq_rm_kernel<D, B, NW_, PF_, true>), grid, dim3(64 * NW_), smr, st, qkv, ctx, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB); \
        } else {                                                                                                            \
            GG_TRY(set_smem(&attn_bwd_dq_rm_kernel<D, B, NW_, PF_, false>, smr));                                           \
            hipLaunchKernelGGL((attn_bwd_dq_rm_kernel<D, B, NW_, PF_, false>), grid, dim3(64 * NW_), smr, st, qkv, ctx, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB); \
        }                                                                                                                   \
    } while (0)
#define GG_BWD(D, B)                                                                                                        \
    do {                                                                                                                    \
        if (rm && nw == 8) {                                                                                         \
            GG_BWD_RM(D, B, 8, false);                                                                                      \
        } else if (rm) {                                                                                                    \
            GG_BWD_RM(D, B, 4, true);                                                                                       \
        } else if (lng && strm && strm_dq) {                                                                                \
            GG_DROPSW(GG_TRY(set_smem(&attn_bwd_dq_stream_kernel<D, B, DROPV>, sms)); hipLaunchKernelGGL((attn_bwd_dq_stream_kernel<D, B, DROPV>), grids, dim3(256), sms, st, qkv, ctx, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB, (int)(N * nh), ngs)); \
        } else if (lng) {                                                                                                          \
            GG_TRY(set_smem(&attn_bwd_dq_long_kernel<D, B>, sml));                                                          \
            hipLaunchKernelGGL((attn_bwd_dq_long_kernel<D, B>), gridl, dim3(256), sml, st, qkv, ctx, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB, ck, (int)(N * nh), nqgl); \
        } else if (use_dq2) {                                                                                                      \
            GG_TRY(set_smem(&attn_bwd_dq2_kernel<D, B>, sm2));                                                              \
            hipLaunchKernelGGL((attn_bwd_dq2_kernel<D, B>), grid, dim3(256), sm2, st, qkv, ctx, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB); \
        } else {                                                                                                            \
            GG_TRY(set_smem(&attn_bwd_dq_kernel<D, B>, sm));                                                                \
            hipLaunchKernelGGL((attn_bwd_dq_kernel<D, B>), grid, dim3(256), sm, st, qkv, ctx, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB); \
        }                                                                                                                   \
        if (ev_mid) GG_CHECK_HIP(hipEventRecord(ev_mid, st));   /* profiling: splits the pair into its two kernels */ \
        if (lng && strm) {                                                                                                  \
            GG_DROPSW(GG_TRY(set_smem(&attn_bwd_dkv_stream_kernel<D, B, DROPV>, smks)); hipLaunchKernelGGL((attn_bwd_dkv_stream_kernel<D, B, DROPV>), grids, dim3(256), smks, st, qkv, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB, (int)(N * nh), ngs)); \
        } else if (rmk) {                                                                                                   \
            GG_TRY(set_smem(&attn_bwd_dkv_rm_kernel<D, B, 4>, smk)); hipLaunchKernelGGL((attn_bwd_dkv_rm_kernel<D, B, 4>), grid, dim3(256), smk, st, qkv, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB); \
        } else {                                                                                                            \
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, B>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, qkv, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, items, qB); \
        }                                                                                                                   \
    } while (0)
    if (io_bf16) {
        if (dh == 64) GG_BWD(64, true);
        else if (dh == 32) GG_BWD(32, true);
        else GG_BWD(16, true);
    } else {
        if (dh == 64) GG_BWD(64, false);
        else if (dh == 32) GG_BWD(32, false);
        else GG_BWD(16, false);
    }
#undef GG_BWD
#undef GG_BWD_RM
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}


// ---- split-operand (bf16x3) attention: fp32 tensors, ns operand parts (2: three products per tile, 3: six) ---------------------
bool flash_attn_x3_supported(int S, int E, int nh) {
    if (nh <= 0 || E % nh || E % 4) return false;
    const int dh = E / nh;
    return (dh == 16 || dh == 32 || dh == 64) && S >= 1 && S <= LONG_MAX_S;
}
const char* flash_attn_x3_kernel_name(int which) {
    return which == 0 ? "attn_fwd_x3_kernel" : which == 1 ? "attn_bwd_dq_x3_kernel" : "attn_bwd_dkv_x3_kernel";
}
int flash_attn_fwd_x3(const float* qkv, const uint8_t* mask, int mask_B, float* ctx, float* lse2, long N, int S, int E, int nh,
                      DropKey drop, hipStream_t st, long qkv_B, int ns) {
    GG_REQUIRE(flash_attn_x3_supported(S, E, nh) && (ns == 2 || ns == 3), "split-operand attention: unsupported shape");
    const int qB = (int)(qkv_B > 0 ? qkv_B : N), dh = E / nh;
    const int nqg = ((S + 31) / 32 + 3) / 4;
    const dim3 grid((unsigned)(((N * nh + 7) / 8) * 8 * nqg));
#define GG_F3C(D, NS_, CK_)                                                                                              \
    do {                                                                                                               \
        const size_t sm = x3_smem<NS_>(S, D, CK_);                                                                     \
        GG_TRY(set_smem(&attn_fwd_x3_kernel<D, NS_, CK_>, sm));                                                         \
        hipLaunchKernelGGL((attn_fwd_x3_kernel<D, NS_, CK_>), grid, dim3(256), sm, st, qkv, mask, mask_B, ctx, lse2, S, E, nh, drop, qB, (int)(N * nh), nqg); \
    } while (0)
#define GG_F3(D, NS_) GG_F3C(D, NS_, X3Chunk<NS_>::CK)
    static const bool ck64 = getenv("GG_X3_FWD_CK64") != nullptr;       // A/B: the one-workgroup-per-CU form of the production-width forward kernel
    if (ns == 3) {
        if (dh == 64 && !ck64) GG_F3C(64, 3, 32);
        else if (dh == 64) GG_F3(64, 3); else if (dh == 32) GG_F3(32, 3); else GG_F3(16, 3);
    } else {
        if (dh == 64) GG_F3(64, 2); else if (dh == 32) GG_F3(32, 2); else GG_F3(16, 2);
    }
#undef GG_F3
#undef GG_F3C
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
int flash_attn_bwd_x3(const float* qkv, const float* ctx, const float* dctx, const float* lse2, float* delta, const uint8_t* mask,
                      int mask_B, float* dqkv, long N, int S, int E, int nh, DropKey drop, hipStream_t st, long qkv_B, int ns,
                      hipEvent_t ev_mid) {
    GG_REQUIRE(flash_attn_x3_supported(S, E, nh) && (ns == 2 || ns == 3), "split-operand attention: unsupported shape");
    const int qB = (int)(qkv_B > 0 ? qkv_B : N), dh = E / nh;
    const int ng = ((S + 31) / 32 + 3) / 4;
    const dim3 grid((unsigned)(((N * nh + 7) / 8) * 8 * ng));
#define GG_B3(D, NS_)                                                                                                    \
    do {                                                                                                               \
        const size_t sm = x3_smem<NS_>(S, D), smk = x3_dkv_smem<NS_>(D);                                                \
        GG_TRY(set_smem(&attn_bwd_dq_x3_kernel<D, NS_>, sm));                                                           \
        hipLaunchKernelGGL((attn_bwd_dq_x3_kernel<D, NS_>), grid, dim3(256), sm, st, qkv, ctx, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB, (int)(N * nh), ng); \
        if (ev_mid) GG_CHECK_HIP(hipEventRecord(ev_mid, st));                                                           \
        GG_TRY(set_smem(&attn_bwd_dkv_x3_kernel<D, NS_>, smk));                                                         \
        hipLaunchKernelGGL((attn_bwd_dkv_x3_kernel<D, NS_>), grid, dim3(256), smk, st, qkv, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB, (int)(N * nh), ng); \
    } while (0)
    // bit 0: dQ kernel, bit 1: dK / dV kernel in the two-workgroups-per-CU form (cfg3 bf16x3 step, interleaved: neither 74.6 ms,
    // dQ only 72.1, dK / dV only 72.0 - its 92 bytes of spills included -, both 69.7); GG_X3_BWD_OCC2=0: the one-workgroup forms
    static const int occ2 = getenv("GG_X3_BWD_OCC2") ? atoi(getenv("GG_X3_BWD_OCC2")) : 3;
    if (ns == 3) {
        if (dh == 64) GG_B3(64, 3); else if (dh == 32) GG_B3(32, 3); else GG_B3(16, 3);
    } else if (dh == 64 && occ2) {
        const size_t sm = (occ2 & 1) ? x3_smem<2>(S, 64, 64) : x3_smem<2>(S, 64), smk = x3_dkv_smem<2>(64);
        if (occ2 & 1) {
            GG_TRY(set_smem(&attn_bwd_dq_x3_kernel<64, 2, 64, 2>, sm));
            hipLaunchKernelGGL((attn_bwd_dq_x3_kernel<64, 2, 64, 2>), grid, dim3(256), sm, st, qkv, ctx, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB, (int)(N * nh), ng);
        } else {
            GG_TRY(set_smem(&attn_bwd_dq_x3_kernel<64, 2>, sm));
            hipLaunchKernelGGL((attn_bwd_dq_x3_kernel<64, 2>), grid, dim3(256), sm, st, qkv, ctx, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB, (int)(N * nh), ng);
        }
        if (ev_mid) GG_CHECK_HIP(hipEventRecord(ev_mid, st));
        if (occ2 & 2) {
            GG_TRY(set_smem(&attn_bwd_dkv_x3_kernel<64, 2, 2>, smk));
            hipLaunchKernelGGL((attn_bwd_dkv_x3_kernel<64, 2, 2>), grid, dim3(256), smk, st, qkv, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB, (int)(N * nh), ng);
        } else {
            GG_TRY(set_smem(&attn_bwd_dkv_x3_kernel<64, 2>, smk));
            hipLaunchKernelGGL((attn_bwd_dkv_x3_kernel<64, 2>), grid, dim3(256), smk, st, qkv, dctx, lse2, delta, mask, mask_B, dqkv, S, E, nh, drop, qB, (int)(N * nh), ng);
        }
    } else {
        if (dh == 64) GG_B3(64, 2); else if (dh == 32) GG_B3(32, 2); else GG_B3(16, 2);
    }
#undef GG_B3
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gg
