// Gradient-penalty kernel chain (R:351-374 and its double backward), closed form of SURVEY.md section 3.3.
//
// For the B interpolate rows, with the critic head  h1 = x^ W1x^T + c W1c^T + b1, a1 = act(h1), a2 = act(a1 W2^T + b2),
// out = a2 w3 + b3  and  m = act'(.) (1 where the post-activation value is positive, `slope` elsewhere):
//
//   gp_front_k : g1 = m1 * ((m2 * w3) W2)                                         [B,H]    (also zeroes nrm2, dg1pre)
//   gp_grad_k  : grad = g1 W1x  == autograd.grad(D(x^), x^)                        [B,G]    + nrm2[b] += sum_g grad^2 in the epilogue
//   gp_coef_k  : coef = w (2/B)(|grad|-1)/|grad| ; loss += mean((|grad|-1)^2) ; g1s = coef * g1
//   (caller)   : dW1x += g1s^T grad        (weight-gradient kernel, side stream)
//   (caller)   : dg1pre += grad W1x^T      (split-K GEMM, atomics into the zeroed buffer)
//   gp_tail_k  : du = m1 * coef * dg1pre ; Q[h,k] = sum_b m2[b,h] du[b,k] ;
//                dW2[h,k] += w3[h] Q[h,k]  (= g2^T du) ;  dw3[h] += sum_k W2[h,k] Q[h,k]  (= sum_b m2 * (du W2^T))
//
// Six launches per critic iteration (was ~18).  Everything here is exact fp32 in BOTH precision modes: the chain is
// launch- and HBM-bound (every tensor is <= B*G*4 bytes), the (|grad| - 1) cancellation is what the parity gate is most
// sensitive to, and bf16 operands would buy nothing.  grad is written once and read twice (dW1x, dg1pre); the row norms
// never re-read it.  The gene dimension is the contiguous one in W1x rows and in grad rows: every access below is a run of
// >= 128 consecutive bytes.
#include "gg_common.h"
#include "kernels.h"
#include <hip/hip_ext.h>
#include <cstdlib>

namespace gg {
namespace {

constexpr int TPB = 256;
// dispatch timestamps of the NEXT launch (gp_time_next), as in tlin.hip: the events are stamped at the kernel's own begin / end
hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
template <typename K, typename... A>
inline void launch(K kernel, dim3 grid, size_t smem, hipStream_t st, A... args) {
    if (g_ev0) {
        hipExtLaunchKernelGGL(kernel, grid, dim3(TPB), (unsigned)smem, st, g_ev0, g_ev1, 0, args...);
        g_ev0 = g_ev1 = nullptr;
    } else {
        hipLaunchKernelGGL(kernel, grid, dim3(TPB), smem, st, args...);
    }
}

__device__ __forceinline__ float actd(float post, float slope) { return post > 0.f ? 1.f : slope; }

// ---- g1 = m1 * ((m2 * w3) W2) : 4 rows per workgroup, thread = output column -------------------------------------
__global__ __launch_bounds__(TPB) void gp_front_k(const float* __restrict__ a1, const float* __restrict__ a2,
                                                   const float* __restrict__ w3, const float* __restrict__ W2,
                                                   float* __restrict__ g1, float* __restrict__ dg1pre, float* __restrict__ nrm2,
                                                   int B, int H, float slope) {
    extern __shared__ float g2s[];      // [4][H]
    const int b0 = blockIdx.x * 4;
    for (int i = threadIdx.x; i < 4 * H; i += TPB) {
        const int r = i / H, h = i - r * H, b = b0 + r;
        g2s[i] = b < B ? actd(a2[(long)b * H + h], slope) * w3[h] : 0.f;
    }
    if (threadIdx.x < 4 && b0 + threadIdx.x < B) nrm2[b0 + threadIdx.x] = 0.f;
    __syncthreads();
    for (int k = threadIdx.x; k < H; k += TPB) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int h = 0;
        for (; h + 8 <= H; h += 8) {          // eight independent L2 loads in flight per thread
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = W2[(long)(h + u) * H + k];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += g2s[r * H + h + u] * w[u];
        }
        for (; h < H; ++h) {
            const float w = W2[(long)h * H + k];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += g2s[r * H + h] * w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + r;
            if (b < B) {
                g1[(long)b * H + k] = actd(a1[(long)b * H + k], slope) * acc[r];
                dg1pre[(long)b * H + k] = 0.f;
            }
        }
    }
}

// ---- grad = g1 W1x (+ row sums of squares) : 64 x 64 tiles, 4 waves x one 32x32 fp32 MFMA accumulator -------------
// K (= H) is consumed in slabs of 64 through LDS; the next slab's global loads are issued before the current slab's
// MFMA chain (register double buffering), 16-byte loads when H, G and the leading dimension are multiples of 4.
constexpr int GT = 64, GK = 64;
template <bool VEC>
__global__ __launch_bounds__(TPB) void gp_grad_k(const float* __restrict__ g1, const float* __restrict__ W1, long ldw,
                                                  float* __restrict__ grad, float* __restrict__ nrm2, int B, int H, int G) {
    __shared__ __attribute__((aligned(16))) float As[GT][GK + 4];        // g1 tile  [row][k]
    __shared__ __attribute__((aligned(16))) float Bs[GK][GT + 4];        // W1x tile [k][gene]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (G + GT - 1) / GT;
    const int m0 = (blockIdx.x / tiles_n) * GT, n0 = (blockIdx.x % tiles_n) * GT;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // thread -> 4 quads of each operand tile: quad f = tid + 256*i, row = f >> 4, column quad = f & 15
    f32x4 ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + TPB * i, row = f >> 4, c4 = (f & 15) * 4;
            {
                const int gr = m0 + row, gk = k0 + c4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gr < B) {
                    const float* src = g1 + (long)gr * H + gk;
                    if (VEC) { if (gk < H) v = *reinterpret_cast<const f32x4*>(src); }
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (gk + j < H) v[j] = src[j];
                    }
                }
                ra[i] = v;
            }
            {
                const int gk = k0 + row, gc = n0 + c4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gk < H) {
                    const float* src = W1 + (long)gk * ldw + gc;
                    if (VEC) { if (gc < G) v = *reinterpret_cast<const f32x4*>(src); }
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (gc + j < G) v[j] = src[j];
                    }
                }
                rb[i] = v;
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + TPB * i, row = f >> 4, c4 = (f & 15) * 4;
            *reinterpret_cast<f32x4*>(&As[row][c4]) = ra[i];
            *reinterpret_cast<f32x4*>(&Bs[row][c4]) = rb[i];
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < H; k0 += GK) {
        stash();
        __syncthreads();
        if (k0 + GK < H) fetch(k0 + GK);
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[wm * 32 + r][kk + hh], Bs[kk + hh][wn * 32 + r], acc, 0, 0, 0);
        __syncthreads();
    }
    const int col = n0 + wn * 32 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        float v = (col < G && row < B) ? acc[i] : 0.f;
        if (col < G && row < B) grad[(long)row * G + col] = v;
        v *= v;
        // sum over the 32 lanes that hold this row (the half-wave hh)
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        if (r == 0 && row < B) atomicAdd(nrm2 + row, v);
    }
}

// ---- grad = g1 W1x, round 3: one 32-gene strip x up to 256 rows per workgroup, split-operand MFMA ---------------------------
// The kernel above feeds each fp32-input MFMA (64 cycles for 4 096 FLOP) from two scalar LDS reads behind bounds-guarded loads:
// 41 us for 0.66 GFLOP and 10.5 MB at B = 256, G = 5 000 - neither launch- nor bandwidth-bound, just slow.  Here
//   * grad^T [gene, row] = W1x^T [gene, k] g1^T [k, row] with both operands as THREE bf16 parts (hi, mid, lo: 24 bits) and the six
//     part products down to 2^-16 of the leading one, fp32 accumulate (tlin3.hip): fp32-grade results (the (|grad| - 1)
//     cancellation is what the parity gates are most sensitive to) at 192 cycles per 32 768 FLOP - 2.7 x the fp32-input rate;
//   * the whole [H, 32] strip of W1x is staged ONCE (unconditional clamped loads, all in flight together, split while written to
//     LDS as row-major part images) and read through ds_read_b64_tr_b16 as the A operand; g1 rows are the B operand straight
//     from global memory (8 consecutive k per lane: two 16-byte loads, L2-resident), split in registers;
//   * the accumulator has genes in registers and the batch row on the lane: the row's sum of squares is 16 in-register
//     multiply-adds and one cross-half shuffle, written per strip (no atomics: gp_coef_k adds the strips in a fixed order, so the
//     penalty is bit-reproducible), grad leaves as 16-byte stores.
// 157 workgroups at G = 5 000: one round on 256 CUs.
constexpr int GS = 32;                         // genes per workgroup
constexpr int GLD = GS;                        // bf16 per LDS row: 16 dwords - the 32 lanes of a transposing read (4 rows x 2 x 8 dwords) tile all 64 banks, as do the staging writes
typedef short s16x4g __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2g __attribute__((ext_vector_type(2)));
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8g(const float (&v)[8], bf16x8 (&out)[3]) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j];
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
        __bf16 b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = (__bf16)r[j];
        const bf16x2g w0 = {b[0], b[1]}, w1 = {b[2], b[3]}, w2 = {b[4], b[5]}, w3 = {b[6], b[7]};
        const u32x4g w = {__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w1), __builtin_bit_cast(unsigned, w2),
                          __builtin_bit_cast(unsigned, w3)};
        out[sp] = __builtin_bit_cast(bf16x8, w);
        if (sp < 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] -= (float)b[j];
        }
    }
}
// A fragment of the 16-k step s2 for the 32 genes of the strip: element j of lane (c, h) = img[16 * s2 + 8 * h + j][c]
__device__ __forceinline__ bf16x8 frag_trg(const __bf16* img, int s2, int lane) {
    const int i = lane & 15, grp = lane >> 4;
    const int hh = grp >> 1, colhalf = grp & 1;
    const __bf16* p0 = img + (16 * s2 + 8 * hh + (i >> 2)) * GLD + 16 * colhalf + 4 * (i & 3);
    typedef __attribute__((address_space(3))) s16x4g lds_s16x4;
    const s16x4g a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4g b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * GLD));
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
// grid: (strips, row blocks of 256); nrm2p [strips][B]
__global__ __launch_bounds__(TPB) void gp_grad3_k(const float* __restrict__ g1, const float* __restrict__ W1, long ldw, float* __restrict__ grad,
                                                   float* __restrict__ nrm2p, int B, int H, int G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    __bf16* const Ws = reinterpret_cast<__bf16*>(gsm);          // [3 parts][H][GLD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int n0 = (int)blockIdx.x * GS;
    const int rb0 = (int)blockIdx.y * 256 + wave * 64;          // this wave's 64 rows: two 32-row tiles
    const int PART = H * GLD;
    // ---- the strip of W1x: H rows x 32 genes fp32, 8 float4 pieces per row; columns past G clamped for the load, zeroed in LDS
    constexpr int SU = 8;                                       // pieces per thread in flight: the whole strip at H = 256 in ONE round trip
    for (int f0 = 0; f0 < H * 8; f0 += SU * TPB) {              // (uniform trip count)
        f32x4 v[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int f = min(f0 + tid + TPB * u, H * 8 - 1), row = f >> 3, pc = f & 7;
            const int gc = min(n0 + 4 * pc, G - 4);
            v[u] = *reinterpret_cast<const f32x4*>(W1 + (long)row * ldw + gc);
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int f = f0 + tid + TPB * u;
            if (f < H * 8) {
                const int row = f >> 3, pc = f & 7;
                f32x4 r = n0 + 4 * pc < G ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};       // (G % 4 == 0: a piece is wholly inside or outside)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    const __bf16 b0 = (__bf16)r[0], b1 = (__bf16)r[1], b2 = (__bf16)r[2], b3 = (__bf16)r[3];
                    const bf16x2g lo = {b0, b1}, hi = {b2, b3};
                    typedef unsigned u32x2g __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u32x2g*>(Ws + sp * PART + row * GLD + 4 * pc) = u32x2g{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
                    r = f32x4{r[0] - (float)b0, r[1] - (float)b1, r[2] - (float)b2, r[3] - (float)b3};
                }
            }
        }
    }
    __syncthreads();
    if (rb0 >= B) return;                                       // (after the only barrier)
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int ks = H / 16;
    const float* gr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) gr[t] = g1 + (long)min(rb0 + 32 * t + c, B - 1) * H + 8 * h;
    // g1 fragments one k-step ahead of their use
    f32x4 gn[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { gn[t][0] = *reinterpret_cast<const f32x4*>(gr[t]); gn[t][1] = *reinterpret_cast<const f32x4*>(gr[t] + 4); }
    for (int s = 0; s < ks; ++s) {
        float gv[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) { gv[t][j] = gn[t][0][j]; gv[t][4 + j] = gn[t][1][j]; }
        if (s + 1 < ks) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                gn[t][0] = *reinterpret_cast<const f32x4*>(gr[t] + 16 * (s + 1));
                gn[t][1] = *reinterpret_cast<const f32x4*>(gr[t] + 16 * (s + 1) + 4);
            }
        }
        bf16x8 wf[3];
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) wf[sp] = frag_trg(Ws + sp * PART, s, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 gf[3];
            split8g(gv[t], gf);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2], gf[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], gf[2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], gf[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], gf[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], gf[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], gf[0], acc[t], 0, 0, 0);
        }
    }
    // ---- epilogue: register i of lane (c, h) = grad[row rb0 + 32 t + c][gene n0 + (i & 3) + 8 (i >> 2) + 4 h]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = rb0 + 32 * t + c;
        float ss = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int gene = n0 + 8 * g + 4 * h;
            f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
            if (gene >= G) v = f32x4{0.f, 0.f, 0.f, 0.f};       // (whole quads: G % 4 == 0; the zeroed W columns give 0 anyway)
            if (row < B && gene < G) *reinterpret_cast<f32x4*>(grad + (long)row * G + gene) = v;
#pragma unroll
            for (int j = 0; j < 4; ++j) ss = __builtin_fmaf(v[j], v[j], ss);
        }
        ss += __shfl_xor(ss, 32, 64);
        if (h == 0 && row < B) nrm2p[(long)blockIdx.x * B + row] = ss;
    }
}

// ---- coef, loss, g1s = coef * g1 : one wave per row -----------------------------------------------------------------
// nparts > 1: nrm2 holds per-strip partial sums [nparts][B] (gp_grad3_k), added here in one fixed order
__global__ __launch_bounds__(TPB) void gp_coef_k(const float* __restrict__ nrm2, const float* __restrict__ g1, float* __restrict__ coef,
                                                  float* __restrict__ g1s, float* __restrict__ loss, int B, int H, float w, int nparts,
                                                  float* __restrict__ nrm2_total) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float n2 = 0.f;
    for (int q = lane; q < nparts; q += 64) n2 += nrm2[(long)q * B + row];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o, 64);
    if (nrm2_total && lane == 0) nrm2_total[row] = n2;
    const float n = sqrtf(n2);
    const float d = n - 1.f;
    const float cf = n > 0.f ? w * (2.f / B) * d / n : 0.f;
    if (lane == 0) {
        coef[row] = cf;
        atomicAdd(loss, d * d / B);
    }
    if (g1s)
        for (int k = lane; k < H; k += 64) g1s[(long)row * H + k] = cf * g1[(long)row * H + k];
}

// ---- du, dW2, dw3 : 8 rows h of Q per workgroup, thread = column k, the batch split over grid.y (32 rows each) ---------
constexpr int TB = 32;
__global__ __launch_bounds__(TPB) void gp_tail_k(const float* __restrict__ dg1pre, const float* __restrict__ coef,
                                                  const float* __restrict__ a1, const float* __restrict__ a2,
                                                  const float* __restrict__ w3, const float* __restrict__ W2,
                                                  float* __restrict__ dW2, float* __restrict__ dw3, int B, int H, float slope) {
    __shared__ float m2s[TB][8];
    __shared__ float cfs[TB];
    __shared__ float red[8][TPB / 64];
    const int h0 = blockIdx.x * 8, b0 = blockIdx.y * TB;
    const int nb = min(TB, B - b0);
    for (int i = threadIdx.x; i < TB * 8; i += TPB) {
        const int b = b0 + (i >> 3), h = h0 + (i & 7);
        m2s[i >> 3][i & 7] = (b < B && h < H) ? actd(a2[(long)b * H + h], slope) : 0.f;
    }
    if ((int)threadIdx.x < nb) cfs[threadIdx.x] = coef[b0 + threadIdx.x];
    __syncthreads();
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // this thread's share of sum_k W2[h,k] Q[h,k]
    for (int k = threadIdx.x; k < H; k += TPB) {
        float Q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = 0;
        for (; b + 4 <= nb; b += 4) {          // four rows' loads in flight
            float av[4], dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long o = (long)(b0 + b + u) * H + k;
                av[u] = a1[o];
                dv[u] = dg1pre[o];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float du = actd(av[u], slope) * cfs[b + u] * dv[u];
#pragma unroll
                for (int i = 0; i < 8; ++i) Q[i] += m2s[b + u][i] * du;
            }
        }
        for (; b < nb; ++b) {
            const long o = (long)(b0 + b) * H + k;
            const float du = actd(a1[o], slope) * cfs[b] * dg1pre[o];
#pragma unroll
            for (int i = 0; i < 8; ++i) Q[i] += m2s[b][i] * du;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int h = h0 + i;
            if (h < H) {
                atomicAdd(dW2 + (long)h * H + k, w3[h] * Q[i]);
                part[i] += W2[(long)h * H + k] * Q[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = part[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8 && h0 + threadIdx.x < H) {
        float v = 0.f;
        for (int w = 0; w < TPB / 64; ++w) v += red[threadIdx.x][w];
        atomicAdd(dw3 + h0 + threadIdx.x, v);
    }
}

}  // namespace

void gp_time_next(hipEvent_t begin, hipEvent_t end) { g_ev0 = begin; g_ev1 = end; }

#define GP_LAUNCH_CHECK()                         \
    do {                                          \
        GG_CHECK_HIP(hipGetLastError());          \
        return 0;                                 \
    } while (0)

int k_gp_front(const float* a1, const float* a2, const float* w3, const float* W2, float* g1, float* dg1pre, float* nrm2, int B,
               int H, float slope, hipStream_t st) {
    GG_REQUIRE((size_t)4 * H * sizeof(float) <= 48 * 1024, "hidden_dims too large for the gradient-penalty front kernel");
    launch(gp_front_k, dim3((B + 3) / 4), 4 * H * sizeof(float), st, a1, a2, w3, W2, g1, dg1pre, nrm2, B, H, slope);
    GP_LAUNCH_CHECK();
}
// split-operand strip kernel: nrm2p [gp_grad3_parts(G)][B] partial sums of squares per strip
int gp_grad3_parts(int G) { return (G + GS - 1) / GS; }
bool gp_grad3_ok(const float* g1, const float* W1, long ldw, const float* grad, int B, int H, int G) {
    static const bool off = getenv("GG_GP_GRAD_V2") != nullptr;
    if (off || H % 16 || H < 16 || H > 512 || G % 4 || G < 4 || ldw % 4 || B < 1) return false;
    return ((reinterpret_cast<uintptr_t>(g1) | reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0;
}
int k_gp_grad3(const float* g1, const float* W1, long ldw, float* grad, float* nrm2p, int B, int H, int G, hipStream_t st) {
    GG_REQUIRE(gp_grad3_ok(g1, W1, ldw, grad, B, H, G), "k_gp_grad3: unsupported shape / alignment");
    const size_t smem = (size_t)3 * H * GLD * 2;
    static size_t smem_set = 0;
    if (smem > smem_set) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gp_grad3_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        smem_set = smem;
    }
    launch(gp_grad3_k, dim3((unsigned)gp_grad3_parts(G), (unsigned)((B + 255) / 256)), smem, st, g1, W1, ldw, grad, nrm2p, B, H, G);
    GP_LAUNCH_CHECK();
}
int k_gp_grad(const float* g1, const float* W1, long ldw, float* grad, float* nrm2, int B, int H, int G, hipStream_t st) {
    const long tiles = (long)((B + GT - 1) / GT) * ((G + GT - 1) / GT);
    const bool vec = H % 4 == 0 && G % 4 == 0 && ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(g1) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(W1) & 15) == 0;
    if (vec) launch(gp_grad_k<true>, dim3((unsigned)tiles), 0, st, g1, W1, ldw, grad, nrm2, B, H, G);
    else launch(gp_grad_k<false>, dim3((unsigned)tiles), 0, st, g1, W1, ldw, grad, nrm2, B, H, G);
    GP_LAUNCH_CHECK();
}
int k_gp_coef_scale(const float* nrm2, const float* g1, float* coef, float* g1s, float* loss, int B, int H, float gp_weight,
                    hipStream_t st, int nparts, float* nrm2_total) {
    launch(gp_coef_k, dim3((B + 3) / 4), 0, st, nrm2, g1, coef, g1s, loss, B, H, gp_weight, nparts, nrm2_total != nrm2 ? nrm2_total : (float*)nullptr);      // the row totals always land in nrm2_total (one strip: a copy)
    GP_LAUNCH_CHECK();
}
int k_gp_tail(const float* dg1pre, const float* coef, const float* a1, const float* a2, const float* w3, const float* W2, float* dW2,
              float* dw3, int B, int H, float slope, hipStream_t st) {
    launch(gp_tail_k, dim3((H + 7) / 8, (B + TB - 1) / TB), 0, st, dg1pre, coef, a1, a2, w3, W2, dW2, dw3, B, H, slope);
    GP_LAUNCH_CHECK();
}

}  // namespace gg
