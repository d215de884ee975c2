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

// ---- coef, loss, g1s = coef * g1 : one wave per row -----------------------------------------------------------------
__global__ __launch_bounds__(TPB) void gp_coef_k(const float* __restrict__ nrm2, const float* __restrict__ g1, float* __restrict__ coef,
                                                  float* __restrict__ g1s, float* __restrict__ loss, int B, int H, float w) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float n = sqrtf(nrm2[row]);
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
int k_gp_grad(const float* g1, const float* W1, long ldw, float* grad, float* nrm2, int B, int H, int G, hipStream_t st) {
    const long tiles = (long)((B + GT - 1) / GT) * ((G + GT - 1) / GT);
    const bool vec = H % 4 == 0 && G % 4 == 0 && ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(g1) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(W1) & 15) == 0;
    if (vec) launch(gp_grad_k<true>, dim3((unsigned)tiles), 0, st, g1, W1, ldw, grad, nrm2, B, H, G);
    else launch(gp_grad_k<false>, dim3((unsigned)tiles), 0, st, g1, W1, ldw, grad, nrm2, B, H, G);
    GP_LAUNCH_CHECK();
}
int k_gp_coef_scale(const float* nrm2, const float* g1, float* coef, float* g1s, float* loss, int B, int H, float gp_weight,
                    hipStream_t st) {
    launch(gp_coef_k, dim3((B + 3) / 4), 0, st, nrm2, g1, coef, g1s, loss, B, H, gp_weight);
    GP_LAUNCH_CHECK();
}
int k_gp_tail(const float* dg1pre, const float* coef, const float* a1, const float* a2, const float* w3, const float* W2, float* dW2,
              float* dw3, int B, int H, float slope, hipStream_t st) {
    launch(gp_tail_k, dim3((H + 7) / 8, (B + TB - 1) / TB), 0, st, dg1pre, coef, a1, a2, w3, W2, dW2, dw3, B, H, slope);
    GP_LAUNCH_CHECK();
}

}  // namespace gg
