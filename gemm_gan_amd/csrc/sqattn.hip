// Single-query multi-head attention over a long key/value sequence WITHOUT projecting the keys and values
// (R:218-219: patch2text_attention(query = text CLS embedding, key = value = encoder output)).
//
// With ONE query per sample the K / V projections of torch's MultiheadAttention fold into the query side:
//     score[h,s] = scale * q_h . (Wk_h x_s + bk_h) = scale * (Wk_h^T q_h) . x_s + const(h)      (const drops out of softmax)
//     ctx_h      = sum_s p[h,s] (Wv_h x_s + bv_h) = Wv_h (sum_s p[h,s] x_s) + bv_h               (sum_s p = 1)
// so instead of a [N*S, E] x [E, 2E] projection (and its two backward GEMMs) the kernel streams the raw
// encoder output x [S, E] of its sample twice (second pass from L2) and does 2*nh dot products per row:
// S*E*2*nh MACs instead of S*E*2*E - a 32x reduction at E = 256, nh = 4 - and the op becomes a pure
// HBM stream of x (forward) / x + dx (backward).  Exact same function as the reference, fp32 throughout.
//
// One workgroup per sample n.  Notation: qt_h = Wk_h^T q_h [E], xbar_h = sum_s p[h,s] x_s [E].
#include "kernels.h"

namespace gg {

namespace {
constexpr int TPB = 256;
constexpr int MAXS = 2048;      // capacity; LDS score arrays are strided by the padded actual length
constexpr int MAXH = 8;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// out[h][e] = sum_d vec[h*dh + d] * W[(row0 + h*dh + d) * E + e]      (W^T applied per head)   -> LDS [nh][E]
__device__ __forceinline__ void headwise_WT_vec(float* out, const float* __restrict__ W, int row0, const float* vec, int E, int nh,
                                                int tid) {
    const int dh = E / nh;
    for (int i = tid; i < nh * E; i += TPB) {
        const int h = i / E, e = i % E;
        float acc = 0.f;
        const float* wp = W + (long)(row0 + h * dh) * E + e;
        int d = 0;
        for (; d + 8 <= dh; d += 8) {            // 8 independent loads in flight (rows are E floats apart, L2 resident)
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = wp[(long)(d + u) * E];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += vec[h * dh + d + u] * t[u];
        }
        for (; d < dh; ++d) acc += vec[h * dh + d] * wp[(long)d * E];
        out[i] = acc;
    }
}
// out[h*dh + d] = sum_e W[(row0 + h*dh + d) * E + e] * in[h][e]  (+ bias)                      (W applied per head)
__device__ __forceinline__ void headwise_W_vec(float* out, const float* __restrict__ W, int row0, const float* in, const float* bias,
                                               int E, int nh, int tid) {
    const int dh = E / nh;
    const int wave = tid >> 6, lane = tid & 63;
    for (int j = wave; j < E; j += TPB / 64) {      // one wave per output feature: coalesced read of a weight row
        const int h = j / dh;
        float acc = 0.f;
        for (int e = lane; e < E; e += 64) acc += W[(long)(row0 + j) * E + e] * in[h * E + e];
        acc = wave_sum(acc);
        if (lane == 0) out[j] = acc + (bias ? bias[j] : 0.f);
    }
}

// scores / generic "dot every row of x with nh vectors":  sc[h][s] = alpha * v_h . x_s
__device__ __forceinline__ void rows_dot(float* sc, int SS, const float* __restrict__ x, const float* vecs /*LDS [nh][E]*/, int S,
                                         int E, int nh, float alpha, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int RB = 4;                                    // rows per batch per wave
    for (int s0 = wave * RB; s0 < S; s0 += (TPB / 64) * RB) {
        float acc[RB][MAXH];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int h = 0; h < MAXH; ++h) acc[r][h] = 0.f;
        for (int e = lane * 4; e < E; e += 256) {
            f32x4 xv[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) xv[r] = *reinterpret_cast<const f32x4*>(x + (long)min(s0 + r, S - 1) * E + e);
#pragma unroll
            for (int h = 0; h < MAXH; ++h)
                if (h < nh) {
                    const f32x4 qv = *reinterpret_cast<const f32x4*>(vecs + h * E + e);
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r][h] += xv[r][0] * qv[0] + xv[r][1] * qv[1] + xv[r][2] * qv[2] + xv[r][3] * qv[3];
                }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int h = 0; h < MAXH; ++h)
                if (h < nh) {
                    const float v = wave_sum(acc[r][h]);
                    if (lane == 0 && s0 + r < S) sc[h * SS + s0 + r] = v * alpha;
                }
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward:  q [N,E] (projected query incl. bias), x [N,S,E], Win [3E,E] / bin [3E] (packed in-proj), mask
//           -> probs [N,nh,S], xbar [N,nh,E] (saved for backward), ctx [N,E]
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void sqx_fwd_kernel(const float* __restrict__ q, const float* __restrict__ x,
                                                       const float* __restrict__ Win, const float* __restrict__ bin,
                                                       const uint8_t* __restrict__ mask, int mask_B, float* __restrict__ probs,
                                                       float* __restrict__ xbar, float* __restrict__ ctx, int S, int E, int nh) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qs = sm;                 // [E]
    float* qt = qs + E;             // [nh][E]
    float* xb = qt + nh * E;        // [nh][E]
    float* sc = xb + nh * E;        // [nh][SS]
    const int SS = (S + 3) & ~3;
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* xn = x + (long)n * S * E;
    const float scale = rsqrtf((float)(E / nh));
    for (int i = tid; i < E; i += TPB) qs[i] = q[(long)n * E + i];
    __syncthreads();
    headwise_WT_vec(qt, Win, E, qs, E, nh, tid);          // Wk = rows E..2E of in_proj
    __syncthreads();
    rows_dot(sc, SS, xn, qt, S, E, nh, scale, tid);
    __syncthreads();
    for (int h = wave; h < nh; h += TPB / 64) {           // masked softmax over the keys, one wave per head
        float m = -INFINITY;
        for (int s = lane; s < S; s += 64) {
            float v = sc[h * SS + s];
            if (mask && mask[(long)(n % mask_B) * S + s]) v = -INFINITY;
            sc[h * SS + s] = v;
            m = fmaxf(m, v);
        }
        m = wave_max(m);
        float sum = 0.f;
        for (int s = lane; s < S; s += 64) {
            const float e = __expf(sc[h * SS + s] - m);
            sc[h * SS + s] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int s = lane; s < S; s += 64) {
            const float p = sc[h * SS + s] * inv;
            sc[h * SS + s] = p;
            probs[((long)n * nh + h) * S + s] = p;
        }
    }
    __syncthreads();
    for (int e = tid; e < E; e += TPB) {                 // xbar_h = sum_s p[h,s] x_s : thread owns a feature column,
        float acc[MAXH];                                 // x re-read once from L2, coalesced over e
#pragma unroll
        for (int h = 0; h < MAXH; ++h) acc[h] = 0.f;
        int s = 0;
        for (; s + 8 <= S; s += 8) {
            float xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = xn[(long)(s + u) * E + e];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int h = 0; h < MAXH; ++h)
                    if (h < nh) acc[h] += sc[h * SS + s + u] * xv[u];
        }
        for (; s < S; ++s) {
            const float xv = xn[(long)s * E + e];
#pragma unroll
            for (int h = 0; h < MAXH; ++h)
                if (h < nh) acc[h] += sc[h * SS + s] * xv;
        }
#pragma unroll
        for (int h = 0; h < MAXH; ++h)
            if (h < nh) {
                xb[h * E + e] = acc[h];
                xbar[((long)n * nh + h) * E + e] = acc[h];
            }
    }
    __syncthreads();
    headwise_W_vec(ctx + (long)n * E, Win, 2 * E, xb, bin + 2 * E, E, nh, tid);     // Wv = rows 2E..3E
}

// ---------------------------------------------------------------------------------------------------------
// backward: dctx [N,E] -> dx [N,S,E] (overwritten), dq [N,E] (grad of the projected query), and the per-sample
//           factors of the weight gradients: dqt [N,nh,E] (dWk_h += q_h (x) dqt_h), xbar saved (dWv_h += dctx_h (x) xbar_h)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void sqx_bwd_kernel(const float* __restrict__ dctx, const float* __restrict__ q,
                                                       const float* __restrict__ x, const float* __restrict__ Win,
                                                       const float* __restrict__ probs, float* __restrict__ dx,
                                                       float* __restrict__ dq, float* __restrict__ dqt_out, int S, int E, int nh) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qs = sm;                 // [E]   projected query
    float* dc = qs + E;             // [E]   dctx
    float* qt = dc + E;             // [nh][E]
    float* dxb = qt + nh * E;       // [nh][E]  d(xbar)
    float* dqt = dxb + nh * E;      // [nh][E]
    const int SS = (S + 3) & ~3;
    float* pr = dqt + nh * E;       // [nh][SS] probabilities
    float* ds = pr + nh * SS;       // [nh][SS] d(score) (already times scale)
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* xn = x + (long)n * S * E;
    float* dxn = dx + (long)n * S * E;
    const float scale = rsqrtf((float)(E / nh));
    for (int i = tid; i < E; i += TPB) {
        qs[i] = q[(long)n * E + i];
        dc[i] = dctx[(long)n * E + i];
    }
    for (int i = tid; i < nh * S; i += TPB) pr[(i / S) * SS + (i % S)] = probs[(long)n * nh * S + i];
    __syncthreads();
    headwise_WT_vec(qt, Win, E, qs, E, nh, tid);           // qt_h  = Wk_h^T q_h
    headwise_WT_vec(dxb, Win, 2 * E, dc, E, nh, tid);      // dxbar_h = Wv_h^T dctx_h
    __syncthreads();
    rows_dot(ds, SS, xn, dxb, S, E, nh, 1.f, tid);             // dp[h,s] = dxbar_h . x_s
    __syncthreads();
    for (int h = wave; h < nh; h += TPB / 64) {            // softmax backward (masked keys have p = 0)
        float dot = 0.f;
        for (int s = lane; s < S; s += 64) dot += ds[h * SS + s] * pr[h * SS + s];
        dot = wave_sum(dot);
        for (int s = lane; s < S; s += 64) ds[h * SS + s] = pr[h * SS + s] * (ds[h * SS + s] - dot) * scale;
    }
    __syncthreads();
    // dx_s = sum_h p[h,s] dxbar_h + ds[h,s] qt_h ;  dqt_h = sum_s ds[h,s] x_s      (thread owns feature columns)
    for (int e = tid; e < E; e += TPB) {
        float a_dxb[MAXH], a_qt[MAXH], a_dqt[MAXH];
#pragma unroll
        for (int h = 0; h < MAXH; ++h) {
            a_dxb[h] = h < nh ? dxb[h * E + e] : 0.f;
            a_qt[h] = h < nh ? qt[h * E + e] : 0.f;
            a_dqt[h] = 0.f;
        }
        for (int s0 = 0; s0 < S; s0 += 8) {
            float xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = xn[(long)min(s0 + u, S - 1) * E + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + u;
                if (s < S) {
                    float o = 0.f;
#pragma unroll
                    for (int h = 0; h < MAXH; ++h)
                        if (h < nh) {
                            const float g = ds[h * SS + s];
                            o += pr[h * SS + s] * a_dxb[h] + g * a_qt[h];
                            a_dqt[h] += g * xv[u];
                        }
                    dxn[(long)s * E + e] = o;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < MAXH; ++h)
            if (h < nh) {
                dqt[h * E + e] = a_dqt[h];
                dqt_out[((long)n * nh + h) * E + e] = a_dqt[h];
            }
    }
    __syncthreads();
    headwise_W_vec(dq + (long)n * E, Win, E, dqt, nullptr, E, nh, tid);     // dq_h = Wk_h dqt_h
}


// =========================================================================================================
// Streaming kernels (production shapes): the per-head projections are hoisted out as small batched GEMMs
// (qt = Wk_h^T q_h and ctx = Wv_h xbar_h + bv before / after the forward, dxbar = Wv_h^T dctx_h and
// dq = Wk_h dqt_h around the backward), and what remains touches the encoder output x exactly ONCE per pass:
//   forward : one sweep with an online softmax - scores, running (max, sum) and sum_s p x_s together
//   backward: one sweep - sum_s p_s dp_s equals dxbar_h . xbar_h (xbar saved by the forward), so d(score) is
//             available row by row and dx is written in the same sweep that accumulates dqt
// Work decomposition: one workgroup per sample; a wave instruction covers 4 rows x 16 lanes, a lane owns
// 16 * (E/256) .. features of one row in 16-byte pieces (each load = 4 rows x 256 contiguous bytes), row
// dot products are reduced inside the 16-lane DPP row (quad_perm / row_half_mirror / row_mirror adds).
// =========================================================================================================
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {      // every lane of a 16-lane row ends with the row's sum
    v += dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);      // row_half_mirror
    v += dpp_mov<0x140>(v);      // row_mirror
    return v;
}
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
constexpr float SQ_LOG2E = 1.4426950408889634f;

// qt [N,NH,E] -> probs [N,NH,S], xbar [N,NH,E]
template <int NV, int NH>       // NV = E / 64 16-byte pieces per lane
__global__ __launch_bounds__(TPB, 2) void sqx2_fwd_kernel(const float* __restrict__ qt, const float* __restrict__ x,
                                                        const uint8_t* __restrict__ mask, int mask_B, float* __restrict__ probs,
                                                        float* __restrict__ xbar, int S, float scale2) {
    constexpr int E = 64 * NV;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int SS = (S + 3) & ~3;
    float* sc = sm;                       // [NH][SS] scaled scores (log2 domain), -inf where masked
    float* part = sc + NH * SS;           // [4 waves][NH][E] partial sum_s p x_s (relative to the wave's max)
    float* ml = part + 4 * NH * E;        // [4 waves][NH][2] (max, sum) per wave
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, slot = lane >> 4;
    const float* xn = x + (long)n * S * E + 4 * li;
    const uint8_t* mrow = mask ? mask + (long)(n % mask_B) * S : nullptr;
    f32x4 qv[NH][NV];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int k = 0; k < NV; ++k) qv[h][k] = *reinterpret_cast<const f32x4*>(qt + ((long)n * NH + h) * E + 4 * li + 64 * k);
    float m[NH], l[NH];
    f32x4 acc[NH][NV];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        m[h] = -INFINITY; l[h] = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[h][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nit = (S + 15) / 16;
    // The next row (and its mask byte) is requested UNCONDITIONALLY from a clamped index: `if (it + 1 < nit) load` and `mrow &&
    // mrow[r]` made the wait-count pass drain the request at once (vmcnt(0)) - the row prefetch did not exist and every iteration
    // paid for a mask-byte round trip.  Without a mask the byte is read from the x rows (any valid address) and ignored.
    const bool use_mask = mrow != nullptr;
    const uint8_t* mr = use_mask ? mrow : reinterpret_cast<const uint8_t*>(x);
    f32x4 nxt[NV];
    uint8_t mnxt;
    {
        const int r0 = min(4 * wave + slot, S - 1);
#pragma unroll
        for (int k = 0; k < NV; ++k) nxt[k] = *reinterpret_cast<const f32x4*>(xn + (long)r0 * E + 64 * k);
        mnxt = mr[r0];
    }
    for (int it = 0; it < nit; ++it) {
        const int r = 16 * it + 4 * wave + slot;
        f32x4 cur[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) cur[k] = nxt[k];
        const uint8_t mcur = mnxt;
        {
            const int rn = min(r + 16, S - 1);
#pragma unroll
            for (int k = 0; k < NV; ++k) nxt[k] = *reinterpret_cast<const f32x4*>(xn + (long)rn * E + 64 * k);
            mnxt = mr[rn];
        }
        const bool valid = r < S && !(use_mask && mcur);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) d += dot4(cur[k], qv[h][k]);
            d = row16_sum(d);
            const float s2 = valid ? d * scale2 : -INFINITY;
            if (li == 0 && r < S) sc[h * SS + r] = s2;
            const float mn = fmaxf(m[h], s2);
            const float mref = (mn == -INFINITY) ? 0.f : mn;
            const float alpha = __builtin_amdgcn_exp2f(m[h] - mref);
            const float p = __builtin_amdgcn_exp2f(s2 - mref);
            l[h] = l[h] * alpha + p;
            m[h] = mn;
#pragma unroll
            for (int k = 0; k < NV; ++k) acc[h][k] = acc[h][k] * alpha + p * cur[k];
        }
    }
    // merge the 4 row slots of the wave (lanes 16 / 32 apart hold the same features)
#pragma unroll
    for (int off = 16; off <= 32; off *= 2) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float mo = __shfl_xor(m[h], off, 64), lo = __shfl_xor(l[h], off, 64);
            const float mn = fmaxf(m[h], mo);
            const float mref = (mn == -INFINITY) ? 0.f : mn;
            const float a = __builtin_amdgcn_exp2f(m[h] - mref), ao = __builtin_amdgcn_exp2f(mo - mref);
            l[h] = l[h] * a + lo * ao;
            m[h] = mn;
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][k][j] = acc[h][k][j] * a + __shfl_xor(acc[h][k][j], off, 64) * ao;
        }
    }
    if (slot == 0) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
#pragma unroll
            for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4*>(&part[(wave * NH + h) * E + 4 * li + 64 * k]) = acc[h][k];
            if (li == 0) { ml[(wave * NH + h) * 2] = m[h]; ml[(wave * NH + h) * 2 + 1] = l[h]; }
        }
    }
    __syncthreads();
    // final merge over the 4 waves: every thread needs (M_h, 1/L_h)
    float Mh[NH], inv[NH], wgt[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, ml[(w * NH + h) * 2]);
        const float mref = (M == -INFINITY) ? 0.f : M;
        float L = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            wgt[h][w] = __builtin_amdgcn_exp2f(ml[(w * NH + h) * 2] - mref);
            L += ml[(w * NH + h) * 2 + 1] * wgt[h][w];
        }
        Mh[h] = mref;
        inv[h] = L > 0.f ? 1.f / L : 0.f;
    }
    for (int e = tid; e < E; e += TPB) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += part[(w * NH + h) * E + e] * wgt[h][w];
            xbar[((long)n * NH + h) * E + e] = v * inv[h];
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
        for (int s = tid; s < S; s += TPB) probs[((long)n * NH + h) * S + s] = __builtin_amdgcn_exp2f(sc[h * SS + s] - Mh[h]) * inv[h];
}

// dxbar [N,NH,E], qt, xbar, probs -> dx [N,S,E] (overwritten), dqt [N,NH,E]
template <int NV, int NH>
__global__ __launch_bounds__(TPB, 2) void sqx2_bwd_kernel(const float* __restrict__ dxbar, const float* __restrict__ qt,
                                                        const float* __restrict__ xbar, const float* __restrict__ x,
                                                        const float* __restrict__ probs, float* __restrict__ dx,
                                                        float* __restrict__ dqt, int S, float scale) {
    constexpr int E = 64 * NV;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int SS = (S + 3) & ~3;
    float* pr = sm;                       // [NH][SS]
    float* part = pr + NH * SS;           // [4 waves][NH][E]
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, slot = lane >> 4;
    const float* xn = x + (long)n * S * E + 4 * li;
    float* dxn = dx + (long)n * S * E + 4 * li;
    for (int i = tid; i < NH * S; i += TPB) pr[(i / S) * SS + (i % S)] = probs[(long)n * NH * S + i];
    f32x4 dxb[NH][NV], qv[NH][NV], dq[NH][NV];
    float dot[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const long o = ((long)n * NH + h) * E + 4 * li + 64 * k;
            dxb[h][k] = *reinterpret_cast<const f32x4*>(dxbar + o);
            qv[h][k] = *reinterpret_cast<const f32x4*>(qt + o);
            dq[h][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            d += dot4(dxb[h][k], *reinterpret_cast<const f32x4*>(xbar + o));
        }
        dot[h] = row16_sum(d);            // sum_s p_s dp_s = dxbar_h . xbar_h
    }
    __syncthreads();
    const int nit = (S + 15) / 16;
    for (int it = 0; it < nit; ++it) {
        const int r = 16 * it + 4 * wave + slot;
        const int rc = min(r, S - 1);
        f32x4 cur[NV], out[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            cur[k] = *reinterpret_cast<const f32x4*>(xn + (long)rc * E + 64 * k);
            out[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float dp = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) dp += dot4(cur[k], dxb[h][k]);
            dp = row16_sum(dp);
            const float p = r < S ? pr[h * SS + rc] : 0.f;
            const float ds = p * (dp - dot[h]) * scale;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                out[k] += p * dxb[h][k] + ds * qv[h][k];
                dq[h][k] += ds * cur[k];
            }
        }
        if (r < S) {
#pragma unroll
            for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4*>(dxn + (long)r * E + 64 * k) = out[k];
        }
    }
#pragma unroll
    for (int off = 16; off <= 32; off *= 2)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) dq[h][k][j] += __shfl_xor(dq[h][k][j], off, 64);
    if (slot == 0) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4*>(&part[(wave * NH + h) * E + 4 * li + 64 * k]) = dq[h][k];
    }
    __syncthreads();
    for (int i = tid; i < NH * E; i += TPB)
        dqt[(long)n * NH * E + i] = part[i] + part[NH * E + i] + part[2 * NH * E + i] + part[3 * NH * E + i];
}

size_t s2_smem(int S, int E, int nh) { return sizeof(float) * ((size_t)nh * ((S + 3) & ~3) + 4 * (size_t)nh * E + 8 * (size_t)nh); }

size_t fwd_smem(int S, int E, int nh) { return sizeof(float) * ((size_t)E + 2 * (size_t)nh * E + (size_t)nh * ((S + 3) & ~3)); }
size_t bwd_smem(int S, int E, int nh) { return sizeof(float) * (2 * (size_t)E + 3 * (size_t)nh * E + 2 * (size_t)nh * ((S + 3) & ~3)); }
}  // namespace

bool sqx_supported(int S, int E, int nh) {
    return nh >= 1 && nh <= MAXH && E % nh == 0 && E % 4 == 0 && S <= MAXS && bwd_smem(S, E, nh) <= 160 * 1024;
}

int sqx_attn_fwd(const float* q, const float* x, const float* Win, const float* bin, const uint8_t* mask, int mask_B, float* probs,
                 float* xbar, float* ctx, int N, int S, int E, int nh, hipStream_t st) {
    GG_REQUIRE(sqx_supported(S, E, nh), "sqx attention: unsupported shape");
    const size_t sm = fwd_smem(S, E, nh);
    GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sqx_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    hipLaunchKernelGGL(sqx_fwd_kernel, dim3(N), dim3(TPB), sm, st, q, x, Win, bin, mask, mask_B > 0 ? mask_B : N, probs, xbar, ctx, S, E, nh);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

int sqx_attn_bwd(const float* dctx, const float* q, const float* x, const float* Win, const float* probs, float* dx, float* dq,
                 float* dqt, int N, int S, int E, int nh, hipStream_t st) {
    GG_REQUIRE(sqx_supported(S, E, nh), "sqx attention: unsupported shape");
    const size_t sm = bwd_smem(S, E, nh);
    GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sqx_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    hipLaunchKernelGGL(sqx_bwd_kernel, dim3(N), dim3(TPB), sm, st, dctx, q, x, Win, probs, dx, dq, dqt, S, E, nh);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}


// ---- streaming variant --------------------------------------------------------------------------------------
bool sqx_stream_supported(int S, int E, int nh) {
    if (!(E == 64 || E == 128 || E == 256)) return false;
    if (!(nh == 1 || nh == 2 || nh == 4)) return false;
    return S >= 1 && S <= MAXS && s2_smem(S, E, nh) <= 64 * 1024;
}
#define SQX2_DISPATCH(KERNEL, ...)                                                                  \
    do {                                                                                            \
        const int nv = E / 64;                                                                      \
        if (nv == 4 && nh == 4) hipLaunchKernelGGL((KERNEL<4, 4>), dim3(N), dim3(TPB), sm, st, __VA_ARGS__);      \
        else if (nv == 4 && nh == 2) hipLaunchKernelGGL((KERNEL<4, 2>), dim3(N), dim3(TPB), sm, st, __VA_ARGS__); \
        else if (nv == 4 && nh == 1) hipLaunchKernelGGL((KERNEL<4, 1>), dim3(N), dim3(TPB), sm, st, __VA_ARGS__); \
        else if (nv == 2 && nh == 4) hipLaunchKernelGGL((KERNEL<2, 4>), dim3(N), dim3(TPB), sm, st, __VA_ARGS__); \
        else if (nv == 2 && nh == 2) hipLaunchKernelGGL((KERNEL<2, 2>), dim3(N), dim3(TPB), sm, st, __VA_ARGS__); \
        else if (nv == 2 && nh == 1) hipLaunchKernelGGL((KERNEL<2, 1>), dim3(N), dim3(TPB), sm, st, __VA_ARGS__); \
        else if (nv == 1 && nh == 4) hipLaunchKernelGGL((KERNEL<1, 4>), dim3(N), dim3(TPB), sm, st, __VA_ARGS__); \
        else if (nv == 1 && nh == 2) hipLaunchKernelGGL((KERNEL<1, 2>), dim3(N), dim3(TPB), sm, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<1, 1>), dim3(N), dim3(TPB), sm, st, __VA_ARGS__);                        \
    } while (0)

int sqx_stream_fwd(const float* qt, const float* x, const uint8_t* mask, int mask_B, float* probs, float* xbar, int N, int S, int E,
                   int nh, hipStream_t st) {
    GG_REQUIRE(sqx_stream_supported(S, E, nh), "sqx streaming attention: unsupported shape");
    const size_t sm = s2_smem(S, E, nh);
    const float scale2 = 1.f / sqrtf((float)(E / nh)) * SQ_LOG2E;
    SQX2_DISPATCH(sqx2_fwd_kernel, qt, x, mask, mask_B > 0 ? mask_B : N, probs, xbar, S, scale2);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
int sqx_stream_bwd(const float* dxbar, const float* qt, const float* xbar, const float* x, const float* probs, float* dx, float* dqt,
                   int N, int S, int E, int nh, hipStream_t st) {
    GG_REQUIRE(sqx_stream_supported(S, E, nh), "sqx streaming attention: unsupported shape");
    const size_t sm = s2_smem(S, E, nh);
    const float scale = 1.f / sqrtf((float)(E / nh));
    SQX2_DISPATCH(sqx2_bwd_kernel, dxbar, qt, xbar, x, probs, dx, dqt, S, scale);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gg
