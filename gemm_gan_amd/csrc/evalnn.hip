// Two nearest neighbours of every query row in a reference set, Euclidean, exact fp32 differences: the device-side math of
// the reference's privacy metrics DCR / NNDR (src/privacy_evaluator.py:9-66), which materialises a [128, N, G] difference
// tensor per batch and sorts every row.  Here: 64 x 64 (query x reference) tiles, the gene dimension streamed through LDS
// 16 columns at a time, 4 x 4 pairs per thread accumulating (q - r)^2 on the VALU (no |q|^2 + |r|^2 - 2qr: the metric is
// about near-duplicates, where that form cancels), a running (smallest, second smallest) per query; the reference set is
// split over grid.y and the partial pairs are merged by a second launch.  Independent of the training engine.
#include "gg_common.h"

namespace gg {
namespace {
constexpr int TQ = 64, TR = 64, TKK = 16, NTH = 256;

__device__ __forceinline__ void top2_push(float& m1, float& m2, float d) {
    if (d < m1) { m2 = m1; m1 = d; }
    else if (d < m2) m2 = d;
}

__global__ __launch_bounds__(NTH) void nn2_partial_kernel(const float* __restrict__ Q, long nq, const float* __restrict__ R, long nr,
                                                           int dim, float* __restrict__ part, int splits) {
    __shared__ __attribute__((aligned(16))) float Qs[TKK][TQ + 4];
    __shared__ __attribute__((aligned(16))) float Rs[TKK][TR + 4];
    __shared__ float red[TQ][16][2];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const long q0 = (long)blockIdx.x * TQ;
    const long per = ((nr + splits - 1) / splits + TR - 1) / TR * TR;          // reference rows per split (whole tiles)
    const long r_lo = (long)blockIdx.y * per, r_hi = min(nr, r_lo + per);
    float m1[4], m2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) m1[i] = m2[i] = INFINITY;
    const int lrow = tid >> 2, lk = (tid & 3) * 4;                               // staging: row of the tile, 4 consecutive columns
    for (long rt = r_lo; rt < r_hi; rt += TR) {
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int k0 = 0; k0 < dim; k0 += TKK) {
            const long qr = min(q0 + lrow, nq - 1), rr = min(rt + lrow, nr - 1);   // clamped rows are masked at the end
            float qv[4], rv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + lk + j;
                qv[j] = k < dim ? Q[qr * dim + k] : 0.f;
                rv[j] = k < dim ? R[rr * dim + k] : 0.f;
            }
            __syncthreads();                                                     // the previous chunk's readers are done
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Qs[lk + j][lrow] = qv[j];
                Rs[lk + j][lrow] = rv[j];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < TKK; ++k) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&Qs[k][4 * ty]);
                const f32x4 b = *reinterpret_cast<const f32x4*>(&Rs[k][4 * tx]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = a[i] - b[j];
                        acc[i][j] += d * d;
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (rt + 4 * tx + j >= r_hi) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) top2_push(m1[i], m2[i], acc[i][j]);
        }
    }
    // merge the 16 column threads of every query row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[4 * ty + i][tx][0] = m1[i];
        red[4 * ty + i][tx][1] = m2[i];
    }
    __syncthreads();
    if (tid < TQ && q0 + tid < nq) {
        float a = INFINITY, b = INFINITY;
        for (int t = 0; t < 16; ++t) {
            top2_push(a, b, red[tid][t][0]);
            top2_push(a, b, red[tid][t][1]);
        }
        float* o = part + ((long)blockIdx.y * nq + q0 + tid) * 2;
        o[0] = a;
        o[1] = b;
    }
}

__global__ void nn2_merge_kernel(const float* __restrict__ part, long nq, int splits, float* __restrict__ d1, float* __restrict__ d2) {
    const long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float a = INFINITY, b = INFINITY;
    for (int s = 0; s < splits; ++s) {
        top2_push(a, b, part[((long)s * nq + q) * 2]);
        top2_push(a, b, part[((long)s * nq + q) * 2 + 1]);
    }
    d1[q] = sqrtf(a);
    d2[q] = sqrtf(b);          // +inf when the reference set has a single row
}

// ---- k smallest distances per query (k-th-neighbour radii of PRDC, src/distribution_distances.py:87-99) and the PRDC
// ---- counting pass (src/distribution_distances.py:102-142).  The reference uses L1 distances there (metric='l1', :64).
template <int K>
__device__ __forceinline__ void topk_push(float (&m)[K], float d) {
    if (d >= m[K - 1]) return;
    m[K - 1] = d;
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
        const float a = m[j - 1], b = m[j];
        m[j - 1] = fminf(a, b);
        m[j] = fmaxf(a, b);
    }
}
template <bool L1>
__device__ __forceinline__ void tile_distances(const float* __restrict__ Q, long nq, long q0, const float* __restrict__ R, long nr, long rt,
                                               int dim, float (*Qs)[TQ + 4], float (*Rs)[TR + 4], float (&acc)[4][4], int tid) {
    const int ty = tid >> 4, tx = tid & 15;
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < dim; k0 += TKK) {
        const long qr = min(q0 + lrow, nq - 1), rr = min(rt + lrow, nr - 1);
        float qv[4], rv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + lk + j;
            qv[j] = k < dim ? Q[qr * dim + k] : 0.f;
            rv[j] = k < dim ? R[rr * dim + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Qs[lk + j][lrow] = qv[j];
            Rs[lk + j][lrow] = rv[j];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TKK; ++k) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&Qs[k][4 * ty]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&Rs[k][4 * tx]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = a[i] - b[j];
                    if (L1) acc[i][j] += fabsf(d);
                    else acc[i][j] += d * d;
                }
        }
    }
}

template <int K, bool L1>
__global__ __launch_bounds__(NTH) void topk_partial_kernel(const float* __restrict__ Q, long nq, const float* __restrict__ R, long nr,
                                                            int dim, float* __restrict__ part, int splits) {
    __shared__ __attribute__((aligned(16))) float Qs[TKK][TQ + 4];
    __shared__ __attribute__((aligned(16))) float Rs[TKK][TR + 4];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const long q0 = (long)blockIdx.x * TQ;
    const long per = ((nr + splits - 1) / splits + TR - 1) / TR * TR;
    const long r_lo = (long)blockIdx.y * per, r_hi = min(nr, r_lo + per);
    float m[4][K];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) m[i][j] = INFINITY;
    for (long rt = r_lo; rt < r_hi; rt += TR) {
        float acc[4][4];
        tile_distances<L1>(Q, nq, q0, R, nr, rt, dim, Qs, Rs, acc, tid);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (rt + 4 * tx + j >= r_hi) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) topk_push<K>(m[i], acc[i][j]);
        }
    }
    // merge the 16 column threads of a query row (consecutive lanes) by exchanging sorted lists
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float other[K];
#pragma unroll
            for (int j = 0; j < K; ++j) other[j] = __shfl_xor(m[i][j], off, 64);
#pragma unroll
            for (int j = 0; j < K; ++j) topk_push<K>(m[i], other[j]);
        }
    }
    if (tx == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long q = q0 + 4 * ty + i;
            if (q >= nq) continue;
#pragma unroll
            for (int j = 0; j < K; ++j) part[((long)blockIdx.y * nq + q) * K + j] = m[i][j];
        }
    }
}
template <int K, bool L1>
__global__ void topk_merge_kernel(const float* __restrict__ part, long nq, int splits, float* __restrict__ out) {
    const long q = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float m[K];
#pragma unroll
    for (int j = 0; j < K; ++j) m[j] = INFINITY;
    for (int s = 0; s < splits; ++s)
        for (int j = 0; j < K; ++j) topk_push<K>(m, part[((long)s * nq + q) * K + j]);
#pragma unroll
    for (int j = 0; j < K; ++j) out[q * K + j] = L1 ? m[j] : sqrtf(m[j]);
}

// counts of one [real x fake] distance matrix against the two radius vectors, never materialised:
//   below_real[j] = #{i : d_ij < rad_real[i]}   (precision: > 0; density: mean / k)
//   any_fake[i]   = any_j d_ij < rad_fake[j]     (recall)
//   min_d[i]      = min_j d_ij                   (coverage: < rad_real[i]); stored as the float's bit pattern (d >= 0)
template <bool L1, bool INC>
__global__ __launch_bounds__(NTH) void prdc_kernel(const float* __restrict__ real, long nr, const float* __restrict__ fake, long nf, int dim,
                                                    const float* __restrict__ rad_real, const float* __restrict__ rad_fake,
                                                    int* __restrict__ below_real, int* __restrict__ any_fake, unsigned* __restrict__ min_d) {
    __shared__ __attribute__((aligned(16))) float Qs[TKK][TQ + 4];
    __shared__ __attribute__((aligned(16))) float Rs[TKK][TR + 4];
    __shared__ int colcnt[TR];
    __shared__ int rowany[TQ];
    __shared__ unsigned rowmin[TQ];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const long i0 = (long)blockIdx.x * TQ, j0 = (long)blockIdx.y * TR;
    if (tid < TR) { colcnt[tid] = 0; rowany[tid] = 0; rowmin[tid] = 0x7f800000u; }
    float acc[4][4];
    tile_distances<L1>(real, nr, i0, fake, nf, j0, dim, Qs, Rs, acc, tid);          // its barriers also publish the zeroing above
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long ri = i0 + 4 * ty + i;
        if (ri >= nr) continue;
        const float rr = rad_real[ri];
        int any = 0;
        float mn = INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long fj = j0 + 4 * tx + j;
            if (fj >= nf) continue;
            const float d = L1 ? acc[i][j] : sqrtf(acc[i][j]);
            if (INC ? d <= rr : d < rr) atomicAdd(&colcnt[4 * tx + j], 1);
            any |= INC ? d <= rad_fake[fj] : d < rad_fake[fj];
            mn = fminf(mn, d);
        }
        if (any) atomicOr(&rowany[4 * ty + i], 1);
        atomicMin(&rowmin[4 * ty + i], __float_as_uint(mn));
    }
    __syncthreads();
    if (tid < TR) {
        if (j0 + tid < nf && colcnt[tid]) atomicAdd(&below_real[j0 + tid], colcnt[tid]);
        if (i0 + tid < nr) {
            if (rowany[tid]) atomicOr(&any_fake[i0 + tid], 1);
            atomicMin(&min_d[i0 + tid], rowmin[tid]);
        }
    }
}
}  // namespace

int nn2_splits(long nq, long nr) {
    const long qt = (nq + TQ - 1) / TQ, rt = (nr + TR - 1) / TR;
    long s = (1024 + qt - 1) / qt;            // aim at >= 1024 workgroups (256 CUs, several per CU)
    if (s > rt) s = rt;
    if (s < 1) s = 1;
    return (int)s;
}
int nn2(const float* Q, long nq, const float* R, long nr, int dim, float* d1, float* d2, float* scratch, long scratch_floats, hipStream_t st) {
    GG_REQUIRE(Q && R && d1 && d2 && scratch && nq > 0 && nr > 0 && dim > 0, "nn2: bad argument");
    const int splits = nn2_splits(nq, nr);
    GG_REQUIRE(scratch_floats >= 2L * splits * nq, "nn2: scratch too small (gg_eval_nn2_scratch)");
    nn2_partial_kernel<<<dim3((unsigned)((nq + TQ - 1) / TQ), (unsigned)splits), NTH, 0, st>>>(Q, nq, R, nr, dim, scratch, splits);
    nn2_merge_kernel<<<(unsigned)((nq + 255) / 256), 256, 0, st>>>(scratch, nq, splits, d1, d2);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int K>
int topk_launch(const float* Q, long nq, const float* R, long nr, int dim, int l1, float* out, float* scratch, int splits, hipStream_t st) {
    const dim3 grid((unsigned)((nq + TQ - 1) / TQ), (unsigned)splits);
    const unsigned mb = (unsigned)((nq + 255) / 256);
    if (l1) {
        topk_partial_kernel<K, true><<<grid, NTH, 0, st>>>(Q, nq, R, nr, dim, scratch, splits);
        topk_merge_kernel<K, true><<<mb, 256, 0, st>>>(scratch, nq, splits, out);
    } else {
        topk_partial_kernel<K, false><<<grid, NTH, 0, st>>>(Q, nq, R, nr, dim, scratch, splits);
        topk_merge_kernel<K, false><<<mb, 256, 0, st>>>(scratch, nq, splits, out);
    }
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
int topk_width(int k) { return k <= 4 ? 4 : (k <= 8 ? 8 : 16); }
int knn(const float* Q, long nq, const float* R, long nr, int dim, int k, int l1, float* out, float* scratch, long scratch_floats, hipStream_t st) {
    GG_REQUIRE(Q && R && out && scratch && nq > 0 && nr > 0 && dim > 0, "knn: bad argument");
    GG_REQUIRE(k >= 1 && k <= 16, "knn: 1 <= k <= 16");
    const int splits = nn2_splits(nq, nr), K = topk_width(k);
    GG_REQUIRE(scratch_floats >= (long)K * splits * nq, "knn: scratch too small (gg_eval_knn_scratch)");
    if (K == 4) return topk_launch<4>(Q, nq, R, nr, dim, l1, out, scratch, splits, st);
    if (K == 8) return topk_launch<8>(Q, nq, R, nr, dim, l1, out, scratch, splits, st);
    return topk_launch<16>(Q, nq, R, nr, dim, l1, out, scratch, splits, st);
}
int prdc_counts(const float* real, long nr, const float* fake, long nf, int dim, int l1, const float* rad_real, const float* rad_fake,
                int* below_real, int* any_fake, float* min_d, hipStream_t st) {
    GG_REQUIRE(real && fake && rad_real && rad_fake && below_real && any_fake && min_d && nr > 0 && nf > 0 && dim > 0, "prdc: bad argument");
    GG_CHECK_HIP(hipMemsetAsync(below_real, 0, sizeof(int) * nf, st));
    GG_CHECK_HIP(hipMemsetAsync(any_fake, 0, sizeof(int) * nr, st));
    GG_CHECK_HIP(hipMemsetAsync(min_d, 0x7f, sizeof(float) * nr, st));            // 0x7f7f7f7f: a huge finite float, any distance is below
    const dim3 grid((unsigned)((nr + TQ - 1) / TQ), (unsigned)((nf + TR - 1) / TR));
    // l1 bit 0: L1 distances (PRDC, src/distribution_distances.py:64), else Euclidean; bit 1: inclusive comparisons d <= radius
    // (ManifoldEstimator.evaluate, src/unsupervised_metrics.py:223) instead of d < radius (compute_prdc, :121-139)
    unsigned* md = reinterpret_cast<unsigned*>(min_d);
    switch (l1 & 3) {
        case 1: prdc_kernel<true, false><<<grid, NTH, 0, st>>>(real, nr, fake, nf, dim, rad_real, rad_fake, below_real, any_fake, md); break;
        case 3: prdc_kernel<true, true><<<grid, NTH, 0, st>>>(real, nr, fake, nf, dim, rad_real, rad_fake, below_real, any_fake, md); break;
        case 2: prdc_kernel<false, true><<<grid, NTH, 0, st>>>(real, nr, fake, nf, dim, rad_real, rad_fake, below_real, any_fake, md); break;
        default: prdc_kernel<false, false><<<grid, NTH, 0, st>>>(real, nr, fake, nf, dim, rad_real, rad_fake, below_real, any_fake, md); break;
    }
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace gg

extern "C" {
long gg_eval_nn2_scratch(long nq, long nr) { return nq > 0 && nr > 0 ? 2L * gg::nn2_splits(nq, nr) * nq : -1; }
long gg_eval_knn_scratch(long nq, long nr, int k) { return nq > 0 && nr > 0 && k >= 1 && k <= 16 ? (long)gg::topk_width(k) * gg::nn2_splits(nq, nr) * nq : -1; }
int gg_eval_knn_width(int k) { return k >= 1 && k <= 16 ? gg::topk_width(k) : -1; }
int gg_eval_knn(const float* queries, long nq, const float* refs, long nr, int dim, int k, int l1, float* out, float* scratch,
                long scratch_floats, void* stream) {
    return gg::knn(queries, nq, refs, nr, dim, k, l1, out, scratch, scratch_floats, reinterpret_cast<hipStream_t>(stream));
}
int gg_eval_prdc_counts(const float* real, long nr, const float* fake, long nf, int dim, int l1, const float* rad_real, const float* rad_fake,
                        int* below_real, int* any_fake, float* min_d, void* stream) {
    return gg::prdc_counts(real, nr, fake, nf, dim, l1, rad_real, rad_fake, below_real, any_fake, min_d, reinterpret_cast<hipStream_t>(stream));
}
int gg_eval_nn2(const float* queries, long nq, const float* refs, long nr, int dim, float* d1, float* d2, float* scratch,
                long scratch_floats, void* stream) {
    return gg::nn2(queries, nq, refs, nr, dim, d1, d2, scratch, scratch_floats, reinterpret_cast<hipStream_t>(stream));
}
}
