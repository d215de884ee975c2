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
}  // namespace gg

extern "C" {
long gg_eval_nn2_scratch(long nq, long nr) { return nq > 0 && nr > 0 ? 2L * gg::nn2_splits(nq, nr) * nq : -1; }
int gg_eval_nn2(const float* queries, long nq, const float* refs, long nr, int dim, float* d1, float* d2, float* scratch,
                long scratch_floats, void* stream) {
    return gg::nn2(queries, nq, refs, nr, dim, d1, d2, scratch, scratch_floats, reinterpret_cast<hipStream_t>(stream));
}
}
