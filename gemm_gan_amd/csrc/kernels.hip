// Row-wise / elementwise kernels (gfx950).  Memory-bound helpers around the MFMA GEMMs: every
// kernel reads and writes contiguous rows with consecutive lanes on consecutive addresses; rows are
// owned by one 64-lane wave so all reductions are wave shuffles (no LDS round trips, no atomics
// except the cross-row parameter-gradient sums).
#include "kernels.h"
#include "drop_rng.h"

namespace gg {

namespace {
constexpr int TPB = 256;
inline unsigned nblocks(long n, int per = TPB, long cap = 1 << 20) {
    long b = (n + per - 1) / per;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

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

// ---- dropout stream: drop_rng.h ---------------------------------------------------------------
}  // namespace

DropKey make_drop_key(float p, uint64_t seed, uint32_t site, uint32_t call) {
    DropKey k;
    k.p = p;
    auto mix = [](uint64_t x) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
        return x;
    };
    uint64_t a = mix(seed ^ (0x9E3779B97F4A7C15ULL * (site + 1)));
    uint64_t b = mix(a ^ (0xD1B54A32D192ED03ULL * ((uint64_t)call + 1)));
    k.k0 = (uint32_t)b ^ (uint32_t)(b >> 32);
    const long t = lrintf(p * 65536.f);
    k.thr = (uint32_t)(t < 0 ? 0 : (t > 65535 ? 65535 : t));
    return k;
}

#define GG_LAUNCH_CHECK()             \
    GG_CHECK_HIP(hipGetLastError()); \
    return 0

// ---- trivial elementwise ------------------------------------------------------------------------
__global__ void fill_k(float* x, long n, float v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = v;
}
int k_fill(float* x, long n, float v, hipStream_t st) {
    if (n <= 0) return 0;
    fill_k<<<nblocks(n, TPB, 4096), TPB, 0, st>>>(x, n, v);
    GG_LAUNCH_CHECK();
}
// device-to-device copy as a plain kernel: a launch is several microseconds cheaper on the host than hipMemcpyAsync
__global__ void copy_k(float* dst, const float* src, long n, int vec) {
    const long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x, step = (long)gridDim.x * blockDim.x;
    if (vec) {
        const long n4 = n >> 2;
        for (long i = i0; i < n4; i += step) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
        for (long i = (n4 << 2) + i0; i < n; i += step) dst[i] = src[i];
    } else {
        for (long i = i0; i < n; i += step) dst[i] = src[i];
    }
}
int k_copy(float* dst, const float* src, long n, hipStream_t st) {
    if (n <= 0) return 0;
    const int vec = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
    copy_k<<<nblocks(vec ? (n + 3) / 4 : n, TPB, 8192), TPB, 0, st>>>(dst, src, n, vec);
    GG_LAUNCH_CHECK();
}
__global__ void copy_rows_bcast_k(float* dst, const float* src, long rows, long src_rows, int cols) {
    const long n = rows * cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        dst[i] = src[(r % src_rows) * cols + c];
    }
}
int k_copy_rows_bcast(float* dst, const float* src, long rows, long src_rows, int cols, hipStream_t st) {
    copy_rows_bcast_k<<<nblocks(rows * cols, TPB, 8192), TPB, 0, st>>>(dst, src, rows, src_rows, cols);
    GG_LAUNCH_CHECK();
}
__global__ void copy_rows_strided_bcast_k(float* dst, long ldd, const float* src, long lds, long rows, long src_rows, int cols) {
    const long n = rows * cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * ldd + c] = src[(r % src_rows) * lds + c];
    }
}
int k_copy_rows_strided_bcast(float* dst, long ldd, const float* src, long lds, long rows, long src_rows, int cols, hipStream_t st) {
    copy_rows_strided_bcast_k<<<nblocks(rows * cols, TPB, 8192), TPB, 0, st>>>(dst, ldd, src, lds, rows, src_rows, cols);
    GG_LAUNCH_CHECK();
}
__global__ void fold_rows_add_k(float* dst, long ldd, const float* src, long B, int R, int cols) {
    const long n = B * cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long b = i / cols;
        const int c = (int)(i - b * cols);
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += src[((long)r * B + b) * cols + c];
        dst[b * ldd + c] += s;
    }
}
int k_fold_rows_add(float* dst, long ldd, const float* src, long B, int R, int cols, hipStream_t st) {
    fold_rows_add_k<<<nblocks(B * cols, TPB, 8192), TPB, 0, st>>>(dst, ldd, src, B, R, cols);
    GG_LAUNCH_CHECK();
}
__global__ void axpy_k(float* y, const float* x, float a, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += a * x[i];
}
int k_axpy(float* y, const float* x, float a, long n, hipStream_t st) {
    if (n <= 0) return 0;
    axpy_k<<<nblocks(n, TPB, 8192), TPB, 0, st>>>(y, x, a, n);
    GG_LAUNCH_CHECK();
}
__global__ void add_bcast_rows_k(float* y, const float* x, long rows, long x_rows, int cols) {
    const long n = rows * cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        y[i] += x[(r % x_rows) * cols + (i - r * cols)];
    }
}
int k_add_bcast_rows(float* y, const float* x, long rows, long x_rows, int cols, hipStream_t st) {
    add_bcast_rows_k<<<nblocks(rows * cols, TPB, 8192), TPB, 0, st>>>(y, x, rows, x_rows, cols);
    GG_LAUNCH_CHECK();
}

// ---- FiLM ---------------------------------------------------------------------------------------
__global__ void film_act_fwd_k(const float* pre, float* gb, int B, int Dp) {
    const long n = (long)B * 2 * Dp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (2 * Dp));
        const float v = pre[i];
        gb[i] = (c < Dp) ? tanhf(v) : fminf(fmaxf(v, -5.f), 5.f);
    }
}
int k_film_act_fwd(const float* gb_pre, float* gb, int B, int Dp, hipStream_t st) {
    film_act_fwd_k<<<nblocks((long)B * 2 * Dp), TPB, 0, st>>>(gb_pre, gb, B, Dp);
    GG_LAUNCH_CHECK();
}
__global__ void film_act_bwd_k(float* dgb, const float* gb, const float* pre, int B, int Dp) {
    const long n = (long)B * 2 * Dp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (2 * Dp));
        const float d = dgb[i];
        if (c < Dp) {
            const float g = gb[i];
            dgb[i] = d * (1.f - g * g);
        } else {
            const float v = pre[i];
            dgb[i] = (v >= -5.f && v <= 5.f) ? d : 0.f;
        }
    }
}
int k_film_act_bwd(float* dgb, const float* gb, const float* gb_pre, int B, int Dp, hipStream_t st) {
    film_act_bwd_k<<<nblocks((long)B * 2 * Dp), TPB, 0, st>>>(dgb, gb, gb_pre, B, Dp);
    GG_LAUNCH_CHECK();
}
__global__ void film_mod_k(const float* patches, const float* gb, float* mod, int B, int P, int Dp) {
    const long n = (long)B * P * Dp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % Dp);
        const long b = i / ((long)P * Dp);
        mod[i] = gb[b * 2 * Dp + d] * patches[i] + gb[b * 2 * Dp + Dp + d];
    }
}
int k_film_mod(const float* patches, const float* gb, float* mod, int B, int P, int Dp, hipStream_t st) {
    film_mod_k<<<nblocks((long)B * P * Dp, TPB, 16384), TPB, 0, st>>>(patches, gb, mod, B, P, Dp);
    GG_LAUNCH_CHECK();
}
// one block per (b, chunk of 256 channels): thread owns channel d, loops over p (coalesced across d)
__global__ void film_bwd_reduce_k(const float* dmod, const float* patches, float* dgb, int B, int P, int Dp) {
    const int b = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= Dp) return;
    float sg = 0.f, sb = 0.f;
    const long base = (long)b * P * Dp + d;
    for (int p = 0; p < P; ++p) {
        const float dm = dmod[base + (long)p * Dp];
        sg += dm * patches[base + (long)p * Dp];
        sb += dm;
    }
    dgb[(long)b * 2 * Dp + d] = sg;
    dgb[(long)b * 2 * Dp + Dp + d] = sb;
}
int k_film_bwd_reduce(const float* dmod, const float* patches, float* dgb, int B, int P, int Dp, hipStream_t st) {
    dim3 grid((Dp + TPB - 1) / TPB, B);
    film_bwd_reduce_k<<<grid, TPB, 0, st>>>(dmod, patches, dgb, B, P, Dp);
    GG_LAUNCH_CHECK();
}

// ---- CLS row / mask -----------------------------------------------------------------------------
__global__ void write_cls_k(float* seq, const float* cls, int B, int S, int E) {
    const long n = (long)B * E;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long b = i / E;
        const int e = (int)(i % E);
        seq[b * S * E + e] = cls[e];
    }
}
int k_write_cls(float* seq, const float* cls, int B, int S, int E, hipStream_t st) {
    write_cls_k<<<nblocks((long)B * E), TPB, 0, st>>>(seq, cls, B, S, E);
    GG_LAUNCH_CHECK();
}
// dcls[e] += sum_b dseq[b, 0, e]: the B rows are S*E floats apart, so the batch is split over blockIdx.y and every
// thread keeps 8 loads in flight; partial sums meet in dcls through atomics (contiguous 256 B per wave)
__global__ void cls_grad_k(const float* dseq, float* dcls, int B, int S, int E) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int per = (B + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
    float s = 0.f;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = dseq[(long)(b + u) * S * E + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; b < b1; ++b) s += dseq[(long)b * S * E + e];
    atomicAdd(dcls + e, s);
}
int k_cls_grad(const float* dseq, float* dcls, int B, int S, int E, hipStream_t st) {
    dim3 grid((E + 63) / 64, B >= 64 ? 16 : 1);
    cls_grad_k<<<grid, 64, 0, st>>>(dseq, dcls, B, S, E);
    GG_LAUNCH_CHECK();
}
__global__ void build_mask_k(const uint8_t* pad, uint8_t* out, int B, int P) {
    const long n = (long)B * (P + 1);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long b = i / (P + 1);
        const int s = (int)(i % (P + 1));
        out[i] = (s == 0) ? 0 : (pad[b * P + s - 1] ? 1 : 0);
    }
}
int k_build_mask(const uint8_t* pad, uint8_t* mask_out, int B, int P, hipStream_t st) {
    build_mask_k<<<nblocks((long)B * (P + 1)), TPB, 0, st>>>(pad, mask_out, B, P);
    GG_LAUNCH_CHECK();
}

// ---- softmax over rows (one wave per row, row cached in LDS) ----------------------------------------
constexpr int SM_MAXC = 2048;
__global__ __launch_bounds__(TPB) void softmax_rows_k(float* S, float* Pd, long rows, int cols, DropKey drop_in) {
    const DropKey drop = drop_live(drop_in);
    __shared__ float buf[4][SM_MAXC];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    for (long row = blockIdx.x * 4L + wave; row < rows; row += gridDim.x * 4L) {
        float* s = S + row * cols;
        float m = -INFINITY;
        for (int c = lane; c < cols; c += 64) {
            const float v = s[c];
            buf[wave][c] = v;
            m = fmaxf(m, v);
        }
        m = wave_max(m);
        float sum = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float e = __expf(buf[wave][c] - m);
            buf[wave][c] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int c = lane; c < cols; c += 64) {
            const float p = buf[wave][c] * inv;
            s[c] = p;
            if (Pd) Pd[row * cols + c] = p * drop_factor(drop, (uint64_t)row * drop_attn_ld(cols) + c, ks);
        }
    }
}
int k_softmax_rows(float* S, float* Pd, long rows, int cols, DropKey drop, hipStream_t st) {
    GG_REQUIRE(cols <= SM_MAXC, "softmax row too long");
    if (drop.p <= 0.f) Pd = nullptr;
    softmax_rows_k<<<nblocks(rows, 4, 65535), TPB, 0, st>>>(S, Pd, rows, cols, drop);
    GG_LAUNCH_CHECK();
}
__global__ __launch_bounds__(TPB) void softmax_bwd_rows_k(float* dP, const float* P, long rows, int cols, float scale,
                                                          DropKey drop_in) {
    const DropKey drop = drop_live(drop_in);
    __shared__ float buf[4][SM_MAXC];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    for (long row = blockIdx.x * 4L + wave; row < rows; row += gridDim.x * 4L) {
        float* d = dP + row * cols;
        const float* p = P + row * cols;
        float dot = 0.f;
        for (int c = lane; c < cols; c += 64) {
            float g = d[c];
            if (drop.p > 0.f) g *= drop_factor(drop, (uint64_t)row * drop_attn_ld(cols) + c, ks);
            buf[wave][c] = g;
            dot += g * p[c];
        }
        dot = wave_sum(dot);
        for (int c = lane; c < cols; c += 64) d[c] = p[c] * (buf[wave][c] - dot) * scale;
    }
}
int k_softmax_bwd_rows(float* dP, const float* P, long rows, int cols, float scale, DropKey drop, hipStream_t st) {
    GG_REQUIRE(cols <= SM_MAXC, "softmax row too long");
    softmax_bwd_rows_k<<<nblocks(rows, 4, 65535), TPB, 0, st>>>(dP, P, rows, cols, scale, drop);
    GG_LAUNCH_CHECK();
}

// ---- residual add + LayerNorm (one wave per row; NJ = ceil(E/64) values per lane) ---------------------
constexpr float LN_EPS = 1e-5f;
template <int NJ>
__global__ __launch_bounds__(TPB) void add_ln_fwd_k(const float* x, long x_rows, float* res, const float* g,
                                                     const float* b, float* y, float* stats, long rows, int E,
                                                     DropKey drop_in) {
    const DropKey drop = drop_live(drop_in);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    for (long row = blockIdx.x * 4L + wave; row < rows; row += gridDim.x * 4L) {
        const float* xr = x + (row % x_rows) * E;
        float* rr = res + row * E;
        float v[NJ];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            float t = 0.f;
            if (c < E) {
                float rv = rr[c];
                if (drop.p > 0.f) rv *= drop_factor(drop, (uint64_t)row * E + c, ks);
                t = xr[c] + rv;
                rr[c] = t;
            }
            v[j] = t;
            sum += t;
        }
        const float mean = wave_sum(sum) / E;
        float var = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            const float dlt = (c < E) ? v[j] - mean : 0.f;
            var += dlt * dlt;
        }
        var = wave_sum(var) / E;
        const float rstd = rsqrtf(var + LN_EPS);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            if (c < E) y[row * E + c] = (v[j] - mean) * rstd * g[c] + b[c];
        }
        if (lane == 0) {
            stats[2 * row] = mean;
            stats[2 * row + 1] = rstd;
        }
    }
}
int k_add_layernorm_fwd(const float* x, long x_rows, float* res, const float* g, const float* b, float* y,
                        float* stats, long rows, int E, DropKey drop, hipStream_t st) {
    const unsigned nb = nblocks(rows, 4, 65535);
#define GG_LN_FWD(NJ) add_ln_fwd_k<NJ><<<nb, TPB, 0, st>>>(x, x_rows, res, g, b, y, stats, rows, E, drop)
    if (E <= 64) GG_LN_FWD(1);
    else if (E <= 128) GG_LN_FWD(2);
    else if (E <= 256) GG_LN_FWD(4);
    else if (E <= 512) GG_LN_FWD(8);
    else if (E <= 1024) GG_LN_FWD(16);
    else { set_error("LayerNorm width > 1024 unsupported"); return -2; }
#undef GG_LN_FWD
    GG_LAUNCH_CHECK();
}

template <int NJ>
__global__ __launch_bounds__(TPB) void ln_bwd_k(const float* dy, const float* r, const float* stats, const float* g,
                                                 float* dr, float* dres, float* dgamma, float* dbeta, float* dbias,
                                                 long rows, int E, DropKey drop_in) {
    const DropKey drop = drop_live(drop_in);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    float pg[NJ], pb[NJ], pc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) pg[j] = pb[j] = pc[j] = 0.f;
    for (long row = blockIdx.x * 4L + wave; row < rows; row += gridDim.x * 4L) {
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        float xh[NJ], dxh[NJ];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            xh[j] = dxh[j] = 0.f;
            if (c < E) {
                const float d = dy[row * E + c];
                xh[j] = (r[row * E + c] - mean) * rstd;
                dxh[j] = d * g[c];
                pg[j] += d * xh[j];
                pb[j] += d;
                s1 += dxh[j];
                s2 += dxh[j] * xh[j];
            }
        }
        s1 = wave_sum(s1) / E;
        s2 = wave_sum(s2) / E;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            if (c < E) {
                const float v = rstd * (dxh[j] - s1 - xh[j] * s2);
                dr[row * E + c] = v;
                const float vb = drop.p > 0.f ? v * drop_factor(drop, (uint64_t)row * E + c, ks) : v;
                if (dres) dres[row * E + c] = vb;
                pc[j] += vb;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        if (c < E) {
            atomicAdd(&dgamma[c], pg[j]);
            atomicAdd(&dbeta[c], pb[j]);
            if (dbias) atomicAdd(&dbias[c], pc[j]);
        }
    }
}
// float4 variant (E % 4 == 0): a lane owns 4 adjacent columns per 256-column group -> one 16-byte access per
// tensor per group (1 KiB per wave instruction), 4x fewer memory instructions than the scalar kernel.
typedef __bf16 bf16x2_k __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_k __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_k(float a, float b) {
    bf16x2_k v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
template <int NV, bool RB16 = false>
__global__ __launch_bounds__(TPB) void ln_bwd_v4_k(const float* dy, const float* r, const float* stats, const float* g,
                                                    float* dr, void* dres, int dres_bf16, float* dgamma, float* dbeta,
                                                    float* dbias, long rows, int E, DropKey drop_in) {
    const DropKey drop = drop_live(drop_in);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float ks = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    f32x4 pg[NV], pb[NV], pc[NV], gw[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        pg[j] = pb[j] = pc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = 4 * lane + 256 * j;
        gw[j] = c < E ? *reinterpret_cast<const f32x4*>(g + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // The next row's loads are issued before this row is reduced (register double buffer; rows past the end re-read the
    // last row and are not used): a load - reduce - store loop exposes one memory round trip per row to each wave.
    const long rstep = gridDim.x * 4L;
    long row = blockIdx.x * 4L + wave;
    f32x4 nd[NV], nr[NV];
    float nmean = 0.f, nrstd = 0.f;
    auto request = [&](long rw) {
        const long rc = min(rw, rows - 1);
        nmean = stats[2 * rc]; nrstd = stats[2 * rc + 1];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = min(4 * lane + 256 * j, E - 4);
            nd[j] = *reinterpret_cast<const f32x4*>(dy + rc * E + c);
            if constexpr (RB16) {          // pre-LayerNorm sums stored in bf16 (engine.hip "rst")
                const u32x2_k w = *reinterpret_cast<const u32x2_k*>(reinterpret_cast<const __bf16*>(r) + rc * E + c);
                nr[j] = f32x4{__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u),
                              __builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xffff0000u)};
            } else {
                nr[j] = *reinterpret_cast<const f32x4*>(r + rc * E + c);
            }
        }
    };
    if (row < rows) request(row);
    for (; row < rows; row += rstep) {
        const float mean = nmean, rstd = nrstd;
        f32x4 cd[NV], cr[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) { cd[j] = nd[j]; cr[j] = nr[j]; }
        request(row + rstep);
        f32x4 xh[NV], dxh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * lane + 256 * j;
            xh[j] = dxh[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < E) {
                const f32x4 d = cd[j];
                const f32x4 rv = cr[j];
                xh[j] = (rv - mean) * rstd;
                dxh[j] = d * gw[j];
                pg[j] += d * xh[j];
                pb[j] += d;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s1 += dxh[j][q];
                    s2 += dxh[j][q] * xh[j][q];
                }
            }
        }
        s1 = wave_sum(s1) / E;
        s2 = wave_sum(s2) / E;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * lane + 256 * j;
            if (c < E) {
                const f32x4 v = (dxh[j] - s1 - xh[j] * s2) * rstd;
                *reinterpret_cast<f32x4*>(dr + row * E + c) = v;
                f32x4 vb = v;
                if (drop.p > 0.f) {
                    float f[4];     // E % 4 == 0 and c % 4 == 0: the four elements are two aligned pairs
                    drop_factor4(drop, (uint64_t)row * E + c, ks, f);
#pragma unroll
                    for (int q = 0; q < 4; ++q) vb[q] = v[q] * f[q];
                }
                if (dres) {
                    if (dres_bf16) {
                        u32x2_k w = {pack2_k(vb[0], vb[1]), pack2_k(vb[2], vb[3])};
                        *reinterpret_cast<u32x2_k*>(reinterpret_cast<__bf16*>(dres) + row * E + c) = w;
                    } else {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dres) + row * E + c) = vb;
                    }
                }
                pc[j] += vb;
            }
        }
    }
    // fold the 4 waves' column partials through LDS so that every atomic wave-instruction covers 256 contiguous
    // bytes (4-byte slots at a 16-byte lane stride take the slow scattered-atomic path)
    __shared__ float red[4][256 * NV];
    float* outs[3] = {dgamma, dbeta, dbias};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (outs[k] == nullptr) continue;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const f32x4 v = k == 0 ? pg[j] : (k == 1 ? pb[j] : pc[j]);
            *reinterpret_cast<f32x4*>(&red[wave][4 * lane + 256 * j]) = v;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < E; c += TPB) atomicAdd(&outs[k][c], red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    }
}

int k_layernorm_bwd(const float* dy, const float* r, const float* stats, const float* g, float* dr, void* dres_out,
                    float* dgamma, float* dbeta, float* dbias, long rows, int E, DropKey drop, hipStream_t st, int dres_flags) {
    const int dres_bf16 = dres_flags & 1, r_bf16 = (dres_flags >> 1) & 1;        // bit 0: bf16 branch-gradient output, bit 1: bf16 r
    // Rows per wave: the kernel ends with 3 x E atomic adds per workgroup into the SAME E addresses (gamma / beta / bias
    // gradients), which serialise per address at the memory side - at 16 rows per wave (2 056 workgroups for the 131 584
    // rows of a backward pass) that tail was a fifth of the kernel (122 us); 32 rows: 99 us (4.4 TB/s), 64 rows: too few
    // waves per CU to cover the load latency (125 us).  GG_LN_ROWS overrides for measurements.
    static const int ln_rows = getenv("GG_LN_ROWS") ? atoi(getenv("GG_LN_ROWS")) : 32;
    const unsigned nb = nblocks(rows, 4 * ln_rows, 4096);
    const bool al = ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(g) |
                      reinterpret_cast<uintptr_t>(dr) | reinterpret_cast<uintptr_t>(dres_out)) & 15) == 0;
    if (E % 4 == 0 && E <= 1024 && al) {
#define GG_LN_BWD4(NV) ln_bwd_v4_k<NV><<<nb, TPB, 0, st>>>(dy, r, stats, g, dr, dres_out, dres_bf16, dgamma, dbeta, dbias, rows, E, drop)
        if (r_bf16) {
            GG_REQUIRE(E == 256, "bf16 pre-LayerNorm sums: production width only");
            ln_bwd_v4_k<1, true><<<nb, TPB, 0, st>>>(dy, r, stats, g, dr, dres_out, dres_bf16, dgamma, dbeta, dbias, rows, E, drop);
        } else
        if (E <= 256) GG_LN_BWD4(1);
        else if (E <= 512) GG_LN_BWD4(2);
        else GG_LN_BWD4(4);
#undef GG_LN_BWD4
        GG_LAUNCH_CHECK();
    }
    GG_REQUIRE(!dres_bf16 && !r_bf16, "bf16 branch-gradient output / bf16 sums need the vectorised LayerNorm backward (E % 4 == 0, aligned)");
#define GG_LN_BWD(NJ) ln_bwd_k<NJ><<<nb, TPB, 0, st>>>(dy, r, stats, g, dr, reinterpret_cast<float*>(dres_out), dgamma, dbeta, dbias, rows, E, drop)
    if (E <= 64) GG_LN_BWD(1);
    else if (E <= 128) GG_LN_BWD(2);
    else if (E <= 256) GG_LN_BWD(4);
    else if (E <= 512) GG_LN_BWD(8);
    else if (E <= 1024) GG_LN_BWD(16);
    else { set_error("LayerNorm width > 1024 unsupported"); return -2; }
#undef GG_LN_BWD
    GG_LAUNCH_CHECK();
}

// ---- column sums (bias gradients) ------------------------------------------------------------------
// grid (row chunks, column groups of 256): a thread owns 4 adjacent columns (one 16-byte load per row) of a
// 4-row-interleaved strip, so a wave reads whole 1 KiB row segments; partial sums are folded through LDS and
// leave as one atomic per column per workgroup.
constexpr int CS_ROWS = 256;
// bf16 input variant (bias gradients of bf16-stored branch gradients): same tiling, 8-byte loads of 4 bf16
__global__ __launch_bounds__(TPB) void colsum_bf16_k(const __bf16* X, long rows, int N, long ld, float* out) {
    __shared__ float red[4][256];
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c0 = blockIdx.y * 256 + cq * 4;
    const long r0 = (long)blockIdx.x * 256;
    const long r1 = min(rows, r0 + 256);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 + 3 < N) {
        for (long r = r0 + rl; r < r1; r += 4) {
            const u32x2_k w = *reinterpret_cast<const u32x2_k*>(X + r * ld + c0);
            s[0] += __builtin_bit_cast(float, w[0] << 16);
            s[1] += __builtin_bit_cast(float, w[0] & 0xffff0000u);
            s[2] += __builtin_bit_cast(float, w[1] << 16);
            s[3] += __builtin_bit_cast(float, w[1] & 0xffff0000u);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][cq * 4 + j] = s[j];
    __syncthreads();
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c < N) atomicAdd(&out[c], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(TPB) void colsum_k(const float* X, const float* ref, long rows, int N, long ld, float slope,
                                                float* out, int rows_per_block) {
    __shared__ float red[4][256];
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;            // column quad, row lane (0..3)
    const int c0 = blockIdx.y * 256 + cq * 4;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    const bool vec = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && (!ref || (reinterpret_cast<uintptr_t>(ref) & 15) == 0);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < N) {
        if (vec && c0 + 3 < N) {
            // 8 rows per thread at a time, all loads issued before the first add (clamped rows, masked adds): a load - add
            // loop costs one memory round trip per row, and the short matrices this kernel mostly sees are 8 rows per thread
            for (long rb = r0 + rl; rb < r1; rb += 32) {
                f32x4 v[8], q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(X + min(rb + 4 * u, rows - 1) * ld + c0);
                if (ref) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) q[u] = *reinterpret_cast<const f32x4*>(ref + min(rb + 4 * u, rows - 1) * ld + c0);
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[j] += rb + 4 * u < r1 ? v[u][j] * (q[u][j] > 0.f ? 1.f : slope) : 0.f;
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[j] += rb + 4 * u < r1 ? v[u][j] : 0.f;
                }
            }
        } else {
            for (long r = r0 + rl; r < r1; r += 4)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c0 + j < N) {
                        const float v = X[r * ld + c0 + j];
                        s[j] += ref ? v * (ref[r * ld + c0 + j] > 0.f ? 1.f : slope) : v;
                    }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][cq * 4 + j] = s[j];
    __syncthreads();
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c < N) atomicAdd(&out[c], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
int k_colsum(const void* X, long rows, int N, long ld, float* out, hipStream_t st, int x_bf16) {
    dim3 grid((unsigned)((rows + CS_ROWS - 1) / CS_ROWS), (unsigned)((N + 255) / 256));
    if (x_bf16) {
        GG_REQUIRE(N % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 7) == 0, "bf16 colsum needs 4-element alignment");
        colsum_bf16_k<<<grid, TPB, 0, st>>>(reinterpret_cast<const __bf16*>(X), rows, N, ld, out);
        GG_LAUNCH_CHECK();
    }
    // short matrices (the [B, E] tensors of the heads and cross-attention): 32 rows per workgroup instead of 256, the
    // 8 loads of a thread are in flight together and the kernel is one memory round trip long instead of 64
    const int rpb = rows <= 8192 ? 32 : CS_ROWS;
    const dim3 grid2((unsigned)((rows + rpb - 1) / rpb), (unsigned)((N + 255) / 256));
    colsum_k<<<grid2, TPB, 0, st>>>(reinterpret_cast<const float*>(X), nullptr, rows, N, ld, 0.f, out, rpb);
    GG_LAUNCH_CHECK();
}
int k_colsum_masked(const float* X, const float* ref, long rows, int N, float slope, float* out, hipStream_t st) {
    const int rpb = rows <= 8192 ? 32 : CS_ROWS;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((N + 255) / 256));
    colsum_k<<<grid, TPB, 0, st>>>(X, ref, rows, N, N, slope, out, rpb);
    GG_LAUNCH_CHECK();
}

__global__ void act_bwd_k(float* y, const float* ref, long n, float slope, float scale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] *= (ref[i] > 0.f ? 1.f : slope) * scale;
}
int k_act_bwd(float* y, const float* ref, long n, float slope, float scale, hipStream_t st) {
    act_bwd_k<<<nblocks(n, TPB, 16384), TPB, 0, st>>>(y, ref, n, slope, scale);
    GG_LAUNCH_CHECK();
}
__global__ void bias_act_k(float* y, const float* bias, long rows, int N, int act, float slope) {
    const long n = rows * N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = y[i];
        if (bias) v += bias[i % N];
        if (act == ACT_LRELU) v = v > 0.f ? v : slope * v;
        y[i] = v;
    }
}
int k_bias_act(float* y, const float* bias, long rows, int N, int act, float slope, hipStream_t st) {
    bias_act_k<<<nblocks(rows * N, TPB, 16384), TPB, 0, st>>>(y, bias, rows, N, act, slope);
    GG_LAUNCH_CHECK();
}
__global__ void dropout_k(float* x, long n, DropKey drop_in) {
    const DropKey drop = drop_live(drop_in);
    const float ks = 1.f / (1.f - drop.p);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] *= drop_factor(drop, (uint64_t)i, ks);
}
int k_dropout(float* x, long n, DropKey drop, hipStream_t st) {
    if (drop.p <= 0.f || n <= 0) return 0;
    dropout_k<<<nblocks(n, TPB, 16384), TPB, 0, st>>>(x, n, drop);
    GG_LAUNCH_CHECK();
}
__global__ void rowscale_k(float* y, const float* s, long rows, int N) {
    const long n = rows * N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] *= s[i / N];
}
int k_rowscale(float* y, const float* s, long rows, int N, hipStream_t st) {
    rowscale_k<<<nblocks(rows * N, TPB, 16384), TPB, 0, st>>>(y, s, rows, N);
    GG_LAUNCH_CHECK();
}
__global__ void mask_times_vec_k(float* out, const float* ref, const float* w, long rows, int N, float slope) {
    const long n = rows * N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (ref[i] > 0.f ? 1.f : slope) * w[i % N];
}
int k_mask_times_vec(float* out, const float* ref, const float* w, long rows, int N, float slope, hipStream_t st) {
    mask_times_vec_k<<<nblocks(rows * N), TPB, 0, st>>>(out, ref, w, rows, N, slope);
    GG_LAUNCH_CHECK();
}

// ---- single-query attention ------------------------------------------------------------------------
// block = one sample b; wave w handles heads w, w+4, ...; scores of one head live in LDS.
constexpr int SQ_MAXS = 2048;
__global__ __launch_bounds__(TPB) void sq_attn_fwd_k(const float* q, const float* kv, const uint8_t* mask, int mask_B,
                                                      float* probs, float* ctx, int S, int E, int nh) {
    __shared__ float sc[4][SQ_MAXS];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dh = E / nh;
    const float scale = rsqrtf((float)dh);
    const float* kvb = kv + (long)b * S * 2 * E;
    for (int h = wave; h < nh; h += 4) {
        const float* qh = q + (long)b * E + h * dh;
        float m = -INFINITY;
        for (int s = lane; s < S; s += 64) {
            const float* kr = kvb + (long)s * 2 * E + h * dh;
            float dot = 0.f;
            for (int d = 0; d < dh; ++d) dot += qh[d] * kr[d];
            dot *= scale;
            if (mask && mask[(long)(b % mask_B) * S + s]) dot = -INFINITY;
            sc[wave][s] = dot;
            m = fmaxf(m, dot);
        }
        m = wave_max(m);
        float sum = 0.f;
        for (int s = lane; s < S; s += 64) {
            const float e = __expf(sc[wave][s] - m);
            sc[wave][s] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int s = lane; s < S; s += 64) {
            const float p = sc[wave][s] * inv;
            sc[wave][s] = p;
            probs[((long)b * nh + h) * S + s] = p;
        }
        // wave-private LDS row: make the writes above visible to all lanes of this wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int d = lane; d < dh; d += 64) {
            float acc = 0.f;
            for (int s = 0; s < S; ++s) acc += sc[wave][s] * kvb[(long)s * 2 * E + E + h * dh + d];
            ctx[(long)b * E + h * dh + d] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Production width (E = 256, 4 heads of 64) with many keys (real text: T = 300): lane l of every wave owns features
// 4l..4l+3 (head l >> 4), the four waves stride over the keys four rows at a time, so every K / V row is one coalesced
// 1 KB access and is read exactly once per pass (the generic kernels above walk a row per lane).
constexpr int SQ256_MAXS = 1024;
template <int CTRL>
__device__ __forceinline__ float dpp_k(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum_k(float v) {      // every lane of a 16-lane row ends with the row's sum
    v += dpp_k<0xB1>(v);
    v += dpp_k<0x4E>(v);
    v += dpp_k<0x141>(v);
    v += dpp_k<0x140>(v);
    return v;
}
__device__ __forceinline__ float dot4_k(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

__global__ __launch_bounds__(TPB) void sq256_fwd_k(const float* __restrict__ q, const float* __restrict__ kv,
                                                    const uint8_t* __restrict__ mask, int mask_B, float* __restrict__ probs,
                                                    float* __restrict__ ctx, int S, int kv_B) {
    constexpr int E = 256, NH = 4;
    __shared__ float sc[NH][SQ256_MAXS];
    __shared__ __attribute__((aligned(16))) float part[4][E];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 4;
    const float* kvb = kv + (long)(b % kv_B) * S * 2 * E + 4 * lane;
    const f32x4 qv = *reinterpret_cast<const f32x4*>(q + (long)b * E + 4 * lane) * 0.125f;       // 1/sqrt(64)
    const uint8_t* mk = mask ? mask + (long)(b % mask_B) * S : nullptr;
    for (int s0 = 4 * wave; s0 < S; s0 += 16) {
        f32x4 kr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) kr[u] = *reinterpret_cast<const f32x4*>(kvb + (long)min(s0 + u, S - 1) * 2 * E);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float d = row16_sum_k(dot4_k(qv, kr[u]));
            if ((lane & 15) == 0 && s0 + u < S) sc[h][s0 + u] = (mk && mk[s0 + u]) ? -INFINITY : d;
        }
    }
    __syncthreads();
    {   // softmax of head `wave`
        float m = -INFINITY;
        for (int s = lane; s < S; s += 64) m = fmaxf(m, sc[wave][s]);
        m = wave_max(m);
        float sum = 0.f;
        for (int s = lane; s < S; s += 64) {
            const float e = __expf(sc[wave][s] - m);
            sc[wave][s] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int s = lane; s < S; s += 64) {
            const float p = sc[wave][s] * inv;
            sc[wave][s] = p;
            probs[((long)b * NH + wave) * S + s] = p;
        }
    }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 4 * wave; s0 < S; s0 += 16) {
        f32x4 vr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vr[u] = *reinterpret_cast<const f32x4*>(kvb + (long)min(s0 + u, S - 1) * 2 * E + E);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (s0 + u < S) acc += sc[h][s0 + u] * vr[u];
    }
    *reinterpret_cast<f32x4*>(&part[wave][4 * lane]) = acc;
    __syncthreads();
    const int t = threadIdx.x;
    ctx[(long)b * E + t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}

__global__ __launch_bounds__(TPB) void sq256_bwd_k(const float* __restrict__ dctx, const float* __restrict__ q,
                                                    const float* __restrict__ kv, const float* __restrict__ probs,
                                                    float* __restrict__ dq, float* __restrict__ dkv, int S) {
    constexpr int E = 256, NH = 4;
    __shared__ float ds[NH][SQ256_MAXS];       // dctx_h . V_s
    __shared__ float ps[NH][SQ256_MAXS];
    __shared__ __attribute__((aligned(16))) float part[4][E];
    __shared__ float red[4][NH];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 4;
    const float* kvb = kv + (long)b * S * 2 * E + 4 * lane;
    float* dkvb = dkv + (long)b * S * 2 * E + 4 * lane;
    const f32x4 dcv = *reinterpret_cast<const f32x4*>(dctx + (long)b * E + 4 * lane);
    const f32x4 qv = *reinterpret_cast<const f32x4*>(q + (long)b * E + 4 * lane);
    const float* ph = probs + ((long)b * NH + h) * S;
    float dotp = 0.f;
    for (int s0 = 4 * wave; s0 < S; s0 += 16) {
        f32x4 vr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vr[u] = *reinterpret_cast<const f32x4*>(kvb + (long)min(s0 + u, S - 1) * 2 * E + E);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float d = row16_sum_k(dot4_k(dcv, vr[u]));
            if (s0 + u < S) {
                const float p = ph[s0 + u];
                dotp += d * p;
                if ((lane & 15) == 0) {
                    ds[h][s0 + u] = d;
                    ps[h][s0 + u] = p;
                }
            }
        }
    }
    if ((lane & 15) == 0) red[wave][h] = dotp;
    __syncthreads();
    const float dot = (red[0][h] + red[1][h]) + (red[2][h] + red[3][h]);        // sum_s p_s dp_s of this lane's head
    f32x4 dqa = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 4 * wave; s0 < S; s0 += 16) {
        f32x4 kr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) kr[u] = *reinterpret_cast<const f32x4*>(kvb + (long)min(s0 + u, S - 1) * 2 * E);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (s0 + u < S) {
                const float p = ps[h][s0 + u];
                const float g = p * (ds[h][s0 + u] - dot) * 0.125f;             // d(score_s)
                *reinterpret_cast<f32x4*>(dkvb + (long)(s0 + u) * 2 * E) = g * qv;
                *reinterpret_cast<f32x4*>(dkvb + (long)(s0 + u) * 2 * E + E) = p * dcv;
                dqa += g * kr[u];
            }
        }
    }
    *reinterpret_cast<f32x4*>(&part[wave][4 * lane]) = dqa;
    __syncthreads();
    const int t = threadIdx.x;
    dq[(long)b * E + t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}
inline bool sq256_ok(const void* a, const void* b_, const void* c_, const void* d, int S, int E, int nh) {
    return E == 256 && nh == 4 && S >= 4 && S <= SQ256_MAXS &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b_) | reinterpret_cast<uintptr_t>(c_) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}
int k_sq_attn_fwd(const float* q, const float* kv, const uint8_t* mask, int mask_B, float* probs, float* ctx, int B,
                  int S, int E, int nh, hipStream_t st, int kv_B) {
    GG_REQUIRE(S <= SQ_MAXS, "single-query attention: too many keys");
    if (sq256_ok(q, kv, ctx, ctx, S, E, nh)) {
        sq256_fwd_k<<<B, TPB, 0, st>>>(q, kv, mask, mask_B > 0 ? mask_B : B, probs, ctx, S, kv_B > 0 ? kv_B : B);
        GG_LAUNCH_CHECK();
    }
    GG_REQUIRE(kv_B <= 0 || kv_B == B, "shared keys / values need the E = 256 kernels (sq_attn_shared_ok)");
    sq_attn_fwd_k<<<B, TPB, 0, st>>>(q, kv, mask, mask_B > 0 ? mask_B : B, probs, ctx, S, E, nh);
    GG_LAUNCH_CHECK();
}
__global__ __launch_bounds__(TPB) void sq_attn_bwd_k(const float* dctx, const float* q, const float* kv,
                                                      const float* probs, float* dq, float* dkv, int S, int E, int nh) {
    __shared__ float ds[4][SQ_MAXS];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dh = E / nh;
    const float scale = rsqrtf((float)dh);
    const float* kvb = kv + (long)b * S * 2 * E;
    float* dkvb = dkv + (long)b * S * 2 * E;
    for (int h = wave; h < nh; h += 4) {
        const float* dc = dctx + (long)b * E + h * dh;
        const float* qh = q + (long)b * E + h * dh;
        const float* ph = probs + ((long)b * nh + h) * S;
        float dot = 0.f;
        for (int s = lane; s < S; s += 64) {
            const float* vr = kvb + (long)s * 2 * E + E + h * dh;
            float dp = 0.f;
            for (int d = 0; d < dh; ++d) dp += dc[d] * vr[d];
            ds[wave][s] = dp;
            dot += dp * ph[s];
        }
        dot = wave_sum(dot);
        for (int s = lane; s < S; s += 64) {
            const float p = ph[s];
            const float g = p * (ds[wave][s] - dot) * scale;     // d(score_s)
            ds[wave][s] = g;
            float* dkr = dkvb + (long)s * 2 * E + h * dh;
            float* dvr = dkr + E;
            for (int d = 0; d < dh; ++d) {
                dkr[d] = g * qh[d];
                dvr[d] = p * dc[d];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int d = lane; d < dh; d += 64) {
            float acc = 0.f;
            for (int s = 0; s < S; ++s) acc += ds[wave][s] * kvb[(long)s * 2 * E + h * dh + d];
            dq[(long)b * E + h * dh + d] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// R replicas of sample b (rows r*B + b of dctx / q / probs / dq) attend to the SAME projected keys / values: one workgroup
// per sample reads every K / V row once and writes the replica-summed dK / dV row once.
template <int R>
__global__ __launch_bounds__(TPB) void sq256_bwd_shared_k(const float* __restrict__ dctx, const float* __restrict__ q,
                                                           const float* __restrict__ kv, const float* __restrict__ probs,
                                                           float* __restrict__ dq, float* __restrict__ dkv, int B, int S) {
    constexpr int E = 256, NH = 4;
    extern __shared__ __attribute__((aligned(16))) float sq_dyn[];
    float* const ds = sq_dyn;                      // [R][NH][S]  dctx_h . V_s
    float* const ps = sq_dyn + R * NH * S;         // [R][NH][S]  probabilities
    __shared__ __attribute__((aligned(16))) float part[4][E];
    __shared__ float red[4][R][NH];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 4;
    const float* kvb = kv + (long)b * S * 2 * E + 4 * lane;
    float* dkvb = dkv + (long)b * S * 2 * E + 4 * lane;
    f32x4 dcv[R], qv[R];
    float dotp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        dcv[r] = *reinterpret_cast<const f32x4*>(dctx + ((long)r * B + b) * E + 4 * lane);
        qv[r] = *reinterpret_cast<const f32x4*>(q + ((long)r * B + b) * E + 4 * lane);
        dotp[r] = 0.f;
    }
    for (int s0 = 4 * wave; s0 < S; s0 += 16) {
        f32x4 vr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vr[u] = *reinterpret_cast<const f32x4*>(kvb + (long)min(s0 + u, S - 1) * 2 * E + E);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float d = row16_sum_k(dot4_k(dcv[r], vr[u]));
                if (s0 + u < S) {
                    const float p = probs[(((long)r * B + b) * NH + h) * S + s0 + u];
                    dotp[r] += d * p;
                    if ((lane & 15) == 0) {
                        ds[(r * NH + h) * S + s0 + u] = d;
                        ps[(r * NH + h) * S + s0 + u] = p;
                    }
                }
            }
        }
    }
    if ((lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) red[wave][r][h] = dotp[r];
    }
    __syncthreads();
    float dot[R];
    f32x4 dqa[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        dot[r] = (red[0][r][h] + red[1][r][h]) + (red[2][r][h] + red[3][r][h]);
        dqa[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int s0 = 4 * wave; s0 < S; s0 += 16) {
        f32x4 kr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) kr[u] = *reinterpret_cast<const f32x4*>(kvb + (long)min(s0 + u, S - 1) * 2 * E);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (s0 + u < S) {
                f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float p = ps[(r * NH + h) * S + s0 + u];
                    const float g = p * (ds[(r * NH + h) * S + s0 + u] - dot[r]) * 0.125f;
                    dk += g * qv[r];
                    dv += p * dcv[r];
                    dqa[r] += g * kr[u];
                }
                *reinterpret_cast<f32x4*>(dkvb + (long)(s0 + u) * 2 * E) = dk;
                *reinterpret_cast<f32x4*>(dkvb + (long)(s0 + u) * 2 * E + E) = dv;
            }
        }
    }
    const int t = threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r > 0) __syncthreads();
        *reinterpret_cast<f32x4*>(&part[wave][4 * lane]) = dqa[r];
        __syncthreads();
        dq[((long)r * B + b) * E + t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    }
}
bool sq_attn_shared_ok(int S, int E, int nh, int R) {
    return E == 256 && nh == 4 && S >= 4 && S <= SQ256_MAXS && R >= 1 && R <= 3 && (size_t)2 * R * 4 * S * sizeof(float) <= 48 * 1024;
}
int k_sq_attn_bwd_shared(const float* dctx, const float* q, const float* kv, const float* probs, float* dq, float* dkv, int B,
                         int R, int S, int E, int nh, hipStream_t st) {
    GG_REQUIRE(sq_attn_shared_ok(S, E, nh, R) && sq256_ok(dctx, q, kv, dkv, S, E, nh), "shared-key single-query attention: unsupported shape");
    const size_t smem = (size_t)2 * R * 4 * S * sizeof(float);
    if (R == 1) sq256_bwd_shared_k<1><<<B, TPB, smem, st>>>(dctx, q, kv, probs, dq, dkv, B, S);
    else if (R == 2) sq256_bwd_shared_k<2><<<B, TPB, smem, st>>>(dctx, q, kv, probs, dq, dkv, B, S);
    else sq256_bwd_shared_k<3><<<B, TPB, smem, st>>>(dctx, q, kv, probs, dq, dkv, B, S);
    GG_LAUNCH_CHECK();
}
int k_sq_attn_bwd(const float* dctx, const float* q, const float* kv, const float* probs, float* dq, float* dkv, int B,
                  int S, int E, int nh, hipStream_t st) {
    GG_REQUIRE(S <= SQ_MAXS, "single-query attention: too many keys");
    if (sq256_ok(dctx, q, kv, dkv, S, E, nh)) {
        sq256_bwd_k<<<B, TPB, 0, st>>>(dctx, q, kv, probs, dq, dkv, S);
        GG_LAUNCH_CHECK();
    }
    sq_attn_bwd_k<<<B, TPB, 0, st>>>(dctx, q, kv, probs, dq, dkv, S, E, nh);
    GG_LAUNCH_CHECK();
}

// ---- replica fold / patch-row gather / dropout copy ---------------------------------------------------
__global__ void fold_k(float* out, const float* in, long n, int R) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += in[(long)r * n + i];
        out[i] = s;
    }
}
int k_fold(float* out, const float* in, long n, int R, hipStream_t st) {
    fold_k<<<nblocks(n, TPB, 16384), TPB, 0, st>>>(out, in, n, R);
    GG_LAUNCH_CHECK();
}
__global__ void gather_patch_rows_k(float* out, const float* seq, int B, int P, int E) {
    const long n = (long)B * P * E;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long row = i / E;
        const int e = (int)(i - row * E);
        const long b = row / P;
        out[i] = seq[(row + b + 1) * E + e];
    }
}
__global__ void scatter_patch_rows_k(float* seq, const float* in, int B, int P, int E) {
    const long n = (long)B * P * E;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long row = i / E;
        const int e = (int)(i - row * E);
        const long b = row / P;
        seq[(row + b + 1) * E + e] = in[i];
    }
}
int k_scatter_patch_rows(float* seq, const float* in, int B, int P, int E, hipStream_t st) {
    scatter_patch_rows_k<<<nblocks((long)B * P * E, TPB, 16384), TPB, 0, st>>>(seq, in, B, P, E);
    GG_LAUNCH_CHECK();
}
// Fold the R replicas of the layer-0 input gradient dx [R][B][S][E] and split it in the same pass: patch rows (s >= 1) go
// to demb [B*(S-1)][E], the CLS rows (s == 0) to cls_rows [B][E].  One read of dx, float4 accesses - replaces fold_k (scalar
// loads, a full [B,S,E] intermediate) + gather_patch_rows_k (a second read and write of the same 67 MB).
__global__ __launch_bounds__(TPB) void fold_gather_k(float* __restrict__ demb, float* __restrict__ cls_rows, const float* __restrict__ dx,
                                                     int B, int S, int E, int R) {
    const long n4 = (long)B * S * E / 4, rep = (long)B * S * E;
    const int E4 = E / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / E4;                       // b * S + s
        const int c = (int)(i - row * E4) * 4;
        const long b = row / S;
        const int sidx = (int)(row - b * S);
        const float* src = dx + row * E + c;
        f32x4 v = *reinterpret_cast<const f32x4*>(src);
        if (R > 1) v += *reinterpret_cast<const f32x4*>(src + rep);
        if (R > 2) v += *reinterpret_cast<const f32x4*>(src + 2 * rep);
        for (int r = 3; r < R; ++r) v += *reinterpret_cast<const f32x4*>(src + r * rep);
        float* dst = sidx == 0 ? cls_rows + b * E + c : demb + (row - b - 1) * E + c;
        *reinterpret_cast<f32x4*>(dst) = v;
    }
}
int k_fold_gather(float* demb, float* cls_rows, const float* dx, int B, int S, int E, int R, hipStream_t st) {
    GG_REQUIRE(E % 4 == 0 && ((reinterpret_cast<uintptr_t>(demb) | reinterpret_cast<uintptr_t>(cls_rows) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
               "fold_gather: 16-byte aligned rows expected");
    fold_gather_k<<<nblocks((long)B * S * E / 4, TPB, 16384), TPB, 0, st>>>(demb, cls_rows, dx, B, S, E, R);
    GG_LAUNCH_CHECK();
}
// c[r, :] = a[r, :] + b[r, :], or NaN for the rows whose sample (r % B) has pad[b] set (a softmax over one masked key)
__global__ void sum2_nan_rows_k(float* __restrict__ c, const float* __restrict__ a, const float* __restrict__ b,
                                const uint8_t* __restrict__ pad, long rows, int B, int E) {
    const long n = rows * E;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / E;
        c[i] = (pad && pad[r % B]) ? __builtin_nanf("") : a[i] + b[i];
    }
}
int k_sum2_nan_rows(float* c, const float* a, const float* b, const uint8_t* pad, long rows, int B, int E, hipStream_t st) {
    sum2_nan_rows_k<<<nblocks(rows * E, TPB, 4096), TPB, 0, st>>>(c, a, b, pad, rows, B, E);
    GG_LAUNCH_CHECK();
}
int k_gather_patch_rows(float* out, const float* seq, int B, int P, int E, hipStream_t st) {
    gather_patch_rows_k<<<nblocks((long)B * P * E, TPB, 16384), TPB, 0, st>>>(out, seq, B, P, E);
    GG_LAUNCH_CHECK();
}
__global__ void dropout_copy_k(float* out, const float* in, long n, DropKey drop_in) {
    const DropKey drop = drop_live(drop_in);
    const float ks = 1.f / (1.f - drop.p);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = in[i] * drop_factor(drop, (uint64_t)i, ks);
}
int k_dropout_copy(float* out, const float* in, long n, DropKey drop, hipStream_t st) {
    dropout_copy_k<<<nblocks(n, TPB, 16384), TPB, 0, st>>>(out, in, n, drop);
    GG_LAUNCH_CHECK();
}

// ---- losses / seeds ----------------------------------------------------------------------------------
__global__ void critic_loss_seed_k(const float* d, float* seed, float* losses, int B) {
    float sf = 0.f, sr = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        sf += d[i];
        sr += d[B + i];
        seed[i] = 1.f / B;
        seed[B + i] = -1.f / B;
    }
    __shared__ float red[2][TPB / 64];
    sf = wave_sum(sf);
    sr = wave_sum(sr);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sf;
        red[1][threadIdx.x >> 6] = sr;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, c = 0.f;
        for (int w = 0; w < TPB / 64; ++w) {
            a += red[0][w];
            c += red[1][w];
        }
        losses[0] += -c / B;   // mean(-d_true)
        losses[1] += a / B;    // mean(d_fake)
    }
}
int k_critic_loss_seed(const float* d, float* seed, float* losses, int B, hipStream_t st) {
    critic_loss_seed_k<<<1, TPB, 0, st>>>(d, seed, losses, B);
    GG_LAUNCH_CHECK();
}
__global__ void gen_loss_seed_k(const float* d, float* seed, float* losses, int B) {
    float sf = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        sf += d[i];
        seed[i] = -1.f / B;
    }
    __shared__ float red[TPB / 64];
    sf = wave_sum(sf);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sf;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int w = 0; w < TPB / 64; ++w) a += red[w];
        losses[3] += -a / B;
    }
}
int k_gen_loss_seed(const float* d, float* seed, float* losses, int B, hipStream_t st) {
    gen_loss_seed_k<<<1, TPB, 0, st>>>(d, seed, losses, B);
    GG_LAUNCH_CHECK();
}

// ---- gradient penalty helpers ------------------------------------------------------------------------
__global__ void lerp_rows_k(const float* Pfr, const float* alpha, float* out, int B, int H) {
    const long n = (long)B * H;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float a = alpha[i / H];
        out[i] = a * Pfr[n + i] + (1.f - a) * Pfr[i];
    }
}
int k_lerp_rows(const float* Pfr, const float* alpha, float* out, int B, int H, hipStream_t st) {
    lerp_rows_k<<<nblocks((long)B * H, TPB, 16384), TPB, 0, st>>>(Pfr, alpha, out, B, H);
    GG_LAUNCH_CHECK();
}
int k_lerp_genes(const float* xfr, const float* alpha, float* out, int B, int G, hipStream_t st) {
    return k_lerp_rows(xfr, alpha, out, B, G, st);
}
__global__ __launch_bounds__(TPB) void row_sumsq_k(const float* X, float* out, long rows, int N) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long row = blockIdx.x * 4L + wave; row < rows; row += gridDim.x * 4L) {
        float s = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float v = X[row * N + c];
            s += v * v;
        }
        s = wave_sum(s);
        if (lane == 0) out[row] = s;
    }
}
int k_row_sumsq(const float* X, float* out, long rows, int N, hipStream_t st) {
    row_sumsq_k<<<nblocks(rows, 4, 65535), TPB, 0, st>>>(X, out, rows, N);
    GG_LAUNCH_CHECK();
}
__global__ void gp_coef_k(const float* nrm2, float* coef, float* losses, int B, float w) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float n = sqrtf(nrm2[i]);
        const float d = n - 1.f;
        acc += d * d;
        coef[i] = n > 0.f ? w * (2.f / B) * d / n : 0.f;
    }
    __shared__ float red[TPB / 64];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int k = 0; k < TPB / 64; ++k) a += red[k];
        losses[2] += a / B;
    }
}
int k_gp_coef(const float* nrm2, float* coef, float* losses, int B, float gp_weight, hipStream_t st) {
    gp_coef_k<<<1, TPB, 0, st>>>(nrm2, coef, losses, B, gp_weight);
    GG_LAUNCH_CHECK();
}

// ---- optimiser -----------------------------------------------------------------------------------------
// Global gradient norm for clip_grad_norm_, DETERMINISTIC: every workgroup writes its partial sum to its own slot (no
// atomics), the optimiser kernel's workgroups each add the slots up in one fixed order.  Data-parallel replicas therefore
// compute bit-identical clip coefficients from their bit-identical all-reduced gradients and never drift apart.
constexpr int SUMSQ_MAX_BLOCKS = 1024;
__global__ __launch_bounds__(TPB) void sumsq_k(const float* x, long n, float* partials) {
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        s += v * v;
    }
    __shared__ float red[TPB / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int k = 0; k < TPB / 64; ++k) a += red[k];
        partials[blockIdx.x] = a;
    }
}
int k_sumsq(const float* x, long n, float* partials, int* n_partials, hipStream_t st) {
    const int nb = nblocks(n, TPB * 8, SUMSQ_MAX_BLOCKS);
    sumsq_k<<<nb, TPB, 0, st>>>(x, n, partials);
    *n_partials = nb;
    GG_LAUNCH_CHECK();
}
__global__ __launch_bounds__(TPB) void opt_step_k(float* w, const float* g, float* s1, float* s2, long n, int kind, float lr, float max_norm,
                                                   const float* partials, int n_partials, float grad_scale, int step_t,
                                                   const uint32_t* __restrict__ t_off) {
    float coef = grad_scale;
    float bc1 = 1.f, bc2s = 1.f;
    if (kind != OPT_RMSPROP) {      // Adam bias corrections; t_off: see gg_engine::dev_words (captured steps)
        const float t = (float)(step_t + (t_off ? (int)*t_off : 0));
        bc1 = 1.f - powf(0.9f, t);
        bc2s = sqrtf(1.f - powf(0.99f, t));
    }
    if (max_norm > 0.f) {
        __shared__ float red[TPB / 64];
        __shared__ float tot;
        float s = 0.f;
        for (int i = threadIdx.x; i < n_partials; i += TPB) s += partials[i];      // same order in every workgroup
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f;
            for (int k = 0; k < TPB / 64; ++k) a += red[k];
            tot = a;
        }
        __syncthreads();
        const float total = sqrtf(tot) * grad_scale;
        coef *= fminf(1.f, max_norm / (total + 1e-6f));
    }
    // four elements per thread and pass, every operand requested before the first use (16-byte accesses; the scalar tail below
    // handles n % 4 and unaligned buffers): the element-wise loop paid two to three dependent 4-byte round trips per element
    auto update = [&](float gr0, float& p, float& a, float& b2) {
        const float gr = gr0 * coef;
        if (kind == OPT_RMSPROP) {
            const float sq = 0.99f * a + 0.01f * gr * gr;
            a = sq;
            p -= lr * gr / (sqrtf(sq) + 1e-8f);
        } else {
            if (kind == OPT_ADAMW) p *= 1.f - lr * 0.01f;
            const float m = 0.9f * a + 0.1f * gr;
            const float v = 0.99f * b2 + 0.01f * gr * gr;
            a = m;
            b2 = v;
            p -= (lr / bc1) * m / (sqrtf(v) / bc2s + 1e-8f);
        }
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(s1) |
                       reinterpret_cast<uintptr_t>(s2)) & 15) == 0;
    const long n4 = vec ? n / 4 : 0;
    const float* s2r = s2 ? s2 : s1;          // RMSprop has no second state: the load stays unconditional and is ignored
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + 4 * i);
        f32x4 pv = *reinterpret_cast<const f32x4*>(w + 4 * i);
        f32x4 av = *reinterpret_cast<const f32x4*>(s1 + 4 * i);
        f32x4 bv = *reinterpret_cast<const f32x4*>(s2r + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float p = pv[j], a = av[j], b2 = bv[j];
            update(gv[j], p, a, b2);
            pv[j] = p; av[j] = a; bv[j] = b2;
        }
        *reinterpret_cast<f32x4*>(w + 4 * i) = pv;
        *reinterpret_cast<f32x4*>(s1 + 4 * i) = av;
        if (kind != OPT_RMSPROP) *reinterpret_cast<f32x4*>(s2 + 4 * i) = bv;
    }
    for (long i = 4 * n4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float p = w[i], a = s1[i], b2 = kind != OPT_RMSPROP ? s2[i] : 0.f;
        update(g[i], p, a, b2);
        w[i] = p;
        s1[i] = a;
        if (kind != OPT_RMSPROP) s2[i] = b2;
    }
}
int k_opt_step(float* w, const float* g, float* s1, float* s2, long n, int kind, float lr, float max_norm,
               const float* partials, int n_partials, float grad_scale, int step_t, const uint32_t* t_off, hipStream_t st) {
    opt_step_k<<<nblocks((n + 3) / 4, TPB, 8192), TPB, 0, st>>>(w, g, s1, s2, n, kind, lr, max_norm, partials, n_partials, grad_scale, step_t, t_off);
    GG_LAUNCH_CHECK();
}

}  // namespace gg
