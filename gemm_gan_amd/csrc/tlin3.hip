// Split-operand ("bf16x3") token-on-lane Linear: the precision mode GG_PREC_BF16X3.
//
//   Y[M,N] = epi( X[M,K] W[N,K]^T ),   X, W, Y fp32 in memory
//
// Every fp32 operand value v is split on its way into LDS into  hi = bf16(v)  and  lo = bf16(v - hi)  (v = hi + lo up to
// 2^-17 relative), and each product tile is three bf16 MFMAs accumulated in fp32:
//
//   X W^T  ~=  Xhi Whi^T + Xlo Whi^T + Xhi Wlo^T        (the dropped Xlo Wlo^T term is 2^-16 relative)
//
// i.e. fp32-grade products (measured 2e-6 .. 1e-5 relative against float64) at 3/16 of the bf16 MFMA rate instead of the 1/16
// of the fp32-input MFMA - on the SAME kernel structure as tlin_res_kernel (tlin.hip): a wave owns 32 tokens whose activations
// are register-resident MFMA B fragments, the weights stream through a double-buffered LDS chunk of 32 output features shared
// by the four waves, accumulators have features in registers and tokens on lanes, so bias, ReLU, dropout (same counter-hash
// stream, same element index), the ReLU gate of a backward product, +=, the residual add and LayerNorm are in-lane epilogue
// code.  Differences: K slices of 128 (two fragment sets per slice), weights read as fp32 from the master copy (no shadow),
// column groups of <= 256 features in grid.y for the wide products (QKV 3 x 256, FFN1 2 x 256), run-time epilogue flags.
// The engine's 1e-3 parity tests (tests/test_engine_golden_gpu.py, tests/test_engine_oracle_gpu.py) run in this mode.
//
// NS = 2 (above; 3 products): 2^-18 per operand, what the BACKWARD products run on.  NS = 3 splits v = hi + mid + lo (24 bits) and
// adds the products down to 2^-16 of the leading one - hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi (6 products): fp32-grade
// results (1e-7), which the FORWARD products need: a ReLU pre-activation that lands on the other side of zero switches that
// unit's whole gradient contribution, and with 2^-18 operands that happens ~64 x as often as in fp32 (measured: one FFN gate
// in ~10^6; visible in a small test as one weight-gradient row off by a token's contribution, harmless in training).
#include "kernels.h"
#include "drop_rng.h"
#include <hip/hip_ext.h>
#include <algorithm>

namespace gg {
namespace {
enum { EM_NONE = 0, EM_RES = 1, EM_ACC = 2, EM_MREF = 3, EM_ANY = 4 };      // epilogue operand modes of tlin3_kernel

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr float LN_EPS = 1e-5f;
// K slice per pass: 64 (12 / 24 MFMAs per 32-feature chunk with two / three operand parts); the next slice's activations are in flight meanwhile
template <int NS> struct Slice { static constexpr int KSL = 64, WLD = KSL + 8; };

// v -> (hi, lo) pairs of two values, packed as bf16x2 words
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    const __bf16 ah = (__bf16)a, bh = (__bf16)b;
    const bf16x2_t h = {ah, bh};
    const bf16x2_t l = {(__bf16)(a - (float)ah), (__bf16)(b - (float)bh)};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
    unsigned h0, l0, h1, l1;
    split2(v[0], v[1], h0, l0);
    split2(v[2], v[3], h1, l1);
    hi = u32x2{h0, h1};
    lo = u32x2{l0, l1};
}

// NS-way split of four values into bf16x2 words: part 0 = hi, 1 = next 8 bits, 2 = next 8 bits
template <int NS>
__device__ __forceinline__ void splitn(const f32x4 v, u32x2 (&out)[NS]) {
    f32x4 r = v;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const __bf16 b0 = (__bf16)r[0], b1 = (__bf16)r[1], b2 = (__bf16)r[2], b3 = (__bf16)r[3];
        const bf16x2_t a = {b0, b1}, b = {b2, b3};
        out[s] = u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
        if (s + 1 < NS) r = f32x4{r[0] - (float)b0, r[1] - (float)b1, r[2] - (float)b2, r[3] - (float)b3};
    }
}

// rows [tok0, tok0 + 32) x [k0, k0 + KSL) of X (fp32, FiLM optional) on their way into the wave-private slabs as NS bf16 images:
// request() issues every load of the slice into registers, commit() (one K slice of MFMAs later: the loads have landed, nothing
// waits on HBM) modulates, splits and writes them to LDS.  Without the split into two phases each slice exposed a full memory
// round trip at one wave per SIMD (the tile loop ran at 23 % of the MFMA rate).
template <int NS>
struct XStage {
    static constexpr int KSL = Slice<NS>::KSL, WLD = Slice<NS>::WLD;
    static constexpr int LPR = KSL / 4, RPI = 64 / LPR, NLD = 32 / RPI;
    f32x4 v[NLD];
    f32x4 ga, ba, gb, bb;
    int tb, rem0;
    __device__ __forceinline__ void request(const TlinP& p, int tok0, int last_tok, int k0, int lane) {
        asm volatile("" : "+v"(tok0));
        const unsigned char* const Xc = reinterpret_cast<const unsigned char*>(p.X);
        const int lrow = lane / LPR, lcol = 4 * (lane % LPR);
        const unsigned ldb = (unsigned)p.ldx * 4u, cb = (unsigned)(k0 + lcol) * 4u;
        if (p.film_g) {         // FiLM rows are per sample (film_group >= 32 tokens): a 32-token tile touches at most two of them
            tb = min(tok0, last_tok);
            const int g0 = tb / p.film_group;
            rem0 = tb - g0 * p.film_group;
            const int g1 = min(g0 + 1, last_tok / p.film_group);
            const unsigned fldb = (unsigned)p.film_ld * 4u;
            const unsigned char* const Gc = reinterpret_cast<const unsigned char*>(p.film_g) + cb;
            const unsigned char* const Bc = reinterpret_cast<const unsigned char*>(p.film_b) + cb;
            ga = *reinterpret_cast<const f32x4*>(Gc + (unsigned)g0 * fldb); ba = *reinterpret_cast<const f32x4*>(Bc + (unsigned)g0 * fldb);
            gb = *reinterpret_cast<const f32x4*>(Gc + (unsigned)g1 * fldb); bb = *reinterpret_cast<const f32x4*>(Bc + (unsigned)g1 * fldb);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) v[i] = *reinterpret_cast<const f32x4*>(Xc + ((unsigned)min(tok0 + lrow + RPI * i, last_tok) * ldb + cb));
    }
    __device__ __forceinline__ void commit(const TlinP& p, __bf16* xs, int tok0, int last_tok, int lane) const {     // xs: NS images, 4 * 32 * WLD apart
        const int lrow = lane / LPR, lcol = 4 * (lane % LPR);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            f32x4 m = v[i];
            if (p.film_g) {
                const int r = min(tok0 + lrow + RPI * i, last_tok);
                const bool second = rem0 + (r - tb) >= p.film_group;
                m = (second ? gb : ga) * m + (second ? bb : ba);
            }
            u32x2 parts[NS];
            splitn<NS>(m, parts);
            const int o = (RPI * i + lrow) * WLD + lcol;
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) *reinterpret_cast<u32x2*>(&xs[sp * (4 * 32 * WLD) + o]) = parts[sp];
        }
    }
};

// OCC: workgroups per CU the register budget is cut for - 128-column groups (NT_RES = 4) with <= 256 registers: the two-part kernel has
// 55 KB of LDS, the three-part one 70 KB with ONE weight buffer (WB = 1: a second barrier per chunk instead of the second image).
// In that form the column group is the fast grid index, so that the workgroups that read the same token rows run together (L2).
template <int NT_RES, int NS, int OCC = 1, int WB = 2>
__global__ __launch_bounds__(256, OCC) void tlin3_kernel(const TlinP p) {
    const DropKey dkey = drop_live(p.drop);
    constexpr int KSL = Slice<NS>::KSL, WLD = Slice<NS>::WLD;
    constexpr int N = 32 * NT_RES;                     // columns of this workgroup's group
    static_assert(KSL == 64, "weight chunk staging: 32 rows x 8 pieces = 256 threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WIMG = WB * 32 * WLD, XIMG = 4 * 32 * WLD;                // elements per weight / activation image (one split part)
    __bf16* const Wsp = reinterpret_cast<__bf16*>(smem_raw);                // [NS][WB][32*WLD]
    __bf16* const Xsp = Wsp + NS * WIMG;                                    // [NS][4][32*WLD]
    float* const Ps = reinterpret_cast<float*>(Xsp + NS * XIMG);            // bias | gamma | beta  [3][N]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    const int gcol = (int)(OCC == 2 ? blockIdx.x : blockIdx.y) * N;
    const int tok0 = (int)(OCC == 2 ? blockIdx.y : blockIdx.x) * 128 + wave * 32;
    const int last_tok = (int)p.M - 1;
    const int nks = p.K / KSL;
    const int nchunks = nks * NT_RES;
    __bf16* const xs = Xsp + wave * 32 * WLD;          // this wave's slab of part 0; part sp at + sp * XIMG

    // weight chunks (32 output features x KSL) travel L2 -> registers -> LDS two chunks ahead of their use.  The weights arrive
    // PRE-SPLIT: NS bf16 images [N][K], w_part_stride elements apart (k_split_weights; the engine refreshes them with the other
    // shadow copies) - splitting fp32 weights here cost ~70 VALU instructions per thread and chunk, outside the MFMA shadow at
    // one wave per SIMD
    u32x4 wreg[2][NS];
    const __bf16* Wp = reinterpret_cast<const __bf16*>(p.W);
    const int wrow = tid >> 3, wpiece = tid & 7;        // 32 rows x 8 pieces of 8 bf16
    auto load_chunk = [&](int set, int ks, int nt) {
        const __bf16* src = Wp + (long)(gcol + nt * 32 + wrow) * p.ldw + ks * KSL + 8 * wpiece;
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) wreg[set][sp] = *reinterpret_cast<const u32x4*>(src + (long)sp * p.w_part_stride);
    };
    auto store_chunk = [&](int set) {        // register set `set` -> weight buffer `set` (WB = 2) / the one buffer (WB = 1)
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) *reinterpret_cast<u32x4*>(&Wsp[sp * WIMG + (WB == 2 ? set : 0) * 32 * WLD + wrow * WLD + 8 * wpiece]) = wreg[set][sp];
    };

    XStage<NS> xst;
    xst.request(p, tok0, last_tok, 0, lane);
    load_chunk(0, 0, 0);
    load_chunk(1, 0, 1);
    for (int i = tid; i < N; i += 256) {
        Ps[i] = p.bias ? p.bias[gcol + i] : 0.f;
        Ps[N + i] = p.ln_g ? p.ln_g[i] : 1.f;
        Ps[2 * N + i] = p.ln_g ? p.ln_b[i] : 0.f;
    }
    store_chunk(0);

    const int tok = tok0 + c;
    const bool valid = tok <= last_tok;
    const int tokc = valid ? tok : last_tok;           // clamped lanes recompute the last row; their stores are predicated off
    const long yrow = p.y_row_group ? (long)tokc + tokc / p.y_row_group + 1 : (long)tokc;
    float* const yb = reinterpret_cast<float*>(p.Y) + yrow * p.ldy + gcol;
    const float* const resp = p.res ? p.res + (long)(tokc % (int)p.res_rows) * p.ldres + gcol : nullptr;
    const float* const mref = p.mask_ref ? reinterpret_cast<const float*>(p.mask_ref) + (long)tokc * p.ldref + gcol : nullptr;
    const bool keep_y = p.y_rows < 0 || tokc < p.y_rows;
    const bool has_pre = resp != nullptr || p.accumulate != 0;

    f32x16 acc[NT_RES];
    bf16x8 xf[NS][KSL / 16];
    // epilogue operands (residual rows / previous output, gate reference) travel one feature tile ahead of their use: tile 0 is
    // requested before the last MFMA pass, tile nt + 1 while tile nt is finished (all of them up front cost 128 registers: spills)
    // The requests carry NO run-time condition: which operands exist is a compile-time mode of the epilogue (EM_*; the kernel
    // branches once, wave-uniformly, into the matching copy).  With `if (resp) load` inside the loop the wait-count pass waited
    // for every request where it was issued (vmcnt(0)): the one-tile-ahead prefetch did not exist.
    f32x4 pre[2][4], mrf[2][4];
    auto load_pre = [&](int nt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nt * 32 + 8 * g + 4 * h;
            if constexpr (MODE == EM_RES) pre[nt & 1][g] = *reinterpret_cast<const f32x4*>(resp + n);
            if constexpr (MODE == EM_ACC) pre[nt & 1][g] = *reinterpret_cast<const f32x4*>(yb + n);
            if constexpr (MODE == EM_MREF) mrf[nt & 1][g] = *reinterpret_cast<const f32x4*>(mref + n);
            if constexpr (MODE == EM_ANY) {
                if (has_pre) {
                    pre[nt & 1][g] = resp ? *reinterpret_cast<const f32x4*>(resp + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p.accumulate) pre[nt & 1][g] += *reinterpret_cast<const f32x4*>(yb + n);
                }
                if (mref) mrf[nt & 1][g] = *reinterpret_cast<const f32x4*>(mref + n);
            }
        }
    };
    const int emode = (!has_pre && !mref) ? EM_NONE : (resp && !p.accumulate && !mref) ? EM_RES : (!resp && p.accumulate && !mref) ? EM_ACC
                      : (!has_pre && mref) ? EM_MREF : EM_ANY;

    int chunk = 0;
    for (int ks = 0; ks < nks; ++ks) {
        xst.commit(p, xs, tok0, last_tok, lane);
        if (ks + 1 < nks) xst.request(p, tok0, last_tok, (ks + 1) * KSL, lane);
        if (ks == nks - 1) {      // (outside the steady state: the branch costs one conservative wait in the last slice)
            if (emode == EM_RES) load_pre(0, std::integral_constant<int, EM_RES>());
            else if (emode == EM_ACC) load_pre(0, std::integral_constant<int, EM_ACC>());
            else if (emode == EM_MREF) load_pre(0, std::integral_constant<int, EM_MREF>());
            else if (emode == EM_ANY) load_pre(0, std::integral_constant<int, EM_ANY>());
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < KSL / 16; ++s)
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) xf[sp][s] = *reinterpret_cast<const bf16x8*>(&xs[sp * XIMG + c * WLD + 16 * s + 8 * h]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __syncthreads();     // chunk `chunk` is in W?[chunk & 1] (and Ps on the first pass)

#pragma unroll
        for (int nt = 0; nt < NT_RES; ++nt) {
            const int buf = nt & 1;      // compile-time after unrolling (NT_RES is even): chunk & 1 == nt & 1
            // unconditional (the k-slice index clamped; the last two requests fetch chunks nobody uses): a run-time condition around
            // these loads makes the wait-count pass drain them at the end of the SAME chunk (vmcnt(0)) instead of one chunk later
            load_chunk(buf, min(nt + 2 >= NT_RES ? ks + 1 : ks, nks - 1), (nt + 2) % NT_RES);
            if (ks == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
            }
            const __bf16* wb0 = Wsp + (WB == 2 ? buf : 0) * 32 * WLD + c * WLD + 8 * h;
            // the fragments of k-step s + 1 are requested before the MFMAs of k-step s: one exposed LDS round trip per chunk, not per step
            bf16x8 wf[2][NS];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) wf[0][sp] = *reinterpret_cast<const bf16x8*>(wb0 + sp * WIMG);
#pragma unroll
            for (int s = 0; s < KSL / 16; ++s) {
                if (s + 1 < KSL / 16) {
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) wf[(s + 1) & 1][sp] = *reinterpret_cast<const bf16x8*>(wb0 + sp * WIMG + 16 * (s + 1));
                }
                const bf16x8(&w)[NS] = wf[s & 1];
                // small terms first, the leading product last
                if constexpr (NS == 3) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], xf[0][s], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], xf[2][s], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], xf[1][s], acc[nt], 0, 0, 0);
                }
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], xf[0][s], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], xf[1][s], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], xf[0][s], acc[nt], 0, 0, 0);
            }
            if constexpr (WB == 1) __syncthreads();      // every wave has read this chunk: the one buffer takes the next
            store_chunk(buf ^ 1);          // (after the last chunk: a buffer nobody reads)
            __syncthreads();
            ++chunk;
        }
    }

    // ---- epilogue: bias -> ReLU -> dropout -> gate -> (+ previous / residual) -> store ; LayerNorm over the N features ----------
    const bool drop_on = p.drop.p > 0.f;
    const float ksd = drop_on ? 1.f / (1.f - p.drop.p) : 1.f;
    const uint64_t dbase = (uint64_t)tokc * p.drop_ld + gcol;
    float sum = 0.f;
    auto epilogue = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
    for (int nt = 0; nt < NT_RES; ++nt) {
        if (nt + 1 < NT_RES) load_pre(nt + 1, mode_tag);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nt * 32 + 8 * g + 4 * h;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[n]);
            f32x4 v = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
            v += bb;
            if (p.act_relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (drop_on) {
                float f[4];
                drop_factor4(dkey, dbase + n, ksd, f);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= f[j];
            }
            if (MODE == EM_MREF || (MODE == EM_ANY && mref)) {
                const f32x4 m = mrf[nt & 1][g];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = m[j] > 0.f ? v[j] * p.mask_scale : 0.f;
            }
            if (MODE == EM_RES || MODE == EM_ACC || (MODE == EM_ANY && has_pre)) v += pre[nt & 1][g];
            if (valid && keep_y) *reinterpret_cast<f32x4*>(yb + n) = v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[nt][4 * g + j] = v[j];
                sum += v[j];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    };
    if (emode == EM_NONE) epilogue(std::integral_constant<int, EM_NONE>());
    else if (emode == EM_RES) epilogue(std::integral_constant<int, EM_RES>());
    else if (emode == EM_ACC) epilogue(std::integral_constant<int, EM_ACC>());
    else if (emode == EM_MREF) epilogue(std::integral_constant<int, EM_MREF>());
    else epilogue(std::integral_constant<int, EM_ANY>());
    if (p.ln_g) {
        const float invn = 1.f / (float)N;
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * invn;
        float var = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT_RES; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float d = acc[nt][i] - mean;
                var += d * d;
            }
        var += __shfl_xor(var, 32, 64);
        const float rstd = rsqrtf(var * invn + LN_EPS);
        float* const lb = p.ln_y + (long)tokc * p.ldy;
#pragma unroll
        for (int nt = 0; nt < NT_RES; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nt * 32 + 8 * g + 4 * h;
                const f32x4 gg_ = *reinterpret_cast<const f32x4*>(&Ps[N + n]);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[2 * N + n]);
                f32x4 y;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = (acc[nt][4 * g + j] - mean) * rstd * gg_[j] + bb[j];
                if (valid) *reinterpret_cast<f32x4*>(lb + n) = y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (h == 0 && valid && keep_y) {
            p.ln_stats[2 * (long)tok] = mean;
            p.ln_stats[2 * (long)tok + 1] = rstd;
        }
    }
}

hipEvent_t g3_ev0 = nullptr, g3_ev1 = nullptr;

template <int NT_RES, int NS, int OCC = 1, int WB = 2>
int launch3(const TlinP& p, int groups, hipStream_t st) {
    constexpr size_t smem = (size_t)NS * (WB + 4) * 32 * Slice<NS>::WLD * 2 + (size_t)3 * 32 * NT_RES * 4;
    static_assert(OCC == 1 || 2 * smem <= 160 * 1024, "two workgroups per CU");
    static_assert(smem <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tlin3_kernel<NT_RES, NS, OCC, WB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const dim3 grid = OCC == 2 ? dim3((unsigned)groups, (unsigned)((p.M + 127) / 128)) : dim3((unsigned)((p.M + 127) / 128), (unsigned)groups);
    if (g3_ev0) {
        hipExtLaunchKernelGGL((tlin3_kernel<NT_RES, NS, OCC, WB>), grid, dim3(256), (unsigned)smem, st, g3_ev0, g3_ev1, 0, p);
        g3_ev0 = g3_ev1 = nullptr;
    } else {
        hipLaunchKernelGGL((tlin3_kernel<NT_RES, NS, OCC, WB>), grid, dim3(256), smem, st, p);
    }
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
}  // namespace

void tlin3_time_next(hipEvent_t begin, hipEvent_t end) { g3_ev0 = begin; g3_ev1 = end; }

// fp32 X / Y (and fp32 gate reference), W as pre-split bf16 part images (k_split_weights: p.W = part 0 [N][K], part sp at + sp * p.w_part_stride
// elements); N a multiple of 32 up to 256, or a multiple of 256 (column groups); K a multiple of 128
bool tlin3_supported(const TlinP& p) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (double)p.M * (double)p.ldx * 4.0 >= 4.0e9) return false;      // 32-bit row offsets
    if (p.x_bf16 || p.y_bf16 || p.fp8 || (p.mask_ref && p.mask_bf16)) return false;
    if (p.K % 128) return false;
    if (!(p.N == 64 || p.N == 128 || p.N == 256 || (p.N > 256 && p.N % 256 == 0))) return false;
    if (p.ln_g && (p.N > 256 || !p.ln_b || !p.ln_y || !p.ln_stats || !al16(p.ln_y))) return false;
    if (!al16(p.X) || !al16(p.W) || !al16(p.Y) || p.ldx % 4 || p.ldy % 4 || p.ldw % 8 || p.w_part_stride % 8 || p.w_part_stride <= 0) return false;
    if (p.film_g && (!al16(p.film_g) || !al16(p.film_b) || p.film_ld % 4 || p.film_group < 32)) return false;
    if (p.mask_ref && (!al16(p.mask_ref) || p.ldref % 4)) return false;
    if (p.res && (!al16(p.res) || p.ldres % 4 || p.res_rows < 1)) return false;
    if (p.drop.p > 0.f && p.drop_ld % 2) return false;
    return true;
}

int tlin3(const TlinP& p, hipStream_t st, int nsplit) {
    GG_REQUIRE(tlin3_supported(p), "tlin3: unsupported shape / alignment");
    GG_REQUIRE(nsplit == 2 || nsplit == 3, "tlin3: 2 (hi, lo: 3 products) or 3 (hi, mid, lo: 6 products) operand parts");
    if (nsplit == 3) {
        if (p.N == 64) return launch3<2, 3>(p, 1, st);
        if (p.N == 128) return launch3<4, 3>(p, 1, st);
        // calls without a LayerNorm epilogue (QKV, FFN1, patch / text encoders): 128-column groups, two workgroups per CU, one weight buffer
        // (cfg3 bf16x3 step, interleaved: 65.8 -> 62.9 ms; 56 bytes of spills included).  GG_TLIN3_FWD_OCC1: the one-workgroup form
        static const bool occ2f = getenv("GG_TLIN3_FWD_OCC1") == nullptr;
        if (occ2f && !p.ln_g && (p.M + 127) / 128 <= 65535) return launch3<4, 3, 2, 1>(p, p.N / 128, st);
        return launch3<8, 3>(p, p.N / 256, st);
    }
    if (p.N == 64) return launch3<2, 2>(p, 1, st);
    if (p.N == 128) return launch3<4, 2>(p, 1, st);
    // calls without a LayerNorm epilogue as 128-column groups, two workgroups per CU (cfg3 bf16x3 step, interleaved: 68.3 -> 66.5 ms);
    // GG_TLIN3_OCC1: the one-workgroup form
    static const bool occ2 = getenv("GG_TLIN3_OCC1") == nullptr;
    if (occ2 && !p.ln_g && (p.M + 127) / 128 <= 65535) return launch3<4, 2, 2>(p, p.N / 128, st);
    return launch3<8, 2>(p, p.N / 256, st);
}

namespace {
// parts[sp][i] = sp-th bf16 part of w[i] (hi, then the bf16 roundings of the successive remainders); optionally the same for the
// transpose of every 2-D tensor of a shadow table (tab == nullptr: one flat array of n elements, no transpose)
__global__ void split_flat_kernel(const float* __restrict__ w, __bf16* __restrict__ parts, long n, long part_stride, int nparts) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float r = w[i];
        for (int sp = 0; sp < nparts; ++sp) {
            const __bf16 b = (__bf16)r;
            parts[sp * part_stride + i] = b;
            r -= (float)b;
        }
    }
}
__global__ void split_tab_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, __bf16* __restrict__ wtp, long part_stride,
                                 const ShadowEntry* __restrict__ tab) {
    __shared__ float t[32][33];
    const ShadowEntry e = tab[blockIdx.y];
    const int tiles_c = (e.cols + 31) / 32, ntile = ((e.rows + 31) / 32) * tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int r0 = (tile / tiles_c) * 32, c0 = (tile % tiles_c) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            const bool in = r < e.rows && c < e.cols;
            float v = in ? w[e.off + (long)r * e.cols + c] : 0.f;
            t[ty + 8 * k][tx] = v;
            if (in) {
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    const __bf16 b = (__bf16)v;
                    wp[sp * part_stride + e.off + (long)r * e.cols + c] = b;
                    v -= (float)b;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < e.rows && c < e.cols) {
                float v = t[tx][ty + 8 * k];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    const __bf16 b = (__bf16)v;
                    wtp[sp * part_stride + e.off + (long)c * e.rows + r] = b;
                    v -= (float)b;
                }
            }
        }
        __syncthreads();
    }
}
}  // namespace

int k_split_weights(const float* w, void* parts, long n, long part_stride, int nparts, hipStream_t st) {
    if (n <= 0) return 0;
    const long blocks = std::min<long>((n + 255) / 256, 2048);
    split_flat_kernel<<<(unsigned)blocks, 256, 0, st>>>(w, reinterpret_cast<__bf16*>(parts), n, part_stride, nparts);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
int k_shadow_parts(const float* w, void* wp, void* wtp, long part_stride, const ShadowEntry* tab_dev, int n_entries, hipStream_t st) {
    if (n_entries <= 0) return 0;
    split_tab_kernel<<<dim3(64, n_entries), 256, 0, st>>>(w, reinterpret_cast<__bf16*>(wp), reinterpret_cast<__bf16*>(wtp), part_stride, tab_dev);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gg
