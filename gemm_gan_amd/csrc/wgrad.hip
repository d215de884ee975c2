// Weight-gradient kernel:  dW[N,K] += dY[M,N]^T . X[M,K]   (reduction over the M ~ 1e5 token rows, bf16 MFMA)
//
// Both operands are stored token-major (the reduction index is the SLOW index of both), which is the worst
// case for an MFMA fragment (8 consecutive reduction elements per lane).  Instead of transposing while
// staging, token chunks are copied row-major into LDS with full-line loads and BOTH operands are fetched
// with ds_read_b64_tr_b16 (hardware transpose read: a 16-lane group reads a 4-token x 16-feature block and
// each lane receives one feature column of 4 tokens).
//
// Decomposition: a workgroup (4 waves) owns a 128 (n) x 256 (k) panel of dW - wave w the 32-row slice
// n in [32w, 32w+32) as 8 MFMA 32x32 accumulators (128 accumulator registers) - and a contiguous range
// of tokens; panels x token-splits fill the chip.  Per 32-token chunk a wave issues 2 x 2 A-fragment and
// 8 x 2 B-fragment transpose reads for 32 MFMAs.  Partial panels are added to dW with fp32 atomics whose wave
// instructions cover two 128-byte row segments (the fast atomic shape).
#include "kernels.h"

namespace gg {

namespace {
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// bf16x3 mode (X3): an fp32 value v is staged as hi = bf16(v) and lo = bf16(v - hi) in two LDS images; a product tile is
// hi*hi + lo*hi + hi*lo (fp32-grade: 2^-16 relative), see tlin3.hip
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
    const __bf16 h0 = (__bf16)v[0], h1 = (__bf16)v[1], h2 = (__bf16)v[2], h3 = (__bf16)v[3];
    const bf16x2_t a = {h0, h1}, b = {h2, h3};
    const bf16x2_t c = {(__bf16)(v[0] - (float)h0), (__bf16)(v[1] - (float)h1)}, d = {(__bf16)(v[2] - (float)h2), (__bf16)(v[3] - (float)h3)};
    hi = u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
    lo = u32x2{__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d)};
}

// (tools/wgrad_probe.hip rebuilds this file with other values; measured alternatives are listed in DESIGN.md section 3)
#ifndef GG_WG_WT
#define GG_WG_WT 1
#endif
#ifndef GG_WG_DEPTH
#define GG_WG_DEPTH 4
#endif
// workgroups per launch (panels x token splits): one per CU.  Fewer (a share of the CUs left to the main chain, the kernel runs on
// side streams) measured 0.3 ms per step faster while the persistent Linears held ALL CUs (192 / 208: 28.75 against 29.06 ms);
// since those leave 9 % of the CUs free themselves (wst.hip) the difference is within the A/B noise (192: 28.44, 224: 28.42,
// 256: 28.56 ms) and the launch alone is 8 % slower at 192.
#ifndef GG_WG_TARGET
#define GG_WG_TARGET 256
#endif
#ifndef GG_WG_T22
#define GG_WG_T22 0
#endif
constexpr int WT = GG_WG_WT;                     // 32-row accumulator tiles per wave along n
#ifndef GG_WG_CT
#define GG_WG_CT 32
#endif
constexpr int PN = 128 * WT, PK = 256, CT = GG_WG_CT;  // panel rows / cols, tokens per chunk
constexpr int LDY = PN + 32, LDX = PK + 32;      // bf16 per LDS row: 144 dwords (16 mod 64) -> conflict-free transpose reads

// A / B fragment of the 16-token step s2 for the 32 features starting at col0: element j of lane (c, h) is
// token 16*s2 + 8*h + j, feature col0 + c   (natural k order: both operands come from LDS)
__device__ __forceinline__ bf16x8 frag_tr(const __bf16* img, int ld, int col0, int s2, int lane) {
    const int i = lane & 15, grp = lane >> 4;
    const int hh = grp >> 1, colhalf = grp & 1;
    const __bf16* p0 = img + (16 * s2 + 8 * hh + (i >> 2)) * ld + col0 + 16 * colhalf + 4 * (i & 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * ld));
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

// global -> registers -> LDS staging of one [CT tokens x W features] chunk (W = 256), 4 x 16-byte pieces / thread
template <bool BF, int W, int NT = 256>
struct Stager {
    static constexpr int PPR = W / (BF ? 8 : 4);             // 16-byte pieces per row
    static constexpr int NP = CT * PPR / NT;                 // pieces per thread
    u32x4 r[NP];
    f32x4 fga, fba, fgb, fbb;      // FiLM (fp32 X only): gamma / beta of the two samples a 32-token chunk can touch, this thread's columns
    int fsplit;                    // first row of the chunk that belongs to the second sample
    // Unconditional loads (rows / columns past the end are clamped to valid addresses so nothing branches and all
    // pieces are in flight together); `zero` blanks the out-of-range pieces afterwards - needed for dY only: a zero
    // dY row/column contributes nothing whatever X holds, and out-of-range X columns only feed unstored outputs.
    // mod > 0: the operand holds `mod` rows that repeat (replicas sharing one input): row t is read at t % mod
    __device__ __forceinline__ void load(const void* base, long ld, long tok0, long tok_end, int col0, int cols_valid, int tid, bool zero,
                                         long mod = 0) {
        if constexpr (BF) {
            const __bf16* p = reinterpret_cast<const __bf16*>(base);
#pragma unroll
            for (int i = 0; i < NP; ++i) {                        // piece = 8 bf16
                const int f = tid + NT * i, row = f / PPR, pc = f % PPR;
                long t = min(tok0 + row, tok_end - 1);
                if (mod > 0) t %= mod;
                const int cc = min(8 * pc, cols_valid - 8);
                r[i] = *reinterpret_cast<const u32x4*>(p + t * ld + col0 + cc);
                if (zero && (tok0 + row >= tok_end || 8 * pc >= cols_valid)) r[i] = u32x4{0u, 0u, 0u, 0u};
            }
        } else {
            const float* p = reinterpret_cast<const float*>(base);
#pragma unroll
            for (int i = 0; i < NP; ++i) {                        // piece = 4 fp32
                const int f = tid + NT * i, row = f / PPR, pc = f % PPR;
                long t = min(tok0 + row, tok_end - 1);
                if (mod > 0) t %= mod;
                const int cc = min(4 * pc, cols_valid - 4);
                r[i] = *reinterpret_cast<const u32x4*>(p + t * ld + col0 + cc);
                if (zero && (tok0 + row >= tok_end || 4 * pc >= cols_valid)) r[i] = u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
    // ---- fast path (whole chunks only): the chunk's first row is a UNIFORM address, the thread's pieces sit at chunk-local
    // 32-bit byte offsets that never change (offsets(), once per kernel).  A load is then ONE instruction: no per-load 64-bit
    // row * ld products, clamps, modulo or zeroing branches.  With the general path above the compiler emitted ~180
    // instructions per load and - because an out-of-range piece was zeroed by overwriting the register the load targets -
    // a wait for each load right after it was issued, which collapsed the DEPTH-deep prefetch ring to a depth of one:
    // the kernel ran at memory LATENCY (measured: removing every MFMA changed its time by 5 %).  Out-of-range columns
    // are blanked when the piece is written to LDS instead (store(.., zmask)).
    static constexpr int ES = BF ? 2 : 4, EPP = BF ? 8 : 4;          // bytes per element, elements per piece
    __device__ __forceinline__ static void offsets(unsigned (&voff)[NP], unsigned& zmask, long ld, int col0, int cols_valid, int tid) {
        zmask = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int f = tid + NT * i, row = f / PPR, pc = f % PPR;
            const int cc = min(EPP * pc, cols_valid - EPP);
            voff[i] = (unsigned)((row * ld + col0 + cc) * ES);
            if (EPP * pc >= cols_valid) zmask |= 1u << i;
        }
    }
    __device__ __forceinline__ void load_fast(const char* __restrict__ chunk_base, const unsigned (&voff)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) r[i] = *reinterpret_cast<const u32x4*>(chunk_base + voff[i]);
    }
    // X' = gamma[sample] * X + beta[sample], sample = token / group: requested together with the chunk's rows
    __device__ __forceinline__ void load_film(const float* fg, const float* fb, long fld, int group, long tok0, long tok_end, int col0,
                                              int cols_valid, int tid) {
        const int pc = tid % PPR;
        const int cc = min(4 * pc, cols_valid - 4);
        const long t0 = min(tok0, tok_end - 1);
        const long g0 = t0 / group, g1 = min(g0 + 1, (tok_end - 1) / group);
        fsplit = (int)((g0 + 1) * group - tok0);
        fga = *reinterpret_cast<const f32x4*>(fg + g0 * fld + col0 + cc);
        fba = *reinterpret_cast<const f32x4*>(fb + g0 * fld + col0 + cc);
        fgb = *reinterpret_cast<const f32x4*>(fg + g1 * fld + col0 + cc);
        fbb = *reinterpret_cast<const f32x4*>(fb + g1 * fld + col0 + cc);
    }
    __device__ __forceinline__ void store_film(__bf16* img, int ld, int tid, __bf16* img_lo = nullptr) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int f = tid + NT * i, row = f / PPR, pc = f % PPR;
            const bool second = row >= fsplit;
            const f32x4 v = (second ? fgb : fga) * __builtin_bit_cast(f32x4, r[i]) + (second ? fbb : fba);
            if (img_lo) {
                u32x2 hi, lo;
                split4(v, hi, lo);
                *reinterpret_cast<u32x2*>(img + row * ld + 4 * pc) = hi;
                *reinterpret_cast<u32x2*>(img_lo + row * ld + 4 * pc) = lo;
            } else {
                u32x2 w = {pack2(v[0], v[1]), pack2(v[2], v[3])};
                *reinterpret_cast<u32x2*>(img + row * ld + 4 * pc) = w;
            }
        }
    }
    __device__ __forceinline__ void store(__bf16* img, int ld, int tid, unsigned zmask = 0, __bf16* img_lo = nullptr) const {
        if constexpr (BF) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int f = tid + NT * i, row = f / PPR, pc = f % PPR;
                *reinterpret_cast<u32x4*>(img + row * ld + 8 * pc) = ((zmask >> i) & 1u) ? u32x4{0u, 0u, 0u, 0u} : r[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int f = tid + NT * i, row = f / PPR, pc = f % PPR;
                const f32x4 v = ((zmask >> i) & 1u) ? f32x4{0.f, 0.f, 0.f, 0.f} : __builtin_bit_cast(f32x4, r[i]);
                if (img_lo) {
                    u32x2 hi, lo;
                    split4(v, hi, lo);
                    *reinterpret_cast<u32x2*>(img + row * ld + 4 * pc) = hi;
                    *reinterpret_cast<u32x2*>(img_lo + row * ld + 4 * pc) = lo;
                } else {
                    u32x2 w = {pack2(v[0], v[1]), pack2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(img + row * ld + 4 * pc) = w;
                }
            }
        }
    }
};

// FGRAD: instead of dW += panel, the panel C_b = dY_b^T X_b of ONE sample b (split = sample) is contracted with the
// weight matrix W [N,K] it belongs to:   dgamma[b,k] += sum_n W[n,k] C_b[n,k],   dbeta[b,k] += sum_n W[n,k] s_b[n],
// s_b[n] = sum_tokens dY_b[token,n]  - the gradients of a FiLM modulation X' = gamma_b * X + beta_b that sits in front of
// the Linear W, without materialising d(X') = dY W  (dgamma = sum_tokens dX' * X, dbeta = sum_tokens dX').
// W8: eight waves per workgroup (two per SIMD) on the same 128 x 256 panel - wave (wn, wk) owns 32 rows x 128 columns (4 accumulator
// tiles, half the staging registers per thread): while one wave of a SIMD waits for its LDS fragments the other multiplies.
template <bool YB, bool XB, bool FILM, bool FGRAD, bool FAST, bool W8 = false, bool X3 = false>
__global__ __launch_bounds__(W8 ? 512 : 256, W8 ? 2 : 1) void wgrad_kernel(const void* __restrict__ dY, long ldy, const void* __restrict__ X, long ldx,
                                                    float* __restrict__ dW, long ldw, long M, int N, int K, int splits, WgradFilm film,
                                                    WgradFilmGrad fg, float* __restrict__ dbias, long x_mod) {
    extern __shared__ __attribute__((aligned(16))) __bf16 wg_smem[];       // 67.6 KB: above the static-LDS limit
    static_assert(!X3 || (!YB && !XB), "bf16x3: fp32 operands");
    auto Ysb = [&](int b) { return wg_smem + b * (CT * LDY); };
    auto Xsb = [&](int b) { return wg_smem + 2 * (CT * LDY) + b * (CT * LDX); };
    // bf16x3: the lo images behind the hi ones (null otherwise: the stagers then write one image)
    auto Ysl = [&](int b) { return X3 ? wg_smem + 2 * CT * (LDY + LDX) + b * (CT * LDY) : (__bf16*)nullptr; };
    auto Xsl = [&](int b) { return X3 ? wg_smem + 2 * CT * (LDY + LDX) + 2 * (CT * LDY) + b * (CT * LDX) : (__bf16*)nullptr; };
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    const int panels_k = (K + PK - 1) / PK;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (b % 8), each with its own L2.  All panels of
    // one token split read the SAME rows of dY / X, so they are made neighbours on one XCD (consecutive b / 8) and
    // the second .. last panel of a split hit that XCD's L2 instead of fetching the rows again.
    const int npanels = panels_k * ((N + PN - 1) / PN);
    const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;
    const int split = (j / npanels) * 8 + xcd, panel = j % npanels;
    if (split >= splits) return;
    const int n0 = (panel / panels_k) * PN, k0 = (panel % panels_k) * PK;
    const int nvalid = min(PN, N - n0), kvalid = min(PK, K - k0);
    // token range of this split, in whole chunks
    const long chunks = (M + CT - 1) / CT;
    long c_beg = chunks * split / splits, c_end = chunks * (split + 1) / splits;
    if constexpr (FGRAD) {                              // split = sample: its tokens are whole chunks (tokens % CT == 0)
        c_beg = (long)split * (fg.tokens / CT);
        c_end = c_beg + fg.tokens / CT;
    }
    if (c_beg >= c_end) return;
    float ssum = 0.f;                                   // sum over tokens of dY[token, row 32*wave + c] (this lane's half of every 16)
    const bool do_bias = dbias != nullptr && k0 == 0;   // bias gradient = column sums of dY: once per row panel

    // Wave tiling of the panel.  4 x 1: wave w owns rows [32w, 32w+32) x all 256 columns (1 A fragment, 8 B fragments per
    // 16-token step: every wave reads the WHOLE X chunk from LDS).  2 x 2 (T22): wave (wn, wk) owns 64 rows x 128 columns
    // (2 A + 4 B fragments): a third fewer LDS fragment reads for the same 8 MFMAs - the LDS read stream is what bounds the
    // kernel once its loads are prefetched.  The FiLM-gradient epilogue is written for the 4 x 1 map.
    static_assert(!W8 || WT == 1, "eight-wave tiling: one 32-row tile per wave");
    constexpr int NT = W8 ? 512 : 256;
    constexpr bool T22 = GG_WG_T22 && !FGRAD && WT == 1 && !W8;
    constexpr int TA = T22 ? 2 : WT, TB = (T22 || W8) ? 4 : 8;
    const int nb = W8 ? (wave & 3) * 32 : T22 ? (wave >> 1) * 64 : wave * (32 * WT), kb = W8 ? (wave >> 2) * 128 : T22 ? (wave & 1) * 128 : 0;
    float ssum2 = 0.f;                                  // T22: column sums of the wave's second 32-row tile
    f32x16 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // Token chunks travel HBM -> registers -> LDS through a ring of DEPTH register stages: chunk r + DEPTH is
    // requested when chunk r is multiplied and written to LDS DEPTH - 1 chunks later, so an HBM round trip is
    // covered by DEPTH - 1 MFMA passes (one pass = 32 MFMAs, far shorter than the memory latency on its own).
#ifndef GG_WG_DEPTH_F32Y
#define GG_WG_DEPTH_F32Y 3
#endif
#ifndef GG_WG_DEPTH_FILM
#define GG_WG_DEPTH_FILM 2
#endif
    // ring depth by register budget (the branch-free loader keeps every stage live): a stage is 40 registers with bf16 dY,
    // 48 with fp32 dY, 64 with FiLM operands on top
#ifndef GG_WG_DEPTH_W8
#define GG_WG_DEPTH_W8 2
#endif
    constexpr int DEPTH = !FAST ? GG_WG_DEPTH : W8 ? GG_WG_DEPTH_W8 : FILM ? GG_WG_DEPTH_FILM : !YB ? GG_WG_DEPTH_F32Y : GG_WG_DEPTH;
    Stager<YB, PN, NT> sy[DEPTH];
    Stager<XB, PK, NT> sx[DEPTH];
    const long nch = c_end - c_beg;
#ifndef GG_WG_ABL
#define GG_WG_ABL 0          // tools/wgrad_probe.hip ablations: 1 = no fragment reads / MFMAs, 2 = no LDS writes of X, 4 = no atomic
                             // panel add (measured: the add costs 15 - 21 us of a 55 - 125 us launch; it scales with
                             // splits x N x K, and smaller panels / fewer splits pay more in L2 re-reads than they save)
#endif
    auto multiply = [&](int buf) {
#pragma unroll
        for (int s2 = 0; s2 < ((GG_WG_ABL & 1) ? 0 : CT / 16); ++s2) {
            bf16x8 af[TA], al[X3 ? TA : 1];
#pragma unroll
            for (int a = 0; a < TA; ++a) af[a] = frag_tr(Ysb(buf), LDY, nb + a * 32, s2, lane);
            if constexpr (X3) {
#pragma unroll
                for (int a = 0; a < TA; ++a) al[a] = frag_tr(Ysl(buf), LDY, nb + a * 32, s2, lane);
            }
            if (FGRAD || do_bias) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ssum += __builtin_bit_cast(float, (unsigned)(unsigned short)af[0][j] << 16);
                if constexpr (X3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ssum += __builtin_bit_cast(float, (unsigned)(unsigned short)al[0][j] << 16);
                }
                if constexpr (T22) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ssum2 += __builtin_bit_cast(float, (unsigned)(unsigned short)af[1][j] << 16);
                }
            }
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const bf16x8 bf = frag_tr(Xsb(buf), LDX, kb + b * 32, s2, lane);
                if constexpr (X3) {
                    const bf16x8 bl = frag_tr(Xsl(buf), LDX, kb + b * 32, s2, lane);
#pragma unroll
                    for (int a = 0; a < TA; ++a) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bf, acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bl, acc[a][b], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int a = 0; a < TA; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf, acc[a][b], 0, 0, 0);
            }
        }
    };
    if constexpr (FAST) {
        // whole chunks only (the host checks M % CT == 0 and x_mod % CT == 0): branch-free issue, see Stager::load_fast.
        // Issues past the last chunk repeat it (their data is written to LDS but never multiplied), so the loop body has
        // no conditional around a load and the compiler's wait counts are those of the steady state.
        unsigned voy[Stager<YB, PN, NT>::NP], vox[Stager<XB, PK, NT>::NP], zmy, zmx;
        Stager<YB, PN, NT>::offsets(voy, zmy, ldy, n0, nvalid, tid);
        Stager<XB, PK, NT>::offsets(vox, zmx, ldx, k0, kvalid, tid);
        const long ystep = (long)CT * ldy * Stager<YB, PN, NT>::ES;
        const char* ycur = reinterpret_cast<const char*>(dY) + c_beg * ystep;
        long xr = x_mod > 0 ? (c_beg * CT) % x_mod : c_beg * CT;          // row of X the next issue starts at
        long issued = 0;
        auto issue = [&](Stager<YB, PN, NT>& ys, Stager<XB, PK, NT>& xs) {
            ys.load_fast(ycur, voy);
            xs.load_fast(reinterpret_cast<const char*>(X) + xr * ldx * Stager<XB, PK, NT>::ES, vox);
            if constexpr (FILM) xs.load_film(film.g, film.b, film.ld, film.group, (c_beg + min(issued, nch - 1)) * CT, M, k0, kvalid, tid);
            if (issued + 1 < nch) {
                ycur += ystep;
                xr += CT;
                if (x_mod > 0 && xr >= x_mod) xr -= x_mod;
            }
            ++issued;
        };
        auto to_lds = [&](int st, int buf) {
            sy[st].store(Ysb(buf), LDY, tid, zmy, Ysl(buf));
            if constexpr (GG_WG_ABL & 2) {
                u32x4 t = sx[st].r[0];
                for (int i = 1; i < Stager<XB, PK, NT>::NP; ++i) t |= sx[st].r[i];
                if (t[0] == 0x12345678u) sx[st].store(Xsb(buf), LDX, tid);
            } else if constexpr (FILM) sx[st].store_film(Xsb(buf), LDX, tid, Xsl(buf));
            else sx[st].store(Xsb(buf), LDX, tid, 0, Xsl(buf));
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(sy[d], sx[d]);
        to_lds(0, 0);
        __syncthreads();
        for (long r0 = 0; r0 < nch; r0 += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const long r = r0 + u;
                if (r >= nch) break;
                const int buf = (int)(r & 1);
                issue(sy[u], sx[u]);                       // chunk r + DEPTH
                multiply(buf);
                to_lds((u + 1) % DEPTH, buf ^ 1);          // chunk r + 1
                __syncthreads();
            }
        }
    } else {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if (d < nch) {
            sy[d].load(dY, ldy, (c_beg + d) * CT, M, n0, nvalid, tid, true);
            sx[d].load(X, ldx, (c_beg + d) * CT, M, k0, kvalid, tid, false, x_mod);
            if constexpr (FILM) sx[d].load_film(film.g, film.b, film.ld, film.group, (c_beg + d) * CT, M, k0, kvalid, tid);
        }
    }
    sy[0].store(Ysb(0), LDY, tid, 0, Ysl(0));
    if constexpr (FILM) sx[0].store_film(Xsb(0), LDX, tid, Xsl(0));
    else sx[0].store(Xsb(0), LDX, tid, 0, Xsl(0));
    __syncthreads();
    for (long r0 = 0; r0 < nch; r0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const long r = r0 + u;
            if (r < nch) {
                const int buf = (int)(r & 1);
                if (r + DEPTH < nch) {
                    sy[u].load(dY, ldy, (c_beg + r + DEPTH) * CT, M, n0, nvalid, tid, true);
                    sx[u].load(X, ldx, (c_beg + r + DEPTH) * CT, M, k0, kvalid, tid, false, x_mod);
                    if constexpr (FILM) sx[u].load_film(film.g, film.b, film.ld, film.group, (c_beg + r + DEPTH) * CT, M, k0, kvalid, tid);
                }
                multiply(buf);
                if (r + 1 < nch) {
                    sy[(u + 1) % DEPTH].store(Ysb(buf ^ 1), LDY, tid, 0, Ysl(buf ^ 1));
                    if constexpr (FILM) sx[(u + 1) % DEPTH].store_film(Xsb(buf ^ 1), LDX, tid, Xsl(buf ^ 1));
                    else sx[(u + 1) % DEPTH].store(Xsb(buf ^ 1), LDX, tid, 0, Xsl(buf ^ 1));
                }
                __syncthreads();
            }
        }
    }
    }
    if constexpr (FGRAD) {
        // s of the wave's 32 rows -> LDS (the staging buffers are free after the last barrier), then per register row
        ssum += __shfl_xor(ssum, 32, 64);
        float* sl = reinterpret_cast<float*>(wg_smem) + wave * 32;
        if (h == 0) sl[c] = ssum;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 sv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) sv[g] = *reinterpret_cast<const f32x4*>(&sl[8 * g + 4 * h]);
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            float vg = 0.f, vb = 0.f;
            const bool colok = kb + b * 32 + c < kvalid;
            // the 16 weights of this lane's column: unconditional loads from clamped addresses, all in flight together (with
            // `ok ? load : 0` per element the compiler waited for every one of the 64 loads where it was issued: vmcnt(0) each)
            const int colc = min(k0 + kb + b * 32 + c, K - 1);
            float wv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = (i & 3) + 8 * (i >> 2);
                wv[i] = fg.W[(long)min(n0 + nb + 4 * h + rr, N - 1) * fg.ldw + colc];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = (i & 3) + 8 * (i >> 2);
                const float w = (colok && nb + 4 * h + rr < nvalid) ? wv[i] : 0.f;
                vg += acc[0][b][i] * w;
                vb += sv[i >> 2][i & 3] * w;
            }
            vg += __shfl_xor(vg, 32, 64);
            vb += __shfl_xor(vb, 32, 64);
            if (h == 0 && colok) {
                atomicAdd(fg.dgamma + (long)split * fg.ld + k0 + kb + b * 32 + c, vg);
                atomicAdd(fg.dbeta + (long)split * fg.ld + k0 + kb + b * 32 + c, vb);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    if (do_bias) {
        if constexpr (T22) {
            if ((wave & 1) == 0) {                       // both column halves hold the same dY rows: one of them adds
                ssum += __shfl_xor(ssum, 32, 64);
                ssum2 += __shfl_xor(ssum2, 32, 64);
                if (h == 0 && nb + c < nvalid) atomicAdd(dbias + n0 + nb + c, ssum);
                if (h == 0 && nb + 32 + c < nvalid) atomicAdd(dbias + n0 + nb + 32 + c, ssum2);
            }
        } else if (!W8 || (wave >> 2) == 0) {                // W8: both column halves hold the same dY rows
            ssum += __shfl_xor(ssum, 32, 64);
            if (h == 0 && nb + c < nvalid) atomicAdd(dbias + n0 + nb + c, ssum);
        }
    }
    // ---- dW[n0 + ..][k0 + ..] += panel : C/D map col = lane&31 (k), row = (i&3) + 8*(i>>2) + 4*h (n)
#ifdef GG_WG_L2ATOM
    // probe (tools/wgrad_probe.hip only): per-XCD partial panels with L2-scope atomics - dW must hold 8 x N x ldw floats
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;                 // HW_REG_XCC_ID[3:0]
    float* wbase = dW + (long)xcc * N * ldw + (long)(n0 + nb + 4 * h) * ldw + k0 + kb + c;
#define GG_WG_PANEL_ADD(ptr, v) __hip_atomic_fetch_add(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#elif defined(GG_WG_STORE)
    // probe (tools/wgrad_probe.hip only): plain stores of the partial panel into slice `split` of dW (splits x N x ldw floats), to be
    // summed by a second pass
    float* wbase = dW + (long)split * N * ldw + (long)(n0 + nb + 4 * h) * ldw + k0 + kb + c;
#define GG_WG_PANEL_ADD(ptr, v) (*(ptr) = (v))
#else
    float* wbase = dW + (long)(n0 + nb + 4 * h) * ldw + k0 + kb + c;
#define GG_WG_PANEL_ADD(ptr, v) atomicAdd(ptr, v)
#endif
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            if (kb + b * 32 + c < kvalid) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rr = a * 32 + (i & 3) + 8 * (i >> 2);       // row inside the wave's slice (minus 4h)
                    if (nb + 4 * h + rr < nvalid && !((GG_WG_ABL & 4) && acc[a][b][i] != 12345.f)) GG_WG_PANEL_ADD(wbase + (long)rr * ldw + b * 32, acc[a][b][i]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // one tile's accumulators at a time: no mass copy-out (spills)
        }
}
inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
}  // namespace

bool wgrad_supported(const void* dY, long ldy, int dy_bf16, const void* X, long ldx, int x_bf16, long M, int N, int K) {
    if (M < 4096 || N % 8 || K % 8 || N < 8 || K < 8) return false;              // short reductions stay on the generic split-K GEMM
    if (!al16(dY) || !al16(X)) return false;
    if (ldy % (dy_bf16 ? 8 : 4) || ldx % (x_bf16 ? 8 : 4)) return false;
    return true;
}

int wgrad(const void* dY, long ldy, int dy_bf16, const void* X, long ldx, int x_bf16, float* dW, long ldw, long M, int N, int K,
          hipStream_t st, const WgradFilm* film_in, const WgradFilmGrad* fgrad_in, float* dbias, long x_mod, int x3) {
    GG_REQUIRE(!x3 || (!dy_bf16 && !x_bf16), "wgrad: the split-operand (bf16x3) form takes fp32 operands");
    WgradFilm film;
    if (film_in) film = *film_in;
    WgradFilmGrad fgrad;
    if (fgrad_in) fgrad = *fgrad_in;
    GG_REQUIRE(!fgrad.W || (!film.g && !dy_bf16 && !x_bf16 && fgrad.tokens >= CT && fgrad.tokens % CT == 0 && M % fgrad.tokens == 0),
               "wgrad: FiLM-gradient mode needs fp32 operands and whole 32-token chunks per sample");
    GG_REQUIRE(!film.g || (!x_bf16 && film.group >= CT && film.ld % 4 == 0 && K % 4 == 0 && al16(film.g) && al16(film.b)),
               "wgrad: FiLM needs fp32 X, 16-byte aligned gamma / beta rows and at least one chunk of tokens per sample");
    GG_REQUIRE(wgrad_supported(dY, ldy, dy_bf16, X, ldx, x_bf16, M, N, K), "wgrad: unsupported shape / alignment");
    const int panels = ((N + PN - 1) / PN) * ((K + PK - 1) / PK);
    const long chunks = (M + CT - 1) / CT;
    // one workgroup per CU (256 accumulator registers => one wave per SIMD): panels x splits ~ 256; every extra
    // split costs a 256 KB atomic panel add, every missing one idles a CU
    int splits = (int)std::max<long>(1, std::min<long>(chunks / 8, (GG_WG_TARGET + panels - 1) / panels));
    if (splits >= 8) splits = splits / 8 * 8;            // whole groups of 8 splits (one per XCD), never more than 256 workgroups
    if (fgrad.W) splits = (int)(M / fgrad.tokens);      // one split per sample
    const int split_groups = (splits + 7) / 8;
    const dim3 grid((unsigned)(panels * split_groups * 8));
    constexpr int SMEM = 2 * CT * (LDY + LDX) * 2;
    // whole chunks and (for replica-shared inputs) a period of whole chunks: the branch-free loader; anything else (a ragged
    // last chunk) the general one
    static const bool no_fast = getenv("GG_WGRAD_GENERAL") != nullptr;
    static const bool w8 = getenv("GG_WGRAD_W4") == nullptr;            // eight-wave tiling unless GG_WGRAD_W4 is set
    const bool fast = !no_fast && M % CT == 0 && (x_mod == 0 || x_mod % CT == 0) && ldy * 4 * CT < (1L << 31) / 1 && ldx * 4 * CT < (1L << 31);
#define GG_WG1(YB, XB, FL, FG, FA, W8_)                                                                                       \
    do {                                                                                                               \
        static bool attr = false;                                                                                      \
        if (!attr) {                                                                                                   \
            GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<YB, XB, FL, FG, FA, W8_>),    \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));                        \
            attr = true;                                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL((wgrad_kernel<YB, XB, FL, FG, FA, W8_>), grid, dim3(W8_ ? 512 : 256), SMEM, st, dY, ldy, X, ldx, dW, ldw, M, N, K, splits, film, fgrad, dbias, x_mod); \
    } while (0)
    // bf16x3: fp32 operands, hi + lo images (2 x the LDS), eight waves when the loader is the branch-free one
#define GG_WG3(FL, FG)                                                                                                  \
    do {                                                                                                               \
        static bool attr3f = false, attr3g = false;                                                                    \
        if (fast) {                                                                                                    \
            if (!attr3f) GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<false, false, FL, FG, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SMEM)); \
            hipLaunchKernelGGL((wgrad_kernel<false, false, FL, FG, true, true, true>), grid, dim3(512), 2 * SMEM, st, dY, ldy, X, ldx, dW, ldw, M, N, K, splits, film, fgrad, dbias, x_mod); \
        } else {                                                                                                       \
            if (!attr3g) GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<false, false, FL, FG, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SMEM)); \
            hipLaunchKernelGGL((wgrad_kernel<false, false, FL, FG, false, false, true>), grid, dim3(256), 2 * SMEM, st, dY, ldy, X, ldx, dW, ldw, M, N, K, splits, film, fgrad, dbias, x_mod); \
        }                                                                                                              \
        (fast ? attr3f : attr3g) = true;                                                                               \
    } while (0)
    if (x3) {
        if (fgrad.W) GG_WG3(false, true);
        else if (film.g) GG_WG3(true, false);
        else GG_WG3(false, false);
        GG_CHECK_HIP(hipGetLastError());
        return 0;
    }
#define GG_WG(YB, XB, FL, FG)                  \
    do {                                       \
        if (fast && w8) GG_WG1(YB, XB, FL, FG, true, true); \
        else if (fast) GG_WG1(YB, XB, FL, FG, true, false); \
        else GG_WG1(YB, XB, FL, FG, false, false);    \
    } while (0)
    if (fgrad.W) {
        GG_WG(false, false, false, true);
    } else if (film.g) {
        if (dy_bf16) GG_WG(true, false, true, false);
        else GG_WG(false, false, true, false);
    } else if (dy_bf16 && x_bf16) GG_WG(true, true, false, false);
    else if (dy_bf16) GG_WG(true, false, false, false);
    else if (x_bf16) GG_WG(false, true, false, false);
    else GG_WG(false, false, false, false);
#undef GG_WG
#undef GG_WG1
#undef GG_WG3
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gg
