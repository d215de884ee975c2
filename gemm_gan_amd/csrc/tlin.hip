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
#include "drop_rng.h"
#include "fp8_util.h"
#include <type_traits>
#include <hip/hip_ext.h>
#include <cstdlib>

// Ablation switches of tools/tlin_probe (GG_TLIN_DBG bits: 1 no stores, 2 no MFMA, 4 no weight loads, 8 no X loads, 16 no epilogue
// operands, 32 no dropout) exist only in a probe build (-DGG_TLIN_DBG_RT): as run-time tests they put a condition around every
// load of the production kernels, and the compiler's wait-count pass then waits for a load where it is issued (vmcnt(0)).
#ifdef GG_TLIN_DBG_RT
#define TLIN_DBG(bit) (p.dbg & (bit))
#else
#define TLIN_DBG(bit) false
#endif

namespace gg {

namespace {
// Optional dispatch timestamps for the next launch (tlin_time_next): hipExtLaunchKernelGGL stamps the two events at the
// kernel's own begin / end, like a profiler, instead of bracketing it with barrier packets on the stream.
hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
template <typename K>
inline void launch_timed(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t st, const TlinP& p) {
    if (g_ev0) {
        hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)smem, st, g_ev0, g_ev1, 0, p);
        g_ev0 = g_ev1 = nullptr;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, smem, st, p);
    }
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

constexpr float LN_EPS = 1e-5f;


// ---------------------------------------------------------------------------------------------------------------
// Both kernels are latency-structured for one / two waves per SIMD: every HBM load a wave tile needs is issued in
// one batch (activation slice, then the residual / accumulate rows for the epilogue), small per-feature vectors
// (bias, LayerNorm gamma / beta) sit in LDS, and no load follows a store inside a tile.  Measured ceiling for this
// traffic shape with batched loads at one workgroup per CU: 4.5 - 5.1 TB/s (tools/bw_probe.hip, mix_probe).
// ---------------------------------------------------------------------------------------------------------------

// Feature order inside a 32-feature chunk.  The MFMA writes output row (i&3) + 8*(i>>2) + 4*h to register i of lane
// half h; feeding A-operand row r with weight row 16*((r>>2)&1) + (r&3) + 4*(r>>3) makes register i of half h hold
// feature 16*h + i: every lane owns 16 CONSECUTIVE features of its token (64 B fp32 / 32 B bf16 per chunk), so all
// epilogue loads and stores are 16-byte accesses to contiguous runs (8-byte pieces measured 3.1 TB/s, these 5.7+).
__device__ __forceinline__ int a_row_of_lane(int r) { return 16 * ((r >> 2) & 1) + (r & 3) + 4 * (r >> 3); }

// stages rows [tok0, tok0+32) x [k0, k0+W) of X (fp32 or bf16, FiLM optional) as bf16 into the wave-private slab xs
// (row stride LD elements).  GB bounds the loads in flight (registers): all of them for bf16, halves for fp32.
template <int W, int LD, bool XB, int ROWS = 32>
__device__ __forceinline__ void stage_x(const TlinP& p, __bf16* xs, int tok0, int last_tok, int k0, int lane) {
    // the row offsets are recomputed per call (32-bit, a few VALU ops) instead of living in registers across the
    // K loop: the empty asm hides their loop invariance from the optimiser
    asm volatile("" : "+v"(tok0));
    const unsigned char* const Xc = reinterpret_cast<const unsigned char*>(p.X);
    if constexpr (XB) {
        constexpr int LPR = W / 8, RPI = 64 / LPR, NLD = ROWS / RPI;
        const int lrow = lane / LPR, lcol = 8 * (lane % LPR);
        const unsigned ldb = (unsigned)p.ldx * 2u, cb = (unsigned)(k0 + lcol) * 2u;
        u32x4 v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) v[i] = *reinterpret_cast<const u32x4*>(Xc + ((unsigned)min(tok0 + lrow + RPI * i, last_tok) * ldb + cb));
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(&xs[(RPI * i + lrow) * LD + lcol]) = v[i];
    } else {
        constexpr int LPR = W / 4, RPI = 64 / LPR, NLD = ROWS / RPI;
        const int lrow = lane / LPR, lcol = 4 * (lane % LPR);
        const unsigned ldb = (unsigned)p.ldx * 4u, cb = (unsigned)(k0 + lcol) * 4u;
        constexpr int GB = NLD < 8 ? NLD : 8;     // loads in flight per batch (register budget)
        if (p.film_g) {
            // FiLM rows are per SAMPLE (film_group tokens): 32 consecutive tokens touch at most two of them
            // (film_group >= 32), so gamma / beta of both groups are fetched once per call for this lane's columns and
            // the activation rows keep the same batched loads as the plain path.
            const int tb = min(tok0, last_tok);
            const int g0 = tb / p.film_group, rem0 = tb - g0 * p.film_group;
            const int g1 = min(g0 + 1, last_tok / p.film_group);
            const unsigned fldb = (unsigned)p.film_ld * 4u;
            const unsigned char* const Gc = reinterpret_cast<const unsigned char*>(p.film_g) + cb;
            const unsigned char* const Bc = reinterpret_cast<const unsigned char*>(p.film_b) + cb;
            const f32x4 ga = *reinterpret_cast<const f32x4*>(Gc + (unsigned)g0 * fldb), ba = *reinterpret_cast<const f32x4*>(Bc + (unsigned)g0 * fldb);
            const f32x4 gb = *reinterpret_cast<const f32x4*>(Gc + (unsigned)g1 * fldb), bb = *reinterpret_cast<const f32x4*>(Bc + (unsigned)g1 * fldb);
            // two batches of GB loads alternate: batch b + 1 is requested before batch b is modulated and written to LDS, so
            // a slice costs one exposed round trip instead of NLD / GB of them (the kernel runs one wave per SIMD)
            f32x4 v[2][GB];
#pragma unroll
            for (int i = 0; i < GB; ++i) v[0][i] = *reinterpret_cast<const f32x4*>(Xc + ((unsigned)min(tok0 + lrow + RPI * i, last_tok) * ldb + cb));
#pragma unroll
            for (int b0 = 0; b0 < NLD; b0 += GB) {
                const int cur = (b0 / GB) & 1;
                if (b0 + GB < NLD) {
#pragma unroll
                    for (int i = 0; i < GB; ++i)
                        v[cur ^ 1][i] = *reinterpret_cast<const f32x4*>(Xc + ((unsigned)min(tok0 + lrow + RPI * (b0 + GB + i), last_tok) * ldb + cb));
                }
#pragma unroll
                for (int i = 0; i < GB; ++i) {
                    const int r = min(tok0 + lrow + RPI * (b0 + i), last_tok);
                    const bool second = rem0 + (r - tb) >= p.film_group;
                    const f32x4 m = (second ? gb : ga) * v[cur][i] + (second ? bb : ba);
                    u32x2 w = {pack2(m[0], m[1]), pack2(m[2], m[3])};
                    *reinterpret_cast<u32x2*>(&xs[(RPI * (b0 + i) + lrow) * LD + lcol]) = w;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            f32x4 v[2][GB];
#pragma unroll
            for (int i = 0; i < GB; ++i) v[0][i] = *reinterpret_cast<const f32x4*>(Xc + ((unsigned)min(tok0 + lrow + RPI * i, last_tok) * ldb + cb));
#pragma unroll
            for (int b0 = 0; b0 < NLD; b0 += GB) {
                const int cur = (b0 / GB) & 1;
                if (b0 + GB < NLD) {
#pragma unroll
                    for (int i = 0; i < GB; ++i)
                        v[cur ^ 1][i] = *reinterpret_cast<const f32x4*>(Xc + ((unsigned)min(tok0 + lrow + RPI * (b0 + GB + i), last_tok) * ldb + cb));
                }
#pragma unroll
                for (int i = 0; i < GB; ++i) {
                    u32x2 w = {pack2(v[cur][i][0], v[cur][i][1]), pack2(v[cur][i][2], v[cur][i][3])};
                    *reinterpret_cast<u32x2*>(&xs[(RPI * (b0 + i) + lrow) * LD + lcol]) = w;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// ---- fp8 (OCP e4m3) operand path ---------------------------------------------------------------------------------
// F8 instantiations keep the structure of the bf16 kernels and change the operand format: activations are quantised to
// e4m3 on their way into the wave-private slab (x * 2^x_exp, clamped to +-448), the weights come from an e4m3 shadow copy
// (w * 2^w_exp, per tensor, *p.w_exp), and the products run on the block-scaled matrix instructions
// v_mfma_scale_f32_{16x16x128,32x32x64}_f8f6f4 - twice the bf16 rate at 4x the K per instruction - whose E8M0 scale
// operands undo the two power-of-two scales inside the instruction (scale byte = 127 - exponent), so accumulators,
// epilogues and outputs are exactly those of the bf16 kernels.  A lane's fragment is 32 CONSECUTIVE k (32 bytes): lane
// (c, g) of a 16x16x128 holds k = 32 g .. 32 g + 31 of row / column c, lane (c, h) of a 32x32x64 k = 32 h .. 32 h + 31.
// rows [tok0, tok0+ROWS) x [k0, k0+W) of X (fp32 or bf16) as e4m3 bytes into the slab xs (row stride LD bytes)
template <int W, int LD, bool XB, int ROWS>
__device__ __forceinline__ void stage_x8(const TlinP& p, unsigned char* xs, int tok0, int last_tok, int k0, int lane, float sc) {
    asm volatile("" : "+v"(tok0));
    const unsigned char* const Xc = reinterpret_cast<const unsigned char*>(p.X);
    if constexpr (XB) {
        constexpr int LPR = W / 8, RPI = 64 / LPR, NLD = ROWS / RPI;
        const int lrow = lane / LPR, lcol = 8 * (lane % LPR);
        const unsigned ldb = (unsigned)p.ldx * 2u, cb = (unsigned)(k0 + lcol) * 2u;
        u32x4 v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) v[i] = *reinterpret_cast<const u32x4*>(Xc + ((unsigned)min(tok0 + lrow + RPI * i, last_tok) * ldb + cb));
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f[2 * j] = __builtin_bit_cast(float, v[i][j] << 16);
                f[2 * j + 1] = __builtin_bit_cast(float, v[i][j] & 0xffff0000u);
            }
            u32x2 w = {cvt4_fp8(f[0], f[1], f[2], f[3], sc), cvt4_fp8(f[4], f[5], f[6], f[7], sc)};
            *reinterpret_cast<u32x2*>(&xs[(RPI * i + lrow) * LD + lcol]) = w;
        }
    } else {
        constexpr int LPR = W / 4, RPI = 64 / LPR, NLD = ROWS / RPI;
        const int lrow = lane / LPR, lcol = 4 * (lane % LPR);
        const unsigned ldb = (unsigned)p.ldx * 4u, cb = (unsigned)(k0 + lcol) * 4u;
        constexpr int GB = NLD < 8 ? NLD : 8;
#pragma unroll
        for (int b0 = 0; b0 < NLD; b0 += GB) {
            f32x4 v[GB];
#pragma unroll
            for (int i = 0; i < GB; ++i) v[i] = *reinterpret_cast<const f32x4*>(Xc + ((unsigned)min(tok0 + lrow + RPI * (b0 + i), last_tok) * ldb + cb));
#pragma unroll
            for (int i = 0; i < GB; ++i)
                *reinterpret_cast<unsigned*>(&xs[(RPI * (b0 + i) + lrow) * LD + lcol]) = cvt4_fp8(v[i][0], v[i][1], v[i][2], v[i][3], sc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- resident: all N = 32*NT_RES accumulators in registers, K streamed in slices of KSL ------------------------
// PRE: what the epilogue adds to the product, prefetched while the last slice is multiplied:
//      0 nothing, 1 residual rows, 2 previous output (accumulate), 3 decided at run time (generic, branchy)
enum { PRE_NONE = 0, PRE_RES = 1, PRE_ACC = 2, PRE_ANY = 3 };
template <int NT_RES, int KSL, bool XB, int PRE>
__global__ __launch_bounds__(256, 1) void tlin_res_kernel(const TlinP p) {
    const DropKey dkey = drop_live(p.drop);
    constexpr int N = 32 * NT_RES;
    constexpr int WLD = KSL + 8;                       // bf16 per LDS row (weights and activations)
    constexpr int PIECES = KSL / 8;
    constexpr int WLOADS = (32 * PIECES + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* const Ws = reinterpret_cast<__bf16*>(smem_raw);                 // [2][32*WLD]
    __bf16* const Xs = Ws + 2 * 32 * WLD;                                   // [4][32*WLD]
    float* const Ps = reinterpret_cast<float*>(Xs + 4 * 32 * WLD);          // bias | gamma | beta  [3][N]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    const int tok0 = (int)blockIdx.x * 128 + wave * 32;
    const int last_tok = (int)p.M - 1;
    const int nks = p.K / KSL;
    const int nchunks = nks * NT_RES;
    __bf16* const xs = Xs + wave * 32 * WLD;

    // weight chunks (32 output features x KSL) travel L2 -> registers -> LDS two chunks ahead of their use: the
    // loads of chunk c+2 are issued when chunk c is multiplied and written to LDS one chunk later, so an L2 round
    // trip is covered by two MFMA passes.  Chunk c uses register set and LDS buffer c & 1 (NT_RES is even).
    u32x4 wreg[2][WLOADS];
    const __bf16* Wp = reinterpret_cast<const __bf16*>(p.W);
    auto load_chunk = [&](int set, int ks, int nt) {
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            const int f = tid + 256 * i;
            const int row = f / PIECES, piece = f % PIECES;
            if ((32 * PIECES) % 256 == 0 || row < 32)
                wreg[set][i] = *reinterpret_cast<const u32x4*>(Wp + (long)(nt * 32 + row) * p.ldw + ks * KSL + 8 * piece);
        }
    };
    auto store_chunk = [&](int set) {
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            const int f = tid + 256 * i;
            const int row = f / PIECES, piece = f % PIECES;
            if ((32 * PIECES) % 256 == 0 || row < 32) *reinterpret_cast<u32x4*>(&Ws[set * 32 * WLD + row * WLD + 8 * piece]) = wreg[set][i];
        }
    };

    load_chunk(0, 0, 0);
    load_chunk(1, 0, 1);
    for (int i = tid; i < N; i += 256) {
        Ps[i] = p.bias ? p.bias[i] : 0.f;
        Ps[N + i] = p.ln_g ? p.ln_g[i] : 1.f;
        Ps[2 * N + i] = p.ln_g ? p.ln_b[i] : 0.f;
    }
    store_chunk(0);

    // epilogue row of this lane.  Rows past the end are clamped to the last token: such lanes recompute that row
    // bit for bit, so their stores are harmless duplicates - except when accumulating, where stores are predicated.
    const int tok = tok0 + c;
    const bool valid = tok <= last_tok;
    const int tokc = valid ? tok : last_tok;
    const long yrow = p.y_row_group ? (long)tokc + tokc / p.y_row_group + 1 : (long)tokc;
    float* const yb = reinterpret_cast<float*>(p.Y) + yrow * p.ldy;
    const float* const resp = p.res ? p.res + (long)(tokc % (int)p.res_rows) * p.ldres : nullptr;
    constexpr bool PRED = PRE >= PRE_ACC;
    const bool keep_y = p.y_rows < 0 || tokc < p.y_rows;

    f32x16 acc[NT_RES];
    bf16x8 xf[KSL / 16];
    f32x4 pre[NT_RES][4];
    // first half with the last activation slice, second half from the middle of the last MFMA pass
    auto load_pre = [&](int nt0, int nt1) {
#pragma unroll
        for (int nt = nt0; nt < nt1; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nt * 32 + 8 * g + 4 * h;
                if constexpr (PRE == PRE_RES) pre[nt][g] = *reinterpret_cast<const f32x4*>(resp + n);
                if constexpr (PRE == PRE_ACC) pre[nt][g] = *reinterpret_cast<const f32x4*>(yb + n);
                if constexpr (PRE == PRE_ANY) {
                    pre[nt][g] = resp ? *reinterpret_cast<const f32x4*>(resp + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p.accumulate) pre[nt][g] += *reinterpret_cast<const f32x4*>(yb + n);
                }
            }
    };

    int chunk = 0;
    for (int ks = 0; ks < nks; ++ks) {
        if (!TLIN_DBG(8)) stage_x<KSL, WLD, XB>(p, xs, tok0, last_tok, ks * KSL, lane);
        if constexpr (PRE != PRE_NONE) {
            if (ks == nks - 1 && !TLIN_DBG(16)) load_pre(0, NT_RES / 2);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < KSL / 16; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(&xs[c * WLD + 16 * s + 8 * h]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __syncthreads();     // chunk `chunk` is in Ws[chunk & 1] (and Ps on the first pass)

#pragma unroll
        for (int nt = 0; nt < NT_RES; ++nt) {
            const int buf = nt & 1;      // compile-time after unrolling
            if (!TLIN_DBG(4)) load_chunk(buf, min(nt + 2 >= NT_RES ? ks + 1 : ks, nks - 1), (nt + 2) % NT_RES);      // unconditional: the last two fetch chunks nobody uses
            if (ks == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
            }
            if constexpr (PRE != PRE_NONE) {
                if (nt == NT_RES / 2 && ks == nks - 1 && !TLIN_DBG(16)) load_pre(NT_RES / 2, NT_RES);
            }
            const __bf16* wsb = Ws + buf * 32 * WLD + c * WLD + 8 * h;
#pragma unroll
            for (int s4 = 0; s4 < KSL / 64; ++s4) {
                bf16x8 wf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) wf[u] = *reinterpret_cast<const bf16x8*>(wsb + 16 * (4 * s4 + u));
                if (!TLIN_DBG(2)) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u], xf[4 * s4 + u], acc[nt], 0, 0, 0);
                }
            }
            store_chunk(buf ^ 1);          // (after the last chunk: a buffer nobody reads)
            __syncthreads();
            ++chunk;
        }
    }

    // ---- epilogue: bias, dropout, residual / accumulate, LayerNorm over the N features; stores only ----------
    float sum = 0.f;
    auto epilogue = [&](auto drop_tag) {
        constexpr bool DROP = decltype(drop_tag)::value;
        const float ksd = DROP ? 1.f / (1.f - p.drop.p) : 1.f;
        const uint64_t dbase = (uint64_t)tokc * p.drop_ld;
#pragma unroll
        for (int nt = 0; nt < NT_RES; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nt * 32 + 8 * g + 4 * h;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[n]);
                f32x4 v = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
                v += bb;
                if constexpr (DROP) {
                    float f[4];
                    drop_factor4(dkey, dbase + n, ksd, f);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= f[j];
                }
                if constexpr (PRE != PRE_NONE) v += pre[nt][g];
                if ((!PRED || valid) && keep_y && !(TLIN_DBG(1) && v[0] != 1234.5f)) *reinterpret_cast<f32x4*>(yb + n) = v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[nt][4 * g + j] = v[j];
                    sum += v[j];
                }
            }
            __builtin_amdgcn_sched_barrier(0);     // one feature tile at a time: LDS reads are not hoisted across tiles
        }
    };
    if (p.drop.p > 0.f && !TLIN_DBG(32)) epilogue(std::true_type{});
    else epilogue(std::false_type{});

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
                if ((!PRED || valid) && !(TLIN_DBG(1) && y[0] != 1234.5f)) *reinterpret_cast<f32x4*>(lb + n) = y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (h == 0 && valid && keep_y) {
            p.ln_stats[2 * (long)tok] = mean;
            p.ln_stats[2 * (long)tok + 1] = rstd;
        }
    }
}

// ---- resident, 16-token wave tiles (v_mfma_f32_16x16x32_bf16) ----------------------------------------------------
// Same contract as tlin_res_kernel, half the registers and LDS per workgroup (64 tokens), so TWO workgroups share a
// CU: while one multiplies / normalises, the other's activation and residual loads are in flight.  The accumulator
// tile is 16 features x 16 tokens: lane = (token lane&15, feature quad lane>>4), four consecutive features per
// register quad, so one store instruction writes 16 rows x 64 contiguous bytes.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int NT_RES, int KSL, bool XB, int PRE, bool F8 = false>
__global__ __launch_bounds__(256, 2) void tlin_res16_kernel(const TlinP p) {
    const DropKey dkey = drop_live(p.drop);
    constexpr int N = 32 * NT_RES;
    constexpr int NF = 2 * NT_RES;                     // 16-feature tiles
    constexpr int ESZ = F8 ? 1 : 2;                    // operand bytes per element
    constexpr int WLD = F8 ? KSL + 16 : KSL + 8;       // elements per LDS row (weights and activations): 4 banks per row step
    constexpr int PIECES = KSL * ESZ / 16;             // 16-byte pieces per weight row
    constexpr int WLOADS = (32 * PIECES + 255) / 256;
    constexpr int KS32 = F8 ? KSL / 128 : KSL / 32;    // MFMA steps per K slice (fragments held)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* const Wsb = smem_raw;                                    // [2][32*WLD] elements
    unsigned char* const Xsb = Wsb + 2 * 32 * WLD * ESZ;                    // [4][16*WLD]
    __bf16* const Ws = reinterpret_cast<__bf16*>(Wsb);
    __bf16* const Xs = reinterpret_cast<__bf16*>(Xsb);
    float* const Ps = reinterpret_cast<float*>(Xsb + 4 * 16 * WLD * ESZ);   // bias | gamma | beta  [3][N]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, q = lane >> 4;
    const int tok0 = (int)blockIdx.x * 64 + wave * 16;
    const int last_tok = (int)p.M - 1;
    const int nks = p.K / KSL;
    const int nchunks = nks * NT_RES;
    __bf16* const xs = Xs + wave * 16 * WLD;

    u32x4 wreg[WLOADS];
    const unsigned char* Wpb = reinterpret_cast<const unsigned char*>(p.W);
    auto load_chunk = [&](int ks, int nt) {
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            const int f = tid + 256 * i;
            const int row = f / PIECES, piece = f % PIECES;
            if ((32 * PIECES) % 256 == 0 || row < 32)
                wreg[i] = *reinterpret_cast<const u32x4*>(Wpb + ((long)(nt * 32 + row) * p.ldw + ks * KSL) * ESZ + 16 * piece);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            const int f = tid + 256 * i;
            const int row = f / PIECES, piece = f % PIECES;
            if ((32 * PIECES) % 256 == 0 || row < 32) *reinterpret_cast<u32x4*>(&Wsb[(buf * 32 * WLD + row * WLD) * ESZ + 16 * piece]) = wreg[i];
        }
    };

    load_chunk(0, 0);
    for (int i = tid; i < N; i += 256) {
        Ps[i] = p.bias ? p.bias[i] : 0.f;
        Ps[N + i] = p.ln_g ? p.ln_g[i] : 1.f;
        Ps[2 * N + i] = p.ln_g ? p.ln_b[i] : 0.f;
    }
    store_chunk(0);
    // fp8: E8M0 scale bytes that undo the operand scales inside the instruction, activation scale for the staging
    const int sc_w = F8 ? 127 - *p.w_exp : 0, sc_x = F8 ? 127 - p.x_exp : 0;
    const float xscale = F8 ? exp2i(p.x_exp) : 1.f;
    unsigned char* const xs8 = Xsb + wave * 16 * WLD;

    const int tok = tok0 + c;
    const bool valid = tok <= last_tok;
    const int tokc = valid ? tok : last_tok;           // clamped lanes recompute the last row bit for bit
    const long yrow = p.y_row_group ? (long)tokc + tokc / p.y_row_group + 1 : (long)tokc;
    float* const yb = reinterpret_cast<float*>(p.Y) + yrow * p.ldy + 4 * q;
    const float* const resp = p.res ? p.res + (long)(tokc % (int)p.res_rows) * p.ldres + 4 * q : nullptr;
    constexpr bool PRED = PRE >= PRE_ACC;
    const bool keep_y = p.y_rows < 0 || tokc < p.y_rows;

    f32x4 acc[NF];
    bf16x8 xf[F8 ? 1 : KS32];
    i32x8 xf8[F8 ? KS32 : 1];
    f32x4 pre[NF];

    int chunk = 0;
    for (int ks = 0; ks < nks; ++ks) {
        if constexpr (F8) stage_x8<KSL, WLD, XB, 16>(p, xs8, tok0, last_tok, ks * KSL, lane, xscale);
        else stage_x<KSL, WLD, XB, 16>(p, xs, tok0, last_tok, ks * KSL, lane);
        if constexpr (PRE != PRE_NONE) {
            if (ks == nks - 1) {
#pragma unroll
                for (int t = 0; t < NF; ++t) {
                    if constexpr (PRE == PRE_RES) pre[t] = *reinterpret_cast<const f32x4*>(resp + 16 * t);
                    if constexpr (PRE == PRE_ACC) pre[t] = *reinterpret_cast<const f32x4*>(yb + 16 * t);
                    if constexpr (PRE == PRE_ANY) {
                        pre[t] = resp ? *reinterpret_cast<const f32x4*>(resp + 16 * t) : f32x4{0.f, 0.f, 0.f, 0.f};
                        if (p.accumulate) pre[t] += *reinterpret_cast<const f32x4*>(yb + 16 * t);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (F8) {
#pragma unroll
            for (int s = 0; s < KS32; ++s) xf8[s] = lds_frag32(xs8 + c * WLD + 128 * s + 32 * q);
        } else {
#pragma unroll
            for (int s = 0; s < KS32; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(&xs[c * WLD + 32 * s + 8 * q]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __syncthreads();     // chunk `chunk` is in Ws[chunk & 1] (and Ps on the first pass)

#pragma unroll
        for (int nt = 0; nt < NT_RES; ++nt) {
            const int buf = nt & 1;      // compile-time after unrolling (NT_RES is even)
            if (chunk + 1 < nchunks) load_chunk(nt + 1 == NT_RES ? ks + 1 : ks, (nt + 1) % NT_RES);
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                if (ks == 0) acc[2 * nt + ft] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (F8) {
                    const unsigned char* wsb8 = Wsb + buf * 32 * WLD + (16 * ft + c) * WLD + 32 * q;
#pragma unroll
                    for (int s = 0; s < KS32; ++s)
                        acc[2 * nt + ft] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(lds_frag32(wsb8 + 128 * s), xf8[s], acc[2 * nt + ft],
                                                                                             0, 0, 0, sc_w, 0, sc_x);
                    // keep this chunk's fragment reads and products together: left alone, the compiler sinks the products
                    // of ALL chunks below the chunk loop and keeps 256 registers of fragments alive (spilled) until then
                    asm volatile("" : "+v"(acc[2 * nt + ft]));
                    continue;
                }
                const __bf16* wsb = Ws + buf * 32 * WLD + (16 * ft + c) * WLD + 8 * q;
#pragma unroll
                for (int s4 = 0; s4 < KS32 / 4; ++s4) {
                    bf16x8 wf[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) wf[u] = *reinterpret_cast<const bf16x8*>(wsb + 32 * (4 * s4 + u));
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[2 * nt + ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf[4 * s4 + u], acc[2 * nt + ft], 0, 0, 0);
                }
                if constexpr (KS32 % 4 != 0) {
#pragma unroll
                    for (int s = KS32 / 4 * 4; s < KS32; ++s) {
                        const bf16x8 wf1 = *reinterpret_cast<const bf16x8*>(wsb + 32 * s);
                        acc[2 * nt + ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1, xf[s], acc[2 * nt + ft], 0, 0, 0);
                    }
                }
                // the tile's products are materialised here (see the fp8 branch): -16 B of spills per lane, -8 % kernel time
                asm volatile("" : "+v"(acc[2 * nt + ft]));
            }
            if (chunk + 1 < nchunks) store_chunk(buf ^ 1);
            __syncthreads();
            ++chunk;
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------
    float sum = 0.f;
    auto epilogue = [&](auto drop_tag) {
        constexpr bool DROP = decltype(drop_tag)::value;
        const float ksd = DROP ? 1.f / (1.f - p.drop.p) : 1.f;
        const uint64_t dbase = (uint64_t)tokc * p.drop_ld + 4 * q;
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[16 * t + 4 * q]);
            f32x4 v = acc[t] + bb;
            if constexpr (DROP) {
                float f[4];
                drop_factor4(dkey, dbase + 16 * t, ksd, f);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= f[j];
            }
            if constexpr (PRE != PRE_NONE) v += pre[t];
            if ((!PRED || valid) && keep_y) *reinterpret_cast<f32x4*>(yb + 16 * t) = v;
            acc[t] = v;
            sum += (v[0] + v[1]) + (v[2] + v[3]);
            if (t % 4 == 3) __builtin_amdgcn_sched_barrier(0);      // bounds the hoisting of LDS reads (registers)
        }
    };
    if (p.drop.p > 0.f) epilogue(std::true_type{});
    else epilogue(std::false_type{});

    if (p.ln_g) {
        const float invn = 1.f / (float)N;
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * invn;
        float var = 0.f;
#pragma unroll
        for (int t = 0; t < NF; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = acc[t][j] - mean;
                var += d * d;
            }
        var += __shfl_xor(var, 16, 64);
        var += __shfl_xor(var, 32, 64);
        const float rstd = rsqrtf(var * invn + LN_EPS);
        float* const lb = p.ln_y + (long)tokc * p.ldy + 4 * q;
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            const f32x4 gg_ = *reinterpret_cast<const f32x4*>(&Ps[N + 16 * t + 4 * q]);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[2 * N + 16 * t + 4 * q]);
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (acc[t][j] - mean) * rstd * gg_[j] + bb[j];
            if (!PRED || valid) *reinterpret_cast<f32x4*>(lb + 16 * t) = y;
            if (t % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (q == 0 && valid && keep_y) {
            p.ln_stats[2 * (long)tok] = mean;
            p.ln_stats[2 * (long)tok + 1] = rstd;
        }
    }
}

// ---- stream: K <= 256 register-resident, N streamed 32 features at a time ----------------------------------------
// EPI: 0 bias (+ReLU), 1 bias (+ReLU) + dropout, 2 bias + sign mask of a reference tensor, 3 decided at run time
enum { EPI_BIAS = 0, EPI_DROP = 1, EPI_MASK = 2, EPI_ANY = 3 };
#ifndef GG_STR_CB
#define GG_STR_CB 4
#endif
#ifndef GG_STR_OCC
#define GG_STR_OCC 2
#endif
template <int KSL, bool XB, bool YB, int EPI, bool F8 = false>
__global__ __launch_bounds__(256, GG_STR_OCC) void tlin_str_kernel(const TlinP p) {
    const DropKey dkey = drop_live(p.drop);
    constexpr int XW = KSL < 128 ? KSL : 128;          // staging window (bounds LDS so two workgroups fit a CU)
    constexpr int ESZ = F8 ? 1 : 2;                    // operand bytes per element
    constexpr int XLDW = F8 ? XW + 32 : XW + 8;
    constexpr int WLD = F8 ? KSL + 32 : KSL + 8;
    constexpr int PIECES = KSL * ESZ / 16;
    constexpr int WLOADS = (32 * PIECES + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* const Wsb = smem_raw;                                    // [2][32*WLD] elements
    unsigned char* const Xsb = Wsb + 2 * 32 * WLD * ESZ;                    // [4][32*XLDW]
    __bf16* const Ws = reinterpret_cast<__bf16*>(Wsb);
    __bf16* const Xs = reinterpret_cast<__bf16*>(Xsb);
    float* const Ps = reinterpret_cast<float*>(Xsb + 4 * 32 * XLDW * ESZ);  // bias [N]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;
    // bf16 outputs use the permuted feature order (16 consecutive features per lane -> 16-byte stores); fp32 outputs
    // keep the MFMA order, where the two lanes of a token write adjacent 16-byte pieces (32 B runs per instruction)
    constexpr bool PERM = YB;
    const int arow = PERM ? a_row_of_lane(c) : c;
    const int hoff = PERM ? 16 * h : 4 * h;      // feature offset of this lane half, group g adds GS * g
    constexpr int GS = PERM ? 4 : 8;
    const int tok0 = (int)blockIdx.x * 128 + wave * 32;
    const int last_tok = (int)p.M - 1;
    const int ntiles = p.N / 32;
    __bf16* const xs = Xs + wave * 32 * XLDW;

    // two-deep register ring for the weight chunks (see the resident kernel); chunk nt uses set / buffer nt & 1
    u32x4 wreg[2][WLOADS];
    const unsigned char* Wpb = reinterpret_cast<const unsigned char*>(p.W);
    auto load_chunk = [&](int set, int nt) {
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            const int f = tid + 256 * i;
            const int row = f / PIECES, piece = f % PIECES;
            if ((32 * PIECES) % 256 == 0 || row < 32) wreg[set][i] = *reinterpret_cast<const u32x4*>(Wpb + (long)(nt * 32 + row) * p.ldw * ESZ + 16 * piece);
        }
    };
    auto store_chunk = [&](int set) {
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            const int f = tid + 256 * i;
            const int row = f / PIECES, piece = f % PIECES;
            if ((32 * PIECES) % 256 == 0 || row < 32) *reinterpret_cast<u32x4*>(&Wsb[(set * 32 * WLD + row * WLD) * ESZ + 16 * piece]) = wreg[set][i];
        }
    };
    const int sc_w = F8 ? 127 - *p.w_exp : 0, sc_x = F8 ? 127 - p.x_exp : 0;
    const float xscale = F8 ? exp2i(p.x_exp) : 1.f;
    unsigned char* const xs8 = Xsb + wave * 32 * XLDW;

    unsigned long long t_start = 0, t_x = 0, t_it0 = 0;
    if (p.stamps) t_start = __builtin_amdgcn_s_memtime();
    load_chunk(0, 0);
    if (ntiles > 1) load_chunk(1, 1);
    for (int i = tid; i < p.N; i += 256) Ps[i] = p.bias ? p.bias[i] : 0.f;
    store_chunk(0);

    bf16x8 xf[F8 ? 1 : KSL / 16];
    i32x8 xf8[F8 ? KSL / 64 : 1];
#pragma unroll
    for (int q = 0; q < KSL / XW; ++q) {
        if constexpr (F8) stage_x8<XW, XLDW, XB, 32>(p, xs8, tok0, last_tok, q * XW, lane, xscale);
        else if (!TLIN_DBG(8)) stage_x<XW, XLDW, XB>(p, xs, tok0, last_tok, q * XW, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (F8) {
#pragma unroll
            for (int s = 0; s < XW / 64; ++s) xf8[q * (XW / 64) + s] = lds_frag32(xs8 + c * XLDW + 64 * s + 32 * h);
        } else {
#pragma unroll
            for (int s = 0; s < XW / 16; ++s) xf[q * (XW / 16) + s] = *reinterpret_cast<const bf16x8*>(&xs[c * XLDW + 16 * s + 8 * h]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (p.stamps) t_x = __builtin_amdgcn_s_memtime();

    const int tok = tok0 + c;
    const bool valid = tok <= last_tok;
    const int tokc = valid ? tok : last_tok;          // clamped lanes recompute the last row bit for bit (see above)
    const long yrow = p.y_row_group ? (long)tokc + tokc / p.y_row_group + 1 : (long)tokc;
    const long ybase = yrow * p.ldy + hoff;
    const long mbase = (long)tokc * p.ldref + hoff;
    const uint64_t dbase = (uint64_t)tokc * p.drop_ld;
    const float floor_ = p.act_relu ? 0.f : -__builtin_inff();
    const bool drop_on = (EPI == EPI_DROP || (EPI == EPI_ANY && p.drop.p > 0.f)) && !TLIN_DBG(32);
    const bool mask_on = EPI == EPI_MASK || (EPI == EPI_ANY && p.mask_ref != nullptr);
    const bool acc_on = EPI == EPI_ANY && !YB && p.accumulate;
    const float ksd = drop_on ? 1.f / (1.f - p.drop.p) : 1.f;
    constexpr bool PRED = EPI == EPI_ANY;

    // bf16 results are held back for CB = 4 feature chunks and written together: per token 4 x 64 contiguous bytes in
    // one burst (full 128-byte lines reach L2 / HBM at once).  Written chunk by chunk, a row's 64-byte pieces arrive
    // microseconds apart and the write stream measured 3.5 TB/s against 5.7 for the burst shape (tools/bw_probe).
    constexpr int CB = GG_STR_CB;
    unsigned held[CB][8];
    auto tile = [&](auto slot_tag, int nt) {
        constexpr int slot = decltype(slot_tag)::value;
        constexpr int buf = slot & 1;
        if (nt + 2 < ntiles && !TLIN_DBG(4)) load_chunk(buf, nt + 2);
        // operands of this tile's epilogue are requested before the MFMA chain
        f32x4 mm[4], yy[4];
        if (mask_on) {
            if (p.mask_bf16) {     // only the sign matters: expand the bf16 values to fp32 bit patterns
                if constexpr (PERM) {
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(p.mask_ref) + mbase + nt * 32 + 8 * g2);
                        mm[2 * g2] = f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xffff0000u),
                                           __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xffff0000u)};
                        mm[2 * g2 + 1] = f32x4{__builtin_bit_cast(float, r[2] << 16), __builtin_bit_cast(float, r[2] & 0xffff0000u),
                                               __builtin_bit_cast(float, r[3] << 16), __builtin_bit_cast(float, r[3] & 0xffff0000u)};
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.mask_ref) + mbase + nt * 32 + 8 * g);
                        mm[g] = f32x4{__builtin_bit_cast(float, r[0] << 16), __builtin_bit_cast(float, r[0] & 0xffff0000u),
                                      __builtin_bit_cast(float, r[1] << 16), __builtin_bit_cast(float, r[1] & 0xffff0000u)};
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) mm[g] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.mask_ref) + mbase + nt * 32 + GS * g);
            }
        }
        if (acc_on) {
#pragma unroll
            for (int g = 0; g < 4; ++g) yy[g] = *reinterpret_cast<const f32x4*>(reinterpret_cast<float*>(p.Y) + ybase + nt * 32 + GS * g);
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if constexpr (F8) {
            const unsigned char* wsb8 = Wsb + buf * 32 * WLD + arow * WLD + 32 * h;
#pragma unroll
            for (int s = 0; s < KSL / 64; ++s)
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(lds_frag32(wsb8 + 64 * s), xf8[s], acc, 0, 0, 0, sc_w, 0, sc_x);
        } else {
        const __bf16* wsb = Ws + buf * 32 * WLD + arow * WLD + 8 * h;
#pragma unroll
        for (int s4 = 0; s4 < KSL / 64; ++s4) {
            bf16x8 wf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) wf[u] = *reinterpret_cast<const bf16x8*>(wsb + 16 * (4 * s4 + u));
            if (!TLIN_DBG(2)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u], xf[4 * s4 + u], acc, 0, 0, 0);
            }
        }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nt * 32 + hoff + GS * g;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[n]);
            f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            v += bb;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], floor_);
            if (drop_on) {
                float f[4];
                drop_factor4(dkey, dbase + n, ksd, f);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= f[j];
            }
            if (mask_on) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = mm[g][j] > 0.f ? v[j] * p.mask_scale : 0.f;
            }
            if (acc_on) v += yy[g];
            if constexpr (YB) {
                held[slot][2 * g] = pack2(v[0], v[1]);
                held[slot][2 * g + 1] = pack2(v[2], v[3]);
            } else {
                if ((!PRED || valid) && !(TLIN_DBG(1) && v[0] != 1234.5f))
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.Y) + ybase + nt * 32 + GS * g) = v;
            }
        }
        if constexpr (YB) {     // 16 consecutive features of this lane's row per chunk: two 16-byte stores each
            if (slot == CB - 1 || nt + 1 == ntiles) {
                if ((!PRED || valid) && !(TLIN_DBG(1) && held[0][0] != 0x12345u)) {
                    __bf16* yp = reinterpret_cast<__bf16*>(p.Y) + ybase + (nt - slot) * 32;
#pragma unroll
                    for (int q = 0; q <= slot; ++q) {
                        if constexpr (PERM) {
                            *reinterpret_cast<u32x4*>(yp + 32 * q) = u32x4{held[q][0], held[q][1], held[q][2], held[q][3]};
                            *reinterpret_cast<u32x4*>(yp + 32 * q + 8) = u32x4{held[q][4], held[q][5], held[q][6], held[q][7]};
                        } else {
#pragma unroll
                            for (int g = 0; g < 4; ++g) *reinterpret_cast<u32x2*>(yp + 32 * q + GS * g) = u32x2{held[q][2 * g], held[q][2 * g + 1]};
                        }
                    }
                }
            }
        }
        if (nt + 1 < ntiles) store_chunk(buf ^ 1);
        __syncthreads();
    };
    for (int nt = 0; nt < ntiles; nt += CB) {
        tile(std::integral_constant<int, 0>{}, nt);
        if (nt + 1 < ntiles) tile(std::integral_constant<int, 1>{}, nt + 1);
        if (p.stamps && nt == 0) t_it0 = __builtin_amdgcn_s_memtime();
        if constexpr (CB > 2) {
            if (nt + 2 < ntiles) tile(std::integral_constant<int, 2>{}, nt + 2);
            if (nt + 3 < ntiles) tile(std::integral_constant<int, 3>{}, nt + 3);
        }
    }
    if (p.stamps && tid == 0) {
        unsigned long long* o = p.stamps + 4 * (long)blockIdx.x;
        o[0] = t_start; o[1] = t_x; o[2] = t_it0; o[3] = __builtin_amdgcn_s_memtime();
    }
}

// one workgroup column per 2-D parameter: wb = bf16(W) [rows][cols], wtb = bf16(W^T) [cols][rows].  32 x 32 tiles through LDS so
// that both images are written in full 64-byte rows (the element-wise version wrote the transposed image one bf16 at a stride
// of `rows`: 36 us per call, after every optimiser step and in front of the next forward pass).
__global__ void shadow_kernel(const float* __restrict__ w, __bf16* __restrict__ wb, __bf16* __restrict__ wtb,
                              const ShadowEntry* __restrict__ tab) {
    __shared__ float t[32][33];
    const ShadowEntry e = tab[blockIdx.y];
    const int tiles_c = (e.cols + 31) / 32, ntile = ((e.rows + 31) / 32) * tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 256 threads: 8 tile rows per pass
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int r0 = (tile / tiles_c) * 32, c0 = (tile % tiles_c) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            const bool in = r < e.rows && c < e.cols;
            const float v = w[e.off + (long)min(r, e.rows - 1) * e.cols + min(c, e.cols - 1)];      // unconditional (clamped): the four loads fly together
            t[ty + 8 * k][tx] = v;
            if (in) wb[e.off + (long)r * e.cols + c] = (__bf16)v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < e.rows && c < e.cols) wtb[e.off + (long)c * e.rows + r] = (__bf16)t[tx][ty + 8 * k];
        }
        __syncthreads();
    }
}

// bf16x3 mode: fp32 W^T [cols][rows] at the same offsets (the data-gradient Linears read the transposed master weights)
__global__ void shadow_t32_kernel(const float* __restrict__ w, float* __restrict__ wt, const ShadowEntry* __restrict__ tab) {
    __shared__ float t[32][33];
    const ShadowEntry e = tab[blockIdx.y];
    const int tiles_c = (e.cols + 31) / 32, ntile = ((e.rows + 31) / 32) * tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int r0 = (tile / tiles_c) * 32, c0 = (tile % tiles_c) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            t[ty + 8 * k][tx] = w[e.off + (long)min(r, e.rows - 1) * e.cols + min(c, e.cols - 1)];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < e.rows && c < e.cols) wt[e.off + (long)c * e.rows + r] = t[tx][ty + 8 * k];
        }
        __syncthreads();
    }
}

// e4m3 shadow of the 2-D weights (forward orientation [rows][cols]) with one power-of-two scale per tensor:
// pass 1: amax[entry] = max |w| (the bit pattern of a non-negative float orders like an unsigned integer);
// pass 2: w_exp[entry] = floor(log2(448 / amax)) (the largest power of two that keeps the tensor inside the e4m3 range),
//         w8[2 * off + i] = e4m3(w[off + i] * 2^w_exp)   (byte offset 2 * off keeps every tensor 16-byte aligned)
__global__ void shadow8_amax_k(const float* __restrict__ w, unsigned* __restrict__ amax, const ShadowEntry* __restrict__ tab) {
    const ShadowEntry e = tab[blockIdx.y];
    const long n = (long)e.rows * e.cols;
    float m = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[e.off + i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(amax + blockIdx.y, __builtin_bit_cast(unsigned, m));
}
__global__ void shadow8_quant_k(const float* __restrict__ w, unsigned char* __restrict__ w8, const unsigned* __restrict__ amax,
                                int* __restrict__ w_exp, const ShadowEntry* __restrict__ tab) {
    const ShadowEntry e = tab[blockIdx.y];
    const float am = __builtin_bit_cast(float, amax[blockIdx.y]);
    int ex = am > 0.f ? ilogbf(448.f / am) : 0;
    ex = max(-24, min(24, ex));
    if (blockIdx.x == 0 && threadIdx.x == 0) w_exp[blockIdx.y] = ex;
    const float sc = exp2i(ex);
    const long n4 = ((long)e.rows * e.cols) / 4;          // slots are multiples of 8 elements
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + e.off + 4 * i);
        *reinterpret_cast<unsigned*>(w8 + 2 * e.off + 4 * i) = cvt4_fp8(v[0], v[1], v[2], v[3], sc);
    }
}

template <int NT_RES, int KSL, bool XB, int PRE>
int launch_res(const TlinP& p, hipStream_t st) {
    constexpr size_t smem = (size_t)6 * 32 * (KSL + 8) * 2 + (size_t)3 * 32 * NT_RES * 4;
    static bool attr_set = false;
    if (!attr_set) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tlin_res_kernel<NT_RES, KSL, XB, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const long blocks = (p.M + 127) / 128;
    launch_timed(tlin_res_kernel<NT_RES, KSL, XB, PRE>, dim3((unsigned)blocks), dim3(256), smem, st, p);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int NT_RES, int KSL, bool XB, int PRE, bool F8 = false>
int launch_res16(const TlinP& p, hipStream_t st) {
    constexpr size_t smem = (size_t)(2 * 32 + 4 * 16) * (F8 ? KSL + 16 : (KSL + 8) * 2) + (size_t)3 * 32 * NT_RES * 4;
    static bool attr_set = false;
    if (!attr_set) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tlin_res16_kernel<NT_RES, KSL, XB, PRE, F8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const long blocks = (p.M + 63) / 64;
    launch_timed(tlin_res16_kernel<NT_RES, KSL, XB, PRE, F8>, dim3((unsigned)blocks), dim3(256), smem, st, p);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
constexpr size_t STREAM_SMEM_MAX = 160 * 1024;
template <int KSL, bool XB, bool YB, int EPI, bool F8 = false>
int launch_str(const TlinP& p, hipStream_t st) {
    constexpr int XW = KSL < 128 ? KSL : 128;
    const size_t smem = F8 ? (size_t)2 * 32 * (KSL + 32) + (size_t)4 * 32 * (XW + 32) + (size_t)p.N * 4
                           : (size_t)2 * 32 * (KSL + 8) * 2 + (size_t)4 * 32 * (XW + 8) * 2 + (size_t)p.N * 4;
    static bool attr_set = false;
    if (!attr_set) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tlin_str_kernel<KSL, XB, YB, EPI, F8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)STREAM_SMEM_MAX));
        attr_set = true;
    }
    const long blocks = (p.M + 127) / 128;
    launch_timed(tlin_str_kernel<KSL, XB, YB, EPI, F8>, dim3((unsigned)blocks), dim3(256), smem, st, p);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int KSL, int EPI>
int launch_str_t(const TlinP& p, hipStream_t st) {
    if (p.x_bf16) return p.y_bf16 ? launch_str<KSL, true, true, EPI>(p, st) : launch_str<KSL, true, false, EPI>(p, st);
    return p.y_bf16 ? launch_str<KSL, false, true, EPI>(p, st) : launch_str<KSL, false, false, EPI>(p, st);
}
// the specialised epilogues exist for the production width (K = 256); other widths take the run-time one
// fp8 operands: the forward Linears of the production width (see tlin_fp8_supported)
inline int GG_FAIL_FP8() { set_error("tlin: fp8 instantiation missing"); return -2; }
int launch_fp8(const TlinP& p, hipStream_t st) {
    if (p.ln_g) return launch_res16<8, 256, true, PRE_RES, true>(p, st);          // out-proj / FFN2 + residual + LayerNorm
    if (p.x_bf16) return GG_FAIL_FP8();
    if (p.drop.p > 0.f) return launch_str<256, false, true, EPI_DROP, true>(p, st);   // FFN1 (+ ReLU + dropout)
    return launch_str<256, false, true, EPI_BIAS, true>(p, st);                       // QKV, FFN1 without dropout
}
int launch_str_256(const TlinP& p, hipStream_t st) {
    if (p.accumulate || (p.mask_ref && p.drop.p > 0.f)) return launch_str_t<256, EPI_ANY>(p, st);
    if (p.mask_ref) return launch_str_t<256, EPI_MASK>(p, st);
    if (p.drop.p > 0.f) return launch_str_t<256, EPI_DROP>(p, st);
    return launch_str_t<256, EPI_BIAS>(p, st);
}
template <int NT_RES, int KSL, int PRE>
int launch_res_t(const TlinP& p, hipStream_t st) {
    return p.x_bf16 ? launch_res<NT_RES, KSL, true, PRE>(p, st) : launch_res<NT_RES, KSL, false, PRE>(p, st);
}
int launch_res_256(const TlinP& p, hipStream_t st) {
    static const bool v32 = getenv("GG_TLIN_RES32") != nullptr;
    if (p.x_bf16 && !v32) {
        if (p.res && p.accumulate) return launch_res16<8, 256, true, PRE_ANY>(p, st);
        if (p.res) return launch_res16<8, 256, true, PRE_RES>(p, st);
        if (p.accumulate) return launch_res16<8, 256, true, PRE_ACC>(p, st);
        return launch_res16<8, 256, true, PRE_NONE>(p, st);
    }
    if (p.res && p.accumulate) return launch_res_t<8, 256, PRE_ANY>(p, st);
    if (p.res) return launch_res_t<8, 256, PRE_RES>(p, st);
    if (p.accumulate) return launch_res_t<8, 256, PRE_ACC>(p, st);
    return launch_res_t<8, 256, PRE_NONE>(p, st);
}
inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
inline bool needs_resident(const TlinP& p) { return p.ln_g || p.res || !(p.K == 64 || p.K == 128 || p.K == 256); }
}  // namespace

// supported instantiations: stream K in {64,128,256} (any N % 32 == 0);
// resident (N, K-slice) in {(256, 256), (128, 128), (64, 64)} with K a multiple of the slice
bool wst_routed(const TlinP& p);
bool tlin_supported(const TlinP& p) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return false;
    if (p.lnb_dres) return wst_routed(p) && wst_kind(p) == 10;        // += then LayerNorm backward: wst.hip only
    if (p.N % 32 || p.K % 64) return false;
    if (!al16(p.X) || !al16(p.W) || !al16(p.Y) || p.ldx % (p.x_bf16 ? 8 : 4) || p.ldy % 4 || p.ldw % 8) return false;
    if (p.bias && !al16(p.bias)) return false;
    if (p.film_g && (p.x_bf16 || !al16(p.film_g) || !al16(p.film_b) || p.film_ld % 4 || p.film_group < 32)) return false;
    if (p.mask_ref && (!al16(p.mask_ref) || p.ldref % 4)) return false;
    if (p.res && (!al16(p.res) || p.ldres % 4)) return false;
    if (p.ln_g && (!al16(p.ln_g) || !al16(p.ln_b) || !al16(p.ln_y))) return false;
    if (p.y_bf16 && p.accumulate) return false;
    if (p.ln_g && (p.res_bf16 || p.ln_y_bf16 || p.y_bf16))       // bf16 residual / LN output / pre-LN sum: wst.hip only
        return p.fp8 ? wst_fp8_kind(p) != 0 : (wst_routed(p) && wst_ln_supported(p));
    if (p.drop.p > 0.f && p.drop_ld % 2) return false;      // the epilogues hash element PAIRS (drop_rng.h)
    if (!needs_resident(p)) return (size_t)p.N * 4 <= 64 * 1024;     // bias vector in LDS
    if (p.mask_ref || p.act_relu || p.y_bf16) return false;   // not implemented in the resident epilogue
    if (p.N == 256) return p.K % 256 == 0;
    if (p.N == 128) return p.K % 128 == 0;
    if (p.N == 64) return p.K % 64 == 0;
    return false;
}

// which kernel tlin() launches for p (profiling classes follow the kernels' own names): 0 tlin_str_kernel, 1 tlin_res_kernel
// (32-token waves), 2 tlin_res16_kernel<..., PRE_RES> (+ residual + LayerNorm), 3 <..., PRE_ACC> (+=), 4 other res16 modes
static thread_local int g_route = 0;      // tests (gg_test_linear, on the calling thread only): 1 = keep the token-on-lane kernels
void tlin_force_route(int route) { g_route = route; }
bool wst_routed(const TlinP& p) {
    if (g_route == 1) return false;
    static const bool off = getenv("GG_NO_WST") != nullptr;
    static const bool off2 = getenv("GG_NO_WST2") != nullptr;
    if (off || p.fp8) return false;
    return wst_ln_supported(p) || (!off2 && wst_kind(p) != 0);
}
// classes >= 32: kernels outside the tlin_str_kernel<256, XB, YB, EPI> family (engine.hip try_tlin names them)
int tlin_kernel_class(const TlinP& p) {
    if (wst_routed(p)) return wst_ln_supported(p) ? (p.K == 256 ? 32 : 33) : (wst_kind(p) >= 6 ? 40 + wst_kind(p) : 33 + wst_kind(p));      // 32 .. 38, 46
    if (p.fp8) {
        static const bool no_wst8 = getenv("GG_NO_WST") != nullptr || getenv("GG_NO_WST8") != nullptr;
        if (!no_wst8 && wst_fp8_kind(p)) return (wst_fp8_kind(p) >= 5 ? 46 : 41) + wst_fp8_kind(p);        // 42 .. 45, 51, 52
        return p.ln_g ? 39 : (p.drop.p > 0.f ? 41 : 40);
    }
    if (!needs_resident(p)) {
        if (p.K != 256) return 0;
        // 16 + the <XB, YB, EPI> instantiation launch_str_256 picks: bit 0 XB, bit 1 YB, bits 2..3 EPI
        int epi = EPI_BIAS;
        if (p.accumulate || (p.mask_ref && p.drop.p > 0.f)) epi = EPI_ANY;
        else if (p.mask_ref) epi = EPI_MASK;
        else if (p.drop.p > 0.f) epi = EPI_DROP;
        return 16 + (p.x_bf16 ? 1 : 0) + (p.y_bf16 ? 2 : 0) + 4 * epi;
    }
    static const bool v32 = getenv("GG_TLIN_RES32") != nullptr;
    if (p.N != 256 || !p.x_bf16 || v32) return 1;
    if (p.res && !p.accumulate) return 2;
    if (p.accumulate && !p.res) return 3;
    return 4;
}

void tlin_time_next(hipEvent_t begin, hipEvent_t end) { g_ev0 = begin; g_ev1 = end; }
bool wst_routed(const TlinP& p);

// fp8 operand path: the four forward Linears of an encoder layer at the production width - stream (QKV, FFN1: fp32 X,
// bf16 Y, K = 256, bias (+ReLU) (+dropout)) and resident + residual + LayerNorm (out-proj, FFN2: bf16 X, N = 256, K % 256 == 0)
bool tlin_fp8_supported(const TlinP& p) {
    if (!p.fp8 || !p.w_exp || !tlin_supported(p)) return false;
    if (p.film_g || p.mask_ref || p.accumulate || p.y_row_group) return false;
    if (p.ln_g) return p.N == 256 && p.K % 256 == 0 && p.x_bf16 && p.res && !p.act_relu && (!p.y_bf16 || wst_fp8_kind(p) != 0);
    return p.K == 256 && (!p.x_bf16 || wst_fp8_kind(p) != 0) && p.y_bf16 && !p.res;
}

int tlin(const TlinP& p_in, hipStream_t st) {
    GG_REQUIRE(tlin_supported(p_in), "tlin: unsupported shape / alignment");
    static const int dbg = getenv("GG_TLIN_DBG") ? atoi(getenv("GG_TLIN_DBG")) : 0;
    TlinP p = p_in;
    p.dbg = dbg;
    if (p.fp8) {
        static const bool no_wst8 = getenv("GG_NO_WST") != nullptr || getenv("GG_NO_WST8") != nullptr;
        if (!no_wst8 && wst_fp8_kind(p)) {
            hipEvent_t a = g_ev0, b = g_ev1;
            g_ev0 = g_ev1 = nullptr;
            return wst_fp8(p, st, a, b);
        }
        GG_REQUIRE(tlin_fp8_supported(p), "tlin: shape has no fp8 instantiation");
        return launch_fp8(p, st);
    }
    if (wst_routed(p)) {
        hipEvent_t a = g_ev0, b = g_ev1;
        g_ev0 = g_ev1 = nullptr;
        return wst_ln_supported(p) ? wst_ln(p, st, a, b) : wst_other(p, st, a, b);
    }
    if (!needs_resident(p)) {
        if (p.K == 256) return launch_str_256(p, st);
        if (p.K == 128) return launch_str_t<128, EPI_ANY>(p, st);
        return launch_str_t<64, EPI_ANY>(p, st);
    }
    if (p.N == 256) return launch_res_256(p, st);
    if (p.N == 128) return launch_res_t<4, 128, PRE_ANY>(p, st);
    return launch_res_t<2, 64, PRE_ANY>(p, st);
}

int k_shadow_weights_fp8(const float* w, void* w8, unsigned* amax, int* w_exp, const ShadowEntry* tab_dev, int n_entries, hipStream_t st) {
    if (n_entries <= 0) return 0;
    GG_CHECK_HIP(hipMemsetAsync(amax, 0, sizeof(unsigned) * n_entries, st));
    shadow8_amax_k<<<dim3(64, n_entries), 256, 0, st>>>(w, amax, tab_dev);
    shadow8_quant_k<<<dim3(64, n_entries), 256, 0, st>>>(w, reinterpret_cast<unsigned char*>(w8), amax, w_exp, tab_dev);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

int k_shadow_weights_t32(const float* w, float* wt, const ShadowEntry* tab_dev, int n_entries, hipStream_t st) {
    if (n_entries <= 0) return 0;
    shadow_t32_kernel<<<dim3(64, n_entries), 256, 0, st>>>(w, wt, tab_dev);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

int k_shadow_weights(const float* w, void* wb, void* wtb, const ShadowEntry* tab_dev, int n_entries, hipStream_t st) {
    if (n_entries <= 0) return 0;
    shadow_kernel<<<dim3(256, n_entries), 256, 0, st>>>(w, reinterpret_cast<__bf16*>(wb), reinterpret_cast<__bf16*>(wtb), tab_dev);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gg
