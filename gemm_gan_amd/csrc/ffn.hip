// Fused position-wise feed-forward block of an encoder layer, forward (torch nn/modules/transformer.py:961-983 via R:213):
//
//   h  = drop(relu(x1 W1^T + b1))              [tok, F = 512]   kept (bf16) only for the rows whose backward will run
//   r2 = x1 + drop(h W2^T + b2)                [tok, E = 256]   kept (fp32) only for those rows
//   x2 = LayerNorm(r2) * gamma + beta          [tok, E]
//
// in ONE kernel: the hidden activations never make the HBM round trip (FFN1 wrote 1 KB per token that FFN2 read back, and FFN2
// re-read the 1 KB residual row FFN1 had just read: 5.5 KB per token and layer in two launches, 2 - 4 KB here).
//
// Token-on-lane orientation (tlin.hip): a wave owns 32 tokens, their x1 rows are register-resident MFMA B fragments, accumulators
// have features in registers and the token on the lane.  The hidden dimension is walked in 16 chunks of 32 features; per chunk
//   acc1 [32 hidden x 32 tok] = W1chunk x^T          16 MFMAs, A = 32 rows of W1 from LDS
//   bias, ReLU, dropout in registers; the accumulator tile IS the B operand of the second product ("accumulator as the next
//   MFMA's operand"): the W1 rows of a chunk are fed in an order (bits 2 and 3 of the row index swapped) that makes register
//   8 s2 + j of lane half hh hold hidden feature 16 s2 + 8 hh + j - eight consecutive k of k-step s2, which is also a 16-byte
//   run of the stored h row;
//   acc2 [256 out x 32 tok] += W2[:, chunk] hchunk    16 MFMAs, A = the 32 rows of W2^T (bf16 transposed shadow) of this chunk from
//   LDS through ds_read_b64_tr_b16 (natural k order).
// 32 MFMAs per chunk and workgroup barrier; the two weight chunks (16 KB each) travel L2 -> registers -> LDS one chunk ahead.
// Persistent workgroups (one per CU, grid = 91 % of the CUs like the weight-stationary Linears) loop over 128-token tiles, the
// chunk pipeline runs on across tile boundaries.  Same dropout streams (sites 2 and 3, element index = token * width + feature)
// and the same arithmetic as the two-launch route (tests/test_kernels_gpu.py compares this kernel with float64,
// tests/test_engine_oracle_gpu.py the two routes inside a critic iteration).
//
// STATUS: correct, but NOT the default.  At the headline shape (197 376 token rows per launch) it takes 346 us against 288 us for
// FFN1 + FFN2 although it moves 0.4 GB less: 496 registers per lane mean one wave per SIMD, so nothing covers the wave's own VALU
// work (368 VALU instructions per chunk: bias, ReLU, dropout hash, bf16 packing) or its waits - MFMA busy 14.6 %, 43 % of the wave
// cycles waiting (profiles/r03_ffn_fused.md).  The engine uses it only when asked to (GG_FFN_FUSED=1 / gg_set_ffn_fused).
#include "kernels.h"
#include "drop_rng.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace gg {
namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
constexpr float LN_EPS = 1e-5f;
constexpr int FE = 256, FF = 512, NCH = FF / 32;
constexpr int XLD = FE + 8, W1LD = FE + 8, W2LD = FE + 32;        // bf16 per LDS row (W2LD: 144 dwords = 16 mod 64: conflict-free transposing reads)
constexpr int W1IMG = 32 * W1LD, W2IMG = 32 * W2LD, XIMG = 32 * XLD;
constexpr size_t FFN_SMEM = (size_t)(2 * W1IMG + 2 * W2IMG + 4 * XIMG) * 2 + (size_t)(FF + 3 * FE) * 4;

// A fragment of k-step s2 for the 32 output features starting at col0, from a row-major [k][feature] image:
// element j of lane (c, hh) = img[16 * s2 + 8 * hh + j][col0 + c]
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

hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;

__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(const FfnP p) {
    const DropKey dk1 = drop_live(p.drop1), dk2 = drop_live(p.drop2);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* const W1s = reinterpret_cast<__bf16*>(smem_raw);                // [2][32][W1LD]   rows in the permuted order
    __bf16* const W2s = W1s + 2 * W1IMG;                                    // [2][32 k][W2LD] rows = hidden features of the chunk
    __bf16* const Xs = W2s + 2 * W2IMG;                                     // [4 waves][32][XLD]
    float* const Ps = reinterpret_cast<float*>(Xs + 4 * XIMG);              // b1 [512] | b2 | gamma | beta [256 each]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const long ntiles = (p.M + 127) / 128;
    const int last_tok = (int)p.M - 1;
    __bf16* const xs = Xs + wave * XIMG;

    // ---- weight chunk pipeline: chunk q (q mod 16 = hidden features 32 q .. 32 q + 31) of W1 (rows permuted: LDS row r holds
    // weight row r with bits 2 and 3 swapped) and of W2^T
    u32x4 wreg[8];             // one chunk in flight: requested when the previous chunk's products start (~1 000 MFMA cycles of cover)
    const __bf16* W1p = reinterpret_cast<const __bf16*>(p.W1);
    const __bf16* W2p = reinterpret_cast<const __bf16*>(p.W2T);
    // per-thread element offsets of its four pieces inside a chunk (chunk q: + 32 * FE * q); issue order W1, W2, (next-tile rows):
    // a wait for the W1 pieces must not also wait for younger loads
    int w1off[4], w2off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i, row = f >> 5, pc = f & 31;
        const int prow = (row & 19) | ((row & 4) << 1) | ((row & 8) >> 1);
        w1off[i] = prow * FE + 8 * pc;
        w2off[i] = row * FE + 8 * pc;
    }
    auto load_chunk = [&](int q) {
        const __bf16* a = W1p + (long)q * (32 * FE);
        const __bf16* b = W2p + (long)q * (32 * FE);
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[i] = *reinterpret_cast<const u32x4*>(a + w1off[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[4 + i] = *reinterpret_cast<const u32x4*>(b + w2off[i]);
    };
    // W1's rows of chunk q are multiplied in iteration q: written to LDS at the end of iteration q - 1.  W2's rows of chunk q are
    // multiplied in iteration q + 1 (the second product runs one chunk behind, see below): written at the START of iteration q into
    // the buffer that held chunk q - 2 - the one nobody reads in iteration q (chunk q - 1 is being read from the other buffer).
    auto store_w1 = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, row = f >> 5, pc = f & 31;
            *reinterpret_cast<u32x4*>(W1s + buf * W1IMG + row * W1LD + 8 * pc) = wreg[i];
        }
    };
    auto store_w2 = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, row = f >> 5, pc = f & 31;
            *reinterpret_cast<u32x4*>(W2s + buf * W2IMG + row * W2LD + 8 * pc) = wreg[4 + i];
        }
    };
    long my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_tiles == 0) return;
    long left = my_tiles * NCH;                 // chunks this workgroup still has to multiply
    load_chunk(0);
    for (int i = tid; i < FF; i += 256) Ps[i] = p.b1 ? p.b1[i] : 0.f;
    for (int i = tid; i < FE; i += 256) {
        Ps[FF + i] = p.b2 ? p.b2[i] : 0.f;
        Ps[FF + FE + i] = p.ln_g[i];
        Ps[FF + 2 * FE + i] = p.ln_b[i];
    }
    store_w1(0);

    const bool drop_on = p.drop1.p > 0.f;
    const float ksd = drop_on ? 1.f / (1.f - p.drop1.p) : 1.f;
    const unsigned char* const Xc = reinterpret_cast<const unsigned char*>(p.X);
    const unsigned lcolb = (unsigned)lane * 16u;            // 64 lanes x 4 floats = one 256-float row per load instruction

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tok0 = (int)(tile * 128) + wave * 32;
        const int tok = tok0 + c;
        const bool valid = tok <= last_tok;
        const int tokc = valid ? tok : last_tok;           // clamped lanes recompute the last row bit for bit: their stores are duplicates
        const bool keep = p.keep_rows < 0 || tokc < p.keep_rows;
        // ---- this wave's 32 x1 rows: fp32 -> bf16 slab -> register-resident B fragments (16 k-steps).  Only the FIRST tile of a
        // workgroup stages here; the rows of every later tile were written to the slab two per chunk while the previous tile was
        // multiplied (the slab is free once the fragments are in registers): no exposed memory round trip between tiles.
        if (tile == (long)blockIdx.x) {
#pragma unroll
            for (int b0 = 0; b0 < 32; b0 += 8) {
                f32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(Xc + ((unsigned)min(tok0 + b0 + i, last_tok) * (unsigned)(FE * 4) + lcolb));
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x2*>(&xs[(b0 + i) * XLD + 4 * lane]) = u32x2{pack2(v[i][0], v[i][1]), pack2(v[i][2], v[i][3])};
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bf16x8 xf[FE / 16];
#pragma unroll
        for (int s = 0; s < FE / 16; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(&xs[c * XLD + 16 * s + 8 * h]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __syncthreads();        // chunk 0 of this tile is in buffer 0 (and Ps on the first tile)

        f32x16 acc2[FE / 32];
#pragma unroll
        for (int nt = 0; nt < FE / 32; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[nt][i] = 0.f;
        f32x4 pre[4][4];                                   // residual rows, four feature tiles ahead of their use in the epilogue
        const float* const resp = p.X + (long)tokc * FE;
        const uint64_t d1base = (uint64_t)tokc * FF;
        __bf16* const hrow = reinterpret_cast<__bf16*>(p.Hs) + (long)tokc * FF;
        const int ntok0 = tok0 + (int)gridDim.x * 128;     // this wave's rows of the workgroup's next tile
        const bool has_next = tile + gridDim.x < ntiles;
        f32x4 nx[2][2];
        bf16x8 pfp[2];                                     // the previous chunk's hidden tile as B fragments, its second product still to run
        bool have_prev = false;
        auto chunk_body = [&](int ch, auto buf_tag) {
            constexpr int buf = decltype(buf_tag)::value;  // LDS buffer and register set of this chunk (the pipeline is continuous across tiles)
            store_w2(buf);
            if (left > 1) load_chunk((ch + 1) % NCH);
            // two rows of the workgroup's next tile per chunk, written to the slab ONE CHUNK LATER (two register sets): an HBM round
            // trip has a whole chunk of cover, and being the youngest loads they never stand between a wait and the weight pieces
            if (has_next) {
#pragma unroll
                for (int i = 0; i < 2; ++i) nx[buf][i] = *reinterpret_cast<const f32x4*>(Xc + ((unsigned)min(ntok0 + 2 * ch + i, last_tok) * (unsigned)(FE * 4) + lcolb));
            }
            // ---- first product: this chunk's 32 hidden features x 32 tokens
            f32x16 acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
            const __bf16* w1b = W1s + buf * W1IMG + c * W1LD + 8 * h;
#pragma unroll
            for (int s4 = 0; s4 < FE / 64; ++s4) {
                bf16x8 wf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) wf[u] = *reinterpret_cast<const bf16x8*>(w1b + 16 * (4 * s4 + u));
#pragma unroll
                for (int u = 0; u < 4; ++u) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u], xf[4 * s4 + u], acc1, 0, 0, 0);
            }
            // ---- software pipeline inside the wave: the SECOND product of the previous chunk (16 MFMAs on eight independent
            // accumulators, operands pfp[] in registers and W2's previous chunk still in the other LDS buffer) is issued in four
            // groups between the four VALU groups of THIS chunk's bias / ReLU / dropout / packing - which depend on the first
            // product just issued.  Without it the wave ran MFMA chain -> ~200 VALU instructions -> MFMA chain strictly in turn.
            // Register 8 s2 + j of half h = hidden feature 32 ch + 16 s2 + 8 h + j.
            const __bf16* w2p = W2s + (buf ^ 1) * W2IMG;
            bf16x8 pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    if (have_prev) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int nt = 4 * s2 + 2 * jj + u;
                            const bf16x8 a0 = frag_tr(w2p, W2LD, 32 * nt, 0, lane), a1 = frag_tr(w2p, W2LD, 32 * nt, 1, lane);
                            acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, pfp[0], acc2[nt], 0, 0, 0);
                            acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, pfp[1], acc2[nt], 0, 0, 0);
                        }
                    }
                    const int f0 = 32 * ch + 16 * s2 + 8 * h + 4 * jj;
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[f0]);
                    float fac[4] = {1.f, 1.f, 1.f, 1.f};
                    if (drop_on) drop_factor4(dk1, d1base + f0, ksd, fac);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[4 * jj + j] = fmaxf(acc1[8 * s2 + 4 * jj + j] + bb[j], 0.f) * fac[j];
                }
                const u32x4 w = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
                pf[s2] = __builtin_bit_cast(bf16x8, w);
                if (keep) *reinterpret_cast<u32x4*>(hrow + 32 * ch + 16 * s2 + 8 * h) = w;
            }
            pfp[0] = pf[0]; pfp[1] = pf[1];
            have_prev = true;
            if (left > 1) store_w1(buf ^ 1);
            if (has_next && ch > 0) {      // the rows requested in the previous chunk
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    *reinterpret_cast<u32x2*>(&xs[(2 * (ch - 1) + i) * XLD + 4 * lane]) =
                        u32x2{pack2(nx[buf ^ 1][i][0], nx[buf ^ 1][i][1]), pack2(nx[buf ^ 1][i][2], nx[buf ^ 1][i][3])};
            }
            --left;
            __syncthreads();
        };
        auto load_pre = [&](int nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[nt & 3][g] = *reinterpret_cast<const f32x4*>(resp + nt * 32 + 8 * g + 4 * h);
        };
#pragma unroll 1
        for (int ch = 0; ch < NCH; ch += 2) {
            if (ch == NCH - 2) {                           // in flight during the last two chunks
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) load_pre(nt);
            }
            chunk_body(ch, std::integral_constant<int, 0>{});
            chunk_body(ch + 1, std::integral_constant<int, 1>{});
        }
        if (has_next) {     // the last two rows of the next tile (requested in the last chunk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                *reinterpret_cast<u32x2*>(&xs[(2 * (NCH - 1) + i) * XLD + 4 * lane]) =
                    u32x2{pack2(nx[1][i][0], nx[1][i][1]), pack2(nx[1][i][2], nx[1][i][3])};
        }
        {   // the last chunk's second product (its W2 rows sit in buffer 1 until the next tile's first chunk has been multiplied)
            const __bf16* w2p = W2s + ((NCH - 1) & 1) * W2IMG;
#pragma unroll
            for (int nt = 0; nt < FE / 32; ++nt) {
                const bf16x8 a0 = frag_tr(w2p, W2LD, 32 * nt, 0, lane), a1 = frag_tr(w2p, W2LD, 32 * nt, 1, lane);
                acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, pfp[0], acc2[nt], 0, 0, 0);
                acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, pfp[1], acc2[nt], 0, 0, 0);
            }
        }

        // ---- epilogue: bias, dropout, residual; pre-LN sum stored for the rows whose backward runs; LayerNorm; x2 stored
        const bool drop2_on = p.drop2.p > 0.f;
        const float ksd2 = drop2_on ? 1.f / (1.f - p.drop2.p) : 1.f;
        const uint64_t d2base = (uint64_t)tokc * FE;
        float* const rb = p.R2 + (long)tokc * FE;
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < FE / 32; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nt * 32 + 8 * g + 4 * h;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[FF + n]);
                f32x4 v = {acc2[nt][4 * g], acc2[nt][4 * g + 1], acc2[nt][4 * g + 2], acc2[nt][4 * g + 3]};
                v += bb;
                if (drop2_on) {
                    float f[4];
                    drop_factor4(dk2, d2base + n, ksd2, f);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= f[j];
                }
                v += pre[nt & 3][g];
                if (keep) *reinterpret_cast<f32x4*>(rb + n) = v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc2[nt][4 * g + j] = v[j];
                    sum += v[j];
                }
            }
            if (nt + 4 < FE / 32) load_pre(nt + 4);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float invn = 1.f / (float)FE;
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * invn;
        float var = 0.f;
#pragma unroll
        for (int nt = 0; nt < FE / 32; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float d = acc2[nt][i] - mean;
                var += d * d;
            }
        var += __shfl_xor(var, 32, 64);
        const float rstd = rsqrtf(var * invn + LN_EPS);
        float* const yb = p.Y + (long)tokc * FE;
#pragma unroll
        for (int nt = 0; nt < FE / 32; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nt * 32 + 8 * g + 4 * h;
                const f32x4 gg_ = *reinterpret_cast<const f32x4*>(&Ps[FF + FE + n]);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(&Ps[FF + 2 * FE + n]);
                f32x4 y;
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = (acc2[nt][4 * g + j] - mean) * rstd * gg_[j] + bt[j];
                *reinterpret_cast<f32x4*>(yb + n) = y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (h == 0 && valid && keep) {
            p.stats[2 * (long)tok] = mean;
            p.stats[2 * (long)tok + 1] = rstd;
        }
    }
}
inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
}  // namespace

void ffn_time_next(hipEvent_t begin, hipEvent_t end) { g_ev0 = begin; g_ev1 = end; }

bool ffn_fused_supported(const FfnP& p) {
    static const bool off = getenv("GG_NO_FFN_FUSED") != nullptr;
    if (off || p.E != FE || p.F != FF || p.M < 1 || (double)p.M * FF >= 2.0e9) return false;
    if (!p.X || !p.W1 || !p.W2T || !p.Hs || !p.R2 || !p.ln_g || !p.ln_b || !p.Y || !p.stats) return false;
    if (!al16(p.X) || !al16(p.W1) || !al16(p.W2T) || !al16(p.Hs) || !al16(p.R2) || !al16(p.Y)) return false;
    return true;
}

int ffn_fused(const FfnP& p, hipStream_t st) {
    GG_REQUIRE(ffn_fused_supported(p), "ffn_fused: unsupported shape / alignment");
    static bool attr_set = false;
    static int n_cu = 0;
    if (!attr_set) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FFN_SMEM));
        int dev = 0;
        GG_CHECK_HIP(hipGetDevice(&dev));
        GG_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_set = true;
    }
    static const int cu_pct = getenv("GG_WST_CU_PCT") ? atoi(getenv("GG_WST_CU_PCT")) : 91;        // see wst.hip: room for the side-stream kernels
    const long cus = std::max<long>(8, (long)n_cu * cu_pct / 100 / 8 * 8);
    const long ntiles = (p.M + 127) / 128;
    const unsigned grid = (unsigned)std::min<long>(ntiles, cus);
    if (g_ev0) {
        hipExtLaunchKernelGGL(ffn_fused_kernel, dim3(grid), dim3(256), (unsigned)FFN_SMEM, st, g_ev0, g_ev1, 0, p);
        g_ev0 = g_ev1 = nullptr;
    } else {
        hipLaunchKernelGGL(ffn_fused_kernel, dim3(grid), dim3(256), FFN_SMEM, st, p);
    }
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gg
