// Fused tail of an encoder layer, forward (torch nn/modules/transformer.py:940-983 via R:213), round 4:
//
//   [ r1 = x + drop(ctx Wo^T + bo) ; x1 = LayerNorm1(r1) ]          (phase O, optional: the attention out-projection)
//   h  = drop(relu(x1 W1^T + b1))                  [tok, 512]      stored (bf16) only for the rows whose backward will run
//   r2 = x1 + drop(h W2^T + b2) ; x2 = LayerNorm2(r2)               r2 / statistics stored for those rows, x2 for every row
//
// in ONE launch.  The two-/three-launch route moved 6.1 KB per token row and layer through HBM for these products (5.1 KB for the
// forward-only replicas); this kernel moves 4.6 KB (1.5 KB): the hidden activations and x1 of a forward-only row never leave the chip.
//
// Structure (what round 3's ffn.hip lacked is marked *):
//   * token-on-lane: a wave owns NG groups of 32 tokens, their rows are register-resident MFMA B fragments; accumulators have the
//     feature in the register index and the token on the lane, so every epilogue (bias, ReLU, dropout, residual, LayerNorm) is
//     lane-local but for one cross-half shuffle, and an accumulator tile IS the next product's B operand after packing to bf16;
//   * the weights arrive as a STREAM of 1-KB MFMA A fragments in exactly the order the waves consume them: a per-layer bf16 image
//     in fragment order (enc_frag_kernel below, refreshed with the other shadow weights), so one wave-instruction of LDS-DMA
//     (global_load_lds_dwordx4: 64 lanes x 16 B, lane-linear on both sides) moves one fragment and a fragment read is one
//     conflict-free ds_read_b128 at an immediate offset - no address arithmetic, no transposing reads;
//   * (*) the stream runs through an LDS ring of NS 16-KB slots filled by LDS-DMA NS - 1 slots ahead (1.5 - 3 us of cover: ffn.hip
//     staged one 32-KB chunk ahead through registers, 0.4 us, and its waves spent 43 % of their time waiting); the DMA is inline
//     asm - invisible to the compiler's wait-count pass, which would otherwise drain it before every fragment read - and is retired
//     by a counted s_waitcnt vmcnt(K) followed by the slot's workgroup barrier (cdna_hip_programming.md 5.7);
//   * (*) NW = 8 waves x NG = 1 group at two waves per SIMD (<= 256 registers), or NW = 4 x NG = 2 at one wave per SIMD with
//     every fragment feeding two MFMAs; both sweep 256 tokens per pass over the 512-KB (640-KB with phase O) stream.
// One slot = 16 fragments = one k-sweep of 16 MFMAs per token group; slot order per hidden chunk q (32 hidden features):
// W1 rows of the chunk (16 k-steps), then the chunk's W2 columns (8 output tiles x 2 k-steps).
#include "kernels.h"
#include "drop_rng.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdlib>

namespace gg {
namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
constexpr float LN_EPS = 1e-5f;
constexpr int FE = 256, FF = 512, NCH = FF / 32;
constexpr int SLOT = 16384;                 // bytes per ring slot: 16 fragments of 1 KB
constexpr int FFN_SLOTS = 2 * NCH;          // 32 slots per sweep of the feed-forward stream
__host__ __device__ __forceinline__ int a_row_of_lane(int r) { return 16 * ((r >> 2) & 1) + (r & 3) + 4 * (r >> 3); }

// ---- fragment-ordered weight image -------------------------------------------------------------------------------------------
// piece (slot, f, lane) = the 16 bytes lane `lane` holds of fragment f of slot `slot`.  lane = (m, hh): A-operand row slot m (weight
// row a_row_of_lane(m) of the tile: accumulator register i of lane half hh is then feature 16 hh + i of the tile) and k half hh.
// The reduction index of a k-step is permuted the same way on both operands: k-step (t, j) of a 256-wide input covers features
// 32 t + 16 hh + 8 j + [0, 8) in lane half hh - the 8 consecutive values of an accumulator tile t that lane half hh owns.
//   even slot 2q  : W1 chunk q, fragment s = 2 t + j : W1[32 q + row][32 t + 16 hh + 8 j ..]
//   odd slot 2q+1 : W2 chunk q, fragment 8 j + t      : W2[32 t + row][32 q + 16 hh + 8 j ..]   (k-step major: the products with the first
//                   half of the hidden tile can start while the second half is still being packed)
struct FragTab { long w1[8], w2[8]; int nl; };
__global__ __launch_bounds__(256) void enc_frag_kernel(const float* __restrict__ w, FragTab tab, unsigned char* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;            // piece index inside the layer: < 32 * 16 * 64
    const int layer = blockIdx.y;
    const int lane = idx & 63, f = (idx >> 6) & 15, slot = idx >> 10;
    const int m = lane & 31, hh = lane >> 5, q = slot >> 1;
    const int t = (slot & 1) ? (f & 7) : (f >> 1), j = (slot & 1) ? (f >> 3) : (f & 1);
    const float* src = (slot & 1) ? w + tab.w2[layer] + (long)(32 * t + a_row_of_lane(m)) * FF + 32 * q + 16 * hh + 8 * j
                                  : w + tab.w1[layer] + (long)(32 * q + a_row_of_lane(m)) * FE + 32 * t + 16 * hh + 8 * j;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    *reinterpret_cast<u32x4*>(out + (size_t)layer * FFN_SLOTS * SLOT + (size_t)idx * 16) =
        u32x4{pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3])};
}

// one LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to LDS bytes [lds_dst, lds_dst + 1024), lane-linear.
// M0 carries the LDS address and is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;

template <int NW, int NG, int NS, bool DROP, bool STAMP = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void ffn2_kernel(const Ffn2P p) {
    constexpr int PPW = 16 / NW;                // 1-KB DMA pieces per wave and slot
    constexpr int UPS = NW * NG;                // 32-token units per sweep
    constexpr int CPI = NS / 2;                 // hidden chunks per unrolled loop body (ring positions are then compile-time constants)
    constexpr int NTH = 64 * NW;
    constexpr int PD = NW == 4 ? 6 : 3;                       // fragment reads in flight ahead of the MFMA that consumes them
    constexpr int KWAIT = PPW * (NS - 2);       // DMA pieces of this wave that may stay in flight when slot g is needed: those of g+1 .. g+NS-2
    static_assert(NS % 2 == 0 && NCH % CPI == 0 && (NS & (NS - 1)) == 0, "ring period must divide the chunk loop");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* const ring = smem_raw;                                           // [NS][SLOT]
    float* const Ps = reinterpret_cast<float*>(smem_raw + NS * SLOT);               // b1 [512] | b2 | gamma | beta [256 each]

    const DropKey dk1 = drop_live(p.drop1), dk2 = drop_live(p.drop2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const long n_units = (p.M + 31) / 32;
    const long G = gridDim.x, blk = blockIdx.x;
    if (blk >= n_units) return;
    const long n_b = (n_units - blk + G - 1) / G;              // units of this workgroup: blk, blk + G, ...
    const int n_sweeps = (int)((n_b + UPS - 1) / UPS);
    const int last_tok = (int)p.M - 1;

    for (int i = tid; i < FF; i += NTH) Ps[i] = p.b1 ? p.b1[i] : 0.f;
    for (int i = tid; i < FE; i += NTH) {
        Ps[FF + i] = p.b2 ? p.b2[i] : 0.f;
        Ps[FF + FE + i] = p.ln_g[i];
        Ps[FF + 2 * FE + i] = p.ln_b[i];
    }

    // STAMP (tools/ffn2_probe.py only): s_memtime deltas per phase, summed over the launch, per wave -> p.stamps[(block * NW + wave) * 8 + phase]
    // phases: 0 DMA wait (slot A), 1 barrier, 2 W1 products, 3 DMA wait (slot B), 4 barrier, 5 hidden-tile epilogue, 6 W2 products, 7 tile epilogue
    unsigned long long t_prev = 0;
    unsigned t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (k >= 0) t_acc[k] += (unsigned)(t - t_prev);
            t_prev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---- weight stream: slot gs (counted over the whole launch) holds stream slot gs mod 32 in ring position gs mod NS
    const unsigned ring_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)ring) + (unsigned)wave * (PPW * 1024);
    const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(p.Wf) + wave * (PPW * 1024) + lane * 16;
    auto dma_slot = [&](int gs) {
        const unsigned char* s = wsrc + (size_t)(gs & (FFN_SLOTS - 1)) * SLOT;
        const unsigned d = ring_lds + (unsigned)(gs & (NS - 1)) * SLOT;
#pragma unroll
        for (int i = 0; i < PPW; ++i) glds16(s + i * 1024, d + i * 1024);
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) dma_slot(i);

    // ---- token rows of a sweep: unit j of the workgroup's list -> wave, group
    const unsigned char* const Xp = reinterpret_cast<const unsigned char*>(p.X);
    bf16x8 xf[NG][16];
    auto unit_tok0 = [&](int sweep, int g) -> long {
        const long j = (long)sweep * UPS + g * NW + wave;
        return j < n_b ? (blk + j * G) * 32 : -1;
    };
    auto load_x = [&](int sweep) {          // unconditional, clamped (wst.hip: a conditional load costs the whole prefetch)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const long t0 = unit_tok0(sweep, g);
            const int tk = t0 < 0 ? last_tok : min((int)t0 + c, last_tok);
            const unsigned char* row = Xp + (size_t)tk * (FE * 2) + 32 * h;
#pragma unroll
            for (int f = 0; f < 16; ++f) xf[g][f] = *reinterpret_cast<const bf16x8*>(row + 64 * (f >> 1) + 16 * (f & 1));
        }
    };
    load_x(0);
    // Ps is written by all waves above and read (the output bias, into acc2) before the first barrier of the chunk loop: without this
    // one a wave that starts late leaves its share unwritten when an early wave reads it - seen once in ~6 runs of the full-size
    // kernel test as 2 % rel-L2 on one row block of the first pass
    __syncthreads();

    // DROP: both dropout sites live (p > 0); a run-time test here would put every hash behind a branch inside the MFMA loop
    constexpr bool drop1_on = DROP, drop2_on = DROP;
    const float ks1 = DROP ? 1.f / (1.f - p.drop1.p) : 1.f, ks2 = DROP ? 1.f / (1.f - p.drop2.p) : 1.f;
    int gs = 0;                                 // next slot to be consumed
    const unsigned char* const fbase = ring + lane * 16;

    for (int sweep = 0; sweep < n_sweeps; ++sweep) {
        int tokc[NG];
        bool valid[NG], keep[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const long t0 = unit_tok0(sweep, g);
            const int tk = (int)t0 + c;
            valid[g] = t0 >= 0 && tk <= last_tok;
            tokc[g] = valid[g] ? tk : last_tok;
            keep[g] = valid[g] && (p.keep_rows < 0 || tokc[g] < p.keep_rows);
        }
        f32x16 acc2[NG][FE / 32];           // start from the output bias
#pragma unroll
        for (int t = 0; t < FE / 32; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&Ps[FF + 32 * t + 16 * h + 4 * g4]);
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc2[g][t][4 * g4 + j] = bb[j];
            }

#pragma unroll 1
        for (int it = 0; it < NCH / CPI; ++it) {
#pragma unroll
            for (int cc = 0; cc < CPI; ++cc) {
                const int q = it * CPI + cc;
                // ---- slot A: the chunk's 32 rows of W1 against the register-resident rows of x1
                stamp(cc == 0 && it == 0 ? -1 : 6);
                wait_vm<KWAIT>();
                stamp(0);
                __syncthreads();
                stamp(1);
                dma_slot(gs + NS - 1);
                ++gs;
                f32x16 acc1[NG];
                {   // the accumulators start from the bias (one LDS read instead of 16 adds in the epilogue)
                    const float* bq = &Ps[32 * q + 16 * h];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 bb = *reinterpret_cast<const f32x4*>(bq + 4 * g4);
#pragma unroll
                        for (int g = 0; g < NG; ++g)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc1[g][4 * g4 + j] = bb[j];
                    }
                }
                {
                    const unsigned char* fa = fbase + (2 * cc) * SLOT;
                    bf16x8 a[PD + 1];
#pragma unroll
                    for (int s = 0; s < PD; ++s) a[s] = *reinterpret_cast<const bf16x8*>(fa + s * 1024);
                    __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);      // the PD reads that open the pipeline stay together, ahead of the first product
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        if (s + PD < 16) a[(s + PD) % (PD + 1)] = *reinterpret_cast<const bf16x8*>(fa + (s + PD) * 1024);
#pragma unroll
                        for (int g = 0; g < NG; ++g) acc1[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s % (PD + 1)], xf[g][s], acc1[g], 0, 0, 0);
                        if (s + PD < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one fragment read, then this step's products
                        __builtin_amdgcn_sched_group_barrier(0x008, NG, 0);
                    }
                }
                // ---- slot B: bias, ReLU, dropout -> the hidden tile as two B fragments; the chunk's columns of W2
                stamp(2);
                wait_vm<KWAIT>();
                stamp(3);
                __syncthreads();
                stamp(4);
                dma_slot(gs + NS - 1);
                ++gs;
                bf16x8 pf[NG][2];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const uint64_t d1 = (uint64_t)tokc[g] * FF + 32 * q + 16 * h;
                    unsigned pk[8];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        float fac[4] = {1.f, 1.f, 1.f, 1.f};
                        if (drop1_on) drop_factor4(dk1, d1 + 4 * g4, ks1, fac);
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc1[g][4 * g4 + j], 0.f) * fac[j];
                        pk[2 * g4] = pack2(v[0], v[1]);
                        pk[2 * g4 + 1] = pack2(v[2], v[3]);
                    }
                    const u32x4 w0 = {pk[0], pk[1], pk[2], pk[3]}, w1 = {pk[4], pk[5], pk[6], pk[7]};
                    pf[g][0] = __builtin_bit_cast(bf16x8, w0);
                    pf[g][1] = __builtin_bit_cast(bf16x8, w1);
                    if (keep[g]) {
                        unsigned char* hr = reinterpret_cast<unsigned char*>(p.Hs) + ((size_t)tokc[g] * FF + 32 * q + 16 * h) * 2;
                        *reinterpret_cast<u32x4*>(hr) = w0;
                        *reinterpret_cast<u32x4*>(hr + 16) = w1;
                    }
                }
                stamp(5);
                {
                    const unsigned char* fb = fbase + (2 * cc + 1) * SLOT;
                    bf16x8 a[PD + 1];
#pragma unroll
                    for (int f = 0; f < PD; ++f) a[f] = *reinterpret_cast<const bf16x8*>(fb + f * 1024);
                    __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
                    for (int f = 0; f < 16; ++f) {          // fragment 8 j + t
                        if (f + PD < 16) a[(f + PD) % (PD + 1)] = *reinterpret_cast<const bf16x8*>(fb + (f + PD) * 1024);
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc2[g][f & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[f % (PD + 1)], pf[g][f >> 3], acc2[g][f & 7], 0, 0, 0);
                        if (f + PD < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, NG, 0);
                    }
                }
            }
        }

        // ---- tile epilogue: bias, dropout, residual (the bf16 x1 rows still in registers); pre-LN sum; LayerNorm; x2
        stamp(6);
        float mean[NG], rstd[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint64_t d2 = (uint64_t)tokc[g] * FE + 16 * h;
            unsigned char* const r2b = reinterpret_cast<unsigned char*>(p.R2) + ((size_t)tokc[g] * FE + 16 * h) * (p.r2_bf16 ? 2 : 4);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < FE / 32; ++t) {
                unsigned rpk[8];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float fac[4] = {1.f, 1.f, 1.f, 1.f};
                    if (drop2_on) drop_factor4(dk2, d2 + 32 * t + 4 * g4, ks2, fac);
                    const u32x4 xw = __builtin_bit_cast(u32x4, xf[g][2 * t + (g4 >> 1)]);
                    const unsigned wa = xw[2 * (g4 & 1)], wb = xw[2 * (g4 & 1) + 1];
                    const float res[4] = {bf_lo(wa), bf_hi(wa), bf_lo(wb), bf_hi(wb)};
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc2[g][t][4 * g4 + j] * fac[j] + res[j];
                        acc2[g][t][4 * g4 + j] = v[j];
                        s1 += v[j];
                        s2 += v[j] * v[j];
                    }
                    if (!p.r2_bf16 && keep[g]) *reinterpret_cast<f32x4*>(r2b + (32 * t + 4 * g4) * 4) = v;
                    rpk[2 * g4] = pack2(v[0], v[1]);
                    rpk[2 * g4 + 1] = pack2(v[2], v[3]);
                }
                if (p.r2_bf16 && keep[g]) {
                    *reinterpret_cast<u32x4*>(r2b + 64 * t) = u32x4{rpk[0], rpk[1], rpk[2], rpk[3]};
                    *reinterpret_cast<u32x4*>(r2b + 64 * t + 16) = u32x4{rpk[4], rpk[5], rpk[6], rpk[7]};
                }
            }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            mean[g] = s1 * (1.f / FE);
            rstd[g] = rsqrtf(fmaxf(s2 * (1.f / FE) - mean[g] * mean[g], 0.f) + LN_EPS);
        }
        int tok_now[NG];
        bool valid_now[NG], keep_now[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { tok_now[g] = tokc[g]; valid_now[g] = valid[g]; keep_now[g] = keep[g]; }
        load_x(sweep + 1);                      // the residual has been added: the next sweep's rows travel under the LayerNorm arithmetic
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            unsigned char* const yb = reinterpret_cast<unsigned char*>(p.Y) + ((size_t)tok_now[g] * FE + 16 * h) * (p.y_bf16 ? 2 : 4);
#pragma unroll
            for (int t = 0; t < FE / 32; ++t) {
                unsigned ypk[8];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int n = 32 * t + 16 * h + 4 * g4;
                    const f32x4 gw = *reinterpret_cast<const f32x4*>(&Ps[FF + FE + n]);
                    const f32x4 bt = *reinterpret_cast<const f32x4*>(&Ps[FF + 2 * FE + n]);
                    f32x4 y;
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = (acc2[g][t][4 * g4 + j] - mean[g]) * rstd[g] * gw[j] + bt[j];
                    if (!p.y_bf16 && valid_now[g]) *reinterpret_cast<f32x4*>(yb + (32 * t + 4 * g4) * 4) = y;
                    ypk[2 * g4] = pack2(y[0], y[1]);
                    ypk[2 * g4 + 1] = pack2(y[2], y[3]);
                }
                if (p.y_bf16 && valid_now[g]) {
                    *reinterpret_cast<u32x4*>(yb + 64 * t) = u32x4{ypk[0], ypk[1], ypk[2], ypk[3]};
                    *reinterpret_cast<u32x4*>(yb + 64 * t + 16) = u32x4{ypk[4], ypk[5], ypk[6], ypk[7]};
                }
            }
            if (h == 0 && keep_now[g]) *reinterpret_cast<float2*>(p.stats + 2 * (size_t)tok_now[g]) = float2{mean[g], rstd[g]};
        }
        stamp(7);
    }
    if constexpr (STAMP) {
        if (lane == 0 && p.stamps)
#pragma unroll
            for (int k = 0; k < 8; ++k) p.stamps[((size_t)blk * NW + wave) * 8 + k] = t_acc[k];
    }
    wait_vm<0>();           // no DMA may still be writing this workgroup's LDS when it is handed to the next one
}


// =================================================================================================================================
// Backward of the same token-local chain, one launch per encoder layer (torch transformer.py:961-983 differentiated; R:412 loss.backward()):
//
//   dr2  = LayerNorm2-backward(dx ; r2, statistics)            dgamma2 += sum dy xhat ; dbeta2 += sum dy
//   dres2 = bf16(dr2 * keep3 / (1 - p))                        db2 += column sums            (stored: operand of dW2)
//   dh   = bf16((dres2 W2) * [h > 0] / (1 - p))                                              (stored: operand of dW1)
//   dx1  = dr2 + dh W1
//   dr1  = LayerNorm1-backward(dx1 ; r1, statistics)           dgamma1, dbeta1               -> written over dx (the residual path of the layer input)
//   dres1 = bf16(dr1 * keep1 / (1 - p))                        dbo += column sums            (stored: operand of dWo)
//   dctx = bf16(dres1 Wo)                                                                    (stored: the attention backward's input)
//
// The four launches this replaces (ln_bwd_v4_k, wst MASK, wst LNB, wst ACT) moved 10.6 KB per token row and layer; this one moves 6.5 KB.
// Same structure as ffn2_kernel: tokens on lanes, 8 waves x 32 tokens, the weights (W2^T rows, W1^T columns per hidden chunk, then Wo^T)
// as a fragment stream through the LDS ring.  The per-feature sums over tokens go through LDS adds (one 6 x 256 float image per
// workgroup) and reach the gradient buffer as 1 536 atomics per workgroup at the end.
//   slot 2q   : W2^T rows of chunk q   fragment 2 t + j : W2[32 t + 16 hh + 8 j + e][32 q + row]      (A rows = hidden features, k = output features of FFN2)
//   slot 2q+1 : W1^T columns of chunk q fragment 8 j + t : W1[32 q + 16 hh + 8 j + e][32 t + row]      (A rows = x1 features, k = hidden features)
//   slot 32+t : Wo^T rows of tile t     fragment 2 t' + j : Wo[32 t' + 16 hh + 8 j + e][32 t + row]     (A rows = ctx features, k = out-proj output features)
constexpr int BWD_SLOTS = FFN_SLOTS + FE / 32;      // 40
struct FragTabB { long w1[8], w2[8], wo[8]; int nl; };
__global__ __launch_bounds__(256) void encb_frag_kernel(const float* __restrict__ w, FragTabB tab, unsigned char* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;            // piece index inside the layer: < 40 * 16 * 64
    const int layer = blockIdx.y;
    const int lane = idx & 63, f = (idx >> 6) & 15, slot = idx >> 10;
    const int m = lane & 31, hh = lane >> 5, row = a_row_of_lane(m);
    const float* src;
    long stride;
    if (slot >= FFN_SLOTS) {
        const int t = slot - FFN_SLOTS, tp = f >> 1, j = f & 1;
        src = w + tab.wo[layer] + (long)(32 * tp + 16 * hh + 8 * j) * FE + 32 * t + row;
        stride = FE;
    } else if (slot & 1) {
        const int q = slot >> 1, t = f & 7, j = f >> 3;
        src = w + tab.w1[layer] + (long)(32 * q + 16 * hh + 8 * j) * FE + 32 * t + row;
        stride = FE;
    } else {
        const int q = slot >> 1, t = f >> 1, j = f & 1;
        src = w + tab.w2[layer] + (long)(32 * t + 16 * hh + 8 * j) * FF + 32 * q + row;
        stride = FF;
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[e * stride];
    *reinterpret_cast<u32x4*>(out + (size_t)layer * BWD_SLOTS * SLOT + (size_t)idx * 16) =
        u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
}

// Sums over the 32 token lanes of a half-wave for 16 per-lane values: a reduce-scatter butterfly (every level halves the values a lane
// carries) on the VALU's own lane networks - v_permlane16_swap for the rows 16 lanes apart, DPP row_ror:8 / row_half_mirror / quad_perm
// inside a row - instead of ds_bpermute: the first version of the backward kernel issued 772 LDS shuffles per token tile, each followed
// by a wait, and spent 60 % of its time in the two LayerNorm phases.  Returns, in every lane, the total of value index
// 8 b4 + 4 b3 + 2 b2 + b0 (b_k = bit k of the lane's token index c); lanes that differ only in bit 1 hold the same total.
__device__ __forceinline__ float dpp_f(float x, int ctrl_const);        // (ctrl must be a literal: specialised below)
#define GG_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xf, 0xf, false))
__device__ __forceinline__ float colsum16v(float (&v)[16], int c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {           // rows 16 lanes apart: one swap gives each row its half of both values
        // (inline asm: element 1 of the builtin's result comes back equal to element 0 with this compiler - tools/lane_probe.hip; the two
        // wait states a VALU write of an operand needs before the swap reads it go inside the string)
        float a = v[i], b = v[i + 8];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v[i] = a + b;
    }
    {
        const bool up = c & 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float send = up ? v[i] : v[i + 4], keep = up ? v[i + 4] : v[i];
            v[i] = keep + GG_DPP(send, 0x128);          // row_ror:8 = the lane 8 away inside the row
        }
    }
    {
        const bool up = c & 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float send = up ? v[i] : v[i + 2], keep = up ? v[i + 2] : v[i];
            v[i] = keep + GG_DPP(send, 0x141);          // row_half_mirror: lane 7 - i of the group of 8 (bits 0 - 2 flipped)
        }
    }
    {
        const bool up = c & 1;
        const float send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
        v[0] = keep + GG_DPP(send, 0xB1);               // quad_perm [1,0,3,2]
    }
    return v[0] + GG_DPP(v[0], 0x4E);                   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float half_sum(float s) {    // s of lane l + s of lane l ^ 32, in every lane
    float a = s, b = s;                 // two registers: the swap exchanges halves BETWEEN its operands
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t srd(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)std::min<size_t>(bytes, 0xffffffffu), 0x00020000);
}
__device__ __forceinline__ u32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0); }

template <bool DROP, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void encb_kernel(const EncBwdP p) {
    constexpr int NW = 8, NS = 4, PPW = 16 / NW, NTH = 64 * NW, PD = 3;
    constexpr int KWAIT = PPW * (NS - 2);           // outside the chunk loop: the weight pieces of the two slots after the one needed
    constexpr int KWAIT_H = KWAIT + 2;              // inside it: + the two gate pieces of the next chunk (see the issue order below)
    static_assert(BWD_SLOTS % NS == 0, "ring positions must be compile-time constants inside a sweep");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* const ring = smem_raw;                                           // [NS][SLOT]
    float* const Ps = reinterpret_cast<float*>(smem_raw + NS * SLOT);               // gamma2 | gamma1 [256 each]
    float* const Cs = Ps + 2 * FE;                                                  // column sums: dgamma2, dbeta2, dbias2, dgamma1, dbeta1, dbias1 [256 each]
    // The gate reference (stored hidden activations: 32 bytes per lane and chunk) comes by LDS-DMA too, one chunk ahead, into a per-wave
    // double buffer: an ordinary load inside the chunk loop makes the compiler wait with a vmcnt that knows nothing of the asm DMAs
    // issued after it - i.e. for the weight slots requested a moment ago (cdna_hip_programming.md 5, "mixing load kinds")
    unsigned char* const Hb = reinterpret_cast<unsigned char*>(Cs + 6 * FE);       // [NW][2][2][1024]

    const DropKey dk1 = drop_live(p.drop1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const long n_units = (p.M + 31) / 32;
    const long G = gridDim.x, blk = blockIdx.x;
    if (blk >= n_units) return;
    const long n_b = (n_units - blk + G - 1) / G;
    const int n_sweeps = (int)((n_b + NW - 1) / NW);
    const int last_tok = (int)p.M - 1;
    for (int i = tid; i < FE; i += NTH) { Ps[i] = 0.f; Ps[FE + i] = p.g1[i]; }
    for (int i = tid; i < 6 * FE; i += NTH) Cs[i] = 0.f;

    // Row tensors through buffer descriptors (scalar registers) and three per-lane byte offsets instead of nine 64-bit row pointers: the
    // pointers alone were 18 registers of a budget that has none to spare (a spill reload is a vector memory LOAD: its wait drains
    // every store issued before it).  Stores of lanes without a token get an offset beyond the descriptor's range and are dropped.
    const size_t M = (size_t)p.M;
    const __amdgpu_buffer_rsrc_t s_dx = srd(p.dx, M * FE * 4), s_dr2 = srd(p.dr2 ? p.dr2 : p.dx, M * FE * 4), s_r1 = srd(p.r1, M * FE * 2), s_dres2 = srd(p.dres2, M * FE * 2),
                                 s_dres1 = srd(p.dres1, M * FE * 2), s_dctx = srd(p.dctx, M * FE * 2), s_dh = srd(p.dh, M * FF * 2),
                                 s_st1 = srd(p.st1, M * 8);

    // STAMP (tools/encb_probe.py): 0 phase 1 (loads + LayerNorm2 backward), 1 DMA waits, 2 barriers, 3 W2^T products, 4 gate + dh stores,
    // 5 W1^T products, 6 phase 3 (LayerNorm1 backward + dr1 stores), 7 phase 4 products + dctx stores
    unsigned long long t_prev = 0;
    unsigned t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (k >= 0) t_acc[k] += (unsigned)(t - t_prev);
            t_prev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const unsigned ring_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)ring) + (unsigned)wave * (PPW * 1024);
    const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(p.Wf) + wave * (PPW * 1024) + lane * 16;
    int gs = 0;                                 // next slot to be consumed (counted over the launch); stream slot = gs mod 40
    int ss = NS - 1;                            // stream slot of the next DMA (gs + NS - 1 mod 40, kept incrementally)
    auto dma_next = [&]() {
        const unsigned char* s = wsrc + (size_t)ss * SLOT;
        const unsigned d = ring_lds + (unsigned)((gs + NS - 1) & (NS - 1)) * SLOT;
#pragma unroll
        for (int i = 0; i < PPW; ++i) glds16(s + i * 1024, d + i * 1024);
        ss = ss + 1 == BWD_SLOTS ? 0 : ss + 1;
    };
    {
        const unsigned char* s0 = wsrc;
#pragma unroll
        for (int i = 0; i < NS - 1; ++i)
#pragma unroll
            for (int k = 0; k < PPW; ++k) glds16(s0 + (size_t)i * SLOT + k * 1024, ring_lds + (unsigned)i * SLOT + k * 1024);
    }
    const float ksd = DROP ? 1.f / (1.f - p.drop1.p) : 1.f;
    const float ksg = p.gate_scale;             // 1 / (1 - p) of the inner dropout: the stored hidden activations are zero where dropped
    const unsigned char* const fbase = ring + lane * 16;
    const unsigned hb_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)Hb) + (unsigned)wave * 4096u;
    const unsigned char* const hb_rd = Hb + wave * 4096 + lane * 16;
    __syncthreads();                            // Ps / Cs initialised

    for (int sweep = 0; sweep < n_sweeps; ++sweep) {
        const long ju = (long)sweep * NW + wave;
        const long t0 = ju < n_b ? (blk + ju * G) * 32 : -1;
        const int tk = (int)t0 + c;
        const bool valid = t0 >= 0 && tk <= last_tok;
        const unsigned tokc = (unsigned)(valid ? tk : last_tok);
        // stores of lanes without a token go beyond the descriptor's range (dropped)
        const unsigned v512 = tokc * 512u + 32u * (unsigned)h, wadd = valid ? 0u : 0x80000000u;      // (every offset is < 2^31: enc_bwd_supported)
        auto o512 = [&]() { return v512; };
        auto o1k = [&]() { return v512 + tokc * 512u; };
        auto o1kx = [&]() { return 2u * v512; };
        auto wr = [&](unsigned off) { return off + wadd; };
        f32x16 acc[FE / 32];
        u32x4 rb[16];                           // bf16 rows as 8-element pieces: B fragments of the branch gradients

        // gate pieces of chunk 0: requested ahead of every other memory operation of the sweep (phase 1's loads are then younger:
        // the counted wait of the first gated slot covers them)
        const unsigned char* const hbase = reinterpret_cast<const unsigned char*>(p.h);
        stamp(-1);
        glds16(hbase + o1k(), hb_lds);
        glds16(hbase + o1k() + 16, hb_lds + 1024);
        // ---- phase 1: the LayerNorm2 backward's outputs (ln_bwd_v4_k): acc <- dr2 (fp32: the residual path), rb <- dres2 (bf16 B fragments)
        {
            const unsigned oa = o512(), ob = o1kx();
#pragma unroll
            for (int t = 0; t < FE / 32; ++t) {
                rb[2 * t] = bload(s_dres2, oa + 64u * t);
                rb[2 * t + 1] = bload(s_dres2, oa + 64u * t + 16u);
            }
#pragma unroll
            for (int t = 0; t < FE / 32; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 v = __builtin_bit_cast(f32x4, bload(s_dr2, ob + 128u * t + 16u * g4));
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][4 * g4 + j] = v[j];
                }
        }

        // ---- phase 2: per hidden chunk, dh = (dres2 W2) gated by the stored hidden activations ; acc (= dr2) += dh W1
        // issue order per chunk q: slot 2q: weights of slot 2q+3, gate pieces of chunk q+1 ; slot 2q+1: weights of slot 2q+4.  At the top of
        // either slot the youngest six DMA pieces (two weight slots + one gate pair) may still be in flight.
        stamp(0);
#pragma unroll 1
        for (int it = 0; it < NCH / 2; ++it) {
#pragma unroll
            for (int cc2 = 0; cc2 < 2; ++cc2) {
                const int q = 2 * it + cc2;
                wait_vm<KWAIT_H>();
                stamp(1);
                __syncthreads();
                stamp(2);
                dma_next();
                ++gs;
                {   // the next chunk's gate pieces (the last chunk re-requests its own: a fixed count of pieces in flight)
                    const int qn = q + 1 < NCH ? q + 1 : q;
                    const unsigned char* hq = hbase + o1k() + 64 * qn;
                    glds16(hq, hb_lds + (unsigned)((cc2 ^ 1) * 2048));
                    glds16(hq + 16, hb_lds + (unsigned)((cc2 ^ 1) * 2048 + 1024));
                }
                f32x16 a1;
#pragma unroll
                for (int i = 0; i < 16; ++i) a1[i] = 0.f;
                {
                    const unsigned char* fa = fbase + (2 * cc2) * SLOT;
                    bf16x8 a[PD + 1];
#pragma unroll
                    for (int s = 0; s < PD; ++s) a[s] = *reinterpret_cast<const bf16x8*>(fa + s * 1024);
                    __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        if (s + PD < 16) a[(s + PD) % (PD + 1)] = *reinterpret_cast<const bf16x8*>(fa + (s + PD) * 1024);
                        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s % (PD + 1)], __builtin_bit_cast(bf16x8, rb[s]), a1, 0, 0, 0);
                        if (s + PD < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                }
                stamp(3);
                wait_vm<KWAIT_H>();
                stamp(1);
                __syncthreads();
                stamp(2);
                dma_next();
                ++gs;
                bf16x8 pf[2];
                {
                    const u32x4 hg0 = *reinterpret_cast<const u32x4*>(hb_rd + cc2 * 2048), hg1 = *reinterpret_cast<const u32x4*>(hb_rd + cc2 * 2048 + 1024);
                    unsigned pk[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const unsigned hw = k < 4 ? hg0[k] : hg1[k - 4];
                        const float v0 = (hw & 0x7fffu) ? a1[2 * k] * ksg : 0.f;
                        const float v1 = (hw & 0x7fff0000u) ? a1[2 * k + 1] * ksg : 0.f;
                        pk[k] = pack2(v0, v1);
                    }
                    const u32x4 w0 = {pk[0], pk[1], pk[2], pk[3]}, w1 = {pk[4], pk[5], pk[6], pk[7]};
                    pf[0] = __builtin_bit_cast(bf16x8, w0);
                    pf[1] = __builtin_bit_cast(bf16x8, w1);
                    const unsigned wo = wr(o1k()) + 64u * q;
                    bstore(s_dh, wo, w0);
                    bstore(s_dh, wo + 16u, w1);
                }
                stamp(4);
                {
                    const unsigned char* fb = fbase + (2 * cc2 + 1) * SLOT;
                    bf16x8 a[PD + 1];
#pragma unroll
                    for (int f = 0; f < PD; ++f) a[f] = *reinterpret_cast<const bf16x8*>(fb + f * 1024);
                    __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
                    for (int f = 0; f < 16; ++f) {
                        if (f + PD < 16) a[(f + PD) % (PD + 1)] = *reinterpret_cast<const bf16x8*>(fb + (f + PD) * 1024);
                        acc[f & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[f % (PD + 1)], pf[f >> 3], acc[f & 7], 0, 0, 0);
                        if (f + PD < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                }
                stamp(5);
            }
        }

        // ---- phase 3: LayerNorm1 backward of dx1 (in acc) against the pre-LN1 sums.  The sums stream through 16 registers (two tiles ahead
        // of their use: a load waited for right behind fresh stores drains them), dr1 is stored tile by tile (its accumulator registers die
        // there) over dr2 in place, the masked bf16 branch gradient replaces the dead dres2 fragments in rb.
        {
            const float mean = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s_st1, tokc * 8u, 0, 0));
            const float rstd = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s_st1, tokc * 8u + 4u, 0, 0));
            const float nmr = -mean * rstd;
            const float* const gam = Ps + FE;
            const unsigned oa = o512();
            u32x4 rq[3][2];                     // ring of three tiles of pre-LN1 sums
            auto rload = [&](int t) __attribute__((always_inline)) {
                rq[t % 3][0] = bload(s_r1, oa + 64u * (t & 7));
                rq[t % 3][1] = bload(s_r1, oa + 64u * (t & 7) + 16u);
            };
            rload(0);
            rload(1);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < FE / 32; ++t) {
                rload(t + 2);                   // (tiles 8, 9 = 0, 1 again: the second pass starts with them)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 gw = *reinterpret_cast<const f32x4*>(&gam[32 * t + 16 * h + 4 * g4]);
                    const u32x4 rw = rq[t % 3][g4 >> 1];
                    const unsigned wa = rw[2 * (g4 & 1)], wb = rw[2 * (g4 & 1) + 1];
                    const float r[4] = {bf_lo(wa), bf_hi(wa), bf_lo(wb), bf_hi(wb)};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float gy = acc[t][4 * g4 + j] * gw[j];
                        s1 += gy;
                        s2 += gy * (r[j] * rstd + nmr);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            s1 = half_sum(s1);
            s2 = half_sum(s2);
            const float m1 = s1 * (1.f / FE), m2 = s2 * (1.f / FE);
            const unsigned wo = wr(oa), wdx = wr(o1kx());
            const unsigned pair0 = tokc * (unsigned)(FE / 2) + 8u * h;      // dropout stream: pair index of this lane's first element of tile 0
            float* const cs = Cs + 3 * FE;
#pragma unroll
            for (int t = 0; t < FE / 32; ++t) {
                if (t + 2 < FE / 32) rload(t + 2 + 8);          // slot (t + 10) % 3 == (t + 1) % 3 ... see the static_assert below
                const u32x4 r0 = rq[(t + 8) % 3][0], r1v = rq[(t + 8) % 3][1];
                float tg, tb, tc;
                {
                    float cg[16];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const u32x4 rw = g4 < 2 ? r0 : r1v;
                        const unsigned wa = rw[2 * (g4 & 1)], wb = rw[2 * (g4 & 1) + 1];
                        const float r[4] = {bf_lo(wa), bf_hi(wa), bf_lo(wb), bf_hi(wb)};
#pragma unroll
                        for (int j = 0; j < 4; ++j) cg[4 * g4 + j] = valid ? acc[t][4 * g4 + j] * (r[j] * rstd + nmr) : 0.f;
                    }
                    tg = colsum16v(cg, c);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    float cb[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) cb[i] = valid ? acc[t][i] : 0.f;
                    tb = colsum16v(cb, c);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    float cc[16];
                    unsigned pk[8];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 gw = *reinterpret_cast<const f32x4*>(&gam[32 * t + 16 * h + 4 * g4]);
                        const u32x4 rw = g4 < 2 ? r0 : r1v;
                        const unsigned wa = rw[2 * (g4 & 1)], wb = rw[2 * (g4 & 1) + 1];
                        const float r[4] = {bf_lo(wa), bf_hi(wa), bf_lo(wb), bf_hi(wb)};
                        f32x4 dr;
                        float v[4];
#pragma unroll
                        for (int j2 = 0; j2 < 2; ++j2) {        // one hash per pair of consecutive elements (drop_rng.h)
                            uint32_t bits = 0;
                            if (DROP) bits = drop_bits((pair0 + 16u * t + 2u * g4 + j2) * DROP_PHI + dk1.k0);
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) {
                                const int j = 2 * j2 + jj;
                                dr[j] = (acc[t][4 * g4 + j] * gw[j] - m1 - (r[j] * rstd + nmr) * m2) * rstd;
                                const bool kp = !DROP || (jj ? drop_keep_odd(bits, dk1.thr) : drop_keep_even(bits, dk1.thr));
                                v[j] = kp ? dr[j] * ksd : 0.f;
                                cc[4 * g4 + j] = valid ? v[j] : 0.f;
                            }
                        }
                        bstore(s_dx, wdx + 128u * t + 16u * g4, __builtin_bit_cast(u32x4, dr));
                        pk[2 * g4] = pack2(v[0], v[1]);
                        pk[2 * g4 + 1] = pack2(v[2], v[3]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    rb[2 * t] = u32x4{pk[0], pk[1], pk[2], pk[3]};
                    rb[2 * t + 1] = u32x4{pk[4], pk[5], pk[6], pk[7]};
                    bstore(s_dres1, wo + 64u * t, rb[2 * t]);
                    bstore(s_dres1, wo + 64u * t + 16u, rb[2 * t + 1]);
                    tc = colsum16v(cc, c);
                }
                if (!(c & 2)) {         // lanes that differ in bit 1 only hold the same totals
                    const int ncol = 16 * h + 8 * ((c >> 4) & 1) + 4 * ((c >> 3) & 1) + 2 * ((c >> 2) & 1) + (c & 1);      // colsum16v's value index of this lane
                    atomicAdd(&cs[32 * t + ncol], tg);
                    atomicAdd(&cs[FE + 32 * t + ncol], tb);
                    atomicAdd(&cs[2 * FE + 32 * t + ncol], tc);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        stamp(6);
        // ---- phase 4: dctx = dres1 Wo : one slot per 32 context features
#pragma unroll 1
        for (int it = 0; it < FE / 32 / NS; ++it) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int t = NS * it + k;
                wait_vm<KWAIT>();
                __syncthreads();
                dma_next();
                ++gs;
                f32x16 d;
#pragma unroll
                for (int i = 0; i < 16; ++i) d[i] = 0.f;
                const unsigned char* fa = fbase + k * SLOT;
                bf16x8 a[PD + 1];
#pragma unroll
                for (int s = 0; s < PD; ++s) a[s] = *reinterpret_cast<const bf16x8*>(fa + s * 1024);
                __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (s + PD < 16) a[(s + PD) % (PD + 1)] = *reinterpret_cast<const bf16x8*>(fa + (s + PD) * 1024);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s % (PD + 1)], __builtin_bit_cast(bf16x8, rb[s]), d, 0, 0, 0);
                    if (s + PD < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                const unsigned wo = wr(o512()) + 64u * t;
                bstore(s_dctx, wo, u32x4{pack2(d[0], d[1]), pack2(d[2], d[3]), pack2(d[4], d[5]), pack2(d[6], d[7])});
                bstore(s_dctx, wo + 16u, u32x4{pack2(d[8], d[9]), pack2(d[10], d[11]), pack2(d[12], d[13]), pack2(d[14], d[15])});
            }
        }
        stamp(7);
    }
    if constexpr (STAMP) {
        if (lane == 0 && p.stamps)
#pragma unroll
            for (int k = 0; k < 8; ++k) p.stamps[((size_t)blk * NW + wave) * 8 + k] = t_acc[k];
    }
    __syncthreads();
    for (int i = tid; i < 6 * FE; i += NTH) {
        float* dst = i < 3 * FE ? nullptr : i < 4 * FE ? p.dg1 : i < 5 * FE ? p.db1 : p.dbias1;
        if (dst) atomicAdd(dst + (i & (FE - 1)), Cs[i]);
    }
    wait_vm<0>();
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int NW, int NG, int NS, bool DROP, bool STAMP = false>
int launch_ffn2(const Ffn2P& p, hipStream_t st, long grid_cap) {
    constexpr size_t smem = (size_t)NS * SLOT + (size_t)(FF + 3 * FE) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn2_kernel<NW, NG, NS, DROP, STAMP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const long n_units = (p.M + 31) / 32;
    const unsigned grid = (unsigned)std::max<long>(1, std::min<long>(n_units, grid_cap));
    if (g_ev0) {
        hipExtLaunchKernelGGL((ffn2_kernel<NW, NG, NS, DROP, STAMP>), dim3(grid), dim3(64 * NW), (unsigned)smem, st, g_ev0, g_ev1, 0, p);
        g_ev0 = g_ev1 = nullptr;
    } else {
        hipLaunchKernelGGL((ffn2_kernel<NW, NG, NS, DROP, STAMP>), dim3(grid), dim3(64 * NW), smem, st, p);
    }
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace

void ffn2_time_next(hipEvent_t begin, hipEvent_t end) { g_ev0 = begin; g_ev1 = end; }

static_assert(FFN_SLOTS * SLOT == 32 * 16384, "kernels.h enc_frag_bytes");

int k_enc_frag_weights(const float* w, const long* w1_off, const long* w2_off, int nl, void* out, hipStream_t st) {
    GG_REQUIRE(nl >= 1 && nl <= 8 && w && out && al16(out), "enc_frag_weights: bad arguments");
    FragTab tab;
    tab.nl = nl;
    for (int l = 0; l < nl; ++l) { tab.w1[l] = w1_off[l]; tab.w2[l] = w2_off[l]; }
    hipLaunchKernelGGL(enc_frag_kernel, dim3(FFN_SLOTS * 16 * 64 / 256, nl), dim3(256), 0, st, w, tab, reinterpret_cast<unsigned char*>(out));
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

static_assert(BWD_SLOTS * SLOT == 40 * 16384, "kernels.h encb_frag_bytes");

int k_encb_frag_weights(const float* w, const long* w1_off, const long* w2_off, const long* wo_off, int nl, void* out, hipStream_t st) {
    GG_REQUIRE(nl >= 1 && nl <= 8 && w && out && al16(out), "encb_frag_weights: bad arguments");
    FragTabB tab;
    tab.nl = nl;
    for (int l = 0; l < nl; ++l) { tab.w1[l] = w1_off[l]; tab.w2[l] = w2_off[l]; tab.wo[l] = wo_off[l]; }
    hipLaunchKernelGGL(encb_frag_kernel, dim3(BWD_SLOTS * 16 * 64 / 256, nl), dim3(256), 0, st, w, tab, reinterpret_cast<unsigned char*>(out));
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

bool enc_bwd_supported(const EncBwdP& p) {
    if (p.M < 1 || (double)p.M * 1024.0 >= 2147483648.0) return false;         // byte offsets of a row + 2^31 (the dropped-store offset) must not wrap
    if (!p.dx || !p.Wf || !p.h || !p.r1 || !p.st1 || !p.g1 || !p.dres2 || !p.dh || !p.dres1 || !p.dctx) return false;
    for (const void* q : {(const void*)p.dx, p.Wf, p.h, p.r1, p.dres2, (const void*)p.dh, (const void*)p.dres1, (const void*)p.dctx})
        if (!al16(q)) return false;
    if (reinterpret_cast<uintptr_t>(p.st1) & 7) return false;
    return true;
}

int enc_bwd(const EncBwdP& p, hipStream_t st) {
    GG_REQUIRE(enc_bwd_supported(p), "enc_bwd: unsupported operands");
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        GG_CHECK_HIP(hipGetDevice(&dev));
        GG_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    constexpr size_t smem = (size_t)4 * SLOT + (size_t)8 * FE * 4 + (size_t)8 * 4096;
    static bool attr_set = false;
    if (!attr_set) {
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&encb_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&encb_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const long n_units = (p.M + 31) / 32;
    const unsigned grid = (unsigned)std::max<long>(1, std::min<long>(n_units, n_cu));
    if (p.stamps) {
        static bool attr2 = false;
        if (!attr2) {
            GG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&encb_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr2 = true;
        }
        hipLaunchKernelGGL((encb_kernel<true, true>), dim3(grid), dim3(512), smem, st, p);
    } else
    if (p.drop1.p > 0.f) hipLaunchKernelGGL(encb_kernel<true>, dim3(grid), dim3(512), smem, st, p);
    else hipLaunchKernelGGL(encb_kernel<false>, dim3(grid), dim3(512), smem, st, p);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

bool ffn2_supported(const Ffn2P& p) {
    if (p.M < 1 || (double)p.M * FF >= 2.0e9) return false;
    if (!p.X || !p.Wf || !p.Hs || !p.R2 || !p.ln_g || !p.ln_b || !p.Y || !p.stats) return false;
    if (!al16(p.X) || !al16(p.Wf) || !al16(p.Hs) || !al16(p.R2) || !al16(p.Y) || (reinterpret_cast<uintptr_t>(p.stats) & 7)) return false;
    return true;
}

namespace {
int g_grid_override = 0;
long enc_grid_cus() {
    if (g_grid_override > 0) return g_grid_override;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 0;
        if (n_cu <= 0) return 256;
    }
    static const int cu_pct = getenv("GG_ENC_CU_PCT") ? atoi(getenv("GG_ENC_CU_PCT")) : 100;
    return std::max<long>(8, (long)n_cu * cu_pct / 100);
}
}  // namespace
void enc_set_grid(int workgroups) { g_grid_override = workgroups > 0 ? workgroups : 0; }
long ffn2_sweep_tokens(int variant) { return enc_grid_cus() * 32 * (variant >= 4 ? 4 : 8); }

int ffn2(const Ffn2P& p, hipStream_t st, int variant) {
    GG_REQUIRE(ffn2_supported(p), "ffn2: unsupported operands");
    const long cus = enc_grid_cus();
    const bool drop = p.drop1.p > 0.f;
    GG_REQUIRE(drop == (p.drop2.p > 0.f), "ffn2: the two dropout sites are on or off together");
    if (p.stamps) {
        GG_REQUIRE(drop, "ffn2: the stamped builds exist with dropout only");
        if (variant == 0) return launch_ffn2<8, 1, 4, true, true>(p, st, cus);
        if (variant == 4) return launch_ffn2<4, 1, 4, true, true>(p, st, cus);
    }
    switch (variant) {
        case 0: return drop ? launch_ffn2<8, 1, 4, true>(p, st, cus) : launch_ffn2<8, 1, 4, false>(p, st, cus);
        case 2: return drop ? launch_ffn2<8, 1, 8, true>(p, st, cus) : launch_ffn2<8, 1, 8, false>(p, st, cus);
        case 4: return drop ? launch_ffn2<4, 1, 4, true>(p, st, cus) : launch_ffn2<4, 1, 4, false>(p, st, cus);
        case 6: return drop ? launch_ffn2<4, 1, 8, true>(p, st, cus) : launch_ffn2<4, 1, 8, false>(p, st, cus);
    }
    set_error("ffn2: unknown variant");
    return -2;
}

}  // namespace gg
