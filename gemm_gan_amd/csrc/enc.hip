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

size_t enc_frag_bytes(int nl) { return (size_t)nl * FFN_SLOTS * SLOT; }

int k_enc_frag_weights(const float* w, const long* w1_off, const long* w2_off, int nl, void* out, hipStream_t st) {
    GG_REQUIRE(nl >= 1 && nl <= 8 && w && out && al16(out), "enc_frag_weights: bad arguments");
    FragTab tab;
    tab.nl = nl;
    for (int l = 0; l < nl; ++l) { tab.w1[l] = w1_off[l]; tab.w2[l] = w2_off[l]; }
    hipLaunchKernelGGL(enc_frag_kernel, dim3(FFN_SLOTS * 16 * 64 / 256, nl), dim3(256), 0, st, w, tab, reinterpret_cast<unsigned char*>(out));
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

bool ffn2_supported(const Ffn2P& p) {
    if (p.M < 1 || (double)p.M * FF >= 2.0e9) return false;
    if (!p.X || !p.Wf || !p.Hs || !p.R2 || !p.ln_g || !p.ln_b || !p.Y || !p.stats) return false;
    if (!al16(p.X) || !al16(p.Wf) || !al16(p.Hs) || !al16(p.R2) || !al16(p.Y) || (reinterpret_cast<uintptr_t>(p.stats) & 7)) return false;
    return true;
}

int ffn2(const Ffn2P& p, hipStream_t st, int variant) {
    GG_REQUIRE(ffn2_supported(p), "ffn2: unsupported operands");
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        GG_CHECK_HIP(hipGetDevice(&dev));
        GG_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    static const int cu_pct = getenv("GG_ENC_CU_PCT") ? atoi(getenv("GG_ENC_CU_PCT")) : 100;
    const long cus = std::max<long>(8, (long)n_cu * cu_pct / 100);
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
