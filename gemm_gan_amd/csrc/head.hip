// MLP heads of the generator / critic (R:226-231) in two launches instead of nine.
// STATUS: correct (tests/test_kernels_gpu.py, tests/test_engine_oracle_gpu.py) and NOT faster: configs[0] 1.435 vs 1.46-1.48 ms, cfg3
// 25.25 vs 25.17 ms, configs[1] 6.62 vs 6.63 ms - a chain of three dependent few-tile products is bound by its three weight
// round trips whether they are three launches or three phases of one.  Opt-in: GG_HEAD_FUSED=1 / gg_set_head_fused.
//
//   forward   a1 = act(a1_pre + c W1c^T + b1) ;  a2 = act(a1 W2^T + b2) ;  out = a2 w3 + b3   (out: critic only, OUT = 1)
//   backward  dh2 = dh2_pre * act'(a2)  (critic: dh2_pre = dout w3^T formed here) ;  dh1 = (dh2 W2) * act'(a1) ;  dc = dh1 W1c
//
// Every one of these products is a few tiles of [rows, 256] x [256, 256]: as separate launches (engine.hip lin_fwd / lin_bwd_data
// on gemm_tiny_kernel, k_act_bwd) each costs a kernel's 5 - 8 us floor and a round trip of a [rows, H] tensor through memory.
// Here a workgroup (8 waves, one 32-feature tile each) owns 32 rows ("tokens on lanes" as in tlin.hip: Y^T = W X^T, accumulator rows = output features, columns =
// the 32 rows) and keeps them for the whole chain: the activation tile between two layers crosses the waves through LDS (each
// wave produces 32-feature slices of it, each consumes all of it as the next product's B operand).
// Operands are rounded to bf16 exactly where gemm_small.hip rounds them (fp32 in memory, bf16 MFMA operands, fp32 accumulate),
// so the fused route differs from the launch-per-product route only by summation order.  bf16 precision mode only: the f32 and
// bf16x3 parity modes keep the generic route.  Weights are read as fp32 from the master copy (W [N][K] row-major for the
// forward products; for the backward products the SAME matrices are read column-wise - lane = output row, so a wave reads 32
// consecutive floats per k: coalesced).
#include "kernels.h"
#include <cstdlib>

namespace gg {
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ bf16x8 frag_f32(const float* p) {          // 8 consecutive floats -> one MFMA fragment
    const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
    u32x4 w = {pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
    return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ bf16x8 frag_col(const float* p, long stride) {      // 8 floats `stride` apart (a column of a row-major matrix)
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[j * stride];
    u32x4 w = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

constexpr int HMAX = 256;                 // hidden width (multiple of 32, <= 256)
constexpr int TLD = HMAX + 8;             // bf16 per LDS row of the activation tile: 132 dwords = 4 mod 64, conflict-free b128 reads

// one layer: acc (features 32 * wave ..) = sum_k A[f][k] B[row][k], A from global memory (row-major, stride lda: COL false; or
// column-wise, A[f][k] = Wm[k * lda + f]: COL true), B either this lane's fp32 row in global memory (Brow, already clamped to a
// valid row) or the LDS tile.  EVERY load of the layer is issued before the first conversion (K <= 256: 16 k-steps, 32 float4 or
// 128 dword registers per lane), unconditionally - steps past K / 16 repeat the last one: the layer costs ONE memory round trip.
// A wave without a tile (H < 256) loads tile 0 again and drops the result.
constexpr int SMAX = HMAX / 16;
// weights of k-steps [S0, S0 + NS) of one 32-feature tile
template <bool COL, int NS>
struct WeightFrags {
    float raw[NS][8];
    __device__ __forceinline__ void request(const float* Wm, long lda, int K, int f, int h, int s0) {
        const int nsteps = K / 16;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = 16 * min(s0 + s, nsteps - 1) + 8 * h;
            if constexpr (COL) {
                const float* q = Wm + (long)k * lda + f;
#pragma unroll
                for (int j = 0; j < 8; ++j) raw[s][j] = q[(long)j * lda];
            } else {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(Wm + (long)f * lda + k), hi = *reinterpret_cast<const f32x4*>(Wm + (long)f * lda + k + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { raw[s][j] = lo[j]; raw[s][4 + j] = hi[j]; }
            }
        }
    }
    __device__ __forceinline__ bf16x8 frag(int s) const {
        u32x4 w = {pack2(raw[s][0], raw[s][1]), pack2(raw[s][2], raw[s][3]), pack2(raw[s][4], raw[s][5]), pack2(raw[s][6], raw[s][7])};
        return __builtin_bit_cast(bf16x8, w);
    }
};
// acc += the k-steps [s0, s0 + NS) of the product
template <bool COL, bool BLDS, int NS>
__device__ __forceinline__ void multiply(f32x16& acc, const WeightFrags<COL, NS>& wf, int K, const float* Brow, const __bf16* Bt, bool has, int c, int h,
                                         int s0) {
    const int nsteps = K / 16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = 16 * min(s0 + s, nsteps - 1) + 8 * h;
        const bf16x8 bf = BLDS ? *reinterpret_cast<const bf16x8*>(Bt + c * TLD + k) : frag_f32(Brow + k);
        if (has && s0 + s < nsteps) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf.frag(s), bf, acc, 0, 0, 0);
    }
}
__device__ __forceinline__ void zero(f32x16& acc) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
}

// ---- forward -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void head_fwd_k(HeadP p) {
    __shared__ __attribute__((aligned(16))) __bf16 Ts[32 * TLD];
    __shared__ float Red[8][32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int H = p.H, ntile = H / 32;
    const bool has = wave < ntile;
    const int tile = has ? wave : 0;
    const long row0 = (long)blockIdx.x * 32;
    const long row = min(row0 + c, (long)p.rows - 1);          // clamped lanes recompute the last row; stores are predicated
    const bool valid = row0 + c < p.rows;
    f32x16 acc;
    // layer 1: a1 = act(a1_pre + c W1c^T + b1); the second layer's weights are requested before its epilogue
    WeightFrags<false, SMAX> w1;
    w1.request(p.W1c, p.ldw1, p.E, 32 * tile + c, h, 0);
    zero(acc);
    multiply<false, false, SMAX>(acc, w1, p.E, p.cvec + row * p.E, nullptr, has, c, h, 0);
    WeightFrags<false, SMAX> w2;
    w2.request(p.W2, H, H, 32 * tile + c, h, 0);
    if (has) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = 32 * tile + 8 * g + 4 * h;
            f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            v += *reinterpret_cast<const f32x4*>(p.a1 + row * H + f);
            v += *reinterpret_cast<const f32x4*>(p.b1 + f);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : p.slope * v[j];
            if (valid) *reinterpret_cast<f32x4*>(p.a1 + row * H + f) = v;
            *reinterpret_cast<unsigned*>(&Ts[c * TLD + f]) = pack2(v[0], v[1]);
            *reinterpret_cast<unsigned*>(&Ts[c * TLD + f + 2]) = pack2(v[2], v[3]);
        }
    }
    __syncthreads();
    // layer 2: a2 = act(a1 W2^T + b2)
    zero(acc);
    multiply<false, true, SMAX>(acc, w2, H, nullptr, Ts, has, c, h, 0);
    float part = 0.f;
    if (has) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = 32 * tile + 8 * g + 4 * h;
            f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            v += *reinterpret_cast<const f32x4*>(p.b2 + f);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : p.slope * v[j];
            if (valid) *reinterpret_cast<f32x4*>(p.a2 + row * H + f) = v;
            if (p.out) {          // the score uses the bf16-rounded operands of the product it replaces
                const f32x4 w = *reinterpret_cast<const f32x4*>(p.w3 + f);
#pragma unroll
                for (int j = 0; j < 4; ++j) part += (float)(__bf16)v[j] * (float)(__bf16)w[j];
            }
        }
    }
    if (p.out) {                  // critic: out = a2 w3 + b3 (one column)
        part += __shfl_xor(part, 32, 64);
        if (h == 0) Red[wave][c] = part;
        __syncthreads();
        if (wave == 0 && h == 0 && valid && row0 + c < p.out_rows) {
            float sum = p.b3[0];
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += Red[w][c];
            p.out[(row0 + c) * p.ldo] = sum;
        }
    }
}

// ---- backward (data path) ------------------------------------------------------------------------------------------------------
template <bool DCOND>
__global__ __launch_bounds__(512) void head_bwd_k(HeadP p) {
    __shared__ __attribute__((aligned(16))) __bf16 Ts[32 * TLD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int H = p.H, ntile = H / 32;
    const bool has = wave < ntile;
    const int tile = has ? wave : 0;
    const long row0 = (long)blockIdx.x * 32;
    const long row = min(row0 + c, (long)p.rows - 1);
    const bool valid = row0 + c < p.rows;
    // the weights of the first product (W2 read column-wise) are requested before anything else
    WeightFrags<true, SMAX / 2> w2;
    w2.request(p.W2, H, H, 32 * tile + c, h, 0);
    // dh2 = dh2_pre * act'(a2): elementwise, each wave its feature tile; the bf16 tile goes to LDS for the next product
    const float dsc = p.dout ? p.dout[row] : 0.f;
    if (has) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = 32 * tile + 8 * g + 4 * h;
            f32x4 v;
            if (p.dout) {         // critic: dout [rows] (one output column) times w3, on the operands the replaced product rounds
                const f32x4 w = *reinterpret_cast<const f32x4*>(p.w3 + f);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (float)(__bf16)dsc * (float)(__bf16)w[j];
            } else {
                v = *reinterpret_cast<const f32x4*>(p.dh2 + row * H + f);
            }
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.a2 + row * H + f);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a[j] > 0.f ? v[j] : p.slope * v[j];
            if (valid) *reinterpret_cast<f32x4*>(p.dh2 + row * H + f) = v;
            *reinterpret_cast<unsigned*>(&Ts[c * TLD + f]) = pack2(v[0], v[1]);
            *reinterpret_cast<unsigned*>(&Ts[c * TLD + f + 2]) = pack2(v[2], v[3]);
        }
    }
    __syncthreads();
    // dh1 = (dh2 W2) * act'(a1):  dh1^T[f'][row] = sum_f W2[f][f'] dh2[row][f]  (A = W2 read column-wise)
    f32x16 acc;
    zero(acc);
    multiply<true, true, SMAX / 2>(acc, w2, H, nullptr, Ts, has, c, h, 0);
    __builtin_amdgcn_sched_barrier(0);          // (one weight set in flight at a time: every column-wise load carries its own address)
    w2.request(p.W2, H, H, 32 * tile + c, h, SMAX / 2);
    multiply<true, true, SMAX / 2>(acc, w2, H, nullptr, Ts, has, c, h, SMAX / 2);
    const bool hasc = DCOND && wave < p.E / 32;
    __builtin_amdgcn_sched_barrier(0);
    WeightFrags<true, SMAX / 2> w1;
    if constexpr (DCOND) w1.request(p.W1c, p.ldw1, H, 32 * (hasc ? wave : 0) + c, h, 0);
    __syncthreads();              // every wave has read the dh2 tile: it is overwritten with dh1 below
    if (has) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = 32 * tile + 8 * g + 4 * h;
            f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.a1 + row * H + f);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a[j] > 0.f ? v[j] : p.slope * v[j];
            if (valid) *reinterpret_cast<f32x4*>(p.dh1 + row * H + f) = v;
            *reinterpret_cast<unsigned*>(&Ts[c * TLD + f]) = pack2(v[0], v[1]);
            *reinterpret_cast<unsigned*>(&Ts[c * TLD + f + 2]) = pack2(v[2], v[3]);
        }
    }
    if constexpr (!DCOND) return;
    __syncthreads();
    // dc = dh1 W1c:  dc^T[e][row] = sum_f W1c[f][e] dh1[row][f]
    zero(acc);
    multiply<true, true, SMAX / 2>(acc, w1, H, nullptr, Ts, hasc, c, h, 0);
    __builtin_amdgcn_sched_barrier(0);
    w1.request(p.W1c, p.ldw1, H, 32 * (hasc ? wave : 0) + c, h, SMAX / 2);
    multiply<true, true, SMAX / 2>(acc, w1, H, nullptr, Ts, hasc, c, h, SMAX / 2);
    if (hasc) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = 32 * wave + 8 * g + 4 * h;
            const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            if (valid) *reinterpret_cast<f32x4*>(p.dcond + row * p.E + f) = v;
        }
    }
}
inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
}  // namespace

bool head_fused_supported(const HeadP& p) {
    if (p.rows < 1 || p.H < 32 || p.H > HMAX || p.H % 32 || p.E < 16 || p.E > 256 || p.E % 32 || p.ldw1 % 4) return false;
    if (!p.W1c || !p.W2 || !p.a1 || !p.a2 || !al16(p.W1c) || !al16(p.W2) || !al16(p.a1) || !al16(p.a2)) return false;
    // every operand the forward / backward launch reads 16 bytes at a time, when given: the engine falls back to the Linear chain
    // on a flat-buffer offset that misses the alignment instead of failing the step in head_fwd / head_bwd
    if (!al16(p.cvec) || !al16(p.b1) || !al16(p.b2) || !al16(p.w3) || !al16(p.dh1) || !al16(p.dh2) || !al16(p.dcond)) return false;
    return true;
}
int head_fwd(const HeadP& p, hipStream_t st) {
    GG_REQUIRE(head_fused_supported(p) && p.cvec && p.b1 && p.b2, "head_fwd: unsupported operands");
    GG_REQUIRE(!p.out || (p.w3 && p.b3), "head_fwd: the score column needs w3 / b3");
    head_fwd_k<<<(unsigned)((p.rows + 31) / 32), 512, 0, st>>>(p);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}
int head_bwd(const HeadP& p, hipStream_t st) {
    GG_REQUIRE(head_fused_supported(p) && p.dh2 && p.dh1, "head_bwd: unsupported operands");
    GG_REQUIRE(!p.dout || p.w3, "head_bwd: dout needs w3");
    if (p.dcond) head_bwd_k<true><<<(unsigned)((p.rows + 31) / 32), 512, 0, st>>>(p);
    else head_bwd_k<false><<<(unsigned)((p.rows + 31) / 32), 512, 0, st>>>(p);
    GG_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gg
