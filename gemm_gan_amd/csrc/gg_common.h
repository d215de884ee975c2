// Common declarations of the gemm_gan_amd HIP engine (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

namespace gg {

// thread-local error text surfaced through gg_last_error()
void set_error(const std::string& s);
#define GG_CHECK_HIP(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            gg::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));               \
            return -1;                                                                       \
        }                                                                                    \
    } while (0)
#define GG_REQUIRE(cond, msg)                                                                \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            gg::set_error(std::string("requirement failed: ") + #cond + " - " + (msg));      \
            return -2;                                                                       \
        }                                                                                    \
    } while (0)
#define GG_TRY(expr)                                                                         \
    do {                                                                                     \
        int _r = (expr);                                                                     \
        if (_r != 0) return _r;                                                              \
    } while (0)

enum Layout { LAY_KC = 0,   // operand stored [rows][K]: reduction index contiguous
              LAY_KS = 1 }; // operand stored [K][rows]: reduction index strided

enum Act { ACT_NONE = 0, ACT_LRELU = 1 };

// One (batched, optionally split-K) GEMM:  C[M,N] (+)= epi( alpha * sum_k A'(m,k) * B'(k,n) )
//   A'(m,k) = A[m*lda + k]      (LAY_KC)   or  A[k*lda + m]   (LAY_KS)
//   B'(k,n) = B[n*ldb + k]      (LAY_KC)   or  B[k*ldb + n]   (LAY_KS)
// batch index z -> (zo, zi) = (z / batch_inner, z % batch_inner); pointer offsets zo*s?o + zi*s?i.
struct GemmP {
    const float* A = nullptr;
    const float* B = nullptr;
    float* C = nullptr;
    int M = 0, N = 0, K = 0;
    long lda = 0, ldb = 0, ldc = 0;
    int layA = LAY_KC, layB = LAY_KC;
    int a_bf16 = 0, b_bf16 = 0;          // gemm_bf16 only, (LAY_KS, LAY_KS) only: operand stored as bf16 (lda/ldb/strides in elements)
    int batch = 1, batch_inner = 1;
    long sAo = 0, sAi = 0, sBo = 0, sBi = 0, sCo = 0, sCi = 0;
    int splitk = 1;             // >1 => partial sums are atomically added into C (epilogue must be linear)
    // epilogue: v = alpha*acc (+ bias[n]) (+ C_old if accumulate) ; act ; colmask -> -inf
    float alpha = 1.f;
    const float* bias = nullptr;
    int accumulate = 0;
    int act = ACT_NONE;
    float slope = 0.f;
    const uint8_t* colmask = nullptr;   // [outer batch][N] bytes, nonzero => write -inf (attention key padding)
    long colmask_stride = 0;
    int colmask_mod = 0;                // >0: mask row = outer batch index % colmask_mod (replica-stacked batches)
    // A-operand transform (LAY_KC only): FiLM  a' = gamma[g][k]*a + beta[g][k], g = m / film_group
    const float* film_gamma = nullptr;
    const float* film_beta = nullptr;
    long film_ld = 0;
    int film_group = 0;
    // C row remap: row m is written to row m + m / c_row_group + 1 (patch rows behind a CLS row)
    int c_row_group = 0;
};

int gemm_f32(const GemmP& p, hipStream_t st);    // exact fp32 (v_mfma_f32_32x32x2_f32)
int gemm_bf16(const GemmP& p, hipStream_t st);   // bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16)
bool gemm_small_wanted(const GemmP& p);          // few-tile problems: use the latency-optimised 64x64 / one-shot-K kernel
int gemm_small(const GemmP& p, hipStream_t st);
bool gemm_small_x3_ok(const GemmP& p);           // bf16x3 mode: the register-direct kernel with six bf16 part products (fp32-grade)
int gemm_small_x3(const GemmP& p, hipStream_t st);

}  // namespace gg
