/* gemmgan_lab.h - C ABI of libgemmgan_lab.so: test hooks and opt-in kernels OUTSIDE the product library.
 *
 * libgemmgan.so (include/gemmgan.h) is what a host links: the engine and the kernels of its default routes.  This second library
 * holds (1) the kernel-level test hooks - one kernel family per call, launched exactly as the engine routes it, for tests/ and tools/ -
 * and (2) the kernels that were built, tested and LOST their A/B against the default routes: the fused
 * feed-forward kernels (csrc/ffn.hip, csrc/enc.hip ffn2_kernel), the fused encoder-layer backward (csrc/enc.hip encb_kernel) and the
 * fused MLP-head chain (csrc/head.hip).  Loading it (after libgemmgan.so) registers those entry points with the engine
 * (gg_lab_register, called from a constructor); only then do gg_set_ffn2 / gg_set_encb / gg_set_ffn_fused / gg_set_head_fused and
 * the GG_FFN2 / GG_ENCB / GG_FFN_FUSED / GG_HEAD_FUSED environment forms succeed - without it they fail loudly, they never fall back.
 */
#ifndef GEMMGAN_LAB_H
#define GEMMGAN_LAB_H
#include "gemmgan.h"
#ifdef __cplusplus
extern "C" {
#endif

/* exported by libgemmgan.so: takes the table of entry points (csrc/kernels.h LabTable) of a lab library of the SAME build */
int gg_lab_register(const void* table, uint64_t bytes);
/* exported by libgemmgan_lab.so: 1 once its constructor has registered the table (a host that dlopen()s it can check) */
int gg_lab_loaded(void);

/* ---- test hooks: individual kernels through the same ABI (tests/ only) ---------------------------- */
int gg_test_gemm(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb,
                 int64_t ldc, int layA, int layB, int splitk, float alpha, const float* bias, int act,
                 float slope, int accumulate, void* stream);
int gg_test_gemm_small(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb,
                       int64_t ldc, int layA, int layB, int splitk, float alpha, const float* bias, int act,
                       float slope, int accumulate, void* stream);
int gg_test_gemm_bf16_stored(const void* A, const void* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb,
                             int64_t ldc, int a_bf16, int b_bf16, int splitk, void* stream);   /* [K,M] / [K,N] operands, bf16-stored */
int gg_test_gemm_bf16(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb,
                      int64_t ldc, int layA, int layB, int splitk, float alpha, const float* bias, int act,
                      float slope, int accumulate, void* stream);
/* ---- kernel-level hooks (csrc/testhooks.hip): ONE kernel family of the hot path on buffers the test supplies, launched
 * through the same host wrappers and routing as the engine; tests/test_kernels_gpu.py compares them with float64 products of
 * host-rounded operands.  Dropout keys are made from (p, seed, site, call) exactly as the engine makes them (csrc/drop_rng.h). */
typedef struct gg_test_linear_args {     /* Y = epi(X W^T): the encoder-layer Linears (torch transformer.py:940-983 via R:213) */
    const void* X; int64_t ldx; int64_t M; int32_t x_bf16;       /* activations [M,K], fp32 or bf16 (ld in elements)          */
    const void* W; int64_t ldw;                                  /* [N,K]: bf16 (routes 0, 1) or fp32 (route 2)                */
    const float* bias;
    void* Y; int64_t ldy; int32_t y_bf16; int64_t y_rows;        /* y_rows: with LayerNorm, rows whose pre-LN sum is stored   */
    int32_t N, K;
    const float* film_g; const float* film_b; int64_t film_ld; int32_t film_group;   /* X' = g[m / group] * X + b[m / group] */
    int32_t y_row_group;                                         /* output row m -> m + m / group + 1 (CLS row per sample)     */
    int32_t act_relu;
    float drop_p; uint64_t drop_seed; uint32_t drop_site, drop_call; int64_t drop_ld;   /* element index = row * drop_ld + n  */
    const void* mask_ref; int64_t ldref; float mask_scale; int32_t mask_bf16;           /* y = ref > 0 ? y * scale : 0        */
    int32_t accumulate;                                          /* y += previous content                                      */
    const float* res; int64_t ldres; int64_t res_rows;           /* + res[row % res_rows]                                      */
    const float* ln_g; const float* ln_b; float* ln_y; float* ln_stats;                  /* LayerNorm of the sum (eps 1e-5)    */
    int32_t res_bf16, ln_y_bf16;                                 /* route 0: `res` / `ln_y` are bf16 arrays (same strides, elements) */
    void* lnb_dres; float* lnb_dgamma; float* lnb_dbeta; float* lnb_dbias;   /* route 0, accumulate: += then LayerNorm backward (wst.hip EPI_LNB):
                          Y = dr_in (read), res = pre-LN sums, ln_stats (read), ln_g; ln_y = dr out, lnb_dres = masked branch gradient (bf16) */
    void* w_parts;     /* routes 2 / 3: scratch for the pre-split weights, route * N * K bf16 elements                                  */
    int32_t route;     /* 0: as the engine routes it (weight-stationary kernel when one takes the shape), 1: token-on-lane kernels only,
                          2 / 3: the split-operand Linear of GG_PREC_BF16X3 with 2 (hi, lo: three products, the backward form) /
                          3 (hi, mid, lo: six products, the forward form) operand parts; W is then the fp32 matrix                  */
} gg_test_linear_args;
int gg_test_linear(const gg_test_linear_args* a, int32_t* kernel_class, void* stream);
/* fused self-attention (torch functional.py:6206-6660): qkv [qkv_B or N, S, 3E] packed, mask [mask_B, S] bytes, ctx [N, S, E],
 * lse2 [N, nh, S] (log2-sum-exp of the scaled scores); backward: dctx -> dqkv [N, S, 3E], delta [N, nh, S] scratch.
 * io_bf16: 0 fp32 tensors / 1 bf16 tensors through the bf16 kernels; 2 / 3: fp32 tensors through the split-operand kernels of
 * GG_PREC_BF16X3 with that many bf16 parts per MFMA operand */
const char* gg_test_attn_kernel_name(int which, int S, int E, int nh);     /* 0 forward, 1 dQ, 2 dK|dV */
int gg_test_attn_fwd(const void* qkv, const uint8_t* mask, int mask_B, void* ctx, float* lse2, int64_t N, int S, int E, int nh,
                     float drop_p, uint64_t drop_seed, uint32_t drop_site, uint32_t drop_call, int io_bf16, int64_t qkv_B, void* stream);
int gg_test_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse2, float* delta, const uint8_t* mask,
                     int mask_B, void* dqkv, int64_t N, int S, int E, int nh, float drop_p, uint64_t drop_seed, uint32_t drop_site,
                     uint32_t drop_call, int io_bf16, int64_t qkv_B, void* stream);
/* fused feed-forward block (csrc/ffn.hip; torch transformer.py:961-983): X [M,256] fp32, W1 bf16 [512][256], W2T = bf16 W2^T [512][256];
 * Hs bf16 [M,512] and R2 fp32 [M,256] written for rows < keep_rows (-1: all), Y = LayerNorm(R2) fp32, stats [M,2] (mean, rstd);
 * dropout keys: (p, seed, site1, call) for the inner dropout, (p, seed, site2, call) for the post-FFN one */
int gg_test_ffn_fused(const float* X, int64_t M, const void* W1, const float* b1, const void* W2T, const float* b2, void* Hs, float* R2,
                      int64_t keep_rows, const float* ln_g, const float* ln_b, float* Y, float* stats, float drop_p, uint64_t drop_seed,
                      uint32_t site1, uint32_t site2, uint32_t drop_call, void* stream);
/* fused feed-forward block, round 4 (csrc/enc.hip): X = x1 as bf16 [M,256]; W1 [512,256] / W2 [256,512] fp32 (the fragment-ordered bf16 image is
   built into wfrag, gg_test_ffn2_frag_bytes() bytes); Hs bf16 [M,512], R2 bf16 or fp32 [M,256], stats [M,2] for rows < keep_rows; Y bf16 or fp32 */
int64_t gg_test_ffn2_frag_bytes(void);
/* caps the persistent grids of the streamed encoder kernels at `workgroups` (0: one per compute unit, the default): a pass of the grid then
   covers workgroups * 256 tokens, which lets small test shapes reach the whole-passes + left-over-rows split of the engine's route */
int gg_test_set_enc_grid(int workgroups);
int gg_test_ffn2(const void* X, int64_t M, const float* W1, const float* b1, const float* W2, const float* b2, void* Hs, void* R2, int r2_bf16,
                 int64_t keep_rows, const float* ln_g, const float* ln_b, void* Y, int y_bf16, float* stats, float drop_p, uint64_t drop_seed,
                 uint32_t site1, uint32_t site2, uint32_t drop_call, void* wfrag, int variant, void* stream);
/* dW [N,K] += dY [M,N]^T X [M,K] over the token rows (+ optional FiLM on X, FiLM-gradient contraction, bias column sums) */
int gg_test_wgrad(const void* dY, int64_t ldy, int dy_bf16, const void* X, int64_t ldx, int x_bf16, float* dW, int64_t ldw, int64_t M,
                  int N, int K, const float* film_g, const float* film_b, int64_t film_ld, int film_group, const float* fgrad_W,
                  int64_t fgrad_ldw, float* dgamma, float* dbeta, int64_t fgrad_ld, int fgrad_tokens, float* dbias, int64_t x_mod,
                  int x3 /* split-operand (bf16x3) products of fp32 operands */, void* stream);
/* projection-free single-query attention sweeps (R:218-219 restated): qt [N,nh,E], x [N,S,E] -> probs [N,nh,S], xbar [N,nh,E] */
int gg_test_sqx_fwd(const float* qt, const float* x, const uint8_t* mask, int mask_B, float* probs, float* xbar, int N, int S, int E,
                    int nh, void* stream);
int gg_test_sqx_bwd(const float* dxbar, const float* qt, const float* xbar, const float* x, const float* probs, float* dx, float* dqt,
                    int N, int S, int E, int nh, void* stream);
/* fused MLP head (csrc/head.hip; R:226-231).  forward: a1 [rows,H] holds the gene / latent part of the first layer on entry and
 * act(a1 + cvec W1c^T + b1) on return, a2 = act(a1 W2^T + b2), out[r] = a2[r] . w3 + b3 for r < out_rows (out may be null).
 * backward: dout [rows] (score gradient; null: dh2 holds dout W3 on entry) -> dh2, dh1 [rows,H], dcond [rows,E] (may be null).
 * W1c: [H][E] slice of the first-layer weight, row stride ldw1.  LeakyReLU(slope). */
/* fused backward of an encoder layer's token-local chain behind LayerNorm2's backward (csrc/enc.hip encb_kernel; torch transformer.py:961-983
 * differentiated): dx [M,256] fp32 in (dr2, the gradient w.r.t. the pre-LN2 sum) and out (dr1, w.r.t. the pre-LN1 sum); dres2 bf16 [M,256] the masked
 * branch gradient of LayerNorm2; Wcat = linear1.weight | linear2.weight | out_proj.weight (fp32, contiguous); r1 bf16 pre-LN1 sums, st1 (mean, rstd)
 * [M,2], g1 norm1.weight, h bf16 [M,512] stored hidden activations; outputs bf16: dh [M,512], dres1 [M,256], dctx [M,256]; colsums [3][256] +=
 * dgamma1, dbeta1, d(out_proj.bias) */
int64_t gg_test_enc_bwd_frag_bytes(void);
int gg_test_enc_bwd(float* dx, int64_t M, const float* Wcat, const void* dres2, const void* h, const void* r1, const float* st1, const float* g1,
                    void* dh, void* dres1, void* dctx, float* colsums, float drop_p, uint64_t drop_seed, uint32_t site1, uint32_t drop_call,
                    void* wfrag, void* stream);
int gg_test_head_fwd(int64_t rows, int H, int E, float slope, const float* W1c, int64_t ldw1, const float* b1, const float* W2,
                     const float* b2, const float* w3, const float* b3, const float* cvec, float* a1, float* a2, float* out,
                     int64_t out_rows, void* stream);
int gg_test_head_bwd(int64_t rows, int H, int E, float slope, const float* W1c, int64_t ldw1, const float* W2, const float* w3,
                     const float* a1, const float* a2, const float* dout, float* dh2, float* dh1, float* dcond, void* stream);
/* LayerNorm backward with the dropout-masked branch gradient (fp32 or bf16) and the fused bias / gamma / beta column sums;
 * dres_bf16: bit 0 = bf16 branch-gradient output, bit 1 = r is a bf16 array (E = 256 only) */
int gg_test_ln_bwd(const float* dy, const float* r, const float* stats, const float* g, float* dr, void* dres_out, float* dgamma,
                   float* dbeta, float* dbias, int64_t rows, int E, float drop_p, uint64_t drop_seed, uint32_t drop_site,
                   uint32_t drop_call, int dres_bf16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEMMGAN_LAB_H */
