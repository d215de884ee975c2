// Row-wise / elementwise HIP kernels of the engine (host launch wrappers).  All pointers are
// device pointers, all tensors fp32 row-major contiguous unless a leading dimension is given.
#pragma once
#include "gg_common.h"

namespace gg {

// counter-hash dropout stream: element i of site `site` in forward call `call` (see kernels.hip)
struct DropKey {
    float p = 0.f;          // drop probability; 0 => disabled
    uint32_t k0 = 0;        // 32-bit mix of (seed, site, call)
    uint32_t thr = 0;       // round(p * 65536): element kept iff its 16-bit uniform >= thr
    // Device word mixed into k0 by every kernel at its start (drop_live): 0 for eagerly enqueued work, the replay number
    // of a captured train step (hipGraph: kernel arguments, hence k0, are frozen at capture) - fresh masks per replay.
    const uint32_t* epoch = nullptr;
    // Added to k0 AFTER the epoch re-mix: a launch over the row range [r0, M) of a tensor whose stream index is row * ld + col
    // passes post = (r0 * ld / 2) * DROP_PHI and draws, with LOCAL row indices, the very words the whole-tensor launch draws for
    // those rows (the pre-mix state is linear in the pair index, drop_rng.h) - drop_key_rows() below.
    uint32_t post = 0;
};
DropKey make_drop_key(float p, uint64_t seed, uint32_t site, uint32_t call);

int k_fill(float* x, long n, float v, hipStream_t st);
int k_copy(float* dst, const float* src, long n, hipStream_t st);
// dst[r, :] = src[r % src_rows, :]   (replica broadcast)
int k_copy_rows_bcast(float* dst, const float* src, long rows, long src_rows, int cols, hipStream_t st);
int k_axpy(float* y, const float* x, float a, long n, hipStream_t st);          // y += a*x
int k_add_bcast_rows(float* y, const float* x, long rows, long x_rows, int cols, hipStream_t st);  // y[r]+=x[r%x_rows]

// FiLM head: gb_pre [B, 2*Dp] -> gb (gamma = tanh(first half), beta = clamp(second half, -5, 5))
int k_film_act_fwd(const float* gb_pre, float* gb, int B, int Dp, hipStream_t st);
// d(gb_pre) from d(gamma|beta) (in dgb, overwritten), needs gb (post) and gb_pre
int k_film_act_bwd(float* dgb, const float* gb, const float* gb_pre, int B, int Dp, hipStream_t st);
// mod[b,p,:] = gamma[b,:]*patches[b,p,:] + beta[b,:]
int k_film_mod(const float* patches, const float* gb, float* mod, int B, int P, int Dp, hipStream_t st);
// dgb[b, 0:Dp] = sum_p dmod*patches ; dgb[b, Dp:2Dp] = sum_p dmod
int k_film_bwd_reduce(const float* dmod, const float* patches, float* dgb, int B, int P, int Dp, hipStream_t st);

// seq[b,0,:] = cls[:]  for b in [0,B)
int k_write_cls(float* seq, const float* cls, int B, int S, int E, hipStream_t st);
// dcls[:] += sum_b dseq[b,0,:]
int k_cls_grad(const float* dseq, float* dcls, int B, int S, int E, hipStream_t st);
// mask_out[b,0]=0 ; mask_out[b,1+p] = pad[b,p]   (bytes)
int k_build_mask(const uint8_t* pad, uint8_t* mask_out, int B, int P, hipStream_t st);

// in-place masked softmax over rows of length cols (masked entries already hold -inf).  If
// drop.p > 0 also writes Pd = P * keep / (1-p) (row r of replica-stacked Pd reads P row r).
int k_softmax_rows(float* S, float* Pd, long rows, int cols, DropKey drop, hipStream_t st);
// dP (in: d(Pd) [rows, cols]) -> dS = P * (dP_eff - sum(dP_eff*P)) * scale, in place
int k_softmax_bwd_rows(float* dP, const float* P, long rows, int cols, float scale, DropKey drop, hipStream_t st);

// r = x[row % x_rows] + drop(res[row]) ; y = LayerNorm(r)*g + b ; res <- r ; stats[row] = (mean, rstd)
int k_add_layernorm_fwd(const float* x, long x_rows, float* res, const float* g, const float* b, float* y,
                        float* stats, long rows, int E, DropKey drop, hipStream_t st);
// dr = LN backward of dy wrt r (r, stats saved) ; dgamma/dbeta accumulated atomically.
// dres_out (may be null) = drop-masked dr (gradient w.r.t. the un-dropped `res` branch)
// dbias (may be null) += column sums of that masked branch gradient (= bias gradient of the Linear
// that produced the residual branch), saving a separate pass over the tensor.
int k_layernorm_bwd(const float* dy, const float* r, const float* stats, const float* g, float* dr,
                    void* dres_out, float* dgamma, float* dbeta, float* dbias, long rows, int E, DropKey drop, hipStream_t st,
                    int dres_bf16 = 0);

// out[n] += sum_rows X[r, n]
int k_colsum(const void* X, long rows, int N, long ld, float* out, hipStream_t st, int x_bf16 = 0);
// out[n] += sum_rows X[r,n] * (ref[r,n] > 0 ? 1 : slope)
int k_colsum_masked(const float* X, const float* ref, long rows, int N, float slope, float* out, hipStream_t st);
// y *= (ref > 0 ? 1 : slope) * scale   (activation backward using the POST-activation value)
int k_act_bwd(float* y, const float* ref, long n, float slope, float scale, hipStream_t st);
// y = act(y + bias[col])   (bias may be null)
int k_bias_act(float* y, const float* bias, long rows, int N, int act, float slope, hipStream_t st);
// x *= keep/(1-p)
int k_dropout(float* x, long n, DropKey drop, hipStream_t st);
// y[r,:] *= s[r]
int k_rowscale(float* y, const float* s, long rows, int N, hipStream_t st);
// out[r,n] = (ref[r,n] > 0 ? 1 : slope) * w[n]
int k_mask_times_vec(float* out, const float* ref, const float* w, long rows, int N, float slope, hipStream_t st);

// single-query multi-head attention (query length 1): q [B,E] (projected), kv [B,S,2E] (K|V projected)
// mask [B,S] bytes (nonzero = ignore).  probs [B,nh,S], ctx [B,E].
// mask row of sample b is b % mask_B (replica-stacked batches share the mask of the original batch)
// kv_B > 0: the keys / values of sample b are block b % kv_B (replicas share one projection; needs sq_attn_shared_ok)
int k_sq_attn_fwd(const float* q, const float* kv, const uint8_t* mask, int mask_B, float* probs, float* ctx,
                  int B, int S, int E, int nh, hipStream_t st, int kv_B = 0);
// dctx [B,E] -> dq [B,E] (overwritten), dkv [B,S,2E] (overwritten)
int k_sq_attn_bwd(const float* dctx, const float* q, const float* kv, const float* probs, float* dq, float* dkv,
                  int B, int S, int E, int nh, hipStream_t st);
// R replica-stacked queries (row r*B + b) over ONE key/value projection kv [B,S,2E]: dq [R*B,E] per replica, dkv [B,S,2E] is
// the SUM over the replicas (what the projection's weight / data gradients need: they are linear in it)
bool sq_attn_shared_ok(int S, int E, int nh, int R);
int k_sq_attn_bwd_shared(const float* dctx, const float* q, const float* kv, const float* probs, float* dq, float* dkv,
                         int B, int R, int S, int E, int nh, hipStream_t st);

// dst[r*ldd + c] = src[(r % src_rows)*lds + c], c < cols   (replicate selected rows of a strided matrix)
int k_copy_rows_strided_bcast(float* dst, long ldd, const float* src, long lds, long rows, long src_rows, int cols, hipStream_t st);
// dst[b*ldd + c] += sum_{r<R} src[(r*B + b)*cols + c]   (fold replica-stacked rows into a strided destination)
int k_fold_rows_add(float* dst, long ldd, const float* src, long B, int R, int cols, hipStream_t st);
// out[i] = sum_{r<R} in[r*n + i]   (fold replica-stacked gradients)
int k_fold(float* out, const float* in, long n, int R, hipStream_t st);
// out[(b*P+p), :] = seq[b, 1+p, :]   (drop the CLS row: [B,P+1,E] -> [B*P,E])
int k_gather_patch_rows(float* out, const float* seq, int B, int P, int E, hipStream_t st);
// c = a + b row-wise ([rows, E]); rows of samples (row % B) with pad[b] != 0 become NaN
int k_sum2_nan_rows(float* c, const float* a, const float* b, const uint8_t* pad, long rows, int B, int E, hipStream_t st);
// demb [B*(S-1), E] = patch rows, cls_rows [B, E] = CLS rows of the sum over the R replicas of dx [R, B, S, E]
int k_fold_gather(float* demb, float* cls_rows, const float* dx, int B, int S, int E, int R, hipStream_t st);
// the inverse: seq[b, 1+p, :] = in[(b*P+p), :]
int k_scatter_patch_rows(float* seq, const float* in, int B, int P, int E, hipStream_t st);
// out = in * keep/(1-p)   (re-materialise dropped attention probabilities in backward)
int k_dropout_copy(float* out, const float* in, long n, DropKey drop, hipStream_t st);

// critic outputs -> losses and seeds: losses[0]+= -mean(d_true) ; losses[1] += mean(d_fake)
// (d layout: [2B] = fake rows then real rows) ; seed[r] = +1/B (fake) / -1/B (real)
int k_critic_loss_seed(const float* d, float* seed, float* losses, int B, hipStream_t st);
// generator: losses[3] = -mean(d_fake) ; seed[r] = -1/B
int k_gen_loss_seed(const float* d, float* seed, float* losses, int B, hipStream_t st);

// h1_hat = alpha*P_real + (1-alpha)*P_fake  (P rows: [fake(B) ; real(B)], alpha [B])
int k_lerp_rows(const float* Pfr, const float* alpha, float* out, int B, int H, hipStream_t st);
// x_hat = alpha*x_real + (1-alpha)*x_fake  (xfr rows: [fake(B) ; real(B)])
int k_lerp_genes(const float* xfr, const float* alpha, float* out, int B, int G, hipStream_t st);
// nrm2[r] = sum_n X[r,n]^2
int k_row_sumsq(const float* X, float* out, long rows, int N, hipStream_t st);
// coef[r] = gp_weight*(2/B)*(nrm-1)/nrm ; losses[2] += mean((nrm-1)^2)
int k_gp_coef(const float* nrm2, float* coef, float* losses, int B, float gp_weight, hipStream_t st);

// fused gradient-penalty chain (gpchain.hip; exact fp32 in both precision modes) -----------------------------------------
// g1 = m1 * ((m2 * w3) W2) for the B interpolate rows (a1 / a2: post-activation head values); zeroes nrm2 [B], dg1pre [B,H]
int k_gp_front(const float* a1, const float* a2, const float* w3, const float* W2, float* g1, float* dg1pre, float* nrm2, int B,
               int H, float slope, hipStream_t st);
// grad [B,G] = g1 [B,H] W1x, W1x[k,g] = W1[k*ldw + g]; nrm2[b] += sum_g grad[b,g]^2 in the epilogue
int k_gp_grad(const float* g1, const float* W1, long ldw, float* grad, float* nrm2, int B, int H, int G, hipStream_t st);
// the same product on the split-operand strip kernel (round 3): 32 genes x up to 256 rows per workgroup, six bf16 part products per tile
// (fp32-grade), nrm2p [gp_grad3_parts(G)][B] = per-strip partial sums of squares (no atomics; k_gp_coef_scale adds them in order)
int gp_grad3_parts(int G);
bool gp_grad3_ok(const float* g1, const float* W1, long ldw, const float* grad, int B, int H, int G);
int k_gp_grad3(const float* g1, const float* W1, long ldw, float* grad, float* nrm2p, int B, int H, int G, hipStream_t st);
// coef[b] = gp_weight*(2/B)*(nrm-1)/nrm ; *loss += mean((nrm-1)^2) ; g1s = coef * g1 (g1s may be null); nparts > 1: nrm2 = [nparts][B] partials
int k_gp_coef_scale(const float* nrm2, const float* g1, float* coef, float* g1s, float* loss, int B, int H, float gp_weight,
                    hipStream_t st, int nparts = 1, float* nrm2_total = nullptr);      // nrm2_total [B] (optional): the summed squares
// du = m1 * coef * dg1pre ; dW2 += (m2*w3)^T du ; dw3 += sum_b m2 * (du W2^T)   (atomic adds)
int k_gp_tail(const float* dg1pre, const float* coef, const float* a1, const float* a2, const float* w3, const float* W2, float* dW2,
              float* dw3, int B, int H, float slope, hipStream_t st);

void gp_time_next(hipEvent_t begin, hipEvent_t end);     // the next k_gp_* launch stamps these at the kernel's own begin / end

// fused self-attention (attention.hip): bf16 MFMA, no [S,S] tensor in HBM -----------------------------------
bool flash_attn_supported(int S, int E, int nh);
const char* flash_attn_kernel_name(int which, int S, int E, int nh);      // 0 forward, 1 dQ, 2 dK/dV: the kernel this shape runs on
// qkv [N,S,3E] packed projections; mask [mask_B,S] bytes (row n % mask_B); ctx [N,S,E]; lse2 [N,nh,S].
// io_bf16: qkv / ctx / dctx / dqkv are bf16 tensors (they only ever feed bf16 MFMA operands), else fp32.
// qkv_B > 0: qkv holds the projection of the first qkv_B samples only and sample n reads sample n % qkv_B (dropout
// replicas stacked on the batch axis share the layer-0 projection); outputs are always per sample
int flash_attn_fwd(const void* qkv, const uint8_t* mask, int mask_B, void* ctx, float* lse2, long N, int S, int E, int nh,
                   DropKey drop, int io_bf16, hipStream_t st, long qkv_B = 0);
// dctx [N,S,E] -> dqkv [N,S,3E] (fully overwritten); delta [N,nh,S] scratch
int flash_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse2, float* delta, const uint8_t* mask,
                   int mask_B, void* dqkv, long N, int S, int E, int nh, DropKey drop, int io_bf16, hipStream_t st, long qkv_B = 0, hipEvent_t ev_mid = nullptr);

// fused feed-forward block of an encoder layer, forward (ffn.hip): x2 = LN(x1 + drop(W2 drop(relu(W1 x1 + b1)) + b2)), bf16 MFMA,
// E = 256, F = 512; h (bf16) and the pre-LayerNorm sum r2 (fp32) are stored for rows < keep_rows only (-1: all)
struct FfnP {
    const float* X = nullptr; long M = 0;           // x1 [M, E] fp32: the input and the residual
    int E = 0, F = 0;
    const void* W1 = nullptr; const float* b1 = nullptr;     // bf16 [F][E]
    const void* W2T = nullptr; const float* b2 = nullptr;    // bf16 W2^T [F][E] (the transposed shadow of linear2.weight [E][F])
    void* Hs = nullptr;                             // bf16 [M, F]
    float* R2 = nullptr;                            // fp32 [M, E]
    long keep_rows = -1;
    const float* ln_g = nullptr; const float* ln_b = nullptr; float* Y = nullptr; float* stats = nullptr;
    DropKey drop1, drop2;                           // inner (element index = token * F + f) and post-FFN (token * E + n) dropout
};
bool ffn_fused_supported(const FfnP& p);
int ffn_fused(const FfnP& p, hipStream_t st);
void ffn_time_next(hipEvent_t begin, hipEvent_t end);

// fused feed-forward block, round 4 (enc.hip): same arithmetic and dropout streams as FFN1 + FFN2 of the weight-stationary route with bf16-stored
// LayerNorm outputs: X = x1 (bf16, input and residual), weights as a fragment-ordered bf16 stream (k_enc_frag_weights), h (bf16) / r2 (bf16 or
// fp32) / statistics stored for rows < keep_rows only (-1: all), Y = x2 (bf16 or fp32) for every row
struct Ffn2P {
    const void* X = nullptr; long M = 0;            // bf16 [M, 256]
    const void* Wf = nullptr;                       // this layer's fragment stream (enc_frag_bytes(1) bytes)
    const float* b1 = nullptr; const float* b2 = nullptr; const float* ln_g = nullptr; const float* ln_b = nullptr;
    void* Hs = nullptr;                             // bf16 [M, 512]
    void* R2 = nullptr; int r2_bf16 = 0;            // pre-LayerNorm sum [M, 256]
    float* stats = nullptr;                         // [M, 2] (mean, rstd)
    void* Y = nullptr; int y_bf16 = 0;              // [M, 256]
    long keep_rows = -1;
    DropKey drop1, drop2;                           // inner (element index = token * 512 + f) and post-FFN (token * 256 + n) dropout
    unsigned* stamps = nullptr;                     // tools/ffn2_probe.py: per-phase cycle sums, [workgroups * waves][8]
};
bool ffn2_supported(const Ffn2P& p);
void enc_set_grid(int workgroups);               // tests: cap the persistent grids of enc.hip (0: one workgroup per compute unit)
long ffn2_sweep_tokens(int variant);            // tokens one pass of the persistent grid covers (workgroup tokens x compute units)
// variant: 0 = 4-slot weight ring (64 KB), 2 = 8-slot ring (128 KB)
int ffn2(const Ffn2P& p, hipStream_t st, int variant = 0);
void ffn2_time_next(hipEvent_t begin, hipEvent_t end);
inline size_t enc_frag_bytes(int nl) { return (size_t)nl * 32 * 16384; }      // bytes of the fragment-ordered image of nl layers: 32 ring slots of 16 KB each (enc.hip FFN_SLOTS, SLOT)
// out[layer] = fragment stream of linear1.weight [512][256] at w + w1_off[layer] and linear2.weight [256][512] at w + w2_off[layer]
int k_enc_frag_weights(const float* w, const long* w1_off, const long* w2_off, int nl, void* out, hipStream_t st);

// fused backward of the token-local chain of an encoder layer behind LayerNorm2's backward (enc.hip): gated hidden gradient -> dx1 += -> LayerNorm1
// backward -> context gradient in one launch; bf16 tensors as in the default bf16 mode
struct EncBwdP {
    float* dx = nullptr; long M = 0;                // out: dr1 [M,256] fp32 (gradient w.r.t. the pre-LN1 sum)
    const float* dr2 = nullptr;                     // in: gradient w.r.t. the pre-LN2 sum, fp32 [M,256] (null: in dx, overwritten in place)
    const void* Wf = nullptr;                       // this layer's backward fragment stream (encb_frag_bytes(1) bytes)
    const void* dres2 = nullptr;                    // bf16 [M,256] masked branch gradient of LayerNorm2 (ln_bwd_v4_k)
    const void* h = nullptr;                        // bf16 [M,512] stored hidden activations (post ReLU and dropout)
    const void* r1 = nullptr; const float* st1 = nullptr; const float* g1 = nullptr;     // bf16 pre-LN1 sums, (mean, rstd) [M,2], norm1.weight
    void* dh = nullptr; void* dres1 = nullptr; void* dctx = nullptr;          // bf16 [M,512], [M,256], [M,256]
    float* dg1 = nullptr; float* db1 = nullptr; float* dbias1 = nullptr;      // += norm1.weight / norm1.bias / out_proj.bias gradients [256]
    DropKey drop1;                                  // post-attention dropout (site 1), element index = token * 256 + n
    float gate_scale = 1.f;                         // 1 / (1 - p) of the inner (FFN) dropout: dh = (dres2 W2) * [h > 0] * gate_scale
    unsigned* stamps = nullptr;                     // tools/encb_probe.py: per-phase cycle sums, [workgroups * 8][8]
};
bool enc_bwd_supported(const EncBwdP& p);
int enc_bwd(const EncBwdP& p, hipStream_t st);
inline size_t encb_frag_bytes(int nl) { return (size_t)nl * 40 * 16384; }     // 40 slots per layer (enc.hip BWD_SLOTS)
// out[layer] = backward fragment stream of linear2.weight [256][512] (as W2^T), linear1.weight [512][256] (as W1^T) and out_proj.weight [256][256] (as Wo^T)
int k_encb_frag_weights(const float* w, const long* w1_off, const long* w2_off, const long* wo_off, int nl, void* out, hipStream_t st);

// split-operand (bf16x3) attention of GG_PREC_BF16X3: fp32 qkv / ctx / dctx / dqkv, `ns` bf16 parts per MFMA operand (2: three
// products per tile - the backward form; 3: six products, fp32-grade - the forward form); any S <= 2048 (keys / queries streamed)
bool flash_attn_x3_supported(int S, int E, int nh);
const char* flash_attn_x3_kernel_name(int which);
int flash_attn_fwd_x3(const float* qkv, const uint8_t* mask, int mask_B, float* ctx, float* lse2, long N, int S, int E, int nh,
                      DropKey drop, hipStream_t st, long qkv_B, int ns);
int flash_attn_bwd_x3(const float* qkv, const float* ctx, const float* dctx, const float* lse2, float* delta, const uint8_t* mask,
                      int mask_B, float* dqkv, long N, int S, int E, int nh, DropKey drop, hipStream_t st, long qkv_B, int ns,
                      hipEvent_t ev_mid = nullptr);

// token-on-lane Linear (tlin.hip): Y[M,N] = epi(X[M,K] W[N,K]^T), bf16 MFMA, activations read once ---------------
struct TlinP {
    const void* X = nullptr; long ldx = 0; long M = 0; int x_bf16 = 0;       // activations fp32 or bf16 (row stride in elements)
    const void* W = nullptr; long ldw = 0;          // bf16 [N][K] (row stride ldw elements)
    long w_part_stride = 0;                         // tlin3 only: W is the first of NS bf16 part images, this many elements apart
    const float* bias = nullptr;
    void* Y = nullptr; long ldy = 0; int y_bf16 = 0;                          // output fp32 or (stream mode only) bf16
    long y_rows = -1;                               // resident + LayerNorm: Y (pre-LN sum) is stored for tokens < y_rows only (-1: all);
                                                    // forward-only replicas never read it back
    int N = 0, K = 0;
    const float* film_g = nullptr; const float* film_b = nullptr; long film_ld = 0; int film_group = 0;   // X' = g*X + b
    int y_row_group = 0;                            // output row m -> m + m / group + 1
    int act_relu = 0;
    DropKey drop; long drop_ld = 0;                 // dropout of the Linear output, element index = token*drop_ld + n
    const void* mask_ref = nullptr; long ldref = 0; float mask_scale = 1.f; int mask_bf16 = 0;   // y = ref > 0 ? y*scale : 0
    int accumulate = 0;                             // y += previous content
    const float* res = nullptr; long ldres = 0; long res_rows = 1;             // + res[token % res_rows]
    const float* ln_g = nullptr; const float* ln_b = nullptr; float* ln_y = nullptr; float* ln_stats = nullptr;
    // weight-stationary "+= then LayerNorm backward" (wst.hip EPI_LNB; N = 256, K = 512):  dy = Y + X W^T is the gradient w.r.t. a LayerNorm
    // output whose pre-LN sum is `res` (fp32) with statistics `ln_stats` (read) and weight `ln_g`; written: ln_y = dr (gradient w.r.t.
    // the pre-LN sum, fp32), lnb_dres = bf16(dr * dropout mask / (1-p)) (the branch gradient; key `drop`, stride drop_ld), and the
    // column sums lnb_dgamma += sum dy*xhat, lnb_dbeta += sum dy, lnb_dbias += sum of the masked dr (atomic adds)
    void* lnb_dres = nullptr; float* lnb_dgamma = nullptr; float* lnb_dbeta = nullptr; float* lnb_dbias = nullptr;
    int res_bf16 = 0, ln_y_bf16 = 0;                // weight-stationary LN kernels only: the residual rows / the LayerNorm output are bf16 arrays (same strides in
                                                    // elements): the encoder's LN outputs feed only MFMA operands and residual adds, so they are stored once, in bf16
    // fp8 (OCP e4m3) operands: W points at the e4m3 shadow copy (ldw in elements), *w_exp (device) is its per-tensor
    // power-of-two exponent (stored value = w * 2^w_exp), activations are quantised as x * 2^x_exp on load
    int fp8 = 0; const int* w_exp = nullptr; int x_exp = 0;
    int grid_pct = 0;                               // weight-stationary kernels: share of the compute units the persistent grid takes (0: 91, wst.hip launch())
    int dbg = 0;                                    // timing experiments only (GG_TLIN_DBG): 1 no stores, 2 no MFMA, 4 no weight loads, 8 no X loads
    unsigned long long* stamps = nullptr;           // tools/tlin_probe: 4 s_memtime stamps per workgroup (wave 0)
};
bool tlin_supported(const TlinP& p);
bool tlin_fp8_supported(const TlinP& p);
// weight-stationary variant of the Linear + dropout + residual + LayerNorm calls (wst.hip): N = 256, K in {256, 512}, bf16 X.
// tlin() routes such calls there (GG_NO_WST=1 in the environment keeps the token-on-lane kernel, for A/B runs)
bool wst_ln_supported(const TlinP& p);
int wst_ln(const TlinP& p, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int wst_kind(const TlinP& p);          // 0: none; 1 += (N 256, K 512); 2 relu/dropout -> bf16 (N 512, K 256, fp32 X); 3 gated -> bf16
int wst_other(const TlinP& p, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int wst_fp8_kind(const TlinP& p);      // fp8 operands: 0 none; 1 / 2 LN (K 256 / 512); 3 FFN1 (N 512); 4 QKV (N 768)
int wst_fp8(const TlinP& p, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);            // p.fp8 set and one of the fp8 instantiations takes the shape
int tlin(const TlinP& p, hipStream_t st);
void tlin_time_next(hipEvent_t begin, hipEvent_t end);     // the next tlin() launch stamps these at the kernel's own begin / end
// 0 stream (K != 256), 1 resident 32-token, 2/3/4 resident 16-token (+res+LN / += / other),
// 16 + XB + 2*YB + 4*EPI: the tlin_str_kernel<256, XB, YB, EPI> instantiation
int tlin_kernel_class(const TlinP& p);
// split-operand (bf16x3) Linear of GG_PREC_BF16X3 (tlin3.hip): same TlinP contract with fp32 X, fp32 W (the master copy, [N][K]),
// fp32 Y and fp32 gate reference; three bf16 MFMAs per product tile on hi / lo splits, fp32 accumulate
bool tlin3_supported(const TlinP& p);
int tlin3(const TlinP& p, hipStream_t st, int nsplit = 2);      // nsplit 3: hi + mid + lo, six products (fp32-grade: forward passes)
void tlin3_time_next(hipEvent_t begin, hipEvent_t end);
// bf16 shadow copies of the 2-D weights: wb = bf16(W) [rows][cols], wtb = bf16(W^T) [cols][rows], same offsets
struct ShadowEntry { long off; int rows, cols; };
int k_shadow_weights(const float* w, void* wb, void* wtb, const ShadowEntry* tab_dev, int n_entries, hipStream_t st);
// bf16x3 mode: wt = W^T [cols][rows] in fp32 at the same offsets
int k_shadow_weights_t32(const float* w, float* wt, const ShadowEntry* tab_dev, int n_entries, hipStream_t st);
// bf16x3 mode: the three bf16 parts (hi, mid, lo) of every shadowed weight in both orientations: wp[sp * part_stride + off ..] = part sp
// of W [rows][cols], wtp likewise of W^T [cols][rows]
int k_shadow_parts(const float* w, void* wp, void* wtp, long part_stride, const ShadowEntry* tab_dev, int n_entries, hipStream_t st);
// parts[sp * part_stride + i] = part sp of w[i], i < n (tests)
int k_split_weights(const float* w, void* parts, long n, long part_stride, int nparts, hipStream_t st);
// e4m3 shadow: w8[2*off + i] = e4m3(w[off + i] * 2^w_exp[entry]), w_exp[entry] = floor(log2(448 / max|w|)); amax: scratch [n_entries]
int k_shadow_weights_fp8(const float* w, void* w8, unsigned* amax, int* w_exp, const ShadowEntry* tab_dev, int n_entries, hipStream_t st);

// fused MLP head (head.hip): forward  a1 = act(a1 + cvec W1c^T + b1), a2 = act(a1 W2^T + b2), out = a2 w3 + b3 (critic);
// backward  dh2 = (dout w3^T | dh2) * act'(a2), dh1 = (dh2 W2) * act'(a1), dcond = dh1 W1c.  fp32 tensors, bf16 MFMA operands.
struct HeadP {
    long rows = 0; int H = 0, E = 0; float slope = 0.f;
    const float* W1c = nullptr; long ldw1 = 0;          // [H][E] slice of the first layer's weight (row stride ldw1)
    const float* b1 = nullptr; const float* W2 = nullptr; const float* b2 = nullptr;      // W2 [H][H] dense
    const float* w3 = nullptr; const float* b3 = nullptr;                                 // [H], [1]: the critic's score column
    const float* cvec = nullptr;                        // [rows][E]
    float* a1 = nullptr; float* a2 = nullptr;           // [rows][H]; forward: a1 holds the gene / latent part on entry
    float* out = nullptr; long ldo = 1; long out_rows = 0;                                // forward: scores of rows < out_rows (null: none)
    const float* dout = nullptr;                        // backward, critic: d(score) [rows]; null: dh2 holds dout W3 on entry
    float* dh2 = nullptr; float* dh1 = nullptr; float* dcond = nullptr;                   // [rows][H], [rows][H], [rows][E] (dcond may be null)
};
bool head_fused_supported(const HeadP& p);
int head_fwd(const HeadP& p, hipStream_t st);
int head_bwd(const HeadP& p, hipStream_t st);

// single-query attention over un-projected keys/values (sqattn.hip) ------------------------------------------------
bool sqx_supported(int S, int E, int nh);
// q [N,E] projected query; x [N,S,E]; Win [3E,E], bin [3E] packed in-proj; probs [N,nh,S], xbar [N,nh,E], ctx [N,E]
int sqx_attn_fwd(const float* q, const float* x, const float* Win, const float* bin, const uint8_t* mask, int mask_B, float* probs,
                 float* xbar, float* ctx, int N, int S, int E, int nh, hipStream_t st);
// dctx [N,E] -> dx [N,S,E], dq [N,E], dqt [N,nh,E]  (dWk_h += q_h (x) dqt_h and dWv_h += dctx_h (x) xbar_h are the caller's GEMMs)
int sqx_attn_bwd(const float* dctx, const float* q, const float* x, const float* Win, const float* probs, float* dx, float* dq,
                 float* dqt, int N, int S, int E, int nh, hipStream_t st);

// streaming variant: the per-head projections are the caller's batched GEMMs; x is read once per pass
//   forward : qt [N,nh,E] (= Wk_h^T q_h) -> probs [N,nh,S], xbar [N,nh,E]
//   backward: dxbar [N,nh,E] (= Wv_h^T dctx_h), qt, xbar, probs -> dx [N,S,E] (overwritten), dqt [N,nh,E]
bool sqx_stream_supported(int S, int E, int nh);
int sqx_stream_fwd(const float* qt, const float* x, const uint8_t* mask, int mask_B, float* probs, float* xbar, int N, int S, int E,
                   int nh, hipStream_t st);
int sqx_stream_bwd(const float* dxbar, const float* qt, const float* xbar, const float* x, const float* probs, float* dx, float* dqt,
                   int N, int S, int E, int nh, hipStream_t st);

// weight gradient dW[N,K] += dY[M,N]^T X[M,K] over long token reductions (wgrad.hip), operands fp32 or bf16 --------
bool wgrad_supported(const void* dY, long ldy, int dy_bf16, const void* X, long ldx, int x_bf16, long M, int N, int K);
// optional FiLM on the X operand: X' = g[token / group] * X + b[token / group]  (rows of g / b are ld floats apart)
struct WgradFilm { const float* g = nullptr; const float* b = nullptr; long ld = 0; int group = 0; };
// FiLM-gradient mode: nothing is added to dW; per sample b (`tokens` rows each) the panel C_b = dY_b^T X_b is contracted
// with W [N,K]:  dgamma[b,k] += sum_n W[n,k] C_b[n,k],  dbeta[b,k] += sum_n W[n,k] sum_tokens dY_b[token,n]
struct WgradFilmGrad { const float* W = nullptr; long ldw = 0; float* dgamma = nullptr; float* dbeta = nullptr; long ld = 0; int tokens = 0; };
int wgrad(const void* dY, long ldy, int dy_bf16, const void* X, long ldx, int x_bf16, float* dW, long ldw, long M, int N, int K,
          hipStream_t st, const WgradFilm* film = nullptr, const WgradFilmGrad* fgrad = nullptr, float* dbias = nullptr,
          long x_mod = 0, int x3 = 0);        // x_mod > 0: X holds x_mod rows, row m is read at m % x_mod
// x3: split-operand (bf16x3) products of fp32 operands: hi / lo images in LDS, three MFMAs per tile (GG_PREC_BF16X3)
// dbias (optional): dbias[n] += sum_m dY[m, n] with dY as the kernel sees it (bf16 operand values) - the Linear's bias gradient

// optimiser ---------------------------------------------------------------------------------------
// partials[0 .. *n_partials) = per-workgroup sums of squares (<= 1024 slots, no atomics: deterministic)
int k_sumsq(const float* x, long n, float* partials, int* n_partials, hipStream_t st);
enum OptKind { OPT_RMSPROP = 0, OPT_ADAM = 1, OPT_ADAMW = 2 };
// clip coefficient = min(1, max_norm/(sqrt(sum of the partials)+1e-6)) when max_norm > 0, else 1 (the partials are added in
// one fixed order by every workgroup).  grad_scale is an extra factor applied to every gradient first (1/world_size after
// a sum all-reduce).
int k_opt_step(float* w, const float* g, float* s1, float* s2, long n, int kind, float lr, float max_norm,
               const float* partials, int n_partials, float grad_scale, int step_t, const uint32_t* t_off, hipStream_t st);
// (t_off: optional device word added to step_t for the Adam bias corrections, which are computed on the device)


// ---- opt-in kernels outside the product library ---------------------------------------------------------------------------------
// The kernels that were built, tested and LOST their A/B against the default routes (ffn.hip, enc.hip, head.hip)
// live in libgemmgan_lab.so together with the kernel-level test hooks.  Loading that library registers its entry points here
// (gg_lab_register, include/gemmgan_lab.h); the engine's opt-in routes run only when the pointer is set, and the switches that
// select them (gg_set_ffn2, gg_set_encb, gg_set_ffn_fused, gg_set_head_fused, the GG_* environment forms) fail loudly when it is not.
struct LabTable {
    bool (*ffn_fused_supported)(const FfnP&) = nullptr;
    int (*ffn_fused)(const FfnP&, hipStream_t) = nullptr;
    bool (*ffn2_supported)(const Ffn2P&) = nullptr;
    int (*ffn2)(const Ffn2P&, hipStream_t, int) = nullptr;
    long (*ffn2_sweep_tokens)(int) = nullptr;
    int (*k_enc_frag_weights)(const float*, const long*, const long*, int, void*, hipStream_t) = nullptr;
    bool (*enc_bwd_supported)(const EncBwdP&) = nullptr;
    int (*enc_bwd)(const EncBwdP&, hipStream_t) = nullptr;
    int (*k_encb_frag_weights)(const float*, const long*, const long*, const long*, int, void*, hipStream_t) = nullptr;
    bool (*head_fused_supported)(const HeadP&) = nullptr;
    int (*head_fwd)(const HeadP&, hipStream_t) = nullptr;
    int (*head_bwd)(const HeadP&, hipStream_t) = nullptr;
};
extern LabTable g_lab;          // engine.hip
}  // namespace gg
