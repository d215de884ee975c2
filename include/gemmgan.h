/* gemmgan.h - C ABI of the MI355X-native WGAN-GP engine (libgemmgan.so).
 *
 * Drop-in boundary for the hot path of francescapia/-GeMM-GAN,
 * /root/reference/src/conditional_gan_cross_attention_with_film.py (cited as R: below).
 * The reference has no FFI of its own: its boundary is the Python object API
 * (generator.forward R:128, discriminator.forward R:197, WGAN_GP.train_disc R:376,
 * WGAN_GP.train_gen R:425, WGAN_GP.train R:463, WGAN_GP.generate_samples R:601).  Each entry
 * point below names the reference function it replaces; gemm_gan_amd/model.py binds them with
 * ctypes behind the reference's own class and method names.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, a hipStream_t passed as void*;
 *   - every function returns 0 on success, non-zero on failure; gg_last_error() gives the text;
 *     nothing aborts the process;
 *   - all float tensors are fp32, row-major, contiguous, batch first; masks are bytes
 *     (non-zero = padded / ignored), exactly the reference's torch.bool convention (R:143);
 *   - the caller owns every buffer (parameters, gradients, optimiser state, workspace are
 *     allocated by the host framework and bound once); inputs are borrowed and never written.
 *   - one host thread drives one engine; kernels are enqueued on the given stream and the
 *     functions return without synchronising.
 */
#ifndef GEMMGAN_H
#define GEMMGAN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GG_ROLE_GENERATOR 0
#define GG_ROLE_CRITIC 1

/* arithmetic of the GEMM-shaped products (everything else is fp32 in both modes) */
#define GG_PREC_F32 0  /* fp32-input MFMA: exact fp32, the parity mode (<= 1e-3 vs the reference) */
#define GG_PREC_BF16 1 /* bf16 MFMA operands, fp32 accumulate: the throughput mode               */
#define GG_PREC_BF16X3 3 /* split-operand bf16: every fp32 operand as hi + lo bf16 halves, three bf16 MFMAs per product tile, fp32  */
                         /* accumulate and fp32 storage everywhere - fp32-grade results (<= 1e-3 vs the reference, the parity */
                         /* gates run in it) through the token-on-lane Linear, fused attention and weight-gradient kernels     */
#define GG_PREC_FP8 2  /* GG_PREC_BF16 with OCP e4m3 operands (block-scaled v_mfma_scale_f32_*_f8f6f4, fp32 accumulate,    */
                       /* per-tensor power-of-two scales) in the forward Linears of the encoder layers: BASELINE configs[4] */

#define GG_OPT_RMSPROP 0 /* torch.optim.RMSprop(lr)                       R:324 */
#define GG_OPT_ADAM 1    /* torch.optim.Adam(lr, betas=(.9,.99))          R:327 */
#define GG_OPT_ADAMW 2   /* torch.optim.AdamW(lr, betas=(.9,.99), wd=.01) R:330 */

/* loss slots written by the iteration entry points (device float[GG_N_LOSSES]) */
#define GG_LOSS_D_REAL 0 /* mean(-D(x))        R:43  */
#define GG_LOSS_D_FAKE 1 /* mean(D(G(z)))      R:44  */
#define GG_LOSS_GP 2     /* mean((|grad|-1)^2) R:374 */
#define GG_LOSS_G 3      /* mean(-D(G(z)))     R:36  */
#define GG_N_LOSSES 8

/* model variants: the same engine, parametrised */
#define GG_VARIANT_XATTN_FILM 0 /* src/conditional_gan_cross_attention_with_film.py: FiLM + encoder + T2I/I2T cross attention  */
#define GG_VARIANT_FILM 1       /* src/conditional_gan_film.py:97-205: FiLM + bias-free encoder (R':115 bias=False), the   */
                                /* conditioning vector is the encoder's CLS row (R':150); text is one vector per sample (T=1) */

#define GG_VARIANT_IMG 2        /* src/conditional_gan_img_transformer.py:95-190: no FiLM, patch encoder Linear -> ReLU ->  */
                                /* LayerNorm (R'':106-110), bias-free encoder, CLS-row conditioning; text is unused         */
#define GG_VARIANT_VANILLA 3    /* src/vanilla_gan_unconditional.py:93-184: MLP generator / critic without conditioning.    */
                                /* The first-layer weights keep embedding_dims zero columns: gg_param_info reports            */
                                /* [H, V + embedding_dims], the reference's tensor is the [:, :V] block; cond inputs ignored  */

typedef struct gg_config {
    /* model shape: WGAN_GP.__init__ kwargs R:258-271, generator/discriminator ctors R:99, R:169 */
    int32_t n_genes;        /* input_dims / vector_dims (G)                           */
    int32_t latent_dims;    /* L                                                      */
    int32_t embedding_dims; /* E (d_model); FFN width is 2E (R:115)                   */
    int32_t hidden_dims;    /* H: generator_dims=[H,H,G], discriminator_dims=[H,H,1]  */
    int32_t text_dims;      /* text_embedding_dims (Dt)                               */
    int32_t patch_dims;     /* patches_embedding_dims (Dp)                            */
    int32_t n_heads;        /* 4 (R:115, R:121)                                       */
    int32_t n_layers;       /* 2 (R:119)                                              */
    float negative_slope;   /* LeakyReLU slope of the MLP blocks (R:56)               */
    float dropout;          /* encoder-layer dropout in train mode (R:116: 0.1)       */
    /* trainer */
    float lr_d, lr_g;       /* R:265 */
    int32_t optimizer;      /* GG_OPT_* (R:320-331) */
    float gp_weight;        /* R:266 */
    float clip_d, clip_g;   /* clip_grad_norm_ max_norm, <=0 disables (R:414: 10, R:457: 2) */
    /* capacity the workspace is sized for */
    int32_t max_batch;      /* B */
    int32_t max_patches;    /* P (encoder sequence is P+1 with the CLS token, R:142) */
    int32_t max_text_tokens;/* T */
    uint64_t seed;          /* dropout stream seed (z / alpha are supplied by the caller) */
    int32_t precision;      /* GG_PREC_* */
    int32_t variant;        /* GG_VARIANT_*: which reference model file the networks follow */
} gg_config;

typedef struct gg_engine gg_engine;

/* conditioning inputs of one minibatch (R:463 arguments after gene_expression) */
typedef struct gg_cond {
    const float* patches;     /* [B,P,Dp] */
    const uint8_t* patch_pad; /* [B,P]  non-zero = padded */
    const float* text;        /* [B,T,Dt], token 0 = text CLS */
    const uint8_t* text_pad;  /* [B,T] */
    int32_t B, P, T;
} gg_cond;

const char* gg_last_error(void);
const char* gg_version(void);

/* ---- construction: WGAN_GP.__init__ + build_WGAN_GP (R:258, R:334) ------------------------- */
int gg_create(const gg_config* cfg, gg_engine** out);
void gg_destroy(gg_engine* e);

/* ---- parameter layout: the live state_dict entries of each network ------- */
int gg_param_count(const gg_engine* e, int role);
const char* gg_param_name(const gg_engine* e, int role, int index);
/* offset (in floats) into the flat parameter buffer, element count, and shape (ndim <= 3) */
int gg_param_info(const gg_engine* e, int role, int index, int64_t* offset, int64_t* numel,
                  int32_t* ndim, int32_t shape[3]);
int64_t gg_flat_numel(const gg_engine* e, int role); /* floats in the flat buffer (16-B aligned slots) */

/* ---- binding of caller-owned device memory ------------------------------------------------------- */
/* params / grads / state1 / state2: flat fp32 buffers of gg_flat_numel floats.  state = optimiser
 * state (RMSprop square_avg | Adam exp_avg, exp_avg_sq); must be zero-initialised by the caller. */
int gg_bind_net(gg_engine* e, int role, float* params, float* grads, float* state1, float* state2);
size_t gg_workspace_bytes(const gg_engine* e);
int gg_bind_workspace(gg_engine* e, void* workspace, size_t bytes);

/* ---- forward passes: generator.forward (R:128) / discriminator.forward (R:197) ---------------- */
/* v = z [B,L] (generator) or gene_expression [B,G] (critic); out = [B,G] or [B,1].
 * train != 0 applies dropout exactly where nn.TransformerEncoderLayer does in train() mode. */
int gg_forward(gg_engine* e, int role, const float* v, const gg_cond* c, float* out, int train, void* stream);

/* ---- one critic update, split at the gradient all-reduce: WGAN_GP.train_disc (R:376-423) --------
 * backward: G(z) (frozen), D(fake), D(real), D_loss, gradient penalty incl. double backward
 * (R:351-374), all critic gradients -> bound grads buffer (overwritten).  losses: device float[8].
 * apply:    clip_grad_norm_(10) + optimiser step (R:414-415).  grad_scale multiplies every gradient
 *           first (1/world_size after a SUM all-reduce of the flat gradient buffer). */
int gg_critic_backward(gg_engine* e, const float* x_real, const float* z, const float* alpha,
                       const gg_cond* c, float* losses, void* stream);
int gg_critic_apply(gg_engine* e, float grad_scale, void* stream);

/* The same iteration in two phases, for data-parallel hosts that overlap the gradient all-reduce with the backward pass:
 * ..._head runs everything up to and including the MLP-head / gradient-penalty gradients (R:391-409 and the head part of
 * loss.backward(), R:412) and returns with the slots gg_mlp_grad_range() names complete on the caller's stream;
 * ..._cond runs the conditioning stack's backward (the rest of R:412).  head + cond == gg_critic_backward. */
int gg_critic_backward_head(gg_engine* e, const float* x_real, const float* z, const float* alpha,
                            const gg_cond* c, float* losses, void* stream);
int gg_critic_backward_cond(gg_engine* e, const gg_cond* c, void* stream);
/* [offset, offset + numel) of the flat gradient buffer: the MLP-head parameters (`<mlp>.0.0`, `<mlp>.1.0`, final_layer),
 * the last entries of the state_dict order; [0, offset) are the conditioning-stack parameters */
int gg_mlp_grad_range(const gg_engine* e, int role, int64_t* offset, int64_t* numel);
/* The conditioning phase in stages, so that a data-parallel host can all-reduce the gradient range of a finished stage while the next one
 * runs (only the last range is exposed): stage 0 = the cross-attention blocks, 1 .. n_layers = encoder layers from the last to the first,
 * n_layers + 1 = replica fold, CLS token, patch encoder, FiLM, text encoder.  gg_cond_stage_count() = n_layers + 2 (0: unconditional model);
 * gg_cond_stage_range() = the contiguous slice of the flat gradient buffer stage `stage` completes (backward order = reverse flat order).
 * The stages of one iteration are called in order 0 .. count-1 after gg_*_backward_head; gg_*_backward_cond runs them all.  A stage's
 * weight-gradient leaves run on the engine's side stream: gg_side_join() makes that stream wait for the caller's, so a collective
 * issued from the side stream sees the whole stage (R:412 loss.backward(); SURVEY 8e "Overlap"). */
int gg_cond_stage_count(const gg_engine* e);
int gg_cond_stage_range(const gg_engine* e, int role, int stage, int64_t* offset, int64_t* numel);
int gg_critic_backward_cond_stage(gg_engine* e, const gg_cond* c, int stage, void* stream);
int gg_generator_backward_cond_stage(gg_engine* e, const gg_cond* c, int stage, void* stream);
int gg_side_join(gg_engine* e, void* stream);
/* The critic's conditioning pass (R:403,404,360: its three dropout draws) of the NEXT gg_critic_backward[_head] call on the
 * same minibatch shape, computed ahead: it depends on the critic's weights and the conditioning inputs only, so a
 * data-parallel host runs it under the generator's gradient all-reduce.  Discarded by anything that changes the critic's
 * weights or reuses its activation arena (gg_critic_apply, gg_forward(critic), gg_generator_backward). */
int gg_critic_cond_prefetch(gg_engine* e, const gg_cond* c, void* stream);

/* ---- WGAN_GP.gradient_penalty (R:351-374) as a call of its own ---------------------------------------------------
 * *gp_out (device float) = mean_b (|grad_x^ D(x^_b)|_2 - 1)^2 for x^ = alpha*real + (1-alpha)*fake, alpha [B] supplied by the
 * caller (R:354 draws it with torch.rand).  train != 0: dropout as in train() mode.  No gradient buffer is written; the
 * gradient itself ([B,G]) is left in the debug buffer "gp_grad". */
int gg_gradient_penalty(gg_engine* e, const float* x_real, const float* x_fake, const float* alpha, const gg_cond* c,
                        int train, float* gp_out, void* stream);

/* ---- one generator update: WGAN_GP.train_gen (R:425-461) ------------------------------------------ */
int gg_generator_backward(gg_engine* e, const float* z, const gg_cond* c, float* losses, void* stream);
int gg_generator_apply(gg_engine* e, float grad_scale, void* stream);
/* two-phase form (see gg_critic_backward_head): head = R:441-455 down to the generator's MLP-head gradients */
int gg_generator_backward_head(gg_engine* e, const float* z, const gg_cond* c, float* losses, void* stream);
int gg_generator_backward_cond(gg_engine* e, const gg_cond* c, void* stream);

/* ---- optional: the frozen generator's outputs for the next n critic iterations, computed ahead -------------
 * WGAN_GP.train (R:463-477) calls train_disc n_critic times on ONE conditioning batch with the generator frozen, so
 * the n generator forward passes (fresh z, fresh dropout draws) do not depend on the critic updates in between.
 * This call runs them as stacked replicas (z_all [n, B, L]); the following n gg_critic_backward calls with the same
 * batch size consume the stored outputs in order instead of running the generator (their z argument is then unused).
 * gg_generator_apply discards what is left.  gg_train_step does this by itself (gg_set_prefetch(e, 0) turns it off);
 * a data-parallel host loop (all-reduce between *_backward and *_apply) calls it once before its critic loop. */
int gg_generator_prefetch(gg_engine* e, const float* z_all, int n, const gg_cond* c, void* stream);
int gg_set_prefetch(gg_engine* e, int on);
/* Parameter gradients (leaves of the backward chain), the frozen critic's forward of the generator iteration and all but the
 * first prefetched generator pass run on engine-owned streams beside the caller's; every entry point returns with the
 * caller's stream waiting for them.  0 serialises everything on the caller's stream (isolated kernel timings). */
int gg_set_side_streams(gg_engine* e, int on);

/* ---- whole step on one GPU: WGAN_GP.train (R:463-477) -----------------------------------------------
 * z_all [n_critic+1, B, L], alpha_all [n_critic, B]; losses hold the LAST critic iteration's
 * D_real / D_fake / GP (what d_batch_loss reports, R:421) and the generator loss. */
int gg_train_step(gg_engine* e, const float* x_real, const gg_cond* c, const float* z_all,
                  const float* alpha_all, int n_critic, float* losses, void* stream);

/* Captured step (hipGraph).  With gg_set_graph(e, 1) gg_train_step runs eagerly the first time it sees a set of argument
 * pointers / shapes / hyper-parameters, captures its ~900 launches (three streams) into a graph the second time and
 * replays the graph from then on: one hipGraphLaunch plus one 1-thread kernel that sets the device words the frozen
 * kernel arguments cannot carry (dropout epoch, Adam step offset).  Up to 8 graphs are kept (least recently used
 * evicted); callers whose buffers move every step simply stay eager.  Results are those of the eager path (same
 * kernels, same order per stream); dropout masks differ between the two routes only in which random stream they draw.
 * gg_graph_stats: graphs captured, replays, captures that failed (those signatures stay eager). */
int gg_set_graph(gg_engine* e, int on);
int gg_graph_stats(const gg_engine* e, int64_t* captures, int64_t* replays, int64_t* failures);

/* ---- knobs mirrored from the reference trainer ------------------------------------------------------ */
int gg_set_lr(gg_engine* e, int role, float lr);     /* optimizer.param_groups[i]['lr'] (R:651-657) */
int gg_set_dropout(gg_engine* e, float p);            /* parity runs use 0 */
int gg_set_seed(gg_engine* e, uint64_t seed);
int gg_set_precision(gg_engine* e, int precision);    /* GG_PREC_* ; may be switched between calls */
int gg_set_flash(gg_engine* e, int on);               /* fused attention kernels in bf16 mode (default on) */
int gg_set_wgrad(gg_engine* e, int on);               /* long-reduction weight-gradient kernel in bf16 mode (default on) */
int gg_set_bstore(gg_engine* e, int on);              /* bf16 storage of MFMA-operand-only tensors in bf16 mode (default on) */
int gg_set_head_fused(gg_engine* e, int on);          /* MLP heads as one forward and one backward launch, bf16 mode (default off: no faster) */
int gg_set_lnb_fused(gg_engine* e, int on);           /* dx1 += and LayerNorm-1 backward in one kernel, bf16 mode at E = 256 (default on) */
int gg_set_xstore(gg_engine* e, int on);              /* bf16 storage of the encoder's LayerNorm outputs and of the pre-LayerNorm sums kept
                                                          for the backward pass, bf16 mode at E = 256 (default 1; 0 off; 3 = outputs only) */
int gg_set_sqx(gg_engine* e, int on);                 /* projection-free single-query T2I attention (default on) */
int gg_set_tlin(gg_engine* e, int on);                /* token-on-lane Linear kernels in bf16 mode (default on) */
int gg_set_encb(gg_engine* e, int on);                 /* fused backward of an encoder layer's token-local chain behind LayerNorm2's backward (csrc/enc.hip): one launch for the gated hidden gradient, dx1 +=, LayerNorm1 backward and the context gradient */
int gg_set_ffn2(gg_engine* e, int mode);               /* streamed fused feed-forward block (csrc/enc.hip): 0 off, 1 on (4-slot weight ring), 3 on (8-slot ring); needs the bf16-stored LayerNorm outputs */
int gg_set_ffn_fused(gg_engine* e, int on);           /* fused feed-forward block (one launch per layer) in bf16 mode at E = 256 (default off: measured slower than the two launches) */
int gg_reset_optimizer_steps(gg_engine* e);           /* after (re)binding zeroed optimiser state */
int gg_get_optimizer_step(const gg_engine* e, int role); /* Adam/AdamW bias-correction step count   */
int gg_set_optimizer_step(gg_engine* e, int role, int step);

/* The kernel-level test hooks (gg_test_*) and the opt-in kernels that lost their A/B against the default routes are in
 * libgemmgan_lab.so: include/gemmgan_lab.h. */
/* device pointer + element count of a named internal activation buffer of the LAST call, e.g.
 * "D.x0", "D.L0.P", "G.c", "X2", "gp_grad" (list in engine.hip); lets tests localise a mismatch. */
int gg_debug_buffer(gg_engine* e, const char* name, void** ptr, int64_t* numel);
int gg_debug_buffer_is_bf16(gg_engine* e, const char* name);   /* 1 if that buffer currently holds bf16 elements */
/* ---- live per-kernel-class timing with HIP events on the launch stream (bench.py roofline) ---------
 * gg_profile_enable(e, 1) brackets every subsequent GEMM launch with an event pair taken from a pool;
 * gg_profile_enable(e, 1 | (mask << 1)) only the launches of the kernel classes whose bit is set in mask (bit i = row i of
 * the aggregate), so that a timed region can carry the events of ONE class without paying for all of them;
 * gg_profile_enable(e, -1) stops taking events and keeps the records (0 stops and the next enable clears them);
 * gg_profile_collect synchronises the events and aggregates per kernel class (= kernel symbol:
 * "gemm_f32<A-layout,B-layout>") launches, total milliseconds, algorithmic FLOPs (2*M*N*K*batch) and
 * algorithmic bytes ((M*K + K*N + M*N)*4*batch).  gg_profile_read returns row i of the aggregate. */
/* phase marks: timing events recorded on the caller's stream at the phase boundaries of gg_train_step (conditioning forward, MLP head +
 * gradient penalty, conditioning backward, optimiser; per critic iteration and for the generator iteration) - the timeline of an
 * UN-profiled step (a tracer slows the host's enqueue enough to open gaps that do not exist otherwise).  bench.py: roofline.step.phases */
int gg_phase_enable(gg_engine* e, int on);
int gg_phase_count(const gg_engine* e);
int gg_phase_read(gg_engine* e, int index, char* name, int name_cap, double* ms);
int gg_profile_enable(gg_engine* e, int on);
int gg_profile_enable_class(gg_engine* e, const char* name);   /* event pairs for ONE class, by its gg_profile_read name */
int gg_profile_add_class(gg_engine* e, const char* name);      /* ... and for this class as well (after gg_profile_enable_class) */
int gg_profile_collect(gg_engine* e);            /* returns number of classes, <0 on error */
int gg_profile_read(gg_engine* e, int index, char* name, int name_cap, int64_t* launches, double* ms,
                    double* flops, double* bytes);
/* gradient-penalty kernels (csrc/gpchain.hip) on the buffers of the last critic iteration with batch B: us[4] / bytes[4] = average
 * microseconds (the dispatches' own timestamps) and algorithmic HBM bytes per launch of gp_front_k, gp_grad_k, gp_coef_k, gp_tail_k */
int gg_gp_profile(gg_engine* e, int B, int reps, double* us, double* bytes, void* stream);
/* kernels launched since the start of the last gg_train_step, or since gg_reset_launch_count (hosts that drive the step
 * through the two-phase entries - the data-parallel loop - reset it themselves at the top of their train()) */
int64_t gg_launch_count(const gg_engine* e);
int gg_reset_launch_count(gg_engine* e);
/* Streams for the engine's concurrent work, owned by the host: `side` carries the parameter-gradient leaves and the frozen
 * critic's forward of the generator iteration, `prefetch` the generator passes computed ahead (R:463-477 allows both: see
 * gg_set_side_streams / gg_generator_prefetch).  Without this call the engine creates its own.  Binding them lets the host
 * framework order the lifetime of borrowed input tensors against them (torch: Tensor.record_stream) - kernels on these
 * streams read the caller's conditioning inputs after an entry point has returned.  Call before the first iteration. */
int gg_bind_streams(gg_engine* e, void* side, void* prefetch);

/* ---- evaluation: nearest reference records ---------------------------------------------
 * The device-side math of the reference's privacy metrics DCR / NNDR (src/privacy_evaluator.py:9-66): for every query row
 * the Euclidean distance to its nearest (d1) and second nearest (d2, +inf if nr == 1) row of `refs`, exact fp32 differences.
 * queries [nq, dim], refs [nr, dim] row-major device pointers; scratch: gg_eval_nn2_scratch(nq, nr) floats.
 * Independent of gg_engine. */
/* PRDC (src/distribution_distances.py:87-142; the reference uses L1 distances there, :64): the k smallest distances of every
 * query row to `refs`, ascending, out [nq, gg_eval_knn_width(k)] (1 <= k <= 16; l1 != 0: sum |q - r|, else Euclidean); and
 * the counting pass over the never-materialised [real x fake] distance matrix: below_real[j] = #{i: d_ij < rad_real[i]},
 * any_fake[i] = any_j d_ij < rad_fake[j], min_d[i] = min_j d_ij.  gg_eval_prdc_counts: l1 bit 0 = L1 distances, bit 1 = inclusive
 * comparisons (d <= radius: ManifoldEstimator.evaluate, src/unsupervised_metrics.py:223, with Euclidean distances). */
long gg_eval_knn_scratch(long nq, long nr, int k);
int gg_eval_knn_width(int k);
int gg_eval_knn(const float* queries, long nq, const float* refs, long nr, int dim, int k, int l1, float* out, float* scratch,
                long scratch_floats, void* stream);
int gg_eval_prdc_counts(const float* real, long nr, const float* fake, long nf, int dim, int l1, const float* rad_real,
                        const float* rad_fake, int* below_real, int* any_fake, float* min_d, void* stream);
long gg_eval_nn2_scratch(long nq, long nr);
int gg_eval_nn2(const float* queries, long nq, const float* refs, long nr, int dim, float* d1, float* d2, float* scratch,
                long scratch_floats, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEMMGAN_H */
