// Kernel-level test entry points (tests/ and tools/ only; declared in include/gemmgan_lab.h; built into libgemmgan_lab.so).
//
// Each hook launches ONE kernel family of the hot path exactly as engine.hip does - same host wrappers, same routing
// (weight-stationary -> token-on-lane Linear, resident -> streaming attention) - on buffers the test supplies, so that
// tests/test_kernels_gpu.py can compare the kernels the bench times against a float64 product of host-rounded
// operands, dropout included (the mask is regenerated on the host from the (seed, site, call) triple: drop_rng.h).
#include "../../include/gemmgan_lab.h"
#include "gg_common.h"
#include "kernels.h"

using namespace gg;

namespace gg {
void tlin_force_route(int route);        // tlin.hip: 0 = as in production, 1 = never the weight-stationary kernels
}

extern "C" {

int gg_test_linear(const gg_test_linear_args* a, int32_t* kernel_class, void* stream) {
    GG_REQUIRE(a && a->X && a->W && a->Y, "null argument");
    TlinP p;
    p.X = a->X; p.ldx = a->ldx; p.M = a->M; p.x_bf16 = a->x_bf16;
    p.W = a->W; p.ldw = a->ldw; p.bias = a->bias;
    p.Y = a->Y; p.ldy = a->ldy; p.y_bf16 = a->y_bf16; p.y_rows = a->y_rows;
    p.N = a->N; p.K = a->K;
    p.film_g = a->film_g; p.film_b = a->film_b; p.film_ld = a->film_ld; p.film_group = a->film_group;
    p.y_row_group = a->y_row_group;
    p.act_relu = a->act_relu;
    p.drop = make_drop_key(a->drop_p, a->drop_seed, a->drop_site, a->drop_call); p.drop_ld = a->drop_ld;
    p.mask_ref = a->mask_ref; p.ldref = a->ldref; p.mask_scale = a->mask_scale; p.mask_bf16 = a->mask_bf16;
    p.accumulate = a->accumulate;
    p.res = a->res; p.ldres = a->ldres; p.res_rows = a->res_rows > 0 ? a->res_rows : 1;
    p.ln_g = a->ln_g; p.ln_b = a->ln_b; p.ln_y = a->ln_y; p.ln_stats = a->ln_stats;
    p.res_bf16 = a->res_bf16; p.ln_y_bf16 = a->ln_y_bf16;
    p.lnb_dres = a->lnb_dres; p.lnb_dgamma = a->lnb_dgamma; p.lnb_dbeta = a->lnb_dbeta; p.lnb_dbias = a->lnb_dbias;
    if (a->route == 2 || a->route == 3) {       // split-operand Linear: fp32 X, fp32 W, fp32 Y; 2 / 3 operand parts (3 / 6 products)
        GG_REQUIRE(a->w_parts && a->ldw == a->K, "gg_test_linear: routes 2 / 3 need the part scratch and a dense W");
        GG_TRY(k_split_weights(reinterpret_cast<const float*>(a->W), a->w_parts, (long)a->N * a->K, (long)a->N * a->K, a->route, (hipStream_t)stream));
        p.W = a->w_parts; p.w_part_stride = (long)a->N * a->K;
        GG_REQUIRE(tlin3_supported(p), "gg_test_linear: no bf16x3 instantiation for this call");
        if (kernel_class) *kernel_class = 62 + a->route;
        return tlin3(p, (hipStream_t)stream, a->route);
    }
    GG_REQUIRE(tlin_supported(p), "gg_test_linear: the Linear kernels do not take this shape / alignment");
    tlin_force_route(a->route);
    if (kernel_class) *kernel_class = tlin_kernel_class(p);
    const int rc = tlin(p, (hipStream_t)stream);
    tlin_force_route(0);
    return rc;
}

const char* gg_test_attn_kernel_name(int which, int S, int E, int nh) { return flash_attn_kernel_name(which, S, E, nh); }

int gg_test_attn_fwd(const void* qkv, const uint8_t* mask, int mask_B, void* ctx, float* lse2, int64_t N, int S, int E, int nh,
                     float drop_p, uint64_t drop_seed, uint32_t drop_site, uint32_t drop_call, int io_bf16, int64_t qkv_B,
                     void* stream) {
    GG_REQUIRE(qkv && ctx && lse2, "null argument");
    if (io_bf16 >= 2) {         // split-operand kernels: fp32 tensors, io_bf16 = number of operand parts (2 / 3)
        GG_REQUIRE(flash_attn_x3_supported(S, E, nh), "gg_test_attn_fwd: unsupported shape");
        return flash_attn_fwd_x3(reinterpret_cast<const float*>(qkv), mask, mask_B, reinterpret_cast<float*>(ctx), lse2, N, S, E, nh,
                                 make_drop_key(drop_p, drop_seed, drop_site, drop_call), (hipStream_t)stream, qkv_B, io_bf16);
    }
    GG_REQUIRE(flash_attn_supported(S, E, nh), "gg_test_attn_fwd: unsupported shape");
    return flash_attn_fwd(qkv, mask, mask_B, ctx, lse2, N, S, E, nh, make_drop_key(drop_p, drop_seed, drop_site, drop_call), io_bf16,
                          (hipStream_t)stream, qkv_B);
}

int gg_test_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse2, float* delta, const uint8_t* mask,
                     int mask_B, void* dqkv, int64_t N, int S, int E, int nh, float drop_p, uint64_t drop_seed, uint32_t drop_site,
                     uint32_t drop_call, int io_bf16, int64_t qkv_B, void* stream) {
    GG_REQUIRE(qkv && ctx && dctx && lse2 && delta && dqkv, "null argument");
    if (io_bf16 >= 2) {
        GG_REQUIRE(flash_attn_x3_supported(S, E, nh), "gg_test_attn_bwd: unsupported shape");
        return flash_attn_bwd_x3(reinterpret_cast<const float*>(qkv), reinterpret_cast<const float*>(ctx), reinterpret_cast<const float*>(dctx),
                                 lse2, delta, mask, mask_B, reinterpret_cast<float*>(dqkv), N, S, E, nh,
                                 make_drop_key(drop_p, drop_seed, drop_site, drop_call), (hipStream_t)stream, qkv_B, io_bf16);
    }
    GG_REQUIRE(flash_attn_supported(S, E, nh), "gg_test_attn_bwd: unsupported shape");
    return flash_attn_bwd(qkv, ctx, dctx, lse2, delta, mask, mask_B, dqkv, N, S, E, nh,
                          make_drop_key(drop_p, drop_seed, drop_site, drop_call), io_bf16, (hipStream_t)stream, qkv_B, nullptr);
}

int gg_test_ffn_fused(const float* X, int64_t M, const void* W1, const float* b1, const void* W2T, const float* b2, void* Hs, float* R2,
                      int64_t keep_rows, const float* ln_g, const float* ln_b, float* Y, float* stats, float drop_p, uint64_t drop_seed,
                      uint32_t site1, uint32_t site2, uint32_t drop_call, void* stream) {
    FfnP f;
    f.X = X; f.M = M; f.E = 256; f.F = 512; f.W1 = W1; f.b1 = b1; f.W2T = W2T; f.b2 = b2; f.Hs = Hs; f.R2 = R2; f.keep_rows = keep_rows;
    f.ln_g = ln_g; f.ln_b = ln_b; f.Y = Y; f.stats = stats;
    f.drop1 = make_drop_key(drop_p, drop_seed, site1, drop_call);
    f.drop2 = make_drop_key(drop_p, drop_seed, site2, drop_call);
    GG_REQUIRE(ffn_fused_supported(f), "gg_test_ffn_fused: unsupported operands");
    return ffn_fused(f, (hipStream_t)stream);
}

// fused feed-forward block of round 4 (enc.hip): the fragment-ordered weight image is built from the fp32 weights into `wfrag`
// (gg_test_ffn2_frag_bytes() bytes) exactly as refresh_shadows does, then the kernel runs as in cond_forward
int64_t gg_test_ffn2_frag_bytes(void) { return (int64_t)enc_frag_bytes(1); }
int gg_test_set_enc_grid(int workgroups) { enc_set_grid(workgroups); return 0; }
int gg_test_ffn2(const void* X, int64_t M, const float* W1, const float* b1, const float* W2, const float* b2, void* Hs, void* R2, int r2_bf16,
                 int64_t keep_rows, const float* ln_g, const float* ln_b, void* Y, int y_bf16, float* stats, float drop_p, uint64_t drop_seed,
                 uint32_t site1, uint32_t site2, uint32_t drop_call, void* wfrag, int variant, void* stream) {
    GG_REQUIRE(X && W1 && W2 && wfrag, "null argument");
    unsigned* stamps = nullptr;
    if (variant >= 64) { stamps = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(wfrag) + enc_frag_bytes(1)); variant -= 64; }   // probe: + 256 * 8 * 8 words behind the image
    const long o1 = 0, o2 = W2 - W1;
    GG_TRY(k_enc_frag_weights(W1, &o1, &o2, 1, wfrag, (hipStream_t)stream));
    Ffn2P f;
    f.X = X; f.M = M; f.Wf = wfrag; f.b1 = b1; f.b2 = b2; f.ln_g = ln_g; f.ln_b = ln_b; f.Hs = Hs; f.R2 = R2; f.r2_bf16 = r2_bf16;
    f.stats = stats; f.Y = Y; f.y_bf16 = y_bf16; f.keep_rows = keep_rows;
    f.drop1 = make_drop_key(drop_p, drop_seed, site1, drop_call);
    f.drop2 = make_drop_key(drop_p, drop_seed, site2, drop_call);
    f.stamps = stamps;
    GG_REQUIRE(ffn2_supported(f), "gg_test_ffn2: unsupported operands");
    return ffn2(f, (hipStream_t)stream, variant);
}

// fused backward of the token-local chain of an encoder layer behind LayerNorm2's backward (enc.hip encb_kernel): Wcat = linear1.weight [512,256] |
// linear2.weight [256,512] | out_proj.weight [256,256] (fp32, contiguous); the backward fragment stream is built into wfrag
// (gg_test_enc_bwd_frag_bytes() bytes); colsums [3][256] += dgamma1, dbeta1, dbias1 (out_proj.bias)
int64_t gg_test_enc_bwd_frag_bytes(void) { return (int64_t)encb_frag_bytes(1); }
int gg_test_enc_bwd(float* dx, int64_t M, const float* Wcat, const void* dres2, const void* h, const void* r1, const float* st1, const float* g1,
                    void* dh, void* dres1, void* dctx, float* colsums, float drop_p, uint64_t drop_seed, uint32_t site1, uint32_t drop_call,
                    void* wfrag, void* stream) {
    GG_REQUIRE(dx && Wcat && wfrag && colsums, "null argument");
    const long o1 = 0, o2 = 512L * 256, oo = 2L * 512 * 256;
    GG_TRY(k_encb_frag_weights(Wcat, &o1, &o2, &oo, 1, wfrag, (hipStream_t)stream));
    EncBwdP p;
    p.dx = dx; p.M = M; p.Wf = wfrag; p.dres2 = dres2; p.h = h; p.r1 = r1; p.st1 = st1; p.g1 = g1;
    p.dh = dh; p.dres1 = dres1; p.dctx = dctx;
    p.dg1 = colsums; p.db1 = colsums + 256; p.dbias1 = colsums + 512;
    p.drop1 = make_drop_key(drop_p, drop_seed, site1, drop_call & 0x3fffffffu);
    p.gate_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    if (drop_call >= 0x40000000u) p.stamps = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(wfrag) + encb_frag_bytes(1));     // probe: + 256 * 8 * 8 words behind the image
    GG_REQUIRE(enc_bwd_supported(p), "gg_test_enc_bwd: unsupported operands");
    return enc_bwd(p, (hipStream_t)stream);
}

int gg_test_head_fwd(int64_t rows, int H, int E, float slope, const float* W1c, int64_t ldw1, const float* b1, const float* W2,
                     const float* b2, const float* w3, const float* b3, const float* cvec, float* a1, float* a2, float* out,
                     int64_t out_rows, void* stream) {
    HeadP h;
    h.rows = rows; h.H = H; h.E = E; h.slope = slope; h.W1c = W1c; h.ldw1 = ldw1; h.b1 = b1; h.W2 = W2; h.b2 = b2; h.w3 = w3; h.b3 = b3;
    h.cvec = cvec; h.a1 = a1; h.a2 = a2; h.out = out; h.ldo = 1; h.out_rows = out_rows;
    GG_REQUIRE(head_fused_supported(h), "gg_test_head_fwd: unsupported operands");
    return head_fwd(h, (hipStream_t)stream);
}
int gg_test_head_bwd(int64_t rows, int H, int E, float slope, const float* W1c, int64_t ldw1, const float* W2, const float* w3,
                     const float* a1, const float* a2, const float* dout, float* dh2, float* dh1, float* dcond, void* stream) {
    HeadP h;
    h.rows = rows; h.H = H; h.E = E; h.slope = slope; h.W1c = W1c; h.ldw1 = ldw1; h.W2 = W2; h.w3 = w3;
    h.a1 = const_cast<float*>(a1); h.a2 = const_cast<float*>(a2); h.dout = dout; h.dh2 = dh2; h.dh1 = dh1; h.dcond = dcond;
    GG_REQUIRE(head_fused_supported(h), "gg_test_head_bwd: unsupported operands");
    return head_bwd(h, (hipStream_t)stream);
}

int gg_test_wgrad(const void* dY, int64_t ldy, int dy_bf16, const void* X, int64_t ldx, int x_bf16, float* dW, int64_t ldw, int64_t M,
                  int N, int K, const float* film_g, const float* film_b, int64_t film_ld, int film_group, const float* fgrad_W,
                  int64_t fgrad_ldw, float* dgamma, float* dbeta, int64_t fgrad_ld, int fgrad_tokens, float* dbias, int64_t x_mod,
                  int x3, void* stream) {
    GG_REQUIRE(dY && X, "null argument");
    GG_REQUIRE(wgrad_supported(dY, ldy, dy_bf16, X, ldx, x_bf16, M, N, K), "gg_test_wgrad: the token-reduction kernel does not take this shape");
    WgradFilm film;
    film.g = film_g; film.b = film_b; film.ld = film_ld; film.group = film_group;
    WgradFilmGrad fg;
    fg.W = fgrad_W; fg.ldw = fgrad_ldw; fg.dgamma = dgamma; fg.dbeta = dbeta; fg.ld = fgrad_ld; fg.tokens = fgrad_tokens;
    return wgrad(dY, ldy, dy_bf16, X, ldx, x_bf16, dW, ldw, M, N, K, (hipStream_t)stream, film_g ? &film : nullptr,
                 fgrad_W ? &fg : nullptr, dbias, x_mod, x3);
}

int gg_test_sqx_fwd(const float* qt, const float* x, const uint8_t* mask, int mask_B, float* probs, float* xbar, int N, int S, int E,
                    int nh, void* stream) {
    GG_REQUIRE(qt && x && probs && xbar, "null argument");
    GG_REQUIRE(sqx_stream_supported(S, E, nh), "gg_test_sqx_fwd: unsupported shape");
    return sqx_stream_fwd(qt, x, mask, mask_B, probs, xbar, N, S, E, nh, (hipStream_t)stream);
}

int gg_test_sqx_bwd(const float* dxbar, const float* qt, const float* xbar, const float* x, const float* probs, float* dx, float* dqt,
                    int N, int S, int E, int nh, void* stream) {
    GG_REQUIRE(dxbar && qt && xbar && x && probs && dx && dqt, "null argument");
    GG_REQUIRE(sqx_stream_supported(S, E, nh), "gg_test_sqx_bwd: unsupported shape");
    return sqx_stream_bwd(dxbar, qt, xbar, x, probs, dx, dqt, N, S, E, nh, (hipStream_t)stream);
}

int gg_test_ln_bwd(const float* dy, const float* r, const float* stats, const float* g, float* dr, void* dres_out, float* dgamma,
                   float* dbeta, float* dbias, int64_t rows, int E, float drop_p, uint64_t drop_seed, uint32_t drop_site,
                   uint32_t drop_call, int dres_bf16, void* stream) {
    GG_REQUIRE(dy && r && stats && g && dr && dgamma && dbeta, "null argument");
    return k_layernorm_bwd(dy, r, stats, g, dr, dres_out, dgamma, dbeta, dbias, rows, E,
                           make_drop_key(drop_p, drop_seed, drop_site, drop_call), (hipStream_t)stream, dres_bf16);
}

int gg_test_gemm(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                 int layA, int layB, int splitk, float alpha, const float* bias, int act, float slope, int accumulate,
                 void* stream) {
    GemmP p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.layA = layA; p.layB = layB; p.splitk = splitk; p.alpha = alpha; p.bias = bias; p.act = act; p.slope = slope;
    p.accumulate = accumulate;
    return gemm_f32(p, (hipStream_t)stream);
}


int gg_test_gemm_small(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                       int layA, int layB, int splitk, float alpha, const float* bias, int act, float slope, int accumulate,
                       void* stream) {
    GemmP p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.layA = layA; p.layB = layB; p.splitk = splitk; p.alpha = alpha; p.bias = bias; p.act = act; p.slope = slope;
    p.accumulate = accumulate;
    return gemm_small(p, (hipStream_t)stream);
}

/* K-strided operands stored as bf16 (the weight-gradient products of bf16-stored branch gradients / activations) */
int gg_test_gemm_bf16_stored(const void* A, const void* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                             int a_bf16, int b_bf16, int splitk, void* stream) {
    GemmP p;
    p.A = reinterpret_cast<const float*>(A); p.B = reinterpret_cast<const float*>(B); p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.layA = LAY_KS; p.layB = LAY_KS; p.a_bf16 = a_bf16; p.b_bf16 = b_bf16; p.splitk = splitk;
    return gemm_bf16(p, (hipStream_t)stream);
}

int gg_test_gemm_bf16(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                      int layA, int layB, int splitk, float alpha, const float* bias, int act, float slope, int accumulate,
                      void* stream) {
    GemmP p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.layA = layA; p.layB = layB; p.splitk = splitk; p.alpha = alpha; p.bias = bias; p.act = act; p.slope = slope;
    p.accumulate = accumulate;
    return gemm_bf16(p, (hipStream_t)stream);
}

}  // extern "C"
