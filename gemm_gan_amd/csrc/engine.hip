// WGAN-GP step engine: parameter layout, workspace arena, forward / hand-written backward of the
// conditioning stack (FiLM -> patch encoder -> CLS -> post-norm encoder layers -> two single-query
// cross attentions), critic / generator MLP heads, closed-form gradient penalty with its double
// backward, global-norm clip + optimiser.  Exposed through the C ABI of include/gemmgan.h.
//
// Reference: /root/reference/src/conditional_gan_cross_attention_with_film.py (R:), restated in
// oracle/numpy_oracle.py whose decomposition this file follows line by line.
#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/gemmgan.h"
#include "gg_common.h"
#include "kernels.h"
#include <cstdlib>

namespace gg {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
LabTable g_lab;        // entry points of libgemmgan_lab.so, when it is loaded (kernels.h)

namespace {
constexpr int MAXL = 8;

struct ParamInfo {
    std::string name;
    long off = 0, numel = 0;
    int ndim = 0;
    int shape[3] = {0, 0, 0};
};

struct AttnP { long inw, inb, ow, ob; };
struct LayerP { AttnP sa; long l1w, l1b, l2w, l2b, n1w, n1b, n2w, n2b; };

struct Net {
    int role = 0;
    std::vector<ParamInfo> ps;
    long total = 0;
    long cls, film_w, film_b, te_w, te_b, pe_w, pe_b, pe_lnw = 0, pe_lnb = 0;
    LayerP layer[MAXL];
    AttnP t2i, i2t;
    long w1, b1, w2, b2, w3, b3;
    int V = 0, OUT = 0;            // first-layer non-conditioning width (L or G), output width (G or 1)
    float *w = nullptr, *g = nullptr, *s1 = nullptr, *s2 = nullptr;
    // bf16 shadow copies of the 2-D conditioning-stack weights (same offsets as the flat fp32 buffer):
    // wb = W [rows][cols], wtb = W^T [cols][rows]; refreshed from the fp32 master at every public entry
    char *wb = nullptr, *wtb = nullptr;
    char *wp3 = nullptr, *wtp3 = nullptr;     // bf16x3 mode: three bf16 parts (hi, mid, lo) of W / W^T, part sp at + 2 * total * sp bytes
    // fp8 mode: e4m3 shadow of the same weights (tensor at byte offset 2 * flat offset), per-tensor exponents
    char* wfragb = nullptr;                   // ... and of linear2^T / linear1^T / out_proj^T for the fused backward kernel
    char* wfrag = nullptr;                    // fragment-ordered bf16 image of linear1 / linear2 of every encoder layer (enc.hip: the fused feed-forward stream)
    char* w8 = nullptr;
    unsigned* w8_amax = nullptr;
    int* w8_exp = nullptr;
    std::map<long, int> tab_index;      // flat offset -> entry of `tab`
    ShadowEntry* tab_dev = nullptr;
    std::vector<ShadowEntry> tab;
    int step_t = 0;
    float lr = 0.f;

    long add(const std::string& name, int d0, int d1 = 0, int d2 = 0) {
        ParamInfo p;
        p.name = name;
        p.ndim = d2 ? 3 : (d1 ? 2 : 1);
        p.shape[0] = d0; p.shape[1] = d1; p.shape[2] = d2;
        p.numel = (long)d0 * (d1 ? d1 : 1) * (d2 ? d2 : 1);
        p.off = total;
        total += (p.numel + 7) / 8 * 8;     // 32-byte aligned slots: 16-byte vector loads on the fp32 weights AND their bf16 shadows
        ps.push_back(p);
        return p.off;
    }
    // slot that exists in the flat buffers but is no parameter of this variant (a bias of a bias-free layer, the text
    // cross-attention of a FiLM-only model): zero, invisible to the state dict, behind `live` so no optimiser touches it
    long add_ghost(int d0, int d1 = 0) {
        const long off = total;
        total += ((long)d0 * (d1 ? d1 : 1) + 7) / 8 * 8;
        return off;
    }
    long live = 0;      // the optimiser, the gradient norm and the state dict cover [0, live)
};

struct LayerActs {
    float *qkv, *P, *ctx, *r1, *x1, *h, *r2, *x2, *st1, *st2, *lse;
};
struct CondActs {
    int B = 0, R = 1, P = 0, T = 0;
    uint32_t call = 0;
    float drop = 0.f;
    bool flash = false;
    bool bst = false;          // qkv / ctx / h (and their gradients) stored as bf16 in this pass
    bool rst = false;          // the pre-LayerNorm sums r1 / r2 (kept for the backward pass) stored as bf16 in this pass
    bool xst = false;          // the LayerNorm outputs x1 (every layer) and x2 (all but the last layer) stored as bf16 in this pass
    float *gbpre, *gb, *tok, *x0, *xrep, *tokrep;
    float *pe_h = nullptr, *pe_y = nullptr, *pe_st = nullptr, *pe_zero = nullptr;     // Linear->ReLU->LayerNorm patch encoder (img variant)
    uint8_t* mask;
    LayerActs L[MAXL];
    float *t2i_q, *t2i_kv, *t2i_P, *t2i_ctx, *t2i_out, *t2i_xbar, *t2i_qt;
    bool sqx = false;          // projection-free single-query T2I attention, generic kernel
    bool share0 = false;       // dropout replicas share the layer-0 input x0 and its QKV projection (no replicated copy)
    bool i2t_shared = false;   // I2T keys / values (projected text tokens) exist once for all replicas (many text tokens)
    bool i2t_t1 = false;       // one text token: the I2T attention is its value projection (see cond_forward)
    bool sqx2 = false;         // ... streaming kernels with the per-head projections hoisted into batched GEMMs
    float *i2t_q, *i2t_kv, *i2t_P, *i2t_ctx, *i2t_out;
    float* c;     // [R*B, E]
};
// generator outputs that may be computed ahead of the critic iterations that consume them (gg_generator_prefetch)
constexpr int GG_MAX_PREFETCH = 8;

struct HeadActs {
    float *a1, *a2, *out;   // [rows,H], [rows,H], [rows,OUT]
};

struct Arena {
    char* base = nullptr;
    size_t off = 0;
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};
}  // namespace
}  // namespace gg

using namespace gg;

struct gg_engine {
    gg_config cfg;
    int E, F, H, G, L, Dt, Dp, nh, nl, dh;
    int maxB, maxP, maxT, maxS, maxR;
    int preR = 5;              // generator passes one batched prefetch pass holds (prefetch arena / head scratch): GG_PREFETCH_R, 3 .. GG_MAX_PREFETCH
    Net net[2];
    float dropout = 0.f;
    int precision = GG_PREC_F32;
    bool x3 = false;           // GG_PREC_BF16X3: the bf16 mode's kernels on hi / lo splits of fp32 operands, fp32 storage (parity mode)
    bool fp8_fwd = false;      // GG_PREC_FP8: bf16 mode with e4m3 operands in the encoder layers' forward Linears
    bool xattn = true;         // text<->image cross attention (conditional_gan_cross_attention_with_film.py); false: CLS row (conditional_gan_film.py)
    bool enc_bias = true;      // encoder layers with biases (bias=False in conditional_gan_film.py:115)
    bool no_cond = false;      // unconditional model (vanilla_gan_unconditional.py): the conditioning vector is identically zero
    bool film = true;          // FiLM modulation of the patches from the text vector; false: conditional_gan_img_transformer.py
    bool pe_ln = false;        // patch encoder = Linear -> ReLU -> LayerNorm (conditional_gan_img_transformer.py:106-110)
    uint64_t seed = 0;
    uint32_t call_counter = 0;
    int64_t launches = 0;
    // workspace
    void* ws = nullptr;
    size_t ws_bytes = 0;
    CondActs actsG, actsD, actsP;      // actsP: generator passes of the pipelined prefetch
    HeadActs headG, headD, headP;
    float* c3P = nullptr;
    float *X2;                 // [2B, G]: fake rows then real rows
    float *Xpre = nullptr;     // [GG_MAX_PREFETCH, B, G]: generator outputs of a whole train() computed in batched passes
    int pre_n = 0, pre_next = 0, pre_B = 0;
    // critic conditioning pass computed ahead (gg_critic_cond_prefetch): valid until the critic's weights or actsD change
    bool dcond_valid = false;
    int dcond_B = 0, dcond_P = 0, dcond_T = 0, dcond_R = 0;
    int crit_R = 0;            // replicas of the critic iteration in flight (between its head and conditioning phases)
    // pipelined prefetch: all but the first output are computed on a third stream in their own activation arena, beside
    // the critic iterations that do not need them yet; pre_ev[k] marks output k ready (pre_wait[k]: not yet waited for)
    hipStream_t pre_stream = nullptr;
    hipEvent_t pre_fork = nullptr, pre_ev[GG_MAX_PREFETCH] = {};
    bool pre_wait[GG_MAX_PREFETCH] = {};
    struct { const float* z_all; const gg_cond* in; int n, rmax; bool pending; } pre_rest = {nullptr, nullptr, 0, 0, false};
    float *c3;                 // [3B, E] conditioning rows for the critic head (fake, real, hat)
    float *Pfr;                // [2B, H] x @ W1x^T for fake / real
    float *dseed;              // [2B]
    float *dA, *dB;            // [3B, H] scratch
    float *dc;                 // [3B, E]
    float *gp_g2, *gp_g1, *gp_g1s, *gp_grad, *gp_nrm2, *gp_coef, *gp_dg1, *gp_dg2;
    float* gp_nrm2p = nullptr; // [gp_grad3_parts(G)][B] per-strip partial row norms of the split-operand gradient kernel
    float *dxfake;             // [B, G]
    float *sumsq;              // [4]
    // conditioning backward scratch
    float *sPd, *sdP, *sdqkv, *sdx, *sdr, *sdres, *sdh, *sdctx;
    float *s_delta;
    int flash = 1;             // use the fused attention kernels when precision == bf16 and the shape allows
    int small_on = 1;          // latency-optimised kernel for few-tile GEMMs (bf16 mode)
    int wgrad_on = 1;          // dedicated long-reduction weight-gradient kernel (bf16 mode)
    int bstore_on = 1;         // store MFMA-operand-only tensors in bf16 (bf16 mode, flash + tlin paths)
    // weight gradients are leaves of the backward chain: they run on a second stream beside the data-gradient kernels
    hipStream_t side = nullptr;
    bool side_own = true, pre_own = true;      // false: the stream was bound by the host (gg_bind_streams) and outlives the engine
    hipEvent_t ev_ready = nullptr, ev_done[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool side_pending[5] = {false, false, false, false, false};
    int side_on = 1;
    int prefetch_on = 1;       // gg_train_step computes the generator outputs of all critic iterations in batched passes
    int sqx_on = 1;            // single-query T2I attention without K/V projections (any precision)
    float *s_dqt, *s_dxbar;
    int tlin_on = 1;           // use the token-on-lane Linear kernels when precision == bf16 and the shape allows
    int encb_on = getenv("GG_ENCB") ? atoi(getenv("GG_ENCB")) : 0;     // fused backward of the token-local chain behind LayerNorm2's backward (enc.hip encb_kernel)
    // 1 + variant: the streamed fused feed-forward block (enc.hip) in bf16 mode with bf16-stored LayerNorm outputs.  Opt-in: it takes the
    // whole passes of its persistent grid and leaves the R * B rows past them to the two Linear launches, so WHICH kernel computes a row
    // depends on the row's position in the batch - the two agree to bf16 rounding ties only (6e-3 on activations), and the full-size
    // row-independence / shard-identity properties of tests/test_engine_oracle_gpu.py then hold to 7e-3 instead of 1e-3.  Not worth
    // 0.19 ms of a 24.9 ms step (profiles/r04_notes.md).
    int ffn2_on = getenv("GG_FFN2") ? atoi(getenv("GG_FFN2")) : 0;
    int ffn_on = getenv("GG_FFN_FUSED") != nullptr;   // fused feed-forward block (ffn.hip), bf16 mode, E = 256: opt-in (or gg_set_ffn_fused) -
                               // measured 20 % slower than the two launches it replaces (profiles/r03_ffn_fused.md)
    float *s_dt, *s_dp, *s_dq, *s_dq2, *s_dkv, *s_dkv2, *s_dtokrep, *s_dtok, *s_dtok0, *s_dx0, *s_demb, *s_mod, *s_dmod, *s_dgb, *s_tmpE;
    hipStream_t st = nullptr;
    // ---- captured train step (hipGraph) ----
    // device words: [0] dropout epoch (DropKey::epoch), [1 + role] offset added to the Adam step number baked into a
    // captured optimiser launch; written by one tiny eager kernel ahead of every replay
    uint32_t* dev_words = nullptr;
    struct StepGraph {
        std::vector<uint64_t> sig;      // everything a captured step bakes in: pointers, shapes, hyper-parameters
        hipGraphExec_t exec = nullptr;
        int seen = 0;                   // eager runs with this signature so far (capture happens on the second sight)
        bool bad = false;               // capture failed once: stay eager
        uint32_t calls = 0;             // forward passes (dropout call numbers) one step consumes
        int t0[2] = {0, 0}, tsteps[2] = {0, 0};   // optimiser step numbers at capture, steps per replay
        int64_t launches = 0;
        uint64_t last_use = 0;
    };
    std::vector<StepGraph> graphs;
    bool graph_on = false, capturing = false;
    hipStream_t cap_stream = nullptr;
    uint32_t epoch_host = 0;
    uint64_t graph_clock = 0;
    int64_t graph_captures = 0, graph_replays = 0, graph_failures = 0;
    // phase marks (gg_phase_enable): timing events on the caller's stream at the phase boundaries of a train step - the un-profiled timeline
    bool phase_on = false;
    std::vector<std::pair<const char*, hipEvent_t>> phase_marks;
    std::vector<hipEvent_t> phase_pool;
    // live profiling
    bool prof_on = false;
    unsigned prof_mask = 0xffffffffu;      // kernel classes that get event pairs (bit = class id)
    struct ProfRec { int cls; double flops, bytes; hipEvent_t e0, e1; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    size_t prof_next = 0;
    struct ProfAgg { std::string name; int64_t launches = 0; double ms = 0, flops = 0, bytes = 0; };
    std::vector<ProfAgg> prof_agg;
    // classes outside the fixed / Linear tables (attention, LayerNorm backward, FiLM-gradient reduction, small GEMMs per
    // instantiation): registered by name on first use, ids 64 + index; taken with the full mask or when singled out
    std::vector<std::string> named_cls;
    bool prof_named_all = true;
    std::vector<bool> prof_named_sel;      // named classes (id 64 + i) that get event pairs when prof_named_all is off
    int str_cls[40] = {0};          // tlin_str_kernel<256,XB,YB,EPI> instantiation -> profiling class id (0: none yet)
    int n_str_cls = 0;
    std::string str_cls_name[14];
    int head_on = getenv("GG_HEAD_FUSED") != nullptr;     // MLP heads as one forward and one backward launch (head.hip), bf16 mode: opt-in -
                               // measured equal to the nine launches it replaces (configs[0] 1.44 vs 1.47 ms, cfg3 25.25 vs 25.17 ms)
    int lnb_on = getenv("GG_NO_WST_LNB") == nullptr;      // dx1 += and LN1 backward in one weight-stationary kernel (production width, bf16 mode)
    int rstore_on = getenv("GG_NO_RSTORE") == nullptr;    // ... and of the pre-LayerNorm sums kept for the backward pass
    int xstore_on = getenv("GG_NO_XSTORE") == nullptr;   // bf16 storage of the encoder's LayerNorm outputs (production width, bf16 mode)
};

namespace {

void drop_graphs(gg_engine* e) {
    for (auto& g : e->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    e->graphs.clear();
}

// ------------------------------------------------------------------------------------------------
// layout
// ------------------------------------------------------------------------------------------------
void build_net(gg_engine* e, int role) {
    Net& n = e->net[role];
    n.role = role;
    const int E = e->E, F = e->F, H = e->H, Dt = e->Dt, Dp = e->Dp;
    n.V = role == GG_ROLE_GENERATOR ? e->L : e->G;
    n.OUT = role == GG_ROLE_GENERATOR ? e->G : 1;
    const std::string mlp = role == GG_ROLE_GENERATOR ? "generator" : "discriminator";
    // visible parameters first, ghosts (see Net::add_ghost) behind them
    const bool xa = e->xattn, eb = e->enc_bias;
    if (e->no_cond) {
        // vanilla_gan_unconditional.py:93-184: MLP heads only.  The first layer keeps its [H, V + E] storage with E zero
        // columns (the conditioning vector is zero, so their gradient is exactly zero and they stay zero); hosts see [:, :V]
        n.w1 = n.add(mlp + ".0.0.weight", H, n.V + E);
        n.b1 = n.add(mlp + ".0.0.bias", H);
        n.w2 = n.add(mlp + ".1.0.weight", H, H);
        n.b2 = n.add(mlp + ".1.0.bias", H);
        n.w3 = n.add("final_layer.weight", n.OUT, H);
        n.b3 = n.add("final_layer.bias", n.OUT);
        n.live = n.total;
        const long gh = n.add_ghost(8);           // never read: cond_forward / cond_backward return before touching them
        n.cls = n.film_w = n.film_b = n.te_w = n.te_b = n.pe_w = n.pe_b = gh;
        for (int l = 0; l < e->nl; ++l) {
            LayerP& L = n.layer[l];
            L.sa.inw = L.sa.inb = L.sa.ow = L.sa.ob = L.l1w = L.l1b = L.l2w = L.l2b = L.n1w = L.n1b = L.n2w = L.n2b = gh;
        }
        n.t2i.inw = n.t2i.inb = n.t2i.ow = n.t2i.ob = gh;
        n.i2t = n.t2i;
        return;
    }
    n.cls = n.add("patches_cls_token", 1, 1, E);
    if (e->film) {
        n.film_w = n.add("film_generator.weight", 2 * Dp, Dt);
        n.film_b = n.add("film_generator.bias", 2 * Dp);
    }
    if (xa) {
        n.te_w = n.add("text_encoder.weight", E, Dt);
        n.te_b = n.add("text_encoder.bias", E);
    }
    if (e->pe_ln) {     // nn.Sequential(Linear, ReLU, LayerNorm): modules 0 and 2 carry parameters
        n.pe_w = n.add("patches_encoder.0.weight", E, Dp);
        n.pe_b = n.add("patches_encoder.0.bias", E);
        n.pe_lnw = n.add("patches_encoder.2.weight", E);
        n.pe_lnb = n.add("patches_encoder.2.bias", E);
    } else {
        n.pe_w = n.add("patches_encoder.weight", E, Dp);
        n.pe_b = n.add("patches_encoder.bias", E);
    }
    for (int l = 0; l < e->nl; ++l) {
        const std::string p = "patches_transformer.layers." + std::to_string(l) + ".";
        LayerP& L = n.layer[l];
        L.sa.inw = n.add(p + "self_attn.in_proj_weight", 3 * E, E);
        if (eb) L.sa.inb = n.add(p + "self_attn.in_proj_bias", 3 * E);
        L.sa.ow = n.add(p + "self_attn.out_proj.weight", E, E);
        if (eb) L.sa.ob = n.add(p + "self_attn.out_proj.bias", E);
        L.l1w = n.add(p + "linear1.weight", F, E);
        if (eb) L.l1b = n.add(p + "linear1.bias", F);
        L.l2w = n.add(p + "linear2.weight", E, F);
        if (eb) L.l2b = n.add(p + "linear2.bias", E);
        L.n1w = n.add(p + "norm1.weight", E);
        if (eb) L.n1b = n.add(p + "norm1.bias", E);
        L.n2w = n.add(p + "norm2.weight", E);
        if (eb) L.n2b = n.add(p + "norm2.bias", E);
    }
    auto attn = [&](const std::string& p, AttnP& a) {
        a.inw = n.add(p + "in_proj_weight", 3 * E, E);
        a.inb = n.add(p + "in_proj_bias", 3 * E);
        a.ow = n.add(p + "out_proj.weight", E, E);
        a.ob = n.add(p + "out_proj.bias", E);
    };
    if (xa) {
        attn("patch2text_attention.", n.t2i);
        attn("text2patch_attention.", n.i2t);
    }
    n.w1 = n.add(mlp + ".0.0.weight", H, n.V + E);
    n.b1 = n.add(mlp + ".0.0.bias", H);
    n.w2 = n.add(mlp + ".1.0.weight", H, H);
    n.b2 = n.add(mlp + ".1.0.bias", H);
    n.w3 = n.add("final_layer.weight", n.OUT, H);
    n.b3 = n.add("final_layer.bias", n.OUT);
    n.live = n.total;
    if (!eb) {
        for (int l = 0; l < e->nl; ++l) {
            LayerP& L = n.layer[l];
            L.sa.inb = n.add_ghost(3 * E); L.sa.ob = n.add_ghost(E); L.l1b = n.add_ghost(F); L.l2b = n.add_ghost(E);
            L.n1b = n.add_ghost(E); L.n2b = n.add_ghost(E);
        }
    }
    if (!e->film) n.film_w = n.film_b = n.add_ghost(8);
    if (!xa) {      // never read: the offsets only have to be valid
        n.te_w = n.te_b = n.add_ghost(8);
        n.t2i.inw = n.t2i.inb = n.t2i.ow = n.t2i.ob = n.te_w;
        n.i2t = n.t2i;
    }
    // 2-D weights of the conditioning stack get bf16 shadows (the MLP head products are short-M and stay generic)
    for (const ParamInfo& pi : n.ps) {
        if (pi.ndim != 2) continue;
        if (pi.off == n.w1 || pi.off == n.w2 || pi.off == n.w3) continue;
        n.tab_index[pi.off] = (int)n.tab.size();
        n.tab.push_back(ShadowEntry{pi.off, pi.shape[0], pi.shape[1]});
    }
}

void carve_cond(gg_engine* e, Arena& a, CondActs& c, int R) {
    const long B = e->maxB, S = e->maxS, T = e->maxT, E = e->E, F = e->F, nh = e->nh;
    const long RB = R * B;
    c.gbpre = a.take<float>(B * 2 * e->Dp);
    c.gb = a.take<float>(B * 2 * e->Dp);
    c.tok = a.take<float>(B * T * E);
    c.x0 = a.take<float>(B * S * E);
    c.xrep = a.take<float>(RB * S * E);
    c.tokrep = a.take<float>(RB * T * E);
    c.mask = a.take<uint8_t>(B * S);
    if (e->pe_ln) {
        c.pe_h = a.take<float>(B * S * E);
        c.pe_y = a.take<float>(B * S * E);
        c.pe_st = a.take<float>(B * S * 2);
        c.pe_zero = a.take<float>(E);
    }
    for (int l = 0; l < e->nl; ++l) {
        LayerActs& L = c.L[l];
        L.qkv = a.take<float>(RB * S * 3 * E);
        L.P = a.take<float>(RB * nh * S * S);
        L.ctx = a.take<float>(RB * S * E);
        L.r1 = a.take<float>(RB * S * E);
        L.x1 = a.take<float>(RB * S * E);
        L.h = a.take<float>(RB * S * F);
        L.r2 = a.take<float>(RB * S * E);
        L.x2 = a.take<float>(RB * S * E);
        L.st1 = a.take<float>(RB * S * 2);
        L.st2 = a.take<float>(RB * S * 2);
        L.lse = a.take<float>(RB * nh * S);
    }
    c.t2i_q = a.take<float>(RB * E);
    c.t2i_kv = a.take<float>(RB * S * 2 * E);
    c.t2i_P = a.take<float>(RB * nh * S);
    c.t2i_ctx = a.take<float>(RB * E);
    c.t2i_out = a.take<float>(RB * E);
    c.t2i_xbar = a.take<float>(RB * nh * E);
    c.t2i_qt = a.take<float>(RB * nh * E);
    c.i2t_q = a.take<float>(RB * E);
    c.i2t_kv = a.take<float>(RB * T * 2 * E);
    c.i2t_P = a.take<float>(RB * nh * T);
    c.i2t_ctx = a.take<float>(RB * E);
    c.i2t_out = a.take<float>(RB * E);
    c.c = a.take<float>(RB * E);
}

size_t carve(gg_engine* e, void* base) {
    Arena a;
    a.base = static_cast<char*>(base);
    const long B = e->maxB, S = e->maxS, T = e->maxT, E = e->E, F = e->F, H = e->H, G = e->G, nh = e->nh;
    const long R = e->maxR, RB = R * B, Rb = (R > 1 ? 2 : 1) * B;   // Rb: rows that take part in backward
    const long P = e->maxP, Dp = e->Dp;
    carve_cond(e, a, e->actsG, 1);
    carve_cond(e, a, e->actsD, (int)R);
    carve_cond(e, a, e->actsP, R > 1 ? e->preR : 1);                       // the prefetch arena holds preR dropout replicas (generator_prefetch)
    for (int r = 0; r < 2; ++r) {
        Net& n = e->net[r];
        n.wb = a.take<char>((size_t)n.total * 2);
        n.wtb = a.take<char>((size_t)n.total * 2);
        n.wp3 = a.take<char>((size_t)n.total * 2 * 3);
        n.wtp3 = a.take<char>((size_t)n.total * 2 * 3);
        n.tab_dev = a.take<ShadowEntry>(n.tab.size() + 1);
        n.w8 = a.take<char>((size_t)n.total * 2);
        n.w8_amax = a.take<unsigned>(n.tab.size() + 1);
        n.w8_exp = a.take<int>(n.tab.size() + 1);
        n.wfrag = a.take<char>(enc_frag_bytes(e->nl));
        n.wfragb = a.take<char>(encb_frag_bytes(e->nl));
    }
    e->headG.a1 = a.take<float>(B * H); e->headG.a2 = a.take<float>(B * H); e->headG.out = nullptr;
    e->headD.a1 = a.take<float>(3 * B * H); e->headD.a2 = a.take<float>(3 * B * H); e->headD.out = a.take<float>(3 * B);
    e->headP.a1 = a.take<float>((long)e->preR * B * H); e->headP.a2 = a.take<float>((long)e->preR * B * H); e->headP.out = nullptr;
    e->c3P = a.take<float>((long)e->preR * B * E);
    e->X2 = a.take<float>(2 * B * G);
    e->Xpre = a.take<float>((long)GG_MAX_PREFETCH * B * G);
    e->c3 = a.take<float>(3 * B * E);
    e->Pfr = a.take<float>(2 * B * H);
    e->dseed = a.take<float>(2 * B);
    e->dA = a.take<float>(3 * B * H);
    e->dB = a.take<float>(3 * B * H);
    e->dc = a.take<float>(3 * B * E);
    e->gp_g2 = a.take<float>(B * H); e->gp_g1 = a.take<float>(B * H); e->gp_g1s = a.take<float>(B * H);
    e->gp_grad = a.take<float>(B * G); e->gp_nrm2 = a.take<float>(B); e->gp_coef = a.take<float>(B);
    e->gp_nrm2p = a.take<float>((long)gp_grad3_parts((int)G) * B);
    e->gp_dg1 = a.take<float>(B * H); e->gp_dg2 = a.take<float>(B * H);
    e->dxfake = a.take<float>(B * G);
    e->sumsq = a.take<float>(2 * 1024 + 8);
    e->dev_words = reinterpret_cast<uint32_t*>(a.take<float>(16));      // per network: <= 1024 partial sums of squares (k_sumsq); [2048..] scratch
    e->sPd = a.take<float>(RB * nh * S * S);
    e->sdP = a.take<float>(Rb * nh * S * S);
    e->sdqkv = a.take<float>(Rb * S * 3 * E);
    e->sdx = a.take<float>(Rb * S * E);
    e->sdr = a.take<float>(Rb * S * E);
    e->sdres = a.take<float>(Rb * S * E);
    e->sdctx = a.take<float>(Rb * S * E);
    e->sdh = a.take<float>(Rb * S * F);
    e->s_dt = a.take<float>(Rb * E); e->s_dp = a.take<float>(Rb * E); e->s_dq = a.take<float>(Rb * E); e->s_dq2 = a.take<float>(Rb * E);   // I2T / T2I query gradients: separate, the side-stream leaves read them late
    e->s_tmpE = a.take<float>(Rb * E);
    e->s_dkv = a.take<float>(Rb * S * 2 * E);
    e->s_dkv2 = a.take<float>(Rb * T * 2 * E);
    e->s_dtokrep = a.take<float>(Rb * T * E);
    e->s_dtok = a.take<float>(B * T * E);
    e->s_dtok0 = a.take<float>(Rb * E);
    e->s_dx0 = a.take<float>(B * S * E);
    e->s_demb = a.take<float>(B * P * E);
    e->s_mod = a.take<float>(B * P * Dp);
    e->s_dmod = a.take<float>(B * P * Dp);
    e->s_dgb = a.take<float>(B * 2 * Dp);
    e->s_delta = a.take<float>(Rb * nh * S);
    e->s_dqt = a.take<float>(Rb * nh * E);
    e->s_dxbar = a.take<float>(Rb * nh * E);
    return a.off + 256;
}

// ------------------------------------------------------------------------------------------------
// GEMM helpers (Linear forward / backward-data / backward-weight)
// ------------------------------------------------------------------------------------------------
struct Ctx {
    gg_engine* e;
    hipStream_t st;
    int grid_pct = 0;      // share of the compute units the persistent Linear grids launched through this context take (0: the kernels' default)
};

inline long tiles_of(long M, long N) { return ((M + 127) / 128) * ((N + 127) / 128); }

inline void phase_mark(Ctx& c, const char* name) {
    gg_engine* e = c.e;
    if (!e->phase_on || e->capturing || e->phase_marks.size() >= 512) return;
    if (e->phase_pool.size() <= e->phase_marks.size()) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return;
        e->phase_pool.push_back(ev);
    }
    hipEvent_t ev = e->phase_pool[e->phase_marks.size()];
    if (hipEventRecord(ev, c.st) == hipSuccess) e->phase_marks.emplace_back(name, ev);
}

int named_class(gg_engine* e, const char* name) {
    for (size_t i = 0; i < e->named_cls.size(); ++i)
        if (e->named_cls[i] == name) return 64 + (int)i;
    e->named_cls.emplace_back(name);
    return 64 + (int)e->named_cls.size() - 1;
}
inline bool prof_wanted(const gg_engine* e, int cls) {
    if (!e->prof_on) return false;
    return cls < 64 ? ((e->prof_mask >> cls) & 1u) != 0
                    : (e->prof_named_all || ((size_t)(cls - 64) < e->prof_named_sel.size() && e->prof_named_sel[(size_t)(cls - 64)]));
}
// event pair around a launch of a named class (the kernels taken this way run for >= tens of microseconds, or are counted
// only in the serialised all-classes step)
struct ProfScope {
    Ctx& c;
    bool on = false;
    gg_engine::ProfRec r;
    ProfScope(Ctx& c_, const char* name, double flops, double bytes) : c(c_) {
        gg_engine* e = c.e;
        if (!e->prof_on) return;
        const int cls = named_class(e, name);
        if (!prof_wanted(e, cls)) return;
        if (e->prof_next + 2 > e->prof_pool.size())
            for (int i = 0; i < 4096; ++i) {
                hipEvent_t ev;
                if (hipEventCreate(&ev) != hipSuccess) return;
                e->prof_pool.push_back(ev);
            }
        r.cls = cls; r.flops = flops; r.bytes = bytes;
        r.e0 = e->prof_pool[e->prof_next++];
        r.e1 = e->prof_pool[e->prof_next++];
        on = hipEventRecord(r.e0, c.st) == hipSuccess;
    }
    ~ProfScope() {
        if (on && hipEventRecord(r.e1, c.st) == hipSuccess) c.e->prof_recs.push_back(r);
    }
};

int run_gemm(Ctx& c, const GemmP& p) {
    gg_engine* e = c.e;
    e->launches++;
    // bf16x3: the few-tile products (MLP heads, cross-attention projections, FiLM head) run on the exact fp32-input MFMA
    const bool bf16 = e->precision == GG_PREC_BF16 && !e->x3;
    if (e->x3 && e->precision == GG_PREC_BF16 && e->small_on && gemm_small_wanted(p) && gemm_small_x3_ok(p)) {
        ProfScope ps(c, "gemm_tiny_kernel<x3>", 6.0 * 2.0 * p.M * p.N * (double)p.K * p.batch,
                     4.0 * p.batch * ((double)p.M * p.K + (double)p.K * p.N + (double)p.M * p.N));
        return gemm_small_x3(p, c.st);
    }
    const bool small = bf16 && e->small_on && gemm_small_wanted(p);
    static const char* small_names[4] = {"gemm_small_kernel<0,0>", "gemm_small_kernel<0,1>", "gemm_small_kernel<1,0>", "gemm_small_kernel<1,1>"};
    const int cls_ = !e->prof_on ? 0 : small ? named_class(e, small_names[p.layA * 2 + p.layB]) : p.layA * 2 + p.layB + (bf16 ? 4 : 0);
    if (!prof_wanted(e, cls_)) return small ? gemm_small(p, c.st) : (bf16 ? gemm_bf16(p, c.st) : gemm_f32(p, c.st));
    if (e->prof_next + 2 > e->prof_pool.size()) {
        for (int i = 0; i < 4096; ++i) {
            hipEvent_t ev;
            GG_CHECK_HIP(hipEventCreate(&ev));
            e->prof_pool.push_back(ev);
        }
    }
    gg_engine::ProfRec r;
    r.cls = p.layA * 2 + p.layB + (bf16 ? 4 : 0);
    const double b = (double)p.batch;
    r.flops = 2.0 * p.M * p.N * (double)p.K * b;
    r.bytes = 4.0 * b * ((double)p.M * p.K + (double)p.K * p.N + (double)p.M * p.N);
    r.e0 = e->prof_pool[e->prof_next++];
    r.e1 = e->prof_pool[e->prof_next++];
    GG_CHECK_HIP(hipEventRecord(r.e0, c.st));
    if (small) r.cls = cls_;
    int rc = small ? gemm_small(p, c.st) : (bf16 ? gemm_bf16(p, c.st) : gemm_f32(p, c.st));
    GG_CHECK_HIP(hipEventRecord(r.e1, c.st));
    e->prof_recs.push_back(r);
    return rc;
}

// C[M,N] = act( A[M,K] @ W[N,K]^T + bias (+ C) )
int lin_fwd(Ctx& c, const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc, int M,
            int N, int K, int act = ACT_NONE, float slope = 0.f, int accumulate = 0) {
    GemmP p;
    p.A = A; p.B = W; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = ldc;
    p.layA = LAY_KC; p.layB = LAY_KC;
    const long tiles = tiles_of(M, N);
    if (tiles < 48 && K >= 1024 && ldc == N) {       // short-and-deep: split K over workgroups, finish elementwise
        int sk = (int)std::min<long>((K + 255) / 256, std::max<long>(1, 256 / tiles));
        if (!accumulate) { GG_TRY(k_fill(C, (long)M * N, 0.f, c.st)); c.e->launches++; }
        p.splitk = sk;
        GG_TRY(run_gemm(c, p));
        if (bias || act != ACT_NONE) { GG_TRY(k_bias_act(C, bias, M, N, act, slope, c.st)); c.e->launches++; }
        return 0;
    }
    p.bias = bias; p.act = act; p.slope = slope; p.accumulate = accumulate;
    return run_gemm(c, p);
}

// dX[M,K] (+)= dY[M,N] @ W[N,K]
int lin_bwd_data(Ctx& c, const float* dY, long ldy, const float* W, long ldw, float* dX, long ldx, int M, int N, int K,
                 int accumulate = 0) {
    GemmP p;
    p.A = dY; p.B = W; p.C = dX; p.M = M; p.N = K; p.K = N; p.lda = ldy; p.ldb = ldw; p.ldc = ldx;
    p.layA = LAY_KC; p.layB = LAY_KS;
    const long tiles = tiles_of(M, K);
    if (tiles < 48 && N >= 1024 && ldx == K) {
        int sk = (int)std::min<long>((N + 255) / 256, std::max<long>(1, 256 / tiles));
        if (!accumulate) { GG_TRY(k_fill(dX, (long)M * K, 0.f, c.st)); c.e->launches++; }
        p.splitk = sk;
        return run_gemm(c, p);
    }
    p.accumulate = accumulate;
    return run_gemm(c, p);
}

// dW[N,K] += dY[M,N]^T @ X[M,K]     (reduction over the M rows, split over workgroups, fp32 atomics)
// film: X is used as  g[m / group] * X + b[m / group]  (the FiLM-modulated patches) without materialising it; only the
// token-reduction kernel implements it - callers check wgrad_film_ok() first
bool wgrad_film_ok(gg_engine* e, const float* dY, long ldy, const float* X, long ldx, int M, int N, int K, int group) {
    return e->wgrad_on && e->precision == GG_PREC_BF16 && group >= 32 && wgrad_supported(dY, ldy, 0, X, ldx, 0, M, N, K);
}
// dbias: bias gradient (column sums of dY) - folded into the token-reduction kernel when dY is stored in bf16 (the sums
// then use exactly the stored values), a separate column-sum launch otherwise
int lin_bwd_weight(Ctx& c, const float* dY, long ldy, const float* X, long ldx, float* dW, long ldw, int M, int N, int K,
                   int dy_bf16 = 0, int x_bf16 = 0, const WgradFilm* film = nullptr, float* dbias = nullptr, long x_mod = 0) {
    gg_engine* e = c.e;
    if (e->wgrad_on && e->precision == GG_PREC_BF16 && wgrad_supported(dY, ldy, dy_bf16, X, ldx, x_bf16, M, N, K)) {
        e->launches++;
        gg_engine::ProfRec r;
        // classes follow the kernel instantiations: wgrad_kernel<dY bf16, X bf16, FiLM on the fly, FiLM gradient>
        const int wcls = film ? 17 : (dy_bf16 ? (x_bf16 ? 15 : 10) : 16);      // (bf16x3: the fp32 / fp32 and FiLM instantiations with X3)
        const bool prof = e->prof_on && ((e->prof_mask >> wcls) & 1u);
        if (prof) {
            if (e->prof_next + 2 > e->prof_pool.size()) {
                for (int i = 0; i < 4096; ++i) {
                    hipEvent_t ev;
                    GG_CHECK_HIP(hipEventCreate(&ev));
                    e->prof_pool.push_back(ev);
                }
            }
            r.cls = wcls;
            r.flops = 2.0 * M * N * (double)K;
            r.bytes = (double)M * N * (dy_bf16 ? 2 : 4) + (double)M * K * (x_bf16 ? 2 : 4) + 4.0 * N * K;
            r.e0 = e->prof_pool[e->prof_next++];
            r.e1 = e->prof_pool[e->prof_next++];
            GG_CHECK_HIP(hipEventRecord(r.e0, c.st));
        }
        GG_TRY(wgrad(dY, ldy, dy_bf16, X, ldx, x_bf16, dW, ldw, M, N, K, c.st, film, nullptr, dy_bf16 ? dbias : nullptr, x_mod, e->x3));
        if (dbias && !dy_bf16) { GG_TRY(k_colsum(dY, M, N, ldy, dbias, c.st, 0)); e->launches++; }
        if (prof) {
            GG_CHECK_HIP(hipEventRecord(r.e1, c.st));
            e->prof_recs.push_back(r);
        }
        return 0;
    }
    GG_REQUIRE(!film && x_mod == 0, "lin_bwd_weight: FiLM / repeated-row operands need the token-reduction kernel");
    if (dbias) { GG_TRY(k_colsum(dY, M, N, ldy, dbias, c.st, dy_bf16)); e->launches++; }
    GemmP p;
    p.A = dY; p.B = X; p.C = dW; p.M = N; p.N = K; p.K = M; p.lda = ldy; p.ldb = ldx; p.ldc = ldw;
    p.layA = LAY_KS; p.layB = LAY_KS; p.a_bf16 = dy_bf16; p.b_bf16 = x_bf16;
    const long tiles = tiles_of(N, K);
    long sk = std::max<long>(1, std::min<long>((M + 255) / 256, (512 + tiles - 1) / tiles));
    p.splitk = (int)sk;
    p.accumulate = 1;
    return run_gemm(c, p);
}

#define KL(call)                 \
    do {                         \
        GG_TRY(call);            \
        c.e->launches++;         \
    } while (0)

inline bool use_tlin(gg_engine* e) { return e->tlin_on && e->precision == GG_PREC_BF16; }

int refresh_shadows(Ctx& c, Net& n) {
    if (!use_tlin(c.e) || n.tab.empty()) return 0;
    if (c.e->x3) {          // the split-operand Linears read pre-split bf16 parts of W and W^T
        KL(k_shadow_parts(n.w, n.wp3, n.wtp3, n.total, n.tab_dev, (int)n.tab.size(), c.st));
        return 0;
    }
    KL(k_shadow_weights(n.w, n.wb, n.wtb, n.tab_dev, (int)n.tab.size(), c.st));
    if (c.e->ffn2_on && g_lab.k_enc_frag_weights && !c.e->fp8_fwd && c.e->bstore_on && c.e->xstore_on && c.e->rstore_on && !c.e->no_cond && c.e->E == 256 && c.e->F == 512) {      // cond_forward's streamed-FFN route
        long o1[MAXL], o2[MAXL];
        for (int l = 0; l < c.e->nl; ++l) { o1[l] = n.layer[l].l1w; o2[l] = n.layer[l].l2w; }
        KL(g_lab.k_enc_frag_weights(n.w, o1, o2, c.e->nl, n.wfrag, c.st));
    }
    if (c.e->encb_on && g_lab.k_encb_frag_weights && !c.e->no_cond && c.e->E == 256 && c.e->F == 512) {
        long o1[MAXL], o2[MAXL], oo[MAXL];
        for (int l = 0; l < c.e->nl; ++l) { o1[l] = n.layer[l].l1w; o2[l] = n.layer[l].l2w; oo[l] = n.layer[l].sa.ow; }
        KL(g_lab.k_encb_frag_weights(n.w, o1, o2, oo, c.e->nl, n.wfragb, c.st));
    }
    if (c.e->fp8_fwd) {
        GG_TRY(k_shadow_weights_fp8(n.w, n.w8, n.w8_amax, n.w8_exp, n.tab_dev, (int)n.tab.size(), c.st));
        c.e->launches += 2;
    }
    return 0;
}
// Switches a forward Linear of an encoder layer to e4m3 operands when the engine is in fp8 mode and an fp8 instantiation
// takes the shape (otherwise the bf16 kernel runs): weights from the e4m3 shadow, activations quantised as x * 2^x_exp.
// Activations entering these Linears are LayerNorm outputs, attention contexts and ReLU outputs - O(1) magnitudes - so a
// static 2^3 keeps |x| <= 56 in range (the conversion clamps) with 2e-3 as the smallest normal value.
constexpr int FP8_X_EXP = 3;
void maybe_fp8(gg_engine* e, const Net& n, TlinP& t, long w_off) {
    if (!e->fp8_fwd) return;
    auto it = n.tab_index.find(w_off);
    if (it == n.tab_index.end()) return;
    TlinP f = t;
    f.fp8 = 1; f.W = n.w8 + 2 * w_off; f.w_exp = n.w8_exp + it->second; f.x_exp = FP8_X_EXP;
    if (tlin_fp8_supported(f)) t = f;
}
inline const void* WB(const Net& n, long off) { return n.wb + 2 * off; }      // bf16 W   at flat offset `off`
inline const void* WTB(const Net& n, long off) { return n.wtb + 2 * off; }    // bf16 W^T at flat offset `off`

// launches the token-on-lane kernel when enabled and the shape qualifies; returns 1 if it ran
// bf16x3: the call sites hand over bf16 shadow weights (W or W^T at the flat offset of the tensor); the split-operand kernel reads
// the three-part shadow of the same orientation at the same offset
// returns the number of operand parts: 3 (six products, fp32-grade) for a forward Linear (W: its results decide ReLU gates), 2 (three
// products) for a backward one (W^T); 0: not a shadow pointer
int x3_weights(gg_engine* e, TlinP& t) {
    const char* w = reinterpret_cast<const char*>(t.W);
    for (Net& n : e->net) {
        if (n.wb && w >= n.wb && w < n.wb + 2 * (size_t)n.total) { t.W = n.wp3 + (w - n.wb); t.w_part_stride = n.total; return 3; }
        if (n.wtb && w >= n.wtb && w < n.wtb + 2 * (size_t)n.total) { t.W = n.wtp3 + (w - n.wtb); t.w_part_stride = n.total; return 2; }
    }
    return 0;
}
int try_tlin3(Ctx& c, const TlinP& p_in) {
    gg_engine* e = c.e;
    TlinP p = p_in;
    if (p.x_bf16 || p.y_bf16 || p.mask_bf16 || p.fp8) return 0;
    const int ns = x3_weights(e, p);
    if (!ns || !tlin3_supported(p)) return 0;
    e->launches++;
    const int cls = e->prof_on ? named_class(e, ns == 3 ? "tlin3_kernel<8,3>" : "tlin3_kernel<8,2>") : 0;
    if (e->prof_on && prof_wanted(e, cls)) {
        if (e->prof_next + 2 > e->prof_pool.size())
            for (int i = 0; i < 4096; ++i) {
                hipEvent_t ev;
                if (hipEventCreate(&ev) != hipSuccess) return -1;
                e->prof_pool.push_back(ev);
            }
        gg_engine::ProfRec r;
        r.cls = cls;
        r.flops = (ns == 3 ? 6.0 : 3.0) * 2.0 * p.M * p.N * (double)p.K;          // bf16 MFMA passes actually issued
        const double MN = (double)p.M * p.N;
        r.bytes = 4.0 * ((double)p.M * p.K + MN + (p.ln_g ? MN : 0.0) + (p.res ? MN : 0.0) + (p.accumulate ? MN : 0.0) + (p.mask_ref ? MN : 0.0) +
                         (double)p.N * p.K);
        r.e0 = e->prof_pool[e->prof_next++];
        r.e1 = e->prof_pool[e->prof_next++];
        tlin3_time_next(r.e0, r.e1);
        if (tlin3(p, c.st, ns) != 0) return -1;
        e->prof_recs.push_back(r);
        return 1;
    }
    return tlin3(p, c.st, ns) == 0 ? 1 : -1;
}
int try_tlin(Ctx& c, const TlinP& p_in) {
    TlinP p = p_in;
    if (p.grid_pct == 0) p.grid_pct = c.grid_pct;
    if (use_tlin(c.e) && c.e->x3) return try_tlin3(c, p);
    if (!use_tlin(c.e) || !tlin_supported(p)) return 0;
    c.e->launches++;
    static const int tlin_cls[5] = {8, 9, 12, 13, 14};
    const int kc = tlin_kernel_class(p);
    int tcls = kc < 16 ? tlin_cls[kc] : 8;
    if (kc >= 16) {     // stream instantiations get a class each, in order of first appearance (ids 18..31)
        gg_engine* e = c.e;
        int& id = e->str_cls[kc - 16];
        if (id == 0 && e->n_str_cls < 12) {      // ids stay below 30: the class mask travels shifted by one in an int
            id = 18 + e->n_str_cls++;
            char nm[64];
            static const char* extra[21] = {"wst_ln_kernel<4,2,16,true,0>", "wst_ln_kernel<8,1,32,true,0>", "wst_ln_kernel<8,1,32,true,1>",
                                            "wst_ln_kernel<8,2,16,false,2>", "wst_ln_kernel<8,2,16,true,3>", "wst_ln_kernel<4,2,16,true,2>",
                                            "wst_ln_kernel<8,1,48,true,1>",
                                            "tlin_res16_kernel<8,256,true,1,true>", "tlin_str_kernel<256,false,true,0,true>",
                                            "tlin_str_kernel<256,false,true,1,true>", "wst_ln_kernel<4,2,16,true,0,true>",
                                            "wst_ln_kernel<8,1,32,true,0,true>", "wst_ln_kernel<8,2,16,false,2,true>",
                                            "wst_ln_kernel<8,3,16,false,2,true>", "wst_ln_kernel<4,2,16,false,2,false,3>", "wst_ln_kernel<4,1,48,true,1,false,2>",
                                            "wst_ln_kernel<8,2,16,true,2>", "wst_ln_kernel<4,2,16,true,2,false,3>", "wst_ln_kernel<8,1,32,true,4>",
                                            "wst_ln_kernel<8,2,16,true,2,true>", "wst_ln_kernel<8,3,16,true,2,true>"};
            if (kc >= 32) snprintf(nm, sizeof nm, "%s", extra[kc - 32]);
            else
            snprintf(nm, sizeof nm, "tlin_str_kernel<256,%s,%s,%d>", ((kc - 16) & 1) ? "true" : "false", ((kc - 16) & 2) ? "true" : "false", (kc - 16) >> 2);
            e->str_cls_name[id - 18] = nm;
        }
        if (id) tcls = id;
    }
    if (c.e->prof_on && ((c.e->prof_mask >> tcls) & 1u)) {
        gg_engine* e = c.e;
        if (e->prof_next + 2 > e->prof_pool.size()) {
            for (int i = 0; i < 4096; ++i) {
                hipEvent_t ev;
                if (hipEventCreate(&ev) != hipSuccess) return -1;
                e->prof_pool.push_back(ev);
            }
        }
        gg_engine::ProfRec r;
        r.cls = tcls;
        r.flops = 2.0 * p.M * p.N * (double)p.K;
        // algorithmic bytes at the element sizes actually stored: X once, Y once (+ LayerNorm output), residual,
        // previous Y when accumulating, sign-mask reference, bf16 weights once
        const double MN = (double)p.M * p.N;
        const double MNy = (p.y_rows >= 0 && p.y_rows < p.M) ? (double)p.y_rows * p.N : MN;     // rows whose pre-LN sum is stored
        r.bytes = (p.x_bf16 ? 2.0 : 4.0) * (double)p.M * p.K + (p.y_bf16 ? 2.0 : 4.0) * MNy + (p.ln_g ? (p.ln_y_bf16 ? 2.0 : 4.0) * MN : 0.0) +
                  (p.res ? (p.res_bf16 ? 2.0 : 4.0) * MN : 0.0) + (p.accumulate ? 4.0 * MN : 0.0) + (p.mask_ref ? (p.mask_bf16 ? 2.0 : 4.0) * MN : 0.0) +
                  2.0 * p.N * p.K + (p.lnb_dres ? 2.0 * MN : 0.0);
        r.e0 = e->prof_pool[e->prof_next++];
        r.e1 = e->prof_pool[e->prof_next++];
        tlin_time_next(r.e0, r.e1);           // dispatch timestamps of the kernel itself (what rocprofv3 reports), no barrier packets
        if (tlin(p, c.st) != 0) return -1;
        e->prof_recs.push_back(r);
        return 1;
    }
    return tlin(p, c.st) == 0 ? 1 : -1;
}
#define TLIN_OR(p_, fallback)                 \
    do {                                      \
        const int _t = try_tlin(c, p_);       \
        if (_t < 0) return -1;                \
        if (_t == 0) { fallback; }            \
    } while (0)

#define TLIN_MUST(p_)                                                                           \
    do {                                                                                        \
        const int _t = try_tlin(c, p_);                                                         \
        if (_t < 0) return -1;                                                                  \
        if (_t == 0) { set_error("bf16-stored operand reached a shape tlin cannot run"); return -2; } \
    } while (0)

DropKey dkey(gg_engine* e, const CondActs& a, int net, int layer, int site) {
    DropKey k = make_drop_key(a.drop, e->seed, (uint32_t)(net * 1000 + layer * 10 + site), a.call);
    if (a.drop > 0.f) k.epoch = e->dev_words;
    return k;
}

// ------------------------------------------------------------------------------------------------
// conditioning stack forward (R:198-224)   [numpy_oracle.cond_fwd]
// ------------------------------------------------------------------------------------------------
// keep: number of leading replicas whose backward will run (their pre-LayerNorm sums are stored); -1 = all
int cond_forward(Ctx& c, Net& n, const gg_cond* in, CondActs& a, int R, float drop, int keep = -1) {
    gg_engine* e = c.e;
    const int B = in->B, P = in->P, T = in->T, S = P + 1, E = e->E, F = e->F, nh = e->nh, dh = e->dh;
    const int Dt = e->Dt, Dp = e->Dp;
    const long RB = (long)R * B;
    const long keep_rows = keep < 0 ? -1 : (long)keep * B * S;
    a.B = B; a.R = R; a.P = P; a.T = T; a.drop = drop; a.call = ++e->call_counter;
    if (e->no_cond) {        // unconditional model: c == 0
        KL(k_fill(a.c, RB * E, 0.f, c.st));
        return 0;
    }
    const float* w = n.w;
    // FiLM parameters from the text CLS token (row b of `text` viewed with ld = T*Dt)
    if (e->film) {
        GG_TRY(lin_fwd(c, in->text, (long)T * Dt, w + n.film_w, Dt, w + n.film_b, a.gbpre, 2 * Dp, B, 2 * Dp, Dt));
        KL(k_film_act_fwd(a.gbpre, a.gb, B, Dp, c.st));
    }
    // text encoder
    if (!e->xattn) {
        // FiLM-only variant: no token encoder, no cross attention (conditional_gan_film.py:130-152)
    } else if ((long)B * T >= 4096) {      // many text tokens: the token-on-lane kernel (a handful of rows stays with the small-GEMM path)
        TlinP t;
        t.X = in->text; t.ldx = Dt; t.M = (long)B * T; t.W = WB(n, n.te_w); t.ldw = Dt; t.bias = w + n.te_b;
        t.Y = a.tok; t.ldy = E; t.N = E; t.K = Dt;
        TLIN_OR(t, GG_TRY(lin_fwd(c, in->text, Dt, w + n.te_w, Dt, w + n.te_b, a.tok, E, B * T, E, Dt)));
    } else {
        GG_TRY(lin_fwd(c, in->text, Dt, w + n.te_w, Dt, w + n.te_b, a.tok, E, B * T, E, Dt));
    }
    if (e->pe_ln) {
        // conditional_gan_img_transformer.py:106-110,126: Linear -> ReLU -> LayerNorm on the raw patches (no FiLM); the
        // ReLU output (LayerNorm's input) and the row statistics are kept for the backward
        TlinP t;
        t.X = in->patches; t.ldx = Dp; t.M = (long)B * P; t.W = WB(n, n.pe_w); t.ldw = Dp; t.bias = w + n.pe_b;
        t.Y = a.pe_h; t.ldy = E; t.N = E; t.K = Dp;
        TLIN_OR(t, GG_TRY(lin_fwd(c, in->patches, Dp, w + n.pe_w, Dp, w + n.pe_b, a.pe_h, E, B * P, E, Dp)));
        KL(k_bias_act(a.pe_h, nullptr, (long)B * P, E, ACT_LRELU, 0.f, c.st));
        KL(k_fill(a.pe_zero, E, 0.f, c.st));
        KL(k_add_layernorm_fwd(a.pe_zero, 1, a.pe_h, w + n.pe_lnw, w + n.pe_lnb, a.pe_y, a.pe_st, (long)B * P, E, DropKey(), c.st));
        KL(k_scatter_patch_rows(a.x0, a.pe_y, B, P, E, c.st));
    } else
    // patch encoder with FiLM fused on the A operand; rows land behind the CLS row of each sample
    {
        TlinP t;
        t.X = in->patches; t.ldx = Dp; t.M = (long)B * P; t.W = WB(n, n.pe_w); t.ldw = Dp; t.bias = w + n.pe_b;
        t.Y = a.x0; t.ldy = E; t.N = E; t.K = Dp;
        t.film_g = a.gb; t.film_b = a.gb + Dp; t.film_ld = 2 * Dp; t.film_group = P; t.y_row_group = P;
        GemmP p;
        p.A = in->patches; p.B = w + n.pe_w; p.C = a.x0; p.M = B * P; p.N = E; p.K = Dp;
        p.lda = Dp; p.ldb = Dp; p.ldc = E; p.bias = w + n.pe_b;
        p.film_gamma = a.gb; p.film_beta = a.gb + Dp; p.film_ld = 2 * Dp; p.film_group = P; p.c_row_group = P;
        TLIN_OR(t, GG_TRY(run_gemm(c, p)));
    }
    KL(k_write_cls(a.x0, w + n.cls, B, S, E, c.st));
    KL(k_build_mask(in->patch_pad, a.mask, B, P, c.st));
    const float scale = 1.f / sqrtf((float)dh);
    // bf16x3: the split-operand kernels (three operand parts in the forward pass, two in the backward one; any S <= 2048)
    const bool use_flash3 = e->flash && e->precision == GG_PREC_BF16 && e->x3 && flash_attn_x3_supported(S, E, nh);
    const bool use_flash = use_flash3 || (e->flash && e->precision == GG_PREC_BF16 && !e->x3 && flash_attn_supported(S, E, nh));
    a.flash = use_flash;
    // bf16 storage of the tensors that are only ever read as bf16 MFMA operands: every producer / consumer
    // must be a tlin / flash kernel, which holds for E in {64,128,256} (see tlin_supported)
    const bool bst = e->bstore_on && !e->x3 && use_flash && use_tlin(e) && (E == 64 || E == 128 || E == 256);
    a.bst = bst;
    // The LayerNorm outputs inside the encoder are read as MFMA operands (QKV, FFN1, the two weight-gradient products: rounded to
    // bf16 on load anyway) and as the residual of the next sub-block: at the production width, where every one of those
    // consumers is a weight-stationary kernel, they are stored ONCE, in bf16 (half the bytes on six passes per layer).  The
    // pre-LayerNorm sums r1 / r2 and the statistics stay fp32 (LayerNorm backward), and so does the last layer's output
    // (cross-attention and CLS read it as fp32).
    static const bool wst_env = getenv("GG_NO_WST") == nullptr && getenv("GG_NO_WST2") == nullptr && getenv("GG_NO_WST_QKV") == nullptr;
    const bool xst = bst && e->xstore_on && wst_env && E == 256 && F == 2 * E;
    a.xst = xst;
    // ... and so are the pre-LayerNorm sums the backward pass re-reads (xhat = (r - mean) * rstd with the fp32 statistics of the
    // unrounded sum: a 2^-9 perturbation of xhat, the same order as the bf16 operands of every product around it)
    const bool rst = xst && e->rstore_on;
    a.rst = rst;
    // The replicas differ only by their dropout draws, and nothing is dropped before the first attention: the layer-0
    // input x0 and its QKV projection are the same for all of them.  With the fused kernels (row / sample indices taken
    // modulo the un-replicated size) neither the R-fold copy of x0 nor R-1 of the R projections exist.
    // the text tokens carry no dropout: their I2T K / V projection is the same for every replica
    static const bool no_i2t_share = getenv("GG_NO_I2T_SHARE") != nullptr;
    a.i2t_shared = e->xattn && !no_i2t_share && sq_attn_shared_ok(T, E, nh, std::max(1, std::min(R, 3)));
    // (bf16x3: the split-operand Linear, attention and weight-gradient kernels take the same modulo-indexed operands; GG_NO_SHARE0_X3: A/B)
    static const bool no_share0_x3 = getenv("GG_NO_SHARE0_X3") != nullptr;
    const bool share0 = R > 1 && (bst || (use_flash3 && !no_share0_x3)) && e->wgrad_on && (long)B * S >= 4096;      // (the weight-gradient kernel must engage)
    a.share0 = share0;
    const float* x_in = a.x0;
    const float* tok = a.tok;
    if (R > 1) {
        if (!share0) {
            KL(k_copy_rows_bcast(a.xrep, a.x0, RB * S, (long)B * S, E, c.st));
            x_in = a.xrep;
        }
        // shared I2T keys: only the text CLS row (token 0) of every replica is read from the replicated layout
        if (!e->xattn) {
        } else if (a.i2t_shared) KL(k_copy_rows_strided_bcast(a.tokrep, (long)T * E, a.tok, (long)T * E, RB, B, E, c.st));
        else KL(k_copy_rows_bcast(a.tokrep, a.tok, RB * T, (long)B * T, E, c.st));
        tok = a.tokrep;
    }
    for (int l = 0; l < e->nl; ++l) {
        LayerActs& L = a.L[l];
        const LayerP& lp = n.layer[l];
        {
            TlinP t;
            const bool shared = share0 && l == 0;
            t.X = x_in; t.ldx = E; t.M = shared ? (long)B * S : RB * S; t.W = WB(n, lp.sa.inw); t.ldw = E; t.bias = w + lp.sa.inb;
            t.Y = L.qkv; t.ldy = 3 * E; t.N = 3 * E; t.K = E; t.y_bf16 = bst; t.x_bf16 = xst && l > 0;
            if (bst) maybe_fp8(e, n, t, lp.sa.inw);
            if (bst) TLIN_MUST(t);
            else TLIN_OR(t, GG_TRY(lin_fwd(c, x_in, E, w + lp.sa.inw, E, w + lp.sa.inb, L.qkv, 3 * E, (int)(RB * S), 3 * E, E)));
        }
        const DropKey kA = dkey(e, a, n.role, l, 0);
        if (use_flash) {
            {   // algorithmic: packed QKV read once (B rows when layer 0 is shared by the replicas), context written once
                const double tok = (double)RB * S, qtok = (share0 && l == 0 ? (double)B : (double)RB) * S, es = bst ? 2.0 : 4.0;
                ProfScope ps(c, use_flash3 ? flash_attn_x3_kernel_name(0) : flash_attn_kernel_name(0, S, E, nh), (use_flash3 ? 6.0 : 1.0) * 4.0 * tok * S * E,
                             es * (3.0 * qtok * E + tok * E) + 4.0 * tok * nh);
                if (use_flash3) KL(flash_attn_fwd_x3(L.qkv, a.mask, B, L.ctx, L.lse, RB, S, E, nh, kA, c.st, share0 && l == 0 ? B : 0, 3));
                else
                KL(flash_attn_fwd(L.qkv, a.mask, B, L.ctx, L.lse, RB, S, E, nh, kA, bst, c.st, share0 && l == 0 ? B : 0));
            }
        } else {
            {   // scores[b,h] = scale * Q_h K_h^T, padded keys -> -inf
                GemmP p;
                p.A = L.qkv; p.B = L.qkv + E; p.C = L.P; p.M = S; p.N = S; p.K = dh;
                p.lda = 3 * E; p.ldb = 3 * E; p.ldc = S; p.layA = LAY_KC; p.layB = LAY_KC;
                p.batch = (int)(RB * nh); p.batch_inner = nh;
                p.sAo = (long)S * 3 * E; p.sAi = dh; p.sBo = (long)S * 3 * E; p.sBi = dh;
                p.sCo = (long)nh * S * S; p.sCi = (long)S * S;
                p.alpha = scale; p.colmask = a.mask; p.colmask_stride = S; p.colmask_mod = B;
                GG_TRY(run_gemm(c, p));
            }
            KL(k_softmax_rows(L.P, e->sPd, RB * nh * S, S, kA, c.st));
            {   // ctx[b,:,h] = Pd[b,h] V_h
                GemmP p;
                p.A = drop > 0.f ? e->sPd : L.P; p.B = L.qkv + 2 * E; p.C = L.ctx; p.M = S; p.N = dh; p.K = S;
                p.lda = S; p.ldb = 3 * E; p.ldc = E; p.layA = LAY_KC; p.layB = LAY_KS;
                p.batch = (int)(RB * nh); p.batch_inner = nh;
                p.sAo = (long)nh * S * S; p.sAi = (long)S * S; p.sBo = (long)S * 3 * E; p.sBi = dh;
                p.sCo = (long)S * E; p.sCi = dh;
                GG_TRY(run_gemm(c, p));
            }
        }
        {   // x1 = LN1(x + drop(ctx Wo^T + bo)) : Linear, dropout, residual and LayerNorm in one kernel
            TlinP t;
            t.X = L.ctx; t.ldx = E; t.M = RB * S; t.W = WB(n, lp.sa.ow); t.ldw = E; t.bias = w + lp.sa.ob;
            t.Y = L.r1; t.ldy = E; t.N = E; t.K = E; t.drop = dkey(e, a, n.role, l, 1); t.drop_ld = E;
            t.res = x_in; t.ldres = E; t.res_rows = (share0 && l == 0) ? (long)B * S : RB * S; t.y_rows = keep_rows;
            t.ln_g = w + lp.n1w; t.ln_b = w + lp.n1b; t.ln_y = L.x1; t.ln_stats = L.st1; t.x_bf16 = bst;
            t.res_bf16 = xst && l > 0; t.ln_y_bf16 = xst; t.y_bf16 = rst;
            if (bst) maybe_fp8(e, n, t, lp.sa.ow);
            if (bst) TLIN_MUST(t);
            else TLIN_OR(t, {
                GG_TRY(lin_fwd(c, L.ctx, E, w + lp.sa.ow, E, w + lp.sa.ob, L.r1, E, (int)(RB * S), E, E));
                KL(k_add_layernorm_fwd(x_in, RB * S, L.r1, w + lp.n1w, w + lp.n1b, L.x1, L.st1, RB * S, E, dkey(e, a, n.role, l, 1), c.st));
            });
        }
        bool ffn_done = false;
        if (bst && !xst && !e->fp8_fwd && e->ffn_on && g_lab.ffn_fused) {   // x2 = LN2(x1 + drop(W2 drop(relu(W1 x1 + b1)) + b2)) in one launch: the hidden tile stays on chip
            FfnP f;
            f.X = L.x1; f.M = RB * S; f.E = E; f.F = F;
            f.W1 = WB(n, lp.l1w); f.b1 = w + lp.l1b; f.W2T = WTB(n, lp.l2w); f.b2 = w + lp.l2b;
            f.Hs = L.h; f.R2 = L.r2; f.keep_rows = keep_rows;
            f.ln_g = w + lp.n2w; f.ln_b = w + lp.n2b; f.Y = L.x2; f.stats = L.st2;
            f.drop1 = dkey(e, a, n.role, l, 2); f.drop2 = dkey(e, a, n.role, l, 3);
            if (g_lab.ffn_fused_supported(f)) {
                const double tokd = (double)RB * S, kept = keep_rows < 0 ? tokd : std::min<double>(tokd, (double)keep_rows);
                ProfScope ps(c, "ffn_fused_kernel", 2.0 * tokd * 2.0 * E * F, tokd * (4.0 * E + 4.0 * E) + kept * (2.0 * F + 4.0 * E) + 4.0 * E * F);
                KL(g_lab.ffn_fused(f, c.st));
                ffn_done = true;
            }
        }
        // rows [r0, r0 + Mt) of the block through the two Linear launches (the whole block, or the rows the streamed kernel leaves)
        auto ffn_two_launches = [&](long r0, long Mt) -> int {
            auto at = [](const void* ptr, long elems, bool bf16) { return static_cast<const char*>(ptr) + elems * (bf16 ? 2 : 4); };
            auto rows_key = [&](DropKey k, long ld) {       // the stream words of rows r0.. with row indices local to the launch (kernels.h DropKey::post)
                k.post = (uint32_t)(((uint64_t)r0 * (uint64_t)ld) >> 1) * 0x9E3779B1u;
                return k;
            };
            const long keep_t = keep_rows < 0 ? -1 : std::max<long>(0, std::min<long>(Mt, keep_rows - r0));
            const bool x2b = xst && l + 1 < e->nl;
            {   // h = drop(relu(x1 W1^T + b1))
                TlinP t;
                t.X = at(L.x1, r0 * E, xst); t.ldx = E; t.M = Mt; t.W = WB(n, lp.l1w); t.ldw = E; t.bias = w + lp.l1b;
                t.Y = const_cast<char*>(at(L.h, r0 * F, bst)); t.ldy = F; t.N = F; t.K = E; t.act_relu = 1;
                t.drop = rows_key(dkey(e, a, n.role, l, 2), F); t.drop_ld = F;
                t.y_bf16 = bst; t.x_bf16 = xst;
                if (bst) maybe_fp8(e, n, t, lp.l1w);
                if (bst) TLIN_MUST(t);
                else TLIN_OR(t, {
                    GG_TRY(lin_fwd(c, L.x1, E, w + lp.l1w, E, w + lp.l1b, L.h, F, (int)Mt, F, E, ACT_LRELU, 0.f));
                    if (drop > 0.f) KL(k_dropout(L.h, Mt * F, dkey(e, a, n.role, l, 2), c.st));
                });
            }
            {   // x2 = LN2(x1 + drop(h W2^T + b2))
                TlinP t;
                t.X = at(L.h, r0 * F, bst); t.ldx = F; t.M = Mt; t.W = WB(n, lp.l2w); t.ldw = F; t.bias = w + lp.l2b;
                t.Y = const_cast<char*>(at(L.r2, r0 * E, rst)); t.ldy = E; t.N = E; t.K = F;
                t.drop = rows_key(dkey(e, a, n.role, l, 3), E); t.drop_ld = E;
                t.res = reinterpret_cast<const float*>(at(L.x1, r0 * E, xst)); t.ldres = E; t.res_rows = Mt; t.y_rows = keep_t;
                t.ln_g = w + lp.n2w; t.ln_b = w + lp.n2b;
                t.ln_y = reinterpret_cast<float*>(const_cast<char*>(at(L.x2, r0 * E, x2b))); t.ln_stats = L.st2 + 2 * r0; t.x_bf16 = bst;
                t.res_bf16 = xst; t.ln_y_bf16 = x2b; t.y_bf16 = rst;
                if (bst) maybe_fp8(e, n, t, lp.l2w);
                if (bst) TLIN_MUST(t);
                else TLIN_OR(t, {
                    GG_TRY(lin_fwd(c, L.h, F, w + lp.l2w, F, w + lp.l2b, L.r2, E, (int)Mt, E, F));
                    KL(k_add_layernorm_fwd(L.x1, Mt, L.r2, w + lp.n2w, w + lp.n2b, L.x2, L.st2, Mt, E, dkey(e, a, n.role, l, 3), c.st));
                });
            }
            return 0;
        };
        if (!ffn_done && xst && rst && !e->fp8_fwd && e->ffn2_on && g_lab.ffn2 && E == 256 && F == 512) {   // the streamed form (enc.hip): bf16 x1 in, weights as MFMA fragments through an LDS ring
            // The streamed kernel is persistent, one workgroup per compute unit: M = R * B * 257 tokens is a whole number of passes of
            // the grid plus R * B rows (the CLS token), and those rows would cost every launch one more pass on a nearly empty chip
            // (measured: 145 us at 3 * 65 536 rows, 175 us at 3 * 65 792).  It takes the whole passes; the few rows left over go through
            // the two Linear launches (rows are independent in this block).
            const long Mtot = RB * S, sweep = g_lab.ffn2_sweep_tokens(e->ffn2_on - 1);
            long Mmain = Mtot / sweep * sweep;
            if (Mmain == 0 || (Mtot - Mmain) * 8 > sweep) Mmain = Mtot;            // no full pass, or a left-over worth a pass of its own
            Ffn2P f;
            f.X = L.x1; f.M = Mmain; f.Wf = n.wfrag + (size_t)l * enc_frag_bytes(1);
            f.b1 = w + lp.l1b; f.b2 = w + lp.l2b; f.ln_g = w + lp.n2w; f.ln_b = w + lp.n2b;
            f.Hs = L.h; f.R2 = L.r2; f.r2_bf16 = 1; f.stats = L.st2; f.Y = L.x2; f.y_bf16 = l + 1 < e->nl; f.keep_rows = keep_rows;
            f.drop1 = dkey(e, a, n.role, l, 2); f.drop2 = dkey(e, a, n.role, l, 3);
            if (g_lab.ffn2_supported(f) && (f.drop1.p > 0.f) == (f.drop2.p > 0.f)) {
                const double tokd = (double)Mmain, kept = keep_rows < 0 ? tokd : std::min<double>(tokd, (double)keep_rows);
                {
                    ProfScope ps(c, "ffn2_kernel", 2.0 * tokd * 2.0 * E * F, tokd * (2.0 * E + (f.y_bf16 ? 2.0 : 4.0) * E) + kept * (2.0 * F + 2.0 * E + 8.0) + 4.0 * E * F);
                    KL(g_lab.ffn2(f, c.st, e->ffn2_on - 1));
                }
                if (Mmain < Mtot) GG_TRY(ffn_two_launches(Mmain, Mtot - Mmain));
                ffn_done = true;
            }
        }
        if (!ffn_done) GG_TRY(ffn_two_launches(0, RB * S));
        x_in = L.x2;
    }
    if (!e->xattn) {    // conditioning vector = the encoder's CLS row (conditional_gan_film.py:150)
        KL(k_copy_rows_strided_bcast(a.c, E, x_in, (long)S * E, RB, RB, E, c.st));
        return 0;
    }
    // T2I: query = text CLS embedding, keys = values = encoder output (R:218)
    GG_TRY(lin_fwd(c, tok, (long)T * E, w + n.t2i.inw, E, w + n.t2i.inb, a.t2i_q, E, (int)RB, E, E));
    a.sqx2 = e->sqx_on && sqx_stream_supported(S, E, nh);
    a.sqx = !a.sqx2 && e->sqx_on && sqx_supported(S, E, nh);
    if (a.sqx2) {
        // K / V projections folded into the query side (sqattn.hip): qt_h = Wk_h^T q_h as a per-head batched GEMM,
        // ONE sweep over the encoder output (scores, online softmax, xbar_h = sum_s p x_s), ctx_h = Wv_h xbar_h + bv
        GemmP p;
        p.M = (int)RB; p.N = E; p.K = dh; p.batch = nh; p.batch_inner = 1;
        p.A = a.t2i_q; p.lda = E; p.layA = LAY_KC; p.sAo = dh;
        p.B = w + n.t2i.inw + (long)E * E; p.ldb = E; p.layB = LAY_KS; p.sBo = (long)dh * E;
        p.C = a.t2i_qt; p.ldc = (long)nh * E; p.sCo = E;
        GG_TRY(run_gemm(c, p));
        KL(sqx_stream_fwd(a.t2i_qt, x_in, a.mask, B, a.t2i_P, a.t2i_xbar, (int)RB, S, E, nh, c.st));
        KL(k_copy_rows_bcast(a.t2i_ctx, w + n.t2i.inb + 2 * E, RB, 1, E, c.st));
        GemmP v;
        v.M = (int)RB; v.N = dh; v.K = E; v.batch = nh; v.batch_inner = 1; v.accumulate = 1;
        v.A = a.t2i_xbar; v.lda = (long)nh * E; v.layA = LAY_KC; v.sAo = E;
        v.B = w + n.t2i.inw + 2L * E * E; v.ldb = E; v.layB = LAY_KC; v.sBo = (long)dh * E;
        v.C = a.t2i_ctx; v.ldc = E; v.sCo = dh;
        GG_TRY(run_gemm(c, v));
    } else if (a.sqx) {
        // K / V projections folded into the query side: one fused per-sample kernel streams the encoder output
        KL(sqx_attn_fwd(a.t2i_q, x_in, w + n.t2i.inw, w + n.t2i.inb, a.mask, B, a.t2i_P, a.t2i_xbar, a.t2i_ctx, (int)RB, S, E, nh, c.st));
    } else {
        {
            TlinP t;
            t.X = x_in; t.ldx = E; t.M = RB * S; t.W = WB(n, n.t2i.inw + (long)E * E); t.ldw = E; t.bias = w + n.t2i.inb + E;
            t.Y = a.t2i_kv; t.ldy = 2 * E; t.N = 2 * E; t.K = E;
            TLIN_OR(t, GG_TRY(lin_fwd(c, x_in, E, w + n.t2i.inw + (long)E * E, E, w + n.t2i.inb + E, a.t2i_kv, 2 * E, (int)(RB * S), 2 * E, E)));
        }
        KL(k_sq_attn_fwd(a.t2i_q, a.t2i_kv, a.mask, B, a.t2i_P, a.t2i_ctx, (int)RB, S, E, nh, c.st));
    }
    GG_TRY(lin_fwd(c, a.t2i_ctx, E, w + n.t2i.ow, E, w + n.t2i.ob, a.t2i_out, E, (int)RB, E, E));
    // I2T: query = that vector, keys = values = encoded text tokens (R:220)
    // One text token (the headline shape): the softmax over a single key is 1 whatever the query, so the attention output is
    // that token's value projection - exactly, gradients included (d softmax = 0: the query projection, the key projection
    // and everything upstream of the query receive exactly zero from this block).  The query projection, the score kernel
    // and, in the backward, their four gradient products are skipped; a padded single token still yields NaN as torch's
    // softmax over one masked key does (k_sum2_nan_rows below).
    static const bool no_t1 = getenv("GG_NO_I2T_T1") != nullptr;
    const bool t1 = T == 1 && !a.i2t_shared && !no_t1;
    a.i2t_t1 = t1;
    if (!t1) GG_TRY(lin_fwd(c, a.t2i_out, E, w + n.i2t.inw, E, w + n.i2t.inb, a.i2t_q, E, (int)RB, E, E));
    if (t1) {        // only V is needed: rows [2E, 3E) of in_proj, into the V half of i2t_kv
        GG_TRY(lin_fwd(c, tok, E, w + n.i2t.inw + 2L * E * E, E, w + n.i2t.inb + 2 * E, a.i2t_kv + E, 2 * E, (int)RB, E, E));
        GG_TRY(lin_fwd(c, a.i2t_kv + E, 2 * E, w + n.i2t.ow, E, w + n.i2t.ob, a.i2t_out, E, (int)RB, E, E));
        KL(k_sum2_nan_rows(a.c, a.t2i_out, a.i2t_out, in->text_pad, RB, B, E, c.st));
        return 0;
    }
    if (a.i2t_shared) {
        TlinP t;
        t.X = a.tok; t.ldx = E; t.M = (long)B * T; t.W = WB(n, n.i2t.inw + (long)E * E); t.ldw = E; t.bias = w + n.i2t.inb + E;
        t.Y = a.i2t_kv; t.ldy = 2 * E; t.N = 2 * E; t.K = E;
        TLIN_OR(t, GG_TRY(lin_fwd(c, a.tok, E, w + n.i2t.inw + (long)E * E, E, w + n.i2t.inb + E, a.i2t_kv, 2 * E, B * T, 2 * E, E)));
    } else {
        GG_TRY(lin_fwd(c, tok, E, w + n.i2t.inw + (long)E * E, E, w + n.i2t.inb + E, a.i2t_kv, 2 * E, (int)(RB * T), 2 * E, E));
    }
    KL(k_sq_attn_fwd(a.i2t_q, a.i2t_kv, in->text_pad, B, a.i2t_P, a.i2t_ctx, (int)RB, T, E, nh, c.st, a.i2t_shared ? B : 0));
    GG_TRY(lin_fwd(c, a.i2t_ctx, E, w + n.i2t.ow, E, w + n.i2t.ob, a.i2t_out, E, (int)RB, E, E));
    KL(k_copy(a.c, a.t2i_out, RB * E, c.st));
    KL(k_axpy(a.c, a.i2t_out, 1.f, RB * E, c.st));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// conditioning stack backward   [numpy_oracle.cond_bwd]
// dc: [Rb*B, E] gradient w.r.t. the conditioning vector of the first Rb replicas
// ------------------------------------------------------------------------------------------------
// ---- second stream for the encoder weight gradients ---------------------------------------------------------------
// dW = dY^T X reads two buffers the main chain has just produced and writes only the gradient buffer, which nothing on
// the main stream touches until the optimiser: it overlaps with the data-gradient kernels that follow (their tails
// and bubbles get filled).  side_begin: the side stream waits for everything enqueued so far on the caller's stream;
// side_end(slot): marks the launch; side_wait(slot): the caller's stream waits for it - called before the kernel that
// OVERWRITES a buffer the pending launch reads (slot 0: sdres, 1: sdh, 2: sdqkv; 3: head / gradient-penalty parameter
// gradients, whose operands live until the next iteration; 4: the second bf16 image in sdres, LN1's branch gradient when the
// fused += / LayerNorm-backward kernel writes it) and at the end of the backward.
// Streams the ENGINE creates (a C host that binds none with gg_bind_streams): default priority, or with GG_SIDE_PRIO=low|high in the
// environment the device's lowest / highest one (A/B runs measured 38.4 / 39.0 ms per step against 38.2 at the default, DESIGN.md
// section 3).  The Python host binds two torch streams instead (engine.py: default priority; GG_SIDE_PRIO=high asks torch for its
// high-priority pool, any other value means the default) and this function is then never reached.
bool create_side_stream(hipStream_t* s) {
    const char* pr = getenv("GG_SIDE_PRIO");
    int least = 0, greatest = 0;
    if (pr && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, pr[0] == 'h' ? greatest : least) == hipSuccess;
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess;
}
bool side_begin(Ctx& c, Ctx& cs) {
    gg_engine* e = c.e;
    static const bool env_off = getenv("GG_NO_SIDE_WGRAD") != nullptr;
    cs = c;
    if (!e->side_on || env_off) return false;
    if (!e->ev_ready) {
        if (!e->side && !create_side_stream(&e->side)) return false;
        bool ok = hipEventCreateWithFlags(&e->ev_ready, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 5; ++i) ok = ok && hipEventCreateWithFlags(&e->ev_done[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) { e->side_on = 0; return false; }
    }
    if (hipEventRecord(e->ev_ready, c.st) != hipSuccess || hipStreamWaitEvent(e->side, e->ev_ready, 0) != hipSuccess) return false;
    cs.st = e->side;
    return true;
}
int side_end(Ctx& c, bool forked, int slot) {
    if (!forked) return 0;
    GG_CHECK_HIP(hipEventRecord(c.e->ev_done[slot], c.e->side));
    c.e->side_pending[slot] = true;
    return 0;
}
int side_wait(Ctx& c, int slot) {
    if (!c.e->side_pending[slot]) return 0;
    GG_CHECK_HIP(hipStreamWaitEvent(c.st, c.e->ev_done[slot], 0));
    c.e->side_pending[slot] = false;
    return 0;
}

// Stages (data-parallel hosts all-reduce a stage's gradient range while the next stage runs; gg_cond_stage_range): 0 = the cross-attention
// blocks (or the CLS-row seed), 1 .. nl = encoder layers nl-1 .. 0, nl + 1 = replica fold, CLS token, patch encoder, FiLM, text encoder.
// [s0, s1] = the stages this call runs; everything a later stage needs lives in the engine's scratch buffers and flags.
int cond_backward(Ctx& c, Net& n, const gg_cond* in, CondActs& a, const float* dc, int Rb, int s0 = 0, int s1 = 1 << 20) {
    gg_engine* e = c.e;
    const int B = a.B, P = a.P, T = a.T, S = P + 1, E = e->E, F = e->F, nh = e->nh, dh = e->dh;
    const int s_tail = e->nl + 1;
    auto in_stage = [&](int k) { return k >= s0 && k <= s1; };
    const int Dt = e->Dt, Dp = e->Dp;
    const long RB = (long)Rb * B;
    const float* w = n.w;
    float* g = n.g;
    const float* tok = a.R > 1 ? a.tokrep : a.tok;
    const float* enc = a.L[e->nl - 1].x2;
    const float drop = a.drop;
    const float ks = drop > 0.f ? 1.f / (1.f - drop) : 1.f;
    const int bst = a.bst ? 1 : 0;

    if (e->no_cond) {        // nothing upstream of the (zero) conditioning vector; join the head's side-stream leaves
        if (in_stage(s_tail))
            for (int i = 0; i < 5; ++i) GG_TRY(side_wait(c, i));
        return 0;
    }
    const bool i2t_sh = e->xattn && !a.i2t_t1 && a.i2t_shared && sq_attn_shared_ok(T, E, nh, Rb);
    if (!in_stage(0)) {
    } else if (!e->xattn) {    // the conditioning vector was the encoder's CLS row: its gradient is the only non-zero row per sample
        KL(k_fill(e->sdx, RB * S * E, 0.f, c.st));
        KL(k_copy_rows_strided_bcast(e->sdx, (long)S * E, dc, E, RB, RB, E, c.st));
    } else {
    // ---- I2T backward: t = out_proj(ctx); scores over text tokens; q from p -----------------------
    if (a.i2t_t1) {
        // ctx = V(token): d(ctx) goes straight to the V half; dq = dk = 0 exactly, so p's gradient is dc alone
        {
            Ctx cs = c;
            const bool fk = side_begin(c, cs);
            GG_TRY(lin_bwd_weight(cs, dc, E, a.i2t_kv + E, 2 * E, g + n.i2t.ow, E, (int)RB, E, E));
            GG_TRY(k_colsum(dc, RB, E, E, g + n.i2t.ob, cs.st)); e->launches++;
            GG_TRY(side_end(c, fk, 3));
        }
        GG_TRY(lin_bwd_data(c, dc, E, w + n.i2t.ow, E, e->s_dkv2 + E, 2 * E, (int)RB, E, E));      // d(V rows), ld 2E
        KL(k_copy(e->s_dp, dc, RB * E, c.st));                                                      // c = t + p
        {
            Ctx cs = c;
            const bool fk = side_begin(c, cs);
            GG_TRY(lin_bwd_weight(cs, e->s_dkv2 + E, 2 * E, tok, E, g + n.i2t.inw + 2L * E * E, E, (int)RB, E, E));
            GG_TRY(k_colsum(e->s_dkv2 + E, RB, E, 2 * E, g + n.i2t.inb + 2 * E, cs.st)); e->launches++;
            GG_TRY(side_end(c, fk, 3));
        }
        GG_TRY(lin_bwd_data(c, e->s_dkv2 + E, 2 * E, w + n.i2t.inw + 2L * E * E, E, e->s_dtokrep, E, (int)RB, E, E));
    } else {
    {   // parameter-gradient leaves: side stream (see side_begin)
        Ctx cs = c;
        const bool fk = side_begin(c, cs);
        GG_TRY(lin_bwd_weight(cs, dc, E, a.i2t_ctx, E, g + n.i2t.ow, E, (int)RB, E, E));
        GG_TRY(k_colsum(dc, RB, E, E, g + n.i2t.ob, cs.st)); e->launches++;
        GG_TRY(side_end(c, fk, 3));
    }
    GG_TRY(lin_bwd_data(c, dc, E, w + n.i2t.ow, E, e->s_tmpE, E, (int)RB, E, E));
    GG_REQUIRE(i2t_sh || !a.i2t_shared, "shared I2T keys: backward replica count not supported");
    if (i2t_sh) KL(k_sq_attn_bwd_shared(e->s_tmpE, a.i2t_q, a.i2t_kv, a.i2t_P, e->s_dq, e->s_dkv2, B, Rb, T, E, nh, c.st));
    else KL(k_sq_attn_bwd(e->s_tmpE, a.i2t_q, a.i2t_kv, a.i2t_P, e->s_dq, e->s_dkv2, (int)RB, T, E, nh, c.st));
    {   // parameter-gradient leaves: side stream (see side_begin)
        Ctx cs = c;
        const bool fk = side_begin(c, cs);
        GG_TRY(lin_bwd_weight(cs, e->s_dq, E, a.t2i_out, E, g + n.i2t.inw, E, (int)RB, E, E));
        GG_TRY(k_colsum(e->s_dq, RB, E, E, g + n.i2t.inb, cs.st)); e->launches++;
        GG_TRY(side_end(c, fk, 3));
    }
    KL(k_copy(e->s_dp, dc, RB * E, c.st));                                        // c = t + p
    GG_TRY(lin_bwd_data(c, e->s_dq, E, w + n.i2t.inw, E, e->s_dp, E, (int)RB, E, E, 1));
    {   // parameter-gradient leaves: side stream (see side_begin)
        Ctx cs = c;
        const bool fk = side_begin(c, cs);
        // shared keys: s_dkv2 is already summed over the replicas, [B*T, 2E] against the un-replicated tokens
        const long Mkv = i2t_sh ? (long)B * T : RB * T;
        GG_TRY(lin_bwd_weight(cs, e->s_dkv2, 2 * E, i2t_sh ? a.tok : tok, E, g + n.i2t.inw + (long)E * E, E, (int)Mkv, 2 * E, E));
        GG_TRY(k_colsum(e->s_dkv2, Mkv, 2 * E, 2 * E, g + n.i2t.inb + E, cs.st)); e->launches++;
        GG_TRY(side_end(c, fk, 3));
    }
    if (i2t_sh) {   // the token gradient of all replicas, [B*T, E]; the T2I query rows are added at the fold below
        TlinP t;
        t.X = e->s_dkv2; t.ldx = 2 * E; t.M = (long)B * T; t.W = WTB(n, n.i2t.inw + E); t.ldw = 3 * E;     // columns E..3E of in_proj^T
        t.Y = e->s_dtokrep; t.ldy = E; t.N = E; t.K = 2 * E;
        TLIN_OR(t, GG_TRY(lin_bwd_data(c, e->s_dkv2, 2 * E, w + n.i2t.inw + (long)E * E, E, e->s_dtokrep, E, B * T, 2 * E, E)));
    } else {
        GG_TRY(lin_bwd_data(c, e->s_dkv2, 2 * E, w + n.i2t.inw + (long)E * E, E, e->s_dtokrep, E, (int)(RB * T), 2 * E, E));
    }
    }   // !i2t_t1
    // ---- T2I backward ---------------------------------------------------------------------------------
    {   // parameter-gradient leaves: side stream (see side_begin)
        Ctx cs = c;
        const bool fk = side_begin(c, cs);
        GG_TRY(lin_bwd_weight(cs, e->s_dp, E, a.t2i_ctx, E, g + n.t2i.ow, E, (int)RB, E, E));
        GG_TRY(k_colsum(e->s_dp, RB, E, E, g + n.t2i.ob, cs.st)); e->launches++;
        GG_TRY(side_end(c, fk, 3));
    }
    GG_TRY(lin_bwd_data(c, e->s_dp, E, w + n.t2i.ow, E, e->s_tmpE, E, (int)RB, E, E));
    if (a.sqx || a.sqx2) {
        if (a.sqx2) {
            // dxbar_h = Wv_h^T dctx_h ; one sweep over the encoder output (dx written, dqt accumulated) ; dq_h = Wk_h dqt_h
            GemmP p;
            p.M = (int)RB; p.N = E; p.K = dh; p.batch = nh; p.batch_inner = 1;
            p.A = e->s_tmpE; p.lda = E; p.layA = LAY_KC; p.sAo = dh;
            p.B = w + n.t2i.inw + 2L * E * E; p.ldb = E; p.layB = LAY_KS; p.sBo = (long)dh * E;
            p.C = e->s_dxbar; p.ldc = (long)nh * E; p.sCo = E;
            GG_TRY(run_gemm(c, p));
            KL(sqx_stream_bwd(e->s_dxbar, a.t2i_qt, a.t2i_xbar, enc, a.t2i_P, e->sdx, e->s_dqt, (int)RB, S, E, nh, c.st));
            GemmP v;
            v.M = (int)RB; v.N = dh; v.K = E; v.batch = nh; v.batch_inner = 1;
            v.A = e->s_dqt; v.lda = (long)nh * E; v.layA = LAY_KC; v.sAo = E;
            v.B = w + n.t2i.inw + (long)E * E; v.ldb = E; v.layB = LAY_KC; v.sBo = (long)dh * E;
            v.C = e->s_dq2; v.ldc = E; v.sCo = dh;
            GG_TRY(run_gemm(c, v));
        } else {
            KL(sqx_attn_bwd(e->s_tmpE, a.t2i_q, enc, w + n.t2i.inw, a.t2i_P, e->sdx, e->s_dq2, e->s_dqt, (int)RB, S, E, nh, c.st));
        }
        {   // dWk_h += q_h (x) dqt_h  and  dWv_h += dctx_h (x) xbar_h, summed over the batch (per-head small GEMMs); leaves
            Ctx cs = c;
            const bool fk = side_begin(c, cs);
            GemmP p;
            p.M = dh; p.N = E; p.K = (int)RB; p.layA = LAY_KS; p.layB = LAY_KS; p.lda = E; p.ldb = (long)nh * E; p.ldc = E;
            p.batch = nh; p.batch_inner = 1; p.sAo = dh; p.sBo = E; p.sCo = (long)dh * E; p.accumulate = 1;
            p.A = a.t2i_q; p.B = e->s_dqt; p.C = g + n.t2i.inw + (long)E * E;
            GG_TRY(run_gemm(cs, p));
            p.A = e->s_tmpE; p.B = a.t2i_xbar; p.C = g + n.t2i.inw + 2L * E * E;
            GG_TRY(run_gemm(cs, p));
            GG_TRY(k_colsum(e->s_tmpE, RB, E, E, g + n.t2i.inb + 2 * E, cs.st)); e->launches++;         // d(bv) = sum dctx ; d(bk) == 0
            GG_TRY(lin_bwd_weight(cs, e->s_dq2, E, tok, (long)T * E, g + n.t2i.inw, E, (int)RB, E, E));
            GG_TRY(k_colsum(e->s_dq2, RB, E, E, g + n.t2i.inb, cs.st)); e->launches++;
            GG_TRY(side_end(c, fk, 3));
        }
        if (i2t_sh) GG_TRY(lin_bwd_data(c, e->s_dq2, E, w + n.t2i.inw, E, e->s_dtok0, E, (int)RB, E, E));
        else GG_TRY(lin_bwd_data(c, e->s_dq2, E, w + n.t2i.inw, E, e->s_dtokrep, (long)T * E, (int)RB, E, E, 1));
    } else {
        KL(k_sq_attn_bwd(e->s_tmpE, a.t2i_q, a.t2i_kv, a.t2i_P, e->s_dq2, e->s_dkv, (int)RB, S, E, nh, c.st));
        GG_TRY(lin_bwd_weight(c, e->s_dq2, E, tok, (long)T * E, g + n.t2i.inw, E, (int)RB, E, E));
        KL(k_colsum(e->s_dq2, RB, E, E, g + n.t2i.inb, c.st));
        if (i2t_sh) GG_TRY(lin_bwd_data(c, e->s_dq2, E, w + n.t2i.inw, E, e->s_dtok0, E, (int)RB, E, E));
        else GG_TRY(lin_bwd_data(c, e->s_dq2, E, w + n.t2i.inw, E, e->s_dtokrep, (long)T * E, (int)RB, E, E, 1));
        GG_TRY(lin_bwd_weight(c, e->s_dkv, 2 * E, enc, E, g + n.t2i.inw + (long)E * E, E, (int)(RB * S), 2 * E, E));
        KL(k_colsum(e->s_dkv, RB * S, 2 * E, 2 * E, g + n.t2i.inb + E, c.st));
        {   // denc = dkv Wkv : reduction over the 2E projected features, W^T = columns E..3E of in_proj^T
            TlinP t;
            t.X = e->s_dkv; t.ldx = 2 * E; t.M = RB * S; t.W = WTB(n, n.t2i.inw + E); t.ldw = 3 * E;
            t.Y = e->sdx; t.ldy = E; t.N = E; t.K = 2 * E;
            TLIN_OR(t, GG_TRY(lin_bwd_data(c, e->s_dkv, 2 * E, w + n.t2i.inw + (long)E * E, E, e->sdx, E, (int)(RB * S), 2 * E, E)));
        }
    }
    if (i2t_sh) {
        // With shared text keys the token gradient is complete here ([B*T, E] from the I2T projection + the T2I query rows):
        // the text encoder's parameter gradients are leaves - on the side stream, beside the encoder layers' backward
        KL(k_fold_rows_add(e->s_dtokrep, (long)T * E, e->s_dtok0, B, Rb, E, c.st));
        Ctx cs = c;
        const bool fk = side_begin(c, cs);
        GG_TRY(lin_bwd_weight(cs, e->s_dtokrep, E, in->text, Dt, g + n.te_w, Dt, B * T, E, Dt));
        GG_TRY(k_colsum(e->s_dtokrep, (long)B * T, E, E, g + n.te_b, cs.st)); e->launches++;
        GG_TRY(side_end(c, fk, 3));
    }
    }   // xattn
    // ---- encoder layers, last to first ---------------------------------------------------------------
    float* dx = e->sdx;          // gradient w.r.t. the layer output (in), w.r.t. its input (out)
    for (int l = e->nl - 1; l >= 0; --l) {
        if (!in_stage(e->nl - l)) continue;
        LayerActs& L = a.L[l];
        const LayerP& lp = n.layer[l];
        const bool shared = a.share0 && l == 0;          // layer-0 input and QKV projection exist once for all replicas
        const float* x_in = l > 0 ? a.L[l - 1].x2 : (a.R > 1 && !a.share0 ? a.xrep : a.x0);
        // LN2
        GG_TRY(side_wait(c, 0));
        {   // reads dy and the saved pre-LN sum (fp32), writes dr (fp32) and the masked branch gradient (bf16 when stored so)
            ProfScope ps(c, "ln_bwd_v4_k", 16.0 * RB * S * E, (double)RB * S * (E * (12.0 + (bst ? 2.0 : 4.0)) + 8.0));
            KL(k_layernorm_bwd(dx, L.r2, L.st2, w + lp.n2w, e->sdr, e->sdres, g + lp.n2w, g + lp.n2b, g + lp.l2b, RB * S, E,
                               dkey(e, a, n.role, l, 3), c.st, bst | (a.rst ? 2 : 0)));
        }
        // FFN: f = h W2^T + b2 ; h = drop(relu(x1 W1^T + b1))   (db2 = column sums of df: fused above)
        {
            Ctx cs = c;
            const bool fk = side_begin(c, cs);
            GG_TRY(lin_bwd_weight(cs, e->sdres, E, L.h, F, g + lp.l2w, F, (int)(RB * S), E, F, bst, bst));
            GG_TRY(side_end(c, fk, 0));
        }
        bool fusedb = false;
        if (bst && a.xst && a.rst && e->encb_on && g_lab.enc_bwd && e->lnb_on && E == 256 && F == 512 && !e->x3) {
            // MASK -> dx1 += -> LayerNorm1 backward -> dctx in one streamed launch (enc.hip): dr2 in sdr, dres2 in sdres; dr1 -> dx, dh -> sdh,
            // dres1 -> the second bf16 image of sdres, dctx -> sdctx
            EncBwdP q;
            q.dx = dx; q.dr2 = e->sdr; q.M = RB * S; q.Wf = n.wfragb + (size_t)l * encb_frag_bytes(1);
            q.dres2 = e->sdres; q.h = L.h; q.r1 = L.r1; q.st1 = L.st1; q.g1 = w + lp.n1w;
            q.dh = e->sdh; q.dres1 = reinterpret_cast<__bf16*>(e->sdres) + RB * S * E; q.dctx = e->sdctx;
            q.dg1 = g + lp.n1w; q.db1 = g + lp.n1b; q.dbias1 = g + lp.sa.ob;
            q.drop1 = dkey(e, a, n.role, l, 1); q.gate_scale = ks;
            if (g_lab.enc_bwd_supported(q)) {
                GG_TRY(side_wait(c, 1));
                GG_TRY(side_wait(c, 4));
                {
                    const double tokd = (double)RB * S;
                    ProfScope ps(c, "encb_kernel", 2.0 * tokd * (2.0 * E * F + (double)E * E), tokd * (4.0 * E * 2 + 2.0 * E * 4 + 2.0 * F * 2) + 2.0 * (2.0 * E * F + E * E));
                    KL(g_lab.enc_bwd(q, c.st));
                }
                fusedb = true;
                Ctx cs = c;
                const bool fk = side_begin(c, cs);
                GG_TRY(lin_bwd_weight(cs, e->sdh, F, L.x1, E, g + lp.l1w, E, (int)(RB * S), F, E, bst, a.xst, nullptr, g + lp.l1b));
                GG_TRY(side_end(c, fk, 1));
                const bool fk2 = side_begin(c, cs);
                GG_TRY(lin_bwd_weight(cs, reinterpret_cast<const float*>(q.dres1), E, L.ctx, E, g + lp.sa.ow, E, (int)(RB * S), E, E, bst, bst));
                GG_TRY(side_end(c, fk2, 4));
            }
        }
        if (!fusedb) {
        GG_TRY(side_wait(c, 1));
        {   // dhpre = (df W2) * [h > 0] / (1-p) : the stored post-dropout h gates both ReLU and the kept-mask
            TlinP t;
            t.X = e->sdres; t.ldx = E; t.M = RB * S; t.W = WTB(n, lp.l2w); t.ldw = E;
            t.Y = e->sdh; t.ldy = F; t.N = F; t.K = E; t.mask_ref = L.h; t.ldref = F; t.mask_scale = ks;
            t.x_bf16 = bst; t.y_bf16 = bst; t.mask_bf16 = bst;
            if (bst) TLIN_MUST(t);
            else TLIN_OR(t, {
                GG_TRY(lin_bwd_data(c, e->sdres, E, w + lp.l2w, F, e->sdh, F, (int)(RB * S), E, F));
                KL(k_act_bwd(e->sdh, L.h, RB * S * F, 0.f, ks, c.st));
            });
        }
        {
            Ctx cs = c;
            const bool fk = side_begin(c, cs);
            GG_TRY(lin_bwd_weight(cs, e->sdh, F, L.x1, E, g + lp.l1w, E, (int)(RB * S), F, E, bst, a.xst, nullptr, g + lp.l1b));
            GG_TRY(side_end(c, fk, 1));
        }
        // dx1 = dr2 + dhpre W1, then LN1 backward.  At the production width both run in ONE weight-stationary kernel (wst.hip EPI_LNB): dx1
        // never travels (a 4-byte write and read per element less, one launch less); LN1's branch gradient goes to the second bf16
        // image inside sdres, so the kernel does not have to wait for the weight-gradient launch that still reads LN2's.
        float* dres1 = e->sdres;
        bool lnb = false;
        {
            TlinP t;
            t.X = e->sdh; t.ldx = F; t.M = RB * S; t.W = WTB(n, lp.l1w); t.ldw = F;
            t.Y = e->sdr; t.ldy = E; t.N = E; t.K = F; t.accumulate = 1; t.x_bf16 = bst;
            if (bst && e->lnb_on) {
                TlinP u = t;
                u.res = L.r1; u.ldres = E; u.res_rows = RB * S; u.res_bf16 = a.rst; u.ln_stats = L.st1; u.ln_g = w + lp.n1w; u.ln_y = dx;
                u.lnb_dres = reinterpret_cast<__bf16*>(e->sdres) + RB * S * E;
                u.lnb_dgamma = g + lp.n1w; u.lnb_dbeta = g + lp.n1b; u.lnb_dbias = g + lp.sa.ob;
                u.drop = dkey(e, a, n.role, l, 1); u.drop_ld = E;
                if (use_tlin(e) && tlin_supported(u)) {
                    GG_TRY(side_wait(c, 4));
                    TLIN_MUST(u);
                    lnb = true;
                    dres1 = reinterpret_cast<float*>(u.lnb_dres);
                }
            }
            if (lnb) {
            } else if (bst) TLIN_MUST(t);
            else TLIN_OR(t, GG_TRY(lin_bwd_data(c, e->sdh, F, w + lp.l1w, E, e->sdr, E, (int)(RB * S), F, E, 1)));
        }
        // LN1
        if (!lnb) {
            GG_TRY(side_wait(c, 0));
            ProfScope ps(c, "ln_bwd_v4_k", 16.0 * RB * S * E, (double)RB * S * (E * (12.0 + (bst ? 2.0 : 4.0)) + 8.0));
            KL(k_layernorm_bwd(e->sdr, L.r1, L.st1, w + lp.n1w, dx, e->sdres, g + lp.n1w, g + lp.n1b, g + lp.sa.ob, RB * S, E,
                               dkey(e, a, n.role, l, 1), c.st, bst | (a.rst ? 2 : 0)));
        }
        // self attention out-proj   (d(out_proj.bias) fused above)
        {
            Ctx cs = c;
            const bool fk = side_begin(c, cs);
            GG_TRY(lin_bwd_weight(cs, dres1, E, L.ctx, E, g + lp.sa.ow, E, (int)(RB * S), E, E, bst, bst));
            GG_TRY(side_end(c, fk, lnb ? 4 : 0));
        }
        {
            TlinP t;
            t.X = dres1; t.ldx = E; t.M = RB * S; t.W = WTB(n, lp.sa.ow); t.ldw = E;
            t.Y = e->sdctx; t.ldy = E; t.N = E; t.K = E; t.x_bf16 = bst; t.y_bf16 = bst;
            if (bst) TLIN_MUST(t);
            else TLIN_OR(t, GG_TRY(lin_bwd_data(c, dres1, E, w + lp.sa.ow, E, e->sdctx, E, (int)(RB * S), E, E)));
        }
        }   // !fusedb
        const DropKey kA = dkey(e, a, n.role, l, 0);
        GG_TRY(side_wait(c, 2));
        if (a.flash) {
            {   // dQ kernel: reads QKV, O and dO (row dots), writes dQ; dK/dV kernel: reads QKV and dO, writes dK | dV
                const double tok = (double)RB * S, qtok = (shared ? (double)B : (double)RB) * S, es = bst ? 2.0 : 4.0;
                hipEvent_t mid = nullptr;
                const bool f3 = e->x3;
                const char* nm1 = f3 ? flash_attn_x3_kernel_name(1) : flash_attn_kernel_name(1, S, E, nh);
                const char* nm2 = f3 ? flash_attn_x3_kernel_name(2) : flash_attn_kernel_name(2, S, E, nh);
                const double fm = f3 ? 3.0 : 1.0;       // MFMA passes per product tile
                if (e->prof_on && (prof_wanted(e, named_class(e, nm1)) || prof_wanted(e, named_class(e, nm2)))) {
                    if (e->prof_next + 1 > e->prof_pool.size())
                        for (int i = 0; i < 4096; ++i) {
                            hipEvent_t ev;
                            GG_CHECK_HIP(hipEventCreate(&ev));
                            e->prof_pool.push_back(ev);
                        }
                    mid = e->prof_pool[e->prof_next++];
                }
                ProfScope ps(c, nm2, fm * 8.0 * tok * S * E, es * (3.0 * qtok * E + tok * E + 2.0 * tok * E) + 8.0 * tok * nh);
                ProfScope pq(c, nm1, fm * 6.0 * tok * S * E, es * (3.0 * qtok * E + 2.0 * tok * E + tok * E) + 8.0 * tok * nh);
                if (f3) KL(flash_attn_bwd_x3(L.qkv, L.ctx, e->sdctx, L.lse, e->s_delta, a.mask, B, e->sdqkv, RB, S, E, nh, kA, c.st, shared ? B : 0, 2, mid));
                else
                KL(flash_attn_bwd(L.qkv, L.ctx, e->sdctx, L.lse, e->s_delta, a.mask, B, e->sdqkv, RB, S, E, nh, kA, bst, c.st, shared ? B : 0, mid));
                if (mid) {          // both scopes opened before the pair: dq = [pq.e0, mid], dkv = [mid, ps.e1]
                    if (pq.on) { pq.r.e1 = mid; e->prof_recs.push_back(pq.r); pq.on = false; }
                    if (ps.on) ps.r.e0 = mid;
                }
            }
            c.e->launches += 2;
        } else {
            const float* Pd = L.P;
            if (drop > 0.f) {
                KL(k_dropout_copy(e->sPd, L.P, RB * nh * S * S, kA, c.st));
                Pd = e->sPd;
            }
            GemmP p;
            const int nb = (int)(RB * nh);
            {   // dPd = dctx_h V_h^T
                p = GemmP();
                p.A = e->sdctx; p.B = L.qkv + 2 * E; p.C = e->sdP; p.M = S; p.N = S; p.K = dh;
                p.lda = E; p.ldb = 3 * E; p.ldc = S; p.layA = LAY_KC; p.layB = LAY_KC;
                p.batch = nb; p.batch_inner = nh;
                p.sAo = (long)S * E; p.sAi = dh; p.sBo = (long)S * 3 * E; p.sBi = dh; p.sCo = (long)nh * S * S; p.sCi = (long)S * S;
                GG_TRY(run_gemm(c, p));
            }
            {   // dV_h = Pd^T dctx_h
                p = GemmP();
                p.A = Pd; p.B = e->sdctx; p.C = e->sdqkv + 2 * E; p.M = S; p.N = dh; p.K = S;
                p.lda = S; p.ldb = E; p.ldc = 3 * E; p.layA = LAY_KS; p.layB = LAY_KS;
                p.batch = nb; p.batch_inner = nh;
                p.sAo = (long)nh * S * S; p.sAi = (long)S * S; p.sBo = (long)S * E; p.sBi = dh; p.sCo = (long)S * 3 * E; p.sCi = dh;
                GG_TRY(run_gemm(c, p));
            }
            KL(k_softmax_bwd_rows(e->sdP, L.P, (long)nb * S, S, 1.f / sqrtf((float)dh), kA, c.st));
            {   // dQ_h = dS K_h
                p = GemmP();
                p.A = e->sdP; p.B = L.qkv + E; p.C = e->sdqkv; p.M = S; p.N = dh; p.K = S;
                p.lda = S; p.ldb = 3 * E; p.ldc = 3 * E; p.layA = LAY_KC; p.layB = LAY_KS;
                p.batch = nb; p.batch_inner = nh;
                p.sAo = (long)nh * S * S; p.sAi = (long)S * S; p.sBo = (long)S * 3 * E; p.sBi = dh; p.sCo = (long)S * 3 * E; p.sCi = dh;
                GG_TRY(run_gemm(c, p));
            }
            {   // dK_h = dS^T Q_h
                p = GemmP();
                p.A = e->sdP; p.B = L.qkv; p.C = e->sdqkv + E; p.M = S; p.N = dh; p.K = S;
                p.lda = S; p.ldb = 3 * E; p.ldc = 3 * E; p.layA = LAY_KS; p.layB = LAY_KS;
                p.batch = nb; p.batch_inner = nh;
                p.sAo = (long)nh * S * S; p.sAi = (long)S * S; p.sBo = (long)S * 3 * E; p.sBi = dh; p.sCo = (long)S * 3 * E; p.sCi = dh;
                GG_TRY(run_gemm(c, p));
            }
        }
        {
            Ctx cs = c;
            const bool fk = side_begin(c, cs);
            GG_TRY(lin_bwd_weight(cs, e->sdqkv, 3 * E, x_in, E, g + lp.sa.inw, E, (int)(RB * S), 3 * E, E, bst, a.xst && l > 0, nullptr, g + lp.sa.inb,
                                  shared ? (long)B * S : 0));
            GG_TRY(side_end(c, fk, 2));
        }
        {   // dx_in = dr1 + dqkv Win
            TlinP t;
            t.X = e->sdqkv; t.ldx = 3 * E; t.M = RB * S; t.W = WTB(n, lp.sa.inw); t.ldw = 3 * E;
            t.Y = dx; t.ldy = E; t.N = E; t.K = 3 * E; t.accumulate = 1; t.x_bf16 = bst;
            if (bst) TLIN_MUST(t);
            else TLIN_OR(t, GG_TRY(lin_bwd_data(c, e->sdqkv, 3 * E, w + lp.sa.inw, E, dx, E, (int)(RB * S), 3 * E, E, 1)));
        }
    }
    if (!in_stage(s_tail)) return 0;
    // ---- fold replicas, CLS token, patch encoder, FiLM, text encoder ----------------------------------
    const float* dtok = e->s_dtokrep;
    if (Rb > 1) {
        if (e->xattn && !i2t_sh) KL(k_fold(e->s_dtok, e->s_dtokrep, (long)B * T * E, Rb, c.st));
        if (!i2t_sh) dtok = e->s_dtok;
    }
    // one pass over dx: replicas summed, patch rows -> s_demb, CLS rows -> the head of s_dx0 (scratch from here on)
    KL(k_fold_gather(e->s_demb, e->s_dx0, dx, B, S, E, Rb, c.st));
    KL(k_cls_grad(e->s_dx0, g + n.cls, B, 1, E, c.st));
    if (e->pe_ln) {
        // Linear -> ReLU -> LayerNorm patch encoder, no FiLM: LayerNorm backward on the saved ReLU output, ReLU mask from the
        // same tensor (h > 0), then the plain weight / bias gradients against the raw patches
        KL(k_layernorm_bwd(e->s_demb, a.pe_h, a.pe_st, w + n.pe_lnw, e->s_dx0 /* scratch: dx0 is consumed */, nullptr, g + n.pe_lnw,
                           g + n.pe_lnb, nullptr, (long)B * P, E, DropKey(), c.st));
        KL(k_act_bwd(e->s_dx0, a.pe_h, (long)B * P * E, 0.f, 1.f, c.st));
        GG_TRY(lin_bwd_weight(c, e->s_dx0, E, in->patches, Dp, g + n.pe_w, Dp, B * P, E, Dp));
        KL(k_colsum(e->s_dx0, (long)B * P, E, E, g + n.pe_b, c.st));
        for (int i = 0; i < 5; ++i) GG_TRY(side_wait(c, i));
        return 0;
    }
    if (wgrad_film_ok(e, e->s_demb, E, in->patches, Dp, B * P, E, Dp, P)) {
        // dW_pe += demb^T (gamma * patches + beta): the modulation is applied while the token chunks are staged
        WgradFilm f;
        f.g = a.gb; f.b = a.gb + Dp; f.ld = 2 * Dp; f.group = P;
        GG_TRY(side_wait(c, 1));
        Ctx cs = c;
        const bool fk = side_begin(c, cs);          // beside the FiLM-gradient contraction below, which reads the same operands
        GG_TRY(lin_bwd_weight(cs, e->s_demb, E, in->patches, Dp, g + n.pe_w, Dp, B * P, E, Dp, 0, 0, &f));
        GG_TRY(side_end(c, fk, 1));
    } else {
        KL(k_film_mod(in->patches, a.gb, e->s_mod, B, P, Dp, c.st));
        GG_TRY(lin_bwd_weight(c, e->s_demb, E, e->s_mod, Dp, g + n.pe_w, Dp, B * P, E, Dp));
    }
    KL(k_colsum(e->s_demb, (long)B * P, E, E, g + n.pe_b, c.st));
    if (wgrad_film_ok(e, e->s_demb, E, in->patches, Dp, B * P, E, Dp, P) && P % 32 == 0) {
        // FiLM gradients without d(modulated input): dgamma_b = sum_e W (demb_b^T patches_b), dbeta_b = sum_e W (sum_p demb_b)
        KL(k_fill(e->s_dgb, (long)B * 2 * Dp, 0.f, c.st));
        WgradFilmGrad f;
        f.W = w + n.pe_w; f.ldw = Dp; f.dgamma = e->s_dgb; f.dbeta = e->s_dgb + Dp; f.ld = 2 * Dp; f.tokens = P;
        {
            ProfScope ps(c, "wgrad_kernel<false,false,false,true>", 2.0 * B * P * E * (double)Dp, 4.0 * B * P * ((double)E + Dp) + 4.0 * E * Dp);
            KL(wgrad(e->s_demb, E, 0, in->patches, Dp, 0, nullptr, 0, (long)B * P, E, Dp, c.st, nullptr, &f, nullptr, 0, e->x3));
        }
    } else {
        TlinP t;
        t.X = e->s_demb; t.ldx = E; t.M = (long)B * P; t.W = WTB(n, n.pe_w); t.ldw = E;
        t.Y = e->s_dmod; t.ldy = Dp; t.N = Dp; t.K = E;
        TLIN_OR(t, GG_TRY(lin_bwd_data(c, e->s_demb, E, w + n.pe_w, Dp, e->s_dmod, Dp, B * P, E, Dp)));
        KL(k_film_bwd_reduce(e->s_dmod, in->patches, e->s_dgb, B, P, Dp, c.st));
    }
    KL(k_film_act_bwd(e->s_dgb, a.gb, a.gbpre, B, Dp, c.st));
    GG_TRY(lin_bwd_weight(c, e->s_dgb, 2 * Dp, in->text, (long)T * Dt, g + n.film_w, Dt, B, 2 * Dp, Dt));
    KL(k_colsum(e->s_dgb, B, 2 * Dp, 2 * Dp, g + n.film_b, c.st));
    if (e->xattn && !i2t_sh) {
        GG_TRY(lin_bwd_weight(c, dtok, E, in->text, Dt, g + n.te_w, Dt, B * T, E, Dt));
        KL(k_colsum(dtok, (long)B * T, E, E, g + n.te_b, c.st));
    }
    for (int i = 0; i < 5; ++i) GG_TRY(side_wait(c, i));      // the gradient buffer is complete on the caller's stream again
    return 0;
}

// ------------------------------------------------------------------------------------------------
// MLP heads (R:226-231)
// ------------------------------------------------------------------------------------------------
// a1 (+)= cvec @ W1c^T + b1, act ; a2 = act(a1 W2^T + b2) ; out = a2 W3^T + b3
// `a1` must already hold v @ W1v^T (the non-conditioning part of the first layer).
inline int fwd_alone_pct() {      // GG_WST_FWD_PCT: grid share of the Linears of a forward pass that runs alone (0: the kernels' default of 91 %)
    static const int pct = getenv("GG_WST_FWD_PCT") ? atoi(getenv("GG_WST_FWD_PCT")) : 100;
    return pct;
}
int head_finish(Ctx& c, Net& n, const float* cvec, float* a1, float* a2, float* out, long ldo, int rows, int out_rows) {
    gg_engine* e = c.e;
    const int E = e->E, H = e->H;
    const float* w = n.w;
    const float slope = e->cfg.negative_slope;
    if (e->head_on && g_lab.head_fwd && e->precision == GG_PREC_BF16 && !e->x3) {       // the chain in one launch (head.hip); the generator's wide output layer stays a GEMM
        HeadP h;
        h.rows = rows; h.H = H; h.E = E; h.slope = slope;
        h.W1c = w + n.w1 + n.V; h.ldw1 = n.V + E; h.b1 = w + n.b1; h.W2 = w + n.w2; h.b2 = w + n.b2;
        h.cvec = cvec; h.a1 = a1; h.a2 = a2;
        const bool score = out && out_rows > 0 && n.OUT == 1;
        if (score) { h.w3 = w + n.w3; h.b3 = w + n.b3; h.out = out; h.ldo = ldo; h.out_rows = out_rows; }
        if (g_lab.head_fused_supported(h)) {
            KL(g_lab.head_fwd(h, c.st));
            if (out && out_rows > 0 && !score) GG_TRY(lin_fwd(c, a2, H, w + n.w3, H, w + n.b3, out, ldo, out_rows, n.OUT, H));
            return 0;
        }
    }
    GG_TRY(lin_fwd(c, cvec, E, w + n.w1 + n.V, n.V + E, w + n.b1, a1, H, rows, H, E, ACT_LRELU, slope, 1));
    GG_TRY(lin_fwd(c, a1, H, w + n.w2, H, w + n.b2, a2, H, rows, H, H, ACT_LRELU, slope));
    if (out && out_rows > 0) GG_TRY(lin_fwd(c, a2, H, w + n.w3, H, w + n.b3, out, ldo, out_rows, n.OUT, H));
    return 0;
}

// backward through one head for `rows` rows.  dout [rows, OUT].  vin: the non-conditioning input
// [rows, V], cvec [rows, E].  If param_grads: accumulate the six parameter gradients.  Outputs:
// dcond [rows,E] (may be null), dv [rows,V] (may be null).
int head_backward(Ctx& c, Net& n, const float* dout, const float* vin, const float* cvec, const float* a1, const float* a2,
                  int rows, bool param_grads, float* dcond, float* dv, bool out_bias_grad = true) {
    gg_engine* e = c.e;
    const int E = e->E, H = e->H, V = n.V, OUT = n.OUT;
    const float* w = n.w;
    float* g = n.g;
    const float slope = e->cfg.negative_slope;
    float* dh2 = e->dA;
    float* dh1 = e->dB;
    // The parameter gradients are leaves (their operands dout / dh2 / dh1 / activations are not rewritten before the
    // backward's final join): they go to the side stream, the data-gradient chain continues on the caller's.
    if (param_grads) {
        Ctx cs = c;
        const bool fk = side_begin(c, cs);
        GG_TRY(lin_bwd_weight(cs, dout, OUT, a2, H, g + n.w3, H, rows, OUT, H));
        if (out_bias_grad) { GG_TRY(k_colsum(dout, rows, OUT, OUT, g + n.b3, cs.st)); e->launches++; }
        GG_TRY(side_end(c, fk, 3));
    }
    bool fused = false;
    if (e->head_on && g_lab.head_bwd && e->precision == GG_PREC_BF16 && !e->x3) {      // dh2, dh1 (and dcond) in one launch (head.hip)
        HeadP h;
        h.rows = rows; h.H = H; h.E = E; h.slope = slope;
        h.W1c = w + n.w1 + V; h.ldw1 = V + E; h.W2 = w + n.w2; h.w3 = w + n.w3;
        h.a1 = const_cast<float*>(a1); h.a2 = const_cast<float*>(a2); h.dh2 = dh2; h.dh1 = dh1; h.dcond = dcond;
        if (g_lab.head_fused_supported(h)) {
            if (OUT == 1) h.dout = dout;
            else GG_TRY(lin_bwd_data(c, dout, OUT, w + n.w3, H, dh2, H, rows, OUT, H));       // the generator's wide layer: dh2 <- dout W3
            KL(g_lab.head_bwd(h, c.st));
            fused = true;
        }
    }
    if (!fused) {
        GG_TRY(lin_bwd_data(c, dout, OUT, w + n.w3, H, dh2, H, rows, OUT, H));
        KL(k_act_bwd(dh2, a2, (long)rows * H, slope, 1.f, c.st));
    }
    if (param_grads) {
        Ctx cs = c;
        const bool fk = side_begin(c, cs);
        GG_TRY(lin_bwd_weight(cs, dh2, H, a1, H, g + n.w2, H, rows, H, H));
        GG_TRY(k_colsum(dh2, rows, H, H, g + n.b2, cs.st)); e->launches++;
        GG_TRY(side_end(c, fk, 3));
    }
    if (!fused) {
        GG_TRY(lin_bwd_data(c, dh2, H, w + n.w2, H, dh1, H, rows, H, H));
        KL(k_act_bwd(dh1, a1, (long)rows * H, slope, 1.f, c.st));
    }
    if (param_grads) {
        Ctx cs = c;
        const bool fk = side_begin(c, cs);
        GG_TRY(lin_bwd_weight(cs, dh1, H, vin, V, g + n.w1, V + E, rows, H, V));
        GG_TRY(lin_bwd_weight(cs, dh1, H, cvec, E, g + n.w1 + V, V + E, rows, H, E));
        GG_TRY(k_colsum(dh1, rows, H, H, g + n.b1, cs.st)); e->launches++;
        GG_TRY(side_end(c, fk, 3));
    }
    if (dcond && !fused) GG_TRY(lin_bwd_data(c, dh1, H, w + n.w1 + V, V + E, dcond, E, rows, H, E));
    if (dv) GG_TRY(lin_bwd_data(c, dh1, H, w + n.w1, V + E, dv, V, rows, H, V));
    return 0;
}

int generator_forward(Ctx& c, const float* z, const gg_cond* in, float* x_out, int train, int keep = -1) {
    gg_engine* e = c.e;
    Net& n = e->net[GG_ROLE_GENERATOR];
    const int B = in->B;
    GG_TRY(cond_forward(c, n, in, e->actsG, 1, train ? e->dropout : 0.f, keep));
    GG_TRY(lin_fwd(c, z, e->L, n.w + n.w1, e->L + e->E, nullptr, e->headG.a1, e->H, B, e->H, e->L));
    GG_TRY(head_finish(c, n, e->actsG.c, e->headG.a1, e->headG.a2, x_out, e->G, B, B));
    return 0;
}

int check_cond(gg_engine* e, const gg_cond* c) {
    GG_REQUIRE(c && c->patches && c->patch_pad && c->text && c->text_pad, "null conditioning input");
    GG_REQUIRE(c->B >= 1 && c->B <= e->maxB, "batch exceeds max_batch");
    GG_REQUIRE(c->P >= 1 && c->P <= e->maxP, "patch count exceeds max_patches");
    GG_REQUIRE(c->T >= 1 && c->T <= e->maxT, "text token count exceeds max_text_tokens");
    GG_REQUIRE(e->ws != nullptr, "workspace not bound");
    GG_REQUIRE(e->net[0].w && e->net[1].w, "parameters not bound");
    return 0;
}

int apply_opt(Ctx& c, Net& n, float max_norm, float grad_scale) {
    gg_engine* e = c.e;
    GG_REQUIRE(n.w && n.g && n.s1, "network buffers not bound");
    GG_REQUIRE(e->cfg.optimizer == GG_OPT_RMSPROP || n.s2, "Adam needs the second state buffer");
    float* ss = e->sumsq + 1024 * n.role;
    int n_partials = 0;
    if (max_norm > 0.f) KL(k_sumsq(n.g, n.live, ss, &n_partials, c.st));
    n.step_t += 1;
    KL(k_opt_step(n.w, n.g, n.s1, n.s2, n.live, e->cfg.optimizer, n.lr, max_norm, ss, n_partials, grad_scale, n.step_t,
                  e->capturing ? e->dev_words + 1 + n.role : nullptr, c.st));
    return 0;
}

// Gradient penalty of the B interpolate rows whose head activations are a1h / a2h (R:351-374) and, if `backward`, its
// double backward into dW1x, dW2, dw3 (the only parameters it reaches).  *gp_loss += the penalty.
int gp_chain(Ctx& c, Net& D, const float* a1h, const float* a2h, int B, float* gp_loss, bool backward) {
    gg_engine* e = c.e;
    const int G = e->G, E = e->E, H = e->H;
    const float slope = e->cfg.negative_slope;
    KL(k_gp_front(a1h, a2h, D.w + D.w3, D.w + D.w2, e->gp_g1, e->gp_dg1, e->gp_nrm2, B, H, slope, c.st));   // g1 = m1 * ((m2*w3) W2)
    const bool g3 = gp_grad3_ok(e->gp_g1, D.w + D.w1, G + E, e->gp_grad, B, H, G);
    if (g3) KL(k_gp_grad3(e->gp_g1, D.w + D.w1, G + E, e->gp_grad, e->gp_nrm2p, B, H, G, c.st));             // grad = g1 W1x, |grad|^2 per strip
    else KL(k_gp_grad(e->gp_g1, D.w + D.w1, G + E, e->gp_grad, e->gp_nrm2, B, H, G, c.st));
    KL(k_gp_coef_scale(g3 ? e->gp_nrm2p : e->gp_nrm2, e->gp_g1, e->gp_coef, backward ? e->gp_g1s : nullptr, gp_loss, B, H,
                       e->cfg.gp_weight, c.st, g3 ? gp_grad3_parts(G) : 1, e->gp_nrm2));
    if (!backward) return 0;
    {
        Ctx cs = c;
        const bool fk = side_begin(c, cs);
        GG_TRY(lin_bwd_weight(cs, e->gp_g1s, H, e->gp_grad, G, D.g + D.w1, G + E, B, H, G));    // dW1x += (coef g1)^T grad
        GG_TRY(side_end(c, fk, 3));
    }
    {   // dg1pre += grad W1x^T : split over the gene dimension, atomics into the buffer gp_front zeroed
        GemmP p;
        p.A = e->gp_grad; p.B = D.w + D.w1; p.C = e->gp_dg1; p.M = B; p.N = H; p.K = G; p.lda = G; p.ldb = G + E; p.ldc = H;
        p.layA = LAY_KC; p.layB = LAY_KC;
        const long tiles = tiles_of(B, H);
        p.splitk = (int)std::max<long>(2, std::min<long>((G + 255) / 256, std::max<long>(1, 256 / tiles)));
        // fp32-grade in every mode (see gpchain.hip): six bf16 part products on the register-direct 32 x 32 kernel (1 280 workgroups,
        // every operand fragment in flight at once) when the operands are aligned for it, the exact fp32-input tile GEMM otherwise
        e->launches++;
        static const bool f32_only = getenv("GG_GP_DG1_F32") != nullptr;
        if (!f32_only && e->precision == GG_PREC_BF16 && gemm_small_x3_ok(p)) GG_TRY(gemm_small_x3(p, c.st));
        else GG_TRY(gemm_f32(p, c.st));
    }
    {   // dW2, dw3: on the side stream like the head's own dW2 / dW3 (they add into the same gradient slots)
        Ctx cs = c;
        const bool fk = side_begin(c, cs);
        GG_TRY(k_gp_tail(e->gp_dg1, e->gp_coef, a1h, a2h, D.w + D.w3, D.w + D.w2, D.g + D.w2, D.g + D.w3, B, H, slope, cs.st));
        e->launches++;
        GG_TRY(side_end(c, fk, 3));
    }
    return 0;
}

// critic conditioning forward computed ahead of the iteration that uses it (gg_critic_cond_prefetch): valid while the critic's
// weights and the minibatch are unchanged
// One critic iteration up to its optimiser step, in two phases so that a data-parallel host can start the all-reduce of the
// MLP-head gradients (complete after the head phase) under the conditioning stack's backward (the cond phase).
int critic_head_phase(Ctx& c, const float* x_real, const float* z, const float* alpha, const gg_cond* in, float* losses,
                      const float* x_fake_pre = nullptr) {      // x_fake_pre: generator output computed ahead (its bf16 shadow
                                                              // weights were refreshed then and the generator is not run here)
    gg_engine* e = c.e;
    Net& D = e->net[GG_ROLE_CRITIC];
    const int B = in->B, G = e->G, E = e->E, H = e->H;
    const int R = e->dropout > 0.f ? 3 : 1;
    GG_REQUIRE(R <= e->maxR, "workspace was sized for dropout == 0; recreate the engine with dropout > 0");
    if (!x_fake_pre) GG_TRY(refresh_shadows(c, e->net[GG_ROLE_GENERATOR]));
    GG_TRY(refresh_shadows(c, D));
    KL(k_fill(losses, GG_N_LOSSES, 0.f, c.st));
    KL(k_fill(D.g, D.total, 0.f, c.st));
    phase_mark(c, "critic: shadow refresh, zero gradients (+ wait for x_fake)");
    // x_fake = G(z) (generator frozen: no activations kept beyond this call)   R:391
    if (x_fake_pre) KL(k_copy(e->X2, x_fake_pre, (long)B * G, c.st));      // computed ahead by generator_prefetch
    else GG_TRY(generator_forward(c, z, in, e->X2, 1, 0));
    KL(k_copy(e->X2 + (long)B * G, x_real, (long)B * G, c.st));
    // critic conditioning: R independent dropout replicas (fake, real, interpolate) R:403,404,360
    {
        const bool have = e->dcond_valid && e->dcond_B == B && e->dcond_P == in->P && e->dcond_T == in->T && e->dcond_R == R;
        e->dcond_valid = false;
        if (!have) {
            // nothing runs beside this pass (the previous iteration's parameter-gradient launches were joined before its optimiser
            // step, the generator passes of the step are done): its persistent Linear grids take every compute unit
            Ctx cf = c;
            cf.grid_pct = fwd_alone_pct();
            GG_TRY(cond_forward(cf, D, in, e->actsD, R, e->dropout, R == 1 ? 1 : 2));
        }
    }
    phase_mark(c, "critic: conditioning forward (3 replicas)");
    e->crit_R = R;
    if (R == 1) KL(k_copy_rows_bcast(e->c3, e->actsD.c, 3L * B, B, E, c.st));
    else KL(k_copy(e->c3, e->actsD.c, 3L * B * E, c.st));
    // first layer, gene part, for fake and real rows at once; the interpolate's is their lerp
    GG_TRY(lin_fwd(c, e->X2, G, D.w + D.w1, G + E, nullptr, e->Pfr, H, 2 * B, H, G));
    KL(k_copy(e->headD.a1, e->Pfr, 2L * B * H, c.st));
    KL(k_lerp_rows(e->Pfr, alpha, e->headD.a1 + 2L * B * H, B, H, c.st));
    GG_TRY(head_finish(c, D, e->c3, e->headD.a1, e->headD.a2, e->headD.out, 1, 3 * B, 2 * B));
    KL(k_critic_loss_seed(e->headD.out, e->dseed, losses, B, c.st));
    // D_loss backward through the head for the 2B fake/real rows
    // d(D_loss)/d(final bias) = sum(+1/B) + sum(-1/B) is identically zero (the reference's two symmetric
    // sums cancel exactly, R:43-45); summing the 2B seeds in one pass would leave ~1e-8 of rounding that
    // RMSprop/Adam normalise into a +-O(lr) drift of the critic's output offset, so it is not computed.
    GG_TRY(head_backward(c, D, e->dseed, e->X2, e->c3, e->headD.a1, e->headD.a2, 2 * B, true, e->dc, nullptr, false));
    // ---- gradient penalty, closed form on the interpolate rows: gpchain.hip, six launches -------------
    GG_TRY(gp_chain(c, D, e->headD.a1 + 2L * B * H, e->headD.a2 + 2L * B * H, B, losses + GG_LOSS_GP, true));
    phase_mark(c, "critic: MLP head forward / backward + gradient penalty");
    return 0;
}
// conditioning backward for the rows that carry gradient (second phase of the critic iteration)
int critic_cond_phase(Ctx& c, const gg_cond* in, int s0 = 0, int s1 = 1 << 20) {
    gg_engine* e = c.e;
    Net& D = e->net[GG_ROLE_CRITIC];
    const int B = in->B, E = e->E;
    const int R = e->crit_R;
    GG_REQUIRE(R == 1 || R == 3, "gg_critic_backward_cond without a preceding gg_critic_backward_head");
    if (s1 >= e->nl + 1) e->crit_R = 0;            // the last stage ends the iteration
    if (R == 1) {
        if (s0 == 0) KL(k_axpy(e->dc, e->dc + (long)B * E, 1.f, (long)B * E, c.st));
        GG_TRY(cond_backward(c, D, in, e->actsD, e->dc, 1, s0, s1));
    } else {
        GG_TRY(cond_backward(c, D, in, e->actsD, e->dc, 2, s0, s1));
    }
    return 0;
}

int critic_backward(Ctx& c, const float* x_real, const float* z, const float* alpha, const gg_cond* in, float* losses,
                    const float* x_fake_pre = nullptr) {
    GG_TRY(critic_head_phase(c, x_real, z, alpha, in, losses, x_fake_pre));
    return critic_cond_phase(c, in);
}

// The generator is frozen during the n_critic critic iterations of a train() (R:463-477) and the conditioning batch is
// the same, so its n forward passes (fresh z, fresh dropout draws) do not depend on the critic updates in between: they
// run here as dropout replicas stacked on the batch axis - up to maxR at a time, in the critic's activation arena, which
// is idle before the first critic iteration - instead of n one-replica passes (fewer, larger launches; better tails).
// Statistically identical to the sequential order: every replica draws its own dropout masks and uses its own z.
// one batched generator pass for outputs [first, first + r): conditioning stack in `acts`, head in `head` / `c3`
int prefetch_chunk(Ctx& c, const float* z_all, int first, int r, const gg_cond* in, CondActs& acts, HeadActs& head, float* c3) {
    gg_engine* e = c.e;
    Net& Gn = e->net[GG_ROLE_GENERATOR];
    const int B = in->B, G = e->G, E = e->E, H = e->H, Lz = e->L;
    const int rc = e->dropout > 0.f ? r : 1;                                // without dropout the conditioning replicas coincide
    {
        Ctx cf = c;
        if (c.st != e->pre_stream) cf.grid_pct = fwd_alone_pct();       // on the caller's stream, before the critic iterations: alone on the chip
        GG_TRY(cond_forward(cf, Gn, in, acts, rc, e->dropout, 0));
    }
    const float* cvec = acts.c;
    if (rc == 1 && r > 1) {
        KL(k_copy_rows_bcast(c3, acts.c, (long)r * B, B, E, c.st));
        cvec = c3;
    }
    GG_TRY(lin_fwd(c, z_all + (long)first * B * Lz, Lz, Gn.w + Gn.w1, Lz + E, nullptr, head.a1, H, r * B, H, Lz));
    GG_TRY(head_finish(c, Gn, cvec, head.a1, head.a2, e->Xpre + (long)first * B * G, G, r * B, r * B));
    return 0;
}
int prefetch_drain(Ctx& c) {        // the caller's stream waits for every output that is still being computed
    gg_engine* e = c.e;
    for (int k = 0; k < GG_MAX_PREFETCH; ++k)
        if (e->pre_wait[k]) {
            GG_CHECK_HIP(hipStreamWaitEvent(c.st, e->pre_ev[k], 0));
            e->pre_wait[k] = false;
        }
    return 0;
}
// the passes 1 .. n-1 of a pipelined prefetch, on the third stream in their own arena
int prefetch_rest(gg_engine* e) {
    if (!e->pre_rest.pending) return 0;
    e->pre_rest.pending = false;
    Ctx cp{e, e->pre_stream};
    const int n = e->pre_rest.n, rmax = e->pre_rest.rmax;
    for (int done = 1; done < n;) {
        const int r = std::min(n - done, rmax);
        GG_TRY(prefetch_chunk(cp, e->pre_rest.z_all, done, r, e->pre_rest.in, e->actsP, e->headP, e->c3P));
        for (int k = done; k < done + r; ++k) {
            GG_CHECK_HIP(hipEventRecord(e->pre_ev[k], e->pre_stream));
            e->pre_wait[k] = true;
        }
        done += r;
    }
    return 0;
}
int generator_prefetch(Ctx& c, const float* z_all, int n, const gg_cond* in, bool defer_rest = false) {
    gg_engine* e = c.e;
    Net& Gn = e->net[GG_ROLE_GENERATOR];
    const int B = in->B;
    GG_TRY(prefetch_drain(c));
    e->pre_n = 0; e->pre_next = 0; e->pre_B = B;
    n = std::min(n, GG_MAX_PREFETCH);
    if (n < 1) return 0;
    GG_TRY(refresh_shadows(c, Gn));
    const int rmax = e->dropout > 0.f ? e->maxR : 3;                        // head scratch holds 3B rows
    // Pipelined form (fused attention path only: the unfused one shares a softmax scratch buffer between arenas): the
    // first output is needed at once and is computed alone on the caller's stream; the others run on a third stream in
    // their own arena while the critic iterations that do not need them yet proceed.
    static const bool pipe_off = getenv("GG_NO_PREFETCH_PIPE") != nullptr;
    bool pipe = !pipe_off && n > 1 && e->side_on && e->flash && e->precision == GG_PREC_BF16 &&
                (e->x3 ? flash_attn_x3_supported(in->P + 1, e->E, e->nh) : flash_attn_supported(in->P + 1, e->E, e->nh));
    // Wide form (fused attention path only, as above): ALL passes as dropout replicas of ONE batched pass in the prefetch arena, on the
    // caller's stream.  A pass costs 0.39 ms + 0.34 ms per replica at cfg3 (FiLM, text and patch encoders and the layer-0 projection run
    // once per pass, and every launch has its fixed prologue): five replicas at once are 2.1 ms where 1 + 3 + 1 (the pipelined form:
    // the four later ones sat behind the parameter-gradient launches in the side streams' hardware queue and ran alone anyway,
    // tools/queue_probe.py) were 2.8 ms and 3 + 2 on one stream 2.45 ms.
    static const bool wide_off = getenv("GG_NO_PREFETCH_WIDE") != nullptr;
    const bool wide = !wide_off && n > 1 && e->flash && e->precision == GG_PREC_BF16 &&
                      (e->x3 ? flash_attn_x3_supported(in->P + 1, e->E, e->nh) : flash_attn_supported(in->P + 1, e->E, e->nh));
    if (wide) {
        for (int done = 0; done < n;) {
            const int r = std::min(n - done, e->preR);
            GG_TRY(prefetch_chunk(c, z_all, done, r, in, e->actsP, e->headP, e->c3P));
            done += r;
        }
        e->pre_n = n;
        return 0;
    }
    // a critic conditioning pass computed ahead lives in the critic's arena: the generator passes take the spare one, in order
    const bool spare = e->dcond_valid;
    if (spare) pipe = false;
    if (pipe && !e->pre_fork) {
        bool ok = e->pre_stream != nullptr || create_side_stream(&e->pre_stream);
        ok = ok && hipEventCreateWithFlags(&e->pre_fork, hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < GG_MAX_PREFETCH; ++k) ok = ok && hipEventCreateWithFlags(&e->pre_ev[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) { e->pre_fork = nullptr; pipe = false; }
    }
    if (!pipe) {
        for (int done = 0; done < n;) {
            const int r = std::min(n - done, rmax);
            if (spare) GG_TRY(prefetch_chunk(c, z_all, done, r, in, e->actsP, e->headP, e->c3P));
            else GG_TRY(prefetch_chunk(c, z_all, done, r, in, e->actsD, e->headD, e->c3));
            done += r;
        }
        e->pre_n = n;
        return 0;
    }
    GG_TRY(prefetch_chunk(c, z_all, 0, 1, in, e->actsD, e->headD, e->c3));
    GG_CHECK_HIP(hipEventRecord(e->pre_fork, c.st));                        // after the shadow refresh and everything before it
    GG_CHECK_HIP(hipStreamWaitEvent(e->pre_stream, e->pre_fork, 0));
    e->pre_n = n;
    // The remaining passes are ENQUEUED later when the caller asks for it (gg_train_step: after the first critic iteration
    // has been enqueued on the caller's stream): ~170 launches on the third stream take the host ~1 ms during which the
    // caller's stream would have nothing to run whenever the host is not far ahead of the GPU.
    e->pre_rest = {z_all, in, n, rmax, true};
    if (!defer_rest) GG_TRY(prefetch_rest(e));
    return 0;
}
// next stored generator output (nullptr: none); the caller's stream waits for it if it is still being computed
inline const float* next_prefetched(Ctx& c, int B) {
    gg_engine* e = c.e;
    if (e->pre_next >= e->pre_n || e->pre_B != B) return nullptr;
    const int k = e->pre_next++;
    if (e->pre_wait[k]) {
        if (hipStreamWaitEvent(c.st, e->pre_ev[k], 0) != hipSuccess) return nullptr;
        e->pre_wait[k] = false;
    }
    return e->Xpre + (long)k * B * e->G;
}

int generator_head_phase(Ctx& c, const float* z, const gg_cond* in, float* losses) {
    gg_engine* e = c.e;
    Net& Gn = e->net[GG_ROLE_GENERATOR];
    Net& D = e->net[GG_ROLE_CRITIC];
    const int B = in->B, G = e->G, E = e->E, H = e->H, L = e->L;
    GG_TRY(prefetch_drain(c));                     // passes of the frozen generator still in flight read its weights / shadows
    e->dcond_valid = false;                        // the frozen critic's forward below takes the critic's arena
    GG_TRY(refresh_shadows(c, Gn));
    GG_TRY(refresh_shadows(c, D));
    KL(k_fill(losses + GG_LOSS_G, 1, 0.f, c.st));
    KL(k_fill(Gn.g, Gn.total, 0.f, c.st));
    // The generator's forward (R:441, activations kept in actsG/headG) and the frozen critic's conditioning forward (R:449,
    // forward only) do not depend on each other: the critic's runs on the side stream (both are one-replica passes whose
    // launches leave most of the chip idle on their own).  Only with the fused attention path: the unfused one shares
    // a softmax scratch buffer between the networks.
    {
        Ctx cs = c;
        bool fk = false;
        if (e->flash && e->precision == GG_PREC_BF16 &&
            (e->x3 ? flash_attn_x3_supported(in->P + 1, E, e->nh) : flash_attn_supported(in->P + 1, E, e->nh))) fk = side_begin(c, cs);
        GG_TRY(cond_forward(cs, D, in, e->actsD, 1, e->dropout, 0));
        GG_TRY(side_end(c, fk, 3));
        GG_TRY(generator_forward(c, z, in, e->X2, 1));
        GG_TRY(side_wait(c, 3));
    }
    phase_mark(c, "generator: G forward beside the frozen critic's conditioning forward");
    GG_TRY(lin_fwd(c, e->X2, G, D.w + D.w1, G + E, nullptr, e->headD.a1, H, B, H, G));
    GG_TRY(head_finish(c, D, e->actsD.c, e->headD.a1, e->headD.a2, e->headD.out, 1, B, B));
    KL(k_gen_loss_seed(e->headD.out, e->dseed, losses, B, c.st));
    // through the frozen critic head down to x_fake, then the generator head and conditioning stack
    GG_TRY(head_backward(c, D, e->dseed, nullptr, nullptr, e->headD.a1, e->headD.a2, B, false, nullptr, e->dxfake));
    GG_TRY(head_backward(c, Gn, e->dxfake, z, e->actsG.c, e->headG.a1, e->headG.a2, B, true, e->dc, nullptr));
    phase_mark(c, "generator: critic head forward, both heads backward");
    (void)L;
    return 0;
}
int generator_cond_phase(Ctx& c, const gg_cond* in, int s0 = 0, int s1 = 1 << 20) {
    gg_engine* e = c.e;
    return cond_backward(c, e->net[GG_ROLE_GENERATOR], in, e->actsG, e->dc, 1, s0, s1);
}
int generator_backward(Ctx& c, const float* z, const gg_cond* in, float* losses) {
    GG_TRY(generator_head_phase(c, z, in, losses));
    return generator_cond_phase(c, in);
}

// WGAN_GP.gradient_penalty (R:351-374) as a call of its own: the penalty of the interpolates alpha*real + (1-alpha)*fake
// under the critic's current weights; no gradient is written.  *gp_out = mean((|grad_x^ D(x^)| - 1)^2).
int gradient_penalty(Ctx& c, const float* x_real, const float* x_fake, const float* alpha, const gg_cond* in, int train, float* gp_out) {
    gg_engine* e = c.e;
    Net& D = e->net[GG_ROLE_CRITIC];
    const int B = in->B, G = e->G, E = e->E, H = e->H;
    e->dcond_valid = false;
    GG_TRY(refresh_shadows(c, D));
    KL(k_copy(e->X2, x_fake, (long)B * G, c.st));
    KL(k_copy(e->X2 + (long)B * G, x_real, (long)B * G, c.st));
    GG_TRY(cond_forward(c, D, in, e->actsD, 1, train ? e->dropout : 0.f, 0));
    // first layer of the interpolate rows by linearity: x^ W1x^T = alpha (x W1x^T) + (1-alpha) (x~ W1x^T)
    GG_TRY(lin_fwd(c, e->X2, G, D.w + D.w1, G + E, nullptr, e->Pfr, H, 2 * B, H, G));
    KL(k_lerp_rows(e->Pfr, alpha, e->headD.a1, B, H, c.st));
    GG_TRY(head_finish(c, D, e->actsD.c, e->headD.a1, e->headD.a2, nullptr, 1, B, 0));
    KL(k_fill(gp_out, 1, 0.f, c.st));
    return gp_chain(c, D, e->headD.a1, e->headD.a2, B, gp_out, false);
}

// The critic's conditioning pass of the NEXT critic iteration (its R dropout replicas), ahead of time: it depends on the
// critic's weights and the minibatch only, so a data-parallel host runs it under the generator's gradient all-reduce.
int critic_cond_prefetch(Ctx& c, const gg_cond* in) {
    gg_engine* e = c.e;
    Net& D = e->net[GG_ROLE_CRITIC];
    const int R = e->dropout > 0.f ? 3 : 1;
    GG_REQUIRE(R <= e->maxR, "workspace was sized for dropout == 0; recreate the engine with dropout > 0");
    GG_TRY(refresh_shadows(c, D));
    GG_TRY(cond_forward(c, D, in, e->actsD, R, e->dropout, R == 1 ? 1 : 2));
    e->dcond_valid = true;
    e->dcond_B = in->B; e->dcond_P = in->P; e->dcond_T = in->T; e->dcond_R = R;
    return 0;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* gg_last_error(void) { return gg::g_err.c_str(); }
const char* gg_version(void) { return "gemm_gan_amd 0.3 (gfx950: bf16-MFMA engine, bf16x3 split-operand and f32-MFMA parity modes)"; }

// libgemmgan_lab.so hands over its entry points when it is loaded (include/gemmgan_lab.h); `bytes` guards against a stale build
int gg_lab_register(const void* table, uint64_t bytes) {
    GG_REQUIRE(table && bytes == sizeof(LabTable), "gg_lab_register: table of another build");
    g_lab = *static_cast<const LabTable*>(table);
    return 0;
}
int gg_create(const gg_config* cfg, gg_engine** out) {
    GG_REQUIRE(cfg && out, "null argument");
    GG_REQUIRE(cfg->n_heads > 0 && cfg->embedding_dims % cfg->n_heads == 0, "embedding_dims must divide by n_heads");
    GG_REQUIRE(cfg->n_layers >= 1 && cfg->n_layers <= MAXL, "n_layers out of range");
    GG_REQUIRE(cfg->embedding_dims <= 1024, "embedding_dims > 1024 unsupported");
    GG_REQUIRE(cfg->max_patches + 1 <= 2048 && cfg->max_text_tokens <= 2048, "sequence too long");
    GG_REQUIRE(cfg->n_genes > 0 && cfg->latent_dims > 0 && cfg->hidden_dims > 0 && cfg->text_dims > 0 && cfg->patch_dims > 0, "bad dims");
    GG_REQUIRE(cfg->max_batch > 0 && cfg->max_patches > 0 && cfg->max_text_tokens > 0, "bad capacity");
    GG_REQUIRE(cfg->dropout >= 0.f && cfg->dropout < 1.f, "bad dropout");
    gg_engine* e = new gg_engine();
    e->cfg = *cfg;
    e->E = cfg->embedding_dims; e->F = 2 * e->E; e->H = cfg->hidden_dims; e->G = cfg->n_genes; e->L = cfg->latent_dims;
    e->Dt = cfg->text_dims; e->Dp = cfg->patch_dims; e->nh = cfg->n_heads; e->nl = cfg->n_layers; e->dh = e->E / e->nh;
    GG_REQUIRE(cfg->variant >= GG_VARIANT_XATTN_FILM && cfg->variant <= GG_VARIANT_VANILLA, "unknown variant");
    GG_REQUIRE((!e->ffn2_on || g_lab.ffn2) && (!e->encb_on || g_lab.enc_bwd) && (!e->ffn_on || g_lab.ffn_fused) && (!e->head_on || g_lab.head_fwd),
               "GG_FFN2 / GG_ENCB / GG_FFN_FUSED / GG_HEAD_FUSED select kernels of libgemmgan_lab.so, which is not loaded");
    e->no_cond = cfg->variant == GG_VARIANT_VANILLA;
    GG_REQUIRE(!e->no_cond || cfg->dropout == 0.f, "the unconditional variant has no dropout site");
    e->xattn = cfg->variant == GG_VARIANT_XATTN_FILM;
    e->enc_bias = cfg->variant == GG_VARIANT_XATTN_FILM;
    e->film = cfg->variant != GG_VARIANT_IMG;
    e->pe_ln = cfg->variant == GG_VARIANT_IMG;
    e->maxB = cfg->max_batch; e->maxP = cfg->max_patches; e->maxT = cfg->max_text_tokens; e->maxS = e->maxP + 1;
    e->maxR = cfg->dropout > 0.f ? 3 : 1;
    {
        const char* pr = getenv("GG_PREFETCH_R");
        e->preR = std::max(3, std::min(GG_MAX_PREFETCH, pr ? atoi(pr) : 5));
    }
    e->dropout = cfg->dropout;
    e->seed = cfg->seed;
    GG_REQUIRE(cfg->precision >= GG_PREC_F32 && cfg->precision <= GG_PREC_BF16X3, "bad precision");
    e->precision = cfg->precision == GG_PREC_F32 ? GG_PREC_F32 : GG_PREC_BF16;
    e->fp8_fwd = cfg->precision == GG_PREC_FP8;
    e->x3 = cfg->precision == GG_PREC_BF16X3;
    build_net(e, GG_ROLE_GENERATOR);
    build_net(e, GG_ROLE_CRITIC);
    e->net[GG_ROLE_GENERATOR].lr = cfg->lr_g;
    e->net[GG_ROLE_CRITIC].lr = cfg->lr_d;
    e->ws_bytes = carve(e, nullptr);
    *out = e;
    return 0;
}

void gg_destroy(gg_engine* e) {
    if (!e) return;
    drop_graphs(e);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    if (e->pre_stream) (void)hipStreamSynchronize(e->pre_stream);
    if (e->pre_fork) {
        (void)hipEventDestroy(e->pre_fork);
        for (int k = 0; k < GG_MAX_PREFETCH; ++k) if (e->pre_ev[k]) (void)hipEventDestroy(e->pre_ev[k]);
    }
    if (e->pre_stream && e->pre_own) (void)hipStreamDestroy(e->pre_stream);
    if (e->side) (void)hipStreamSynchronize(e->side);
    if (e->ev_ready) {
        (void)hipEventDestroy(e->ev_ready);
        for (int i = 0; i < 5; ++i) if (e->ev_done[i]) (void)hipEventDestroy(e->ev_done[i]);
    }
    if (e->side && e->side_own) (void)hipStreamDestroy(e->side);
    for (hipEvent_t ev : e->phase_pool) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->prof_pool) (void)hipEventDestroy(ev);
    delete e;
}

int gg_param_count(const gg_engine* e, int role) { return e && (role == 0 || role == 1) ? (int)e->net[role].ps.size() : -1; }
const char* gg_param_name(const gg_engine* e, int role, int i) {
    if (!e || role < 0 || role > 1 || i < 0 || i >= (int)e->net[role].ps.size()) return nullptr;
    return e->net[role].ps[i].name.c_str();
}
int gg_param_info(const gg_engine* e, int role, int i, int64_t* offset, int64_t* numel, int32_t* ndim, int32_t shape[3]) {
    GG_REQUIRE(e && (role == 0 || role == 1) && i >= 0 && i < (int)e->net[role].ps.size(), "bad parameter index");
    const ParamInfo& p = e->net[role].ps[i];
    if (offset) *offset = p.off;
    if (numel) *numel = p.numel;
    if (ndim) *ndim = p.ndim;
    if (shape) { shape[0] = p.shape[0]; shape[1] = p.shape[1]; shape[2] = p.shape[2]; }
    return 0;
}
int64_t gg_flat_numel(const gg_engine* e, int role) { return e && (role == 0 || role == 1) ? e->net[role].total : -1; }

int gg_bind_net(gg_engine* e, int role, float* params, float* grads, float* s1, float* s2) {
    GG_REQUIRE(e && (role == 0 || role == 1), "bad role");
    GG_REQUIRE(params && grads, "null buffer");
    GG_REQUIRE(((uintptr_t)params % 16 == 0) && ((uintptr_t)grads % 16 == 0), "buffers must be 16-byte aligned");
    Net& n = e->net[role];
    n.w = params; n.g = grads; n.s1 = s1; n.s2 = s2;
    return 0;
}
size_t gg_workspace_bytes(const gg_engine* e) { return e ? e->ws_bytes : 0; }
int gg_bind_workspace(gg_engine* e, void* ws, size_t bytes) {
    GG_REQUIRE(e && ws, "null argument");
    GG_REQUIRE(bytes >= e->ws_bytes, "workspace too small");
    GG_REQUIRE((uintptr_t)ws % 256 == 0, "workspace must be 256-byte aligned");
    e->ws = ws;
    carve(e, ws);
    drop_graphs(e);
    GG_CHECK_HIP(hipMemset(e->dev_words, 0, 16 * sizeof(uint32_t)));
    for (int r = 0; r < 2; ++r) {
        Net& n = e->net[r];
        if (!n.tab.empty())
            GG_CHECK_HIP(hipMemcpy(n.tab_dev, n.tab.data(), n.tab.size() * sizeof(ShadowEntry), hipMemcpyHostToDevice));
    }
    return 0;
}

int gg_forward(gg_engine* e, int role, const float* v, const gg_cond* in, float* out, int train, void* stream) {
    GG_REQUIRE(e && v && out, "null argument");
    GG_REQUIRE(role == 0 || role == 1, "bad role");
    GG_TRY(check_cond(e, in));
    Ctx c{e, (hipStream_t)stream};
    GG_TRY(refresh_shadows(c, e->net[role]));
    if (role == GG_ROLE_GENERATOR) return generator_forward(c, v, in, out, train);
    e->dcond_valid = false;
    Net& D = e->net[GG_ROLE_CRITIC];
    const int B = in->B;
    GG_TRY(cond_forward(c, D, in, e->actsD, 1, train ? e->dropout : 0.f));
    GG_TRY(lin_fwd(c, v, e->G, D.w + D.w1, e->G + e->E, nullptr, e->headD.a1, e->H, B, e->H, e->G));
    GG_TRY(head_finish(c, D, e->actsD.c, e->headD.a1, e->headD.a2, out, 1, B, B));
    return 0;
}

int gg_critic_backward(gg_engine* e, const float* x_real, const float* z, const float* alpha, const gg_cond* in,
                       float* losses, void* stream) {
    GG_REQUIRE(e && x_real && z && alpha && losses, "null argument");
    GG_TRY(check_cond(e, in));
    Ctx c{e, (hipStream_t)stream};
    return critic_backward(c, x_real, z, alpha, in, losses, next_prefetched(c, in->B));
}
int gg_critic_backward_head(gg_engine* e, const float* x_real, const float* z, const float* alpha, const gg_cond* in,
                            float* losses, void* stream) {
    GG_REQUIRE(e && x_real && z && alpha && losses, "null argument");
    GG_TRY(check_cond(e, in));
    Ctx c{e, (hipStream_t)stream};
    GG_TRY(critic_head_phase(c, x_real, z, alpha, in, losses, next_prefetched(c, in->B)));
    return side_wait(c, 3);       // the MLP-head gradient slots are complete on the caller's stream
}
int gg_critic_backward_cond(gg_engine* e, const gg_cond* in, void* stream) {
    GG_REQUIRE(e, "null argument");
    GG_TRY(check_cond(e, in));
    Ctx c{e, (hipStream_t)stream};
    return critic_cond_phase(c, in);
}
int gg_critic_cond_prefetch(gg_engine* e, const gg_cond* in, void* stream) {
    GG_REQUIRE(e, "null argument");
    GG_TRY(check_cond(e, in));
    Ctx c{e, (hipStream_t)stream};
    return critic_cond_prefetch(c, in);
}
int gg_gradient_penalty(gg_engine* e, const float* x_real, const float* x_fake, const float* alpha, const gg_cond* in, int train,
                        float* gp_out, void* stream) {
    GG_REQUIRE(e && x_real && x_fake && alpha && gp_out, "null argument");
    GG_TRY(check_cond(e, in));
    Ctx c{e, (hipStream_t)stream};
    return gradient_penalty(c, x_real, x_fake, alpha, in, train, gp_out);
}
// bench.py: the gradient-penalty kernels on the buffers of the last critic iteration, `reps` times each, timed by their own
// dispatch timestamps.  us[4] = average microseconds of gp_front_k, gp_grad_k, gp_coef_k, gp_tail_k; bytes[4] = their
// algorithmic HBM bytes per launch.  Scratch outputs only (the tail adds into the gradient buffer, which the next
// iteration zeroes).
int gg_gp_profile(gg_engine* e, int B, int reps, double* us, double* bytes, void* stream) {
    GG_REQUIRE(e && us && bytes && reps > 0 && B > 0 && B <= e->maxB, "bad argument");
    GG_REQUIRE(e->ws && e->net[1].w, "engine not bound");
    hipStream_t st = (hipStream_t)stream;
    Net& D = e->net[GG_ROLE_CRITIC];
    const int G = e->G, E = e->E, H = e->H;
    const float slope = e->cfg.negative_slope;
    const float* a1h = e->headD.a1 + 2L * B * H;
    const float* a2h = e->headD.a2 + 2L * B * H;
    hipEvent_t ev[2];
    GG_CHECK_HIP(hipEventCreate(&ev[0]));
    GG_CHECK_HIP(hipEventCreate(&ev[1]));
    for (int k = 0; k < 4; ++k) {
        double acc = 0.0;
        for (int r = 0; r < reps; ++r) {
            gp_time_next(ev[0], ev[1]);
            int rc = 0;
            if (k == 0) rc = k_gp_front(a1h, a2h, D.w + D.w3, D.w + D.w2, e->gp_g1, e->gp_dg1, e->gp_nrm2, B, H, slope, st);
            const bool g3 = gp_grad3_ok(e->gp_g1, D.w + D.w1, G + E, e->gp_grad, B, H, G);
            if (k == 1) rc = g3 ? k_gp_grad3(e->gp_g1, D.w + D.w1, G + E, e->gp_grad, e->gp_nrm2p, B, H, G, st)
                                : k_gp_grad(e->gp_g1, D.w + D.w1, G + E, e->gp_grad, e->gp_nrm2, B, H, G, st);
            if (k == 2) rc = k_gp_coef_scale(g3 ? e->gp_nrm2p : e->gp_nrm2, e->gp_g1, e->gp_coef, e->gp_g1s, e->sumsq + 2048, B, H, e->cfg.gp_weight, st,
                                             g3 ? gp_grad3_parts(G) : 1, e->gp_nrm2);
            if (k == 3) rc = k_gp_tail(e->gp_dg1, e->gp_coef, a1h, a2h, D.w + D.w3, D.w + D.w2, D.g + D.w2, D.g + D.w3, B, H, slope, st);
            if (rc != 0) return rc;
            GG_CHECK_HIP(hipEventSynchronize(ev[1]));
            float ms = 0.f;
            GG_CHECK_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
            acc += ms * 1e3;
        }
        us[k] = acc / reps;
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    const double BH = 4.0 * B * H, HH = 4.0 * H * H, BG = 4.0 * B * G, HG = 4.0 * H * G;
    bytes[0] = 2 * BH + HH + 2 * BH;          // a1, a2, W2 -> g1, dg1pre
    bytes[1] = BH + HG + BG;                  // g1, W1x -> grad (row norms by atomics)
    bytes[2] = 3 * BH;                        // g1 -> g1s (+ B scalars)
    bytes[3] = 3 * BH + 2 * HH;               // dg1pre, a1, a2, W2 -> dW2
    return 0;
}
int gg_generator_backward_head(gg_engine* e, const float* z, const gg_cond* in, float* losses, void* stream) {
    GG_REQUIRE(e && z && losses, "null argument");
    GG_TRY(check_cond(e, in));
    Ctx c{e, (hipStream_t)stream};
    GG_TRY(generator_head_phase(c, z, in, losses));
    return side_wait(c, 3);
}
int gg_generator_backward_cond(gg_engine* e, const gg_cond* in, void* stream) {
    GG_REQUIRE(e, "null argument");
    GG_TRY(check_cond(e, in));
    Ctx c{e, (hipStream_t)stream};
    return generator_cond_phase(c, in);
}
int gg_cond_stage_count(const gg_engine* e) { return !e ? -1 : (e->no_cond ? 0 : e->nl + 2); }
int gg_cond_stage_range(const gg_engine* e, int role, int stage, int64_t* offset, int64_t* numel) {
    GG_REQUIRE(e && (role == 0 || role == 1) && offset && numel, "bad argument");
    GG_REQUIRE(!e->no_cond && stage >= 0 && stage <= e->nl + 1, "gg_cond_stage_range: no such stage");
    const Net& n = e->net[role];
    // flat order (build_net): [CLS, FiLM, text encoder, patch encoder][layer 0] .. [layer nl-1][T2I, I2T][MLP head]; ghosts behind `live`
    const long x0 = e->xattn ? n.t2i.inw : n.w1, l0 = n.layer[0].sa.inw;
    long a, b;
    if (stage == 0) { a = x0; b = n.w1; }
    else if (stage <= e->nl) { const int l = e->nl - stage; a = n.layer[l].sa.inw; b = l + 1 < e->nl ? n.layer[l + 1].sa.inw : x0; }
    else { a = 0; b = l0; }
    *offset = a;
    *numel = b - a;
    return 0;
}
int gg_critic_backward_cond_stage(gg_engine* e, const gg_cond* in, int stage, void* stream) {
    GG_REQUIRE(e, "null argument");
    GG_TRY(check_cond(e, in));
    GG_REQUIRE(stage >= 0 && stage <= e->nl + 1, "gg_critic_backward_cond_stage: no such stage");
    Ctx c{e, (hipStream_t)stream};
    return critic_cond_phase(c, in, stage, stage);
}
int gg_generator_backward_cond_stage(gg_engine* e, const gg_cond* in, int stage, void* stream) {
    GG_REQUIRE(e, "null argument");
    GG_TRY(check_cond(e, in));
    GG_REQUIRE(stage >= 0 && stage <= e->nl + 1, "gg_generator_backward_cond_stage: no such stage");
    Ctx c{e, (hipStream_t)stream};
    return generator_cond_phase(c, in, stage, stage);
}
// the engine's side stream (weight-gradient leaves) waits for everything enqueued on `stream` so far: a collective a host then issues
// from the side stream depends on both streams' share of a stage without the caller's stream waiting for the leaves
int gg_side_join(gg_engine* e, void* stream) {
    GG_REQUIRE(e, "null argument");
    Ctx c{e, (hipStream_t)stream}, cs = c;
    (void)side_begin(c, cs);
    return 0;
}
int gg_mlp_grad_range(const gg_engine* e, int role, int64_t* offset, int64_t* numel) {
    GG_REQUIRE(e && (role == 0 || role == 1) && offset && numel, "bad argument");
    const Net& n = e->net[role];
    *offset = n.w1;
    *numel = n.live - n.w1;
    return 0;
}
int gg_generator_prefetch(gg_engine* e, const float* z_all, int n, const gg_cond* in, void* stream) {
    GG_REQUIRE(e && z_all && n >= 0, "bad argument");
    GG_TRY(check_cond(e, in));
    Ctx c{e, (hipStream_t)stream};
    return generator_prefetch(c, z_all, n, in);
}
int gg_critic_apply(gg_engine* e, float grad_scale, void* stream) {
    GG_REQUIRE(e, "null argument");
    Ctx c{e, (hipStream_t)stream};
    e->dcond_valid = false;
    return apply_opt(c, e->net[GG_ROLE_CRITIC], e->cfg.clip_d, grad_scale);
}
int gg_generator_backward(gg_engine* e, const float* z, const gg_cond* in, float* losses, void* stream) {
    GG_REQUIRE(e && z && losses, "null argument");
    GG_TRY(check_cond(e, in));
    Ctx c{e, (hipStream_t)stream};
    return generator_backward(c, z, in, losses);
}
int gg_generator_apply(gg_engine* e, float grad_scale, void* stream) {
    GG_REQUIRE(e, "null argument");
    Ctx c{e, (hipStream_t)stream};
    if (prefetch_drain(c) != 0) return -1;
    e->pre_n = e->pre_next = 0;                       // the generator changes: outputs computed ahead are stale
    return apply_opt(c, e->net[GG_ROLE_GENERATOR], e->cfg.clip_g, grad_scale);
}

namespace {
int train_step_body(Ctx& c, const float* x_real, const gg_cond* in, const float* z_all, const float* alpha_all, int n_critic,
                    float* losses) {
    gg_engine* e = c.e;
    e->launches = 0;
    e->phase_marks.clear();
    phase_mark(c, "step begin");
    const long zs = (long)in->B * e->L;
    static const bool defer = getenv("GG_NO_PREFETCH_DEFER") == nullptr;
    if (n_critic > 1 && e->prefetch_on) GG_TRY(generator_prefetch(c, z_all, n_critic, in, defer));
    phase_mark(c, "generator: passes computed ahead (all of the step in one batched pass)");
    for (int k = 0; k < n_critic; ++k) {
        if (k == 1) GG_TRY(prefetch_rest(e));           // (no-op unless deferred) output 1 is waited for just below
        GG_TRY(critic_backward(c, x_real, z_all + k * zs, alpha_all + (long)k * in->B, in, losses, next_prefetched(c, in->B)));
        phase_mark(c, "critic: conditioning backward (2 replicas)");
        e->dcond_valid = false;
        GG_TRY(apply_opt(c, e->net[GG_ROLE_CRITIC], e->cfg.clip_d, 1.f));
        phase_mark(c, "critic: clip + optimiser");
    }
    GG_TRY(prefetch_rest(e));
    e->pre_n = e->pre_next = 0;
    GG_TRY(generator_backward(c, z_all + n_critic * zs, in, losses));
    phase_mark(c, "generator: conditioning backward");
    GG_TRY(apply_opt(c, e->net[GG_ROLE_GENERATOR], e->cfg.clip_g, 1.f));
    phase_mark(c, "generator: clip + optimiser");
    return 0;
}

// Everything a captured step freezes: if any of it differs, it is another graph.
void step_signature(const gg_engine* e, const float* x_real, const gg_cond* in, const float* z_all,
                    const float* alpha_all, int n_critic, const float* losses, std::vector<uint64_t>& v) {
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint64_t)u; };
    v = {
        (uint64_t)(uintptr_t)x_real, (uint64_t)(uintptr_t)z_all, (uint64_t)(uintptr_t)alpha_all, (uint64_t)(uintptr_t)losses,
        (uint64_t)(uintptr_t)in->patches, (uint64_t)(uintptr_t)in->patch_pad, (uint64_t)(uintptr_t)in->text,
        (uint64_t)(uintptr_t)in->text_pad, (uint64_t)in->B, (uint64_t)in->P, (uint64_t)in->T, (uint64_t)n_critic,
        (uint64_t)e->precision | (uint64_t)e->fp8_fwd << 8 | (uint64_t)e->side_on << 9 | (uint64_t)e->prefetch_on << 10 |
            (uint64_t)e->flash << 11 | (uint64_t)e->tlin_on << 12 | (uint64_t)e->wgrad_on << 13 | (uint64_t)e->bstore_on << 14 |
            (uint64_t)e->small_on << 15 | (uint64_t)e->sqx_on << 16 | (uint64_t)e->x3 << 17 | (uint64_t)e->ffn_on << 18 | (uint64_t)e->xstore_on << 19 | (uint64_t)e->lnb_on << 20 | (uint64_t)e->rstore_on << 21 | (uint64_t)e->head_on << 22 | (uint64_t)e->ffn2_on << 23 | (uint64_t)e->encb_on << 26,
        bits(e->dropout), bits(e->net[0].lr), bits(e->net[1].lr), (uint64_t)e->seed, (uint64_t)(uintptr_t)e->ws};
    for (int r = 0; r < 2; ++r)
        for (const float* q : {e->net[r].w, e->net[r].g, e->net[r].s1, e->net[r].s2}) v.push_back((uint64_t)(uintptr_t)q);
}

__global__ void set_words_k(uint32_t* w, uint32_t a, uint32_t b, uint32_t c) { w[0] = a; w[1] = b; w[2] = c; }

constexpr size_t GG_MAX_GRAPHS = 8;
}  // namespace

int gg_train_step(gg_engine* e, const float* x_real, const gg_cond* in, const float* z_all, const float* alpha_all,
                  int n_critic, float* losses, void* stream) {
    GG_REQUIRE(e && x_real && z_all && alpha_all && losses, "null argument");
    GG_REQUIRE(n_critic >= 0, "bad n_critic");
    GG_TRY(check_cond(e, in));
    Ctx c{e, (hipStream_t)stream};
    if (!e->graph_on || e->prof_on) return train_step_body(c, x_real, in, z_all, alpha_all, n_critic, losses);

    // ---- captured step: first sight of a signature runs eagerly (lazy initialisation happens there), the second is
    // captured, later ones replay.  A caller whose buffers move every step never reaches the second sight: it stays eager.
    std::vector<uint64_t> sig;
    step_signature(e, x_real, in, z_all, alpha_all, n_critic, losses, sig);
    gg_engine::StepGraph* g = nullptr;
    for (auto& q : e->graphs)
        if (q.sig == sig) { g = &q; break; }
    if (!g) {
        if (e->graphs.size() >= GG_MAX_GRAPHS) {
            size_t old = 0;
            for (size_t i = 1; i < e->graphs.size(); ++i)
                if (e->graphs[i].last_use < e->graphs[old].last_use) old = i;
            if (e->graphs[old].exec) (void)hipGraphExecDestroy(e->graphs[old].exec);
            e->graphs.erase(e->graphs.begin() + (long)old);
        }
        e->graphs.emplace_back();
        g = &e->graphs.back();
        g->sig = sig;
    }
    g->last_use = ++e->graph_clock;
    if (g->bad || g->seen++ == 0) return train_step_body(c, x_real, in, z_all, alpha_all, n_critic, losses);

    uint32_t toff[2] = {0, 0};
    if (!g->exec) {
        if (!e->cap_stream) GG_CHECK_HIP(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
        const uint32_t call0 = e->call_counter;
        const int t0[2] = {e->net[0].step_t, e->net[1].step_t};
        GG_CHECK_HIP(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed));
        Ctx cc{e, e->cap_stream};
        e->capturing = true;
        const int rc = train_step_body(cc, x_real, in, z_all, alpha_all, n_critic, losses);
        e->capturing = false;
        hipGraph_t graph = nullptr;
        const hipError_t ec = hipStreamEndCapture(e->cap_stream, &graph);
        hipError_t ei = hipErrorUnknown;
        if (rc == 0 && ec == hipSuccess && graph) ei = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) {     // nothing ran: restore the host-side counters and take the eager route from now on
            (void)hipGetLastError();
            g->exec = nullptr;
            g->bad = true;
            e->graph_failures++;
            e->call_counter = call0;
            e->net[0].step_t = t0[0]; e->net[1].step_t = t0[1];
            for (bool& b : e->side_pending) b = false;
            for (bool& b : e->pre_wait) b = false;
            e->pre_rest.pending = false;
            e->pre_n = e->pre_next = 0;
            e->dcond_valid = false;
            if (rc != 0) return rc;
            return train_step_body(c, x_real, in, z_all, alpha_all, n_critic, losses);
        }
        g->calls = e->call_counter - call0;
        for (int r = 0; r < 2; ++r) { g->t0[r] = t0[r]; g->tsteps[r] = e->net[r].step_t - t0[r]; }
        g->launches = e->launches;
        e->graph_captures++;
    } else {
        e->call_counter += g->calls;
        for (int r = 0; r < 2; ++r) {
            toff[r] = (uint32_t)(e->net[r].step_t - g->t0[r]);
            e->net[r].step_t += g->tsteps[r];
        }
        e->launches = g->launches;
        e->pre_n = e->pre_next = 0;
        e->dcond_valid = false;
        e->graph_replays++;
    }
    hipLaunchKernelGGL(set_words_k, dim3(1), dim3(1), 0, c.st, e->dev_words, ++e->epoch_host, toff[0], toff[1]);
    GG_CHECK_HIP(hipGraphLaunch(g->exec, c.st));
    return 0;
}

int gg_set_graph(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    e->graph_on = on != 0;
    if (!on) drop_graphs(e);
    return 0;
}
int gg_graph_stats(const gg_engine* e, int64_t* captures, int64_t* replays, int64_t* failures) {
    GG_REQUIRE(e, "null argument");
    if (captures) *captures = e->graph_captures;
    if (replays) *replays = e->graph_replays;
    if (failures) *failures = e->graph_failures;
    return 0;
}

int gg_set_lr(gg_engine* e, int role, float lr) {
    GG_REQUIRE(e && (role == 0 || role == 1), "bad role");
    e->net[role].lr = lr;
    return 0;
}
int gg_set_dropout(gg_engine* e, float p) {
    GG_REQUIRE(e && p >= 0.f && p < 1.f, "bad dropout");
    GG_REQUIRE(!(p > 0.f && e->maxR < 3), "engine was created with dropout == 0 (workspace sized for one replica)");
    e->dropout = p;
    return 0;
}
int gg_set_precision(gg_engine* e, int precision) {
    GG_REQUIRE(e && precision >= GG_PREC_F32 && precision <= GG_PREC_BF16X3, "bad precision");
    e->precision = precision == GG_PREC_F32 ? GG_PREC_F32 : GG_PREC_BF16;
    e->fp8_fwd = precision == GG_PREC_FP8;
    e->x3 = precision == GG_PREC_BF16X3;
    e->dcond_valid = false;
    e->pre_n = e->pre_next = 0;
    return 0;
}
int gg_set_side_streams(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    e->side_on = on != 0;
    return 0;
}
int gg_set_prefetch(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    e->prefetch_on = on != 0;
    e->pre_n = e->pre_next = 0;
    return 0;
}
int gg_set_flash(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    e->flash = on != 0;
    return 0;
}
int gg_set_encb(gg_engine* e, int on) {
    GG_REQUIRE(e, "null engine");
    GG_REQUIRE(!on || g_lab.enc_bwd, "gg_set_encb: this kernel lives in libgemmgan_lab.so, which is not loaded");
    e->encb_on = on != 0;
    drop_graphs(e);
    return 0;
}
int gg_set_ffn2(gg_engine* e, int mode) {
    GG_REQUIRE(e, "null engine");
    GG_REQUIRE(mode == 0 || mode == 1 || mode == 3, "gg_set_ffn2: 0 off, 1 / 3 on (4- / 8-slot weight ring)");
    GG_REQUIRE(mode == 0 || g_lab.ffn2, "gg_set_ffn2: this kernel lives in libgemmgan_lab.so, which is not loaded");
    e->ffn2_on = mode;
    drop_graphs(e);
    return 0;
}
int gg_set_ffn_fused(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    GG_REQUIRE(!on || g_lab.ffn_fused, "gg_set_ffn_fused: this kernel lives in libgemmgan_lab.so, which is not loaded");
    e->ffn_on = on != 0;
    return 0;
}
int gg_set_tlin(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    e->tlin_on = on != 0;
    return 0;
}
int gg_set_wgrad(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    e->wgrad_on = on != 0;
    return 0;
}
int gg_set_bstore(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    e->bstore_on = on != 0;
    return 0;
}
int gg_set_head_fused(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    GG_REQUIRE(!on || g_lab.head_fwd, "gg_set_head_fused: this kernel lives in libgemmgan_lab.so, which is not loaded");
    e->head_on = on != 0;
    return 0;
}
int gg_set_lnb_fused(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    e->lnb_on = on != 0;
    return 0;
}
int gg_set_xstore(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    e->xstore_on = on & 1;
    e->rstore_on = (on & 1) && !(on & 2);        // bit 1: keep the pre-LayerNorm sums in fp32
    return 0;
}
int gg_debug_buffer_is_bf16(gg_engine* e, const char* name) {
    if (!e || !name) return -1;
    std::string s(name);
    if (s.size() > 5 && s[1] == '.' && (s[0] == 'G' || s[0] == 'D') && s[2] == 'L') {
        const CondActs& a = s[0] == 'G' ? e->actsG : e->actsD;
        const std::string k = s.substr(5);
        if (a.bst && (k == "qkv" || k == "ctx" || k == "h")) return 1;
        if (a.xst && (k == "x1" || (k == "x2" && s[3] - '0' + 1 < e->nl))) return 1;
        if (a.rst && (k == "r1" || k == "r2")) return 1;
    }
    return 0;
}
int gg_set_sqx(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    e->sqx_on = on != 0;
    return 0;
}
int gg_set_seed(gg_engine* e, uint64_t seed) {
    GG_REQUIRE(e, "null argument");
    e->seed = seed;
    e->call_counter = 0;
    return 0;
}
int gg_reset_optimizer_steps(gg_engine* e) {
    GG_REQUIRE(e, "null argument");
    e->net[0].step_t = e->net[1].step_t = 0;
    return 0;
}
int gg_get_optimizer_step(const gg_engine* e, int role) { return e && (role == 0 || role == 1) ? e->net[role].step_t : -1; }
int gg_set_optimizer_step(gg_engine* e, int role, int step) {
    GG_REQUIRE(e && (role == 0 || role == 1) && step >= 0, "bad argument");
    e->net[role].step_t = step;
    return 0;
}
int64_t gg_launch_count(const gg_engine* e) { return e ? e->launches : -1; }
int gg_reset_launch_count(gg_engine* e) {
    GG_REQUIRE(e, "null argument");
    e->launches = 0;
    return 0;
}
// Host-owned streams for the engine's concurrent work (see gemmgan.h).  Must precede the first call that forks onto them.
int gg_bind_streams(gg_engine* e, void* side, void* prefetch) {
    GG_REQUIRE(e && side && prefetch && side != prefetch, "two distinct streams expected");
    GG_REQUIRE(!e->ev_ready && !e->pre_fork, "gg_bind_streams must be called before the engine's first iteration");
    e->side = (hipStream_t)side; e->side_own = false;
    e->pre_stream = (hipStream_t)prefetch; e->pre_own = false;
    return 0;
}

int gg_phase_enable(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    e->phase_on = on != 0;
    e->phase_marks.clear();
    return 0;
}
int gg_phase_count(const gg_engine* e) { return e ? (int)e->phase_marks.size() : -1; }
// mark `index` of the last gg_train_step: its name and the milliseconds since the previous mark (synchronises on the mark's event)
int gg_phase_read(gg_engine* e, int index, char* name, int name_cap, double* ms) {
    GG_REQUIRE(e && name && ms && index >= 0 && index < (int)e->phase_marks.size(), "bad argument");
    snprintf(name, (size_t)name_cap, "%s", e->phase_marks[(size_t)index].first);
    *ms = 0.0;
    if (index > 0) {
        GG_CHECK_HIP(hipEventSynchronize(e->phase_marks[(size_t)index].second));
        float f = 0.f;
        GG_CHECK_HIP(hipEventElapsedTime(&f, e->phase_marks[(size_t)index - 1].second, e->phase_marks[(size_t)index].second));
        *ms = f;
    }
    return 0;
}
int gg_profile_enable(gg_engine* e, int on) {
    GG_REQUIRE(e, "null argument");
    if (on < 0) { e->prof_on = false; return 0; }      // pause: no more event pairs, the records stay for gg_profile_collect
    e->prof_on = on != 0;
    e->prof_mask = on > 1 ? (unsigned)on >> 1 : 0xffffffffu;      // on = 1 | (class bit mask << 1) restricts the classes
    e->prof_named_all = on == 1;
    e->prof_named_sel.clear();
    if (on) { e->prof_recs.clear(); e->prof_next = 0; }
    return 0;
}
// event pairs for the named class only (add = 0), or for it as well as the classes already selected (add = 1); names as
// gg_profile_read reported them
namespace {
int profile_select_class(gg_engine* e, const char* name, bool add) {
    if (!add) { e->prof_mask = 0; e->prof_named_sel.clear(); }
    e->prof_named_all = false;
    bool found = false;
    for (size_t i = 0; i < e->named_cls.size() && !found; ++i)
        if (e->named_cls[i] == name) {
            if (e->prof_named_sel.size() <= i) e->prof_named_sel.resize(i + 1, false);
            e->prof_named_sel[i] = true; found = true;
        }
    const size_t fixed = std::min<size_t>(32, 18 + (size_t)e->n_str_cls);       // prof_agg = fixed ids, Linear ids, named classes
    for (size_t i = 0; i < e->prof_agg.size() && i < fixed && !found; ++i)
        if (e->prof_agg[i].name == name) { e->prof_mask |= 1u << i; found = true; }
    GG_REQUIRE(found, "gg_profile_enable_class: unknown class (names come from gg_profile_read after a full collect)");
    return 0;
}
}  // namespace
int gg_profile_enable_class(gg_engine* e, const char* name) {
    GG_REQUIRE(e && name, "null argument");
    GG_TRY(profile_select_class(e, name, false));
    e->prof_on = true;
    e->prof_recs.clear(); e->prof_next = 0;
    return 0;
}
int gg_profile_add_class(gg_engine* e, const char* name) {
    GG_REQUIRE(e && name, "null argument");
    GG_REQUIRE(e->prof_on && !e->prof_named_all, "gg_profile_add_class: call gg_profile_enable_class first");
    return profile_select_class(e, name, true);
}
int gg_profile_collect(gg_engine* e) {
    if (!e) return -1;
    static const char* names[8] = {"gemm_f32_kernel<KC,KC>", "gemm_f32_kernel<KC,KS>", "gemm_f32_kernel<KS,KC>", "gemm_f32_kernel<KS,KS>",
                                   "gemm_bf16_kernel<KC,KC>", "gemm_bf16_kernel<KC,KS>", "gemm_bf16_kernel<KS,KC>", "gemm_bf16_kernel<KS,KS>"};
    const int named0 = 18 + e->n_str_cls;
    e->prof_agg.assign(named0 + e->named_cls.size(), gg_engine::ProfAgg());
    for (size_t i = 0; i < e->named_cls.size(); ++i) e->prof_agg[named0 + i].name = e->named_cls[i];
    for (int i = 0; i < e->n_str_cls; ++i) e->prof_agg[18 + i].name = e->str_cls_name[i];
    e->prof_agg[15].name = "wgrad_kernel<true,true,false,false>";
    e->prof_agg[16].name = "wgrad_kernel<false,false,false,false>";
    e->prof_agg[17].name = "wgrad_kernel<false,false,true,false>";
    e->prof_agg[12].name = "tlin_res16_kernel<8,256,true,1>";      // + residual + LayerNorm epilogue
    e->prof_agg[13].name = "tlin_res16_kernel<8,256,true,2>";      // += (data gradients)
    e->prof_agg[14].name = "tlin_res16_kernel<8,256,true,0|3>";
    e->prof_agg[11].name = "gemm_small_kernel";
    e->prof_agg[10].name = "wgrad_kernel<true,false,false,false>";
    for (int i = 0; i < 8; ++i) e->prof_agg[i].name = names[i];
    e->prof_agg[8].name = "tlin_str_kernel";
    e->prof_agg[9].name = "tlin_res_kernel";
    for (auto& r : e->prof_recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) { set_error("hipEventSynchronize failed"); return -1; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { set_error("hipEventElapsedTime failed"); return -1; }
        auto& a = e->prof_agg[r.cls >= 64 ? named0 + (r.cls - 64) : r.cls];
        a.launches++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    e->prof_recs.clear();
    e->prof_next = 0;
    return (int)e->prof_agg.size();
}
int gg_profile_read(gg_engine* e, int index, char* name, int name_cap, int64_t* launches, double* ms, double* flops, double* bytes) {
    GG_REQUIRE(e && index >= 0 && index < (int)e->prof_agg.size(), "bad index");
    const auto& a = e->prof_agg[index];
    if (name && name_cap > 0) { strncpy(name, a.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (launches) *launches = a.launches;
    if (ms) *ms = a.ms;
    if (flops) *flops = a.flops;
    if (bytes) *bytes = a.bytes;
    return 0;
}

int gg_debug_buffer(gg_engine* e, const char* name, void** ptr, int64_t* numel) {
    GG_REQUIRE(e && name && ptr && numel, "null argument");
    GG_REQUIRE(e->ws, "workspace not bound");
    std::string s(name);
    auto ret = [&](void* p, long n) { *ptr = p; *numel = n; return 0; };
    const long E = e->E, F = e->F, H = e->H, G = e->G, nh = e->nh, Dp = e->Dp;
    if (s.size() > 2 && s[1] == '.' && (s[0] == 'G' || s[0] == 'D' || s[0] == 'P')) {      // P: the arena of the generator passes computed ahead
        CondActs& a = s[0] == 'G' ? e->actsG : (s[0] == 'D' ? e->actsD : e->actsP);
        const long B = a.B, RB = (long)a.R * a.B, S = a.P + 1, T = a.T;
        std::string f = s.substr(2);
        if (f == "gbpre") return ret(a.gbpre, B * 2 * Dp);
        if (f == "gb") return ret(a.gb, B * 2 * Dp);
        if (f == "tok") return ret(a.tok, B * T * E);
        if (f == "x0") return ret(a.x0, B * S * E);
        if (f == "c") return ret(a.c, RB * E);
        if (f == "t2i_out") return ret(a.t2i_out, RB * E);
        if (f == "i2t_out") return ret(a.i2t_out, RB * E);
        if (f == "t2i_P") return ret(a.t2i_P, RB * nh * S);
        if (f == "i2t_P") return ret(a.i2t_P, RB * nh * T);
        if (f == "t2i_kv") return ret(a.t2i_kv, RB * S * 2 * E);
        if (f.size() > 3 && f[0] == 'L' && f[2] == '.') {
            const int l = f[1] - '0';
            GG_REQUIRE(l >= 0 && l < e->nl, "bad layer index");
            LayerActs& L = a.L[l];
            std::string k = f.substr(3);
            if (k == "qkv") return ret(L.qkv, RB * S * 3 * E);
            if (k == "P") return ret(L.P, RB * nh * S * S);
            if (k == "ctx") return ret(L.ctx, RB * S * E);
            if (k == "r1") return ret(L.r1, RB * S * E);
            if (k == "x1") return ret(L.x1, RB * S * E);
            if (k == "h") return ret(L.h, RB * S * F);
            if (k == "x2") return ret(L.x2, RB * S * E);
        }
    } else {
        const long B = e->actsD.B ? e->actsD.B : e->actsG.B;
        if (s == "X2") return ret(e->X2, 2 * B * G);
        if (s == "Pfr") return ret(e->Pfr, 2 * B * H);
        if (s == "headD.a1") return ret(e->headD.a1, 3 * B * H);
        if (s == "headD.a2") return ret(e->headD.a2, 3 * B * H);
        if (s == "headD.out") return ret(e->headD.out, 2 * B);
        if (s == "headG.a1") return ret(e->headG.a1, B * H);
        if (s == "headG.a2") return ret(e->headG.a2, B * H);
        if (s == "gp_grad") return ret(e->gp_grad, B * G);
        if (s == "gp_nrm2") return ret(e->gp_nrm2, B);
        if (s == "gp_coef") return ret(e->gp_coef, B);
        if (s == "dc") return ret(e->dc, 2 * B * E);
        if (s == "dxfake") return ret(e->dxfake, B * G);
        if (s == "Xpre") return ret(e->Xpre, (long)e->pre_n * e->pre_B * G);          // generator outputs computed ahead, [n, B, G]
        if (s == "sPd") return ret(e->sPd, (long)e->actsD.R * B * nh * (e->actsD.P + 1) * (e->actsD.P + 1));
    }
    set_error("unknown debug buffer '" + s + "'");
    return -3;
}

}  // extern "C"
