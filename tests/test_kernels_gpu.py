"""Kernel-level oracle tests for the kernels the bench times.

Every test launches ONE kernel family of the hot path through the C ABI (`gg_test_*`, csrc/testhooks.hip - the same host
wrappers and routing the engine uses) and compares it with a float64 computation on the SAME operand values the kernel
multiplies: operands the kernel converts to bf16 are rounded to bf16 on the host first, dropout masks are regenerated on
the host from the (seed, site, call) triple with a numpy restatement of csrc/drop_rng.h.  What remains is accumulation order
in fp32 and the one rounding of a bf16-stored output, hence the bounds

    fp32 outputs   rel-L2 <= 1e-4   (measured 1e-7 .. 1e-6)
    bf16 outputs   rel-L2 <= 2e-3   (one bf16 rounding, unit roundoff 2^-8: 1.6 - 1.7e-3 rms measured) and max-norm <= 2^-8
    attention      rel-L2 <= 3.2e-3 (bf16 storage) / 2.5e-3 (fp32 storage): besides the stored result the kernels round ONE
                   intermediate tile to bf16 - the probabilities (forward, dV) or dS (dQ, dK), the B operand of the second
                   product - relative to a reference exponent only the kernel knows (lazy online-softmax maximum), so the host
                   cannot reproduce that rounding bit for bit; two independent roundings give 2.3e-3 rms (measured 1.8 - 2.5e-3)

- one to two orders tighter than the mode-level gates of tests/test_bf16_parity_gpu.py (whose gradient bounds carry the
ReLU-gate flips of a whole network).  A wrong fragment map, tile tail, mask, dropout index or epilogue gives O(1) here.

Reference semantics: torch nn.TransformerEncoderLayer / F.multi_head_attention_forward (transformer.py:940-983,
functional.py:6206-6660) as used by R:213, restated per kernel below."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from gemm_gan_amd import _lib as L
from gpu_util import diag

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PHI = 0x9E3779B1
M32 = 0xFFFFFFFF


# ---- host restatement of the dropout stream (csrc/kernels.hip make_drop_key, csrc/drop_rng.h) ------------------------------
def drop_key(p, seed, site, call):
    def mix(x):
        x &= (1 << 64) - 1
        x ^= x >> 33
        x = (x * 0xff51afd7ed558ccd) & ((1 << 64) - 1)
        x ^= x >> 33
        x = (x * 0xc4ceb9fe1a85ec53) & ((1 << 64) - 1)
        x ^= x >> 33
        return x
    a = mix(seed ^ ((0x9E3779B97F4A7C15 * (site + 1)) & ((1 << 64) - 1)))
    b = mix(a ^ ((0xD1B54A32D192ED03 * (call + 1)) & ((1 << 64) - 1)))
    k0 = (b ^ (b >> 32)) & M32
    thr = min(65535, max(0, int(np.rint(np.float32(p) * np.float32(65536.0)))))
    return k0, thr


def drop_keep(k0, thr, idx):
    """keep flags of the elements with (int64) indices `idx`: u16(i) >= thr, u16 = low / high half of fmix32((i >> 1) * PHI + k0)."""
    idx = np.asarray(idx, dtype=np.uint64)
    h = ((idx >> np.uint64(1)) * np.uint64(PHI) + np.uint64(k0)) & np.uint64(M32)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7feb352d)) & np.uint64(M32)
    h ^= h >> np.uint64(15)
    u16 = np.where((idx & np.uint64(1)) == 0, h & np.uint64(0xFFFF), h >> np.uint64(16))
    return u16 >= np.uint64(thr)


def keep_scale(p):
    return float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))


def bf(t):
    """bf16-round (round-to-nearest-even, what v_cvt_pk_bf16_f32 does) and return float64."""
    return t.float().to(torch.bfloat16).double()


def rel_l2(got, want):
    return float((got.double() - want).norm() / want.norm().clamp_min(1e-30))


def rel_max(got, want):
    return float((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))


def check(name, got, want, bf16_out, l2_bound=None, max_bound=None):
    got = got.detach().cpu()
    assert torch.isfinite(got.float()).all(), name
    l2, mx = rel_l2(got, want), rel_max(got, want)
    diag(f"   {name:44s} rel-L2 {l2:.2e}  max-norm {mx:.2e}  ({'bf16' if bf16_out else 'fp32'} output)")
    l2_bound = l2_bound or (2e-3 if bf16_out else 1e-4)
    max_bound = max_bound or (2.0 ** -8 if bf16_out else 2e-4)
    assert l2 <= l2_bound and mx <= max_bound, (name, l2, mx)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


LIN_CLASS = {32: "wst_ln_kernel<4,2,16,true,LN>", 33: "wst_ln_kernel<8,1,32,true,LN>", 34: "wst_ln_kernel<8,1,32,true,ACC>",
             35: "wst_ln_kernel<8,2,16,false,ACT>", 36: "wst_ln_kernel<8,2,16,true,MASK>", 37: "wst_ln_kernel<4,2,16,true,ACT>",
             46: "wst_ln_kernel<4,2,16,false,ACT,groups 3>", 47: "wst_ln_kernel<4,1,48,true,ACC,groups 2>", 48: "wst_ln_kernel<8,2,16,true,ACT>",
             49: "wst_ln_kernel<4,2,16,true,ACT,groups 3>", 1: "tlin_res_kernel",
             2: "tlin_res16_kernel<PRE_RES>", 3: "tlin_res16_kernel<PRE_ACC>", 4: "tlin_res16_kernel<other>", 0: "tlin_str_kernel"}


def run_linear(*, X, W, N, K, bias=None, y_bf16=False, x_bf16=False, relu=False, drop=None, mask_ref=None, mask_scale=1.0,
               accumulate=None, res=None, res_rows=0, ln=None, y_rows=-1, film=None, y_row_group=0, route=0, res_bf16=False,
               ln_y_bf16=False):
    """One Linear call.  X [M,K] float64-exact values (already bf16-representable when x_bf16), W [N,K] likewise.
    Returns (dict of outputs, kernel class)."""
    lib = L.load()
    M = X.shape[0]
    a = L.GGTestLinear()
    keep = []

    def dev(t, dt):
        t = t.to(DEV, dt).contiguous()
        keep.append(t)
        return t
    Xd = dev(X, torch.bfloat16 if x_bf16 else torch.float32)
    Wd = dev(W, torch.float32 if route >= 2 else torch.bfloat16)
    rows_out = M + (M // y_row_group if y_row_group else 0)
    ydt = torch.bfloat16 if y_bf16 else torch.float32
    if accumulate is not None:
        Yd = dev(accumulate, ydt).clone()
    else:
        Yd = torch.full((rows_out, N), float("nan"), dtype=ydt, device=DEV)
    a.X, a.ldx, a.M, a.x_bf16 = Xd.data_ptr(), K, M, int(x_bf16)
    a.W, a.ldw = Wd.data_ptr(), K
    if bias is not None:
        a.bias = dev(bias, torch.float32).data_ptr()
    a.Y, a.ldy, a.y_bf16, a.y_rows = Yd.data_ptr(), N, int(y_bf16), y_rows
    a.N, a.K = N, K
    if film is not None:
        g, b, group = film
        gd, bd = dev(g, torch.float32), dev(b, torch.float32)
        a.film_g, a.film_b, a.film_ld, a.film_group = gd.data_ptr(), bd.data_ptr(), K, group
    a.y_row_group = y_row_group
    a.act_relu = int(relu)
    if drop is not None:
        a.drop_p, a.drop_seed, a.drop_site, a.drop_call = drop
        a.drop_ld = N
    if mask_ref is not None:
        mr = dev(mask_ref, torch.float32 if route >= 2 else torch.bfloat16)
        a.mask_ref, a.ldref, a.mask_scale, a.mask_bf16 = mr.data_ptr(), N, mask_scale, int(route < 2)
    a.accumulate = int(accumulate is not None)
    if res is not None:
        rd = dev(res, torch.bfloat16 if res_bf16 else torch.float32)
        a.res, a.ldres, a.res_rows, a.res_bf16 = rd.data_ptr(), N, res_rows or M, int(res_bf16)
    out = {"Y": Yd}
    if ln is not None:
        g, b = ln
        gd, bd = dev(g, torch.float32), dev(b, torch.float32)
        out["ln_y"] = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16 if ln_y_bf16 else torch.float32)
        a.ln_y_bf16 = int(ln_y_bf16)
        out["ln_stats"] = torch.full((M, 2), float("nan"), device=DEV)
        a.ln_g, a.ln_b, a.ln_y, a.ln_stats = gd.data_ptr(), bd.data_ptr(), out["ln_y"].data_ptr(), out["ln_stats"].data_ptr()
    a.route = route
    if route >= 2:
        wparts = torch.empty(route * N * K, dtype=torch.bfloat16, device=DEV)
        keep.append(wparts)
        a.w_parts = wparts.data_ptr()
    cls = C.c_int32(-1)
    L.check(lib.gg_test_linear(C.byref(a), C.byref(cls), stream()))
    torch.cuda.synchronize()
    return out, cls.value


def ref_linear(*, X, W, N, K, bias=None, y_bf16=False, relu=False, drop=None, mask_ref=None, mask_scale=1.0, accumulate=None,
               res=None, res_rows=0, ln=None, film=None, exact=False, **_):
    """float64 restatement of the epilogue chain of tlin.hip / wst.hip: bias -> ReLU -> dropout -> gate -> (+ previous) ->
    (+ residual) -> LayerNorm(eps 1e-5)."""
    M = X.shape[0]
    Xe = X.float().double() if exact else bf(X)                             # fp32 activations are converted to bf16 on load
    if film is not None:
        Xe = X.float().double()
        g, b, group = film
        idx = torch.arange(M) // group
        Xe = (g.double()[idx] * Xe + b.double()[idx]).float()              # modulated in fp32,
        Xe = Xe.double() if exact else bf(Xe)                              # rounded to bf16 on the way to LDS (bf16x3: split, not rounded)
    y = Xe @ W.double().T
    if bias is not None:
        y = y + bias.double()
    if relu:
        y = y.clamp_min(0)
    if drop is not None:
        p, seed, site, call = drop
        k0, thr = drop_key(p, seed, site, call)
        keep = drop_keep(k0, thr, np.arange(M * N, dtype=np.int64)).reshape(M, N)
        y = y * torch.from_numpy(keep).double() * keep_scale(p)
    if mask_ref is not None:
        y = torch.where(mask_ref.double() > 0, y * float(np.float32(mask_scale)), torch.zeros_like(y))
    if accumulate is not None:
        y = y + accumulate.double()
    out = {}
    if res is not None:
        rr = res_rows or M
        y = y + res.double()[torch.arange(M) % rr]
    out["Y"] = y
    if ln is not None:
        g, b = ln
        mu = y.mean(-1, keepdim=True)
        var = ((y - mu) ** 2).mean(-1, keepdim=True)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        out["ln_y"] = (y - mu) * rstd * g.double() + b.double()
        out["ln_stats"] = torch.cat([mu, rstd], dim=1)
    return out


def rnd(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


# The Linear calls of one encoder layer, forward and backward, as engine.hip cond_forward / cond_backward issue them at the
# production width (E = 256, F = 512), plus the FiLM-fused patch encoder (K = 1024).  M = 8 samples x 257 tokens: not a
# multiple of the 32-token tile, so every tail path runs.
E_, F_ = 256, 512
M_ = 8 * 257
LINEAR_CASES = {
    # name: (builder returning kwargs), expected kernel class under route 0
    "qkv_projection": (lambda g: dict(X=rnd(g, M_, E_), W=bf(rnd(g, 3 * E_, E_, scale=0.06)), N=3 * E_, K=E_, bias=rnd(g, 3 * E_, scale=0.1),
                                      y_bf16=True), 46),
    "out_proj_dropout_residual_layernorm": (lambda g: dict(X=bf(rnd(g, M_, E_)), x_bf16=True, W=bf(rnd(g, E_, E_, scale=0.06)), N=E_, K=E_,
                                                           bias=rnd(g, E_, scale=0.1), drop=(0.1, 7, 1011, 3), res=rnd(g, M_, E_),
                                                           ln=(1 + rnd(g, E_, scale=0.1), rnd(g, E_, scale=0.1))), 32),
    "out_proj_layer0_shared_residual_keep_rows": (lambda g: dict(X=bf(rnd(g, M_, E_)), x_bf16=True, W=bf(rnd(g, E_, E_, scale=0.06)), N=E_, K=E_,
                                                                 bias=rnd(g, E_, scale=0.1), drop=(0.1, 7, 1011, 4), res=rnd(g, 4 * 257, E_),
                                                                 res_rows=4 * 257, y_rows=4 * 257,
                                                                 ln=(1 + rnd(g, E_, scale=0.1), rnd(g, E_, scale=0.1))), 32),
    "ffn1_relu_dropout": (lambda g: dict(X=rnd(g, M_, E_), W=bf(rnd(g, F_, E_, scale=0.06)), N=F_, K=E_, bias=rnd(g, F_, scale=0.1), relu=True,
                                         drop=(0.1, 7, 1012, 3), y_bf16=True), 35),
    "ffn2_dropout_residual_layernorm": (lambda g: dict(X=bf(rnd(g, M_, F_).clamp_min(0)), x_bf16=True, W=bf(rnd(g, E_, F_, scale=0.05)), N=E_, K=F_,
                                                       bias=rnd(g, E_, scale=0.1), drop=(0.1, 7, 1013, 3), res=rnd(g, M_, E_),
                                                       ln=(1 + rnd(g, E_, scale=0.1), rnd(g, E_, scale=0.1))), 33),
    "dhidden_gated_by_stored_activation": (lambda g: dict(X=bf(rnd(g, M_, E_)), x_bf16=True, W=bf(rnd(g, F_, E_, scale=0.05)), N=F_, K=E_,
                                                          mask_ref=bf(rnd(g, M_, F_).clamp_min(0)), mask_scale=keep_scale(0.1), y_bf16=True), 36),
    "dx1_accumulate_k512": (lambda g: dict(X=bf(rnd(g, M_, F_)), x_bf16=True, W=bf(rnd(g, E_, F_, scale=0.05)), N=E_, K=F_,
                                           accumulate=rnd(g, M_, E_)), 34),
    "dctx_bf16_to_bf16": (lambda g: dict(X=bf(rnd(g, M_, E_)), x_bf16=True, W=bf(rnd(g, E_, E_, scale=0.06)), N=E_, K=E_, y_bf16=True), 37),
    "dx_accumulate_k768": (lambda g: dict(X=bf(rnd(g, M_, 3 * E_)), x_bf16=True, W=bf(rnd(g, E_, 3 * E_, scale=0.04)), N=E_, K=3 * E_,
                                          accumulate=rnd(g, M_, E_)), 47),
    # the same calls as the engine issues them with the LayerNorm outputs stored in bf16 (engine.hip "xst"): bf16 X for QKV / FFN1,
    # bf16 residual rows and / or bf16 LayerNorm output for the two LayerNorm kernels
    "qkv_projection_bf16_x": (lambda g: dict(X=bf(rnd(g, M_, E_)), x_bf16=True, W=bf(rnd(g, 3 * E_, E_, scale=0.06)), N=3 * E_, K=E_,
                                             bias=rnd(g, 3 * E_, scale=0.1), y_bf16=True), 49),
    "ffn1_relu_dropout_bf16_x": (lambda g: dict(X=bf(rnd(g, M_, E_)), x_bf16=True, W=bf(rnd(g, F_, E_, scale=0.06)), N=F_, K=E_,
                                                bias=rnd(g, F_, scale=0.1), relu=True, drop=(0.1, 7, 1012, 5), y_bf16=True), 48),
    "out_proj_layer0_fp32_residual_bf16_output": (lambda g: dict(X=bf(rnd(g, M_, E_)), x_bf16=True, W=bf(rnd(g, E_, E_, scale=0.06)), N=E_, K=E_,
                                                                 bias=rnd(g, E_, scale=0.1), drop=(0.1, 7, 1011, 5), res=rnd(g, 4 * 257, E_),
                                                                 res_rows=4 * 257, y_rows=4 * 257, ln_y_bf16=True,
                                                                 ln=(1 + rnd(g, E_, scale=0.1), rnd(g, E_, scale=0.1))), 32),
    "out_proj_bf16_residual_bf16_output": (lambda g: dict(X=bf(rnd(g, M_, E_)), x_bf16=True, W=bf(rnd(g, E_, E_, scale=0.06)), N=E_, K=E_,
                                                          bias=rnd(g, E_, scale=0.1), drop=(0.1, 7, 1011, 6), res=bf(rnd(g, M_, E_)), res_bf16=True,
                                                          ln_y_bf16=True, ln=(1 + rnd(g, E_, scale=0.1), rnd(g, E_, scale=0.1))), 32),
    "ffn2_bf16_residual_fp32_output_last_layer": (lambda g: dict(X=bf(rnd(g, M_, F_).clamp_min(0)), x_bf16=True, W=bf(rnd(g, E_, F_, scale=0.05)),
                                                                 N=E_, K=F_, bias=rnd(g, E_, scale=0.1), drop=(0.1, 7, 1013, 5),
                                                                 res=bf(rnd(g, M_, E_)), res_bf16=True,
                                                                 ln=(1 + rnd(g, E_, scale=0.1), rnd(g, E_, scale=0.1))), 33),
    "ffn2_bf16_residual_bf16_output": (lambda g: dict(X=bf(rnd(g, M_, F_).clamp_min(0)), x_bf16=True, W=bf(rnd(g, E_, F_, scale=0.05)), N=E_, K=F_,
                                                      bias=rnd(g, E_, scale=0.1), drop=(0.1, 7, 1013, 6), res=bf(rnd(g, M_, E_)), res_bf16=True,
                                                      ln_y_bf16=True, y_rows=4 * 257, ln=(1 + rnd(g, E_, scale=0.1), rnd(g, E_, scale=0.1))), 33),
    "out_proj_bf16_everything": (lambda g: dict(X=bf(rnd(g, M_, E_)), x_bf16=True, W=bf(rnd(g, E_, E_, scale=0.06)), N=E_, K=E_,
                                                bias=rnd(g, E_, scale=0.1), drop=(0.1, 7, 1011, 7), res=bf(rnd(g, M_, E_)), res_bf16=True,
                                                ln_y_bf16=True, y_bf16=True, y_rows=5 * 257,
                                                ln=(1 + rnd(g, E_, scale=0.1), rnd(g, E_, scale=0.1))), 32),
    "ffn2_bf16_everything": (lambda g: dict(X=bf(rnd(g, M_, F_).clamp_min(0)), x_bf16=True, W=bf(rnd(g, E_, F_, scale=0.05)), N=E_, K=F_,
                                            bias=rnd(g, E_, scale=0.1), drop=(0.1, 7, 1013, 7), res=bf(rnd(g, M_, E_)), res_bf16=True,
                                            ln_y_bf16=True, y_bf16=True, ln=(1 + rnd(g, E_, scale=0.1), rnd(g, E_, scale=0.1))), 33),
    "patch_encoder_film_cls_rows": (lambda g: dict(X=rnd(g, 8 * 256, 1024), W=bf(rnd(g, E_, 1024, scale=0.03)), N=E_, K=1024,
                                                   bias=rnd(g, E_, scale=0.1), y_row_group=256,
                                                   film=(torch.tanh(rnd(g, 8, 1024)), rnd(g, 8, 1024).clamp(-5, 5), 256)), 1),
}


def _linear_case(name, route):
    build, want_cls = LINEAR_CASES[name]
    g = torch.Generator().manual_seed(abs(hash(name)) % 1000 + 11)
    kw = build(g)
    got, cls = run_linear(**kw, route=route)
    want = ref_linear(**kw)
    label = LIN_CLASS.get(cls, str(cls))
    diag(f"== Linear {name} (route {route}): {label}")
    if route == 0:
        assert cls == want_cls, (name, cls, want_cls)          # the hot-path kernel really ran (not a fallback)
    else:
        assert cls < 32 or cls == want_cls == 1, (name, cls)
    M = kw["X"].shape[0]
    yw = want["Y"]
    yg = got["Y"].float().cpu()
    if kw.get("y_row_group"):
        grp = kw["y_row_group"]
        rows = torch.arange(M) + torch.arange(M) // grp + 1
        skipped = torch.ones(yg.shape[0], dtype=torch.bool)
        skipped[rows] = False
        assert torch.isnan(yg[skipped]).all(), "rows in front of each sample (the CLS slots) must not be written"
        yg = yg[rows]
    if kw.get("y_rows", -1) >= 0:
        yr = kw["y_rows"]
        assert torch.isnan(yg[yr:]).all(), "pre-LayerNorm sums of forward-only rows must not be stored"
        check("pre-LN sum (stored rows)", yg[:yr], yw[:yr], bool(kw.get("y_bf16")))
    else:
        check("Y", yg, yw, bool(kw.get("y_bf16")))
    if "ln_y" in want:
        check("LayerNorm output", got["ln_y"].float(), want["ln_y"], bool(kw.get("ln_y_bf16")))
        yr = kw["y_rows"] if kw.get("y_rows", -1) >= 0 else M
        st = got["ln_stats"].cpu()
        check("LayerNorm statistics (mean, rstd)", st[:yr], want["ln_stats"][:yr], False)


@pytest.mark.parametrize("name", list(LINEAR_CASES))
def test_linear_kernels_of_the_hot_path_equal_the_fp64_product(name):
    """Route 0 = what the engine launches at the production width: the weight-stationary kernels (csrc/wst.hip, every epilogue:
    LayerNorm K = 256 / 512, +=, ReLU + dropout -> bf16, gate -> bf16, column groups 3 x 256 and 2 x 128) and tlin_res_kernel
    for the FiLM-fused patch encoder."""
    _linear_case(name, 0)


XSTORE_CASES = [n for n in LINEAR_CASES if "bf16_residual" in n or "bf16_output" in n or "bf16_everything" in n]        # weight-stationary kernels only


@pytest.mark.parametrize("name", [n for n in LINEAR_CASES if n != "patch_encoder_film_cls_rows" and n not in XSTORE_CASES])
def test_token_on_lane_linear_kernels_equal_the_fp64_product(name):
    """Route 1 = the token-on-lane kernels (csrc/tlin.hip: tlin_str_kernel<256, XB, YB, EPI>, tlin_res16_kernel<PRE_*>) the same calls
    run on when no weight-stationary instantiation takes them (other widths, GG_NO_WST)."""
    _linear_case(name, 1)


@pytest.mark.parametrize("parts", [2, 3])
@pytest.mark.parametrize("name", [n for n in LINEAR_CASES if n not in XSTORE_CASES and not n.endswith("_bf16_x")])
def test_split_operand_linear_equals_the_fp64_product_of_the_fp32_operands(name, parts):
    """Routes 2 / 3 = tlin3_kernel (csrc/tlin3.hip), the Linear of the bf16x3 parity mode: the same calls with fp32 activations,
    fp32 weights and fp32 outputs.  NO operand rounding in the reference.  Two operand parts (hi, lo; three MFMAs per tile, the
    backward form): measured 2 - 5e-6, bound 2e-5 - three orders below what one bf16 operand rounding costs.  Three parts (hi,
    mid, lo; six MFMAs, the forward form, whose results decide ReLU gates): fp32-grade, bound 1e-6 (measured ~1e-7)."""
    build, _ = LINEAR_CASES[name]
    g = torch.Generator().manual_seed(abs(hash(name)) % 1000 + 77)
    kw = build(g)
    kw["W"] = kw["W"] + 1e-3 * rnd(g, *kw["W"].shape).double()          # fp32 weights that are NOT bf16-representable
    kw["X"] = kw["X"].float() + 1e-3 * rnd(g, *kw["X"].shape)
    kw.update(x_bf16=False, y_bf16=False)
    got, cls = run_linear(**kw, route=parts)
    want = ref_linear(**kw, exact=True)
    diag(f"== Linear {name} (bf16x3, tlin3_kernel, {parts} operand parts)")
    bl2, bmx = (2e-5, 4e-5) if parts == 2 else (1e-6, 2e-6)
    M = kw["X"].shape[0]
    yg = got["Y"].float().cpu()
    if kw.get("y_row_group"):
        grp = kw["y_row_group"]
        rows = torch.arange(M) + torch.arange(M) // grp + 1
        yg = yg[rows]
    yr = kw["y_rows"] if kw.get("y_rows", -1) >= 0 else M
    assert yr == M or torch.isnan(yg[yr:]).all()
    check("Y", yg[:yr], want["Y"][:yr], False, bl2, bmx)
    if "ln_y" in want:
        check("LayerNorm output", got["ln_y"], want["ln_y"], False, bl2, bmx)
        check("LayerNorm statistics (mean, rstd)", got["ln_stats"].cpu()[:yr], want["ln_stats"][:yr], False, bl2, bmx)


# ---- fused self-attention ---------------------------------------------------------------------------------------------------
def attn_inputs(N, S, E, nh, seed, pad):
    g = torch.Generator().manual_seed(seed)
    qkv = rnd(g, N, S, 3 * E)
    qkv[:, :, :E] *= 1.5                                     # scores with a real spread (softmax not near-uniform)
    mask = torch.zeros(N, S, dtype=torch.bool)
    if pad:
        mask[0, S - S // 5:] = True                          # ragged: a padded tail,
        mask[1, 1:S:3] = True                                # every third key,
        if N > 2:
            mask[2, 1:] = True                               # and a sample whose only valid key is the CLS row
    dctx = rnd(g, N, S, E)
    return qkv, mask, dctx


def ref_attention(qkv, mask, dctx, ctx_stored, nh, drop, dctx_stored=None):
    """float64 forward and backward on the values the kernels multiply (qkv / dctx already rounded as stored).  Backward follows the
    kernels' own formulas: delta = rowsum(dO * O_stored), P from the forward's log-sum-exp."""
    N, S, E3 = qkv.shape
    E = E3 // 3
    dh = E // nh
    q, k, v = (qkv[:, :, i * E:(i + 1) * E].reshape(N, S, nh, dh).permute(0, 2, 1, 3).double() for i in range(3))
    sc = 1.0 / math.sqrt(dh)
    s = q @ k.transpose(-1, -2) * sc
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    lse = torch.logsumexp(s, dim=-1, keepdim=True)
    p = torch.exp(s - lse)
    keep = torch.ones_like(p)
    ks = 1.0
    if drop is not None:
        pr, seed, site, call = drop
        k0, thr = drop_key(pr, seed, site, call)
        ld = (S + 3) & ~3
        bh = torch.arange(N * nh, dtype=torch.int64).reshape(N, nh, 1, 1)
        qi = torch.arange(S, dtype=torch.int64).reshape(1, 1, S, 1)
        ki = torch.arange(S, dtype=torch.int64).reshape(1, 1, 1, S)
        idx = ((bh * S + qi) * ld + ki).numpy()
        keep = torch.from_numpy(drop_keep(k0, thr, idx)).double()
        ks = keep_scale(pr)
    pd = p * keep * ks
    o = pd @ v                                               # [N, nh, S, dh]
    ctx = o.permute(0, 2, 1, 3).reshape(N, S, E)
    out = {"ctx": ctx, "lse2": (lse.squeeze(-1) / math.log(2.0))}
    if dctx is not None:
        do = dctx.reshape(N, S, nh, dh).permute(0, 2, 1, 3).double()
        ost = ctx_stored.reshape(N, S, nh, dh).permute(0, 2, 1, 3).double()
        # delta = rowsum(dO * O) is VALU arithmetic on the tensors as stored (fp32 storage: the un-rounded dO)
        dst = do if dctx_stored is None else dctx_stored.reshape(N, S, nh, dh).permute(0, 2, 1, 3).double()
        delta = (dst * ost).sum(-1, keepdim=True)
        dv = pd.transpose(-1, -2) @ do
        dp = (do @ v.transpose(-1, -2)) * keep * ks
        ds = p * (dp - delta) * sc
        dq = ds @ k
        dk = ds.transpose(-1, -2) @ q
        pack = lambda t: t.permute(0, 2, 1, 3).reshape(N, S, E)
        out["dqkv"] = torch.cat([pack(dq), pack(dk), pack(dv)], dim=-1)
    return out


ATTN_SHAPES = [(257, 64), (258, 64), (1025, 64), (257, 16), (258, 16), (1025, 16)]


@pytest.mark.parametrize("S,dh", ATTN_SHAPES)
@pytest.mark.parametrize("io_bf16,pad,drop_on", [(1, True, True), (1, False, False), (0, True, True), (2, True, True), (3, True, True), (3, False, False)])
def test_fused_attention_forward_and_backward_equal_the_fp64_result(S, dh, io_bf16, pad, drop_on):
    """attn_fwd_rm / attn_bwd_dq_rm / attn_bwd_dkv_rm (S <= 512: K, V resident in LDS) and attn_*_stream_kernel (S = 1025:
    BASELINE configs[4]) against float64 softmax attention on the stored operand values: S = 257 (one CLS row past eight tiles),
    258, 1025; dh 64 (E = 256) and 16 (E = 64); ragged key masks incl. a sample with a single valid key; dropout on."""
    lib = L.load()
    nh = 4
    E = nh * dh
    N = 3 if S < 1000 else 2
    qkv, mask, dctx = attn_inputs(N, S, E, nh, seed=S + dh, pad=pad)
    drop = (0.1, 99, 2010, 5) if drop_on else None
    x3 = io_bf16 >= 2                                        # split-operand kernels: fp32 tensors, io_bf16 = operand parts
    dt = torch.bfloat16 if io_bf16 == 1 else torch.float32
    qkv_d, dctx_d = qkv.to(DEV, dt).contiguous(), dctx.to(DEV, dt).contiguous()
    mask_d = mask.to(DEV).view(torch.uint8).contiguous()
    ctx_d = torch.full((N, S, E), float("nan"), dtype=dt, device=DEV)
    lse_d = torch.full((N, nh, S), float("nan"), device=DEV)
    dp = drop or (0.0, 0, 0, 0)
    names = [lib.gg_test_attn_kernel_name(w, S, E, nh).decode() for w in range(3)]
    diag(f"== attention S={S} dh={dh} io_bf16={io_bf16} pad={pad} dropout={drop_on}: {names}")
    if x3:
        names = ["attn_fwd_x3_kernel", "attn_bwd_dq_x3_kernel", "attn_bwd_dkv_x3_kernel"]
    elif dh == 64:      # the production head width: the kernels the bench times (cfg3: *_rm, configs[4]: *_stream)
        assert names[0] == ("attn_fwd_rm_kernel" if S < 1000 else "attn_fwd_stream_kernel")
        assert names[1] == ("attn_bwd_dq_rm_kernel" if S < 1000 else "attn_bwd_dq_stream_kernel")
        assert names[2] == ("attn_bwd_dkv_rm_kernel" if S < 1000 else "attn_bwd_dkv_stream_kernel")
    L.check(lib.gg_test_attn_fwd(P(qkv_d), P(mask_d), N, P(ctx_d), P(lse_d), N, S, E, nh, C.c_float(dp[0]), dp[1], dp[2], dp[3],
                                 io_bf16, 0, stream()))
    delta_d = torch.zeros(N, nh, S, device=DEV)
    dqkv_d = torch.full((N, S, 3 * E), float("nan"), dtype=dt, device=DEV)
    L.check(lib.gg_test_attn_bwd(P(qkv_d), P(ctx_d), P(dctx_d), P(lse_d), P(delta_d), P(mask_d), N, P(dqkv_d), N, S, E, nh,
                                 C.c_float(dp[0]), dp[1], dp[2], dp[3], io_bf16, 0, stream()))
    torch.cuda.synchronize()
    # the kernels convert fp32 operands to bf16 on load: the reference multiplies those values in both storage modes
    rd = (lambda t: t.float().double()) if x3 else bf       # split-operand kernels: NO operand rounding in the reference
    want = ref_attention(rd(qkv_d.cpu()), mask, rd(dctx_d.cpu()), ctx_d.cpu().double(), nh, drop, dctx_stored=dctx_d.cpu().double())
    bl2, bmx = (3.2e-3, 2.0 ** -7) if io_bf16 == 1 else (2.5e-3, 2.0 ** -7)
    if io_bf16 == 2:
        bl2, bmx = 4e-5, 2e-4           # two parts per operand: 2^-18 each (measured 3e-6 .. 2e-5, max-norm <= 1.1e-4)
    if io_bf16 == 3:
        bl2, bmx = 4e-6, 8e-6           # three parts: fp32 grade (measured 1e-7 .. 8e-7)
    check("context", ctx_d, want["ctx"], io_bf16 == 1, bl2, bmx)
    valid = ~mask[:, None, :].expand(N, nh, S)
    check("log2-sum-exp", lse_d.cpu(), want["lse2"], False)
    dq, dk, dv = (dqkv_d[:, :, i * E:(i + 1) * E] for i in range(3))
    wq, wk, wv = (want["dqkv"][:, :, i * E:(i + 1) * E] for i in range(3))
    check("dQ", dq, wq, io_bf16 == 1, bl2, bmx)
    check("dK", dk, wk, io_bf16 == 1, bl2, bmx)
    check("dV", dv, wv, io_bf16 == 1, bl2, bmx)
    # masked keys receive exactly zero gradient
    mk = mask[:, :, None].expand(N, S, E)
    assert float(dk.float().cpu()[mk].abs().max() if mk.any() else 0.0) == 0.0
    assert float(dv.float().cpu()[mk].abs().max() if mk.any() else 0.0) == 0.0
    assert valid.any()


def test_attention_replicas_share_the_layer0_projection():
    """qkv_B > 0: sample n reads the projection of sample n % qkv_B (dropout replicas stacked on the batch axis share layer 0)."""
    lib = L.load()
    S, dh, nh = 257, 64, 4
    E, B, R = nh * dh, 2, 3
    qkv, mask, dctx = attn_inputs(B, S, E, nh, seed=5, pad=True)
    dctx = rnd(torch.Generator().manual_seed(6), R * B, S, E)
    drop = (0.1, 99, 2010, 6)
    qkv_d, dctx_d = qkv.to(DEV, torch.bfloat16), dctx.to(DEV, torch.bfloat16)
    mask_d = mask.to(DEV).view(torch.uint8).contiguous()
    ctx_d = torch.empty(R * B, S, E, dtype=torch.bfloat16, device=DEV)
    lse_d = torch.empty(R * B, nh, S, device=DEV)
    L.check(lib.gg_test_attn_fwd(P(qkv_d), P(mask_d), B, P(ctx_d), P(lse_d), R * B, S, E, nh, C.c_float(drop[0]), drop[1], drop[2], drop[3], 1, B,
                                 stream()))
    delta_d = torch.zeros(R * B, nh, S, device=DEV)
    dqkv_d = torch.empty(R * B, S, 3 * E, dtype=torch.bfloat16, device=DEV)
    L.check(lib.gg_test_attn_bwd(P(qkv_d), P(ctx_d), P(dctx_d), P(lse_d), P(delta_d), P(mask_d), B, P(dqkv_d), R * B, S, E, nh,
                                 C.c_float(drop[0]), drop[1], drop[2], drop[3], 1, B, stream()))
    torch.cuda.synchronize()
    want = ref_attention(bf(qkv_d.cpu()).repeat(R, 1, 1), mask.repeat(R, 1), bf(dctx_d.cpu()), ctx_d.cpu().double(), nh, drop)
    diag("== attention with a replica-shared projection (qkv_B = 2, N = 6)")
    check("context", ctx_d, want["ctx"], True, 3.2e-3, 2.0 ** -7)
    check("dqkv", dqkv_d, want["dqkv"], True, 3.2e-3, 2.0 ** -7)
    assert rel_l2(ctx_d[:B].cpu(), ctx_d[B:2 * B].cpu().double()) > 1e-2       # replicas draw their own dropout masks


# ---- weight gradient over the token rows ------------------------------------------------------------------------------------
WGRAD_CASES = {
    # name: (M, N, K, dy_bf16, x_bf16, extras)
    "ffn2_bf16_bf16": (4096 + 2048, 256, 512, 1, 1, {}),
    "ffn1_bf16_fp32_bias_ragged": (257 * 24, 512, 256, 1, 0, {"bias": True}),
    "qkv_bf16_fp32_shared_rows": (3 * 2048, 768, 256, 1, 0, {"bias": True, "x_mod": 2048}),
    "out_proj_fp32_fp32": (4096, 256, 256, 0, 0, {}),
    "patch_encoder_film_on_the_fly": (16 * 256, 256, 1024, 0, 0, {"film": 256}),
    "film_gradient_contraction": (16 * 256, 256, 1024, 0, 0, {"fgrad": 256}),
    "narrow_panel_tail": (4096, 200, 328, 0, 0, {}),
    # the split-operand (bf16x3) form of the parity mode: fp32 operands as hi / lo images, three MFMAs per tile - the reference
    # is the float64 product of the UN-rounded fp32 values
    "x3_ffn1_bias_ragged": (257 * 24, 512, 256, 0, 0, {"bias": True, "x3": 1}),
    "x3_qkv_shared_rows": (3 * 2048, 768, 256, 0, 0, {"bias": True, "x_mod": 2048, "x3": 1}),
    "x3_patch_encoder_film_on_the_fly": (16 * 256, 256, 1024, 0, 0, {"film": 256, "x3": 1}),
    "x3_film_gradient_contraction": (16 * 256, 256, 1024, 0, 0, {"fgrad": 256, "x3": 1}),
    "x3_narrow_panel_tail": (4096, 200, 328, 0, 0, {"x3": 1}),
}


@pytest.mark.parametrize("name", list(WGRAD_CASES))
def test_weight_gradient_kernel_equals_the_fp64_reduction(name):
    """wgrad_kernel (csrc/wgrad.hip): dW += dY^T X over >= 4096 token rows, both operands through hardware transpose reads; bf16 /
    fp32 stored operands, bias column sums, replica-shared X rows, FiLM applied on the fly, the FiLM-gradient contraction, a ragged
    last chunk (general loader) and panels narrower than 128 x 256."""
    lib = L.load()
    M, N, K, yb, xb, ex = WGRAD_CASES[name]
    g = torch.Generator().manual_seed(M + N + K)
    dY = rnd(g, M, N, scale=0.1)
    x_rows = ex.get("x_mod", M)
    X = rnd(g, x_rows, K)
    dW0 = rnd(g, N, K, scale=0.01)
    dYd = dY.to(DEV, torch.bfloat16 if yb else torch.float32).contiguous()
    Xd = X.to(DEV, torch.bfloat16 if xb else torch.float32).contiguous()
    dWd = dW0.to(DEV).contiguous()
    x3 = ex.get("x3", 0)
    rd = (lambda t: t.float().double()) if x3 else bf
    dyv, xv = rd(dYd.cpu()), rd(Xd.cpu())                   # the values the MFMAs see
    film_g = film_b = fW = dgam = dbet = dbias = None
    rows = torch.arange(M) % x_rows
    if "film" in ex:
        grp = ex["film"]
        film_g, film_b = torch.tanh(rnd(g, M // grp, K)).to(DEV), rnd(g, M // grp, K).clamp(-5, 5).to(DEV)
        idx = torch.arange(M) // grp
        xv = rd((film_g.cpu().double()[idx] * Xd.cpu().double() + film_b.cpu().double()[idx]).float())
    if "fgrad" in ex:
        grp = ex["fgrad"]
        fW = rnd(g, N, K, scale=0.05).to(DEV)
        dgam, dbet = torch.zeros(M // grp, K, device=DEV), torch.zeros(M // grp, K, device=DEV)
    if ex.get("bias"):
        dbias = torch.zeros(N, device=DEV)
    diag(f"== wgrad {name}: M={M} N={N} K={K} dY {'bf16' if yb else 'fp32'} X {'bf16' if xb else 'fp32'} {ex}")
    L.check(lib.gg_test_wgrad(P(dYd), N, yb, P(Xd), K, xb, P(dWd), K, M, N, K, P(film_g), P(film_b), K, ex.get("film", 0),
                              P(fW), K, P(dgam), P(dbet), K, ex.get("fgrad", 0), P(dbias), ex.get("x_mod", 0), x3, stream()))
    torch.cuda.synchronize()
    if "fgrad" in ex:
        grp = ex["fgrad"]
        nb = M // grp
        Cb = torch.einsum("btn,btk->bnk", dyv.reshape(nb, grp, N), xv.reshape(nb, grp, K))
        check("dgamma", dgam, (Cb * fW.cpu().double()).sum(1), False)
        sb = dyv.reshape(nb, grp, N).sum(1)
        check("dbeta", dbet, sb @ fW.cpu().double(), False)
        assert torch.equal(dWd.cpu(), dW0), "FiLM-gradient mode adds nothing to dW"
        return
    want = dW0.double() + dyv.T @ xv[rows]
    check("dW", dWd, want, False)
    if dbias is not None:
        check("bias gradient (column sums of dY)", dbias, dyv.sum(0), False)


# ---- projection-free single-query attention (T2I) -----------------------------------------------------------------------------
@pytest.mark.parametrize("S,E,nh", [(257, 256, 4), (258, 256, 4), (1025, 256, 4), (257, 64, 4), (33, 128, 2)])
def test_single_query_sweeps_equal_the_fp64_result(S, E, nh):
    """sqx2_fwd / sqx2_bwd (csrc/sqattn.hip): scores, softmax and sum_s p x_s in one sweep over the encoder output; backward writes
    dx in the sweep that accumulates dqt.  fp32 VALU arithmetic: 1e-4 bound, measured ~1e-6."""
    lib = L.load()
    N, B = 5, 3                                             # replica-stacked queries over B masks
    g = torch.Generator().manual_seed(S + E)
    qt, x, dxbar = rnd(g, N, nh, E, scale=0.5), rnd(g, N, S, E), rnd(g, N, nh, E)
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[1, S // 2:] = True
    mask[2, 1:] = True
    qt_d, x_d, dxbar_d = qt.to(DEV), x.to(DEV), dxbar.to(DEV)
    mask_d = mask.to(DEV).view(torch.uint8)
    probs_d, xbar_d = torch.empty(N, nh, S, device=DEV), torch.empty(N, nh, E, device=DEV)
    L.check(lib.gg_test_sqx_fwd(P(qt_d), P(x_d), P(mask_d), B, P(probs_d), P(xbar_d), N, S, E, nh, stream()))
    dx_d, dqt_d = torch.empty(N, S, E, device=DEV), torch.empty(N, nh, E, device=DEV)
    L.check(lib.gg_test_sqx_bwd(P(dxbar_d), P(qt_d), P(xbar_d), P(x_d), P(probs_d), P(dx_d), P(dqt_d), N, S, E, nh, stream()))
    torch.cuda.synchronize()
    sc = 1.0 / math.sqrt(E // nh)
    xd, qd, dd = x.double(), qt.double(), dxbar.double()
    s = torch.einsum("nhe,nse->nhs", qd, xd) * sc
    s = s.masked_fill(mask[torch.arange(N) % B][:, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    xbar = torch.einsum("nhs,nse->nhe", p, xd)
    dp = torch.einsum("nhe,nse->nhs", dd, xd)
    ds = p * (dp - (p * dp).sum(-1, keepdim=True)) * sc
    dx = torch.einsum("nhs,nhe->nse", p, dd) + torch.einsum("nhs,nhe->nse", ds, qd)
    dqt = torch.einsum("nhs,nse->nhe", ds, xd)
    diag(f"== single-query sweeps S={S} E={E} nh={nh}")
    check("probabilities", probs_d, p, False)
    check("xbar", xbar_d, xbar, False)
    check("dx", dx_d, dx, False)
    check("dqt", dqt_d, dqt, False)


# ---- LayerNorm backward with the fused column sums -----------------------------------------------------------------------------
@pytest.mark.parametrize("dres_bf16,drop_on,r_bf16", [(1, True, False), (0, False, False), (1, True, True)])
def test_layernorm_backward_with_fused_column_sums(dres_bf16, drop_on, r_bf16):
    """r_bf16: the pre-LayerNorm sums as the engine keeps them at the production width (bf16 array, statistics of the unrounded sums)."""
    lib = L.load()
    rows, E = 8 * 257, 256
    g = torch.Generator().manual_seed(3)
    r, dy, gam = rnd(g, rows, E), rnd(g, rows, E, scale=0.1), 1 + rnd(g, E, scale=0.1)
    mu = r.double().mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((r.double() - mu) ** 2).mean(-1, keepdim=True) + 1e-5)
    stats = torch.cat([mu, rstd], 1).float()
    drop = (0.1, 5, 1013, 2) if drop_on else (0.0, 0, 0, 0)
    d = lambda t: t.to(DEV).contiguous()
    r_d, dy_d, g_d, st_d = d(r), d(dy), d(gam), d(stats)
    if r_bf16:
        r_d = r_d.to(torch.bfloat16)
        r = r_d.float().cpu()              # the values the kernel reads (the statistics above stay those of the unrounded sums)
    dr_d = torch.empty(rows, E, device=DEV)
    dres_d = torch.empty(rows, E, dtype=torch.bfloat16 if dres_bf16 else torch.float32, device=DEV)
    dg_d, db_d, dbias_d = torch.zeros(E, device=DEV), torch.zeros(E, device=DEV), torch.zeros(E, device=DEV)
    L.check(lib.gg_test_ln_bwd(P(dy_d), P(r_d), P(st_d), P(g_d), P(dr_d), P(dres_d), P(dg_d), P(db_d), P(dbias_d), rows, E,
                               C.c_float(drop[0]), drop[1], drop[2], drop[3], dres_bf16 | (2 if r_bf16 else 0), stream()))
    torch.cuda.synchronize()
    mu, rstd = stats[:, :1].double(), stats[:, 1:].double()
    xh = (r.double() - mu) * rstd
    dyg = dy.double() * gam.double()
    dr = rstd * (dyg - dyg.mean(-1, keepdim=True) - xh * (dyg * xh).mean(-1, keepdim=True))
    keep = torch.ones(rows, E, dtype=torch.float64)
    ks = 1.0
    if drop_on:
        k0, thr = drop_key(*drop)
        keep = torch.from_numpy(drop_keep(k0, thr, np.arange(rows * E, dtype=np.int64)).reshape(rows, E)).double()
        ks = keep_scale(drop[0])
    dres = dr * keep * ks
    diag(f"== LayerNorm backward (dres {'bf16' if dres_bf16 else 'fp32'}, dropout {drop_on})")
    check("dr", dr_d, dr, False)
    check("masked branch gradient", dres_d, dres, bool(dres_bf16))
    check("dgamma", dg_d, (dy.double() * xh).sum(0), False)
    check("dbeta", db_d, dy.double().sum(0), False)
    check("fused bias gradient", dbias_d, dres.sum(0), False)        # column sums of the fp32 values, before the bf16 store


@pytest.mark.parametrize("M,drop_on,r_bf16", [(8 * 257, True, False), (70000 + 5, True, True), (3 * 257, False, True)])
def test_accumulate_then_layernorm_backward_kernel_equals_the_fp64_result(M, drop_on, r_bf16):
    """wst_ln_kernel<8,1,32,true,EPI_LNB> (csrc/wst.hip): dy = dr_in + dh W1 (K = 512, bf16 dh) followed, in the same launch, by the
    backward of LayerNorm-1: dr (fp32), the dropout-masked branch gradient (bf16) and the gamma / beta / out-proj-bias column sums
    (reduce-scatter butterfly over the token lanes, one atomic per feature and workgroup).  M = 70 005: several tiles per persistent
    workgroup and a ragged last tile (clamped lanes must not reach the column sums)."""
    lib = L.load()
    E, F = 256, 512
    g = torch.Generator().manual_seed(M + 9)
    dh, W = bf(rnd(g, M, F, scale=0.1)), bf(rnd(g, E, F, scale=0.05))
    dr_in, r, gam = rnd(g, M, E, scale=0.1), rnd(g, M, E), 1 + rnd(g, E, scale=0.1)
    mu = r.double().mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((r.double() - mu) ** 2).mean(-1, keepdim=True) + 1e-5)
    stats = torch.cat([mu, rstd], 1).float()
    drop = (0.1, 5, 1011, 2) if drop_on else (0.0, 0, 0, 0)
    d = lambda t, dt=torch.float32: t.to(DEV, dt).contiguous()
    X_d, W_d, Y_d, r_d, g_d, st_d = d(dh, torch.bfloat16), d(W, torch.bfloat16), d(dr_in), d(r), d(gam), d(stats)
    if r_bf16:          # pre-LayerNorm sums as the engine keeps them (bf16), statistics of the unrounded sums
        r_d = r_d.to(torch.bfloat16)
        r = r_d.float().cpu()
    dr_d = torch.full((M, E), float("nan"), device=DEV)
    dres_d = torch.full((M, E), float("nan"), dtype=torch.bfloat16, device=DEV)
    dg_d, db_d, dbias_d = torch.zeros(E, device=DEV), torch.zeros(E, device=DEV), torch.zeros(E, device=DEV)
    a = L.GGTestLinear()
    a.X, a.ldx, a.M, a.x_bf16 = X_d.data_ptr(), F, M, 1
    a.W, a.ldw = W_d.data_ptr(), F
    a.Y, a.ldy, a.y_bf16, a.y_rows = Y_d.data_ptr(), E, 0, -1
    a.N, a.K, a.accumulate = E, F, 1
    a.res, a.ldres, a.res_rows, a.res_bf16 = r_d.data_ptr(), E, M, int(r_bf16)
    a.ln_g, a.ln_y, a.ln_stats = g_d.data_ptr(), dr_d.data_ptr(), st_d.data_ptr()
    a.lnb_dres, a.lnb_dgamma, a.lnb_dbeta, a.lnb_dbias = dres_d.data_ptr(), dg_d.data_ptr(), db_d.data_ptr(), dbias_d.data_ptr()
    a.drop_p, a.drop_seed, a.drop_site, a.drop_call = drop
    a.drop_ld = E
    a.route = 0
    cls = C.c_int32(-1)
    Y_before = Y_d.clone()
    L.check(lib.gg_test_linear(C.byref(a), C.byref(cls), stream()))
    torch.cuda.synchronize()
    assert cls.value == 50, cls.value                          # the fused kernel ran
    assert torch.equal(Y_d, Y_before)                          # dr_in is read, not rewritten
    dy = dr_in.double() + dh.double() @ W.double().T
    mu, rstd = stats[:, :1].double(), stats[:, 1:].double()
    xh = (r.double() - mu) * rstd
    dyg = dy * gam.double()
    dr = rstd * (dyg - dyg.mean(-1, keepdim=True) - xh * (dyg * xh).mean(-1, keepdim=True))
    keep, ks = torch.ones(M, E, dtype=torch.float64), 1.0
    if drop_on:
        k0, thr = drop_key(*drop)
        keep = torch.from_numpy(drop_keep(k0, thr, np.arange(M * E, dtype=np.int64)).reshape(M, E)).double()
        ks = keep_scale(drop[0])
    dres = dr * keep * ks
    diag(f"== += then LayerNorm backward (M = {M}, dropout {drop_on})")
    check("dr", dr_d, dr, False)
    check("masked branch gradient", dres_d, dres, True)
    check("dgamma", dg_d, (dy * xh).sum(0), False)
    check("dbeta", db_d, dy.sum(0), False)
    check("fused bias gradient", dbias_d, dres.sum(0), False)


# ---- fused MLP head -------------------------------------------------------------------------------------------------------------
def lrelu(x, slope):
    return torch.where(x > 0, x, x * slope)


@pytest.mark.parametrize("rows,H,E,V,slope,score", [(768, 256, 256, 5000, 0.0, True), (70, 256, 256, 64, 0.2, True), (515, 256, 256, 256, 0.2, False),
                                                    (96, 64, 64, 40, 0.0, True), (33, 128, 32, 8, 0.2, False)])
def test_fused_mlp_head_equals_the_fp64_chain(rows, H, E, V, slope, score):
    """head_fwd_k / head_bwd_k (csrc/head.hip): the three-layer MLP head of R:226-231 (first layer = stored gene / latent part +
    conditioning part) and its data-path backward, each in one launch.  Reference: float64 on operands rounded to bf16 exactly where
    the launch-per-product route (gemm_small.hip) rounds them: every product's two operands.  Ragged row counts (clamped lanes),
    narrow widths (waves without a tile), the first-layer weight as a column slice of [H, V + E]."""
    lib = L.load()
    g = torch.Generator().manual_seed(rows + H)
    W1 = rnd(g, H, V + E, scale=0.08)
    b1, W2, b2 = rnd(g, H, scale=0.1), rnd(g, H, H, scale=0.08), rnd(g, H, scale=0.1)
    w3, b3 = rnd(g, H, scale=0.1), rnd(g, 1, scale=0.1)
    cvec, a1_pre = rnd(g, rows, E), rnd(g, rows, H)
    d = lambda t: t.to(DEV, torch.float32).contiguous()
    W1_d, b1_d, W2_d, b2_d, w3_d, b3_d, c_d = d(W1), d(b1), d(W2), d(b2), d(w3), d(b3), d(cvec)
    a1_d = d(a1_pre).clone()
    a2_d = torch.full((rows, H), float("nan"), device=DEV)
    out_rows = (2 * rows) // 3 if score else 0
    out_d = torch.full((rows,), float("nan"), device=DEV)
    W1c_ptr = C.c_void_p(W1_d.data_ptr() + 4 * V)
    L.check(lib.gg_test_head_fwd(rows, H, E, C.c_float(slope), W1c_ptr, V + E, P(b1_d), P(W2_d), P(b2_d), P(w3_d), P(b3_d), P(c_d), P(a1_d),
                                 P(a2_d), P(out_d) if score else None, out_rows, stream()))
    torch.cuda.synchronize()
    W1c = W1[:, V:]
    a1 = lrelu(a1_pre.double() + bf(cvec) @ bf(W1c).T + b1.double(), slope)
    a2 = lrelu(bf(a1.float()) @ bf(W2).T + b2.double(), slope)
    diag(f"== fused MLP head rows={rows} H={H} E={E} slope={slope}")
    check("a1", a1_d, a1, False)
    # a2's operand is the bf16 rounding of the kernel's own fp32 a1: where that sits on a rounding boundary the two runs round apart
    check("a2", a2_d, a2, False, 2e-4, 3e-3)
    if score:
        out = bf(a2.float()) @ bf(w3) + b3.double()
        got = out_d.cpu()
        assert torch.isnan(got[out_rows:]).all(), "scores of rows past out_rows must not be written"
        check("critic score", got[:out_rows], out[:out_rows], False, 2e-4, 3e-3)
    # backward on the kernel's own activations
    a1_k, a2_k = a1_d.cpu(), a2_d.cpu()
    if score:
        dout = rnd(g, rows, scale=0.5)
        dout_d = d(dout)
        dh2_d = torch.full((rows, H), float("nan"), device=DEV)
        dh2_pre = bf(dout)[:, None] * bf(w3)[None, :]
    else:
        dh2_pre = rnd(g, rows, H, scale=0.5).double()
        dh2_d = d(dh2_pre.float()).clone()
        dout_d = None
    dh1_d = torch.full((rows, H), float("nan"), device=DEV)
    dc_d = torch.full((rows, E), float("nan"), device=DEV)
    L.check(lib.gg_test_head_bwd(rows, H, E, C.c_float(slope), W1c_ptr, V + E, P(W2_d), P(w3_d), P(a1_d), P(a2_d), P(dout_d), P(dh2_d), P(dh1_d),
                                 P(dc_d), stream()))
    torch.cuda.synchronize()
    gate = lambda a: torch.where(a.double() > 0, torch.ones_like(a, dtype=torch.float64), torch.full_like(a, slope, dtype=torch.float64))
    dh2 = dh2_pre * gate(a2_k)
    dh1 = (bf(dh2.float()) @ bf(W2)) * gate(a1_k)
    dc = bf(dh1.float()) @ bf(W1c)
    check("dh2", dh2_d, dh2, False)
    check("dh1", dh1_d, dh1, False, 2e-4, 3e-3)
    check("dcond", dc_d, dc, False, 2e-4, 3e-3)


# ---- fused feed-forward block ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,keep_rows,drop_on", [(8 * 257, -1, True), (8 * 257, 3 * 257, True), (5 * 257 + 3, 0, False), (70000, 1000, True)])
def test_fused_feed_forward_block_equals_the_fp64_result(M, keep_rows, drop_on):
    """ffn_fused_kernel (csrc/ffn.hip): x2 = LN(x1 + drop(W2 drop(relu(W1 x1 + b1)) + b2)) in one launch; the hidden tile goes from the
    first product's accumulators straight into the second product's operand registers.  Reference on the values the MFMAs see:
    x1 rounded to bf16 for the first product (fp32 for the residual), h rounded to bf16 (it is both stored and multiplied as bf16).
    h (bf16): one rounding; r2, x2, statistics (fp32): 1e-4.  Rows >= keep_rows: h / r2 / statistics untouched.  M = 70 000: the
    persistent grid walks several tiles per workgroup (chunk pipeline across tile boundaries) and ends in a ragged tile."""
    lib = L.load()
    E, F = 256, 512
    g = torch.Generator().manual_seed(M + 5)
    x1 = rnd(g, M, E)
    W1, b1 = bf(rnd(g, F, E, scale=0.06)), rnd(g, F, scale=0.1)
    W2, b2 = bf(rnd(g, E, F, scale=0.05)), rnd(g, E, scale=0.1)
    gam, bet = 1 + rnd(g, E, scale=0.1), rnd(g, E, scale=0.1)
    drop = (0.1, 31, 1012, 1013, 4) if drop_on else (0.0, 0, 0, 0, 0)
    d = lambda t, dt=torch.float32: t.to(DEV, dt).contiguous()
    x_d, W1_d, W2T_d = d(x1), d(W1, torch.bfloat16), d(W2.T, torch.bfloat16)
    b1_d, b2_d, g_d, bt_d = d(b1), d(b2), d(gam), d(bet)
    h_d = torch.full((M, F), float("nan"), dtype=torch.bfloat16, device=DEV)
    r2_d = torch.full((M, E), float("nan"), device=DEV)
    y_d = torch.full((M, E), float("nan"), device=DEV)
    st_d = torch.full((M, 2), float("nan"), device=DEV)
    L.check(lib.gg_test_ffn_fused(P(x_d), M, P(W1_d), P(b1_d), P(W2T_d), P(b2_d), P(h_d), P(r2_d), keep_rows, P(g_d), P(bt_d), P(y_d), P(st_d),
                                  C.c_float(drop[0]), drop[1], drop[2], drop[3], drop[4], stream()))
    torch.cuda.synchronize()
    ks = keep_scale(drop[0]) if drop_on else 1.0
    h = (bf(x1) @ W1.T + b1.double()).clamp_min(0)
    if drop_on:
        k0, thr = drop_key(drop[0], drop[1], drop[2], drop[4])
        h = h * torch.from_numpy(drop_keep(k0, thr, np.arange(M * F, dtype=np.int64)).reshape(M, F)).double() * ks
    hb = bf(h.float())
    y = hb @ W2.T + b2.double()
    if drop_on:
        k0, thr = drop_key(drop[0], drop[1], drop[3], drop[4])
        y = y * torch.from_numpy(drop_keep(k0, thr, np.arange(M * E, dtype=np.int64)).reshape(M, E)).double() * ks
    r2 = x1.double() + y
    mu = r2.mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((r2 - mu) ** 2).mean(-1, keepdim=True) + 1e-5)
    x2 = (r2 - mu) * rstd * gam.double() + bet.double()
    kr = M if keep_rows < 0 else keep_rows
    diag(f"== fused feed-forward block M={M} keep_rows={keep_rows} dropout={drop_on}")
    # (an h element within fp32 accumulation noise of a bf16 rounding boundary rounds the other way than the float64 reference:
    # one such element moves an output by 2^-8 |h w2| ~ 2e-4 of the row scale - hence the max-norm bound of 2e-3 beside rel-L2 1e-4)
    check("x2 = LayerNorm(r2)", y_d, x2, False, 1e-4, 2e-3)
    if kr > 0:
        check("hidden activations (stored rows)", h_d[:kr], h[:kr], True)
        check("pre-LayerNorm sum (stored rows)", r2_d[:kr], r2[:kr], False, 1e-4, 2e-3)
        check("LayerNorm statistics", st_d[:kr].cpu(), torch.cat([mu, rstd], 1)[:kr], False)
    if kr < M:
        assert torch.isnan(h_d[kr:].float()).all() and torch.isnan(r2_d[kr:]).all() and torch.isnan(st_d[kr:]).all()


# ---- fused feed-forward block, round 4 (csrc/enc.hip) ------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", [0, 2, 4, 6])
@pytest.mark.parametrize("M,keep_rows,drop_on,r2_bf16,y_bf16", [
    (8 * 257, -1, True, 1, 1), (8 * 257, 3 * 257, True, 1, 0), (5 * 257 + 3, 0, False, 0, 0), (33, -1, True, 0, 1),
    (70000, 1024, True, 1, 1), (768 * 257, 512 * 257, True, 1, 1)])
def test_fused_feed_forward_stream_kernel_equals_the_fp64_result(M, keep_rows, drop_on, r2_bf16, y_bf16, variant):
    """ffn2_kernel (csrc/enc.hip): x2 = LN(x1 + drop(W2 drop(relu(W1 x1 + b1)) + b2)) in one launch with the weights streamed as MFMA
    fragments through an LDS ring (LDS-DMA, counted waits).  x1 is the bf16-stored LayerNorm output (operand AND residual), h is
    rounded to bf16 (stored and multiplied as bf16), r2 / x2 are stored in bf16 or fp32.  Rows >= keep_rows: h / r2 / statistics
    untouched.  M = 70 000 and 768 * 257: several sweeps per workgroup, a partial last sweep, a ragged last unit; M = 33: two units,
    one of them a single row."""
    if M > 100000 and variant != 0:
        pytest.skip("full-size case once")
    lib = L.load()
    E, F = 256, 512
    g = torch.Generator().manual_seed(M + 11)
    x1 = bf(rnd(g, M, E)).float()
    W1, b1 = bf(rnd(g, F, E, scale=0.06)).float(), rnd(g, F, scale=0.1)
    W2, b2 = bf(rnd(g, E, F, scale=0.05)).float(), rnd(g, E, scale=0.1)
    gam, bet = 1 + rnd(g, E, scale=0.1), rnd(g, E, scale=0.1)
    drop = (0.1, 31, 1012, 1013, 4) if drop_on else (0.0, 0, 0, 0, 0)
    d = lambda t, dt=torch.float32: t.to(DEV, dt).contiguous()
    x_d = d(x1, torch.bfloat16)
    Wcat = d(torch.cat([W1.reshape(-1), W2.reshape(-1)]))          # W2 follows W1 in one buffer (the hook takes offsets from W1)
    W1_d, W2_d = Wcat[:F * E], Wcat[F * E:]
    b1_d, b2_d, g_d, bt_d = d(b1), d(b2), d(gam), d(bet)
    h_d = torch.full((M, F), float("nan"), dtype=torch.bfloat16, device=DEV)
    r2_d = torch.full((M, E), float("nan"), dtype=torch.bfloat16 if r2_bf16 else torch.float32, device=DEV)
    y_d = torch.full((M, E), float("nan"), dtype=torch.bfloat16 if y_bf16 else torch.float32, device=DEV)
    st_d = torch.full((M, 2), float("nan"), device=DEV)
    wf = torch.empty(lib.gg_test_ffn2_frag_bytes(), dtype=torch.uint8, device=DEV)
    L.check(lib.gg_test_ffn2(P(x_d), M, P(W1_d), P(b1_d), P(W2_d), P(b2_d), P(h_d), P(r2_d), r2_bf16, keep_rows, P(g_d), P(bt_d), P(y_d), y_bf16,
                             P(st_d), C.c_float(drop[0]), drop[1], drop[2], drop[3], drop[4], P(wf), variant, stream()))
    torch.cuda.synchronize()
    ks = keep_scale(drop[0]) if drop_on else 1.0
    chunk = 16384
    diag(f"== fused feed-forward stream kernel M={M} keep_rows={keep_rows} dropout={drop_on} r2_bf16={r2_bf16} y_bf16={y_bf16} variant={variant}")
    kr = M if keep_rows < 0 else keep_rows
    W1d, W2d = W1.double(), W2.double()
    worst = {}
    for r0 in range(0, M, chunk):           # row blocks: the float64 reference of 197 376 x 512 would not fit comfortably in one piece
        r1 = min(M, r0 + chunk)
        xb = x1[r0:r1].double()
        h = (xb @ W1d.T + b1.double()).clamp_min(0)
        if drop_on:
            k0, thr = drop_key(drop[0], drop[1], drop[2], drop[4])
            h = h * torch.from_numpy(drop_keep(k0, thr, np.arange(r0 * F, r1 * F, dtype=np.int64)).reshape(r1 - r0, F)).double() * ks
        hb = bf(h.float())
        y = hb @ W2d.T + b2.double()
        if drop_on:
            k0, thr = drop_key(drop[0], drop[1], drop[3], drop[4])
            y = y * torch.from_numpy(drop_keep(k0, thr, np.arange(r0 * E, r1 * E, dtype=np.int64)).reshape(r1 - r0, E)).double() * ks
        r2 = xb + y
        mu = r2.mean(-1, keepdim=True)
        rstd = 1.0 / torch.sqrt(((r2 - mu) ** 2).mean(-1, keepdim=True) + 1e-5)
        x2 = (r2 - mu) * rstd * gam.double() + bet.double()
        def acc(name, got, want, bfo, l2b=None, mxb=None):
            got = got.detach().cpu()
            assert torch.isfinite(got.float()).all(), (name, r0)
            l2, mx = rel_l2(got, want), rel_max(got, want)
            w = worst.setdefault(name, [0.0, 0.0])
            w[0], w[1] = max(w[0], l2), max(w[1], mx)
            assert l2 <= (l2b or (2e-3 if bfo else 1e-4)) and mx <= (mxb or (2.0 ** -8 if bfo else 2e-4)), (name, r0, l2, mx)
        # (an h element within fp32 accumulation noise of a bf16 rounding boundary rounds the other way than the float64 reference:
        # one such element moves an output by 2^-8 |h w2| ~ 2e-4 of the row scale - hence the max-norm bound of 2e-3 beside rel-L2 1e-4)
        acc("x2 = LayerNorm(r2)", y_d[r0:r1], x2, bool(y_bf16), None, 2.0 ** -7 if y_bf16 else 2e-3)
        ka, kb = r0, min(r1, kr)
        if kb > ka:
            acc("hidden activations (stored rows)", h_d[ka:kb], h[: kb - ka], True)
            acc("pre-LayerNorm sum (stored rows)", r2_d[ka:kb], r2[: kb - ka], bool(r2_bf16), None, 2.0 ** -7 if r2_bf16 else 2e-3)
            acc("LayerNorm statistics", st_d[ka:kb], torch.cat([mu, rstd], 1)[: kb - ka], False)
    for k, v in worst.items():
        diag(f"   {k:44s} rel-L2 {v[0]:.2e}  max-norm {v[1]:.2e}")
    if kr < M:
        assert torch.isnan(h_d[kr:].float()).all() and torch.isnan(r2_d[kr:].float()).all() and torch.isnan(st_d[kr:]).all()


# ---- fused backward of the token-local chain of an encoder layer (csrc/enc.hip encb_kernel) -------------------------------------------
@pytest.mark.parametrize("M,drop_on", [(8 * 257, True), (5 * 257 + 3, False), (33, True), (70000, True), (512 * 257, True)])
def test_fused_encoder_layer_backward_equals_the_fp64_result(M, drop_on):
    """encb_kernel: dh = (dres2 W2) gated by the stored hidden activations -> dx1 = dr2 + dh W1 -> LayerNorm1 backward (dr1 over dr2 in place,
    masked bf16 branch gradient dres1, three per-feature sums) -> dctx = dres1 Wo.  Inputs dr2 / dres2 are what ln_bwd_v4_k (LayerNorm2
    backward) leaves.  Reference in float64 on the values the MFMAs see (dres2 / dh / dres1 as bf16); dropout mask regenerated on the host.
    M = 70 000 and 512 * 257 (the cfg3 backward shape): several sweeps per workgroup and a partial last sweep; M = 33: a one-row unit."""
    lib = L.load()
    E, F = 256, 512
    g = torch.Generator().manual_seed(M + 23)
    dr2 = rnd(g, M, E)
    dres2 = bf(rnd(g, M, E)).float()
    W1, W2, Wo = bf(rnd(g, F, E, scale=0.06)).float(), bf(rnd(g, E, F, scale=0.05)).float(), bf(rnd(g, E, E, scale=0.06)).float()
    r1 = bf(rnd(g, M, E)).float()
    g1 = 1 + rnd(g, E, scale=0.1)
    hid = bf((rnd(g, M, F)).clamp_min(0) * (torch.rand(M, F, generator=g) > 0.1)).float()        # ReLU output with dropped entries
    drop = (0.1, 31, 1013, 1011, 4) if drop_on else (0.0, 0, 0, 0, 0)
    ks = keep_scale(drop[0]) if drop_on else 1.0
    mu = r1.double().mean(-1, keepdim=True)
    st1 = torch.cat([mu, 1.0 / torch.sqrt(((r1.double() - mu) ** 2).mean(-1, keepdim=True) + 1e-5)], 1)
    d = lambda t, dt=torch.float32: t.to(DEV, dt).contiguous()
    dx_d = d(dr2)
    Wcat = d(torch.cat([W1.reshape(-1), W2.reshape(-1), Wo.reshape(-1)]))
    dres2_d, r1_d, h_d = d(dres2, torch.bfloat16), d(r1, torch.bfloat16), d(hid, torch.bfloat16)
    st1_d, g1_d = d(st1.float()), d(g1)
    nan16 = lambda *s: torch.full(s, float("nan"), dtype=torch.bfloat16, device=DEV)
    dh_d, dres1_d, dctx_d = nan16(M, F), nan16(M, E), nan16(M, E)
    cs_d = torch.zeros(3, E, device=DEV)
    wf = torch.empty(lib.gg_test_enc_bwd_frag_bytes(), dtype=torch.uint8, device=DEV)
    L.check(lib.gg_test_enc_bwd(P(dx_d), M, P(Wcat), P(dres2_d), P(h_d), P(r1_d), P(st1_d), P(g1_d), P(dh_d), P(dres1_d), P(dctx_d), P(cs_d),
                                C.c_float(drop[0]), drop[1], drop[3], drop[4], P(wf), stream()))
    torch.cuda.synchronize()
    diag(f"== fused encoder-layer backward M={M} dropout={drop_on}")
    st1f = st1.float().double()
    W1d, W2d, Wod = W1.double(), W2.double(), Wo.double()
    sums = torch.zeros(3, E, dtype=torch.float64)
    worst = {}

    def acc(name, got, want, bfo, l2b=None, mxb=None):
        got = got.detach().cpu()
        assert torch.isfinite(got.float()).all(), name
        l2, mx = rel_l2(got, want), rel_max(got, want)
        w = worst.setdefault(name, [0.0, 0.0])
        w[0], w[1] = max(w[0], l2), max(w[1], mx)
        assert l2 <= (l2b or (2e-3 if bfo else 1e-4)) and mx <= (mxb or (2.0 ** -8 if bfo else 2e-4)), (name, l2, mx)
    chunk = 16384
    for a in range(0, M, chunk):
        b = min(M, a + chunk)
        m1 = 1.0
        if drop_on:
            k0, thr = drop_key(drop[0], drop[1], drop[3], drop[4])
            m1 = torch.from_numpy(drop_keep(k0, thr, np.arange(a * E, b * E, dtype=np.int64)).reshape(b - a, E)).double() * ks
        dh = (dres2[a:b].double() @ W2d) * (hid[a:b].double() > 0) * ks
        dx1 = dr2[a:b].double() + bf(dh.float()) @ W1d
        st = st1f[a:b]
        xh = (r1[a:b].double() - st[:, :1]) * st[:, 1:]
        gy = dx1 * g1.double()
        dr1 = st[:, 1:] * (gy - gy.mean(-1, keepdim=True) - xh * (gy * xh).mean(-1, keepdim=True))
        dres1 = dr1 * m1
        dctx = bf(dres1.float()) @ Wod
        for i, v in enumerate(((dx1 * xh).sum(0), dx1.sum(0), dres1.sum(0))):
            sums[i] += v
        acc("dh (bf16)", dh_d[a:b], dh, True, None, 2.0 ** -7)
        acc("dres1 (bf16)", dres1_d[a:b], dres1, True, None, 2.0 ** -7)
        acc("dctx (bf16)", dctx_d[a:b], dctx, True, None, 2.0 ** -7)
        # dr1 carries the bf16 rounding of dh through a 512-term product: 2^-9 / sqrt(512) per term relative to the branch, ~1e-4 of dr1
        acc("dr1 over dr2 (fp32)", dx_d[a:b], dr1, False, 5e-4, 2e-3)
    for k, v in worst.items():
        diag(f"   {k:44s} rel-L2 {v[0]:.2e}  max-norm {v[1]:.2e}")
    for i, nm in enumerate(("dgamma1", "dbeta1", "d out_proj.bias")):
        check(nm, cs_d[i], sums[i], False, 2e-3, 4e-3)      # sums of M terms of either sign whose inputs carry the bf16 rounding of dh
