"""HIP engine vs golden vectors produced by the REAL reference (tests/golden/*.npz): stage
activations, losses, GP gradient, every parameter gradient, post-step parameters for the three
optimisers.  Tolerance: 1e-3 relative (BASELINE.json north_star), fp32, dropout 0."""
import numpy as np
import pytest
import torch

from golden_util import FIXTURES, Golden, comparable
from gemm_gan_amd import _lib as L
from gpu_util import Checker, dev, engine_from_cfg

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("parity_mode")]      # every test in f32 AND in bf16x3 mode
TOL = 1e-3


def well_conditioned(grad):
    """The first RMSprop step is -lr*g/(0.1|g|+1e-8): for |g| <~ 1e-6 the update depends on the VALUE of a
    gradient that is itself rounding noise in fp32 (any summation order moves it), so post-step parameters
    are compared where the golden gradient is at least 1e-5 of its tensor's scale and above 1e-6 absolute."""
    a = np.abs(np.asarray(grad, dtype=np.float64)).reshape(-1)
    return (a > 1e-5 * max(a.max(), 1e-30)) & (a > 1e-6)


def make(g: Golden, opt="rms_prop"):
    cfg = g.cfg(opt)
    d = g.dims
    eng = engine_from_cfg(cfg, d["B"], d["P"], d["T"], dropout=0.0, optimizer=opt)
    eng.load_state(L.ROLE_GENERATOR, g.state("init_gen"))
    eng.load_state(L.ROLE_CRITIC, g.state("init_disc"))
    x, text, text_pad, patches, patch_pad = dev(*g.inputs())
    return eng, x, text, text_pad, patches, patch_pad


@pytest.mark.parametrize("name", FIXTURES)
def test_forward_stages(name):
    g = Golden(name)
    d = g.dims
    eng, x, text, text_pad, patches, patch_pad = make(g)
    ck = Checker(f"golden forward {name}", TOL)
    out = eng.forward(L.ROLE_CRITIC, x, patches, patch_pad, text, text_pad, train=True)
    ref = g.group("disc_fwd")
    B, P, S, E, Dp = d["B"], d["P"], d["P"] + 1, d["E"], d["Dp"]
    if g.variant == "vanilla":       # no conditioning stack: outputs only
        ck.check("critic score", out, ref["out"])
        xg = eng.forward(L.ROLE_GENERATOR, g.t("gen_fwd/z").cuda(), patches, patch_pad, text, text_pad, train=True)
        ck.check("generated genes", xg, g.z["gen_fwd/out"])
        xe = eng.forward(L.ROLE_GENERATOR, g.t("infer/z").cuda(), patches, patch_pad, text, text_pad, train=False)
        ck.check("generate_samples (eval)", xe, g.z["infer/x_gen"])
        ck.done()
        return
    if g.variant != "img":
        gb = eng.debug_buffer("D.gb").view(B, 2 * Dp)
        ck.check("film gamma", gb[:, :Dp], np.tanh(ref["film_pre"][:, :Dp]))
        ck.check("film beta", gb[:, Dp:], np.clip(ref["film_pre"][:, Dp:], -5, 5))
    if g.variant == "xattn_film":
        ck.check("text encoder", eng.debug_buffer("D.tok").view(B, d["T"], E), ref["text_enc"])
    ck.check("patch encoder (FiLM fused)", eng.debug_buffer("D.x0").view(B, S, E)[:, 1:], ref["patch_emb"])
    ck.check("encoder layer 0", eng.debug_buffer("D.L0.x2").view(B, S, E), ref["enc_layer0"])
    ck.check("encoder layer 1", eng.debug_buffer("D.L1.x2").view(B, S, E), ref["enc_layer1"])
    if g.variant == "xattn_film":
        ck.check("T2I attention", eng.debug_buffer("D.t2i_out").view(B, E), ref["t2i"])
        ck.check("I2T attention", eng.debug_buffer("D.i2t_out").view(B, E), ref["i2t"])
    ck.check("critic score", out, ref["out"])
    xg = eng.forward(L.ROLE_GENERATOR, g.t("gen_fwd/z").cuda(), patches, patch_pad, text, text_pad, train=True)
    ck.check("generated genes", xg, g.z["gen_fwd/out"])
    xe = eng.forward(L.ROLE_GENERATOR, g.t("infer/z").cuda(), patches, patch_pad, text, text_pad, train=False)
    ck.check("generate_samples (eval)", xe, g.z["infer/x_gen"])
    ck.done()


@pytest.mark.parametrize("name", FIXTURES)
def test_critic_iteration(name):
    g = Golden(name)
    eng, x, text, text_pad, patches, patch_pad = make(g)
    ck = Checker(f"golden critic iteration {name}", TOL)
    eng.critic_backward(x, g.t("critic1/z").cuda(), g.t("critic1/alpha").cuda(), patches, patch_pad, text, text_pad)
    l = eng.losses.tolist()
    los = g.z["critic1/losses"]      # total, d_loss, d_real, d_fake
    ck.check("losses (d_real, d_fake)", np.array([l[0], l[1]]), los[2:4])
    ck.check("total loss", np.array([l[0] + l[1] + 10.0 * l[2]]), los[0:1])
    ck.check("grad_x_hat", eng.debug_buffer("gp_grad").view(g.dims["B"], g.dims["G"]), g.z["critic1/grad_x_hat"])
    grads = eng.state(L.ROLE_CRITIC, "g")
    for n, r in g.group("critic1/grad").items():
        ck.check("grad " + n, grads[n], r)
    eng.critic_apply(1.0)
    post = eng.state(L.ROLE_CRITIC, "w")
    init = g.group("init_disc")
    for n, r in g.group("critic1/post_disc").items():
        if n not in post:
            continue          # dead patches_transformer_layer.* template copy (never trained, not in the engine)
        keep = comparable(n, post[n], g.dims["E"])
        ck.check_post("post-step " + n, post[n].reshape(-1).cpu().numpy()[keep], r.reshape(-1)[keep], init[n].reshape(-1)[keep],
                      "rms_prop", 5e-4, 1)
        wc = keep & well_conditioned(g.z["critic1/grad/" + n])          # where the gradient is not noise: elementwise
        ck.check("post-step (well-conditioned) " + n, post[n].reshape(-1).cpu().numpy()[wc], r.reshape(-1)[wc])
    ck.done()


@pytest.mark.parametrize("name", FIXTURES)
def test_generator_iteration(name):
    g = Golden(name)
    eng, x, text, text_pad, patches, patch_pad = make(g)
    ck = Checker(f"golden generator iteration {name}", TOL)
    eng.generator_backward(g.t("gen1/z").cuda(), patches, patch_pad, text, text_pad)
    ck.check("g_loss", np.array([eng.losses.tolist()[3]]), np.array([float(g.z["gen1/loss"])]))
    grads = eng.state(L.ROLE_GENERATOR, "g")
    for n, r in g.group("gen1/grad").items():
        ck.check("grad " + n, grads[n], r)
    eng.generator_apply(1.0)
    post = eng.state(L.ROLE_GENERATOR, "w")
    init = g.group("init_gen")
    for n, r in g.group("gen1/post_gen").items():
        if n not in post:
            continue
        keep = comparable(n, post[n], g.dims["E"])
        ck.check_post("post-step " + n, post[n].reshape(-1).cpu().numpy()[keep], r.reshape(-1)[keep], init[n].reshape(-1)[keep],
                      "rms_prop", 5e-4, 1)
        wc = keep & well_conditioned(g.z["gen1/grad/" + n])
        ck.check("post-step (well-conditioned) " + n, post[n].reshape(-1).cpu().numpy()[wc], r.reshape(-1)[wc])
    ck.done()


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("opt", ["rms_prop", "adam", "adamw"])
def test_full_train_step(name, opt):
    g = Golden(name)
    if f"step_{opt}/z" not in g.z.files:
        pytest.skip("the reference file of this fixture has no such optimiser branch")
    eng, x, text, text_pad, patches, patch_pad = make(g, opt)
    ck = Checker(f"golden full train() {name} {opt}", TOL)
    z_all = g.t(f"step_{opt}/z").cuda().contiguous()
    alpha_all = g.t(f"step_{opt}/alpha").cuda().reshape(z_all.shape[0] - 1, -1).contiguous()
    eng.train_step(x, patches, patch_pad, text, text_pad, z_all, alpha_all)
    l = eng.losses.tolist()
    # Multi-step gate: 1e-3 on the losses like every single-iteration quantity (tests/test_numpy_oracle.py::
    # test_multi_step_conditioning: rounding-level gradient noise moves them by < 1e-4); post-step parameters, EVERY element,
    # through Checker.check_post.
    ck.check("d_batch_loss", np.array([l[0] + l[1], l[0], l[1]]), g.z[f"step_{opt}/d_batch_loss"])
    ck.check("gen_loss", np.array([l[3]]), np.array([float(g.z[f"step_{opt}/gen_loss"])]))
    for role, prefix in ((L.ROLE_GENERATOR, "gen"), (L.ROLE_CRITIC, "disc")):
        post = eng.state(role, "w")
        init = g.group(f"init_{prefix}")
        steps = 1 if prefix == "gen" else g.dims["n_critic"]
        for n, r in g.group(f"step_{opt}/post_{prefix}").items():
            if n not in post:
                continue
            keep = comparable(n, post[n], g.dims["E"])
            ck.check_post(f"post {prefix} {n}", post[n].reshape(-1).cpu().numpy()[keep], r.reshape(-1)[keep],
                          init[n].reshape(-1)[keep], opt, 5e-4, steps)
    ck.done()
