"""Pin oracle/torch_oracle.py against golden vectors produced by the REAL reference
(oracle/make_golden.py).  CPU only; the whole file runs in a few seconds."""
import numpy as np
import pytest
import torch

from golden_util import FIXTURES, Golden, rel_err
from oracle.torch_oracle import live_parameters

TOL = 1e-6   # same stock torch modules, same thread count -> expected bitwise; allow reassociation


@pytest.fixture(autouse=True)
def _one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


@pytest.mark.parametrize("name", FIXTURES)
def test_state_dict_keys_and_dead_layer(name):
    g = Golden(name)
    tr = g.trainer()
    assert set(tr.gen.state_dict()) == set(g.group("init_gen"))
    assert set(tr.disc.state_dict()) == set(g.group("init_disc"))
    dead = [k for k in g.group("critic1/grad_none")]
    n_dead = {"xattn_film": 12, "vanilla": 0}.get(g.variant, 6)       # the template layer of the bias-free encoder has weights only
    assert len(dead) == n_dead and all(k.startswith("patches_transformer_layer.") for k in dead)
    assert not any(n.startswith("patches_transformer_layer.") for n, _ in live_parameters(tr.disc))


@pytest.mark.parametrize("name", FIXTURES)
def test_forward_stages(name):
    g = Golden(name)
    tr = g.trainer()
    x, text, text_pad, patches, patch_pad = g.inputs()
    tr.disc.train()
    taps = {}
    with torch.no_grad():
        out = tr.disc(x, patches, patch_pad, text, text_pad, taps)
    ref = g.group("disc_fwd")
    Dp = g.dims["Dp"]
    if g.variant == "vanilla":
        taps["seq0"] = taps["enc"] = None
    if g.variant not in ("img", "vanilla"):
        assert rel_err(taps["gamma"], np.tanh(ref["film_pre"][:, :Dp])) < TOL
        assert rel_err(taps["beta"], np.clip(ref["film_pre"][:, Dp:], -5, 5)) < TOL
    if g.variant != "vanilla":
        assert rel_err(taps["seq0"][:, 1:], ref["patch_emb"]) < TOL
        assert rel_err(taps["enc"], ref["enc_layer1"]) < TOL
    if g.variant == "xattn_film":
        assert rel_err(taps["text_enc"], ref["text_enc"]) < TOL
        assert rel_err(taps["t2i"], ref["t2i"]) < TOL
        assert rel_err(taps["i2t"], ref["i2t"]) < TOL
    assert rel_err(taps["mlp_pre0"], ref["mlp_pre0"]) < TOL
    assert rel_err(taps["mlp_pre1"], ref["mlp_pre1"]) < TOL
    assert rel_err(out, ref["out"]) < TOL
    tr.gen.train()
    with torch.no_grad():
        xg = tr.gen(g.t("gen_fwd/z"), patches, patch_pad, text, text_pad)
    assert rel_err(xg, g.z["gen_fwd/out"]) < TOL
    # eval-mode inference takes torch's nested-tensor fast path: ~3e-8 abs apart
    assert rel_err(tr.generate(g.t("infer/z"), g.cond()), g.z["infer/x_gen"]) < 1e-5


@pytest.mark.parametrize("name", FIXTURES)
def test_critic_iteration(name):
    g = Golden(name)
    tr = g.trainer()
    x = g.inputs()[0]
    r = tr.critic_iteration(x, g.t("critic1/z"), g.t("critic1/alpha"), g.cond())
    los = g.z["critic1/losses"]
    assert rel_err([r["total"].item(), r["d_loss"].item(), r["d_real"].item(), r["d_fake"].item()], los) < TOL
    assert rel_err(r["grad_x_hat"].detach(), g.z["critic1/grad_x_hat"]) < TOL
    if g.variant == "xattn_film":     # the sibling files do not clip
        assert abs(float(r["grad_norm_total"]) - float(g.z["critic1/grad_total_norm"])) < 1e-5 * float(g.z["critic1/grad_total_norm"])
    else:
        assert "grad_norm_total" not in r
    for n, ref in g.group("critic1/grad").items():
        assert rel_err(r["grads"][n], ref) < 1e-5, n
    for n in g.group("critic1/grad_none"):
        assert r["grads"][n] is None
    post = tr.disc.state_dict()
    for n, ref in g.group("critic1/post_disc").items():
        assert rel_err(post[n].reshape(-1), ref.reshape(-1)) < 1e-5, n


@pytest.mark.parametrize("name", FIXTURES)
def test_generator_iteration(name):
    g = Golden(name)
    tr = g.trainer()
    r = tr.generator_iteration(g.t("gen1/z"), g.cond())
    assert abs(r["g_loss"].item() - float(g.z["gen1/loss"])) < 1e-6 * max(1.0, abs(float(g.z["gen1/loss"])))
    for n, ref in g.group("gen1/grad").items():
        assert rel_err(r["grads"][n], ref) < 1e-5, n
    post = tr.gen.state_dict()
    for n, ref in g.group("gen1/post_gen").items():
        assert rel_err(post[n].reshape(-1), ref.reshape(-1)) < 1e-5, n


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("opt", ["rms_prop", "adam", "adamw"])
def test_full_train_step(name, opt):
    g = Golden(name)
    if f"step_{opt}/z" not in g.z.files:
        pytest.skip("the reference file of this fixture has no such optimiser branch")
    tr = g.trainer(opt)
    x, text, text_pad, patches, patch_pad = g.inputs()
    zs = list(g.t(f"step_{opt}/z"))
    al = list(g.t(f"step_{opt}/alpha"))
    out = tr.train_step(x, text, text_pad, patches, patch_pad, zs, al)
    c = out["critic"]
    assert rel_err([c["d_loss"].item(), c["d_real"].item(), c["d_fake"].item()], g.z[f"step_{opt}/d_batch_loss"]) < 1e-5
    assert abs(out["gen"]["g_loss"].item() - float(g.z[f"step_{opt}/gen_loss"])) < 1e-5
    for role, net in (("gen", tr.gen), ("disc", tr.disc)):
        sd = net.state_dict()
        for n, ref in g.group(f"step_{opt}/post_{role}").items():
            assert rel_err(sd[n].reshape(-1), ref.reshape(-1)) < 2e-5, (role, n)
