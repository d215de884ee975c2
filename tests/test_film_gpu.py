"""Host-side mirror of the FiLM-only sibling (gemm_gan_amd/film.py <-> the reference's src/conditional_gan_film.py)
against the golden vectors produced by that reference file (tests/golden/film_*.npz, oracle/make_golden.py): class
surface, state_dict keys, a full train() with the recorded noise, eval-mode inference, fit() + checkpoints."""
import numpy as np
import pytest
import torch

from gemm_gan_amd import film, img_transformer
from golden_util import Golden, comparable
from gpu_util import Checker

pytestmark = pytest.mark.gpu
FILM_FIXTURES = ["film_P1", "film_P7", "img_P9"]      # the 4-argument sibling files: FiLM-only and image-transformer


PREC = None


@pytest.fixture(params=[None, "f32"], ids=["default-precision", "f32"], autouse=True)
def facade_precision(request):
    """Every facade test runs twice: with the drop-in default (`precision` not given: bf16x3 since round 4) and in the exact f32 mode."""
    global PREC
    PREC = request.param
    yield
    PREC = None


def build(g: Golden, opt="rms_prop", **kw):
    d = g.dims
    mod = img_transformer if g.variant == "img" else film
    w = mod.WGAN_GP(d["G"], d["L"], d["E"], [d["H"], d["H"], d["G"]], [d["H"], d["H"], 1], text_embedding_dims=d["Dt"],
                    patches_embedding_dims=d["Dp"], negative_slope=g.slope, optimizer=opt, n_critic=d["n_critic"],
                    dropout=0.0, device="cuda:0", **({} if PREC is None else {"precision": PREC}), **kw)
    w.build_WGAN_GP()
    w.init_train()
    return w


@pytest.mark.parametrize("name", FILM_FIXTURES)
def test_state_dict_keys_and_golden_train_step(name):
    g = Golden(name)
    w = build(g)
    assert set(w.gen.state_dict()) == set(g.group("init_gen"))          # incl. the dead template layer, as upstream
    assert set(w.disc.state_dict()) == set(g.group("init_disc"))
    w.gen.load_state_dict(g.state("init_gen"))
    w.disc.load_state_dict(g.state("init_disc"))
    x, text, text_pad, patches, patch_pad = (t.cuda() for t in g.inputs())
    ck = Checker(f"film mirror {name}", 1e-3)
    # eval-mode inference through the reference's forward signature (x, text_embedding, patches, padding_mask)
    w.gen.eval()
    ck.check("generator forward (eval)", w.gen(g.t("infer/z").cuda(), text[:, 0, :], patches, patch_pad), g.z["infer/x_gen"])
    w.disc.train()
    ck.check("critic forward (train)", w.disc(x, text[:, 0, :], patches, patch_pad), g.group("disc_fwd")["out"])
    z_all = g.t("step_rms_prop/z").cuda().contiguous()
    alpha_all = g.t("step_rms_prop/alpha").cuda().reshape(z_all.shape[0] - 1, -1).contiguous()
    w.train_with_noise(x, text.contiguous(), text_pad, patches.contiguous(), patch_pad, z_all, alpha_all)
    # several normalised-gradient steps: same gates as tests/test_engine_golden_gpu.py::test_full_train_step
    ck.check("d_batch_loss", w.d_batch_loss, g.z["step_rms_prop/d_batch_loss"])
    ck.check("gen_loss", np.array([float(w.gen_loss)]), np.array([float(g.z["step_rms_prop/gen_loss"])]))
    for role, net in (("gen", w.gen), ("disc", w.disc)):
        sd = net.state_dict()
        steps = 1 if role == "gen" else g.dims["n_critic"]
        for n, ref in g.group(f"step_rms_prop/post_{role}").items():
            if n.startswith("patches_transformer_layer."):
                continue
            keep = comparable(n, sd[n], g.dims["E"])
            ck.check_post(f"post {role} {n}", sd[n].detach().cpu().numpy().reshape(-1)[keep], ref.reshape(-1)[keep],
                          g.z[f"init_{role}/{n}"].reshape(-1)[keep], "rms_prop", 5e-4, steps)
    ck.done()


def test_reference_call_signatures_and_fit(tmp_path):
    g = Golden("film_P7")
    d = g.dims
    w = build(g, results_dire=str(tmp_path))
    x, text, text_pad, patches, patch_pad = g.inputs()
    emb = text[:, 0, :]
    z = torch.randn(d["B"], d["L"])
    w.train_disc(x, z, emb, patches, patch_pad)
    w.train_gen(z, emb, patches, patch_pad)
    w.train(x, emb, patches, patch_pad)
    assert np.isfinite(w.d_batch_loss).all() and np.isfinite(w.g_batch_loss).all()
    x_real, x_gen = w.generate_samples(x, emb, patches, patch_pad)
    assert tuple(x_gen.shape) == (d["B"], d["G"]) and torch.isfinite(x_gen).all()
    w.freq_compute_test = 1
    loader = [(emb, x, patches, patch_pad)] * 2                 # F:632-635 batch layout
    w2 = film.WGAN_GP(d["G"], d["L"], d["E"], [d["H"], d["H"], d["G"]], [d["H"], d["H"], 1], text_embedding_dims=d["Dt"],
                      patches_embedding_dims=d["Dp"], n_critic=2, dropout=0.1, device="cuda:0", results_dire=str(tmp_path),
                      freq_compute_test=1)
    hist = w2.fit(loader, epochs=2)
    assert len(hist["d loss"]) == 2 and all(np.isfinite(v) for v in hist["d loss"] + hist["g loss"])
    sd = torch.load(tmp_path / "generator_last_epoch.pt")
    assert set(sd) == set(g.group("init_gen"))                  # a checkpoint the reference's load_state_dict accepts
    # F:322-345 gradient_penalty(real, fake, text_embedding, patches, padding_mask): real == fake makes the interpolate
    # independent of alpha, so the value equals the penalty of the critic at x, whatever torch.rand draws
    gp = w.gradient_penalty(x, x, emb, patches, patch_pad)
    gp2 = w.gradient_penalty(x, x, emb, patches, patch_pad)
    assert gp.dim() == 0 and float(gp) >= 0.0 and abs(float(gp) - float(gp2)) <= 1e-5 * max(1.0, float(gp))


def test_vanilla_mirror_golden_train_step(tmp_path):
    """gemm_gan_amd.vanilla <-> src/vanilla_gan_unconditional.py (BASELINE configs[0]): names, shapes (the engine's padded
    first-layer weight shows up as the reference's [H, V] block), a full train() with the recorded noise, inference, fit()."""
    from gemm_gan_amd import vanilla
    g = Golden("vanilla_G60")
    d = g.dims
    w = vanilla.WGAN_GP_nocond(d["G"], d["L"], [], [d["H"], d["H"], d["G"]], [d["H"], d["H"], 1], negative_slope=g.slope,
                               n_critic=d["n_critic"], device="cuda:0", results_dire=str(tmp_path))
    w.build_WGAN_GP_nocond()
    w.init_train()
    for net, prefix in ((w.gen, "init_gen"), (w.disc, "init_disc")):
        ref = g.group(prefix)
        sd = net.state_dict()
        assert set(sd) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)
        net.load_state_dict(g.state(prefix))
    x = g.inputs()[0].cuda()
    ck = Checker("vanilla mirror", 1e-3)
    w.gen.eval()
    ck.check("generator forward (eval)", w.gen(g.t("infer/z").cuda()), g.z["infer/x_gen"])
    w.disc.train()
    ck.check("critic forward", w.disc(x), g.group("disc_fwd")["out"])
    z_all = g.t("step_rms_prop/z").cuda().contiguous()
    alpha_all = g.t("step_rms_prop/alpha").cuda().reshape(z_all.shape[0] - 1, -1).contiguous()
    w.train_with_explicit_noise(x, z_all, alpha_all)
    ck.check("d_batch_loss", w.d_batch_loss, g.z["step_rms_prop/d_batch_loss"])
    ck.check("gen_loss", np.array([float(w.gen_loss)]), np.array([float(g.z["step_rms_prop/gen_loss"])]))
    for role, net in (("gen", w.gen), ("disc", w.disc)):
        sd = net.state_dict()
        steps = 1 if role == "gen" else d["n_critic"]
        for n, ref in g.group(f"step_rms_prop/post_{role}").items():
            ck.check_post(f"post {role} {n}", sd[n].detach().cpu().numpy().reshape(-1), ref.reshape(-1),
                          g.z[f"init_{role}/{n}"].reshape(-1), "rms_prop", 5e-4, steps)
    # the padded columns of the engine's first-layer weights are still exactly zero after six optimiser steps
    for role in (0, 1):
        name = [k for k in w.engine.layout[role] if k.endswith(".0.0.weight")][0]
        off, numel, shape = w.engine.layout[role][name]
        full = w.engine.flat[role]["w"][off:off + numel].view(shape)
        assert float(full[:, shape[1] - w.engine.cfg.embedding_dims:].abs().max()) == 0.0
    ck.done()
    w.train(x)
    w.train_disc(x, torch.randn(d["B"], d["L"]))
    w.train_gen(torch.randn(d["B"], d["L"]))
    x_real, x_gen = w.generate_samples(x)
    assert tuple(x_gen.shape) == (d["B"], d["G"]) and torch.isfinite(x_gen).all()
    w2 = vanilla.WGAN_GP_nocond(d["G"], d["L"], [], [d["H"], d["H"], d["G"]], [d["H"], d["H"], 1], n_critic=2, device="cuda:0",
                                results_dire=str(tmp_path), freq_compute_test=1)
    hist = w2.fit([(x.cpu(),)] * 2, epochs=2)
    assert len(hist["d loss"]) == 2 and all(np.isfinite(v) for v in hist["d loss"] + hist["g loss"])
    assert set(torch.load(tmp_path / "generator_last_epoch.pt")) == set(g.group("init_gen"))
