"""The MAIN facade (gemm_gan_amd.WGAN_GP, the mirror of src/conditional_gan_cross_attention_with_film.py's class) driven through
its public per-iteration methods on the GPU - train_disc (R:376), train_gen (R:425), generate_samples (R:601),
gradient_penalty (R:351), generate_samples_all (R:561-599) - against the golden values recorded from the real reference.
The methods draw alpha / z with the reference's own torch.rand / torch.normal calls; the test substitutes the recorded draws."""
import os

import numpy as np
import pytest
import torch

import gemm_gan_amd as gga
from golden_util import XATTN_FIXTURES, Golden, comparable
from gpu_util import Checker

pytestmark = pytest.mark.gpu


class Replay:
    """torch.rand / torch.normal return the recorded draws (the reverse of oracle/make_golden.py's Recorder)."""

    def __init__(self, rand=None, normal=None):
        self.rand, self.normal = list(rand or []), list(normal or [])

    def __enter__(self):
        self._r, self._n = torch.rand, torch.normal

        def rand(*a, **k):
            return self.rand.pop(0).to(k.get("device", "cpu"))

        def normal(*a, **k):
            return self.normal.pop(0).to(k.get("device", "cpu"))
        torch.rand, torch.normal = rand, normal
        return self

    def __exit__(self, *exc):
        torch.rand, torch.normal = self._r, self._n


PREC = None


@pytest.fixture(params=[None, "f32"], ids=["default-precision", "f32"], autouse=True)
def facade_precision(request):
    """Every facade test runs twice: with the drop-in default (`precision` not given: bf16x3 since round 4) and in the exact f32 mode."""
    global PREC
    PREC = request.param
    yield
    PREC = None


def build(g: Golden, optimizer="rms_prop"):
    d = g.dims
    w = gga.WGAN_GP(d["G"], d["L"], d["E"], [d["H"], d["H"], d["G"]], [d["H"], d["H"], 1], text_embedding_dims=d["Dt"],
                    patches_embedding_dims=d["Dp"], negative_slope=g.slope, optimizer=optimizer, n_critic=d["n_critic"],
                    dropout=0.0, device="cuda:0", **({} if PREC is None else {"precision": PREC}))
    assert w.precision == (PREC or "bf16x3")
    w.build_WGAN_GP()
    w.init_train()
    w.gen.load_state_dict(g.state("init_gen"))
    w.disc.load_state_dict(g.state("init_disc"))
    return w


def check_post(ck, tag, net, want_group, grad_group, E):
    sd = net.state_dict()
    for n, r in want_group.items():
        if n.startswith("patches_transformer_layer."):
            continue
        keep = comparable(n, sd[n], E)
        a = np.abs(np.asarray(grad_group[n], dtype=np.float64)).reshape(-1)
        keep &= ((a > 1e-5 * max(a.max(), 1e-30)) & (a > 1e-6))             # see test_engine_golden_gpu.well_conditioned
        ck.check(f"{tag} {n}", sd[n].reshape(-1).cpu().numpy()[keep], r.reshape(-1)[keep])


@pytest.mark.parametrize("name", XATTN_FIXTURES)
def test_train_disc_train_gen_generate_samples(name):
    g = Golden(name)
    d = g.dims
    x, text, text_pad, patches, patch_pad = g.inputs()
    ck = Checker(f"facade {name}", 1e-3)
    w = build(g)
    with Replay(rand=[g.t("critic1/alpha")]):
        w.train_disc(x, g.t("critic1/z"), text, text_pad, patches, patch_pad)
    los = g.z["critic1/losses"]                       # disc_loss, then d_batch_loss = (d_loss, d_real, d_fake)
    ck.check("d_batch_loss", w.d_batch_loss, los[1:4])
    ck.check("disc_loss", np.array([w.disc_loss.item()]), los[0:1])
    check_post(ck, "post train_disc", w.disc, g.group("critic1/post_disc"), g.group("critic1/grad"), d["E"])

    w = build(g)
    w.train_gen(g.t("gen1/z"), text, text_pad, patches, patch_pad)
    ck.check("gen_loss", np.array([w.gen_loss.item(), w.g_batch_loss[0]]), np.array([float(g.z["gen1/loss"])] * 2))
    check_post(ck, "post train_gen", w.gen, g.group("gen1/post_gen"), g.group("gen1/grad"), d["E"])

    w = build(g)
    with Replay(normal=[g.t("infer/z")]):
        x_real, x_gen = w.generate_samples(x, text, text_pad, patches, patch_pad)
    assert not w.gen.training
    ck.check("generate_samples x_gen", x_gen, g.z["infer/x_gen"])
    ck.check("generate_samples x_real", x_real, x.numpy(), tol=0.0)
    ck.done()


@pytest.mark.parametrize("name", XATTN_FIXTURES)
def test_gradient_penalty_method(name):
    """R:351-374 with the reference's argument order; value and gradient against the recorded train_disc internals."""
    g = Golden(name)
    d = g.dims
    x, text, text_pad, patches, patch_pad = g.inputs()
    w = build(g)
    w.gen.train(); w.disc.train()
    fake = w.gen(g.t("critic1/z").cuda(), patches, patch_pad, text, text_pad)          # R:391, dropout 0
    before = {k: v.clone() for k, v in w.disc.state_dict().items()}
    with Replay(rand=[g.t("critic1/alpha")]):
        gp = w.gradient_penalty(x, fake, patches, patch_pad, text, text_pad)
    assert gp.dim() == 0 and gp.device.type == "cuda"
    los = g.z["critic1/losses"]
    want = (los[0] - los[1]) / 10.0                     # disc_loss = d_loss + gp_weight * gp (R:409)
    ck = Checker(f"gradient_penalty {name}", 1e-3)
    ck.check("gp", np.array([gp.item()]), np.array([want]))
    ck.check("grad_x_hat", w.engine.debug_buffer("gp_grad").view(d["B"], d["G"]), g.z["critic1/grad_x_hat"])
    for k, v in w.disc.state_dict().items():
        assert torch.equal(v, before[k])                # a pure evaluation: no parameter moved
    ck.done()
    with pytest.raises(ValueError):
        w.gradient_penalty(x[:, :-1], fake, patches, patch_pad, text, text_pad)


def test_generate_samples_all_and_dumps(tmp_path):
    g = Golden("xattn_film_T3")
    d = g.dims
    x, text, text_pad, patches, patch_pad = g.inputs()
    w = gga.WGAN_GP(d["G"], d["L"], d["E"], [d["H"], d["H"], d["G"]], [d["H"], d["H"], 1], text_embedding_dims=d["Dt"],
                    patches_embedding_dims=d["Dp"], n_critic=1, dropout=0.0, device="cuda:0", results_dire=str(tmp_path),
                    freq_compute_test=1)
    dis, site = torch.arange(d["B"]) % 3, torch.arange(d["B"]) % 2
    loader = [(text[:3], text_pad[:3], x[:3], patches[:3], patch_pad[:3], dis[:3], site[:3]),
              (text[3:], text_pad[3:], x[3:], patches[3:], patch_pad[3:], dis[3:], site[3:])]
    w.fit(loader, None, loader, epochs=1, val=True)
    real, gen, dr, dg, sr, sg = w.generate_samples_all(loader, num_repeats=2)
    assert real.shape == (d["B"], d["G"]) and gen.shape == (2 * d["B"], d["G"]) and np.array_equal(real, x.numpy())
    assert np.array_equal(dr, dis.numpy()) and np.array_equal(dg, np.tile(dis.numpy(), 2)) and np.array_equal(sg, np.tile(site.numpy(), 2))
    with pytest.raises(NotImplementedError):
        w.generate_samples_all(loader, balanced=True)
    for run in range(2):                                 # R:786-806: twelve files per run, loadable with np.load
        dd = tmp_path / f"test_{run}_epoch_1"
        names = sorted(os.listdir(dd))
        assert names == sorted(f"{a}.npy" for a in ("data_real", "data_gen", "test_real", "test_gen", "train_labels_real",
                                                    "train_labels_gen", "test_labels_real", "test_labels_gen",
                                                    "train_primary_site_real", "train_primary_site_gen",
                                                    "test_primary_site_real", "test_primary_site_gen"))
        assert np.load(dd / "data_gen.npy").shape == (d["B"], d["G"]) and np.array_equal(np.load(dd / "test_real.npy"), x.numpy())


def test_train_draws_the_reference_noise_stream():
    """train() (R:463-477) draws (z, alpha) x n_critic and then z with torch.normal / torch.rand in the reference's order from the
    default generator; the facade draws them straight into their slots (`out=`), which must leave the stream unchanged."""
    g = Golden(XATTN_FIXTURES[0])
    d = g.dims
    w = build(g)
    x, text, text_pad, patches, patch_pad = g.inputs()
    B = x.shape[0]
    seen = {}
    w.train_with_noise = lambda x, text, tpad, pat, ppad, z_all, alpha_all, **k: seen.update(z=z_all.clone(), a=alpha_all.clone())
    dev = torch.device("cuda:0")
    torch.manual_seed(77)
    w.train(x, text, text_pad, patches, patch_pad)
    torch.manual_seed(77)
    for k in range(d["n_critic"]):
        z = torch.normal(0, 1, size=(B, d["L"]), device=dev)                 # R:473
        a = torch.rand(B, 1, device=dev)                                    # R:354
        assert torch.equal(seen["z"][k], z) and torch.equal(seen["a"][k], a.view(B)), k
    assert torch.equal(seen["z"][d["n_critic"]], torch.normal(0, 1, size=(B, d["L"]), device=dev))      # R:476
