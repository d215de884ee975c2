"""The training loop of the host mirror (WGAN_GP.fit, R:619-711) on the GPU: loss bookkeeping, LR plumbing, checkpoints
that round-trip through state_dict (the reference's own keys), inference after loading them."""
import os
import numpy as np
import pytest
import torch

import gemm_gan_amd as gga

pytestmark = pytest.mark.gpu


def _batches(n, B, G, P, T, Dt, Dp, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        text = torch.randn(B, T, Dt, generator=g)
        tpad = torch.zeros(B, T, dtype=torch.bool)
        x = torch.randn(B, G, generator=g)
        patches = torch.randn(B, P, Dp, generator=g)
        ppad = torch.zeros(B, P, dtype=torch.bool)
        ppad[:, P - 2:] = True                                   # ragged: the last two patches are padding
        out.append((text, tpad, x, patches, ppad))               # the reference's dataloader order (R:667-673)
    return out


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_fit_two_epochs_checkpoints_and_reload(tmp_path, precision):
    G, Lz, E, H, Dt, Dp, B, P, T = 120, 32, 64, 48, 40, 72, 10, 12, 3
    torch.manual_seed(3)
    w = gga.WGAN_GP(G, Lz, E, [H, H, G], [H, H, 1], text_embedding_dims=Dt, patches_embedding_dims=Dp, optimizer="rms_prop",
                    n_critic=2, freq_print=10, freq_compute_test=1, dropout=0.1, seed=9, device="cuda:0",
                    results_dire=str(tmp_path), precision=precision)
    data = _batches(3, B, G, P, T, Dt, Dp, seed=5)
    hist = w.fit(data, epochs=2)
    assert all(len(hist[k]) == 2 for k in ("d loss", "d real loss", "d fake loss", "g loss"))
    assert all(np.isfinite(v) for k in hist for v in hist[k])
    # the attributes a caller of the reference reads after train()
    assert w.disc_loss.dim() == 0 and w.gen_loss.dim() == 0 and w.d_batch_loss.shape == (3,) and w.g_batch_loss.shape == (1,)
    assert abs(float(w.disc_loss) - (w.d_batch_loss[0] + w.gp_weight * w.gp_value)) < 1e-3 * max(1.0, abs(float(w.disc_loss)))
    for name in ("generator_epoch_1.pt", "discriminator_epoch_2.pt", "generator_last_epoch.pt", "discriminator_last_epoch.pt"):
        assert os.path.exists(os.path.join(str(tmp_path), name)), name
    # a fresh model that loads the checkpoints generates what the trained one generates (eval mode, same z)
    w2 = gga.WGAN_GP(G, Lz, E, [H, H, G], [H, H, 1], text_embedding_dims=Dt, patches_embedding_dims=Dp, optimizer="rms_prop",
                     n_critic=2, dropout=0.1, seed=9, device="cuda:0", results_dire="", precision=precision)
    w2.build_WGAN_GP()
    w2.init_train()
    w2.gen.load_state_dict(torch.load(os.path.join(str(tmp_path), "generator_last_epoch.pt")), strict=True)
    w2.disc.load_state_dict(torch.load(os.path.join(str(tmp_path), "discriminator_last_epoch.pt")), strict=True)
    text, tpad, x, patches, ppad = (t.cuda() for t in data[0])
    z = torch.randn(B, Lz, generator=torch.Generator().manual_seed(1)).cuda()
    w.gen.eval(); w2.gen.eval(); w.disc.eval(); w2.disc.eval()
    a = w.gen(z, patches, ppad, text, tpad)
    b = w2.gen(z, patches, ppad, text, tpad)
    assert a.shape == (B, G) and torch.isfinite(a).all()
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), float((a - b).abs().max())
    da = w.disc(x, patches, ppad, text, tpad)
    db = w2.disc(x, patches, ppad, text, tpad)
    assert torch.allclose(da, db, rtol=1e-5, atol=1e-6)


def test_fit_from_the_device_resident_loader(tmp_path):
    """gemm_gan_amd/data.py: the reference's on-disk case files -> HBM-resident cache -> minibatches assembled on the device in
    the reference loader's tuple order -> WGAN_GP.fit, unchanged."""
    from gemm_gan_amd.data import DeviceCaseCache
    rng = np.random.default_rng(0)
    G, Lz, E, H, Dt, Dp, T, P = 90, 16, 64, 32, 24, 40, 6, 12
    pdir, tdir = tmp_path / "patches", tmp_path / "tokens"
    pdir.mkdir(); tdir.mkdir()
    ids = []
    for i, n in enumerate([5, 30, 12, 40, 7, 19, 3, 25, 14]):
        cid = f"c{i}"
        np.save(pdir / f"{cid}.npy", rng.standard_normal((n, Dp)))
        np.save(tdir / f"{cid}.npy", rng.standard_normal((1, T, Dt)).astype(np.float32))
        m = np.ones((1, T), dtype=np.int64); m[0, T - (i % 3):] = 0
        np.save(tdir / f"{cid}_attention_mask.npy", m)
        ids.append(cid)
    cache = DeviceCaseCache(ids, tdir, pdir, rng.standard_normal((len(ids), G)), num_patches=P, device="cuda:0")
    loader = cache.loader(batch_size=4, shuffle=True, seed=1)
    b = next(iter(loader))
    assert all(t.is_cuda for t in b) and b[3].shape == (4, P, Dp) and b[4].dtype == torch.bool
    w = gga.WGAN_GP(G, Lz, E, [H, H, G], [H, H, 1], text_embedding_dims=Dt, patches_embedding_dims=Dp, n_critic=2, dropout=0.1,
                    seed=2, device="cuda:0", results_dire="")
    hist = w.fit(loader, epochs=2)
    assert len(hist["d loss"]) == 2 and all(np.isfinite(v) for v in hist["d loss"] + hist["g loss"])
