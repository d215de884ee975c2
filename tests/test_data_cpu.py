"""gemm_gan_amd/data.py against the reference loader's per-item semantics (src/multi_patch_multi_token_gan_dataloader.py:11-55,
restated here: the reference tree is not read): on-disk formats, float64 -> float32 patches, inverted attention mask,
subsample without replacement / zero padding with mask, tuple order, epoch permutations."""
import numpy as np
import torch

from gemm_gan_amd.data import DeviceCaseCache


def _write_cases(tmp_path, counts, Dp=6, T=5, Dt=4, G=7, seed=0):
    rng = np.random.default_rng(seed)
    pdir, tdir = tmp_path / "patches", tmp_path / "tokens"
    pdir.mkdir(); tdir.mkdir()
    ids, raw = [], {}
    for i, n in enumerate(counts):
        cid = f"case{i}"
        p = rng.standard_normal((n, Dp))                                   # float64, as the preprocessing writes it
        t = rng.standard_normal((1, T, Dt)).astype(np.float32)
        m = np.ones((1, T), dtype=np.int64); m[0, T - (i % 3):] = 0        # Hugging Face: 1 = token, 0 = padding
        np.save(pdir / f"{cid}.npy", p); np.save(tdir / f"{cid}.npy", t); np.save(tdir / f"{cid}_attention_mask.npy", m)
        ids.append(cid); raw[cid] = (p, t, m)
    genes = rng.standard_normal((len(counts), G))
    return ids, raw, genes, pdir, tdir


def test_cache_reproduces_the_reference_item_semantics(tmp_path):
    P = 8
    counts = [3, 8, 20, 9, 1]
    ids, raw, genes, pdir, tdir = _write_cases(tmp_path, counts)
    cache = DeviceCaseCache(ids, tdir, pdir, genes, disease_types=np.arange(5), primary_site=np.arange(5) + 10, num_patches=P,
                            device="cpu")
    g = torch.Generator().manual_seed(3)
    tok, tpad, x, patches, pad, dis, site = cache.batch([0, 1, 2, 3, 4], g)
    assert patches.shape == (5, P, 6) and patches.dtype == torch.float32 and pad.dtype == torch.bool
    assert torch.equal(dis, torch.arange(5)) and torch.equal(site, torch.arange(5) + 10)
    assert torch.allclose(x, torch.tensor(genes, dtype=torch.float32))
    for b, cid in enumerate(ids):
        p64, t, m = raw[cid]
        ref32 = torch.tensor(p64, dtype=torch.float32)                      # D:52
        n = p64.shape[0]
        assert torch.equal(tok[b], torch.tensor(t, dtype=torch.float32).squeeze(0))
        assert torch.equal(tpad[b], ~torch.tensor(m, dtype=torch.bool).squeeze(0))      # D:47
        if n > P:                                                           # D:32-35: P distinct rows of the case, nothing padded
            assert not pad[b].any()
            hits = [(ref32 == patches[b, j]).all(dim=1).nonzero().flatten().tolist() for j in range(P)]
            assert all(len(h) == 1 for h in hits) and len({h[0] for h in hits}) == P
        else:                                                               # D:36-40: file order, then zero rows with mask True
            assert torch.equal(patches[b, :n], ref32) and torch.equal(patches[b, n:], torch.zeros(P - n, 6))
            assert torch.equal(pad[b], torch.tensor([False] * n + [True] * (P - n)))


def test_subsample_is_uniform_and_loader_covers_every_case_once_per_epoch(tmp_path):
    P = 4
    ids, raw, genes, pdir, tdir = _write_cases(tmp_path, [12, 12, 2, 5, 12, 12, 3], seed=1)
    cache = DeviceCaseCache(ids, tdir, pdir, genes, num_patches=P, device="cpu")
    g = torch.Generator().manual_seed(0)
    ref = torch.tensor(raw["case0"][0], dtype=torch.float32)
    seen = torch.zeros(12)
    firsts = torch.zeros(12)
    for _ in range(600):
        patches = cache.batch([0], g)[3][0]
        rows = [(ref == patches[j]).all(dim=1).nonzero().item() for j in range(P)]
        seen[rows] += 1
        firsts[rows[0]] += 1
    assert (seen / 600 - P / 12).abs().max() < 0.08          # every row equally likely to be drawn ...
    assert (firsts / 600 - 1 / 12).abs().max() < 0.05        # ... and to come first (random order, like np.random.choice)
    loader = cache.loader(batch_size=3, shuffle=True, seed=5)
    assert len(loader) == 3
    epochs = []
    for _ in range(2):
        xs = torch.cat([b[2] for b in loader])
        assert xs.shape[0] == 7
        order = [int((torch.tensor(genes, dtype=torch.float32) == r).all(dim=1).nonzero().item()) for r in xs]
        assert sorted(order) == list(range(7))
        epochs.append(order)
    assert epochs[0] != epochs[1]                            # a fresh permutation per epoch
