"""gemm_gan_amd/data.py against the reference loader (src/multi_patch_multi_token_gan_dataloader.py): per-item semantics
pinned by items recorded from the REAL `MultiPatchMultiTokenGANDataset` (tests/golden/aux_loader_items.npz, written by
oracle/make_golden_aux.py), the front half (`dataloader_multi_patch_conditional_gan`: gene filter, split, z-score, label
encodings) by tests/golden/aux_loader_split.npz; plus on-disk formats, subsampling statistics, epoch permutations."""
import os
import pickle
from pathlib import Path

import numpy as np
import torch

from gemm_gan_amd.data import DeviceCaseCache, dataloader_multi_patch_conditional_gan

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_items_equal_the_recorded_reference_items(tmp_path):
    """Every 7-tuple the reference Dataset returned for cases with fewer / exactly / more rows than num_patches."""
    z = np.load(os.path.join(GOLDEN, "aux_loader_items.npz"))
    counts, P = z["counts"], int(z["num_patches"])
    pdir, tdir = tmp_path / "p", tmp_path / "t"
    pdir.mkdir(); tdir.mkdir()
    ids = []
    for i in range(len(counts)):
        cid = f"case{i}"
        np.save(pdir / f"{cid}.npy", z[f"in/patches{i}"]); np.save(tdir / f"{cid}.npy", z[f"in/tokens{i}"])
        np.save(tdir / f"{cid}_attention_mask.npy", z[f"in/mask{i}"])
        ids.append(cid)
    cache = DeviceCaseCache(ids, tdir, pdir, z["in/genes"], z["in/disease"], z["in/site"], num_patches=P, device="cpu")
    tok, tpad, x, patches, pad, dis, site = cache.batch(list(range(len(counts))), torch.Generator().manual_seed(0))
    for i, n in enumerate(counts):
        assert np.array_equal(tok[i].numpy(), z[f"item{i}/tokens"]) and np.array_equal(tpad[i].numpy(), z[f"item{i}/token_pad"])
        assert np.array_equal(x[i].numpy(), z[f"item{i}/genes"])
        assert int(dis[i]) == int(z[f"item{i}/disease"]) and int(site[i]) == int(z[f"item{i}/site"])
        assert np.array_equal(pad[i].numpy(), z[f"item{i}/patch_pad"])            # all False, also for the zero-padded cases
        want = z[f"item{i}/patches"]
        assert patches[i].dtype == torch.float32 and want.dtype == np.float32
        if n <= P:
            assert np.array_equal(patches[i].numpy(), want)                       # rows in file order, then zero rows
        else:       # a random subset: same row set semantics (P distinct rows of the case), order and choice are the RNG's
            src = z[f"in/patches{i}"].astype(np.float32)
            for rows in (patches[i].numpy(), want):
                hits = [np.flatnonzero((src == r).all(axis=1)) for r in rows]
                assert all(len(h) == 1 for h in hits) and len({int(h[0]) for h in hits}) == P
    # the documented opt-in: padded slots masked
    fixed = DeviceCaseCache(ids, tdir, pdir, z["in/genes"], num_patches=P, device="cpu", mask_padding=True)
    pad2 = fixed.batch([0, 3], torch.Generator().manual_seed(0))[4]
    assert pad2[0].tolist() == [False] * 3 + [True] * 2 and pad2[1].tolist() == [False] + [True] * 4


def test_front_half_equals_the_recorded_reference_split(tmp_path):
    import pandas as pd
    z = np.load(os.path.join(GOLDEN, "aux_loader_split.npz"), allow_pickle=False)
    spec = {k[5:]: z[k] for k in z.files if k.startswith("spec/")}
    root = Path(tmp_path)
    case_ids = [str(c) for c in spec["case_ids"]]
    pd.DataFrame(spec["expr"], index=case_ids, columns=[str(g) for g in spec["gene_names"]]).to_parquet(root / "rna_seq.parquet")
    (root / "case_ids.txt").write_text("\n".join(str(c) for c in spec["listed_case_ids"]) + "\n")
    txt = [str(c) for c in spec["text_case_ids"]]
    pd.DataFrame(np.zeros((len(txt), 2), dtype=np.float32), index=txt).rename(columns=str).to_parquet(root / "text_emb.parquet")
    (root / "patch_emb").mkdir(); (root / "token_emb").mkdir()
    for i, cid in enumerate(str(c) for c in spec["patch_case_ids"]):
        np.save(root / "patch_emb" / f"{cid}.npy", spec["patches"][spec["patch_off"][i]:spec["patch_off"][i + 1]])
        np.save(root / "token_emb" / f"{cid}.npy", spec["tokens"][i][None])
        np.save(root / "token_emb" / f"{cid}_attention_mask.npy", spec["token_mask"][i][None])
    with open(root / "metainfos.pkl", "wb") as f:
        pickle.dump({str(c): {"disease_type": str(d), "primary_site": str(s)}
                     for c, d, s in zip(spec["case_ids"], spec["disease"], spec["site"])}, f)
    tr, va, te, n_genes = dataloader_multi_patch_conditional_gan(
        root, batch_size=4, num_workers=0, num_patches=3, text_embedding_file="text_emb.parquet",
        patch_embeddings_folder="patch_emb", token_embeddings_folder="token_emb", device="cpu")
    assert n_genes == int(z["n_genes"])
    for name, loader in (("train", tr), ("validation", va), ("test", te)):
        ds = loader.dataset
        assert [str(c) for c in ds.case_ids] == [str(c) for c in z[f"{name}/case_ids"]], name
        assert np.allclose(ds.gene_expressions.numpy(), z[f"{name}/gene_expressions"].astype(np.float32), rtol=1e-6, atol=1e-6)
        assert np.array_equal(ds.disease_types.numpy(), z[f"{name}/disease_types"])
        assert np.array_equal(ds.primary_site.numpy(), z[f"{name}/primary_site"])
        assert len(loader) == int(z[f"{name}/n_batches"])
    b = next(iter(tr))
    assert b[3].shape == (4, 3, spec["patches"].shape[1]) and b[2].shape == (4, n_genes) and not b[4].any()


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
        else:                                                               # D:36-40: file order, then zero rows; mask all False
            assert torch.equal(patches[b, :n], ref32) and torch.equal(patches[b, n:], torch.zeros(P - n, 6))
            assert not pad[b].any()


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
