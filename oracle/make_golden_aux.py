#!/usr/bin/env python3
"""Golden vectors for the rows SURVEY.md 8(f) marks "next": the dataloader hand-off and the evaluation math.

TEST INFRASTRUCTURE ONLY, like make_golden.py: runs the REAL reference modules in the build container (inert stand-ins for
the third-party imports the image lacks) on tiny synthetic inputs and stores inputs + outputs as .npz.  No reference source
text is written anywhere; /root/reference does not exist on the GPU box.

    python oracle/make_golden_aux.py        # writes tests/golden/aux_*.npz

Fixtures
  aux_loader_items.npz   src/multi_patch_multi_token_gan_dataloader.py:11-55  MultiPatchMultiTokenGANDataset.__getitem__ on
                         cases with fewer / exactly / more rows than num_patches (the 7-tuple of every item)
  aux_loader_split.npz   same file :58-187  dataloader_multi_patch_conditional_gan on a synthetic dataset directory: gene
                         filter, 64/16/20 split, z-score on train statistics, label encodings (the three Dataset objects)
  aux_prdc.npz           src/distribution_distances.py:102-142  compute_prdc (L1 distances) for three k
  aux_knn_pr.npz         src/unsupervised_metrics.py:141-303  ManifoldEstimator / knn_precision_recall_features /
                         get_precision_recall (squared Euclidean distances, <= radius)
  aux_privacy.npz        src/privacy_evaluator.py:9-66  dcr / nndr run by the reference itself (its `.cuda()` calls made
                         no-ops for the duration of the call: the container has no GPU; the arithmetic is the same torch
                         expressions on the CPU), scores plus the per-sample distance / ratio vectors it concatenates
"""
from __future__ import annotations

import os
import pickle
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT_DIR, import_reference  # noqa: E402


def write_dataset_dir(root: Path, spec):
    """Materialise a synthetic dataset directory in the reference's on-disk formats from the arrays in `spec`."""
    import pandas as pd
    root.mkdir(parents=True, exist_ok=True)
    case_ids = [str(c) for c in spec["case_ids"]]
    genes = [str(g) for g in spec["gene_names"]]
    pd.DataFrame(spec["expr"], index=case_ids, columns=genes).to_parquet(root / "rna_seq.parquet")
    (root / "case_ids.txt").write_text("\n".join(str(c) for c in spec["listed_case_ids"]) + "\n")
    txt_ids = [str(c) for c in spec["text_case_ids"]]
    pd.DataFrame(np.zeros((len(txt_ids), 2), dtype=np.float32), index=txt_ids).rename(columns=str).to_parquet(root / "text_emb.parquet")
    pdir, tdir = root / "patch_emb", root / "token_emb"
    pdir.mkdir(exist_ok=True)
    tdir.mkdir(exist_ok=True)
    for i, cid in enumerate(str(c) for c in spec["patch_case_ids"]):
        np.save(pdir / f"{cid}.npy", spec["patches"][spec["patch_off"][i]:spec["patch_off"][i + 1]])
        np.save(tdir / f"{cid}.npy", spec["tokens"][i][None])
        np.save(tdir / f"{cid}_attention_mask.npy", spec["token_mask"][i][None])
    meta = {str(c): {"disease_type": str(d), "primary_site": str(s)}
            for c, d, s in zip(spec["case_ids"], spec["disease"], spec["site"])}
    with open(root / "metainfos.pkl", "wb") as f:
        pickle.dump(meta, f)


def loader_items():
    ref = import_reference("multi_patch_multi_token_gan_dataloader")
    rng = np.random.default_rng(7)
    num_patches, Dp, T, Dt, G = 5, 6, 4, 5, 7
    counts = [3, 5, 9, 1, 12]
    case_ids = [f"case{i}" for i in range(len(counts))]
    out = {"num_patches": np.int64(num_patches), "counts": np.array(counts)}
    with tempfile.TemporaryDirectory() as d:
        d = Path(d)
        (d / "p").mkdir()
        (d / "t").mkdir()
        genes = rng.standard_normal((len(counts), G))
        for i, (cid, n) in enumerate(zip(case_ids, counts)):
            p = rng.standard_normal((n, Dp))                       # float64 on disk, as the UNI embeddings are
            tok = rng.standard_normal((1, T, Dt)).astype(np.float32)
            am = np.ones((1, T), dtype=np.int64)
            am[0, T - (i % 3):] = 0 if i % 3 else 1                # Hugging Face convention: 1 = token
            np.save(d / "p" / f"{cid}.npy", p)
            np.save(d / "t" / f"{cid}.npy", tok)
            np.save(d / "t" / f"{cid}_attention_mask.npy", am)
            out[f"in/patches{i}"], out[f"in/tokens{i}"], out[f"in/mask{i}"] = p, tok, am
        out["in/genes"] = genes
        out["in/disease"] = np.arange(len(counts)) % 3
        out["in/site"] = np.arange(len(counts)) % 2
        ds = ref.MultiPatchMultiTokenGANDataset(case_ids, d / "t", d / "p", genes, out["in/disease"], out["in/site"], num_patches=num_patches)
        assert len(ds) == len(counts)
        for i in range(len(ds)):
            np.random.seed(100 + i)
            item = ds[i]
            for name, v in zip(("tokens", "token_pad", "genes", "patches", "patch_pad", "disease", "site"), item):
                out[f"item{i}/{name}"] = v.numpy()
    path = os.path.join(OUT_DIR, "aux_loader_items.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


def loader_split():
    ref = import_reference("multi_patch_multi_token_gan_dataloader")
    rng = np.random.default_rng(11)
    n_all, G, Dp, T, Dt = 23, 9, 4, 3, 4
    case_ids = [f"TCGA-{i:02d}" for i in range(n_all)]
    expr = rng.gamma(2.0, 1.5, size=(n_all, G))
    expr[:, 2] = 0.0                                    # all zeros -> removed
    expr[rng.random(n_all) < 0.95, 5] = 0.0             # ~95 % zeros -> removed (> 90 %)
    expr[rng.random(n_all) < 0.5, 7] = 0.0              # 50 % zeros -> kept
    expr[:, 4] = 3.25                                   # constant gene: std 0 -> NaN -> fillna(0)
    listed = case_ids[:-1]                              # one case missing from case_ids.txt
    text_ids = case_ids[1:]                             # one without text embedding
    patch_ids = [c for i, c in enumerate(case_ids) if i != 5]   # one without patch file
    counts = rng.integers(1, 7, size=len(patch_ids))
    off = np.concatenate([[0], np.cumsum(counts)])
    spec = dict(case_ids=np.array(case_ids), gene_names=np.array([f"g{i}" for i in range(G)]), expr=expr,
                listed_case_ids=np.array(listed), text_case_ids=np.array(text_ids), patch_case_ids=np.array(patch_ids),
                patches=rng.standard_normal((int(off[-1]), Dp)), patch_off=off,
                tokens=rng.standard_normal((len(patch_ids), T, Dt)).astype(np.float32),
                token_mask=np.ones((len(patch_ids), T), dtype=np.int64),
                disease=np.array([f"d{(i * 7) % 4}" for i in range(n_all)]), site=np.array([f"s{(i * 3) % 5}" for i in range(n_all)]))
    out = {f"spec/{k}": v for k, v in spec.items()}
    with tempfile.TemporaryDirectory() as d:
        root = Path(d)
        write_dataset_dir(root, spec)
        tr, va, te, n_genes = ref.dataloader_multi_patch_conditional_gan(
            root, batch_size=4, num_workers=0, num_patches=3, text_embedding_file="text_emb.parquet",
            patch_embeddings_folder="patch_emb", token_embeddings_folder="token_emb")
        out["n_genes"] = np.int64(n_genes)
        for name, loader in (("train", tr), ("validation", va), ("test", te)):
            ds = loader.dataset
            out[f"{name}/case_ids"] = np.array([str(c) for c in ds.case_ids])
            out[f"{name}/gene_expressions"] = np.asarray(ds.gene_expressions, dtype=np.float64)
            out[f"{name}/disease_types"] = np.asarray(ds.disease_types, dtype=np.int64)
            out[f"{name}/primary_site"] = np.asarray(ds.primary_site, dtype=np.int64)
            out[f"{name}/n_batches"] = np.int64(len(loader))
    path = os.path.join(OUT_DIR, "aux_loader_split.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "n_genes", int(out["n_genes"]))


def prdc():
    ref = import_reference("distribution_distances")
    rng = np.random.default_rng(3)
    real = rng.standard_normal((70, 12)).astype(np.float32)
    fake = (0.8 * rng.standard_normal((55, 12)) + 0.3).astype(np.float32)
    out = {"real": real, "fake": fake}
    for k in (1, 5, 10):
        r = ref.compute_prdc(real, fake, nearest_k=k)
        out[f"k{k}"] = np.array([r["precision"], r["recall"], r["density"], r["coverage"]], dtype=np.float64)
    path = os.path.join(OUT_DIR, "aux_prdc.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


def knn_pr():
    ref = import_reference("unsupervised_metrics")
    rng = np.random.default_rng(5)
    real = torch.from_numpy(rng.standard_normal((64, 10)).astype(np.float32))
    fake = torch.from_numpy((0.7 * rng.standard_normal((48, 10)) + 0.4).astype(np.float32))
    out = {"real": real.numpy(), "fake": fake.numpy()}
    for k in (3, 10):
        est = ref.ManifoldEstimator(real, 25000, 50000, [k])
        out[f"k{k}/radii_real"] = est.D.copy()
        pred, realism = est.evaluate(fake, return_realism=True)
        out[f"k{k}/fake_in_real_manifold"] = pred.copy()
        out[f"k{k}/realism"] = realism.copy()
        p, r = ref.get_precision_recall(real, fake, nb_nn=[k])
        out[f"k{k}/precision_recall"] = np.array([p, r], dtype=np.float64)
    path = os.path.join(OUT_DIR, "aux_knn_pr.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


def privacy():
    """dcr / nndr of src/privacy_evaluator.py:9-66 on small sets that exercise what the metric exists for (exact and near
    copies of training records), the reference's batching (fewer than, exactly, and more than one batch of 128 generated rows)
    (no exact ties: a strict comparison at a tie is fp32 rounding in any implementation).  torch.Tensor.cuda is an identity while the reference runs; torch.cat is wrapped to keep the per-sample vectors."""
    ref = import_reference("privacy_evaluator")
    rng = np.random.default_rng(11)
    out = {}
    cases = {"small": (40, 90, 33, 17), "one_batch": (128, 60, 70, 24), "multi_batch": (300, 150, 120, 64)}
    out["cases"] = np.array(sorted(cases))
    for name, (nq, nr, nt, dim) in cases.items():
        real = rng.standard_normal((nr, dim)).astype(np.float32)
        test = rng.standard_normal((nt, dim)).astype(np.float32)
        gen = rng.standard_normal((nq, dim)).astype(np.float32)
        gen[0] = real[3]                                                          # exact copy of a training record
        gen[2] = real[0] + 1e-3 * rng.standard_normal(dim).astype(np.float32)     # near copy
        gen[5] = test[1]                                                          # exact copy of a test record
        cat_rec = []
        orig_cuda, orig_cat = torch.Tensor.cuda, torch.cat

        def cat(xs, *a, **k):
            o = orig_cat(xs, *a, **k)
            cat_rec.append(o.detach().clone().numpy())
            return o
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.cat = cat
        try:
            s_dcr = ref.dcr(real, gen, test)
            s_nndr = ref.nndr(real, gen, test)
        finally:
            torch.Tensor.cuda, torch.cat = orig_cuda, orig_cat
        assert len(cat_rec) == 4
        out[f"{name}/real"], out[f"{name}/test"], out[f"{name}/gen"] = real, test, gen
        out[f"{name}/scores"] = np.array([s_dcr, s_nndr], dtype=np.float64)
        out[f"{name}/dcr_real"], out[f"{name}/dcr_test"] = cat_rec[0], cat_rec[1]
        out[f"{name}/nndr_real"], out[f"{name}/nndr_test"] = cat_rec[2], cat_rec[3]
    path = os.path.join(OUT_DIR, "aux_privacy.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {n: out[f"{n}/scores"].tolist() for n in cases})


if __name__ == "__main__":
    torch.set_num_threads(1)
    only = set(sys.argv[1:])
    for fn in (loader_items, loader_split, prdc, knn_pr, privacy):
        if not only or fn.__name__ in only:
            fn()
