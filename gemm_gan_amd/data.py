"""Dataloader hand-off: the reference feeds `fit()` from a torch `DataLoader` whose workers read one
`.npy` per case and subsample / pad the patches on the host (`src/multi_patch_multi_token_gan_dataloader.py:11-55`, D below).
With a GPU step of tens of milliseconds that host path starves the device, and 288 GB of HBM hold the whole training set:
`DeviceCaseCache` reads every case ONCE, keeps the embeddings resident on the device and builds each minibatch there -
same on-disk formats, same per-item semantics, same tuple order as the reference loader, so `WGAN_GP.fit(loader)` takes it
unchanged.

On-disk formats (D:31, D:43-47): `<patches_path>/<case>.npy` float64 `[N_i, Dp]`; `<tokens_path>/<case>.npy` float32
`[1, T, Dt]`; `<tokens_path>/<case>_attention_mask.npy` `[1, T]` with the Hugging Face convention 1 = token, inverted to
torch's True = padded (D:47).  Per-item patch semantics (D:32-40): more than `num_patches` rows -> `num_patches` of them,
uniformly without replacement, in random order; otherwise the rows in file order followed by zero rows.

Padding mask: the reference builds it AFTER rebinding `patches` to the zero-padded array (D:38-40), so
`patches.shape[0] == num_patches` there and the mask is all False - the zero rows are attended like real patches.  That is
what the reference trains with, so it is the default here (`mask_padding=False`; pinned by tests/golden/aux_loader_items.npz,
recorded from the real Dataset).  `mask_padding=True` marks the padded slots True instead (what the code presumably meant).

`dataloader_multi_patch_conditional_gan` is the front half (D:58-187): case intersection, gene filter, 64/16/20 split,
z-score on the training statistics, label encodings - same arguments, same return tuple, loaders backed by the device cache.
"""
import pickle
import random
from pathlib import Path
from typing import Iterator, Optional, Sequence

import numpy as np
import torch


class DeviceCaseCache:
    def __init__(self, case_ids: Sequence[str], tokens_path, patches_path, gene_expressions, disease_types=None,
                 primary_site=None, num_patches: int = 256, device="cuda:0", patch_dtype=torch.float32, mask_padding: bool = False):
        self.device = torch.device(device)
        self.num_patches = int(num_patches)
        self.mask_padding = bool(mask_padding)
        self.case_ids = list(case_ids)
        tokens_path, patches_path = Path(tokens_path), Path(patches_path)
        n = len(case_ids)
        counts, chunks, toks, masks = [], [], [], []
        for cid in case_ids:
            p = np.load(patches_path / f"{cid}.npy")                                   # float64 [N_i, Dp]
            counts.append(p.shape[0])
            chunks.append(torch.from_numpy(np.ascontiguousarray(p)).to(torch.float32))   # == torch.tensor(p, dtype=float32), D:52
            toks.append(torch.from_numpy(np.load(tokens_path / f"{cid}.npy")).to(torch.float32).squeeze(0))
            m = torch.from_numpy(np.load(tokens_path / f"{cid}_attention_mask.npy")).to(torch.bool).squeeze(0)
            masks.append(~m)                                                             # D:47
        self.counts = torch.tensor(counts, dtype=torch.long, device=self.device)
        self.offsets = torch.cumsum(self.counts, 0) - self.counts
        # one flat [sum N_i + 1, Dp] tensor; the extra last row is the zero row every padded slot points at
        flat = torch.cat(chunks + [torch.zeros(1, chunks[0].shape[1])], dim=0)
        self.patches = flat.to(self.device, patch_dtype)
        self.zero_row = self.patches.shape[0] - 1
        self.tokens = torch.stack(toks).to(self.device)
        self.token_pad = torch.stack(masks).to(self.device)
        self.gene_expressions = torch.as_tensor(np.asarray(gene_expressions), dtype=torch.float32).to(self.device)
        z = torch.zeros(n, dtype=torch.long)
        self.disease_types = torch.as_tensor(np.asarray(disease_types), dtype=torch.long).to(self.device) if disease_types is not None else z.to(self.device)
        self.primary_site = torch.as_tensor(np.asarray(primary_site), dtype=torch.long).to(self.device) if primary_site is not None else z.to(self.device)
        self.max_count = int(max(counts))

    def __len__(self):
        return int(self.counts.shape[0])

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.patches, self.tokens, self.token_pad, self.gene_expressions))

    def patch_indices(self, idx: torch.Tensor, generator: Optional[torch.Generator] = None):
        """[B, num_patches] rows of the flat patch tensor (zero_row where padded) and the padding mask, built on the device:
        one sort per minibatch instead of one host-side `np.random.choice` per sample."""
        P = self.num_patches
        cnt = self.counts[idx]                                                    # [B]
        W = max(self.max_count, P)
        pos = torch.arange(W, device=self.device).expand(idx.shape[0], W)
        valid = pos < cnt[:, None]
        keys = torch.rand(idx.shape[0], W, device=self.device, generator=generator)       # random order for the subsampled cases
        keys = torch.where((cnt > P)[:, None], keys, pos.to(keys.dtype) / W)                # file order for the padded ones
        keys = torch.where(valid, keys, torch.full_like(keys, 2.0))                         # rows past N_i sort last
        order = torch.topk(keys, P, dim=1, largest=False, sorted=True).indices              # [B, P]
        pad = order >= cnt[:, None]
        rows = torch.where(pad, torch.full_like(order, self.zero_row), self.offsets[idx][:, None] + order)
        if not self.mask_padding:
            pad = torch.zeros_like(pad)          # the reference's mask: all False, zero rows attended (D:38-40)
        return rows, pad

    def batch(self, idx, generator: Optional[torch.Generator] = None):
        """The reference loader's 7-tuple (D:55) for the cases `idx`, every tensor on the device."""
        idx = torch.as_tensor(idx, dtype=torch.long, device=self.device)
        rows, pad = self.patch_indices(idx, generator)
        patches = self.patches[rows].to(torch.float32)                            # [B, P, Dp]
        return (self.tokens[idx], self.token_pad[idx], self.gene_expressions[idx], patches, pad,
                self.disease_types[idx], self.primary_site[idx])

    def loader(self, batch_size: int, shuffle: bool = True, seed: int = 42, drop_last: bool = False) -> "DeviceLoader":
        return DeviceLoader(self, batch_size, shuffle, seed, drop_last)


class DeviceLoader:
    """Iterable with `len()`, like the `DataLoader` the reference hands to `fit()` (D:178): a fresh permutation per epoch
    (shuffle=True), minibatches assembled on the device by `DeviceCaseCache.batch`."""

    def __init__(self, cache: DeviceCaseCache, batch_size: int, shuffle: bool = True, seed: int = 42, drop_last: bool = False):
        self.cache, self.batch_size, self.shuffle, self.drop_last = cache, int(batch_size), shuffle, drop_last
        self.dataset = cache                    # DataLoader.dataset, as the reference's callers read it
        self.gen = torch.Generator(device=cache.device)
        self.gen.manual_seed(seed)

    def __len__(self):
        n = len(self.cache)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator:
        n = len(self.cache)
        order = torch.randperm(n, device=self.cache.device, generator=self.gen) if self.shuffle else torch.arange(n, device=self.cache.device)
        for i in range(len(self)):
            yield self.cache.batch(order[i * self.batch_size:(i + 1) * self.batch_size], self.gen)


def split_data(n_samples, train_rate=0.80, validation_rate=0.20, seed=42, shuffle=True):
    """src/multi_patch_gan_dataloader.py:77-102: one legacy-MT19937 shuffle of arange(n) under `seed`, cut at 64 % / 80 %.
    (The reference reseeds the GLOBAL numpy / random generators to get this permutation; a private RandomState gives the
    identical permutation without that side effect.)"""
    idxs = np.arange(n_samples)
    if shuffle:
        np.random.RandomState(seed).shuffle(idxs)
    t_tr = int(train_rate * (1 - validation_rate) * n_samples)
    t_val = t_tr + int(train_rate * validation_rate * n_samples)
    return idxs[:t_tr], idxs[t_tr:t_val], idxs[t_val:]


def dataloader_multi_patch_conditional_gan(dataset_path, normalize: bool = True, percentage_to_remove: float = 90,
                                           norm_type: str = "standardize", num_patches: int = 256, batch_size: int = 8,
                                           seed: int = 42, num_workers: int = 4, embedding_dim: int = 256,
                                           text_embedding_file: Optional[str] = None, patch_embeddings_folder: Optional[str] = None,
                                           token_embeddings_folder: Optional[str] = None, device="cuda:0", mask_padding: bool = False):
    """D:58-187 with device-resident loaders: (train_loader, validation_loader, test_loader, n_genes).  `num_workers` is
    accepted and unused (there are no host workers: minibatches are assembled on the device)."""
    import pandas as pd
    dataset_path = Path(dataset_path)
    text_embedding_file = text_embedding_file or f"text_embeddings_contrastive_{embedding_dim}.parquet"
    patch_embeddings_folder = patch_embeddings_folder or f"patch_embeddings_contrastive_{embedding_dim}"
    token_embeddings_folder = token_embeddings_folder or f"../text_embeddings_contrastive_{embedding_dim}"
    df_expr = pd.read_parquet(dataset_path / "rna_seq.parquet")
    with open(dataset_path / "case_ids.txt") as f:
        listed = {c.strip() for c in f.read().splitlines()}
    text_ids = set(pd.read_parquet(dataset_path / text_embedding_file).index.tolist())
    img_ids = {p.stem for p in (dataset_path / patch_embeddings_folder).glob("*.npy")}
    case_ids = sorted(listed & img_ids & text_ids & set(df_expr.index.tolist()))                     # D:93-94
    zero_percent = (df_expr == 0).sum() / len(df_expr) * 100                                       # over ALL rows, D:98
    df_expr = df_expr.loc[:, zero_percent <= percentage_to_remove]
    n_genes = df_expr.shape[1]
    parts = split_data(len(case_ids), seed=42)          # the reference calls split_data(n) with its default seed (D:105)
    ids = [[case_ids[i] for i in p] for p in parts]
    frames = [df_expr.loc[i] for i in ids]
    if normalize and norm_type == "standardize":                                                   # D:116-122
        mean, std = np.mean(frames[0], axis=0), np.std(frames[0], axis=0)
        frames = [((f - mean) / std).fillna(0) for f in frames]
    elif normalize and norm_type == "min-max":                                                     # D:124-129
        mx, mn = np.max(frames[0], axis=0), np.min(frames[0], axis=0)
        frames = [((f - mn) / (mx - mn)).fillna(0) for f in frames]
    with open(dataset_path / "metainfos.pkl", "rb") as f:
        meta = pickle.load(f)
    labels = []
    for key in ("disease_type", "primary_site"):                                                   # D:138-160
        raw = [[meta[c][key] for c in part] for part in ids]
        table = {v: i for i, v in enumerate(sorted(set(raw[0] + raw[1] + raw[2])))}
        labels.append([[table[v] for v in part] for part in raw])
    loaders = []
    for k in range(3):
        cache = DeviceCaseCache(ids[k], dataset_path / token_embeddings_folder, dataset_path / patch_embeddings_folder,
                                frames[k].values, labels[0][k], labels[1][k], num_patches=num_patches, device=device,
                                mask_padding=mask_padding)
        loaders.append(cache.loader(batch_size, shuffle=(k < 2), seed=seed))                       # D:178-185
    return loaders[0], loaders[1], loaders[2], n_genes
