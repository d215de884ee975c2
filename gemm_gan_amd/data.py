"""Dataloader hand-off: the reference feeds `fit()` from a torch `DataLoader` whose workers read one
`.npy` per case and subsample / pad the patches on the host (`src/multi_patch_multi_token_gan_dataloader.py:11-55`, D below).
With a GPU step of tens of milliseconds that host path starves the device, and 288 GB of HBM hold the whole training set:
`DeviceCaseCache` reads every case ONCE, keeps the embeddings resident on the device and builds each minibatch there -
same on-disk formats, same per-item semantics, same tuple order as the reference loader, so `WGAN_GP.fit(loader)` takes it
unchanged.

On-disk formats (D:31, D:43-47): `<patches_path>/<case>.npy` float64 `[N_i, Dp]`; `<tokens_path>/<case>.npy` float32
`[1, T, Dt]`; `<tokens_path>/<case>_attention_mask.npy` `[1, T]` with the Hugging Face convention 1 = token, inverted to
torch's True = padded (D:47).  Per-item patch semantics (D:32-40): more than `num_patches` rows -> `num_patches` of them,
uniformly without replacement, in random order; otherwise the rows in file order followed by zero rows, mask True on the
padding.
"""
from pathlib import Path
from typing import Iterator, Optional, Sequence

import numpy as np
import torch


class DeviceCaseCache:
    def __init__(self, case_ids: Sequence[str], tokens_path, patches_path, gene_expressions, disease_types=None,
                 primary_site=None, num_patches: int = 256, device="cuda:0", patch_dtype=torch.float32):
        self.device = torch.device(device)
        self.num_patches = int(num_patches)
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
