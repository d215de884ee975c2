"""Helpers for the -m gpu parity tests (HIP engine vs oracles / golden vectors)."""
import os

import numpy as np
import torch

from gemm_gan_amd import _lib as L
from gemm_gan_amd.engine import Engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "gpurun_out", "parity_diag.txt")


def diag(msg):
    os.makedirs(os.path.dirname(DIAG), exist_ok=True)
    with open(DIAG, "a") as f:
        f.write(msg + "\n")


def rel(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    ref = float(np.abs(b).max()) if b.size else 0.0
    if ref < 1e-12:      # reference is (numerically) exactly zero: accept fp32 rounding-level noise
        return 0.0 if (a.size == 0 or float(np.abs(a).max()) < 1e-6) else float("inf")
    return float(np.abs(a - b).max() / ref)


class Checker:
    """Collects stage errors; asserts at the end so one run reports every failing stage."""

    def __init__(self, tag, tol):
        self.tag, self.tol, self.bad = tag, tol, []
        diag(f"== {tag} (tol {tol:g})")

    def check(self, name, got, want, tol=None):
        tol = self.tol if tol is None else tol
        try:
            e = rel(got, want)
        except AssertionError as ex:
            e = float("inf")
            diag(f"   {name}: SHAPE MISMATCH {ex}")
        flag = "" if e <= tol else "   <-- FAIL"
        if not np.isfinite(e):
            flag = "   <-- FAIL (non-finite)"
        diag(f"   {name:55s} rel_err {e:.3e}{flag}")
        if flag:
            try:
                ga = got.detach().cpu().numpy().reshape(-1) if isinstance(got, torch.Tensor) else np.asarray(got).reshape(-1)
                wa = want.detach().cpu().numpy().reshape(-1) if isinstance(want, torch.Tensor) else np.asarray(want).reshape(-1)
                diag(f"        got  {ga[:6]}\n        want {wa[:6]}")
            except Exception:
                pass
        if flag:
            self.bad.append((name, e))

    def done(self):
        assert not self.bad, f"{self.tag}: " + ", ".join(f"{n} ({e:.2e})" for n, e in self.bad[:12])


def engine_from_cfg(cfg, B, P, T, dropout=0.0, seed=0, optimizer=None):
    return Engine(n_genes=cfg.n_genes, latent_dims=cfg.latent_dims, embedding_dims=cfg.embedding_dims,
                  hidden_dims=cfg.hidden_dims, text_dims=cfg.text_dims, patch_dims=cfg.patch_dims,
                  n_heads=cfg.n_heads, n_layers=cfg.n_layers, negative_slope=cfg.negative_slope, dropout=dropout,
                  lr_d=cfg.lr_d, lr_g=cfg.lr_g, optimizer=optimizer or cfg.optimizer, gp_weight=cfg.gp_weight,
                  clip_d=cfg.clip_d or 0.0, clip_g=cfg.clip_g or 0.0, max_batch=B, max_patches=P, max_text_tokens=T,
                  seed=seed, device="cuda:0", variant=getattr(cfg, "variant", "xattn_film"))


def load_oracle_state(eng, trainer):
    eng.load_state(L.ROLE_GENERATOR, trainer.gen.state_dict())
    eng.load_state(L.ROLE_CRITIC, trainer.disc.state_dict())


def dev(*ts):
    return tuple(t.cuda().contiguous() for t in ts)
