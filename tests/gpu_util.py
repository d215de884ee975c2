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


ATOL_FRAC = 1e-2      # absolute floor of the elementwise gate, as a fraction of rtol * max|want|


def rel_elementwise(a, b, atol_frac=ATOL_FRAC):
    """max_i |a_i - b_i| / (|b_i| + atol_frac * max|b|): the smallest rtol for which EVERY element satisfies
    |a - b| <= rtol * |b| + rtol * atol_frac * max|b|  (rtol 1e-3, atol_frac 1e-2: 1e-3 relative per element with an
    absolute floor of 1e-5 of the tensor's scale for entries that are cancellation residues)."""
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.size == 0:
        return 0.0
    ref = float(np.abs(b).max())
    if ref < 1e-12:
        return 0.0 if float(np.abs(a).max()) < 1e-6 else float("inf")
    if not np.isfinite(a).all():
        return float("inf")
    return float((np.abs(a - b) / (np.abs(b) + atol_frac * ref)).max())


def optimiser_travel(opt, lr, steps):
    """Largest distance `steps` optimiser steps can move one parameter: RMSprop(alpha=.99) divides by sqrt((1-.99^t) g^2)
    when |g| is constant - 10 lr on the first step; Adam / AdamW are bias-corrected to ~lr per step."""
    if opt == "rms_prop":
        return lr * sum(1.0 / np.sqrt(1.0 - 0.99 ** t) for t in range(1, steps + 1))
    return 1.05 * lr * steps


OUTLIER_SHARE = 5e-3   # share of a tensor's elements that may miss the elementwise bound (they still obey the max-norm one)


def outlier_share(a, b, tol, atol_frac=ATOL_FRAC):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    if b.size == 0:
        return 0.0
    ref = float(np.abs(b).max())
    return float((np.abs(a - b) > tol * (np.abs(b) + atol_frac * ref)).mean())


class Checker:
    """Collects stage errors; asserts at the end so one run reports every failing stage.

    metric="elementwise" (the parity gates: HIP f32 mode against the reference's golden vectors / the CPU oracle): EVERY
    element must satisfy |a - b| <= tol * |b| + tol * 1e-2 * max|b|, except for a share of at most 0.5 % of a tensor's
    elements, which - like everything - must still satisfy the max-norm bound |a - b| <= tol * max|b|.  The exception exists
    for ReLU gates: a pre-activation within fp32 rounding of zero may fall on either side in two correct implementations
    (expected a few times per 10^6 gates), which moves the one weight-gradient row of that unit by one token's
    contribution.  metric="max" (self-consistency of two routes of the SAME precision mode, bf16 bounds): max-norm only."""

    def __init__(self, tag, tol, metric="elementwise"):
        self.tag, self.tol, self.bad, self.metric = tag, tol, [], metric
        diag(f"== {tag} (tol {tol:g}, {metric})")

    def check_post(self, name, got, want, init, opt, lr, steps, rtol=1e-3, share=0.03):
        """Post-optimiser-step parameters, every element: RMSprop / Adam divide by |g|, so an entry whose gradient is
        rounding noise moves by a full step whose SIGN is noise in any implementation.  Gate: all but a few entries agree
        to `rtol` of the tensor's largest update, none is off by more than twice the optimiser's largest travel
        (tests/test_numpy_oracle.py::test_multi_step_conditioning measures why)."""
        g = got.detach().double().cpu().numpy().reshape(-1) if isinstance(got, torch.Tensor) else np.asarray(got, dtype=np.float64).reshape(-1)
        w = np.asarray(want, dtype=np.float64).reshape(-1)
        i = np.asarray(init, dtype=np.float64).reshape(-1)
        err = np.abs(g - w)
        move = float(np.abs(w - i).max()) if w.size else 0.0
        bad = int((err > rtol * move + 1e-7).sum())
        worst = float(err.max()) if err.size else 0.0
        ok = np.isfinite(g).all() and bad <= max(2, share * err.size) and worst <= 2.0 * optimiser_travel(opt, lr, steps) + 1e-6
        diag(f"   {name:55s} post-step: max err {worst:.3e}, largest update {move:.3e}, {bad}/{err.size} beyond {rtol:g} of it"
             + ("" if ok else "   <-- FAIL"))
        if not ok:
            self.bad.append((name, worst))

    def check(self, name, got, want, tol=None):
        tol = self.tol if tol is None else tol
        try:
            e_max = rel(got, want)
            e_el = rel_elementwise(got, want)
            e = e_max
            if self.metric == "elementwise" and e_max <= tol and e_el > tol and outlier_share(got, want, tol) > OUTLIER_SHARE:
                e = e_el
        except AssertionError as ex:
            e = float("inf")
            diag(f"   {name}: SHAPE MISMATCH {ex}")
        flag = "" if e <= tol else "   <-- FAIL"
        if not np.isfinite(e):
            flag = "   <-- FAIL (non-finite)"
        if np.isfinite(e):
            diag(f"   {name:55s} rel_err max-norm {e_max:.3e} elementwise {e_el:.3e}"
                 + (f" (outliers {outlier_share(got, want, tol):.1e})" if e_el > tol >= e_max else "") + flag)
        else:
            diag(f"   {name:55s} rel_err {e}{flag}")
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


PARITY_PRECISION = "f32"      # set by the `parity_mode` fixture (tests/conftest.py): "f32" or "bf16x3"


def engine_from_cfg(cfg, B, P, T, dropout=0.0, seed=0, optimizer=None, precision=None):
    return Engine(precision=precision or PARITY_PRECISION, n_genes=cfg.n_genes, latent_dims=cfg.latent_dims, embedding_dims=cfg.embedding_dims,
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
