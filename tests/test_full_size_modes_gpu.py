"""The arithmetic modes the bench quotes, held to the on-chip fp32 reference at BASELINE cfg3's FULL size.

cfg3: B = 256, 5 000 genes, 256 x 1024 patch tokens (S = 257), 1 x 512 text token, E = H = L = 256, dropout 0 - the shape
`bench.py` times.  At this size no CPU oracle finishes in reasonable time, but the box has an exact reference of its own: the
`f32` mode (fp32-input MFMA, unfused attention), which the golden / oracle suites hold to the reference's tolerance at the sizes
the oracle can run.  On ONE set of weights, inputs and noise (z, alpha), one critic iteration and one generator iteration
(R:376-461) are run in every mode:

  * `bf16x3` (`parity_mode` of the bench line) vs `f32`: the SAME 1e-3 elementwise gate as the golden / oracle suites - losses,
    x_fake, the conditioning vector, grad_x_hat and every gradient tensor of both networks;
  * `bf16` (the headline mode) vs `f32`: activations and losses <= 1e-2 rel-L2, the share of ReLU gates that land on the other
    side is reported (encoder FFN, both heads), gradient tensors within the mode's stated bounds (0.15 / 0.25 rel-L2: gate
    flips, see tests/test_gate_flips_gpu.py) and their direction (cosine of the flat gradient >= 0.99).
"""
import numpy as np
import pytest
import torch

from gemm_gan_amd import _lib as L
from gpu_util import Checker, dev, diag, engine_from_cfg, load_oracle_state
from oracle.torch_oracle import PathConfig, Trainer, synthetic_batch
from test_bf16_parity_gpu import ACT_TOL, GEN_GRAD_TOL, GRAD_TOL, LOSS_TOL, Gate, l2, significant

pytestmark = pytest.mark.gpu
B_, P_, T_ = 256, 256, 1


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))


_cache = {}


def run_mode(mode):
    """One critic + one generator iteration at cfg3 in `mode`; tensors moved to the host (the three engines never coexist)."""
    if mode in _cache:
        return _cache[mode]
    cfg = PathConfig(dropout=0.0)
    torch.manual_seed(3)
    tr = Trainer(cfg)
    eng = engine_from_cfg(cfg, B_, P_, T_, dropout=0.0, precision=mode)
    load_oracle_state(eng, tr)
    x, text, text_pad, patches, patch_pad = dev(*synthetic_batch(cfg, B_, P_, T_, seed=9, pad_patches=True))
    g = torch.Generator().manual_seed(1)
    z = torch.randn(B_, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B_, generator=g).cuda()
    S, F, H, nl = P_ + 1, 2 * cfg.embedding_dims, cfg.hidden_dims, cfg.n_layers
    out = {}
    eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
    out["d_losses"] = eng.losses[:3].cpu().clone()
    out["x_fake"] = eng.debug_buffer("X2").view(2 * B_, -1)[:B_].float().cpu()
    out["D.c"] = eng.debug_buffer("D.c").float().cpu()
    out["grad_x_hat"] = eng.debug_buffer("gp_grad").view(B_, -1).cpu()
    out["dD"] = {k: v.detach().cpu().clone() for k, v in eng.state(L.ROLE_CRITIC, "g").items()}
    out["dD_flat"] = eng.flat[L.ROLE_CRITIC]["g"].cpu().clone()
    out["gates_D_enc"] = [(eng.debug_buffer(f"D.L{l}.h").float() > 0).cpu() for l in range(nl)]
    out["gates_D_head"] = [(eng.debug_buffer(n) > 0).cpu() for n in ("headD.a1", "headD.a2")]
    eng.generator_backward(z, patches, patch_pad, text, text_pad)
    out["g_loss"] = eng.losses[3:4].cpu().clone()
    out["G.c"] = eng.debug_buffer("G.c").float().cpu()
    out["dG"] = {k: v.detach().cpu().clone() for k, v in eng.state(L.ROLE_GENERATOR, "g").items()}
    out["dG_flat"] = eng.flat[L.ROLE_GENERATOR]["g"].cpu().clone()
    out["gates_G_enc"] = [(eng.debug_buffer(f"G.L{l}.h").float() > 0).cpu() for l in range(nl)]
    out["gates_G_head"] = [(eng.debug_buffer(n) > 0).cpu() for n in ("headG.a1", "headG.a2")]
    for k in ("d_losses", "g_loss", "x_fake", "D.c", "G.c", "grad_x_hat", "dD_flat", "dG_flat"):
        assert torch.isfinite(out[k]).all(), (mode, k)
    del eng
    torch.cuda.empty_cache()
    _cache[mode] = out
    return out


def test_cfg3_bf16x3_meets_the_parity_gate_against_the_on_chip_fp32_mode():
    ref, got = run_mode("f32"), run_mode("bf16x3")
    ck = Checker("cfg3 full size: bf16x3 vs f32 (one critic + one generator iteration)", 1e-3)
    ck.check("d_real,d_fake,gp", got["d_losses"], ref["d_losses"])
    ck.check("g_loss", got["g_loss"], ref["g_loss"])
    ck.check("x_fake", got["x_fake"], ref["x_fake"])
    ck.check("critic conditioning vector", got["D.c"], ref["D.c"])
    ck.check("generator conditioning vector", got["G.c"], ref["G.c"])
    ck.check("grad_x_hat", got["grad_x_hat"], ref["grad_x_hat"])
    for net in ("dD", "dG"):
        for n, r in ref[net].items():
            if significant(r):
                ck.check(f"{net} {n}", got[net][n], r)
    ck.done()


def test_cfg3_bf16_headline_mode_against_the_on_chip_fp32_mode():
    ref, got = run_mode("f32"), run_mode("bf16")
    gate = Gate("cfg3 full size: bf16 vs f32 (one critic + one generator iteration)")
    gate.check("loss", "d_real,d_fake,gp", got["d_losses"], ref["d_losses"], LOSS_TOL)
    gate.check("loss", "g_loss", got["g_loss"], ref["g_loss"], LOSS_TOL)
    for k in ("x_fake", "D.c", "G.c"):
        gate.check("act", k, got[k], ref[k], ACT_TOL)
    for tag, key in (("critic encoder FFN", "gates_D_enc"), ("critic head", "gates_D_head"), ("generator encoder FFN", "gates_G_enc"),
                     ("generator head", "gates_G_head")):
        flips = sum(int((a != b).sum()) for a, b in zip(got[key], ref[key]))
        n = sum(a.numel() for a in got[key])
        diag(f"   gate flips bf16 vs f32, {tag:22s}: {flips} of {n} ({flips / n:.2e})")
        assert flips < 2e-2 * n, (tag, flips, n)
    gate.check("grad", "grad_x_hat", got["grad_x_hat"], ref["grad_x_hat"], GRAD_TOL)
    for net, tol in (("dD", GRAD_TOL), ("dG", GEN_GRAD_TOL)):
        for n, r in ref[net].items():
            if significant(r) and not n.endswith("in_proj_bias"):
                gate.check("grad", f"{net} {n}", got[net][n], r, tol)
    cd, cg = _cos(got["dD_flat"], ref["dD_flat"]), _cos(got["dG_flat"], ref["dG_flat"])
    diag(f"   flat gradient cosine bf16 vs f32 at full size: critic {cd:.6f}, generator {cg:.6f}")
    assert cd >= 0.99 and cg >= 0.99, (cd, cg)
    gate.done()
