"""The gradient error of the bf16 mode is ReLU gates, not arithmetic - as a test.

tests/test_bf16_parity_gpu.py bounds the bf16 mode's gradient tensors at 0.15 rel-L2 against the fp32 oracle and explains
the 5 - 14 % it measures by ReLU / LeakyReLU gates whose pre-activation lies within bf16 rounding of zero: such a gate lands
on the other side in the two runs and the unit's whole gradient contribution switches on or off.  That was prose.  Here:

  1. one critic iteration in bf16 mode and one in f32 mode on the same weights and inputs; the gate patterns of both runs
     are read back (encoder FFN: stored hidden activations > 0; critic head: post-activations > 0) and the gates that
     DIFFER are counted - a fraction of a per cent, as predicted;
  2. the fp32 CPU oracle (stock torch modules + autograd, double backward included) is run with its activations replaced
     by  y = x * gate  with the gate pattern OF THE BF16 RUN held fixed - the same piecewise-linear function the bf16 kernels
     differentiated;
  3. against that reference every significant gradient tensor of the bf16 run agrees to <= 1e-2 rel-L2 (operand rounding
     only: 2e-3 .. 6e-3 measured), where the free-running oracle differs by several per cent.

So a defect worth even 1 - 2 % of a gradient tensor's norm in attn_bwd_dkv_rm, wgrad_kernel or a wst data-gradient epilogue
fails this test although it would pass the 0.15 gate.  Reference semantics: R:376-423 (train_disc), R:351-374 (gradient_penalty)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from gemm_gan_amd import _lib as L
from gpu_util import dev, diag, engine_from_cfg, load_oracle_state
from test_bf16_parity_gpu import l2, significant
from test_engine_oracle_gpu import setup

pytestmark = pytest.mark.gpu
MASKED_TOL = 1e-2


class ForcedGateFn:
    """y = x * (gate ? 1 : slope) with `gate` fixed: the activation of one call site; masks cycle over the site's calls."""

    def __init__(self, masks, slope=0.0):
        self.masks, self.slope, self.calls = masks, slope, 0

    def __call__(self, x):
        m = self.masks[self.calls % len(self.masks)]
        self.calls += 1
        assert m.shape == x.shape, (m.shape, x.shape)
        return x * torch.where(m, torch.ones_like(x), torch.full_like(x, self.slope))


class ForcedGate(nn.Module):          # the same as a module (a slot of the head's nn.Sequential blocks)
    def __init__(self, masks, slope=0.0):
        super().__init__()
        self.fn = ForcedGateFn(masks, slope)

    def forward(self, x):
        return self.fn(x)


def read_gates(eng, B, S, F, H, nl):
    enc = [eng.debug_buffer(f"D.L{l}.h").view(B, S, F) > 0 for l in range(nl)]
    a1 = eng.debug_buffer("headD.a1").view(3, B, H) > 0
    a2 = eng.debug_buffer("headD.a2").view(3, B, H) > 0
    return [m.cpu() for m in enc], a1.cpu(), a2.cpu()


@pytest.mark.parametrize("case", ["hot_tiles_E256", "cls_tail_S257"])
def test_bf16_gradients_match_fp32_autograd_once_the_gate_pattern_is_held_fixed(case):
    cfg, tr, eng, (x, text, text_pad, patches, patch_pad) = setup(case)
    B, P = x.shape[0], patches.shape[1]
    S, E, F, H, nl = P + 1, cfg.embedding_dims, 2 * cfg.embedding_dims, cfg.hidden_dims, cfg.n_layers
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g)
    alpha = torch.rand(B, 1, generator=g)
    cond = (patches, patch_pad, text, text_pad)
    xg, tg, tpg, pg, ppg, zg, ag = dev(x, text, text_pad, patches, patch_pad, z, alpha)

    # (1) the two engine runs and their gate patterns
    f32 = engine_from_cfg(cfg, B, P, text.shape[1], dropout=0.0)
    load_oracle_state(f32, tr)
    f32.critic_backward(xg, zg, ag, pg, ppg, tg, tpg)
    enc32, a1_32, a2_32 = read_gates(f32, B, S, F, H, nl)
    eng.set_precision("bf16")
    eng.critic_backward(xg, zg, ag, pg, ppg, tg, tpg)
    enc16, a1_16, a2_16 = read_gates(eng, B, S, F, H, nl)
    grads16 = {n: v.detach().cpu().clone() for n, v in eng.state(L.ROLE_CRITIC, "g").items()}
    gx16 = eng.debug_buffer("gp_grad").view(B, -1).cpu()
    n_enc = sum(m.numel() for m in enc16)
    flips_enc = sum(int((a != b).sum()) for a, b in zip(enc16, enc32))
    flips_head = int((a1_16 != a1_32).sum()) + int((a2_16 != a2_32).sum())
    n_head = a1_16.numel() + a2_16.numel()
    diag(f"== gate flips bf16 vs f32 run, {case}: encoder FFN {flips_enc} of {n_enc} ({flips_enc / n_enc:.2e}), "
         f"critic head {flips_head} of {n_head} ({flips_head / n_head:.2e})")
    assert 0 < flips_enc < 2e-2 * n_enc, (flips_enc, n_enc)          # a fraction of a per cent of the gates - and not none
    assert flips_head < 5e-2 * n_head

    # (2) fp32 autograd, free-running and with the bf16 run's gates held fixed
    free = tr.critic_iteration(x, z, alpha, cond, apply=False)
    layers = list(tr.disc.patches_transformer.layers)
    saved = [(lay, lay.activation) for lay in layers]
    blocks = getattr(tr.disc, "discriminator")
    saved_head = [blk[1] for blk in blocks]
    try:
        for lay, m in zip(layers, enc16):
            lay.activation = ForcedGateFn([m], 0.0)        # a plain callable: stays an ordinary attribute
        blocks[0][1] = ForcedGate([a1_16[i] for i in range(3)], cfg.negative_slope)      # calls: D(fake), D(real), D(x^)  (R:403,404,360)
        blocks[1][1] = ForcedGate([a2_16[i] for i in range(3)], cfg.negative_slope)
        forced = tr.critic_iteration(x, z, alpha, cond, apply=False)
    finally:
        for lay, act in saved:
            lay.activation = act
        for blk, act in zip(blocks, saved_head):
            blk[1] = act

    # (3) the residual once the flipped units cannot differ
    bad, worst_free, worst_forced = [], 0.0, 0.0
    rows = [("grad_x_hat", gx16, free["grad_x_hat"].detach(), forced["grad_x_hat"].detach())]
    for n, ref in forced["grads"].items():
        if significant(ref) and not n.endswith("in_proj_bias") and n in grads16:
            rows.append(("dD " + n, grads16[n], free["grads"][n], ref))
    for name, got, ref_free, ref_forced in rows:
        e_free, e_forced = l2(got, ref_free), l2(got, ref_forced)
        worst_free, worst_forced = max(worst_free, e_free), max(worst_forced, e_forced)
        flag = "" if e_forced <= MASKED_TOL else "   <-- FAIL"
        diag(f"   {name:58s} rel-L2 vs free-running fp32 {e_free:.3e} -> gates held fixed {e_forced:.3e}{flag}")
        if flag:
            bad.append((name, e_forced))
    diag(f"   worst: free-running {worst_free:.3e}, gates held fixed {worst_forced:.3e} (bound {MASKED_TOL:g})")
    assert not bad, bad[:8]
    # the losses do not depend on which side of zero a near-zero pre-activation falls (ReLU is continuous)
    l = eng.losses.tolist()
    want = np.array([forced["d_real"].item(), forced["d_fake"].item(), forced["gp"].item()])
    assert np.allclose(np.array(l[:3]), want, rtol=2e-2, atol=2e-3), (l[:3], want)


GEN_MASKED_TOL = 1e-2


@pytest.mark.parametrize("case", ["hot_tiles_E256", "cls_tail_S257"])
def test_bf16_generator_gradients_match_fp32_autograd_once_the_gate_pattern_is_held_fixed(case):
    """The same for the GENERATOR iteration (R:425-461; VERDICT round 3, "missing" 6): its gradient runs backward through the frozen
    critic's head (two layers of gates) into x_fake, then through the generator's own head (the H -> G output layer and two more
    layers of gates) and its conditioning stack (encoder FFN gates).  The fp32 autograd oracle is re-run with all of those gates held
    at the bf16 run's pattern; every significant generator gradient tensor of the bf16 run must then agree to <= 1e-2 rel-L2 (measured 7.7e-3 / 8.3e-3; four
    layers of bf16 products in series: measured below), where the free-running oracle differs by up to tens of per cent on these
    small batches.  A defect worth a few per cent in the generator's backward kernels (head backward through W1x / W3, the
    replica-free conditioning backward) fails here although it passes the 0.25 free-running bound."""
    cfg, tr, eng, (x, text, text_pad, patches, patch_pad) = setup(case)
    B, P = x.shape[0], patches.shape[1]
    S, F, H, nl = P + 1, 2 * cfg.embedding_dims, cfg.hidden_dims, cfg.n_layers
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g)
    cond = (patches, patch_pad, text, text_pad)
    xg, tg, tpg, pg, ppg, zg = dev(x, text, text_pad, patches, patch_pad, z)

    eng.set_precision("bf16")
    eng.generator_backward(zg, pg, ppg, tg, tpg)
    enc16 = [(eng.debug_buffer(f"G.L{l}.h").view(B, S, F) > 0).cpu() for l in range(nl)]
    g1, g2 = (eng.debug_buffer("headG.a1").view(B, H) > 0).cpu(), (eng.debug_buffer("headG.a2").view(B, H) > 0).cpu()
    d1 = (eng.debug_buffer("headD.a1").view(3, B, H)[0] > 0).cpu()
    d2 = (eng.debug_buffer("headD.a2").view(3, B, H)[0] > 0).cpu()
    grads16 = {n: v.detach().cpu().clone() for n, v in eng.state(L.ROLE_GENERATOR, "g").items()}
    loss16 = eng.losses.tolist()[3]

    free = tr.generator_iteration(z, cond, apply=False)
    layers = list(tr.gen.patches_transformer.layers)
    saved = [(lay, lay.activation) for lay in layers]
    gblocks, dblocks = getattr(tr.gen, "generator"), getattr(tr.disc, "discriminator")
    saved_g, saved_d = [blk[1] for blk in gblocks], [blk[1] for blk in dblocks]
    try:
        for lay, m in zip(layers, enc16):
            lay.activation = ForcedGateFn([m], 0.0)
        gblocks[0][1], gblocks[1][1] = ForcedGate([g1], cfg.negative_slope), ForcedGate([g2], cfg.negative_slope)
        dblocks[0][1], dblocks[1][1] = ForcedGate([d1], cfg.negative_slope), ForcedGate([d2], cfg.negative_slope)
        forced = tr.generator_iteration(z, cond, apply=False)
    finally:
        for lay, act in saved:
            lay.activation = act
        for blk, act in zip(gblocks, saved_g):
            blk[1] = act
        for blk, act in zip(dblocks, saved_d):
            blk[1] = act

    diag(f"== generator iteration, bf16 run against fp32 autograd with the bf16 run's gates held fixed, {case}")
    bad, worst_free, worst_forced = [], 0.0, 0.0
    for n, ref in forced["grads"].items():
        if not significant(ref) or n.endswith("in_proj_bias") or n not in grads16:
            continue
        e_free, e_forced = l2(grads16[n], free["grads"][n]), l2(grads16[n], ref)
        worst_free, worst_forced = max(worst_free, e_free), max(worst_forced, e_forced)
        flag = "" if e_forced <= GEN_MASKED_TOL else "   <-- FAIL"
        diag(f"   dG {n:58s} rel-L2 vs free-running fp32 {e_free:.3e} -> gates held fixed {e_forced:.3e}{flag}")
        if flag:
            bad.append((n, e_forced))
    diag(f"   worst: free-running {worst_free:.3e}, gates held fixed {worst_forced:.3e} (bound {GEN_MASKED_TOL:g})")
    assert not bad, bad[:8]
    assert np.allclose(loss16, forced["g_loss"].item(), rtol=2e-2, atol=2e-3), (loss16, forced["g_loss"].item())
