"""HIP engine vs oracle #1 (torch CPU autograd) on seeded inputs at sizes the oracle finishes in
seconds, incl. the hot-path tile shapes (E=256, dh=64, S=65), ragged masks, T>1; plus
size-independent properties at BASELINE.json's full cfg3 size."""
import numpy as np
import pytest
import torch

from gemm_gan_amd import _lib as L
from gpu_util import Checker, dev, engine_from_cfg, load_oracle_state
from oracle.torch_oracle import PathConfig, Trainer, film_config, img_config, set_dropout, synthetic_batch

pytestmark = pytest.mark.gpu
TOL = 1e-3

CASES = {
    "mid_T5_ragged": dict(cfg=PathConfig(n_genes=300, latent_dims=48, embedding_dims=64, hidden_dims=96, text_dims=40,
                                          patch_dims=72, dropout=0.0), B=12, P=37, T=5),
    "hot_tiles_E256": dict(cfg=PathConfig(n_genes=1000, latent_dims=256, embedding_dims=256, hidden_dims=256,
                                           text_dims=512, patch_dims=1024, dropout=0.0), B=8, P=64, T=1),
    "leaky_T300": dict(cfg=PathConfig(n_genes=130, latent_dims=32, embedding_dims=32, hidden_dims=64, text_dims=24,
                                       patch_dims=16, dropout=0.0, negative_slope=0.2), B=6, P=9, T=300),
    # production width with many text tokens (real text has T = 300): the coalesced single-query attention kernels;
    # 77 = four 16-key rounds + a ragged 13
    "text_T77_E256": dict(cfg=PathConfig(n_genes=120, latent_dims=32, embedding_dims=256, hidden_dims=64, text_dims=48,
                                          patch_dims=40, dropout=0.0), B=5, P=20, T=77),
    # FiLM-only sibling (src/conditional_gan_film.py): bias-free encoder, CLS-row conditioning, no clip
    "film_P1": dict(cfg=film_config(n_genes=150, latent_dims=32, embedding_dims=64, hidden_dims=48, text_dims=40,
                                    patch_dims=56, dropout=0.0), B=9, P=1, T=1),
    "film_P33_E256": dict(cfg=film_config(n_genes=90, latent_dims=24, embedding_dims=256, hidden_dims=64, text_dims=48,
                                          patch_dims=64, dropout=0.0), B=4, P=33, T=1),
    # image-transformer sibling (src/conditional_gan_img_transformer.py): Linear-ReLU-LayerNorm patch encoder, no FiLM
    "img_P40_E256": dict(cfg=img_config(n_genes=70, latent_dims=16, embedding_dims=256, hidden_dims=48, text_dims=20,
                                        patch_dims=72, dropout=0.0), B=4, P=40, T=1),
    # B * S = 4 112 rows >= 4 096: the dropout replicas share the layer-0 input and its QKV projection (modulo-indexed operands in the
    # Linear, attention and weight-gradient kernels) - in bf16 mode and, since round 4, in bf16x3
    "share0_S257_E256": dict(cfg=PathConfig(n_genes=200, latent_dims=32, embedding_dims=256, hidden_dims=64, text_dims=48,
                                             patch_dims=96, dropout=0.0), B=16, P=256, T=1),
    # smallest shapes: one sample, one patch (S = 2), one text token - every kernel with a one-workgroup grid
    "single_sample": dict(cfg=PathConfig(n_genes=17, latent_dims=8, embedding_dims=32, hidden_dims=16, text_dims=12,
                                          patch_dims=20, dropout=0.0), B=1, P=1, T=1),
    # S = 601 > the LDS-resident key range of the short attention kernels: in bf16 mode the key-streaming ("long") kernels,
    # in f32 parity mode the unfused (GEMM + softmax + GEMM) route
    "long_S601": dict(cfg=PathConfig(n_genes=64, latent_dims=16, embedding_dims=64, hidden_dims=32, text_dims=24,
                                      patch_dims=40, dropout=0.0), B=2, P=600, T=2),
    # the same at the production head width (dh = 64), several query groups of the long dQ kernel (18 query tiles > 12)
    "long_S545_E256": dict(cfg=PathConfig(n_genes=48, latent_dims=16, embedding_dims=256, hidden_dims=32, text_dims=24,
                                           patch_dims=32, dropout=0.0), B=2, P=544, T=1),
    # the longest sequence the key-streaming kernels take (64 key tiles), ragged tail
    "long_S2047": dict(cfg=PathConfig(n_genes=40, latent_dims=8, embedding_dims=64, hidden_dims=16, text_dims=16,
                                      patch_dims=24, dropout=0.0), B=1, P=2046, T=1),
    # image-transformer sibling with S > 512: the key-streaming attention kernels under GG_VARIANT_IMG (BASELINE configs[4] has
    # S = 1025), Linear-ReLU-LayerNorm patch encoder, bias-free layers, ragged masks
    "img_long_S577_E256": dict(cfg=img_config(n_genes=48, latent_dims=16, embedding_dims=256, hidden_dims=32, text_dims=20,
                                              patch_dims=48, dropout=0.0), B=2, P=576, T=1),
    # S = 257 = eight full 32-row tiles + the CLS row: the left-over query tile is shared by the four waves of a workgroup
    "cls_tail_S257": dict(cfg=PathConfig(n_genes=200, latent_dims=64, embedding_dims=256, hidden_dims=128, text_dims=64,
                                          patch_dims=64, dropout=0.0), B=3, P=256, T=1),
    # edge shapes of the row-major attention kernels (attention.hip, *_rm): S a multiple of 32 (no clamped tile, no zero row),
    # dh = 32; S a multiple of 8 but not of 32 (the zero row is an EXTRA row); S = 129 / 130 / 258: the left-over tile of one /
    # two rows shared by the waves (dQ, dK/dV at four waves: 5 and 9 tiles; forward at eight waves: 9 tiles only), dh = 16
    "tiles_S64_dh32": dict(cfg=PathConfig(n_genes=60, latent_dims=16, embedding_dims=128, hidden_dims=32, text_dims=24,
                                           patch_dims=40, dropout=0.0), B=3, P=63, T=1),
    "rows8_S40": dict(cfg=PathConfig(n_genes=60, latent_dims=16, embedding_dims=256, hidden_dims=32, text_dims=24,
                                      patch_dims=40, dropout=0.0), B=3, P=39, T=1),
    "coop_S129": dict(cfg=PathConfig(n_genes=60, latent_dims=16, embedding_dims=256, hidden_dims=32, text_dims=24,
                                      patch_dims=40, dropout=0.0), B=2, P=128, T=1),
    "coop2_S130_dh16": dict(cfg=PathConfig(n_genes=60, latent_dims=16, embedding_dims=64, hidden_dims=32, text_dims=24,
                                            patch_dims=40, dropout=0.0), B=2, P=129, T=1),
    "coop2_S258": dict(cfg=PathConfig(n_genes=60, latent_dims=16, embedding_dims=256, hidden_dims=32, text_dims=24,
                                       patch_dims=40, dropout=0.0), B=2, P=257, T=1),
}


def setup(case, seed=11):
    c = CASES[case]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(seed)
    tr = Trainer(cfg)
    set_dropout(tr.gen, 0.0)
    set_dropout(tr.disc, 0.0)
    batch = synthetic_batch(cfg, B, P, T, seed=seed + 1, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.0)
    load_oracle_state(eng, tr)
    return cfg, tr, eng, batch


@pytest.mark.usefixtures("parity_mode")
@pytest.mark.parametrize("case", list(CASES))
def test_critic_and_generator_iteration_vs_autograd(case):
    cfg, tr, eng, (x, text, text_pad, patches, patch_pad) = setup(case)
    B = x.shape[0]
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g)
    alpha = torch.rand(B, 1, generator=g)
    cond = (patches, patch_pad, text, text_pad)
    ck = Checker(f"oracle critic/gen iteration {case}", TOL)
    r = tr.critic_iteration(x, z, alpha, cond, apply=False)
    xg, tg, tpg, pg, ppg, zg, ag = dev(x, text, text_pad, patches, patch_pad, z, alpha)
    eng.critic_backward(xg, zg, ag, pg, ppg, tg, tpg)
    l = eng.losses.tolist()
    ck.check("d_real,d_fake,gp", np.array(l[:3]), np.array([r["d_real"].item(), r["d_fake"].item(), r["gp"].item()]))
    ck.check("x_fake", eng.debug_buffer("X2").view(2 * B, -1)[:B], r["x_fake"])
    ck.check("grad_x_hat", eng.debug_buffer("gp_grad").view(B, -1), r["grad_x_hat"].detach())
    grads = eng.state(L.ROLE_CRITIC, "g")
    for n, ref in r["grads"].items():
        if ref is not None:
            ck.check("dD " + n, grads[n], ref)
    rg = tr.generator_iteration(z, cond, apply=False)
    eng.generator_backward(zg, pg, ppg, tg, tpg)
    ck.check("g_loss", np.array([eng.losses.tolist()[3]]), np.array([rg["g_loss"].item()]))
    grads = eng.state(L.ROLE_GENERATOR, "g")
    for n, ref in rg["grads"].items():
        if ref is not None:
            ck.check("dG " + n, grads[n], ref)
    ck.done()


@pytest.mark.usefixtures("parity_mode")
@pytest.mark.parametrize("case", ["text_T77_E256", "mid_T5_ragged", "hot_tiles_E256", "film_P33_E256", "img_P40_E256", "share0_S257_E256"])
def test_replica_stacked_passes_vs_autograd(case):
    """With dropout on, the three critic passes of an iteration run as replicas stacked on the batch axis (two of them
    carry gradient; shared layer inputs, shared text keys, replica-summed gradients).  A drop probability of 1e-7 keeps
    every element (threshold round(p * 65536) = 0, scale 1 + 1e-7), so that route must reproduce the dropout-free
    autograd reference within the parity tolerance."""
    c = CASES[case]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    set_dropout(tr.gen, 0.0)
    set_dropout(tr.disc, 0.0)
    x, text, text_pad, patches, patch_pad = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=1e-7)
    load_oracle_state(eng, tr)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g)
    alpha = torch.rand(B, 1, generator=g)
    cond = (patches, patch_pad, text, text_pad)
    ck = Checker(f"replica-stacked critic/gen iteration {case}", TOL)
    r = tr.critic_iteration(x, z, alpha, cond, apply=False)
    xg, tg, tpg, pg, ppg, zg, ag = dev(x, text, text_pad, patches, patch_pad, z, alpha)
    eng.critic_backward(xg, zg, ag, pg, ppg, tg, tpg)
    l = eng.losses.tolist()
    ck.check("d_real,d_fake,gp", np.array(l[:3]), np.array([r["d_real"].item(), r["d_fake"].item(), r["gp"].item()]))
    grads = eng.state(L.ROLE_CRITIC, "g")
    for n, ref in r["grads"].items():
        if ref is not None:
            ck.check("dD " + n, grads[n], ref)
    rg = tr.generator_iteration(z, cond, apply=False)
    eng.generator_backward(zg, pg, ppg, tg, tpg)
    ck.check("g_loss", np.array([eng.losses.tolist()[3]]), np.array([rg["g_loss"].item()]))
    grads = eng.state(L.ROLE_GENERATOR, "g")
    for n, ref in rg["grads"].items():
        if ref is not None:
            ck.check("dG " + n, grads[n], ref)
    ck.done()


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
def test_full_size_properties_cfg3(mode):
    """BASELINE cfg3 (B=256, G=5000, P=256, Dp=1024, T=1, Dt=512): size-independent properties, in every arithmetic mode the bench
    quotes (f32: the exact reference; bf16x3: `parity_mode`; bf16: the headline).
    (1) critic score is independent of the batch composition (row b of a half batch == row b of the
    full batch); (2) data-parallel identity: the gradient of the full batch equals the mean of the two
    half-batch gradients; (3) everything finite; (4) GP gradient norm matches the row norm of grad_x_hat.
    Bounds: f32 / bf16x3 1e-5 and 2e-4 (fp32 accumulation order); bf16 1e-3 (bf16 operand products whose split over workgroups
    depends on the row count: accumulation order of 8-bit-mantissa terms)."""
    cfg = PathConfig(dropout=0.0)
    B, P, T = 256, 256, 1
    torch.manual_seed(3)
    tr = Trainer(cfg)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.0, precision=mode)
    t_row, t_dp = (1e-3, 1e-3) if mode == "bf16" else (1e-5, 2e-4)
    load_oracle_state(eng, tr)
    x, text, text_pad, patches, patch_pad = dev(*synthetic_batch(cfg, B, P, T, seed=9, pad_patches=True))
    g = torch.Generator().manual_seed(1)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    ck = Checker(f"full-size cfg3 properties ({mode})", 1e-3, metric="max")
    full = eng.forward(L.ROLE_CRITIC, x, patches, patch_pad, text, text_pad, train=False)
    h = B // 2
    half = eng.forward(L.ROLE_CRITIC, x[h:].contiguous(), patches[h:].contiguous(), patch_pad[h:].contiguous(),
                       text[h:].contiguous(), text_pad[h:].contiguous(), train=False)
    assert torch.isfinite(full).all()
    ck.check("critic rows independent of batch", half, full[h:], tol=t_row)
    eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
    gfull = eng.flat[L.ROLE_CRITIC]["g"].clone()
    nrm = eng.debug_buffer("gp_grad").view(B, -1).norm(dim=1)
    ck.check("|grad_x_hat| vs gp_nrm2", nrm ** 2, eng.debug_buffer("gp_nrm2"), tol=1e-5)
    assert torch.isfinite(gfull).all() and torch.isfinite(eng.losses).all()
    acc = torch.zeros_like(gfull)
    for s in (slice(0, h), slice(h, B)):
        eng.critic_backward(x[s].contiguous(), z[s].contiguous(), alpha[s].contiguous(), patches[s].contiguous(),
                            patch_pad[s].contiguous(), text[s].contiguous(), text_pad[s].contiguous())
        acc += eng.flat[L.ROLE_CRITIC]["g"]
    ck.check("DP identity: mean of shard grads == full-batch grad", acc / 2, gfull, tol=t_dp)
    ck.done()


def test_full_size_properties_cfg5_rank():
    """Per-GPU shape of BASELINE configs[4] (conditional_gan_img_transformer.py: B = 128 per rank, 18 000 genes, 1 024 patch
    tokens -> S = 1 025, the key-streaming attention kernels), bf16 mode as that configuration runs: (1) finite outputs and
    gradients; (2) critic rows independent of the batch composition; (3) data-parallel identity (mean of the shard
    gradients == full-batch gradient); (4) the fused long-sequence attention route against the unfused GEMM + softmax +
    GEMM route of the same mode; (5) |grad_x^| against the row norms of the stored gradient."""
    cfg = img_config(n_genes=18000, dropout=0.0)
    B, P, T = 128, 1024, 1
    torch.manual_seed(3)
    tr = Trainer(cfg)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.0)
    eng.set_precision("bf16")
    load_oracle_state(eng, tr)
    x, text, text_pad, patches, patch_pad = dev(*synthetic_batch(cfg, B, P, T, seed=9, pad_patches=True))
    g = torch.Generator().manual_seed(1)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    ck = Checker("full-size cfg5-per-rank properties (img variant, bf16)", 1e-3, metric="max")
    full = eng.forward(L.ROLE_CRITIC, x, patches, patch_pad, text, text_pad, train=False)
    h = B // 2
    half = eng.forward(L.ROLE_CRITIC, x[h:].contiguous(), patches[h:].contiguous(), patch_pad[h:].contiguous(),
                       text[h:].contiguous(), text_pad[h:].contiguous(), train=False)
    assert torch.isfinite(full).all()
    # 18 000-term bf16 dot products whose split over workgroups depends on the row count: accumulation order only
    ck.check("critic rows independent of batch", half, full[h:], tol=1e-3)
    xgen = eng.forward(L.ROLE_GENERATOR, z, patches, patch_pad, text, text_pad, train=False)
    assert tuple(xgen.shape) == (B, 18000) and torch.isfinite(xgen).all()
    eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
    gfull = eng.flat[L.ROLE_CRITIC]["g"].clone()
    lfull = eng.losses.clone()
    nrm = eng.debug_buffer("gp_grad").view(B, -1).norm(dim=1)
    ck.check("|grad_x_hat| vs gp_nrm2", nrm ** 2, eng.debug_buffer("gp_nrm2"), tol=1e-5)
    assert torch.isfinite(gfull).all() and torch.isfinite(lfull).all()
    acc = torch.zeros_like(gfull)
    for s in (slice(0, h), slice(h, B)):
        eng.critic_backward(x[s].contiguous(), z[s].contiguous(), alpha[s].contiguous(), patches[s].contiguous(),
                            patch_pad[s].contiguous(), text[s].contiguous(), text_pad[s].contiguous())
        acc += eng.flat[L.ROLE_CRITIC]["g"]
    ck.check("DP identity: mean of shard grads == full-batch grad", acc / 2, gfull, tol=1e-3)
    eng.set_flash(False)                      # unfused attention route of the same precision mode
    unf = eng.forward(L.ROLE_CRITIC, x, patches, patch_pad, text, text_pad, train=False)
    ck.check("critic score: key-streaming attention vs unfused", full, unf, tol=2e-2)
    eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
    gunf = eng.flat[L.ROLE_CRITIC]["g"]
    cs = _cos(gfull, gunf)
    from gpu_util import diag
    diag(f"   flat critic gradient cosine, key-streaming vs unfused attention at S = 1025: {cs:.5f}")
    assert cs >= 0.995, cs
    ck.check("losses: key-streaming vs unfused", lfull[:3], eng.losses[:3], tol=2e-2)
    ck.done()


def test_dropout_statistics_and_replicas():
    """p=0.1: attention-probability dropout zeroes ~10 % of the unmasked probabilities, survivors are
    scaled by 1/(1-p); the three critic passes of one iteration draw independent masks; backward
    regenerates the same masks (finite-difference free check: gradients stay finite and the loss
    of a second identical call with the same seed reproduces)."""
    cfg = PathConfig(n_genes=200, latent_dims=32, embedding_dims=64, hidden_dims=64, text_dims=24, patch_dims=48, dropout=0.1)
    B, P, T = 16, 31, 2
    torch.manual_seed(2)
    tr = Trainer(cfg)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=77)
    load_oracle_state(eng, tr)
    x, text, text_pad, patches, patch_pad = dev(*synthetic_batch(cfg, B, P, T, seed=4))
    g = torch.Generator().manual_seed(1)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
    S = P + 1
    # backward re-materialises the dropped probabilities layer by layer, ending with layer 0, for the
    # two replicas that carry gradient: the first 2B slabs of sPd are layer 0's regenerated mask
    Pm = eng.debug_buffer("D.L0.P").view(3 * B, 4, S, S)[:2 * B]
    Pd = eng.debug_buffer("sPd").view(3 * B, 4, S, S)[:2 * B]
    zero_frac = float(((Pd == 0) & (Pm > 0)).float().sum() / (Pm > 0).float().sum())
    assert 0.09 < zero_frac < 0.11, zero_frac
    kept = Pd != 0
    assert torch.allclose(Pd[kept], Pm[kept] / 0.9, rtol=1e-5)
    c = eng.debug_buffer("D.c").view(3, B, -1)
    assert (c[0] - c[1]).abs().max() > 1e-4 and (c[1] - c[2]).abs().max() > 1e-4   # independent draws
    l1 = eng.losses.clone()
    g1 = eng.flat[L.ROLE_CRITIC]["g"].clone()
    assert torch.isfinite(g1).all()
    eng.set_seed(77)
    eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
    assert torch.allclose(eng.losses, l1, rtol=1e-5, atol=1e-6)
    assert (eng.flat[L.ROLE_CRITIC]["g"] - g1).abs().max() <= 1e-4 * g1.abs().max()


def _cos(a, b):
    a = a.double().reshape(-1).cpu()
    b = b.double().reshape(-1).cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


@pytest.mark.parametrize("flash", [False, True])
def test_bf16_mode_tracks_fp32_oracle(flash):
    """Throughput mode (bf16 MFMA operands, fp32 accumulate), with and without the fused attention
    kernels.  bf16 keeps 8 significant bits; on an 8-sample batch a handful of ReLU gates flip under that
    rounding, so individual gradient tensors move by several percent.  Gates: losses / generated genes
    within 4e-2 of the fp32 oracle, every gradient tensor's direction within cos >= 0.9 and the whole
    critic gradient within cos >= 0.97 - a mapping or logic bug gives cos ~ 0."""
    cfg, tr, eng, (x, text, text_pad, patches, patch_pad) = setup("hot_tiles_E256")
    eng.set_precision("bf16")
    eng.set_flash(flash)
    B = x.shape[0]
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g)
    alpha = torch.rand(B, 1, generator=g)
    cond = (patches, patch_pad, text, text_pad)
    ck = Checker(f"bf16 mode vs fp32 oracle (hot_tiles_E256, flash={flash})", 4e-2, metric="max")
    r = tr.critic_iteration(x, z, alpha, cond, apply=False)
    xg, tg, tpg, pg, ppg, zg, ag = dev(x, text, text_pad, patches, patch_pad, z, alpha)
    eng.critic_backward(xg, zg, ag, pg, ppg, tg, tpg)
    l = eng.losses.tolist()
    ck.check("d_real,d_fake,gp", np.array(l[:3]), np.array([r["d_real"].item(), r["d_fake"].item(), r["gp"].item()]))
    ck.check("x_fake", eng.debug_buffer("X2").view(2 * B, -1)[:B], r["x_fake"])
    grads = eng.state(L.ROLE_CRITIC, "g")
    flat_a, flat_b, worst = [], [], (1.0, "")
    for n, ref in r["grads"].items():
        if ref is None or n.endswith("in_proj_bias") or ref.abs().max() < 1e-12:
            continue
        cs = _cos(grads[n], ref)
        worst = min(worst, (cs, n))
        flat_a.append(grads[n].reshape(-1).cpu())
        flat_b.append(ref.reshape(-1))
    total = _cos(torch.cat(flat_a), torch.cat(flat_b))
    from gpu_util import diag
    diag(f"   gradient cosine: total {total:.4f}, worst tensor {worst[1]} {worst[0]:.4f}")
    assert total >= 0.97 and worst[0] >= 0.9, (total, worst)
    ck.done()


@pytest.mark.parametrize("case", ["hot_tiles_E256", "mid_T5_ragged", "cls_tail_S257", "long_S601", "long_S545_E256", "long_S2047",
                                  "img_long_S577_E256", "tiles_S64_dh32", "rows8_S40", "coop_S129", "coop2_S130_dh16", "coop2_S258"])
@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_flash_attention_matches_unfused_path(case, dropout):
    """Fused attention forward/backward (attention.hip) against the unfused bf16 path (GEMM + softmax +
    GEMM) inside the same engine: same operands, same rounding points (bf16 Q/K/V/P, fp32 softmax), and -
    because both index the dropout hash by ((n*nh+h)*S+q)*S+key - the SAME dropout masks, so the
    comparison holds with dropout on.  dh = 64 (S=65) and dh = 16 with ragged key padding."""
    c = CASES[case]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=5)
    load_oracle_state(eng, tr)
    eng.set_dropout(dropout)
    eng.set_precision("bf16")
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    R = 3 if dropout > 0 else 1
    S, E = P + 1, cfg.embedding_dims
    out = {}
    for flash in (False, True):
        eng.set_flash(flash)
        eng.set_seed(5)
        eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
        out[flash] = dict(losses=eng.losses.clone(), ctx0=eng.debug_buffer("D.L0.ctx"), ctx1=eng.debug_buffer("D.L1.ctx"),
                          c=eng.debug_buffer("D.c"), g=eng.flat[L.ROLE_CRITIC]["g"].clone())
    ck = Checker(f"flash vs unfused bf16 {case} dropout={dropout}", 2e-2, metric="max")
    a, b = out[True], out[False]
    ck.check("layer-0 attention context", a["ctx0"], b["ctx0"], tol=1e-2)
    ck.check("layer-1 attention context", a["ctx1"], b["ctx1"], tol=1e-2)
    ck.check("conditioning vector", a["c"], b["c"], tol=1e-2)
    ck.check("losses", a["losses"][:3], b["losses"][:3])
    cs = _cos(a["g"], b["g"])
    from gpu_util import diag
    diag(f"   flat critic gradient cosine flash vs unfused: {cs:.5f}")
    # two-sample batches: ONE ReLU gate of the 2 x H head units landing on the other side of zero between the two routes
    # (they agree to 3e-3 on the activations) moves the whole upstream gradient by percents - see tests/test_bf16_parity_gpu.py
    assert cs > (0.95 if CASES[case]["B"] <= 2 else 0.995), cs
    ck.done()


@pytest.mark.parametrize("xstore", [True, False])
@pytest.mark.parametrize("case", ["hot_tiles_E256", "mid_T5_ragged"])
@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_token_on_lane_linear_matches_tile_gemm(case, dropout, xstore):
    """tlin.hip (activations register-resident, fused bias/ReLU/dropout/residual/LayerNorm/mask epilogues)
    against the generic tile GEMM + separate row kernels, both in bf16 mode: the operands are rounded to
    the same bf16 values and the dropout hash is indexed identically, so only the fp32 summation order
    differs.  Covers K = 1024 with fused FiLM and the CLS row remap (patch encoder), K = 768 accumulate
    (in_proj backward), N = 1024 stream (FiLM backward) on the E = 256 case; E = 64 exercises NT = 2."""
    c = CASES[case]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=5)
    load_oracle_state(eng, tr)
    eng.set_dropout(dropout)
    eng.set_precision("bf16")
    # xstore: bf16 storage of the encoder's LayerNorm outputs and pre-LayerNorm sums (the default at the production width).  With it OFF the
    # round-2 gates apply unchanged (activations 2e-3, gradient cosine 0.999): the looser ones below are for the extra rounding only
    eng.set_xstore(xstore)
    xs = xstore and cfg.embedding_dims == 256
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    out = {}
    names = ["D.x0", "D.L0.qkv", "D.L0.x1", "D.L0.h", "D.L0.x2", "D.L1.x2", "D.t2i_kv", "D.c", "G.c"]
    gnames = ["G.x0", "G.L0.qkv", "G.L0.ctx", "G.L0.x1", "G.L0.h", "G.L0.x2", "G.L1.x2", "G.c", "D.c", "dxfake", "dc"]
    for on in (False, True):
        eng.set_tlin(on)
        eng.set_seed(5)
        eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
        out[on] = dict(losses=eng.losses.clone(), g=eng.flat[L.ROLE_CRITIC]["g"].clone(),
                       **{n: eng.debug_buffer(n) for n in names})
        eng.generator_backward(z, patches, patch_pad, text, text_pad)
        out[on]["gg"] = eng.flat[L.ROLE_GENERATOR]["g"].clone()
        out[on]["gl"] = eng.losses.clone()
        out[on]["gstate"] = {k: v.clone() for k, v in eng.state(L.ROLE_GENERATOR, "g").items()}
        for nme in gnames:
            out[on]["gen:" + nme] = eng.debug_buffer(nme)
    ck = Checker(f"tlin vs tile GEMM (bf16) {case} dropout={dropout}", 2e-3, metric="max")
    a, b = out[True], out[False]
    S_, F_ = P + 1, 2 * cfg.embedding_dims
    for n in names:
        # qkv / h are stored in bf16 on the tlin path (2^-9 element rounding), fp32 on the tile-GEMM path
        u, v = a[n], b[n]          # every row: the two-launch feed-forward route stores the forward-only replica's hidden rows too
        # x1 / x2 (all but the last layer's) are bf16-stored too at the production width when xstore is on (one 2^-9 rounding; the rounded
        # residual moves the last layer's fp32 output and the conditioning vector by a little over 2e-3 of their largest element)
        loose = ("qkv", "h", "ctx", "x1", "x2", "c") if xs else ("qkv", "h", "ctx")
        ck.check(n, u, v, tol=6e-3 if n.split(".")[-1] in loose else None)
    for n in gnames:
        # the critic's conditioning vector differs by ~5e-4 between the two bf16 paths (values that sit on a bf16
        # rounding boundary re-round differently); on these 6-12 sample batches that flips a few ReLU gates of
        # the critic head, which moves d(loss)/d(x_fake) and everything downstream by percents
        ck.check("generator pass " + n, a["gen:" + n], b["gen:" + n],
                 tol=0.25 if n in ("dxfake", "dc") else (6e-3 if n.split(".")[-1] in loose else None))
    for k in a["gstate"]:
        from gpu_util import diag as _d
        _d(f"      dG {k:60s} cos {_cos(a['gstate'][k], b['gstate'][k]):.6f}")
    ck.check("critic losses", a["losses"][:3], b["losses"][:3], tol=1e-2)
    ck.check("generator loss", a["gl"][3:4], b["gl"][3:4], tol=1e-2)
    from gpu_util import diag
    cd, cg = _cos(a["g"], b["g"]), _cos(a["gg"], b["gg"])
    diag(f"   flat gradient cosine tlin vs generic: critic {cd:.6f} generator {cg:.6f}")
    # 0.997: with the bf16-stored LayerNorm outputs the rounded residual flips as many FFN gates again as the operand rounding does
    # (tests/test_gate_flips_gpu.py holds the gates fixed and bounds what is left at 1e-2); measured 0.9982 at dropout 0.1
    assert cd > (0.997 if xs else 0.999) and cg > 0.99, (cd, cg)
    ck.done()


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_bf16_storage_of_the_layernorm_outputs_changes_only_the_residual_rounding(dropout):
    """engine.hip "xst": at the production width the encoder's LayerNorm outputs x1 (both layers) and x2 (all but the last) are
    stored in bf16.  Their MFMA consumers round to bf16 on load anyway (identical products); the only new rounding is the residual
    operand of the next sub-block: stored values equal the fp32 ones to one bf16 rounding (2^-9), downstream fp32 tensors move by
    parts in a thousand, losses and the gradient direction stay put."""
    c = CASES["hot_tiles_E256"]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=5)
    load_oracle_state(eng, tr)
    eng.set_dropout(dropout)
    eng.set_precision("bf16")
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    names = ["D.L0.x1", "D.L0.x2", "D.L1.x1", "D.L1.x2", "D.c"]
    out = {}
    for on in (False, True):
        eng.set_xstore(on)          # (the pre-LayerNorm sums r1 / r2 follow: bf16 too)
        eng.set_seed(5)
        eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
        out[on] = dict(losses=eng.losses.clone(), g=eng.flat[L.ROLE_CRITIC]["g"].clone(),
                       **{n: eng.debug_buffer(n) for n in names},
                       **{"bf16:" + n: eng.lib.gg_debug_buffer_is_bf16(eng.h, n.encode()) for n in names})
    a, b = out[True], out[False]
    assert [a["bf16:" + n] for n in names] == [1, 1, 1, 0, 0] and not any(b["bf16:" + n] for n in names)     # the last layer's output stays fp32
    ck = Checker(f"bf16 LayerNorm-output storage on vs off, dropout={dropout}", 6e-3, metric="max")
    for n in names:
        ck.check(n, a[n].float(), b[n])
    ck.check("critic losses", a["losses"][:3], b["losses"][:3], tol=1e-2)
    cs = _cos(a["g"], b["g"])
    from gpu_util import diag
    diag(f"   flat critic gradient cosine bf16 LN-output storage on vs off: {cs:.6f}")
    assert cs > 0.997, cs           # gate flips on this 6-sample batch (see test_gate_flips_gpu.py); measured 0.9981 at dropout 0.1
    ck.done()


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_fused_accumulate_layernorm_backward_matches_the_two_launch_route(dropout):
    """wst.hip EPI_LNB inside a critic iteration against `dx1 += dh W1` followed by ln_bwd_v4_k: the same fp32 arithmetic up to summation
    order; where that moves a value across a rounding boundary of the bf16-stored branch gradient, downstream gradient tensors move
    by up to ~2e-3 of their largest element (measured 1.7e-3 on film_generator.weight; most tensors 1e-5).  One launch fewer per
    encoder layer.  The kernel itself is held to float64 in tests/test_kernels_gpu.py."""
    c = CASES["hot_tiles_E256"]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=5)
    load_oracle_state(eng, tr)
    eng.set_dropout(dropout)
    eng.set_precision("bf16")
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    out = {}
    for on in (False, True):
        eng.set_lnb_fused(on)
        eng.set_seed(5)
        eng.reset_launch_count()
        eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
        out[on] = dict(losses=eng.losses.clone(), launches=eng.launch_count(),
                       g={k: v.clone() for k, v in eng.state(L.ROLE_CRITIC, "g").items()})
    a, b = out[True], out[False]
    assert b["launches"] - a["launches"] == cfg.n_layers, (a["launches"], b["launches"])
    assert torch.allclose(a["losses"], b["losses"], rtol=1e-5, atol=1e-6)      # same forward launches (the loss slots are atomic sums)
    ck = Checker(f"+= / LayerNorm backward in one kernel vs two, dropout={dropout}", 5e-3, metric="max")
    for k in a["g"]:
        ck.check("dD " + k, a["g"][k], b["g"][k])
    ck.done()


@pytest.mark.parametrize("case", ["hot_tiles_E256", "mid_T5_ragged"])
def test_fused_mlp_head_route_matches_the_launch_per_product_route(case):
    """head.hip (opt-in: gg_set_head_fused) inside a critic and a generator iteration against the lin_fwd / lin_bwd_data / k_act_bwd
    chain: same bf16 operand roundings, only fp32 summation order differs (a value of a1 / dh2 / dh1 on a bf16 rounding boundary
    may round the other way as an operand of the next product)."""
    c = CASES[case]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.0, seed=5)
    load_oracle_state(eng, tr)
    eng.set_precision("bf16")
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    out = {}
    for on in (False, True):
        eng.set_head_fused(on)
        eng.set_seed(5)
        eng.reset_launch_count()
        eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
        n_crit = eng.launch_count()
        out[on] = dict(losses=eng.losses.clone(), launches=n_crit, g={k: v.clone() for k, v in eng.state(L.ROLE_CRITIC, "g").items()})
        eng.generator_backward(z, patches, patch_pad, text, text_pad)
        out[on]["gl"] = eng.losses.clone()
        out[on]["gg"] = {k: v.clone() for k, v in eng.state(L.ROLE_GENERATOR, "g").items()}
    a, b = out[True], out[False]
    if cfg.embedding_dims % 32 == 0 and cfg.hidden_dims % 32 == 0:
        assert a["launches"] < b["launches"], (a["launches"], b["launches"])          # the fused kernels really ran
    ck = Checker(f"fused MLP head vs launch-per-product, {case}", 5e-3, metric="max")
    ck.check("critic losses", a["losses"][:3], b["losses"][:3], tol=2e-3)
    ck.check("generator loss", a["gl"][3:4], b["gl"][3:4], tol=2e-3)
    from test_bf16_parity_gpu import significant
    for k in a["g"]:
        if significant(b["g"][k]):
            ck.check("dD " + k, a["g"][k], b["g"][k])
    for k in a["gg"]:
        if significant(b["gg"][k]):
            ck.check("dG " + k, a["gg"][k], b["gg"][k], tol=5e-2)        # a flipped critic-head gate moves d(loss)/d(x_fake): see the tlin test above
    ck.done()


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_fused_feed_forward_route_matches_the_two_launch_route(dropout):
    """ffn.hip inside a critic iteration (opt-in: gg_set_ffn_fused) against FFN1 + FFN2 as two token-on-lane Linears: same
    operands, same dropout streams; only the bf16 rounding of a hidden value that sits on a tie and the fp32 summation order
    of the second product differ."""
    c = CASES["hot_tiles_E256"]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=5)
    load_oracle_state(eng, tr)
    eng.set_dropout(dropout)
    eng.set_precision("bf16")
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    names = ["D.L0.x1", "D.L0.h", "D.L0.x2", "D.L1.x2", "D.c"]
    out = {}
    eng.set_xstore(False)          # the fused kernel reads x1 as fp32: the engine uses it only without bf16 LayerNorm-output storage
    for on in (False, True):
        eng.set_ffn_fused(on)
        eng.set_seed(5)
        eng.reset_launch_count()
        eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
        out[on] = dict(losses=eng.losses.clone(), g=eng.flat[L.ROLE_CRITIC]["g"].clone(), launches=eng.launch_count(),
                       **{n: eng.debug_buffer(n) for n in names})
    a, b = out[True], out[False]
    saved = b["launches"] - a["launches"]        # one launch fewer per encoder layer and encoder pass (critic's, generator's)
    assert saved > 0 and saved % cfg.n_layers == 0, (a["launches"], b["launches"])
    ck = Checker(f"fused feed-forward vs two launches, dropout={dropout}", 2e-3, metric="max")
    S_, F_ = P + 1, 2 * cfg.embedding_dims
    for n in names:
        u, v = a[n], b[n]
        if n.endswith(".h"):        # the forward-only replica's hidden tile is not stored by the fused kernel
            u, v = u.view(-1, F_)[: 2 * B * S_], v.view(-1, F_)[: 2 * B * S_]
        ck.check(n, u, v, tol=6e-3 if n.endswith(".h") else None)
    ck.check("critic losses", a["losses"][:3], b["losses"][:3], tol=1e-2)
    cs = _cos(a["g"], b["g"])
    from gpu_util import diag
    diag(f"   flat critic gradient cosine fused vs two launches: {cs:.6f}")
    assert cs > 0.999, cs
    ck.done()


@pytest.mark.parametrize("case", ["hot_tiles_E256", "cls_tail_S257"])
@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_fused_layer_backward_route_matches_the_three_launch_route(dropout, case):
    """enc.hip encb_kernel (gg_set_encb) inside a critic and a generator iteration against the gate, += / LayerNorm-backward and context-gradient
    Linears as three weight-stationary launches: same operands and dropout streams, same bf16 storage; fp32 summation order and the bf16
    rounding of values on a tie differ."""
    c = CASES[case]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=5)
    load_oracle_state(eng, tr)
    eng.set_dropout(dropout)
    eng.set_precision("bf16")
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    out = {}
    for on in (False, True):
        eng.set_encb(on)
        eng.set_seed(5)
        eng.reset_launch_count()
        eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
        out[on] = dict(losses=eng.losses.clone(), g=eng.flat[L.ROLE_CRITIC]["g"].clone(), launches=eng.launch_count(),
                       gs={k: v.clone() for k, v in eng.state(L.ROLE_CRITIC, "g").items()})
        eng.generator_backward(z, patches, patch_pad, text, text_pad)
        out[on]["gg"] = eng.flat[L.ROLE_GENERATOR]["g"].clone()
    eng.set_encb(False)
    a, b = out[True], out[False]
    assert a["launches"] < b["launches"], (a["launches"], b["launches"])           # two launches less per layer, one more per shadow refresh
    ck = Checker(f"fused layer backward vs three launches, {case}, dropout={dropout}", 5e-3, metric="max")
    ck.check("critic losses", a["losses"][:3], b["losses"][:3], tol=1e-6)        # the forward pass is the same code
    from test_bf16_parity_gpu import significant
    for k in a["gs"]:
        if significant(b["gs"][k]):
            ck.check("dD " + k, a["gs"][k], b["gs"][k])
    cd, cg = _cos(a["g"], b["g"]), _cos(a["gg"], b["gg"])
    from gpu_util import diag
    diag(f"   flat gradient cosine fused vs three launches: critic {cd:.6f}, generator {cg:.6f}")
    assert cd > 0.9999 and cg > 0.9999, (cd, cg)
    ck.done()


@pytest.mark.parametrize("grid", [0, 2])
@pytest.mark.parametrize("mode", [1, 3])
@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_streamed_feed_forward_route_matches_the_two_launch_route(dropout, mode, grid):
    """enc.hip (gg_set_ffn2) inside a critic iteration against FFN1 + FFN2 as two weight-stationary Linears, default bf16 storage
    (bf16 x1 / r2 / x2): same operands and dropout streams; only the bf16 rounding of a hidden value on a tie and the fp32 summation
    order differ.  Every stored row is compared (h / r2 of the forward-only replica are written by neither route... by the
    two-launch route only for h: compared on the rows both store).
    grid = 2 caps the persistent grid at two workgroups (512 tokens per pass): the 3 * 8 * 65 = 1 560 token rows then split into three
    whole passes for the streamed kernel and 24 left-over rows for the two Linear launches, as 3 * 256 * 257 rows do on 256 compute
    units - the left-over rows run the SAME kernels with the SAME dropout words as the all-two-launch route (DropKey::post), so they
    must come out bit-identical."""
    c = CASES["hot_tiles_E256"]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=5)
    load_oracle_state(eng, tr)
    eng.set_dropout(dropout)
    eng.set_precision("bf16")
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    names = ["D.L0.x1", "D.L0.h", "D.L0.x2", "D.L1.x2", "D.c"]
    out = {}
    lib = L.load()
    for on in (0, mode):
        eng.set_ffn2(on)
        L.check(lib.gg_test_set_enc_grid(grid if on else 0))
        eng.set_seed(5)
        eng.reset_launch_count()
        eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
        out[on] = dict(losses=eng.losses.clone(), g=eng.flat[L.ROLE_CRITIC]["g"].clone(), launches=eng.launch_count(),
                       **{n: eng.debug_buffer(n) for n in names})
    eng.set_ffn2(0)
    L.check(lib.gg_test_set_enc_grid(0))
    a, b = out[mode], out[0]
    assert a["launches"] != b["launches"], (a["launches"], b["launches"])          # the fused kernel really ran (one launch less per layer, one more per shadow refresh)
    if grid:
        R0 = 3 if dropout > 0 else 1
        M0, sweep = R0 * B * (P + 1), grid * 256
        main = M0 // sweep * sweep
        assert 0 < main < M0 and (M0 - main) * 8 <= sweep, (M0, main)               # the split is really taken at this shape
        u, v = a["D.L0.x2"].view(-1, cfg.embedding_dims)[main:M0], b["D.L0.x2"].view(-1, cfg.embedding_dims)[main:M0]
        assert torch.equal(u, v), "left-over rows of layer 0 differ from the two-launch route (same kernels, same dropout words expected)"
    ck = Checker(f"streamed feed-forward vs two launches, dropout={dropout}, mode={mode}", 6e-3, metric="max")
    R_ = 3 if dropout > 0 else 1
    S_, F_ = P + 1, 2 * cfg.embedding_dims
    kept = (2 if R_ == 3 else 1) * B * S_
    for n in names:
        u, v = a[n].float(), b[n].float()
        if n.endswith(".h"):
            u, v = u.view(-1, F_)[:kept], v.view(-1, F_)[:kept]
        ck.check(n, u, v)
    ck.check("critic losses", a["losses"][:3], b["losses"][:3], tol=1e-2)
    cs = _cos(a["g"], b["g"])
    from gpu_util import diag
    diag(f"   flat critic gradient cosine streamed vs two launches: {cs:.6f}")
    assert cs > 0.999, cs
    ck.done()


@pytest.mark.parametrize("case", ["hot_tiles_E256", "mid_T5_ragged"])
def test_bf16_operand_storage_is_numerically_transparent(case):
    """Storing the MFMA-operand-only tensors (qkv, attention context, FFN hidden and the matching branch
    gradients) in bf16 instead of rounding them at load time must not change the results beyond the bf16
    rounding of the few fp32 uses of those tensors (softmax delta = sum dO*O, bias column sums)."""
    c = CASES[case]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=5)
    load_oracle_state(eng, tr)
    eng.set_precision("bf16")
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    out = {}
    names = ["D.L0.qkv", "D.L0.ctx", "D.L0.x1", "D.L0.h", "D.L1.x2", "D.c"]
    for on in (False, True):
        eng.set_bstore(on)
        eng.set_seed(5)
        eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
        out[on] = dict(losses=eng.losses.clone(), g=eng.flat[L.ROLE_CRITIC]["g"].clone(), **{n: eng.debug_buffer(n) for n in names})
    ck = Checker(f"bf16 operand storage on vs off {case}", 6e-3, metric="max")
    a, b = out[True], out[False]
    S, F = P + 1, 2 * cfg.embedding_dims
    for n in names:
        # the hidden tile of the forward-only replica (D(fake), no backward) is never stored by the fused feed-forward kernel
        rows = 2 * B * S * F if n.endswith(".h") else None
        ck.check(n, a[n].flatten()[:rows], b[n].flatten()[:rows])
    ck.check("critic losses", a["losses"][:3], b["losses"][:3], tol=1e-2)
    cs = _cos(a["g"], b["g"])
    from gpu_util import diag
    diag(f"   flat critic gradient cosine bf16-storage on vs off: {cs:.6f}")
    assert cs > 0.997, cs           # E = 256: bf16 storage includes the LayerNorm outputs (residual rounding -> gate flips); measured 0.9982
    ck.done()


@pytest.mark.parametrize("case", ["hot_tiles_E256", "mid_T5_ragged"])
def test_weight_gradient_kernel_matches_split_k_gemm(case):
    """wgrad.hip (row-major LDS chunks, both MFMA operands via hardware transpose reads, panel accumulators)
    against the generic split-K tile GEMM: same bf16 operand values, only the fp32 summation order differs."""
    c = CASES[case]
    cfg, B, P, T = c["cfg"], max(c["B"], 70), c["P"], c["T"]          # >= 4096 token rows so the kernel engages
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=5)
    load_oracle_state(eng, tr)
    eng.set_precision("bf16")
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g).cuda()
    alpha = torch.rand(B, generator=g).cuda()
    out = {}
    for on in (False, True):
        eng.set_wgrad(on)
        eng.set_seed(5)
        eng.critic_backward(x, z, alpha, patches, patch_pad, text, text_pad)
        out[on] = {k: v.clone() for k, v in eng.state(L.ROLE_CRITIC, "g").items()}
    ck = Checker(f"wgrad kernel vs split-K GEMM {case}", 2e-3, metric="max")
    for k in out[True]:
        # encoder weights; the patch encoder, whose X operand is FiLM-modulated on the fly by the kernel; and the FiLM
        # generator, whose gradient comes from the kernel's per-sample contraction mode (no d(modulated input) tensor)
        if k.endswith("weight") and ("transformer" in k or "patches_encoder" in k):
            ck.check(k, out[True][k], out[False][k])
        elif k.endswith("bias") and ("in_proj" in k or "linear1" in k) and "transformer" in k:
            ck.check(k, out[True][k], out[False][k])        # column sums of dY folded into the kernel
        elif "film_generator" in k:
            # different association: (demb^T patches) . W with bf16 patches, against (demb W) * fp32 patches summed over
            # tokens - one more operand rounded to bf16, so the bf16 tolerance applies, not the summation-order one
            ck.check(k, out[True][k], out[False][k], tol=1e-2)
    ck.done()


PREC = "f32"


def test_generator_prefetch_equals_sequential_passes():
    """gg_train_step computes the frozen generator's outputs of all critic iterations ahead, as stacked replicas; with
    dropout 0 (identical arithmetic per row) a step must match the sequential order - through gg_train_step with the
    switch off, and through the host loop (generator_prefetch + critic_backward/apply) of the data-parallel path."""
    c = CASES["hot_tiles_E256"]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(7)
    n = 4
    z_all = torch.randn(n + 1, B, cfg.latent_dims, generator=g).cuda()
    alpha_all = torch.rand(n, B, generator=g).cuda()
    res = {}
    for mode in ("step_prefetch", "step_sequential", "step_sequential_again", "host_loop"):
        eng = engine_from_cfg(cfg, B, P, T, dropout=0.0)
        load_oracle_state(eng, tr)
        eng.set_precision(PREC)
        if mode == "host_loop":
            eng.generator_prefetch(z_all[:n].contiguous(), patches, patch_pad, text, text_pad)
            for k in range(n):
                eng.critic_backward(x, z_all[k].contiguous(), alpha_all[k].contiguous(), patches, patch_pad, text, text_pad)
                eng.critic_apply(1.0)
            eng.generator_backward(z_all[n].contiguous(), patches, patch_pad, text, text_pad)
            eng.generator_apply(1.0)
        else:
            eng.set_prefetch(mode == "step_prefetch")
            eng.train_step(x, patches, patch_pad, text, text_pad, z_all, alpha_all)
        res[mode] = dict(losses=eng.losses.clone(), wd=eng.flat[L.ROLE_CRITIC]["w"].clone(), wg=eng.flat[L.ROLE_GENERATOR]["w"].clone())
    ck = Checker(f"generator prefetch vs sequential ({PREC}, dropout 0)", 5e-3, metric="max")
    ck.check("run-to-run: losses", res["step_sequential_again"]["losses"], res["step_sequential"]["losses"])
    ck.check("run-to-run: critic parameters", res["step_sequential_again"]["wd"], res["step_sequential"]["wd"])
    for other in ("step_sequential", "host_loop"):
        ck.check(f"losses vs {other}", res["step_prefetch"]["losses"], res[other]["losses"])
        ck.check(f"critic parameters vs {other}", res["step_prefetch"]["wd"], res[other]["wd"])
        ck.check(f"generator parameters vs {other}", res["step_prefetch"]["wg"], res[other]["wg"])
    ck.done()


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_batched_generator_pass_of_five_replicas(prec):
    """With dropout on, the n_critic generator passes of a step run as dropout replicas of ONE batched pass in the prefetch arena
    (R = 5; the dropout-free case above shares one conditioning pass and never stacks replicas).  (1) A drop probability of 1e-7 keeps
    every element (threshold round(p * 65536) = 0): the five stacked conditioning vectors and outputs must then equal five one-replica
    generator forwards with the same z - the R = 5 arithmetic against R = 1.  (2) At p = 0.1 the replicas draw independent masks:
    their conditioning vectors differ pairwise, everything stays finite."""
    c = CASES["hot_tiles_E256"]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    x, text, text_pad, patches, patch_pad = dev(*synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True))
    n = 5
    z_all = torch.randn(n, B, cfg.latent_dims, generator=torch.Generator().manual_seed(7)).cuda()
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=5)
    load_oracle_state(eng, tr)
    eng.set_precision(prec)
    eng.set_dropout(1e-7)
    eng.generator_prefetch(z_all.contiguous(), patches, patch_pad, text, text_pad)
    torch.cuda.synchronize()
    cP = eng.debug_buffer("P.c").view(n, B, -1).clone()
    xP = eng.debug_buffer("Xpre").view(n, B, -1).clone()
    tol = 1e-5                     # measured: bit-identical in both modes (the kernels' arithmetic per row does not depend on the replica count)
    ck = Checker(f"batched generator pass, 5 replicas, all-keep dropout ({prec})", tol, metric="max")
    for k in range(n):
        xk = eng.forward(L.ROLE_GENERATOR, z_all[k].contiguous(), patches, patch_pad, text, text_pad, train=True)
        ck.check(f"pass {k}: generated genes", xP[k], xk)
        ck.check(f"pass {k}: conditioning vector", cP[k], eng.debug_buffer("G.c").view(B, -1))
    ck.done()
    eng.set_dropout(0.1)
    eng.set_seed(5)
    eng.generator_prefetch(z_all.contiguous(), patches, patch_pad, text, text_pad)
    torch.cuda.synchronize()
    cP = eng.debug_buffer("P.c").view(n, B, -1)
    assert torch.isfinite(cP).all() and torch.isfinite(eng.debug_buffer("Xpre")).all()
    for a in range(n):
        for b in range(a + 1, n):
            assert (cP[a] - cP[b]).abs().max() > 1e-4, (a, b)              # independent draws per replica


@pytest.mark.parametrize("case", ["hot_tiles_E256", "cls_tail_S257"])
def test_side_streams_do_not_change_results(case):
    """The engine's side streams (parameter-gradient leaves beside the data-gradient chain, the generator iteration's two
    forward passes, the pipelined prefetch) only reorder independent work: a full train() must give the same parameters
    as with everything serialised on the caller's stream (same seeds => same dropout masks; fp32 atomics reorder)."""
    c = CASES[case]
    cfg, B, P, T = c["cfg"], max(c["B"], 70), c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    batch = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    x, text, text_pad, patches, patch_pad = dev(*batch)
    g = torch.Generator().manual_seed(7)
    n = 3
    z_all = torch.randn(n + 1, B, cfg.latent_dims, generator=g).cuda()
    alpha_all = torch.rand(n, B, generator=g).cuda()
    res = {}
    for side in (True, False):
        eng = engine_from_cfg(cfg, B, P, T, dropout=0.1, seed=5)
        load_oracle_state(eng, tr)
        eng.set_precision("bf16")
        eng.set_side_streams(side)
        eng.critic_backward(x, z_all[0].contiguous(), alpha_all[0].contiguous(), patches, patch_pad, text, text_pad)
        gd = eng.flat[L.ROLE_CRITIC]["g"].clone()
        eng.generator_backward(z_all[n].contiguous(), patches, patch_pad, text, text_pad)
        gg = eng.flat[L.ROLE_GENERATOR]["g"].clone()
        res[side] = dict(gd=gd, gg=gg, losses=eng.losses.clone())
    ck = Checker(f"side streams on vs off {case}", 1e-4, metric="max")
    ck.check("critic gradient", res[True]["gd"], res[False]["gd"])
    ck.check("generator gradient", res[True]["gg"], res[False]["gg"])
    ck.check("losses", res[True]["losses"], res[False]["losses"])
    ck.done()


@pytest.mark.parametrize("case", ["mid_T5_ragged", "leaky_T300", "single_sample", "long_S601", "text_T77_E256", "film_P1",
                                  "film_P33_E256", "img_P40_E256", "long_S545_E256", "img_long_S577_E256"])
def test_bf16_mode_on_generic_shapes(case):
    """bf16 mode on shapes that do not qualify for the fused kernels (E = 32: head dim 8, no token-on-lane / flash path;
    one-sample batches): the generic bf16 GEMM route must still follow the fp32 oracle's gradient direction."""
    cfg, tr, eng, (x, text, text_pad, patches, patch_pad) = setup(case)
    eng.set_precision("bf16")
    B = x.shape[0]
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, cfg.latent_dims, generator=g)
    alpha = torch.rand(B, 1, generator=g)
    cond = (patches, patch_pad, text, text_pad)
    r = tr.critic_iteration(x, z, alpha, cond, apply=False)
    xg, tg, tpg, pg, ppg, zg, ag = dev(x, text, text_pad, patches, patch_pad, z, alpha)
    eng.critic_backward(xg, zg, ag, pg, ppg, tg, tpg)
    ck = Checker(f"bf16 mode vs fp32 oracle ({case})", 6e-2, metric="max")
    l = eng.losses.tolist()
    ck.check("d_real,d_fake", np.array(l[:2]), np.array([r["d_real"].item(), r["d_fake"].item()]))
    grads = eng.state(L.ROLE_CRITIC, "g")
    fa, fb = [], []
    for n, ref in r["grads"].items():
        if ref is None or n.endswith("in_proj_bias") or ref.abs().max() < 1e-12:
            continue
        fa.append(grads[n].reshape(-1).cpu())
        fb.append(ref.reshape(-1))
    total = _cos(torch.cat(fa), torch.cat(fb))
    from gpu_util import diag
    diag(f"   flat critic gradient cosine: {total:.5f}")
    assert total >= 0.95, total
    ck.done()


def test_call_time_errors_and_nan_semantics():
    """Boundary behaviour (SURVEY 8b 'Errors' / 'Tensor conventions'): shape mistakes raise Python exceptions, exceeding the
    reserved capacity is an error code (never an abort), and a fully padded text row yields NaN exactly as torch does."""
    c = CASES["mid_T5_ragged"]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    set_dropout(tr.gen, 0.0)
    set_dropout(tr.disc, 0.0)
    x, text, text_pad, patches, patch_pad = synthetic_batch(cfg, B, P, T, seed=12, pad_patches=True, pad_text=True)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.0)
    load_oracle_state(eng, tr)
    xg, tg, tpg, pg, ppg = dev(x, text, text_pad, patches, patch_pad)
    with pytest.raises(ValueError):
        eng.forward(L.ROLE_CRITIC, xg, pg[:, :, :-1].contiguous(), ppg, tg, tpg)                  # wrong patch width
    with pytest.raises(ValueError):
        eng.forward(L.ROLE_CRITIC, xg, pg, ppg[:, :-1].contiguous(), tg, tpg)                     # mask / patches disagree
    big = torch.cat([pg, pg], dim=0)
    with pytest.raises(RuntimeError):
        eng.forward(L.ROLE_CRITIC, torch.cat([xg, xg]), big, torch.cat([ppg, ppg]), torch.cat([tg, tg]), torch.cat([tpg, tpg]))
    ok = eng.forward(L.ROLE_CRITIC, xg, pg, ppg, tg, tpg)                                          # the engine is still usable
    assert torch.isfinite(ok).all()
    tp2 = text_pad.clone()
    tp2[2, :] = True                                                                                # every text key of sample 2 padded
    tr.disc.eval()
    with torch.no_grad():
        ref = tr.disc(x, patches, patch_pad, text, tp2)
    got = eng.forward(L.ROLE_CRITIC, xg, pg, ppg, tg, tp2.cuda())
    assert torch.isnan(ref[2]).all() and torch.isnan(got[2]).all()
    keep = [i for i in range(B) if i != 2]
    assert torch.allclose(got[keep].cpu(), ref[keep], rtol=1e-3, atol=1e-5)
