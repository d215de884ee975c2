"""Captured train step (gg_set_graph, include/gemmgan.h): gg_train_step replayed from a hipGraph.

(1) With dropout 0 a sequence of steps through the graph route (eager, capture, replay, replay ...) must give the
    parameters and losses of the eager route - same kernels, same order per stream; fp32 atomics reorder as they do between
    two eager runs.  Adam is in the matrix because its step number is the one optimiser quantity a frozen kernel argument
    cannot carry (device-side offset word), RMSprop because it is the reference's default.
(2) With dropout > 0 and frozen weights (lr = 0) the replays of one graph on identical inputs must draw FRESH masks (the
    dropout epoch word): losses differ from replay to replay, by no more than mask noise.
(3) A caller whose input buffers move does not capture (and still computes the right thing); a second resident batch gets a
    second graph.
(4) The variants with their own launch sequences (FiLM-only, image transformer, unconditional) capture and replay."""
import pytest
import torch

from gemm_gan_amd import _lib as L
from gpu_util import Checker, dev, diag, engine_from_cfg, load_oracle_state
from oracle.torch_oracle import Trainer, synthetic_batch
from test_engine_oracle_gpu import CASES

pytestmark = pytest.mark.gpu


def _inputs(cfg, B, P, T, n, seed=12):
    x, text, text_pad, patches, patch_pad = dev(*synthetic_batch(cfg, B, P, T, seed=seed, pad_patches=True, pad_text=True))
    g = torch.Generator().manual_seed(seed + 100)
    zs = [torch.randn(n + 1, B, cfg.latent_dims, generator=g).cuda() for _ in range(8)]
    als = [torch.rand(n, B, generator=g).cuda() for _ in range(8)]
    return (x, patches, patch_pad, text, text_pad), zs, als


@pytest.mark.parametrize("case,precision,optimizer", [
    ("hot_tiles_E256", "f32", "rms_prop"), ("hot_tiles_E256", "bf16", "adam"), ("cls_tail_S257", "bf16", "rms_prop"),
    ("mid_T5_ragged", "f32", "adamw"), ("film_P33_E256", "bf16", "adam"), ("img_P40_E256", "bf16", "rms_prop"),
])
def test_graph_route_equals_eager_route(case, precision, optimizer):
    """Lock-step: before every step the graph engine takes the eager engine's parameters and optimiser state (same
    addresses, so the captured graph stays valid), then both run the step on the same inputs.  That keeps the comparison a
    ONE-step one - RMSprop / Adam turn the rounding noise of a near-zero gradient into a full +-lr step, so free-running
    trajectories of two bit-different but equally right runs drift apart (tests/test_numpy_oracle.py measures it).  In bf16
    mode up to ~10 % of the generator's entries (gene-output weights whose gradient is rounding noise) take such a step in
    either direction under Adam, in two eager runs as much as between the routes: the gate allows 15 %."""
    c = CASES[case]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    # n_critic = 1: the critic losses of a step are then computed from the weights both engines were GIVEN (tight gate);
    # the generator loss follows one critic update and is compared loosely in bf16, the parameters by count
    n, steps = 1, 6
    (x, patches, patch_pad, text, text_pad), zs, als = _inputs(cfg, B, P, T, n)
    eager = engine_from_cfg(cfg, B, P, T, dropout=0.0, optimizer=optimizer)
    graph = engine_from_cfg(cfg, B, P, T, dropout=0.0, optimizer=optimizer)
    for e in (eager, graph):
        load_oracle_state(e, tr)
        e.set_precision(precision)
    graph.set_graph(True)
    f32 = precision == "f32"
    ck = Checker(f"graph route vs eager route, lock-step: {case} {precision} {optimizer}", 1e-4 if f32 else 2e-2, metric="max")
    lr = {L.ROLE_CRITIC: cfg.lr_d, L.ROLE_GENERATOR: cfg.lr_g}
    for s in range(steps):
        for r in (L.ROLE_GENERATOR, L.ROLE_CRITIC):
            for k in ("w", "s1", "s2"):
                graph.flat[r][k].copy_(eager.flat[r][k])
        init = {r: eager.flat[r]["w"].clone() for r in (L.ROLE_GENERATOR, L.ROLE_CRITIC)}
        for e in (eager, graph):
            e.train_step(x, patches, patch_pad, text, text_pad, zs[s], als[s])
        ck.check(f"step {s}: critic losses", graph.losses[:3], eager.losses[:3])
        ck.check(f"step {s}: generator loss", graph.losses[3:4], eager.losses[3:4], 1e-3 if f32 else 0.3)
        for r, name, k in ((L.ROLE_CRITIC, "critic", n), (L.ROLE_GENERATOR, "generator", 1)):
            ck.check_post(f"step {s}: {name} parameters", graph.flat[r]["w"], eager.flat[r]["w"].cpu().numpy(), init[r].cpu().numpy(),
                          optimizer, lr[r], k if optimizer == "rms_prop" else s * k + k, rtol=1e-3 if f32 else 5e-2, share=0.03 if f32 else 0.15)
    st = graph.graph_stats()
    diag(f"   {st}, {graph.launch_count()} launches per step")
    assert st == {"captures": 1, "replays": steps - 2, "failures": 0}, st
    assert (graph.optimizer_step(L.ROLE_GENERATOR), graph.optimizer_step(L.ROLE_CRITIC)) == (steps, steps * n)
    ck.done()


def test_replays_draw_fresh_dropout_masks():
    c = CASES["hot_tiles_E256"]
    cfg, B, P, T = c["cfg"], max(c["B"], 48), c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    n = 2
    (x, patches, patch_pad, text, text_pad), zs, als = _inputs(cfg, B, P, T, n)
    out = {}
    for p in (0.0, 0.1):
        eng = engine_from_cfg(cfg, B, P, T, dropout=p if p else 0.0, seed=5)
        if p == 0.0 and eng.dropout != 0.0:
            eng.set_dropout(0.0)
        load_oracle_state(eng, tr)
        eng.set_precision("bf16")
        eng.set_lr(L.ROLE_CRITIC, 0.0)
        eng.set_lr(L.ROLE_GENERATOR, 0.0)
        eng.set_graph(True)
        w0 = eng.flat[L.ROLE_CRITIC]["w"].clone()
        ls = []
        for s in range(6):
            eng.train_step(x, patches, patch_pad, text, text_pad, zs[0], als[0])       # identical inputs every step
            ls.append(eng.losses.clone().cpu())
        assert torch.equal(w0, eng.flat[L.ROLE_CRITIC]["w"]), "lr = 0 must freeze the weights"
        st = eng.graph_stats()
        assert st["captures"] == 1 and st["replays"] == 4 and st["failures"] == 0, st
        out[p] = torch.stack(ls)[:, :2]          # D_real, D_fake of the last critic iteration
    diag("== replays on identical inputs, frozen weights: D_real / D_fake per step")
    for p, v in out.items():
        diag(f"   dropout {p}: " + "  ".join(f"({a:+.5f}, {b:+.5f})" for a, b in v.tolist()))
    d0 = out[0.0]
    assert float((d0 - d0[0]).abs().max()) <= 1e-3 * float(d0.abs().max()) + 1e-6, "without dropout the replays repeat"
    d1 = out[0.1]
    for i in range(1, 6):
        for j in range(i):
            assert float((d1[i] - d1[j]).abs().max()) > 0.0, f"steps {i} and {j} drew the same masks"
    # ... and mask noise only: every draw stays near the dropout-free value
    scale = float(d0.abs().max()) + 1e-3
    assert float((d1 - d0[0]).abs().max()) <= 0.5 * scale + 0.05, (d1, d0[0])


def test_moving_buffers_stay_eager_and_a_second_batch_gets_its_own_graph():
    c = CASES["mid_T5_ragged"]
    cfg, B, P, T = c["cfg"], c["B"], c["P"], c["T"]
    torch.manual_seed(11)
    tr = Trainer(cfg)
    n = 2
    (x, patches, patch_pad, text, text_pad), zs, als = _inputs(cfg, B, P, T, n)
    (x2, patches2, patch_pad2, text2, text_pad2), _, _ = _inputs(cfg, B, P, T, n, seed=31)
    ref = engine_from_cfg(cfg, B, P, T, dropout=0.0)
    eng = engine_from_cfg(cfg, B, P, T, dropout=0.0)
    for e in (ref, eng):
        load_oracle_state(e, tr)
        e.set_precision("f32")
    eng.set_graph(True)
    seq = [0, 1, 0, 1, 0, 1, 0]
    batches = [(x, patches, patch_pad, text, text_pad), (x2, patches2, patch_pad2, text2, text_pad2)]
    for s, b in enumerate(seq):
        for e in (ref, eng):
            e.train_step(*batches[b], zs[s], als[s])
    st = eng.graph_stats()
    diag(f"== two resident batches alternating: {st}")
    assert st == {"captures": 2, "replays": 3, "failures": 0}, st
    ck = Checker("graph route, two resident batches vs eager", 5e-3, metric="max")
    ck.check("critic parameters", eng.flat[L.ROLE_CRITIC]["w"], ref.flat[L.ROLE_CRITIC]["w"])
    ck.check("generator parameters", eng.flat[L.ROLE_GENERATOR]["w"], ref.flat[L.ROLE_GENERATOR]["w"])
    # buffers that move every step: clones at fresh addresses, the old ones kept alive so the allocator cannot hand them back
    hold = []
    for s in range(4):
        moved = tuple(t.clone() for t in batches[0])
        hold.append(moved)
        for e, bt in ((ref, batches[0]), (eng, moved)):
            e.train_step(*bt, zs[s], als[s])
    st2 = eng.graph_stats()
    assert st2["captures"] == 2 and st2["replays"] == 3, st2
    ck.check("critic parameters after 4 steps on moving buffers", eng.flat[L.ROLE_CRITIC]["w"], ref.flat[L.ROLE_CRITIC]["w"])
    ck.done()


def test_facade_train_replays_when_asked_to():
    """The drop-in facade: WGAN_GP_nocond.train(x) (BASELINE configs[0] family) draws its noise into fresh tensors, which
    Engine.train_step stages into engine-lifetime buffers, so a resident batch replays from its third step on."""
    from gemm_gan_amd.vanilla import WGAN_GP_nocond
    torch.manual_seed(3)
    m = WGAN_GP_nocond(100, 16, [], [32, 32, 100], [32, 32, 1], n_critic=3, device="cuda:0", precision="bf16")
    m.use_graph = True
    m.build_WGAN_GP_nocond()
    x = torch.randn(24, 100, device="cuda:0")
    for _ in range(5):
        m.train(x)
    st = m.engine.graph_stats()
    assert st == {"captures": 1, "replays": 3, "failures": 0}, st
    assert all(torch.isfinite(m.engine.flat[r]["w"]).all() for r in (L.ROLE_GENERATOR, L.ROLE_CRITIC))
