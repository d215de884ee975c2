"""Data-parallel orchestration on CPU: 2 ranks over `gloo`, each with half of the minibatch, must end
one train() with the same parameters as 1 rank on the full minibatch.

What is under test is gemm_gan_amd.model.WGAN_GP's DP plumbing (shard -> *_backward -> SUM all-reduce of
the flat gradient buffer -> *_apply(1/N) -> clip on the averaged gradient).  The GPU engine is replaced
by a duck-typed stand-in that computes with oracle #1 on CPU (tests may use the oracle as the checker;
the product path never does)."""
import os
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gemm_gan_amd as gga
from gemm_gan_amd import _lib as L
from oracle.torch_oracle import PathConfig, Trainer, live_parameters, set_dropout, synthetic_batch

CFG = PathConfig(n_genes=60, latent_dims=12, embedding_dims=16, hidden_dims=24, text_dims=10, patch_dims=14,
                 dropout=0.0, n_critic=2)
B, P, T = 8, 5, 2


class OracleEngine:
    """Same call surface as gemm_gan_amd.engine.Engine, arithmetic by oracle #1 (CPU)."""

    def __init__(self, trainer: Trainer):
        self.tr = trainer
        self.cfg = types.SimpleNamespace(max_batch=1 << 20, max_patches=1 << 20, max_text_tokens=1 << 20)
        self.nets = {L.ROLE_GENERATOR: trainer.gen, L.ROLE_CRITIC: trainer.disc}
        self.flat = {r: {"g": torch.zeros(sum(p.numel() for _, p in live_parameters(n)))} for r, n in self.nets.items()}
        self.losses = torch.zeros(L.N_LOSSES)
        self.lr = {}
        # the MLP-head parameters (<mlp>.0.0, <mlp>.1.0, final_layer) are the last six entries of the flat layout
        self.mlp_range = {}
        for r, n in self.nets.items():
            sizes = [p.numel() for _, p in live_parameters(n)]
            self.mlp_range[r] = (sum(sizes[:-6]), sum(sizes[-6:]))
        # stages of the conditioning backward and the contiguous gradient range each completes (include/gemmgan.h gg_cond_stage_range):
        # 0 cross-attention, 1 .. nl encoder layers last to first, nl + 1 CLS / FiLM / text encoder / patch encoder
        nl = trainer.cfg.n_layers
        self.cond_stages = nl + 2
        self.stage_range = {}
        for r, n in self.nets.items():
            names = [(nm, p.numel()) for nm, p in live_parameters(n)]
            offs = np.cumsum([0] + [k for _, k in names])

            def rng(pred):
                idx = [i for i, (nm, _) in enumerate(names) if pred(nm)]
                assert idx == list(range(idx[0], idx[-1] + 1)), "a stage's parameters must be contiguous in the flat buffer"
                return int(offs[idx[0]]), int(offs[idx[-1] + 1] - offs[idx[0]])
            st = [rng(lambda nm: nm.startswith(("patch2text_attention.", "text2patch_attention.")))]
            st += [rng(lambda nm, l=l: nm.startswith(f"patches_transformer.layers.{l}.")) for l in range(nl - 1, -1, -1)]
            st.append(rng(lambda nm: nm.startswith(("patches_cls_token", "film_generator.", "text_encoder.", "patches_encoder."))))
            self.stage_range[r] = st
            covered = sorted(st + [self.mlp_range[r]])
            assert covered[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(covered, covered[1:])) and \
                covered[-1][0] + covered[-1][1] == offs[-1], "stage ranges + MLP range must tile the flat gradient buffer"
        self.calls = []

    def set_lr(self, role, lr):
        self.lr[role] = lr

    def _store(self, role, grads):
        flat = torch.cat([grads[n].reshape(-1) for n, _ in live_parameters(self.nets[role])])
        self.flat[role]["g"].copy_(flat)

    def _apply(self, role, scale, clip, opt):
        off = 0
        for n, p in live_parameters(self.nets[role]):
            p.grad = (self.flat[role]["g"][off:off + p.numel()] * scale).view_as(p).clone()
            off += p.numel()
        torch.nn.utils.clip_grad_norm_([p for _, p in live_parameters(self.nets[role])], clip)
        for g in opt.param_groups:
            g["lr"] = self.lr[role]
        opt.step()

    def critic_backward(self, x, z, alpha, pat, ppad, text, tpad):
        r = self.tr.critic_iteration(x, z, alpha.view(-1, 1), (pat, ppad.bool(), text, tpad.bool()), apply=False)
        # undo the oracle's in-place clipping: store the raw gradients
        self._store(L.ROLE_CRITIC, r["grads"])
        self.losses[0], self.losses[1], self.losses[2] = r["d_real"].item(), r["d_fake"].item(), r["gp"].item()

    # two-phase form used by the data-parallel host loop: the stand-in has the whole gradient after "head"
    def critic_backward_head(self, x, z, alpha, pat, ppad, text, tpad):
        self.calls.append(("critic_head", z.clone()))
        self.critic_backward(x, z, alpha, pat, ppad, text, tpad)

    def critic_backward_cond(self, pat, ppad, text, tpad):
        self.calls.append(("critic_cond",))

    def critic_backward_cond_stage(self, stage, pat, ppad, text, tpad):
        self.calls.append((f"critic_cond{stage}",))

    def generator_backward_cond_stage(self, stage, pat, ppad, text, tpad):
        self.calls.append((f"gen_cond{stage}",))

    def critic_cond_prefetch(self, pat, ppad, text, tpad):
        self.calls.append(("critic_cond_prefetch", pat.clone()))

    def generator_backward_head(self, z, pat, ppad, text, tpad):
        self.calls.append(("gen_head", z.clone()))
        self.generator_backward(z, pat, ppad, text, tpad)

    def generator_backward_cond(self, pat, ppad, text, tpad):
        self.calls.append(("gen_cond",))

    def critic_apply(self, scale):
        self._apply(L.ROLE_CRITIC, scale, self.tr.cfg.clip_d, self.tr.opt_d)

    def reset_launch_count(self):
        self.resets = getattr(self, "resets", 0) + 1

    def generator_prefetch(self, z_all, pat, ppad, text, tpad):
        pass        # a scheduling hint of the HIP engine (batched generator passes): same results without it

    def generator_backward(self, z, pat, ppad, text, tpad):
        r = self.tr.generator_iteration(z, (pat, ppad.bool(), text, tpad.bool()), apply=False)
        self._store(L.ROLE_GENERATOR, r["grads"])
        self.losses[3] = r["g_loss"].item()

    def generator_apply(self, scale):
        self._apply(L.ROLE_GENERATOR, scale, self.tr.cfg.clip_g, self.tr.opt_g)

    def train_step(self, x, pat, ppad, text, tpad, z_all, alpha_all):
        n = alpha_all.shape[0]
        for k in range(n):
            self.critic_backward(x, z_all[k], alpha_all[k], pat, ppad, text, tpad)
            self.critic_apply(1.0)
        self.generator_backward(z_all[n], pat, ppad, text, tpad)
        self.generator_apply(1.0)


def make_wgan():
    torch.manual_seed(5)
    tr = Trainer(CFG)
    set_dropout(tr.gen, 0.0)
    set_dropout(tr.disc, 0.0)
    w = gga.WGAN_GP(CFG.n_genes, CFG.latent_dims, CFG.embedding_dims, [24, 24, CFG.n_genes], [24, 24, 1],
                    text_embedding_dims=CFG.text_dims, patches_embedding_dims=CFG.patch_dims, n_critic=CFG.n_critic,
                    dropout=0.0, device="cpu")
    w.engine = OracleEngine(tr)          # stand-in for build_WGAN_GP() (which needs the GPU)
    w.init_train()
    return w, tr


def batch_and_noise():
    x, text, text_pad, patches, patch_pad = synthetic_batch(CFG, B, P, T, seed=8, pad_patches=True, pad_text=True)
    g = torch.Generator().manual_seed(9)
    z_all = torch.randn(CFG.n_critic + 1, B, CFG.latent_dims, generator=g)
    alpha_all = torch.rand(CFG.n_critic, B, generator=g)
    return (x, text, text_pad, patches, patch_pad), z_all, alpha_all


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, tr = make_wgan()
        (x, text, text_pad, patches, patch_pad), z_all, alpha_all = batch_and_noise()
        n = B // world
        s = slice(rank * n, (rank + 1) * n)
        # every gradient all-reduce is logged between the engine calls: which range of which flat buffer, and when
        flat_ptr = {w.engine.flat[r]["g"].untyped_storage().data_ptr(): r for r in (L.ROLE_GENERATOR, L.ROLE_CRITIC)}
        real_all_reduce = dist.all_reduce

        def logged_all_reduce(t, *a, **k):
            r = flat_ptr.get(t.untyped_storage().data_ptr())
            if r is not None:
                w.engine.calls.append(("all_reduce", r, t.storage_offset(), t.numel()))
            return real_all_reduce(t, *a, **k)
        dist.all_reduce = logged_all_reduce
        w.train_with_noise(x[s], text[s], text_pad[s], patches[s], patch_pad[s], z_all[:, s].contiguous(),
                           alpha_all[:, s].contiguous(), next_cond=(patches[s], patch_pad[s], text[s], text_pad[s]))
        dist.all_reduce = real_all_reduce
        n_st = CFG.n_layers + 2

        def iteration(role, head, cond):        # head phase, its MLP bucket, then every backward stage followed at once by ITS range
            exp = [(head,), ("all_reduce", role) + tuple(w.engine.mlp_range[role])]
            for i in range(n_st):
                exp += [(cond + str(i),), ("all_reduce", role) + tuple(w.engine.stage_range[role][i])]
            return exp
        seen = [c if c[0] == "all_reduce" else (c[0],) for c in w.engine.calls]
        expected = iteration(L.ROLE_CRITIC, "critic_head", "critic_cond") * CFG.n_critic + \
            iteration(L.ROLE_GENERATOR, "gen_head", "gen_cond") + [("critic_cond_prefetch",)]
        assert seen == expected, (seen, expected)
        # reverse-layer order: the stage ranges walk the flat buffer from the MLP block down to offset 0, without gaps
        for r in (L.ROLE_GENERATOR, L.ROLE_CRITIC):
            end = w.engine.mlp_range[r][0]
            for off, numel in w.engine.stage_range[r]:
                assert off + numel == end, (r, off, numel, end)
                end = off
            assert end == 0
        sd = {k: v.detach().numpy().copy() for k, v in {**{"g." + k: v for k, v in tr.gen.state_dict().items()},
                                                         **{"d." + k: v for k, v in tr.disc.state_dict().items()}}.items()}
        d_loss_global = float(w.d_batch_loss[0])
        # train() draws its own noise: every rank must use a different z / alpha stream (SURVEY 8e partitioning)
        torch.manual_seed(123)
        w.engine.calls.clear()
        w.train(x[s], text[s], text_pad[s], patches[s], patch_pad[s])
        z0 = [c[1] for c in w.engine.calls if c[0] == "critic_head"][0]
        q.put((rank, sd, d_loss_global, z0.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_rank_full_batch():
    torch.set_num_threads(1)
    w, tr = make_wgan()
    (x, text, text_pad, patches, patch_pad), z_all, alpha_all = batch_and_noise()
    w.train_with_noise(x, text, text_pad, patches, patch_pad, z_all, alpha_all)
    ref = {**{"g." + k: v for k, v in tr.gen.state_dict().items()}, **{"d." + k: v for k, v in tr.disc.state_dict().items()}}

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sds = {r: sd for r, sd, _, _ in got}
    zs = {r: z for r, _, _, z in got}
    assert zs[0].shape == zs[1].shape and not np.allclose(zs[0], zs[1]), "ranks drew identical latent vectors"
    # reported losses are global means (identical on both ranks)
    dl = {r: d for r, _, d, _ in got}
    assert abs(dl[0] - dl[1]) <= 1e-6 * max(1.0, abs(dl[0])), dl
    E = CFG.embedding_dims
    for k, v in ref.items():
        if "patches_transformer_layer." in k:
            continue
        a, b = sds[0][k], sds[1][k]
        assert np.array_equal(a, b), f"ranks diverged on {k}"       # replicas stay bit-identical
        keep = np.ones(v.numel(), dtype=bool)
        if k.endswith("in_proj_bias"):
            keep[E:2 * E] = False                                    # zero-gradient slice (see golden_util.comparable)
        d = np.abs(a.reshape(-1) - v.reshape(-1).numpy())[keep].max()
        assert d <= 2e-4 * max(float(v.abs().max()), 1e-3) + 1e-6, (k, d)
