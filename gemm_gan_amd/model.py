"""Drop-in host-side mirror of the reference's hot-path object API, backed by the HIP engine.

Mirrors /root/reference/src/conditional_gan_cross_attention_with_film.py (R:):
    generator (R:97), discriminator (R:167), WGAN_GP_model (R:236), WGAN_GP (R:256) with
    init_train (R:320), build_WGAN_GP (R:334), train_disc (R:376), train_gen (R:425), train (R:463),
    generate_samples (R:601), fit (R:619; training loop, LR schedule and checkpoints only).

Same class names, constructor kwargs, method signatures (including the differing argument orders of
train_disc / gradient_penalty / forward), tensor conventions (bool masks, True = padded) and
``state_dict`` keys - including the twelve dead ``patches_transformer_layer.*`` entries (R:114) -
so a reference checkpoint loads here and vice versa.

What runs where: ``torch.nn`` modules below are PARAMETER CONTAINERS ONLY (they give the reference's
initialisation, names and ``state_dict`` for free); their storage is re-pointed into the engine's flat
HBM buffers and their ``forward`` is never called.  Every forward / backward / optimiser computation is
done by libgemmgan.so.  Without a GPU or without the built library these classes raise.

Not supported (by design): autograd through ``forward`` (training goes through ``WGAN_GP.train*``,
whose backward is hand-written in the engine); ``is_bn=True`` (never enabled by the reference's
``__main__``, R:949); ``p_aug != 0`` (raises NameError in the reference, R:401).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import _lib as L
from .engine import Engine


def _block(n_in, n_out, slope):
    return nn.Sequential(nn.Linear(n_in, n_out), nn.LeakyReLU(negative_slope=slope))


class _CondNet(nn.Module):
    """Parameter container with the reference's attribute names + engine-backed forward."""

    _role = None          # 'generator' | 'discriminator'
    _variant = "xattn_film"   # engine variant (film.py overrides: "film")

    def __init__(self, first_dims, embedding_dims, mlp_dims, text_embedding_dims=768,
                 patches_embedding_dims=1024, negative_slope=0.0, is_bn=False):
        super().__init__()
        if is_bn:
            raise NotImplementedError("is_bn=True is not part of the accelerated hot path (never enabled upstream)")
        E = embedding_dims
        self.embedding_dims = E
        self.text_embedding_dims = text_embedding_dims
        self.patches_embedding_dims = patches_embedding_dims
        self.negative_slope = negative_slope
        self.is_bn = is_bn
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        # construction order == reference order (R:111-126), so a given torch seed gives the same init
        # src/conditional_gan_film.py:111-124 and src/conditional_gan_img_transformer.py:105-117: no token encoder / cross
        # attention, bias-free encoder layers; the image-transformer file has no FiLM and a Linear-ReLU-LayerNorm patch encoder
        film_only = self._variant in ("film", "img")
        if self._variant != "img":
            self.film_generator = nn.Linear(text_embedding_dims, patches_embedding_dims * 2)
        if not film_only:
            self.text_encoder = nn.Linear(text_embedding_dims, E)
        if self._variant == "img":
            self.patches_encoder = nn.Sequential(nn.Linear(patches_embedding_dims, E), nn.ReLU(), nn.LayerNorm(E))
        else:
            self.patches_encoder = nn.Linear(patches_embedding_dims, E)
        self.patches_transformer_layer = nn.TransformerEncoderLayer(
            d_model=E, nhead=4, dim_feedforward=E * 2, dropout=0.1, activation="relu", batch_first=True,
            bias=not film_only)
        self.patches_cls_token = nn.Parameter(torch.empty(1, 1, E))
        torch.nn.init.trunc_normal_(self.patches_cls_token, std=0.02)
        self.patches_transformer = nn.TransformerEncoder(self.patches_transformer_layer, num_layers=2)
        if not film_only:
            self.patch2text_attention = nn.MultiheadAttention(embed_dim=E, num_heads=4, batch_first=True)
            self.text2patch_attention = nn.MultiheadAttention(embed_dim=E, num_heads=4, batch_first=True)
        self.input_dims = first_dims + E
        dims = list(mlp_dims)
        blocks = nn.ModuleList()
        prev = self.input_dims
        for d in dims[:-1]:
            blocks.append(_block(prev, d, negative_slope))
            prev = d
        if len(blocks) != 2 or dims[0] != dims[1]:
            raise NotImplementedError("the engine implements the reference's [H, H, out] MLP heads")
        setattr(self, self._role, blocks)
        self.final_layer = nn.Linear(dims[-2], dims[-1])
        self._engine: Optional[Engine] = None
        self._engine_role = L.ROLE_GENERATOR if self._role == "generator" else L.ROLE_CRITIC

    # ---- engine binding ---------------------------------------------------------------------
    def _bind(self, engine: Engine):
        """Copy current values into the engine's flat buffer and re-point every live parameter at it."""
        self._engine = engine
        role = self._engine_role
        params = dict(self.named_parameters())
        with torch.no_grad():
            for name in engine.layout[role]:
                view = engine.view(role, name)
                view.copy_(params[name].detach().to(view.device, torch.float32))
                params[name].data = view
            for name, p in params.items():           # dead template layer: plain device tensors
                if name.startswith("patches_transformer_layer."):
                    p.data = p.data.to(engine.device)
        return self

    def _require_engine(self) -> Engine:
        if self._engine is None:
            raise RuntimeError("network is not bound to a HIP engine; build it through WGAN_GP.build_WGAN_GP() "
                               "or gemm_gan_amd.standalone_engine(...) (there is no torch/CPU fallback)")
        return self._engine

    def forward(self, gene_expression, patches, patches_padding_mask, text_tokens, text_padding_mask):
        eng = self._require_engine()
        owner = getattr(eng, "_owner", None)
        if owner is not None:
            owner._ensure_capacity(patches.shape[0], patches.shape[1], text_tokens.shape[1])
            eng = owner.engine
        with torch.no_grad():
            return eng.forward(self._engine_role, gene_expression.to(eng.device), patches.to(eng.device).contiguous(),
                               patches_padding_mask.to(eng.device), text_tokens.to(eng.device).contiguous(),
                               text_padding_mask.to(eng.device), train=self.training)


class generator(_CondNet):
    _role = "generator"

    def __init__(self, latent_dims, embedding_dims, generator_dims, text_embedding_dims=768,
                 patches_embedding_dims=1024, negative_slope=0.0, is_bn=False):
        super().__init__(latent_dims, embedding_dims, generator_dims, text_embedding_dims,
                         patches_embedding_dims, negative_slope, is_bn)
        self.latent_dims = latent_dims
        self.generator_dims = generator_dims


class discriminator(_CondNet):
    _role = "discriminator"

    def __init__(self, vector_dims, embedding_dims, discriminator_dims, text_embedding_dims=768,
                 patches_embedding_dims=1024, negative_slope=0.0, is_bn=False):
        super().__init__(vector_dims, embedding_dims, discriminator_dims, text_embedding_dims,
                         patches_embedding_dims, negative_slope, is_bn)
        self.vector_dims = vector_dims
        self.discriminator_dims = discriminator_dims


def rccl_process_group_options():
    """`pg_options` for torch.distributed.init_process_group("nccl", ...) in a data-parallel host: RCCL's internal stream from torch's
    HIGH-priority pool.  The HIP runtime multiplexes a process's streams onto 4 hardware queues in creation order and then re-uses them
    (DESIGN.md section 6): a default-priority RCCL stream can land on the hardware queue of the compute stream, and its wait for the side
    stream's event - the gradient all-reduces are issued from the engine's side stream, one per backward stage - then stalls every
    compute kernel queued behind it until the side stream has drained.  Measured with one rank (real RCCL, the full host loop):
    26.5 ms per step with the default stream against 24.9 ms with a high-priority one (24.3 without any collective).  High-priority
    streams get hardware queues of their own."""
    import torch.distributed as dist
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
    return opts


def WGAN_GP_model(latent_dims, vector_dims, embedding_dims, generator_dims, discriminator_dims,
                  text_embedding_dims=768, patches_embedding_dims=1024, negative_slope=0.0, is_bn=False):
    gen = generator(latent_dims, embedding_dims, generator_dims, text_embedding_dims,
                    patches_embedding_dims, negative_slope, is_bn)
    disc = discriminator(vector_dims, embedding_dims, discriminator_dims, text_embedding_dims,
                         patches_embedding_dims, negative_slope, is_bn)
    return gen, disc


class _EngineOptimizer:
    """Stand-in for torch.optim.* exposing what fit() touches: mutable param_groups[i]['lr'] (R:651-657)."""

    def __init__(self, kind, lr):
        self.kind = kind
        self.param_groups = [{"lr": lr}]

    def zero_grad(self, set_to_none=True):
        return None          # gradients live in the engine's flat buffer and are rewritten every iteration


class WGAN_GP:
    _variant = "xattn_film"       # engine variant
    _clip = (10.0, 2.0)           # clip_grad_norm_ max_norm of the critic / generator step (R:414, R:457)

    def __init__(self, input_dims, latent_dims, embedding_dims, generator_dims, discriminator_dims,
                 text_embedding_dims=768, patches_embedding_dims=1024, negative_slope=0.0, is_bn=False,
                 lr_d=5e-4, lr_g=5e-4, optimizer="rms_prop", gp_weight=10, p_aug=0, norm_scale=0.5, train=True,
                 n_critic=5, freq_print=2, freq_compute_test=50, freq_visualize_test=100, patience=10,
                 normalization="standardize", log2=False, rpm=False, results_dire="",
                 # --- extensions (keyword-only in spirit; defaults reproduce the reference) ---
                 dropout=0.1, seed=0, device=None, process_group=None, precision="bf16x3"):
        self.input_dims = input_dims
        self.latent_dims = latent_dims
        self.embedding_dims = embedding_dims
        self.generator_dims = generator_dims
        self.discriminator_dims = discriminator_dims
        self.text_embedding_dims = text_embedding_dims
        self.patches_embedding_dims = patches_embedding_dims
        self.negative_slope = negative_slope
        self.is_bn = is_bn
        self.gp_weight = gp_weight
        self.isTrain = train
        self.p_aug = p_aug
        self.norm_scale = norm_scale
        self.n_genes = input_dims
        self.n_critic = n_critic
        self.freq_print = freq_print
        self.freq_compute_test = freq_compute_test
        self.freq_visualize_test = freq_visualize_test
        self.result_dire = results_dire
        if results_dire:
            os.makedirs(self.result_dire, exist_ok=True)
            self.results_dire_fig = os.path.join(self.result_dire, "figures")
            os.makedirs(self.results_dire_fig, exist_ok=True)
        self.lr_d, self.lr_g = lr_d, lr_g
        self.optimizer = optimizer
        self.patience = patience
        if p_aug != 0:
            raise NotImplementedError("p_aug != 0 raises NameError in the reference (R:401); unsupported")
        if optimizer.lower() not in L.OPT_KINDS:
            raise ValueError(f"unknown optimizer {optimizer!r}")
        if device is None:
            device = "cuda:0" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        self.loss_dict = {"d loss": [], "d real loss": [], "d fake loss": [], "g loss": []}
        self.normalization, self.log2, self.rpm = normalization, log2, rpm
        self.dropout = dropout
        self.seed = seed
        self.process_group = process_group
        # "bf16x3" (default): the fused kernels on fp32 operands split into bf16 parts - the reference's <= 1e-3 tolerance at 2.5x the speed of
        # "f32" (exact fp32-input MFMA on the generic kernels; shapes the split-operand kernels do not take fall back to it per call);
        # "bf16": bf16 MFMA operands, fp32 accumulate (the throughput mode of bench.py); "fp8": bf16 with e4m3 forward Linears
        self.precision = precision
        self.engine: Optional[Engine] = None
        self.gen = self.disc = None
        self.optimizer_disc = self.optimizer_gen = None
        self._noise_gen = None
        # Opt-in (use_graph = True before the first step, or GG_GRAPH=1): gg_train_step replayed from a captured hipGraph once
        # a resident batch has been seen twice (include/gemmgan.h).  Off by default: on ROCm 7.2 a replay halves the host's
        # enqueue time (4.4 -> 2.2 ms per cfg3 step) but the GPU runs the graph's nodes SLOWER than the same kernels enqueued
        # on three streams (38.5 vs 36.0 ms at cfg3, 2.03 vs 1.88 ms at configs[0]), and the host is not the bottleneck.
        self.use_graph = os.environ.get("GG_GRAPH", "0") == "1"
        self.measure_comm = False
        self._comm_events = []

    # ---- construction (R:334-349, R:320-331) -------------------------------------------------------
    def _build_nets(self):
        return WGAN_GP_model(self.latent_dims, self.input_dims, self.embedding_dims, self.generator_dims,
                             self.discriminator_dims, self.text_embedding_dims, self.patches_embedding_dims,
                             self.negative_slope, self.is_bn)

    def build_WGAN_GP(self):
        self.gen, self.disc = self._build_nets()
        if self.device.type != "cuda":
            raise RuntimeError("gemm_gan_amd.WGAN_GP needs a ROCm GPU: there is no CPU fallback "
                               "(use oracle/torch_oracle.py for CPU checks)")
        self._make_engine(8, 16, 1)

    def init_train(self):
        kind = self.optimizer.lower()
        self.optimizer_disc = _EngineOptimizer(kind, self.lr_d)
        self.optimizer_gen = _EngineOptimizer(kind, self.lr_g)

    def _make_engine(self, B, P, T):
        H = self.generator_dims[0]
        old = self.engine
        eng = Engine(n_genes=self.input_dims, latent_dims=self.latent_dims, embedding_dims=self.embedding_dims,
                     hidden_dims=H, text_dims=self.text_embedding_dims, patch_dims=self.patches_embedding_dims,
                     negative_slope=self.negative_slope, dropout=self.dropout, lr_d=self.lr_d, lr_g=self.lr_g,
                     optimizer=self.optimizer, gp_weight=float(self.gp_weight), max_batch=B, max_patches=P,
                     max_text_tokens=T, seed=self.seed, device=self.device, precision=self.precision,
                     variant=self._variant, clip_d=self._clip[0], clip_g=self._clip[1])
        eng._owner = self
        eng.set_graph(self.use_graph)
        if old is not None:          # grow: keep parameters, gradients, optimiser state and step counters
            for r in (L.ROLE_GENERATOR, L.ROLE_CRITIC):
                for k in ("w", "g", "s1", "s2"):
                    eng.flat[r][k].copy_(old.flat[r][k])
                L.check(eng.lib.gg_set_optimizer_step(eng.h, r, old.lib.gg_get_optimizer_step(old.h, r)))
        self.engine = eng
        self.gen._bind(eng)
        self.disc._bind(eng)
        if old is not None:
            old.close()

    def reserve(self, batch, patches, tokens):
        """Size the workspace for the largest minibatch that will be seen (grow-only)."""
        self._ensure_capacity(batch, patches, tokens)

    def _ensure_capacity(self, B, P, T):
        c = self.engine.cfg
        if B > c.max_batch or P > c.max_patches or T > c.max_text_tokens:
            self._make_engine(max(B, c.max_batch), max(P, c.max_patches), max(T, c.max_text_tokens))

    def _sync_lr(self):
        if self.optimizer_disc is not None:
            self.engine.set_lr(L.ROLE_CRITIC, self.optimizer_disc.param_groups[0]["lr"])
            self.engine.set_lr(L.ROLE_GENERATOR, self.optimizer_gen.param_groups[0]["lr"])

    # ---- data-parallel plumbing ----------------------------------------------------------------
    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.process_group)
        return 1

    @staticmethod
    def _collectives_at_world_1():
        """GG_FORCE_DP_LOOP=1 GG_FORCE_DP_COLLECTIVES=1 with an initialised ONE-rank process group: the data-parallel host loop issues
        every all-reduce it would issue on N ranks (measurement on a one-GPU box: RCCL's launch path and the stream pattern, not its
        transport)."""
        import torch.distributed as dist
        return os.environ.get("GG_FORCE_DP_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized()

    def _rank(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.process_group)
        return 0

    def _noise_generator(self):
        """z / alpha stream of this rank.  One rank: the default torch generator, i.e. exactly the reference's draws
        (R:473, R:354, R:476).  Data parallel: every rank owns a shard of the global minibatch and must draw ITS rows'
        noise, so each rank gets its own generator seeded from (torch.initial_seed(), rank) - a global batch of N*B then
        has N*B distinct latent vectors, as one GPU on that batch would."""
        if self._world() == 1:
            return None
        if self._noise_gen is None:
            g = torch.Generator(device=self.device)
            g.manual_seed((torch.initial_seed() + 1000003 * (self._rank() + 1)) % (2 ** 63))
            self._noise_gen = g
        return self._noise_gen

    def _allreduce(self, role):
        import torch.distributed as dist
        dist.all_reduce(self.engine.flat[role]["g"], op=dist.ReduceOp.SUM, group=self.process_group)

    def _allreduce_bucket(self, role, which):
        """Asynchronous SUM all-reduce of one of the two gradient buckets: 'mlp' = the MLP-head slots (complete after the
        head phase of the backward), 'cond' = the conditioning-stack slots (complete at its end)."""
        import torch.distributed as dist
        if self._world() == 1 and not self._collectives_at_world_1():      # GG_FORCE_DP_LOOP=1 at world size 1: the host loop without its collectives (measurement)
            return None
        off, numel = self.engine.mlp_range[role]
        g = self.engine.flat[role]["g"]
        t = g[off:off + numel] if which == "mlp" else g[:off]
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)

    def _allreduce_stage(self, role, stage):
        """Asynchronous SUM all-reduce of the gradient range conditioning-backward stage `stage` has just enqueued.  The range's
        weight gradients are leaves on the engine's side stream: the collective is issued FROM that stream after it has joined the
        caller's (it then depends on both streams' share of the stage), so the caller's stream goes on with the next stage at once."""
        import torch.distributed as dist
        if self._world() == 1 and not self._collectives_at_world_1():
            return None
        off, numel = self.engine.stage_range[role][stage]
        if numel == 0:
            return None
        t = self.engine.flat[role]["g"][off:off + numel]
        side = getattr(self.engine, "_side_stream", None)
        if side is not None and t.is_cuda:
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)

    def _wait(self, *works):
        """The compute stream waits for the collectives; with `measure_comm` the exposed wait is timed with an event pair."""
        timed = self.measure_comm and self.device.type == "cuda"
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in works:
            if w is not None:
                w.wait()
        if timed:
            e1.record()
            self._comm_events.append((e0, e1))

    def comm_wait_ms(self, reset=True):
        """Milliseconds the compute stream spent waiting for gradient all-reduces since the last reset (synchronises)."""
        ms = 0.0
        for e0, e1 in self._comm_events:
            e1.synchronize()
            ms += e0.elapsed_time(e1)
        if reset:
            self._comm_events = []
        return ms

    def _prep(self, gene_expression, text_token, text_token_padding, patches, padding_mask):
        dev = self.device
        x = gene_expression.to(torch.float32).to(dev).contiguous()
        text = text_token.to(dev, torch.float32).contiguous()
        tpad = text_token_padding.to(dev)
        pat = patches.to(dev, torch.float32).contiguous()
        ppad = padding_mask.to(dev)
        self._ensure_capacity(x.shape[0], pat.shape[1], text.shape[1])
        return x, text, tpad, pat, ppad

    # ---- gradient penalty as a public call (R:351-374; note the argument order, which differs from train_disc) -----------
    def gradient_penalty(self, real_data, fake_data, patches, padding_mask, text_token, text_token_padding):
        """0-d tensor mean((||grad_x^ D(x^)||_2 - 1)^2), x^ = alpha*real + (1-alpha)*fake with alpha ~ U[0,1) drawn here
        by the same torch.rand call as R:354.  The reference returns a tensor with an autograd graph; training on this
        path goes through train_disc, whose double backward is hand-written, so the value carries no graph."""
        x, text, tpad, pat, ppad = self._prep(real_data, text_token, text_token_padding, patches, padding_mask)
        alpha = torch.rand(x.shape[0], 1, device=self.device)
        return self.engine.gradient_penalty(x, fake_data, alpha, pat, ppad, text, tpad, train=self.disc.training)

    # ---- trainer (R:376-477) ------------------------------------------------------------------------------------
    def train_disc(self, real_data, z, text_token, text_token_padding, patches, padding_mask):
        x, text, tpad, pat, ppad = self._prep(real_data, text_token, text_token_padding, patches, padding_mask)
        self._sync_lr()
        alpha = torch.rand(x.shape[0], 1, device=self.device)             # R:354 (same draw, same place)
        eng = self.engine
        z = z.to(self.device, torch.float32).contiguous()
        w = self._world()
        if w == 1:
            eng.critic_backward(x, z, alpha, pat, ppad, text, tpad)
        else:
            self._critic_iteration_dp(x, z, alpha, pat, ppad, text, tpad)
        eng.critic_apply(1.0 / w)
        self._publish_critic_losses()

    def train_gen(self, z, text_token, text_token_padding, patches, padding_mask):
        dev = self.device
        text = text_token.to(dev, torch.float32).contiguous()
        pat = patches.to(dev, torch.float32).contiguous()
        self._ensure_capacity(z.shape[0], pat.shape[1], text.shape[1])
        self._sync_lr()
        eng = self.engine
        z = z.to(dev, torch.float32).contiguous()
        ppad, tpad = padding_mask.to(dev), text_token_padding.to(dev)
        w = self._world()
        if w == 1:
            eng.generator_backward(z, pat, ppad, text, tpad)
        else:
            self._generator_iteration_dp(z, pat, ppad, text, tpad)
        eng.generator_apply(1.0 / w)
        self._publish_gen_loss()

    def _critic_iteration_dp(self, x, z, alpha, pat, ppad, text, tpad):
        """Backward of one critic iteration with the all-reduce in two buckets: the MLP-head + gradient-penalty gradients
        travel while the conditioning stack's backward runs (SURVEY 8e "Overlap")."""
        eng = self.engine
        eng.critic_backward_head(x, z, alpha, pat, ppad, text, tpad)
        hs = [self._allreduce_bucket(L.ROLE_CRITIC, "mlp")]
        n_st = getattr(eng, "cond_stages", 0)
        if n_st > 0 and not os.environ.get("GG_DP_TWO_BUCKETS"):
            # reverse-layer buckets (round 4): cross-attention, encoder layers last to first, then CLS / patch encoder / FiLM / text
            # encoder - contiguous ranges of the flat buffer in backward order; only the last one is exposed
            for st in range(n_st):
                eng.critic_backward_cond_stage(st, pat, ppad, text, tpad)
                hs.append(self._allreduce_stage(L.ROLE_CRITIC, st))
        else:
            eng.critic_backward_cond(pat, ppad, text, tpad)
            hs.append(self._allreduce_bucket(L.ROLE_CRITIC, "cond"))
        self._wait(*hs)

    def _generator_iteration_dp(self, z, pat, ppad, text, tpad, next_cond=None):
        """Same for the generator; `next_cond` = (patches, pad, text, text_pad) of the NEXT train(): the critic's conditioning
        forward of its first critic iteration does not depend on the generator and runs under the all-reduce."""
        eng = self.engine
        eng.generator_backward_head(z, pat, ppad, text, tpad)
        hs = [self._allreduce_bucket(L.ROLE_GENERATOR, "mlp")]
        n_st = getattr(eng, "cond_stages", 0)
        if n_st > 0 and not os.environ.get("GG_DP_TWO_BUCKETS"):
            for st in range(n_st):
                eng.generator_backward_cond_stage(st, pat, ppad, text, tpad)
                hs.append(self._allreduce_stage(L.ROLE_GENERATOR, st))
        else:
            eng.generator_backward_cond(pat, ppad, text, tpad)
            hs.append(self._allreduce_bucket(L.ROLE_GENERATOR, "cond"))
        if next_cond is not None:
            eng.critic_cond_prefetch(*next_cond)
        self._wait(*hs)

    # ---- losses (R:421-423, R:458-460) -------------------------------------------------------------------------------
    # The reference publishes disc_loss / gen_loss (0-d tensors) and d_batch_loss / g_batch_loss (numpy) after every call,
    # with three .item() host syncs per critic iteration.  Here a call leaves a DEVICE snapshot of the engine's loss
    # slots (an async 24-byte copy on the same stream); the attributes are computed from it when read, so a loop that
    # does not look at the losses never stalls the host, and one that does (fit(), like the reference) syncs once.
    def _global_mean(self, t):
        """Data parallel: the batch means of the loss slots are per shard; equal shards => mean of means = global mean."""
        w = self._world()
        if w > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group)
            t /= w
        return t

    def _publish_critic_losses(self):
        self._crit_snap = self._global_mean(self.engine.losses.clone())
        self._crit_host = None

    def _publish_gen_loss(self):
        self._gen_snap = self._global_mean(self.engine.losses[L.LOSS_G:L.LOSS_G + 1].clone())[0]
        self._gen_host = None

    def _crit_values(self):
        if self._crit_host is None:
            l = self._crit_snap.tolist()                                    # one host sync (reference: three)
            self._crit_host = (l[L.LOSS_D_REAL], l[L.LOSS_D_FAKE], l[L.LOSS_GP])
        return self._crit_host

    @property
    def disc_loss(self):
        s = self._crit_snap
        return s[L.LOSS_D_REAL] + s[L.LOSS_D_FAKE] + self.gp_weight * s[L.LOSS_GP]

    @property
    def d_batch_loss(self):
        d_real, d_fake, _ = self._crit_values()
        return np.array([d_real + d_fake, d_real, d_fake])

    @property
    def gp_value(self):
        return self._crit_values()[2]

    @property
    def gen_loss(self):
        return self._gen_snap

    @property
    def g_batch_loss(self):
        if self._gen_host is None:
            self._gen_host = float(self._gen_snap)
        return np.array([self._gen_host])

    def train(self, gene_expression, text_token, text_token_padding, patches, padding_mask, next_batch=None):
        """One WGAN-GP step (R:463-477).  `next_batch` (extension, data parallel only): the (gene_expression, text_token,
        text_token_padding, patches, padding_mask) of the following call, whose critic conditioning forward then runs
        under this step's generator all-reduce; results are identical with or without it."""
        x, text, tpad, pat, ppad = self._prep(gene_expression, text_token, text_token_padding, patches, padding_mask)
        self._sync_lr()
        B, n = x.shape[0], self.n_critic
        dev = self.device
        gen = self._noise_generator()
        # same torch RNG draws, in the same order, as the reference loop (z, alpha) x n_critic, then z (R:472-476)
        z_all = torch.empty(n + 1, B, self.latent_dims, device=dev)
        alpha_all = torch.empty(n, B, device=dev)
        # (drawn straight into their slots: `out=` takes the same draws from the generator as the reference's allocating calls - one
        # kernel per draw instead of a draw and a copy)
        for k in range(n):
            torch.normal(0, 1, size=(B, self.latent_dims), generator=gen, out=z_all[k])
            torch.rand(B, 1, generator=gen, out=alpha_all[k].view(B, 1))
        torch.normal(0, 1, size=(B, self.latent_dims), generator=gen, out=z_all[n])
        next_cond = None
        if next_batch is not None and (self._world() > 1 or self._collectives_at_world_1()):
            _, ntext, ntpad, npat, nppad = self._prep(*next_batch)
            next_cond = (npat, nppad, ntext, ntpad)
        self.train_with_noise(x, text, tpad, pat, ppad, z_all, alpha_all, next_cond=next_cond)

    def train_with_noise(self, x, text, tpad, pat, ppad, z_all, alpha_all, sync_losses=True, next_cond=None):
        """One train() with explicit z [n_critic+1,B,L] / alpha [n_critic,B] (parity and DP tests)."""
        eng = self.engine
        self._sync_lr()
        w = self._world()
        n = alpha_all.shape[0]
        if w == 1 and os.environ.get("GG_FORCE_DP_LOOP") != "1":
            eng.train_step(x, pat, ppad, text, tpad, z_all, alpha_all)     # whole step enqueued by ONE C call
        else:
            # One flat-buffer SUM all-reduce per optimiser step, issued as two buckets so that it overlaps the backward
            # (SURVEY 8e); clip + step on the averaged gradient, identical on every rank.
            eng.reset_launch_count()                                                    # gg_train_step does this by itself
            if n > 1:
                eng.generator_prefetch(z_all[:n].contiguous(), pat, ppad, text, tpad)   # frozen generator: all n passes at once
            for k in range(n):
                self._critic_iteration_dp(x, z_all[k], alpha_all[k], pat, ppad, text, tpad)
                eng.critic_apply(1.0 / w)
            self._generator_iteration_dp(z_all[n], pat, ppad, text, tpad, next_cond)
            eng.generator_apply(1.0 / w)
        if sync_losses:
            self._publish_critic_losses()
            self._publish_gen_loss()

    # ---- inference (R:601-608) -------------------------------------------------------------------------------------
    def generate_samples(self, gene_expression, text_embedding, text_padding, patches, padding_mask):
        with torch.no_grad():
            self.gen.eval()
            x_real = gene_expression.clone().to(torch.float32)
            z = torch.normal(0, 1, size=(x_real.shape[0], self.latent_dims), device=self.device)
            x_gen = self.gen(z, patches, padding_mask, text_embedding, text_padding)
        return x_real, x_gen

    def _fit_batch(self, data, nxt=None):
        nb = None if nxt is None else (nxt[2], nxt[0], nxt[1], nxt[3], nxt[4])
        self.train(data[2], data[0], data[1], data[3], data[4], next_batch=nb)             # R:667-673

    # ---- sample dumps consumed by the reference's evaluators (R:561-599 balanced=False branch, R:793-806) -------------
    def generate_samples_all(self, data_loader, num_repeats=1, balanced=False, balanced_max_oversample=5):
        """(all_real_x, all_gen_x, disease_real, disease_gen, primary_site_real, primary_site_gen) as numpy arrays, one
        generator pass per minibatch and repeat (R:561-599).  `balanced=True` raises NameError in the reference (R:531:
        `text_padding` used before assignment) and is not offered."""
        if balanced:
            raise NotImplementedError("generate_samples_all(balanced=True) is broken upstream (R:531); use balanced=False")
        all_real, all_gen, dis_real, dis_gen, site_real, site_gen = [], [], [], [], [], []
        for i in range(num_repeats):
            for batch in data_loader:
                x_real, x_gen = self._generate_from_batch(batch)
                dis, site = self._labels_of(batch)
                all_gen.append(x_gen.detach().cpu().numpy())
                dis_gen.append(dis)
                site_gen.append(site)
                if i == 0:
                    all_real.append(x_real.detach().cpu().numpy())
                    dis_real.append(dis)
                    site_real.append(site)
        return (np.vstack(all_real), np.vstack(all_gen), np.concatenate(dis_real, axis=0), np.concatenate(dis_gen, axis=0),
                np.concatenate(site_real, axis=0), np.concatenate(site_gen, axis=0))

    def _generate_from_batch(self, batch):
        dev = self.device
        return self.generate_samples(batch[2].to(dev), batch[0].to(dev), batch[1].to(dev), batch[3].to(dev), batch[4].to(dev))

    @staticmethod
    def _labels_of(batch):
        return batch[5].detach().cpu().numpy(), batch[6].detach().cpu().numpy()

    def dump_generated(self, train_data, test_data, epoch, n_runs=2):
        """The twelve .npy files per run that the reference writes at the last epoch (R:786-806) and its evaluation scripts
        read back: <results_dire>/test_<run>_epoch_<epoch+1>/{data,test}_{real,gen}.npy, {train,test}_labels_{real,gen}.npy,
        {train,test}_primary_site_{real,gen}.npy."""
        out = []
        for run in range(n_runs):
            tr = self.generate_samples_all(train_data)
            te = self.generate_samples_all(test_data)
            d = os.path.join(self.result_dire, f"test_{run}_epoch_{epoch + 1}")
            os.makedirs(d, exist_ok=True)
            files = {"data_real": tr[0], "data_gen": tr[1], "test_real": te[0], "test_gen": te[1],
                     "train_labels_real": tr[2], "train_labels_gen": tr[3], "test_labels_real": te[2], "test_labels_gen": te[3],
                     "train_primary_site_real": tr[4], "train_primary_site_gen": tr[5],
                     "test_primary_site_real": te[4], "test_primary_site_gen": te[5]}
            for name, arr in files.items():
                with open(os.path.join(d, name + ".npy"), "wb") as f:
                    np.save(f, arr)
            out.append(d)
        return out

    # ---- epoch loop (R:619-711): training, LR halving, loss bookkeeping, checkpoints ----------------------------
    def fit(self, train_data, val_data=None, test_data=None, epochs=1, val=False):
        """Training part of the reference fit(): LR schedule, loss bookkeeping, checkpoints and - with `val` and a test
        loader - the generated-sample dumps of the last epoch (R:786-806).  Metrics / plots (R:712-785, R:807-894) are the
        sklearn / matplotlib side of the reference and stay out of scope."""
        self.build_WGAN_GP()
        if self.isTrain:
            self.init_train()
        lookahead = self._world() > 1
        for epoch in range(epochs):
            if epoch % 100 == 0 and epoch != 0:                                  # R:649-657
                for opt in (self.optimizer_disc, self.optimizer_gen):
                    for group in opt.param_groups:
                        group["lr"] = group["lr"] * 0.50
            self.epoch = epoch
            d_loss_all, d_batch_loss, g_batch_loss, nb = 0.0, None, None, 0
            it = iter(train_data)
            data = next(it, None)
            i = 0
            while data is not None:
                nxt = next(it, None)
                self._fit_batch(data, nxt if lookahead else None)
                d_loss_all += self.disc_loss.item()
                d_batch_loss = self.d_batch_loss if d_batch_loss is None else d_batch_loss + self.d_batch_loss
                g_batch_loss = self.g_batch_loss if g_batch_loss is None else g_batch_loss + self.g_batch_loss
                nb += 1
                if (i + 1) % self.freq_print == 0:
                    print("[Epoch %d/%d] [Batch %d/%d] [D loss : %f] [G loss : %f]"
                          % (epoch + 1, epochs, i + 1, len(train_data), self.disc_loss.item(), self.gen_loss.item()))
                data = nxt
                i += 1
            d_batch_loss = d_batch_loss / max(nb, 1)
            self.loss_dict["d loss"].append(d_batch_loss[0])
            self.loss_dict["d real loss"].append(d_batch_loss[1])
            self.loss_dict["d fake loss"].append(d_batch_loss[2])
            self.loss_dict["g loss"].append(g_batch_loss[0])
            print("Averge D Loss:", d_loss_all / max(nb, 1))
            if self.result_dire and (epoch + 1) % self.freq_compute_test == 0:       # R:710-711
                torch.save(self.gen.state_dict(), os.path.join(self.result_dire, f"generator_epoch_{epoch + 1}.pt"))
                torch.save(self.disc.state_dict(), os.path.join(self.result_dire, f"discriminator_epoch_{epoch + 1}.pt"))
                if val and test_data is not None and (epoch + 1) == epochs:              # R:739, R:786-806
                    self.dump_generated(train_data, test_data, epoch)
        if self.result_dire:                                                           # R:743-744
            torch.save(self.gen.state_dict(), os.path.join(self.result_dire, "generator_last_epoch.pt"))
            torch.save(self.disc.state_dict(), os.path.join(self.result_dire, "discriminator_last_epoch.pt"))
        return self.loss_dict
