"""Thin torch-side owner of the device buffers the C-ABI engine works on.

PyTorch is plumbing here: it allocates HBM (flat parameter / gradient / optimiser-state buffers and
one workspace arena), provides the HIP stream, and runs ``torch.distributed`` for the data-parallel
gradient all-reduce.  All arithmetic of the hot path happens in libgemmgan.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import _lib as L


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    """One engine = both networks + trainer state for one GPU (one process per GPU)."""

    def __init__(self, *, n_genes, latent_dims, embedding_dims, hidden_dims, text_dims, patch_dims,
                 n_heads=4, n_layers=2, negative_slope=0.0, dropout=0.1, lr_d=5e-4, lr_g=5e-4,
                 optimizer="rms_prop", gp_weight=10.0, clip_d=10.0, clip_g=2.0, max_batch=8, max_patches=256,
                 max_text_tokens=1, seed=0, device="cuda:0", precision="bf16x3", variant="xattn_film"):
        self.lib = L.load()
        if L.lab_wanted_by_env():        # an environment switch selects a kernel of libgemmgan_lab.so: register it before gg_create
            L.load_lab()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("gemm_gan_amd needs a ROCm GPU (cuda:N device); there is no CPU fallback")
        self.cfg = L.GGConfig(n_genes, latent_dims, embedding_dims, hidden_dims, text_dims, patch_dims, n_heads,
                              n_layers, negative_slope, dropout, lr_d, lr_g, L.OPT_KINDS[optimizer.lower()], gp_weight,
                              clip_d if clip_d else 0.0, clip_g if clip_g else 0.0, max_batch, max_patches,
                              max_text_tokens, seed, L.PRECISIONS[precision], L.VARIANTS[variant])
        self.precision = precision
        self.variant = variant
        self.h = C.c_void_p()
        L.check(self.lib.gg_create(C.byref(self.cfg), C.byref(self.h)))
        self.layout = {r: self._read_layout(r) for r in (L.ROLE_GENERATOR, L.ROLE_CRITIC)}
        self.flat: Dict[int, Dict[str, torch.Tensor]] = {}
        with torch.cuda.device(self.device):
            for r in (L.ROLE_GENERATOR, L.ROLE_CRITIC):
                n = int(self.lib.gg_flat_numel(self.h, r))
                bufs = {k: torch.zeros(n, dtype=torch.float32, device=self.device) for k in ("w", "g", "s1", "s2")}
                self.flat[r] = bufs
                L.check(self.lib.gg_bind_net(self.h, r, _ptr(bufs["w"]), _ptr(bufs["g"]), _ptr(bufs["s1"]), _ptr(bufs["s2"])))
            self._alloc_workspace()
            self.losses = torch.zeros(L.N_LOSSES, dtype=torch.float32, device=self.device)
            self.graph, self._stage, self._zeros = False, {}, {}
            # The engine's concurrent work (parameter-gradient leaves, generator passes computed ahead) runs on two streams
            # this object owns and binds: kernels on them read the caller's input tensors after an entry point has returned,
            # and the caching allocator orders a block's reuse only against streams the tensor was recorded on (_borrow).
            # GG_SIDE_PRIO=high: torch's high-priority stream pool (the one ProcessGroupNCCL also draws from); anything else, "low" included,
            # is the default priority - torch exposes no lower one (the C side's own streams know low|high: engine.hip create_side_stream)
            prio = -1 if os.environ.get("GG_SIDE_PRIO", "") .startswith("h") else 0
            self._side_stream = torch.cuda.Stream(self.device, priority=prio)
            self._pre_stream = torch.cuda.Stream(self.device, priority=prio)
            L.check(self.lib.gg_bind_streams(self.h, C.c_void_p(self._side_stream.cuda_stream), C.c_void_p(self._pre_stream.cuda_stream)))
        self.dropout = float(dropout)
        off, numel = C.c_int64(), C.c_int64()
        self.mlp_range = {}
        for r in (L.ROLE_GENERATOR, L.ROLE_CRITIC):
            L.check(self.lib.gg_mlp_grad_range(self.h, r, C.byref(off), C.byref(numel)))
            self.mlp_range[r] = (off.value, numel.value)
        # stages of the conditioning backward and the gradient range each completes (reverse flat order: data-parallel hosts all-reduce
        # a finished stage's range while the next stage runs; include/gemmgan.h gg_cond_stage_range)
        self.cond_stages = self.lib.gg_cond_stage_count(self.h)
        self.stage_range = {}
        for r in (L.ROLE_GENERATOR, L.ROLE_CRITIC):
            self.stage_range[r] = []
            for st in range(self.cond_stages):
                L.check(self.lib.gg_cond_stage_range(self.h, r, st, C.byref(off), C.byref(numel)))
                self.stage_range[r].append((off.value, numel.value))

    # -- construction helpers ------------------------------------------------------------------
    def _read_layout(self, role):
        out = {}
        for i in range(self.lib.gg_param_count(self.h, role)):
            off, numel, ndim = C.c_int64(), C.c_int64(), C.c_int32()
            shape = (C.c_int32 * 3)()
            L.check(self.lib.gg_param_info(self.h, role, i, C.byref(off), C.byref(numel), C.byref(ndim), shape))
            out[self.lib.gg_param_name(self.h, role, i).decode()] = (off.value, numel.value, tuple(shape[:ndim.value]))
        return out

    def _alloc_workspace(self):
        nbytes = int(self.lib.gg_workspace_bytes(self.h))
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = self.workspace.data_ptr()
        aligned = (base + 255) // 256 * 256
        L.check(self.lib.gg_bind_workspace(self.h, C.c_void_p(aligned), C.c_size_t(nbytes)))
        self.workspace_bytes = nbytes

    def close(self):
        if getattr(self, "h", None):
            self.lib.gg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameter access ------------------------------------------------------------------------
    def view(self, role, name, which="w") -> torch.Tensor:
        off, numel, shape = self.layout[role][name]
        t = self.flat[role][which][off:off + numel].view(shape)
        if self.variant == "vanilla" and name.endswith(".0.0.weight"):
            # GG_VARIANT_VANILLA keeps embedding_dims zero columns behind the reference's [H, V] first-layer weight
            t = t[:, : shape[1] - self.cfg.embedding_dims]
        return t

    def load_state(self, role, state: Dict[str, torch.Tensor]):
        for name in self.layout[role]:
            self.view(role, name).copy_(state[name].to(self.device, torch.float32))

    def state(self, role, which="w") -> Dict[str, torch.Tensor]:
        return {name: self.view(role, name, which) for name in self.layout[role]}

    # -- calls ----------------------------------------------------------------------------------------
    def _cond(self, patches, patch_pad, text, text_pad):
        for t in (patches, text):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device, "fp32 contiguous GPU tensor expected"
        if patches.dim() != 3 or text.dim() != 3 or patches.shape[0] != text.shape[0]:
            raise ValueError("patches [B,P,Dp] / text [B,T,Dt] expected")
        if patches.shape[2] != self.cfg.patch_dims or text.shape[2] != self.cfg.text_dims:
            raise ValueError("embedding width mismatch")
        B, P, T = patches.shape[0], patches.shape[1], text.shape[1]
        if tuple(patch_pad.shape) != (B, P) or tuple(text_pad.shape) != (B, T):
            raise ValueError("padding mask shape mismatch")
        pp, tp = self._mask_bytes(patch_pad), self._mask_bytes(text_pad)
        keep = (patches, pp, text, tp)
        self._borrow(*keep)
        return L.GGCond(patches.data_ptr(), pp.data_ptr(), text.data_ptr(), tp.data_ptr(), B, P, T), keep

    def _borrow(self, *tensors):
        """Inputs are borrowed for the call, but kernels on the engine's side / prefetch streams may read them after it has
        returned (and after the caller dropped its reference).  Recording the tensor on those streams makes the caching
        allocator hold the block until the work enqueued on them so far has finished - an ordering guarantee, where
        earlier rounds kept the last 16 argument sets alive and hoped."""
        for t in tensors:
            if t is not None and t.is_cuda:
                t.record_stream(self._side_stream)
                t.record_stream(self._pre_stream)

    def _mask_bytes(self, m: torch.Tensor) -> torch.Tensor:
        """torch.bool masks are one byte per element: the kernels read the caller's tensor in place (no uint8 temporary)."""
        if m.device != self.device:
            m = m.to(self.device)
        if not m.is_contiguous():
            m = m.contiguous()
        if m.dtype == torch.bool:
            return m.view(torch.uint8)
        return m if m.dtype == torch.uint8 else (m != 0).view(torch.uint8)

    def forward(self, role, v, patches, patch_pad, text, text_pad, train=False) -> torch.Tensor:
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        width = self.cfg.latent_dims if role == L.ROLE_GENERATOR else self.cfg.n_genes
        if v.dim() != 2 or v.shape[0] != cond.B or v.shape[1] != width:
            raise ValueError(f"first input must be [B,{width}]")
        v = v.to(torch.float32).contiguous()
        out = torch.empty(cond.B, self.cfg.n_genes if role == L.ROLE_GENERATOR else 1, dtype=torch.float32, device=self.device)
        L.check(self.lib.gg_forward(self.h, role, _ptr(v), C.byref(cond), _ptr(out), int(train), _stream()))
        return out

    def critic_backward(self, x_real, z, alpha, patches, patch_pad, text, text_pad):
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        alpha = alpha.reshape(-1).contiguous()
        self._borrow(x_real, z, alpha)
        L.check(self.lib.gg_critic_backward(self.h, _ptr(x_real), _ptr(z), _ptr(alpha), C.byref(cond), _ptr(self.losses), _stream()))

    def critic_backward_head(self, x_real, z, alpha, patches, patch_pad, text, text_pad):
        """First phase of a critic iteration: returns with the MLP-head gradient slots (`mlp_range`) complete."""
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        alpha = alpha.reshape(-1).contiguous()
        self._borrow(x_real, z, alpha)
        L.check(self.lib.gg_critic_backward_head(self.h, _ptr(x_real), _ptr(z), _ptr(alpha), C.byref(cond), _ptr(self.losses), _stream()))

    def critic_backward_cond(self, patches, patch_pad, text, text_pad):
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        L.check(self.lib.gg_critic_backward_cond(self.h, C.byref(cond), _stream()))

    def critic_backward_cond_stage(self, stage, patches, patch_pad, text, text_pad):
        """Stage `stage` (0 .. cond_stages - 1, in order) of the conditioning phase: returns with stage_range[critic][stage] enqueued
        (its weight-gradient leaves on the side stream)."""
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        L.check(self.lib.gg_critic_backward_cond_stage(self.h, C.byref(cond), int(stage), _stream()))

    def generator_backward_cond_stage(self, stage, patches, patch_pad, text, text_pad):
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        L.check(self.lib.gg_generator_backward_cond_stage(self.h, C.byref(cond), int(stage), _stream()))

    def critic_cond_prefetch(self, patches, patch_pad, text, text_pad):
        """The critic's conditioning pass of the next critic iteration on this minibatch, computed ahead."""
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        L.check(self.lib.gg_critic_cond_prefetch(self.h, C.byref(cond), _stream()))

    def gradient_penalty(self, x_real, x_fake, alpha, patches, patch_pad, text, text_pad, train=True) -> torch.Tensor:
        """0-d tensor mean((|grad_x^ D(x^)| - 1)^2) for x^ = alpha*real + (1-alpha)*fake (R:351-374); no gradients written."""
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        for t in (x_real, x_fake):
            if t.dim() != 2 or t.shape[0] != cond.B or t.shape[1] != self.cfg.n_genes:
                raise ValueError(f"real / fake must be [B,{self.cfg.n_genes}]")
        x_real = x_real.to(self.device, torch.float32).contiguous()
        x_fake = x_fake.to(self.device, torch.float32).contiguous()
        alpha = alpha.to(self.device, torch.float32).reshape(-1).contiguous()
        if alpha.numel() != cond.B:
            raise ValueError("alpha must have one entry per sample")
        out = torch.zeros(1, dtype=torch.float32, device=self.device)
        L.check(self.lib.gg_gradient_penalty(self.h, _ptr(x_real), _ptr(x_fake), _ptr(alpha), C.byref(cond), int(train), _ptr(out), _stream()))
        self._borrow(x_real, x_fake, alpha)
        return out[0]

    def generator_backward_head(self, z, patches, patch_pad, text, text_pad):
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        self._borrow(z)
        L.check(self.lib.gg_generator_backward_head(self.h, _ptr(z), C.byref(cond), _ptr(self.losses), _stream()))

    def generator_backward_cond(self, patches, patch_pad, text, text_pad):
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        L.check(self.lib.gg_generator_backward_cond(self.h, C.byref(cond), _stream()))

    def critic_apply(self, grad_scale=1.0):
        L.check(self.lib.gg_critic_apply(self.h, C.c_float(grad_scale), _stream()))

    def generator_backward(self, z, patches, patch_pad, text, text_pad):
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        self._borrow(z)
        L.check(self.lib.gg_generator_backward(self.h, _ptr(z), C.byref(cond), _ptr(self.losses), _stream()))

    def generator_apply(self, grad_scale=1.0):
        L.check(self.lib.gg_generator_apply(self.h, C.c_float(grad_scale), _stream()))

    def generator_prefetch(self, z_all, patches, patch_pad, text, text_pad):
        """Generator outputs for the next len(z_all) critic iterations, computed ahead as stacked replicas (the
        generator is frozen in between); the following critic_backward calls consume them in order."""
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        assert z_all.dim() == 3 and z_all.is_contiguous()
        self._borrow(z_all)
        L.check(self.lib.gg_generator_prefetch(self.h, _ptr(z_all), int(z_all.shape[0]), C.byref(cond), _stream()))

    def set_side_streams(self, on):
        L.check(self.lib.gg_set_side_streams(self.h, int(bool(on))))

    def set_prefetch(self, on):
        L.check(self.lib.gg_set_prefetch(self.h, int(bool(on))))

    def zeros(self, *shape, dtype=torch.float32):
        """A read-only all-zero tensor that lives as long as the engine (all-False masks and dummy conditioning inputs of the
        sibling facades: the same address every call, so a resident batch keeps its captured graph)."""
        key = (tuple(shape), dtype)
        t = self._zeros.get(key)
        if t is None:
            t = self._zeros[key] = torch.zeros(*shape, dtype=dtype, device=self.device)
        return t

    def set_graph(self, on):
        """Replay gg_train_step from a captured hipGraph once a set of input buffers has been seen twice (gemmgan.h)."""
        L.check(self.lib.gg_set_graph(self.h, int(bool(on))))
        self.graph = bool(on)
        self._stage = {}

    def graph_stats(self):
        v = [C.c_int64(0) for _ in range(3)]
        L.check(self.lib.gg_graph_stats(self.h, *[C.byref(x) for x in v]))
        return {"captures": v[0].value, "replays": v[1].value, "failures": v[2].value}

    def _staged(self, name, t):
        """A captured step reads its inputs from the addresses it was captured with: the per-step noise goes through
        engine-lifetime buffers (two small copies per step)."""
        key = (name, tuple(t.shape))
        buf = self._stage.get(key)
        if buf is None:
            buf = self._stage[key] = torch.empty_like(t)
        buf.copy_(t)
        return buf

    def train_step(self, x_real, patches, patch_pad, text, text_pad, z_all, alpha_all):
        cond, keep = self._cond(patches, patch_pad, text, text_pad)
        n_critic = alpha_all.shape[0]
        assert z_all.shape[0] == n_critic + 1 and z_all.is_contiguous() and alpha_all.is_contiguous()
        if getattr(self, "graph", False):
            z_all, alpha_all = self._staged("z", z_all), self._staged("alpha", alpha_all)
        self._borrow(x_real, z_all, alpha_all)
        L.check(self.lib.gg_train_step(self.h, _ptr(x_real), C.byref(cond), _ptr(z_all), _ptr(alpha_all), n_critic,
                                       _ptr(self.losses), _stream()))

    def set_lr(self, role, lr):
        L.check(self.lib.gg_set_lr(self.h, role, C.c_float(lr)))

    def set_dropout(self, p):
        L.check(self.lib.gg_set_dropout(self.h, C.c_float(p)))
        self.dropout = float(p)

    def set_precision(self, precision: str):
        L.check(self.lib.gg_set_precision(self.h, L.PRECISIONS[precision]))
        self.precision = precision

    def set_flash(self, on: bool):
        L.check(self.lib.gg_set_flash(self.h, int(on)))

    def set_wgrad(self, on: bool):
        L.check(self.lib.gg_set_wgrad(self.h, int(on)))

    def set_bstore(self, on: bool):
        L.check(self.lib.gg_set_bstore(self.h, int(on)))

    def set_sqx(self, on: bool):
        L.check(self.lib.gg_set_sqx(self.h, int(on)))

    def set_head_fused(self, on: bool):
        """Fused MLP-head chain (csrc/head.hip, libgemmgan_lab.so: loaded here when switched on)."""
        if on:
            L.load_lab()
        L.check(self.lib.gg_set_head_fused(self.h, int(on)))

    def set_lnb_fused(self, on: bool):
        L.check(self.lib.gg_set_lnb_fused(self.h, int(on)))

    def set_xstore(self, on: bool):
        L.check(self.lib.gg_set_xstore(self.h, int(on)))

    def phase_enable(self, on: bool):
        """Timing events at the phase boundaries of train_step (include/gemmgan.h gg_phase_*)."""
        L.check(self.lib.gg_phase_enable(self.h, int(on)))

    def phase_times(self):
        """[(name, ms since the previous mark)] of the last train_step (synchronises)."""
        out = []
        buf, ms = C.create_string_buffer(128), C.c_double()
        for i in range(self.lib.gg_phase_count(self.h)):
            L.check(self.lib.gg_phase_read(self.h, i, buf, 128, C.byref(ms)))
            out.append((buf.value.decode(), ms.value))
        return out

    def set_encb(self, on: bool):
        """Fused backward of the token-local chain of an encoder layer (csrc/enc.hip encb_kernel, libgemmgan_lab.so)."""
        if on:
            L.load_lab()
        L.check(self.lib.gg_set_encb(self.h, int(on)))

    def set_ffn2(self, mode: int):
        """Streamed fused feed-forward block (csrc/enc.hip, libgemmgan_lab.so): 0 off, 1 on (4-slot weight ring), 3 on (8-slot ring)."""
        if mode:
            L.load_lab()
        L.check(self.lib.gg_set_ffn2(self.h, int(mode)))

    def set_ffn_fused(self, on: bool):
        """Fused feed-forward block of round 3 (csrc/ffn.hip, libgemmgan_lab.so)."""
        if on:
            L.load_lab()
        L.check(self.lib.gg_set_ffn_fused(self.h, int(on)))

    def set_tlin(self, on: bool):
        L.check(self.lib.gg_set_tlin(self.h, int(on)))

    def set_seed(self, seed):
        L.check(self.lib.gg_set_seed(self.h, C.c_uint64(seed)))

    def debug_buffer(self, name: str) -> torch.Tensor:
        """Copy of a named internal activation buffer of the last call (tests only)."""
        ptr, n = C.c_void_p(), C.c_int64()
        L.check(self.lib.gg_debug_buffer(self.h, name.encode(), C.byref(ptr), C.byref(n)))
        is_bf16 = self.lib.gg_debug_buffer_is_bf16(self.h, name.encode()) == 1
        out = torch.empty(n.value, dtype=torch.bfloat16 if is_bf16 else torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        rt = C.CDLL("libamdhip64.so")
        rc = rt.hipMemcpy(C.c_void_p(out.data_ptr()), ptr, C.c_size_t(out.element_size() * n.value), C.c_int(3))   # DeviceToDevice
        if rc != 0:
            raise RuntimeError(f"hipMemcpy failed: {rc}")
        return out.float()

    PROFILE_CLASSES = ["gemm_f32_kernel<KC,KC>", "gemm_f32_kernel<KC,KS>", "gemm_f32_kernel<KS,KC>", "gemm_f32_kernel<KS,KS>",
                       "gemm_bf16_kernel<KC,KC>", "gemm_bf16_kernel<KC,KS>", "gemm_bf16_kernel<KS,KC>", "gemm_bf16_kernel<KS,KS>",
                       "tlin_str_kernel", "tlin_res_kernel", "wgrad_kernel<true,false,false,false>", "gemm_small_kernel",
                       "tlin_res16_kernel<8,256,true,1>", "tlin_res16_kernel<8,256,true,2>", "tlin_res16_kernel<8,256,true,0|3>",
                       "wgrad_kernel<true,true,false,false>", "wgrad_kernel<false,false,false,false>",
                       "wgrad_kernel<false,false,true,false>"]

    def profile(self, on: bool, classes=None):
        """HIP-event pair around every GEMM-class launch (classes=None) or only around the named classes."""
        if on and classes:
            L.check(self.lib.gg_profile_enable_class(self.h, classes[0].encode()))      # names as profile_collect reports them
            for name in classes[1:]:
                L.check(self.lib.gg_profile_add_class(self.h, name.encode()))
            return
        arg = int(bool(on))
        L.check(self.lib.gg_profile_enable(self.h, arg))

    def profile_pause(self):
        """Stop taking event pairs but keep the records (profile(False) followed by a collect would do the same; this
        form does not clear them)."""
        L.check(self.lib.gg_profile_enable(self.h, -1))

    def profile_collect(self):
        """[{name, launches, ms, flops, bytes}] per kernel class since profile(True)."""
        n = self.lib.gg_profile_collect(self.h)
        if n < 0:
            L.check(n)
        rows = []
        self._prof_ids = {}
        for i in range(n):
            name = C.create_string_buffer(128)
            launches, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            L.check(self.lib.gg_profile_read(self.h, i, name, 128, C.byref(launches), C.byref(ms), C.byref(fl), C.byref(by)))
            rows.append(dict(name=name.value.decode(), launches=launches.value, ms=ms.value, flops=fl.value, bytes=by.value))
            self._prof_ids[rows[-1]["name"]] = i
        return rows

    def gp_profile(self, B, reps=20):
        """[{kernel, us, bytes}] of the four gradient-penalty kernels (buffers of the last critic iteration)."""
        us, by = (C.c_double * 4)(), (C.c_double * 4)()
        L.check(self.lib.gg_gp_profile(self.h, int(B), int(reps), us, by, _stream()))
        return [dict(kernel=n, us=us[i], bytes=by[i]) for i, n in enumerate(("gp_front_k", "gp_grad_k", "gp_coef_k", "gp_tail_k"))]

    def optimizer_step(self, role):
        return int(self.lib.gg_get_optimizer_step(self.h, role))

    def launch_count(self):
        return int(self.lib.gg_launch_count(self.h))

    def reset_launch_count(self):
        L.check(self.lib.gg_reset_launch_count(self.h))
