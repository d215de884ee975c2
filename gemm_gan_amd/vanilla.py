"""Unconditional sibling: the reference's ``src/vanilla_gan_unconditional.py`` (V below; BASELINE ``configs[0]``) behind the
same HIP engine (``GG_VARIANT_VANILLA``).  Same names and call signatures as that file:

    generator_nocond (V:135) / discriminator_nocond (V:93): forward(x)
    WGAN_GP_model_nocond (V:186), WGAN_GP_nocond (V:211): build_WGAN_GP_nocond (V:291), init_train (V:276),
    train_disc(x, z) (V:330), train_gen(z) (V:383), train(x_GE) (V:421), generate_samples(x_GE) (V:462),
    fit (V:543; training loop, LR schedule, checkpoints)

The engine's entry points still take conditioning tensors; this module feeds them one dummy patch / text row per sample, and
the engine returns before touching them (its conditioning vector is identically zero for this variant).  The first-layer
weights live in the engine with ``embedding_dims`` extra zero columns; the modules here expose the reference's ``[H, V]``
block (a strided view), so ``state_dict`` keys AND shapes are the reference's.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import model as _m

_E = 8          # width of the (zero) conditioning vector the engine carries: the smallest that divides by 4 heads


class _NoCondNet(nn.Module):
    _role = None

    def __init__(self, first_dims, numerical_dims, vocab_sizes, mlp_dims, negative_slope=0.0, is_bn=False):
        super().__init__()
        if is_bn:
            raise NotImplementedError("is_bn=True is not part of the accelerated hot path (never enabled upstream)")
        self.numerical_dims = len(numerical_dims)
        self.vocab_sizes = vocab_sizes
        self.negative_slope = negative_slope
        self.n_cat_vars = len(vocab_sizes)
        self.input_dims = first_dims
        dims = list(mlp_dims)
        blocks = nn.ModuleList()
        prev = first_dims
        for d in dims[:-1]:
            blocks.append(_m._block(prev, d, negative_slope))
            prev = d
        if len(blocks) != 2 or dims[0] != dims[1]:
            raise NotImplementedError("the engine implements the reference's [H, H, out] MLP heads")
        setattr(self, self._role, blocks)
        self.final_layer = nn.Linear(dims[-2], dims[-1])
        self._engine = None
        self._engine_role = L.ROLE_GENERATOR if self._role == "generator" else L.ROLE_CRITIC

    def _bind(self, engine):
        self._engine = engine
        params = dict(self.named_parameters())
        with torch.no_grad():
            for name in engine.layout[self._engine_role]:
                view = engine.view(self._engine_role, name)        # first-layer weight: the [:, :V] block of the padded matrix
                view.copy_(params[name].detach().to(view.device, torch.float32))
                params[name].data = view
        return self

    def forward(self, x):
        if self._engine is None:
            raise RuntimeError("network is not bound to a HIP engine; build it through WGAN_GP_nocond.build_WGAN_GP_nocond() "
                               "(there is no torch/CPU fallback)")
        owner = getattr(self._engine, "_owner", None)
        if owner is not None:
            owner._ensure_capacity(x.shape[0], 1, 1)
        eng = owner.engine if owner is not None else self._engine
        pat, ppad, text, tpad = _dummies(eng, x.shape[0])
        with torch.no_grad():
            return eng.forward(self._engine_role, x.to(eng.device, torch.float32).contiguous(), pat, ppad, text, tpad,
                               train=self.training)


def _dummies(eng, B):
    mask = eng.zeros(B, 1, dtype=torch.bool)
    return eng.zeros(B, 1, eng.cfg.patch_dims), mask, eng.zeros(B, 1, eng.cfg.text_dims), mask


class generator_nocond(_NoCondNet):
    _role = "generator"

    def __init__(self, latent_dims, numerical_dims, vocab_sizes, generator_dims, negative_slope=0.0, is_bn=False):
        super().__init__(latent_dims, numerical_dims, vocab_sizes, generator_dims, negative_slope, is_bn)
        self.latent_dims = latent_dims
        self.generator_dims = generator_dims


class discriminator_nocond(_NoCondNet):
    _role = "discriminator"

    def __init__(self, vector_dims, numerical_dims, vocab_sizes, discriminator_dims, negative_slope=0.0, is_bn=False):
        super().__init__(vector_dims, numerical_dims, vocab_sizes, discriminator_dims, negative_slope, is_bn)
        self.vector_dims = vector_dims
        self.discriminator_dims = discriminator_dims


def WGAN_GP_model_nocond(latent_dims, vector_dims, numerical_dims, vocab_sizes, generator_dims, discriminator_dims,
                         negative_slope=0.0, is_bn=False):
    gen = generator_nocond(latent_dims, numerical_dims, vocab_sizes, generator_dims, negative_slope, is_bn)
    disc = discriminator_nocond(vector_dims, numerical_dims, vocab_sizes, discriminator_dims, negative_slope, is_bn)
    return gen, disc


class WGAN_GP_nocond(_m.WGAN_GP):
    _variant = "vanilla"
    _clip = (0.0, 0.0)            # V:330-420: optimiser steps without clip_grad_norm_

    def __init__(self, input_dims, latent_dims, vocab_sizes, generator_dims, discriminator_dims, negative_slope=0.0,
                 is_bn=False, numerical_dims=(), lr_d=5e-4, lr_g=5e-4, optimizer="rms_prop", gp_weight=10, p_aug=0,
                 norm_scale=0.5, train=True, n_critic=5, freq_print=2, freq_compute_test=10, freq_visualize_test=100,
                 patience=10, normalization="standardize", log2=False, rpm=False, results_dire="",
                 seed=0, device=None, process_group=None, precision="bf16x3"):
        super().__init__(input_dims, latent_dims, _E, generator_dims, discriminator_dims, text_embedding_dims=8,
                         patches_embedding_dims=8, negative_slope=negative_slope, is_bn=is_bn, lr_d=lr_d, lr_g=lr_g,
                         optimizer=optimizer, gp_weight=gp_weight, p_aug=p_aug, norm_scale=norm_scale, train=train,
                         n_critic=n_critic, freq_print=freq_print, freq_compute_test=freq_compute_test,
                         freq_visualize_test=freq_visualize_test, patience=patience, normalization=normalization, log2=log2,
                         rpm=rpm, results_dire=results_dire, dropout=0.0, seed=seed, device=device,
                         process_group=process_group, precision=precision)
        self.vocab_sizes = vocab_sizes
        self.numerical_dims = list(numerical_dims)

    def _build_nets(self):
        return WGAN_GP_model_nocond(self.latent_dims, self.input_dims, [], self.vocab_sizes, self.generator_dims,
                                    self.discriminator_dims, self.negative_slope, self.is_bn)

    def build_WGAN_GP_nocond(self):
        self.build_WGAN_GP()

    def gradient_penalty(self, real_data, fake_data):
        """V:304-327: the penalty value (0-d tensor, no autograd graph; alpha drawn by the same torch.rand call)."""
        text, tpad, pat, ppad = self._cond(real_data.shape[0])
        return super().gradient_penalty(real_data, fake_data, pat, ppad, text, tpad)

    def generate_samples_all(self, data):
        """V:433-459: (all_real_x, all_gen_x) over a loader whose items carry the expression matrix first."""
        all_real, all_gen = [], []
        for item in data:
            x_real, x_gen = self.generate_samples((item[0] if isinstance(item, (tuple, list)) else item).to(self.device))
            all_real.append(x_real.cpu().detach().numpy())
            all_gen.append(x_gen.cpu().detach().numpy())
        return np.vstack(all_real), np.vstack(all_gen)

    def _cond(self, B):
        self._ensure_capacity(B, 1, 1)
        pat, ppad, text, tpad = _dummies(self.engine, B)
        return text, tpad, pat, ppad

    def train_disc(self, x, z):
        return super().train_disc(x, z, *self._cond(z.shape[0]))

    def train_gen(self, z):
        return super().train_gen(z, *self._cond(z.shape[0]))

    def train(self, x_GE):
        return super().train(x_GE, *self._cond(x_GE.shape[0]))

    def train_with_explicit_noise(self, x, z_all, alpha_all, sync_losses=True):
        """One train() with explicit z [n_critic+1,B,L] / alpha [n_critic,B] (parity tests)."""
        text, tpad, pat, ppad = self._cond(x.shape[0])
        return self.train_with_noise(x.to(self.device, torch.float32).contiguous(), text, tpad, pat, ppad, z_all, alpha_all, sync_losses)

    def generate_samples(self, x_GE):
        with torch.no_grad():
            self.gen.eval()
            x_real = x_GE.clone().to(torch.float32)
            z = torch.normal(0, 1, size=(x_real.shape[0], self.latent_dims), device=self.device)
            x_gen = self.gen(z)
        return x_real, x_gen

    def _fit_batch(self, data, nxt=None):
        self.train(data[0] if isinstance(data, (tuple, list)) else data)     # V:576-579: data[0] is the expression matrix
