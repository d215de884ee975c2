"""CPU oracle #1 (torch, autograd) for the GeMM-GAN conditional WGAN-GP hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gemm_gan_amd/`` may import this file; it
is the checker used by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  It never runs on the product path.

What it restates (all citations are into /root/reference/src/
conditional_gan_cross_attention_with_film.py unless another file is named):

* ``CondNet``            - ``generator`` (:97-164) / ``discriminator`` (:167-233): FiLM from the
                            text CLS token (:129-137), Linear text/patch encoders (:139-140),
                            learnable CLS + 2x post-norm TransformerEncoderLayer (:142-144),
                            single-query patch<-text-CLS attention then text<-that attention
                            (:149-152), sum (:155), concat with z / genes (:157), MLP (:159-162).
* ``critic_losses``      - ``D_loss`` (:41-46), ``gradient_penalty`` (:351-374).
* ``Trainer``            - ``init_train`` (:320-331), ``train_disc`` (:376-423),
                            ``train_gen`` (:425-461), ``train`` (:463-477), with z / alpha
                            supplied explicitly so that a step is a pure function of its inputs.

Sibling files (``PathConfig.variant``, each pinned by its own fixtures): ``film`` = src/conditional_gan_film.py
(``film_config``), ``img`` = src/conditional_gan_img_transformer.py (``img_config``), ``vanilla`` =
src/vanilla_gan_unconditional.py (``vanilla_config``).

The arithmetic of the reference lives in stock ``torch.nn`` modules;
the oracle therefore composes the same stock modules under the same attribute names, so a
reference ``state_dict`` loads with ``strict=True`` (including the dead
``patches_transformer_layer.*`` copy, :114).  Parity of this file against the real reference is
pinned by ``tests/golden/*.npz`` (made by ``oracle/make_golden.py`` importing the reference in
the build container) - see ``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn


@dataclasses.dataclass
class PathConfig:
    """Shape/hyper-parameter bundle of one hot-path configuration (SURVEY.md conventions)."""
    n_genes: int = 5000          # G
    latent_dims: int = 256       # L
    embedding_dims: int = 256    # E
    hidden_dims: int = 256       # H  (generator_dims=[H,H,G], discriminator_dims=[H,H,1], :940-951)
    text_dims: int = 512         # Dt (reference default 768, :262)
    patch_dims: int = 1024       # Dp
    n_heads: int = 4             # :115, :121
    n_layers: int = 2            # :119
    negative_slope: float = 0.0  # :945
    dropout: float = 0.1         # :116 (encoder layers only; the two cross attentions use 0.0)
    lr_d: float = 5e-4
    lr_g: float = 5e-4
    optimizer: str = "rms_prop"
    gp_weight: float = 10.0
    n_critic: int = 5
    clip_d: Optional[float] = 10.0   # :414
    clip_g: Optional[float] = 2.0    # :457
    # "xattn_film": src/conditional_gan_cross_attention_with_film.py (line numbers above and below);
    # "film": src/conditional_gan_film.py - no token encoder / cross attention, the conditioning vector is the encoder's
    # CLS row (F:150), bias-free encoder layers (F:115 bias=False), no gradient clipping (F:383-385): use
    # film_config() so the clip fields follow
    # "img": src/conditional_gan_img_transformer.py (I:95-190) - like "film" but without FiLM: the patch encoder is
    # Linear -> ReLU -> LayerNorm on the raw patches (I:106-110), the text input is unused
    # "vanilla": src/vanilla_gan_unconditional.py (V:93-184) - MLP generator / critic, no conditioning inputs at all
    variant: str = "xattn_film"

    @property
    def ffn_dims(self) -> int:
        return 2 * self.embedding_dims   # dim_feedforward=E*2, :115


def film_config(**kw) -> PathConfig:
    """PathConfig of the FiLM-only sibling (src/conditional_gan_film.py): no clipping, reference default text width."""
    kw.setdefault("clip_d", None)
    kw.setdefault("clip_g", None)
    return PathConfig(variant="film", **kw)


def img_config(**kw) -> PathConfig:
    """PathConfig of the image-transformer sibling (src/conditional_gan_img_transformer.py): no FiLM, no clipping."""
    kw.setdefault("clip_d", None)
    kw.setdefault("clip_g", None)
    return PathConfig(variant="img", **kw)


def vanilla_config(**kw) -> PathConfig:
    """PathConfig of the unconditional model (src/vanilla_gan_unconditional.py): no conditioning, no dropout, no clipping."""
    kw.setdefault("clip_d", None)
    kw.setdefault("clip_g", None)
    kw.setdefault("dropout", 0.0)
    return PathConfig(variant="vanilla", **kw)


def _mlp_block(n_in: int, n_out: int, slope: float) -> nn.Sequential:
    # build_linear_block (:56-74) with is_bn=False: Linear followed by LeakyReLU(slope)
    return nn.Sequential(nn.Linear(n_in, n_out), nn.LeakyReLU(negative_slope=slope))


class CondNet(nn.Module):
    """Generator (role='generator') or critic (role='discriminator') of the xattn+FiLM model.

    Attribute creation order follows the reference constructors (:111-126 / :180-195) so that
    ``torch.manual_seed(s)`` followed by construction reproduces the reference initialisation.
    """

    def __init__(self, role: str, cfg: PathConfig):
        super().__init__()
        assert role in ("generator", "discriminator")
        self.role, self.cfg = role, cfg
        E, Dt, Dp = cfg.embedding_dims, cfg.text_dims, cfg.patch_dims
        assert cfg.variant in ("xattn_film", "film", "img", "vanilla")
        if cfg.variant == "vanilla":                      # V:110-113 / V:158-161: the MLP on the raw vector
            first = cfg.latent_dims if role == "generator" else cfg.n_genes
            H = cfg.hidden_dims
            setattr(self, role, nn.ModuleList([_mlp_block(first, H, cfg.negative_slope), _mlp_block(H, H, cfg.negative_slope)]))
            self.final_layer = nn.Linear(H, cfg.n_genes if role == "generator" else 1)
            return
        film_only = cfg.variant in ("film", "img")       # no token encoder / cross attention, bias-free encoder
        if cfg.variant != "img":
            self.film_generator = nn.Linear(Dt, 2 * Dp)
        if not film_only:
            self.text_encoder = nn.Linear(Dt, E)
        if cfg.variant == "img":                          # I:106-110
            self.patches_encoder = nn.Sequential(nn.Linear(Dp, E), nn.ReLU(), nn.LayerNorm(E))
        else:
            self.patches_encoder = nn.Linear(Dp, E)
        self.patches_transformer_layer = nn.TransformerEncoderLayer(
            d_model=E, nhead=cfg.n_heads, dim_feedforward=cfg.ffn_dims, dropout=cfg.dropout,
            activation="relu", batch_first=True, bias=not film_only)      # F:113-115: bias=False
        self.patches_cls_token = nn.Parameter(torch.empty(1, 1, E))
        nn.init.trunc_normal_(self.patches_cls_token, std=0.02)
        self.patches_transformer = nn.TransformerEncoder(
            self.patches_transformer_layer, num_layers=cfg.n_layers)
        if not film_only:
            self.patch2text_attention = nn.MultiheadAttention(E, cfg.n_heads, batch_first=True)
            self.text2patch_attention = nn.MultiheadAttention(E, cfg.n_heads, batch_first=True)
        first = (cfg.latent_dims if role == "generator" else cfg.n_genes) + E
        H = cfg.hidden_dims
        blocks = nn.ModuleList([_mlp_block(first, H, cfg.negative_slope),
                                _mlp_block(H, H, cfg.negative_slope)])
        setattr(self, role, blocks)     # attribute is literally 'generator' / 'discriminator'
        self.final_layer = nn.Linear(H, cfg.n_genes if role == "generator" else 1)

    # -- conditioning stack shared by both roles (:129-155 == :198-224) ------------------------
    def conditioning(self, patches, patch_pad, text, text_pad, taps: Optional[dict] = None):
        if self.cfg.variant == "vanilla":
            return None
        Dp = self.cfg.patch_dims
        if self.cfg.variant == "img":                     # I:126: the encoder sees the raw patches; text is unused
            gamma = beta = None
            emb = self.patches_encoder(patches)
        else:
            gb = self.film_generator(text[:, 0, :])
            gamma = torch.tanh(gb[:, :Dp])
            beta = torch.clamp(gb[:, Dp:], min=-5.0, max=5.0)
            mod = gamma[:, None, :] * patches + beta[:, None, :]
            emb = self.patches_encoder(mod)
        B = emb.shape[0]
        seq = torch.cat((self.patches_cls_token.expand(B, -1, -1), emb), dim=1)
        mask = torch.cat((patch_pad.new_zeros(B, 1, dtype=torch.bool), patch_pad), dim=1)
        enc = self.patches_transformer(seq, src_key_padding_mask=mask)
        if self.cfg.variant in ("film", "img"):      # F:130-152 / I:124-136: text is one vector per sample (here [B,1,Dt]); c = CLS row
            c = enc[:, 0, :]
            if taps is not None:
                taps.update(gamma=gamma, beta=beta, seq0=seq, enc=enc, cond=c)
            return c
        tok = self.text_encoder(text)
        p, _ = self.patch2text_attention(tok[:, 0:1, :], enc, enc, key_padding_mask=mask)
        t, _ = self.text2patch_attention(p[:, 0:1, :], tok, tok, key_padding_mask=text_pad)
        c = t[:, 0, :] + p[:, 0, :]
        if taps is not None:
            taps.update(gamma=gamma, beta=beta, text_enc=tok, seq0=seq, enc=enc,
                        t2i=p[:, 0, :], i2t=t[:, 0, :], cond=c)
        return c

    def head(self, v, c, taps: Optional[dict] = None):
        h = v if c is None else torch.cat((v, c), dim=1)
        for i, blk in enumerate(getattr(self, self.role)):
            pre = blk[0](h)
            h = blk[1](pre)
            if taps is not None:
                taps[f"mlp_pre{i}"] = pre
        return self.final_layer(h)

    def forward(self, v, patches, patch_pad, text, text_pad, taps: Optional[dict] = None):
        return self.head(v, self.conditioning(patches, patch_pad, text, text_pad, taps), taps)


def live_parameters(net: CondNet) -> List[Tuple[str, nn.Parameter]]:
    """Parameters that take part in forward (everything except the dead template layer, :114)."""
    return [(n, p) for n, p in net.named_parameters()
            if not n.startswith("patches_transformer_layer.")]


def set_dropout(net: nn.Module, p: float) -> None:
    """Set every dropout site of the encoder stack (4 per layer incl. attention-prob dropout)."""
    for m in net.modules():
        if isinstance(m, nn.Dropout):
            m.p = p
        if isinstance(m, nn.MultiheadAttention) and any(
                m is l.self_attn for l in _encoder_layers(net)):
            m.dropout = p


def _encoder_layers(net):
    out = []
    for m in net.modules():
        if isinstance(m, nn.TransformerEncoderLayer):
            out.append(m)
    return out


def critic_losses(disc: CondNet, x_real, x_fake, alpha, cond, gp_weight: float):
    """D_loss (:41-46) + gp_weight * gradient_penalty (:351-374), alpha [B,1] supplied."""
    patches, patch_pad, text, text_pad = cond
    d_fake = disc(x_fake, patches, patch_pad, text, text_pad)
    d_true = disc(x_real, patches, patch_pad, text, text_pad)
    loss_real = torch.mean(-d_true)
    loss_fake = torch.mean(d_fake)
    a = alpha.detach().clone().requires_grad_(True)
    x_hat = a * x_real + (1 - a) * x_fake
    out = disc(x_hat, patches, patch_pad, text, text_pad)
    grad = torch.autograd.grad(out, x_hat, torch.ones_like(out),
                               create_graph=True, retain_graph=True)[0]
    nrm = grad.reshape(grad.shape[0], -1).norm(2, 1)
    gp = torch.mean((nrm - 1) ** 2)
    total = loss_real + loss_fake + gp_weight * gp
    return dict(total=total, d_loss=loss_real + loss_fake, d_real=loss_real, d_fake=loss_fake,
                gp=gp, grad_x_hat=grad, grad_norm=nrm, d_fake_out=d_fake, d_true_out=d_true)


def _make_opt(name: str, params, lr: float):
    name = name.lower()
    if name == "rms_prop":
        return torch.optim.RMSprop(params, lr=lr)
    if name == "adam":
        return torch.optim.Adam(params, lr=lr, betas=(0.9, 0.99))
    if name == "adamw":
        return torch.optim.AdamW(params, lr=lr, betas=(0.9, 0.99), weight_decay=0.01)
    raise ValueError(name)


class Trainer:
    """One WGAN-GP ``train()`` (:463-477) as a pure function of (inputs, z list, alpha list)."""

    def __init__(self, cfg: PathConfig, gen: Optional[CondNet] = None,
                 disc: Optional[CondNet] = None):
        self.cfg = cfg
        self.gen = gen if gen is not None else CondNet("generator", cfg)
        self.disc = disc if disc is not None else CondNet("discriminator", cfg)
        self.opt_d = _make_opt(cfg.optimizer, self.disc.parameters(), cfg.lr_d)
        self.opt_g = _make_opt(cfg.optimizer, self.gen.parameters(), cfg.lr_g)
        self.last: Dict[str, object] = {}

    def critic_iteration(self, x_real, z, alpha, cond, apply: bool = True):
        cfg = self.cfg
        self.disc.train()
        self.opt_d.zero_grad()
        for w in self.disc.parameters():
            w.requires_grad = True
        for w in self.gen.parameters():
            w.requires_grad = False
        patches, patch_pad, text, text_pad = cond
        x_fake = self.gen(z, patches, patch_pad, text, text_pad)
        res = critic_losses(self.disc, x_real, x_fake, alpha, cond, cfg.gp_weight)
        res["total"].backward()
        res["x_fake"] = x_fake.detach()
        res["grads"] = {n: (p.grad.detach().clone() if p.grad is not None else None)
                        for n, p in self.disc.named_parameters()}
        if cfg.clip_d is not None:
            res["grad_norm_total"] = torch.nn.utils.clip_grad_norm_(
                self.disc.parameters(), max_norm=cfg.clip_d).detach()
        if apply:
            self.opt_d.step()
        return res

    def generator_iteration(self, z, cond, apply: bool = True):
        cfg = self.cfg
        self.gen.train()
        self.opt_g.zero_grad()
        for w in self.disc.parameters():
            w.requires_grad = False
        for w in self.gen.parameters():
            w.requires_grad = True
        patches, patch_pad, text, text_pad = cond
        x_fake = self.gen(z, patches, patch_pad, text, text_pad)
        d_fake = self.disc(x_fake, patches, patch_pad, text, text_pad)
        g_loss = torch.mean(-d_fake)
        g_loss.backward()
        res = dict(g_loss=g_loss.detach(), x_fake=x_fake.detach(),
                   grads={n: (p.grad.detach().clone() if p.grad is not None else None)
                          for n, p in self.gen.named_parameters()})
        if cfg.clip_g is not None:
            res["grad_norm_total"] = torch.nn.utils.clip_grad_norm_(
                self.gen.parameters(), max_norm=cfg.clip_g).detach()
        if apply:
            self.opt_g.step()
        return res

    def train_step(self, x_real, text, text_pad, patches, patch_pad,
                   z_list: Sequence[torch.Tensor], alpha_list: Sequence[torch.Tensor]):
        """z_list: n_critic+1 tensors [B,L]; alpha_list: n_critic tensors [B,1]."""
        cond = (patches, patch_pad, text, text_pad)
        x_real = x_real.to(torch.float32)
        out = {}
        for k in range(self.cfg.n_critic):
            r = self.critic_iteration(x_real, z_list[k], alpha_list[k], cond)
            out["critic"] = r
        out["gen"] = self.generator_iteration(z_list[self.cfg.n_critic], cond)
        self.last = out
        return out

    @torch.no_grad()
    def generate(self, z, cond):
        self.gen.eval()
        patches, patch_pad, text, text_pad = cond
        return self.gen(z, patches, patch_pad, text, text_pad)


def synthetic_batch(cfg: PathConfig, B: int, P: int, T: int, seed: int = 42,
                    pad_patches: bool = False, pad_text: bool = False, device="cpu"):
    """Synthetic inputs of SURVEY.md section 8d: N(0,1) genes / patches / text, bool masks (True=pad)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(B, cfg.n_genes, generator=g)
    patches = torch.randn(B, P, cfg.patch_dims, generator=g)
    text = torch.randn(B, T, cfg.text_dims, generator=g)
    patch_pad = torch.zeros(B, P, dtype=torch.bool)
    text_pad = torch.zeros(B, T, dtype=torch.bool)
    if pad_patches and P > 1:
        for b in range(0, B, 4):                      # 25 % of rows: last quarter (>=1) padded
            patch_pad[b, P - max(1, P // 4):] = True
    if pad_text and T > 1:
        for b in range(1, B, 3):
            text_pad[b, T - max(1, T // 3):] = True   # token 0 (CLS) is never padded
    return tuple(t.to(device) for t in (x, text, text_pad, patches, patch_pad))
