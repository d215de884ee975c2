"""CPU oracle #2 (numpy, hand-derived backward) for the GeMM-GAN WGAN-GP hot path.

TEST INFRASTRUCTURE ONLY - see the header of ``oracle/torch_oracle.py``.  Never imported by the
product path.

This file restates the SAME algorithm without autograd: every forward stage and its backward are
written out as plain matrix algebra (float64 by default), in exactly the decomposition the HIP
kernels implement.  It is the blueprint for ``gemm_gan_amd/csrc`` and the checker at shapes where
autograd on CPU would take minutes.  It is pinned (tests/test_numpy_oracle.py) against
  (1) the golden vectors of the real reference (tests/golden/*.npz), and
  (2) oracle #1 (torch autograd) on random shapes.

Citations: /root/reference/src/conditional_gan_cross_attention_with_film.py (``R:``) and the
installed torch (``T:`` = torch/nn/...).

  film                R:198-206     tanh / clamp(-5,5) FiLM from the text CLS token
  encoder layer       T:modules/transformer.py:940-983 (post-norm, ReLU FFN), MHA
                      T:functional.py:6206-6660 (packed in-proj, key-padding -> -inf, 1/sqrt(dh))
  single-query MHA    R:218-221
  head (MLP)          R:226-231, LeakyReLU(slope) R:56-74
  gradient penalty    R:351-374, closed form of SURVEY.md section 3.3
  losses              R:32-46
  clip + optimisers   R:414, R:457, R:320-331  (torch.optim RMSprop / Adam / AdamW defaults)
"""
from __future__ import annotations

import numpy as np

LN_EPS = 1e-5


# ------------------------------------------------------------------------------------------------
# parameter container: plain dict name -> ndarray using the reference state_dict keys
# ------------------------------------------------------------------------------------------------
def params_from_state(sd, dtype=np.float64):
    return {k: np.asarray(v, dtype=dtype).copy() for k, v in sd.items()}


def live_names(params, role):
    return [k for k in params if not k.startswith("patches_transformer_layer.")]


def zeros_like_params(params, role):
    return {k: np.zeros_like(params[k]) for k in live_names(params, role)}


# ------------------------------------------------------------------------------------------------
# primitive stages (forward returns (out, cache); backward consumes the cache)
# ------------------------------------------------------------------------------------------------
def linear(x, W, b=None):
    y = x @ W.T
    return y if b is None else y + b


def layernorm_fwd(x, g, b):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + LN_EPS)
    xh = (x - mu) * rstd
    return xh * g + b, (xh, rstd)


def layernorm_bwd(dy, g, cache):
    xh, rstd = cache
    dxh = dy * g
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xh * (dxh * xh).mean(-1, keepdims=True))
    red = tuple(range(dy.ndim - 1))
    return dx, (dy * xh).sum(red), dy.sum(red)


def _split_heads(x, nh):          # [B,S,E] -> [B,nh,S,dh]
    B, S, E = x.shape
    return x.reshape(B, S, nh, E // nh).transpose(0, 2, 1, 3)


def _merge_heads(x):              # [B,nh,S,dh] -> [B,S,E]
    B, nh, S, dh = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, S, nh * dh)


def _softmax_masked(scores, key_pad):
    """scores [B,nh,Sq,Sk]; key_pad [B,Sk] bool (True = ignore)."""
    s = np.where(key_pad[:, None, None, :], -np.inf, scores)
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    return e / e.sum(-1, keepdims=True)


def mha_fwd(q_in, kv_in, key_pad, Win, bin_, Wo, bo, nh, drop_mask=None, drop_scale=1.0):
    """torch MultiheadAttention(batch_first) with key == value.  q_in [B,Sq,E], kv_in [B,Sk,E]."""
    E = q_in.shape[-1]
    dh = E // nh
    scale = 1.0 / np.sqrt(dh)
    q = _split_heads(linear(q_in, Win[:E], bin_[:E]), nh)
    k = _split_heads(linear(kv_in, Win[E:2 * E], bin_[E:2 * E]), nh)
    v = _split_heads(linear(kv_in, Win[2 * E:], bin_[2 * E:]), nh)
    P = _softmax_masked(scale * (q @ k.transpose(0, 1, 3, 2)), key_pad)
    Pd = P if drop_mask is None else P * drop_mask * drop_scale
    ctx = _merge_heads(Pd @ v)
    out = linear(ctx, Wo, bo)
    return out, (q_in, kv_in, q, k, v, P, Pd, ctx, scale, drop_mask, drop_scale)


def mha_bwd(dout, Win, Wo, nh, cache):
    q_in, kv_in, q, k, v, P, Pd, ctx, scale, drop_mask, drop_scale = cache
    E = q_in.shape[-1]
    g = {}
    g["out_proj.weight"] = np.einsum("bse,bsf->ef", dout, ctx)
    g["out_proj.bias"] = dout.sum((0, 1))
    dctx = _split_heads(dout @ Wo, nh)
    dPd = dctx @ v.transpose(0, 1, 3, 2)
    dv = Pd.transpose(0, 1, 3, 2) @ dctx
    dP = dPd if drop_mask is None else dPd * drop_mask * drop_scale
    dS = P * (dP - (dP * P).sum(-1, keepdims=True))
    dq = _merge_heads(scale * (dS @ k))
    dk = _merge_heads(scale * (dS.transpose(0, 1, 3, 2) @ q))
    dv = _merge_heads(dv)
    gW = np.zeros_like(Win)
    gb = np.zeros(3 * E, dtype=Win.dtype)
    gW[:E] = np.einsum("bse,bsf->ef", dq, q_in)
    gW[E:2 * E] = np.einsum("bse,bsf->ef", dk, kv_in)
    gW[2 * E:] = np.einsum("bse,bsf->ef", dv, kv_in)
    gb[:E], gb[E:2 * E], gb[2 * E:] = dq.sum((0, 1)), dk.sum((0, 1)), dv.sum((0, 1))
    g["in_proj_weight"], g["in_proj_bias"] = gW, gb
    dq_in = dq @ Win[:E]
    dkv_in = dk @ Win[E:2 * E] + dv @ Win[2 * E:]
    return dq_in, dkv_in, g


def encoder_layer_fwd(x, key_pad, p, pre, nh, drops=None):
    """Post-norm layer: x1 = LN1(x + SA(x)); x2 = LN2(x1 + W2 relu(W1 x1 + b1) + b2)."""
    d = drops or {}
    sa, c_sa = mha_fwd(x, x, key_pad, p[pre + "self_attn.in_proj_weight"], p[pre + "self_attn.in_proj_bias"],
                       p[pre + "self_attn.out_proj.weight"], p[pre + "self_attn.out_proj.bias"], nh,
                       d.get("attn"), d.get("scale", 1.0))
    if "post_sa" in d:
        sa = sa * d["post_sa"] * d["scale"]
    x1, c_n1 = layernorm_fwd(x + sa, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
    hpre = linear(x1, p[pre + "linear1.weight"], p[pre + "linear1.bias"])
    h = np.maximum(hpre, 0)
    if "ffn" in d:
        h = h * d["ffn"] * d["scale"]
    f = linear(h, p[pre + "linear2.weight"], p[pre + "linear2.bias"])
    if "post_ffn" in d:
        f = f * d["post_ffn"] * d["scale"]
    x2, c_n2 = layernorm_fwd(x1 + f, p[pre + "norm2.weight"], p[pre + "norm2.bias"])
    return x2, (c_sa, c_n1, x1, hpre, h, c_n2, d)


def encoder_layer_bwd(dx2, p, pre, nh, cache, g):
    c_sa, c_n1, x1, hpre, h, c_n2, d = cache
    dr, g[pre + "norm2.weight"], g[pre + "norm2.bias"] = layernorm_bwd(dx2, p[pre + "norm2.weight"], c_n2)
    df = dr if "post_ffn" not in d else dr * d["post_ffn"] * d["scale"]
    g[pre + "linear2.weight"] = np.einsum("bse,bsf->ef", df, h)
    g[pre + "linear2.bias"] = df.sum((0, 1))
    dh = df @ p[pre + "linear2.weight"]
    if "ffn" in d:
        dh = dh * d["ffn"] * d["scale"]
    dhpre = dh * (hpre > 0)
    g[pre + "linear1.weight"] = np.einsum("bse,bsf->ef", dhpre, x1)
    g[pre + "linear1.bias"] = dhpre.sum((0, 1))
    dx1 = dr + dhpre @ p[pre + "linear1.weight"]
    dr1, g[pre + "norm1.weight"], g[pre + "norm1.bias"] = layernorm_bwd(dx1, p[pre + "norm1.weight"], c_n1)
    dsa = dr1 if "post_sa" not in d else dr1 * d["post_sa"] * d["scale"]
    dq_in, dkv_in, ga = mha_bwd(dsa, p[pre + "self_attn.in_proj_weight"], p[pre + "self_attn.out_proj.weight"], nh, c_sa)
    for k, v in ga.items():
        g[pre + "self_attn." + k] = v
    return dr1 + dq_in + dkv_in


# ------------------------------------------------------------------------------------------------
# conditioning stack (R:198-224) and MLP head (R:226-231)
# ------------------------------------------------------------------------------------------------
def cond_fwd(p, patches, patch_pad, text, text_pad, nh=4, n_layers=2, drops=None):
    Dp = patches.shape[-1]
    B = patches.shape[0]
    text_cls = text[:, 0, :]
    gb = linear(text_cls, p["film_generator.weight"], p["film_generator.bias"])
    gamma = np.tanh(gb[:, :Dp])
    beta = np.clip(gb[:, Dp:], -5.0, 5.0)
    mod = gamma[:, None, :] * patches + beta[:, None, :]
    tok = linear(text, p["text_encoder.weight"], p["text_encoder.bias"])
    emb = linear(mod, p["patches_encoder.weight"], p["patches_encoder.bias"])
    E = emb.shape[-1]
    seq = np.concatenate([np.broadcast_to(p["patches_cls_token"].reshape(1, 1, E), (B, 1, E)), emb], axis=1)
    mask = np.concatenate([np.zeros((B, 1), dtype=bool), patch_pad], axis=1)
    xs, caches = [seq], []
    for l in range(n_layers):
        y, c = encoder_layer_fwd(xs[-1], mask, p, f"patches_transformer.layers.{l}.", nh,
                                 None if drops is None else drops[l])
        xs.append(y)
        caches.append(c)
    enc = xs[-1]
    t2i, c_t2i = mha_fwd(tok[:, 0:1, :], enc, mask, p["patch2text_attention.in_proj_weight"],
                         p["patch2text_attention.in_proj_bias"], p["patch2text_attention.out_proj.weight"],
                         p["patch2text_attention.out_proj.bias"], nh)
    i2t, c_i2t = mha_fwd(t2i, tok, text_pad, p["text2patch_attention.in_proj_weight"],
                         p["text2patch_attention.in_proj_bias"], p["text2patch_attention.out_proj.weight"],
                         p["text2patch_attention.out_proj.bias"], nh)
    c = i2t[:, 0, :] + t2i[:, 0, :]
    cache = dict(text_cls=text_cls, gb=gb, gamma=gamma, beta=beta, patches=patches, mod=mod, text=text,
                 tok=tok, seq=seq, mask=mask, layers=caches, enc=enc, c_t2i=c_t2i, c_i2t=c_i2t, xs=xs,
                 t2i=t2i[:, 0, :], i2t=i2t[:, 0, :], nh=nh, n_layers=n_layers)
    return c, cache


def cond_bwd(dc, p, cache, g):
    """Accumulate d(loss)/d(params of the conditioning stack) into dict g (sets the entries)."""
    nh = cache["nh"]
    Dp = cache["patches"].shape[-1]
    dt = dc[:, None, :]
    dq_in, dtok, ga = mha_bwd(dt, p["text2patch_attention.in_proj_weight"], p["text2patch_attention.out_proj.weight"], nh, cache["c_i2t"])
    for k, v in ga.items():
        g["text2patch_attention." + k] = v
    dp_ = dc[:, None, :] + dq_in                       # c = t + p ; p also feeds the I2T query
    dq_tok, denc, ga = mha_bwd(dp_, p["patch2text_attention.in_proj_weight"], p["patch2text_attention.out_proj.weight"], nh, cache["c_t2i"])
    for k, v in ga.items():
        g["patch2text_attention." + k] = v
    dtok = dtok.copy()
    dtok[:, 0:1, :] += dq_tok
    dx = denc
    for l in reversed(range(cache["n_layers"])):
        dx = encoder_layer_bwd(dx, p, f"patches_transformer.layers.{l}.", nh, cache["layers"][l], g)
    g["patches_cls_token"] = dx[:, 0, :].sum(0).reshape(1, 1, -1)
    demb = dx[:, 1:, :]
    g["patches_encoder.weight"] = np.einsum("bpe,bpd->ed", demb, cache["mod"])
    g["patches_encoder.bias"] = demb.sum((0, 1))
    dmod = demb @ p["patches_encoder.weight"]
    g["text_encoder.weight"] = np.einsum("bte,btd->ed", dtok, cache["text"])
    g["text_encoder.bias"] = dtok.sum((0, 1))
    dgamma = (dmod * cache["patches"]).sum(1)
    dbeta = dmod.sum(1)
    gbeta_pre = cache["gb"][:, Dp:]
    dgb = np.concatenate([dgamma * (1 - cache["gamma"] ** 2),
                          dbeta * ((gbeta_pre >= -5.0) & (gbeta_pre <= 5.0))], axis=1)
    g["film_generator.weight"] = dgb.T @ cache["text_cls"]
    g["film_generator.bias"] = dgb.sum(0)
    return g


def _act(h, slope):
    return np.where(h > 0, h, slope * h)


def _dact(h, slope):
    return np.where(h > 0, 1.0, slope)


def head_fwd(p, role, v, c, slope):
    W1, b1 = p[f"{role}.0.0.weight"], p[f"{role}.0.0.bias"]
    W2, b2 = p[f"{role}.1.0.weight"], p[f"{role}.1.0.bias"]
    W3, b3 = p["final_layer.weight"], p["final_layer.bias"]
    inp = np.concatenate([v, c], axis=1)
    h1 = linear(inp, W1, b1)
    a1 = _act(h1, slope)
    h2 = linear(a1, W2, b2)
    a2 = _act(h2, slope)
    out = linear(a2, W3, b3)
    return out, dict(inp=inp, h1=h1, a1=a1, h2=h2, a2=a2, nv=v.shape[1])


def head_bwd(dout, p, role, slope, cache, g, accumulate=False):
    """Returns (dv, dc).  Sets (or adds to, accumulate=True) the five head gradients in g."""
    W1, W2, W3 = p[f"{role}.0.0.weight"], p[f"{role}.1.0.weight"], p["final_layer.weight"]
    def put(k, val):
        g[k] = g[k] + val if (accumulate and k in g) else val
    put("final_layer.weight", dout.T @ cache["a2"])
    put("final_layer.bias", dout.sum(0))
    dh2 = (dout @ W3) * _dact(cache["h2"], slope)
    put(f"{role}.1.0.weight", dh2.T @ cache["a1"])
    put(f"{role}.1.0.bias", dh2.sum(0))
    dh1 = (dh2 @ W2) * _dact(cache["h1"], slope)
    put(f"{role}.0.0.weight", dh1.T @ cache["inp"])
    put(f"{role}.0.0.bias", dh1.sum(0))
    dinp = dh1 @ W1
    nv = cache["nv"]
    return dinp[:, :nv], dinp[:, nv:]


# ------------------------------------------------------------------------------------------------
# gradient penalty, closed form (R:351-374)
# ------------------------------------------------------------------------------------------------
def gp_closed_form(p, x_hat, c, slope, gp_weight, g=None):
    """Returns (gp, grad_x_hat, grad_norm).  If g is given, ADDS gp_weight * d(gp)/d(W1x, W2, w3)."""
    role = "discriminator"
    B, G = x_hat.shape
    W1, W2, w3 = p[f"{role}.0.0.weight"], p[f"{role}.1.0.weight"], p["final_layer.weight"]
    W1x = W1[:, :G]
    _, ch = head_fwd(p, role, x_hat, c, slope)
    m1, m2 = _dact(ch["h1"], slope), _dact(ch["h2"], slope)
    g2 = m2 * w3                       # [B,H]
    u = g2 @ W2
    g1 = m1 * u
    grad = g1 @ W1x                    # == autograd.grad(D(x_hat), x_hat)
    nrm = np.sqrt((grad ** 2).sum(1))
    gp = ((nrm - 1.0) ** 2).mean()
    if g is not None:
        safe = np.where(nrm > 0, nrm, 1.0)
        s = (2.0 / B) * ((nrm - 1.0) / safe)[:, None] * grad * gp_weight
        gW1 = np.zeros_like(W1)
        gW1[:, :G] = g1.T @ s
        dg1 = s @ W1x.T
        du = m1 * dg1
        dg2 = du @ W2.T
        g[f"{role}.0.0.weight"] = g[f"{role}.0.0.weight"] + gW1
        g[f"{role}.1.0.weight"] = g[f"{role}.1.0.weight"] + g2.T @ du
        g["final_layer.weight"] = g["final_layer.weight"] + (m2 * dg2).sum(0, keepdims=True)
    return gp, grad, nrm


# ------------------------------------------------------------------------------------------------
# one critic / generator iteration, clip, optimisers, full step
# ------------------------------------------------------------------------------------------------
def net_forward(p, role, v, patches, patch_pad, text, text_pad, slope=0.0, nh=4, n_layers=2):
    c, _ = cond_fwd(p, patches, patch_pad, text, text_pad, nh, n_layers)
    out, _ = head_fwd(p, role, v, c, slope)
    return out


def critic_iteration_grads(pg, pd, x_real, z, alpha, cond, slope=0.0, gp_weight=10.0, nh=4, n_layers=2):
    """R:376-412 with dropout 0: returns (losses dict, grads dict over live critic params, x_fake)."""
    patches, patch_pad, text, text_pad = cond
    B = x_real.shape[0]
    x_fake = net_forward(pg, "generator", z, patches, patch_pad, text, text_pad, slope, nh, n_layers)
    c, cc = cond_fwd(pd, patches, patch_pad, text, text_pad, nh, n_layers)   # shared by 3 passes (p=0)
    d_fake, ch_f = head_fwd(pd, "discriminator", x_fake, c, slope)
    d_true, ch_t = head_fwd(pd, "discriminator", x_real, c, slope)
    loss_real, loss_fake = (-d_true).mean(), d_fake.mean()
    g = {}
    ones = np.ones_like(d_fake) / B
    _, dc_f = head_bwd(ones, pd, "discriminator", slope, ch_f, g)
    _, dc_t = head_bwd(-ones, pd, "discriminator", slope, ch_t, g, accumulate=True)
    x_hat = alpha * x_real + (1 - alpha) * x_fake
    gp, grad, nrm = gp_closed_form(pd, x_hat, c, slope, gp_weight, g)
    cond_bwd(dc_f + dc_t, pd, cc, g)
    losses = dict(total=loss_real + loss_fake + gp_weight * gp, d_loss=loss_real + loss_fake,
                  d_real=loss_real, d_fake=loss_fake, gp=gp)
    return losses, g, x_fake, grad


def generator_iteration_grads(pg, pd, z, cond, slope=0.0, nh=4, n_layers=2):
    """R:425-455 with dropout 0."""
    patches, patch_pad, text, text_pad = cond
    B = z.shape[0]
    cg, ccg = cond_fwd(pg, patches, patch_pad, text, text_pad, nh, n_layers)
    x_fake, chg = head_fwd(pg, "generator", z, cg, slope)
    cd, _ = cond_fwd(pd, patches, patch_pad, text, text_pad, nh, n_layers)
    d_fake, chd = head_fwd(pd, "discriminator", x_fake, cd, slope)
    g_loss = (-d_fake).mean()
    dx_fake, _ = head_bwd(-np.ones_like(d_fake) / B, pd, "discriminator", slope, chd, {})
    g = {}
    _, dcg = head_bwd(dx_fake, pg, "generator", slope, chg, g)
    cond_bwd(dcg, pg, ccg, g)
    return g_loss, g, x_fake


def clip_coef(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_: total 2-norm, coef = min(1, max_norm / (total + 1e-6))."""
    total = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in grads.values()))
    return total, min(1.0, max_norm / (total + 1e-6))


class Optim:
    """torch.optim RMSprop(alpha=.99, eps=1e-8) / Adam(betas=(.9,.99), eps=1e-8) / AdamW(wd=.01)."""

    def __init__(self, kind, lr):
        self.kind, self.lr, self.t, self.state = kind.lower(), lr, 0, {}

    def step(self, params, grads):
        self.t += 1
        for k, gr in grads.items():
            w = params[k]
            st = self.state.setdefault(k, {})
            if self.kind == "rms_prop":
                sq = st.get("sq", np.zeros_like(w))
                sq = 0.99 * sq + 0.01 * gr * gr
                st["sq"] = sq
                params[k] = w - self.lr * gr / (np.sqrt(sq) + 1e-8)
            else:
                b1, b2 = 0.9, 0.99
                if self.kind == "adamw":
                    w = w * (1 - self.lr * 0.01)
                m = b1 * st.get("m", np.zeros_like(w)) + (1 - b1) * gr
                v = b2 * st.get("v", np.zeros_like(w)) + (1 - b2) * gr * gr
                st["m"], st["v"] = m, v
                bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
                params[k] = w - (self.lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + 1e-8)


def train_step(pg, pd, opt_g, opt_d, x_real, cond, z_list, alpha_list, slope=0.0, gp_weight=10.0,
               clip_d=10.0, clip_g=2.0, nh=4, n_layers=2, grad_hook=None):
    """R:463-477.  Mutates pg / pd in place; returns last critic losses and the generator loss.
    grad_hook(grads) -> grads (optional) sees every gradient dict before clipping: conditioning experiments (how far
    rounding-level gradient noise moves a multi-step result) run through it."""
    n_critic = len(alpha_list)
    losses = None
    for k in range(n_critic):
        losses, g, _, _ = critic_iteration_grads(pg, pd, x_real, z_list[k], alpha_list[k], cond,
                                                 slope, gp_weight, nh, n_layers)
        if grad_hook is not None:
            g = grad_hook(g)
        if clip_d is not None:
            _, coef = clip_coef(g, clip_d)
            g = {n: v * coef for n, v in g.items()}
        opt_d.step(pd, g)
    g_loss, g, _ = generator_iteration_grads(pg, pd, z_list[n_critic], cond, slope, nh, n_layers)
    if grad_hook is not None:
        g = grad_hook(g)
    if clip_g is not None:
        _, coef = clip_coef(g, clip_g)
        g = {n: v * coef for n, v in g.items()}
    opt_g.step(pg, g)
    return losses, g_loss
