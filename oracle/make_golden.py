#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference in the build container.

TEST INFRASTRUCTURE ONLY (never imported by the product path, never run on the GPU box:
/root/reference does not exist there).  Run from the repo root:

    python oracle/make_golden.py            # writes tests/golden/*.npz

What it does
  * registers inert stand-ins in ``sys.modules`` for evaluation-only third-party imports the
    image lacks (lightgbm, umap, seaborn, catboost, ot, torch_geometric ...; none is touched by
    the model/trainer code) and imports
    ``/root/reference/src/conditional_gan_cross_attention_with_film.py`` (and, for the FiLM-only sibling fixtures,
    ``/root/reference/src/conditional_gan_film.py``) unmodified,
  * zeroes every dropout probability (bitwise RNG parity with torch dropout is not a goal;
    parity runs are p=0, SURVEY.md section 7 "Hard parts" (b)),
  * records - without editing the reference - the ``z`` / ``alpha`` draws (wrapping
    ``torch.normal`` / ``torch.rand``), the GP gradient (wrapping ``torch.autograd.grad``) and the
    pre-clip gradients + total norm (wrapping ``torch.nn.utils.clip_grad_norm_``),
  * stores inputs, initial ``state_dict``s, stage activations (forward hooks), losses, gradients
    and post-step parameters as small ``.npz`` fixtures.

Only DATA leaves this container: no reference source text is written anywhere.
"""
from __future__ import annotations

import copy
import importlib
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

REF_SRC = "/root/reference/src"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class _Inert(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {})


def import_reference(module="conditional_gan_cross_attention_with_film"):
    for m in ["lightgbm", "umap", "seaborn", "catboost", "ot", "rnaseq_contrastive_model",
              "torch_geometric", "torch_geometric.nn", "geomloss", "timm", "openslide"]:
        if m not in sys.modules:
            mod = _Inert(m)
            mod.__path__ = []
            mod.__spec__ = importlib.machinery.ModuleSpec(m, None)
            sys.modules[m] = mod
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    return importlib.import_module(module)


def zero_dropout(net):
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0


class Recorder:
    """Context manager wrapping the torch entry points the reference's trainer calls."""

    def __init__(self):
        self.z, self.alpha, self.gp_grads, self.clips = [], [], [], []

    def __enter__(self):
        self._normal, self._rand = torch.normal, torch.rand
        self._agrad, self._clip = torch.autograd.grad, torch.nn.utils.clip_grad_norm_
        rec = self

        def normal(*a, **k):
            out = rec._normal(*a, **k)
            rec.z.append(out.detach().clone())
            return out

        def rand(*a, **k):
            out = rec._rand(*a, **k)
            rec.alpha.append(out.detach().clone())
            return out

        def agrad(*a, **k):
            out = rec._agrad(*a, **k)
            rec.gp_grads.append(out[0].detach().clone())
            return out

        def clip(params, max_norm, *a, **k):
            params = list(params)
            pre = [None if p.grad is None else p.grad.detach().clone() for p in params]
            tot = rec._clip(params, max_norm, *a, **k)
            rec.clips.append((pre, float(tot), float(max_norm)))
            return tot

        torch.normal, torch.rand = normal, rand
        torch.autograd.grad, torch.nn.utils.clip_grad_norm_ = agrad, clip
        return self

    def __exit__(self, *exc):
        torch.normal, torch.rand = self._normal, self._rand
        torch.autograd.grad, torch.nn.utils.clip_grad_norm_ = self._agrad, self._clip


def synth_inputs(dims, seed):
    rng = np.random.default_rng(seed)
    B, G, P, T, Dt, Dp = (dims[k] for k in ("B", "G", "P", "T", "Dt", "Dp"))
    x = rng.standard_normal((B, G), dtype=np.float32)
    patches = rng.standard_normal((B, P, Dp), dtype=np.float32)
    text = rng.standard_normal((B, T, Dt), dtype=np.float32)
    patch_pad = np.zeros((B, P), dtype=bool)
    text_pad = np.zeros((B, T), dtype=bool)
    if P > 2:
        patch_pad[0, P - 2:] = True          # ragged: two padded patches in row 0
        patch_pad[B - 1, 1:] = True          # all but the first patch padded in the last row
    if T > 1:
        text_pad[1, T - 1:] = True           # token 0 (text CLS) always valid
    t = torch.from_numpy
    return dict(x=t(x), text=t(text), text_pad=t(text_pad), patches=t(patches), patch_pad=t(patch_pad))


def sd_to_np(prefix, sd, out, stride=1):
    for k, v in sd.items():
        a = v.detach().cpu().numpy().astype(np.float32)
        out[f"{prefix}/{k}"] = a if stride == 1 else a.reshape(-1)[::stride].copy()


class Calls:
    """The two reference files differ in argument order / text rank; fixtures store text as [B,T,Dt] for both."""

    def __init__(self, film, vanilla=False):
        self.film = film
        self.vanilla = vanilla

    def net(self, net, v, inp):
        if self.vanilla:    # vanilla_gan_unconditional.py:122 / :169: forward(x)
            return net(v)
        if self.film:       # conditional_gan_film.py:130 / :183: (x, text_embedding [B,Dt], patches, padding_mask)
            return net(v, inp["text"][:, 0, :], inp["patches"], inp["patch_pad"])
        return net(v, inp["patches"], inp["patch_pad"], inp["text"], inp["text_pad"])

    def train_disc(self, w, x, z, inp):
        if self.vanilla:
            return w.train_disc(x, z)
        if self.film:
            return w.train_disc(x, z, inp["text"][:, 0, :], inp["patches"], inp["patch_pad"])
        return w.train_disc(x, z, inp["text"], inp["text_pad"], inp["patches"], inp["patch_pad"])

    def train_gen(self, w, z, inp):
        if self.vanilla:
            return w.train_gen(z)
        if self.film:
            return w.train_gen(z, inp["text"][:, 0, :], inp["patches"], inp["patch_pad"])
        return w.train_gen(z, inp["text"], inp["text_pad"], inp["patches"], inp["patch_pad"])

    def train(self, w, x, inp):
        if self.vanilla:
            return w.train(x)
        if self.film:
            return w.train(x, inp["text"][:, 0, :], inp["patches"], inp["patch_pad"])
        return w.train(x, inp["text"], inp["text_pad"], inp["patches"], inp["patch_pad"])

    def generate(self, w, x, inp):
        if self.vanilla:
            return w.generate_samples(x)
        if self.film:
            return w.generate_samples(x, inp["text"][:, 0, :], inp["patches"], inp["patch_pad"])
        return w.generate_samples(x, inp["text"], inp["text_pad"], inp["patches"], inp["patch_pad"])


def grads_of(net):
    return [None if p.grad is None else p.grad.detach().clone() for p in net.parameters()]


def build(ref, dims, optimizer, init=None):
    if dims.get("variant") == "vanilla":
        # WGAN_GP_nocond(input_dims, latent_dims, vocab_sizes, generator_dims, discriminator_dims, ...) (V:214)
        w = ref.WGAN_GP_nocond(dims["G"], dims["L"], [], [dims["H"], dims["H"], dims["G"]], [dims["H"], dims["H"], 1],
                               negative_slope=dims.get("slope", 0.0), optimizer=optimizer, n_critic=dims["n_critic"],
                               results_dire="/tmp/_gemmgan_golden")
        w.build_WGAN_GP_nocond()
        if init is not None:
            w.gen.load_state_dict(init[0])
            w.disc.load_state_dict(init[1])
        w.init_train()
        return w
    w = ref.WGAN_GP(dims["G"], dims["L"], dims["E"], [dims["H"], dims["H"], dims["G"]],
                    [dims["H"], dims["H"], 1], text_embedding_dims=dims["Dt"],
                    patches_embedding_dims=dims["Dp"], negative_slope=dims.get("slope", 0.0),
                    optimizer=optimizer, n_critic=dims["n_critic"], results_dire="/tmp/_gemmgan_golden")
    w.build_WGAN_GP()
    if init is not None:
        w.gen.load_state_dict(init[0])
        w.disc.load_state_dict(init[1])
    w.init_train()
    zero_dropout(w.gen)
    zero_dropout(w.disc)
    return w


def stage_hooks(net, role, taps):
    hs = []

    def tap(name, pick=lambda o: o):
        def fn(_m, _i, o):
            taps[name] = pick(o).detach().clone()
        return fn
    if hasattr(net, "film_generator"):
        hs.append(net.film_generator.register_forward_hook(tap("film_pre")))
    if not hasattr(net, "patches_encoder"):          # unconditional nets: the MLP only
        blocks = getattr(net, role)
        for i, blk in enumerate(blocks):
            hs.append(blk[0].register_forward_hook(tap(f"mlp_pre{i}")))
        hs.append(net.final_layer.register_forward_hook(tap("out")))
        return hs
    if hasattr(net, "text_encoder"):
        hs.append(net.text_encoder.register_forward_hook(tap("text_enc")))
    hs.append(net.patches_encoder.register_forward_hook(tap("patch_emb")))
    for i, layer in enumerate(net.patches_transformer.layers):
        hs.append(layer.register_forward_hook(tap(f"enc_layer{i}")))
    if hasattr(net, "patch2text_attention"):
        hs.append(net.patch2text_attention.register_forward_hook(tap("t2i", lambda o: o[0][:, 0, :])))
        hs.append(net.text2patch_attention.register_forward_hook(tap("i2t", lambda o: o[0][:, 0, :])))
    blocks = getattr(net, role)
    for i, blk in enumerate(blocks):
        hs.append(blk[0].register_forward_hook(tap(f"mlp_pre{i}")))
    hs.append(net.final_layer.register_forward_hook(tap("out")))
    return hs


def make_fixture(ref, name, dims, seed):
    out = {}
    out["dims"] = np.array([dims[k] for k in ("B", "G", "P", "T", "Dt", "Dp", "E", "H", "L", "n_critic")],
                           dtype=np.int64)
    out["slope"] = np.float32(dims.get("slope", 0.0))
    film = dims.get("variant", "xattn_film") in ("film", "img")      # the 4-argument files
    out["variant"] = np.array(dims.get("variant", "xattn_film"))
    call = Calls(film, dims.get("variant") == "vanilla")
    torch.manual_seed(seed)
    w0 = build(ref, dims, "rms_prop")
    init = (copy.deepcopy(w0.gen.state_dict()), copy.deepcopy(w0.disc.state_dict()))
    sd_to_np("init_gen", init[0], out)
    sd_to_np("init_disc", init[1], out)
    inp = synth_inputs(dims, seed + 1)
    for k, v in inp.items():
        out[f"in/{k}"] = v.numpy()
    x, text, text_pad, patches, patch_pad = (inp[k] for k in ("x", "text", "text_pad", "patches", "patch_pad"))

    # ---- stage activations of one critic forward on the real genes and one generator forward
    w0.disc.train()
    w0.gen.train()
    taps = {}
    hs = stage_hooks(w0.disc, "discriminator", taps)
    with torch.no_grad():
        call.net(w0.disc, x, inp)
    for h in hs:
        h.remove()
    for k, v in taps.items():
        out[f"disc_fwd/{k}"] = v.numpy()
    g = torch.Generator().manual_seed(seed + 2)
    z0 = torch.randn(dims["B"], dims["L"], generator=g)
    taps = {}
    hs = stage_hooks(w0.gen, "generator", taps)
    with torch.no_grad():
        call.net(w0.gen, z0, inp)
    for h in hs:
        h.remove()
    out["gen_fwd/z"] = z0.numpy()
    for k, v in taps.items():
        out[f"gen_fwd/{k}"] = v.numpy()
    # eval-mode inference (generate_samples, :601-608) with z recorded
    torch.manual_seed(seed + 3)
    with Recorder() as rec:
        _, x_gen = call.generate(w0, x, inp)
    out["infer/z"] = rec.z[0].numpy()
    out["infer/x_gen"] = x_gen.numpy()

    # ---- one critic iteration (train_disc :376-423) from the initial state
    w1 = build(ref, dims, "rms_prop", init)
    z1 = torch.randn(dims["B"], dims["L"], generator=g)
    torch.manual_seed(seed + 4)
    with Recorder() as rec:
        call.train_disc(w1, x, z1, inp)
    out["critic1/z"] = z1.numpy()
    out["critic1/alpha"] = rec.alpha[0].numpy()
    out["critic1/grad_x_hat"] = rec.gp_grads[0].numpy()
    out["critic1/losses"] = np.array([float(w1.disc_loss), *w1.d_batch_loss], dtype=np.float64)
    if film or call.vanilla:        # no clipping in these files: the gradients are still on the parameters
        pre = grads_of(w1.disc)
    else:
        pre, tot, mx = rec.clips[0]
        out["critic1/grad_total_norm"] = np.float64(tot)
    for (n, _p), gpre in zip(w1.disc.named_parameters(), pre):
        if gpre is not None:
            out[f"critic1/grad/{n}"] = gpre.numpy()
        else:
            out[f"critic1/grad_none/{n}"] = np.zeros(0, dtype=np.float32)
    sd_to_np("critic1/post_disc", w1.disc.state_dict(), out)

    # ---- one generator iteration (train_gen :425-461) from the initial state
    w2 = build(ref, dims, "rms_prop", init)
    z2 = torch.randn(dims["B"], dims["L"], generator=g)
    with Recorder() as rec:
        call.train_gen(w2, z2, inp)
    out["gen1/z"] = z2.numpy()
    out["gen1/loss"] = np.float64(float(w2.gen_loss))
    if film or call.vanilla:
        pre = grads_of(w2.gen)
    else:
        pre, tot, mx = rec.clips[0]
        out["gen1/grad_total_norm"] = np.float64(tot)
    for (n, _p), gpre in zip(w2.gen.named_parameters(), pre):
        if gpre is not None:
            out[f"gen1/grad/{n}"] = gpre.numpy()
    sd_to_np("gen1/post_gen", w2.gen.state_dict(), out)

    # ---- full train() (:463-477) for the three optimisers
    # conditional_gan_img_transformer.py:277-286 knows rms_prop and adam only
    for opt in (("rms_prop", "adam") if dims.get("variant") in ("img", "vanilla") else ("rms_prop", "adam", "adamw")):
        w = build(ref, dims, opt, init)
        torch.manual_seed(seed + 5)
        with Recorder() as rec:
            call.train(w, x, inp)
        assert len(rec.z) == dims["n_critic"] + 1 and len(rec.alpha) == dims["n_critic"]
        out[f"step_{opt}/z"] = np.stack([t.numpy() for t in rec.z])
        out[f"step_{opt}/alpha"] = np.stack([t.numpy() for t in rec.alpha])
        out[f"step_{opt}/d_batch_loss"] = np.asarray(w.d_batch_loss, dtype=np.float64)
        out[f"step_{opt}/g_batch_loss"] = np.asarray(w.g_batch_loss, dtype=np.float64)
        out[f"step_{opt}/disc_loss"] = np.float64(float(w.disc_loss))
        out[f"step_{opt}/gen_loss"] = np.float64(float(w.gen_loss))
        out[f"step_{opt}/grad_norms"] = np.array([c[1] for c in rec.clips], dtype=np.float64)
        sd_to_np(f"step_{opt}/post_gen", w.gen.state_dict(), out)          # every element (no subsampling)
        sd_to_np(f"step_{opt}/post_disc", w.disc.state_dict(), out)

    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)/1024:.0f} KiB")


FIXTURES = {
    # ragged patches + padded text token, T>1, odd gene count, dh = E/4 = 8
    "xattn_film_T3": dict(B=5, G=37, P=6, T=3, Dt=24, Dp=40, E=32, H=32, L=16, n_critic=5),
    # the BASELINE shape family in miniature: single text token, slope != 0 exercises LeakyReLU
    "xattn_film_T1_leaky": dict(B=4, G=50, P=9, T=1, Dt=16, Dp=24, E=32, H=24, L=8, n_critic=2, slope=0.2),
    # FiLM-only sibling (src/conditional_gan_film.py; BASELINE configs[1]: one patch, text vector)
    "film_P1": dict(B=6, G=41, P=1, T=1, Dt=20, Dp=28, E=32, H=24, L=12, n_critic=5, variant="film"),
    "film_P7": dict(B=4, G=33, P=7, T=1, Dt=16, Dp=24, E=32, H=16, L=8, n_critic=2, variant="film", slope=0.1),
    # image-transformer sibling (src/conditional_gan_img_transformer.py; BASELINE configs[4] family)
    "img_P9": dict(B=5, G=45, P=9, T=1, Dt=12, Dp=28, E=32, H=24, L=10, n_critic=3, variant="img"),
    # unconditional model (src/vanilla_gan_unconditional.py; BASELINE configs[0] family).  P, T, Dt, Dp, E only size the
    # dummy conditioning inputs the engine's entry points still take
    "vanilla_G60": dict(B=7, G=60, P=1, T=1, Dt=8, Dp=8, E=8, H=20, L=12, n_critic=5, variant="vanilla", slope=0.2),
}
REF_MODULE = {"xattn_film": "conditional_gan_cross_attention_with_film", "film": "conditional_gan_film",
              "img": "conditional_gan_img_transformer", "vanilla": "vanilla_gan_unconditional"}


def main():
    torch.set_num_threads(1)           # bit-reproducible reductions
    only = set(sys.argv[1:])
    for i, (name, dims) in enumerate(FIXTURES.items()):
        if only and name not in only:
            continue
        ref = import_reference(REF_MODULE[dims.get("variant", "xattn_film")])
        make_fixture(ref, name, dims, seed=1234 + 100 * i)


if __name__ == "__main__":
    main()
