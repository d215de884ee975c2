"""Helpers shared by the parity tests: load a golden fixture, build oracles from it."""
import os

import numpy as np
import torch

from oracle.torch_oracle import CondNet, PathConfig, Trainer, film_config, img_config, set_dropout, vanilla_config

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# xattn_film_*: from src/conditional_gan_cross_attention_with_film.py; film_*: from src/conditional_gan_film.py
FIXTURES = ["xattn_film_T3", "xattn_film_T1_leaky", "film_P1", "film_P7", "img_P9", "vanilla_G60"]
XATTN_FIXTURES = [f for f in FIXTURES if f.startswith("xattn_")]


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        d = dict(zip(("B", "G", "P", "T", "Dt", "Dp", "E", "H", "L", "n_critic"),
                     (int(v) for v in self.z["dims"])))
        self.dims = d
        self.slope = float(self.z["slope"])
        self.variant = str(self.z["variant"]) if "variant" in self.z.files else "xattn_film"

    def cfg(self, optimizer="rms_prop") -> PathConfig:
        d = self.dims
        make = {"film": film_config, "img": img_config, "vanilla": vanilla_config}.get(self.variant, PathConfig)
        return make(n_genes=d["G"], latent_dims=d["L"], embedding_dims=d["E"], hidden_dims=d["H"],
                    text_dims=d["Dt"], patch_dims=d["Dp"], negative_slope=self.slope, dropout=0.0,
                    optimizer=optimizer, n_critic=d["n_critic"])

    def t(self, key):
        return torch.from_numpy(np.asarray(self.z[key]))

    def group(self, prefix):
        p = prefix + "/"
        return {k[len(p):]: self.z[k] for k in self.z.files if k.startswith(p)}

    def state(self, prefix):
        return {k: torch.from_numpy(v) for k, v in self.group(prefix).items()}

    def inputs(self):
        return tuple(self.t("in/" + k) for k in ("x", "text", "text_pad", "patches", "patch_pad"))

    def cond(self):
        x, text, text_pad, patches, patch_pad = self.inputs()
        return (patches, patch_pad, text, text_pad)

    def trainer(self, optimizer="rms_prop") -> Trainer:
        cfg = self.cfg(optimizer)
        gen, disc = CondNet("generator", cfg), CondNet("discriminator", cfg)
        gen.load_state_dict(self.state("init_gen"), strict=True)
        disc.load_state_dict(self.state("init_disc"), strict=True)
        set_dropout(gen, 0.0)
        set_dropout(disc, 0.0)
        return Trainer(cfg, gen, disc)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / den)


def comparable(name, arr, E):
    """Post-optimiser-step comparisons skip the key-bias slice [E:2E) of every ``in_proj_bias``:
    its true gradient is identically zero (softmax is invariant to a per-query shift of the scores),
    so the reference's value there is rounding noise that RMSprop/Adam normalise to a +-O(lr) step."""
    keep = np.ones(int(np.prod(tuple(arr.shape))), dtype=bool)
    if name.endswith("in_proj_bias"):
        keep[E:2 * E] = False
    return keep
