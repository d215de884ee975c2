"""ctypes binding of libgemmgan.so (C ABI in include/gemmgan.h) and of libgemmgan_lab.so (include/gemmgan_lab.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing the
engine raises.  ``build()`` compiles both in-tree with hipcc for gfx950 (no GPU needed to build).
libgemmgan_lab.so - the kernel-level test hooks and the opt-in kernels that lost their A/B against the default
routes - is loaded only on demand: by a gg_test_* call through the handle ``load()`` returns, or by ``load_lab()``
(the Engine's set_ffn2 / set_encb / set_ffn_fused / set_head_fused wrappers, and its constructor when one of the
GG_FFN2 / GG_ENCB / GG_FFN_FUSED / GG_HEAD_FUSED environment switches is set).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgemmgan.so")
LAB_PATH = os.path.join(_HERE, "libgemmgan_lab.so")
CSRC = os.path.join(_HERE, "csrc")

ROLE_GENERATOR, ROLE_CRITIC = 0, 1
OPT_KINDS = {"rms_prop": 0, "adam": 1, "adamw": 2}
LOSS_D_REAL, LOSS_D_FAKE, LOSS_GP, LOSS_G, N_LOSSES = 0, 1, 2, 3, 8
LAY_KC, LAY_KS = 0, 1
PRECISIONS = {"f32": 0, "bf16": 1, "fp8": 2, "bf16x3": 3}
VARIANTS = {"xattn_film": 0, "film": 1, "img": 2, "vanilla": 3}      # GG_VARIANT_* (include/gemmgan.h)


class GGConfig(C.Structure):
    _fields_ = [("n_genes", C.c_int32), ("latent_dims", C.c_int32), ("embedding_dims", C.c_int32),
                ("hidden_dims", C.c_int32), ("text_dims", C.c_int32), ("patch_dims", C.c_int32),
                ("n_heads", C.c_int32), ("n_layers", C.c_int32), ("negative_slope", C.c_float),
                ("dropout", C.c_float), ("lr_d", C.c_float), ("lr_g", C.c_float), ("optimizer", C.c_int32),
                ("gp_weight", C.c_float), ("clip_d", C.c_float), ("clip_g", C.c_float),
                ("max_batch", C.c_int32), ("max_patches", C.c_int32), ("max_text_tokens", C.c_int32),
                ("seed", C.c_uint64), ("precision", C.c_int32), ("variant", C.c_int32)]


class GGTestLinear(C.Structure):       # gg_test_linear_args (include/gemmgan.h)
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int64), ("M", C.c_int64), ("x_bf16", C.c_int32),
                ("W", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p),
                ("Y", C.c_void_p), ("ldy", C.c_int64), ("y_bf16", C.c_int32), ("y_rows", C.c_int64),
                ("N", C.c_int32), ("K", C.c_int32),
                ("film_g", C.c_void_p), ("film_b", C.c_void_p), ("film_ld", C.c_int64), ("film_group", C.c_int32),
                ("y_row_group", C.c_int32), ("act_relu", C.c_int32),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_site", C.c_uint32), ("drop_call", C.c_uint32),
                ("drop_ld", C.c_int64),
                ("mask_ref", C.c_void_p), ("ldref", C.c_int64), ("mask_scale", C.c_float), ("mask_bf16", C.c_int32),
                ("accumulate", C.c_int32),
                ("res", C.c_void_p), ("ldres", C.c_int64), ("res_rows", C.c_int64),
                ("ln_g", C.c_void_p), ("ln_b", C.c_void_p), ("ln_y", C.c_void_p), ("ln_stats", C.c_void_p),
                ("res_bf16", C.c_int32), ("ln_y_bf16", C.c_int32),
                ("lnb_dres", C.c_void_p), ("lnb_dgamma", C.c_void_p), ("lnb_dbeta", C.c_void_p), ("lnb_dbias", C.c_void_p),
                ("w_parts", C.c_void_p), ("route", C.c_int32)]


class GGCond(C.Structure):
    _fields_ = [("patches", C.c_void_p), ("patch_pad", C.c_void_p), ("text", C.c_void_p),
                ("text_pad", C.c_void_p), ("B", C.c_int32), ("P", C.c_int32), ("T", C.c_int32)]


# name -> (restype, argtypes): every symbol include/gemmgan.h declares
SYMBOLS = {
    "gg_last_error": (C.c_char_p, []),
    "gg_version": (C.c_char_p, []),
    "gg_create": (C.c_int, [C.POINTER(GGConfig), C.POINTER(C.c_void_p)]),
    "gg_destroy": (None, [C.c_void_p]),
    "gg_param_count": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_param_name": (C.c_char_p, [C.c_void_p, C.c_int, C.c_int]),
    "gg_param_info": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gg_flat_numel": (C.c_int64, [C.c_void_p, C.c_int]),
    "gg_bind_net": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gg_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "gg_bind_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "gg_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(GGCond), C.c_void_p, C.c_int, C.c_void_p]),
    "gg_critic_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(GGCond), C.c_void_p, C.c_void_p]),
    "gg_critic_backward_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(GGCond), C.c_void_p, C.c_void_p]),
    "gg_critic_backward_cond": (C.c_int, [C.c_void_p, C.POINTER(GGCond), C.c_void_p]),
    "gg_critic_cond_prefetch": (C.c_int, [C.c_void_p, C.POINTER(GGCond), C.c_void_p]),
    "gg_mlp_grad_range": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gg_cond_stage_count": (C.c_int, [C.c_void_p]),
    "gg_cond_stage_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gg_critic_backward_cond_stage": (C.c_int, [C.c_void_p, C.POINTER(GGCond), C.c_int, C.c_void_p]),
    "gg_generator_backward_cond_stage": (C.c_int, [C.c_void_p, C.POINTER(GGCond), C.c_int, C.c_void_p]),
    "gg_side_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gg_gradient_penalty": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(GGCond), C.c_int, C.c_void_p, C.c_void_p]),
    "gg_generator_backward_head": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(GGCond), C.c_void_p, C.c_void_p]),
    "gg_generator_backward_cond": (C.c_int, [C.c_void_p, C.POINTER(GGCond), C.c_void_p]),
    "gg_critic_apply": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p]),
    "gg_generator_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(GGCond), C.c_void_p, C.c_void_p]),
    "gg_generator_apply": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p]),
    "gg_generator_prefetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(GGCond), C.c_void_p]),
    "gg_set_prefetch": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_graph": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_graph_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gg_set_side_streams": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(GGCond), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gg_set_lr": (C.c_int, [C.c_void_p, C.c_int, C.c_float]),
    "gg_set_dropout": (C.c_int, [C.c_void_p, C.c_float]),
    "gg_set_seed": (C.c_int, [C.c_void_p, C.c_uint64]),
    "gg_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_flash": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_tlin": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_ffn_fused": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_ffn2": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_encb": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_xstore": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_lnb_fused": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_head_fused": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_test_ffn_fused": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_void_p]),
    "gg_test_ffn2_frag_bytes": (C.c_int64, []),
    "gg_test_set_enc_grid": (C.c_int, [C.c_int]),
    "gg_test_ffn2": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                               C.c_void_p, C.c_int, C.c_void_p]),
    "gg_test_enc_bwd_frag_bytes": (C.c_int64, []),
    "gg_test_enc_bwd": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 10 + [C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "gg_set_sqx": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_bstore": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_wgrad": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_debug_buffer_is_bf16": (C.c_int, [C.c_void_p, C.c_char_p]),
    "gg_reset_optimizer_steps": (C.c_int, [C.c_void_p]),
    "gg_get_optimizer_step": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_set_optimizer_step": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gg_eval_knn_scratch": (C.c_long, [C.c_long, C.c_long, C.c_int]),
    "gg_eval_knn_width": (C.c_int, [C.c_int]),
    "gg_eval_knn": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "gg_eval_prdc_counts": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "gg_eval_nn2_scratch": (C.c_long, [C.c_long, C.c_long]),
    "gg_eval_nn2": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "gg_test_gemm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                               C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_float,
                               C.c_int, C.c_void_p]),
    "gg_test_gemm_small": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                     C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_float,
                                     C.c_int, C.c_void_p]),
    "gg_test_gemm_bf16_stored": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                           C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gg_test_gemm_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                    C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_float,
                                    C.c_int, C.c_void_p]),
    "gg_test_linear": (C.c_int, [C.POINTER(GGTestLinear), C.POINTER(C.c_int32), C.c_void_p]),
    "gg_test_attn_kernel_name": (C.c_char_p, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "gg_test_attn_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                   C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int64, C.c_void_p]),
    "gg_test_attn_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int,
                                   C.c_int64, C.c_void_p]),
    "gg_test_wgrad": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int64,
                                C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "gg_test_sqx_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p]),
    "gg_test_sqx_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gg_test_head_fwd": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gg_test_head_bwd": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gg_test_ln_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]),
    "gg_launch_count": (C.c_int64, [C.c_void_p]),
    "gg_reset_launch_count": (C.c_int, [C.c_void_p]),
    "gg_bind_streams": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gg_gp_profile": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "gg_phase_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_phase_count": (C.c_int, [C.c_void_p]),
    "gg_phase_read": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double)]),
    "gg_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "gg_profile_enable_class": (C.c_int, [C.c_void_p, C.c_char_p]),
    "gg_profile_add_class": (C.c_int, [C.c_void_p, C.c_char_p]),
    "gg_profile_collect": (C.c_int, [C.c_void_p]),
    "gg_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gg_debug_buffer": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
}

SYMBOLS["gg_lab_register"] = (C.c_int, [C.c_void_p, C.c_uint64])
# include/gemmgan_lab.h: what libgemmgan_lab.so exports
LAB_SYMBOLS = {k: SYMBOLS.pop(k) for k in list(SYMBOLS) if k.startswith("gg_test_")}
LAB_SYMBOLS["gg_lab_loaded"] = (C.c_int, [])
LAB_ENV = ("GG_FFN2", "GG_ENCB", "GG_FFN_FUSED", "GG_HEAD_FUSED")      # environment switches that select kernels of the lab library

_lib = None
_core = None
_lab = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip into gemm_gan_amd/libgemmgan.so and libgemmgan_lab.so for gfx950 (hipcc, in-tree)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libgemmgan.so failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)
    return LIB_PATH


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    return lib


class _Handle:
    """What load() returns: attribute access resolves a symbol of include/gemmgan.h in libgemmgan.so, and a symbol of
    include/gemmgan_lab.h in libgemmgan_lab.so (loaded at that moment if it was not)."""

    def __getattr__(self, name):
        if name in LAB_SYMBOLS:
            fn = getattr(load_lab(), name)
        else:
            fn = getattr(_core, name)
        object.__setattr__(self, name, fn)
        return fn


def load():
    """Load the HIP engine.  Raises (never falls back) when the library is absent or broken."""
    global _lib, _core
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: the HIP engine is not built "
                          f"(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C {CSRC}`). "
                          "There is no CPU fallback.")
    _core = _bind(C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL), SYMBOLS)      # global: the lab library resolves its references here
    _lib = _Handle()
    return _lib


def load_lab():
    """Load libgemmgan_lab.so (after the engine library); its constructor registers the opt-in kernels with the engine."""
    global _lab
    if _lab is not None:
        return _lab
    load()
    if not os.path.exists(LAB_PATH):
        raise ImportError(f"{LAB_PATH} not found: build it with `make -C {CSRC}` (test hooks and opt-in kernels; the engine "
                          "itself does not need it)")
    lab = _bind(C.CDLL(LAB_PATH, mode=C.RTLD_GLOBAL), LAB_SYMBOLS)
    if lab.gg_lab_loaded() != 1:
        raise ImportError("libgemmgan_lab.so did not register with libgemmgan.so (libraries of different builds?)")
    _lab = lab
    return lab


def lab_wanted_by_env() -> bool:
    return any(k in os.environ for k in LAB_ENV)


def check(rc: int):
    if rc != 0:
        raise RuntimeError("gemmgan: " + load().gg_last_error().decode("utf-8", "replace"))
