"""CPU-side checks of the C-ABI boundary: the libraries load without a GPU, export every symbol
include/gemmgan.h / include/gemmgan_lab.h declare, and the parameter layout is the reference's live state_dict."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from golden_util import FIXTURES, Golden
from gemm_gan_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(header="gemmgan.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gg_[a-z_0-9]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = L.load()
    names = _header_symbols()
    assert len(names) >= 25
    core = C.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(core, n), f"{n} declared in gemmgan.h but not exported by libgemmgan.so"
    assert set(names) | {"gg_lab_register"} == set(L.SYMBOLS), "ctypes table and header disagree"
    assert not [n for n in names if n.startswith("gg_test_")], "test hooks belong to gemmgan_lab.h / libgemmgan_lab.so"
    assert b"gfx950" in lib.gg_version()


def test_lab_library_loads_registers_and_exports_every_declared_symbol():
    """include/gemmgan_lab.h: gg_lab_register is the product library's, everything else libgemmgan_lab.so's; loading the lab
    library registers its opt-in kernels (gg_lab_loaded() == 1) - without a GPU."""
    names = [n for n in _header_symbols("gemmgan_lab.h")]
    assert "gg_lab_register" in names and hasattr(C.CDLL(L.LIB_PATH), "gg_lab_register")
    lab = L.load_lab()
    assert lab.gg_lab_loaded() == 1
    lab_names = [n for n in names if n != "gg_lab_register"]
    for n in lab_names:
        assert hasattr(lab, n), f"{n} declared in gemmgan_lab.h but not exported by libgemmgan_lab.so"
    assert set(lab_names) == set(L.LAB_SYMBOLS), "ctypes table and header disagree"
    core = C.CDLL(L.LIB_PATH)
    assert not [n for n in lab_names if hasattr(core, n)], "a test hook is still exported by the product library"


def test_opt_in_kernel_switches_fail_loudly_without_the_lab_library():
    """In a process that has not loaded libgemmgan_lab.so the switches of its kernels return an error (no silent fallback to the
    default route): checked in a child process, this one has the library loaded by other tests."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "from gemm_gan_amd import _lib as L\n"
        "lib = L.load()\n"
        "vals = dict(n_genes=64, latent_dims=16, embedding_dims=32, hidden_dims=24, text_dims=16, patch_dims=16, n_heads=4, n_layers=2,\n"
        "            negative_slope=0.01, dropout=0.0, lr_d=5e-4, lr_g=5e-4, optimizer=0, gp_weight=10.0, clip_d=10.0, clip_g=2.0,\n"
        "            max_batch=4, max_patches=8, max_text_tokens=1, seed=0, precision=1, variant=0)\n"
        "cfg = L.GGConfig(*[vals[f[0]] for f in L.GGConfig._fields_])\n"
        "h = C.c_void_p()\n"
        "assert lib.gg_create(C.byref(cfg), C.byref(h)) == 0\n"
        "for fn, arg in ((lib.gg_set_ffn2, 1), (lib.gg_set_encb, 1), (lib.gg_set_ffn_fused, 1), (lib.gg_set_head_fused, 1)):\n"
        "    assert fn(h, arg) != 0 and b'libgemmgan_lab.so' in lib.gg_last_error()\n"
        "    assert fn(h, 0) == 0\n"
        "print('ok')\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in L.LAB_ENV}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
    env["GG_FFN2"] = "1"          # the environment form: gg_create itself refuses
    code2 = code.replace("assert lib.gg_create(C.byref(cfg), C.byref(h)) == 0", "assert lib.gg_create(C.byref(cfg), C.byref(h)) != 0 and b'libgemmgan_lab.so' in lib.gg_last_error(); print('ok'); sys.exit(0)")
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def _create(g: Golden, **over):
    d = g.dims
    vals = dict(n_genes=d["G"], latent_dims=d["L"], embedding_dims=d["E"], hidden_dims=d["H"], text_dims=d["Dt"],
                patch_dims=d["Dp"], n_heads=4, n_layers=2, negative_slope=g.slope, dropout=0.0, lr_d=5e-4, lr_g=5e-4,
                optimizer=0, gp_weight=10.0, clip_d=10.0, clip_g=2.0, max_batch=d["B"], max_patches=d["P"],
                max_text_tokens=d["T"], seed=0, precision=0, variant=L.VARIANTS[g.variant])
    vals.update(over)
    cfg = L.GGConfig(*[vals[f[0]] for f in L.GGConfig._fields_])
    h = C.c_void_p()
    rc = L.load().gg_create(C.byref(cfg), C.byref(h))
    return rc, h


@pytest.mark.parametrize("name", FIXTURES)
def test_layout_is_the_live_state_dict(name):
    g = Golden(name)
    d = g.dims
    lib = L.load()
    rc, h = _create(g)
    assert rc == 0, lib.gg_last_error()
    for role, prefix in ((L.ROLE_GENERATOR, "init_gen"), (L.ROLE_CRITIC, "init_disc")):
        ref = {k: v for k, v in g.group(prefix).items() if not k.startswith("patches_transformer_layer.")}
        seen, end = {}, 0
        for i in range(lib.gg_param_count(h, role)):
            off, numel, ndim = C.c_int64(), C.c_int64(), C.c_int32()
            shape = (C.c_int32 * 3)()
            assert lib.gg_param_info(h, role, i, C.byref(off), C.byref(numel), C.byref(ndim), shape) == 0
            nm = lib.gg_param_name(h, role, i).decode()
            seen[nm] = tuple(shape[:ndim.value])
            assert off.value % 4 == 0 and off.value >= end       # 16-byte aligned, non-overlapping
            end = off.value + numel.value
        assert end <= lib.gg_flat_numel(h, role)
        assert set(seen) == set(ref)
        for k, shp in seen.items():
            if g.variant == "vanilla" and k.endswith(".0.0.weight"):       # GG_VARIANT_VANILLA: embedding_dims zero columns appended
                assert shp == (ref[k].shape[0], ref[k].shape[1] + d["E"]), k
            else:
                assert shp == tuple(ref[k].shape), k
    assert lib.gg_workspace_bytes(h) > 0
    lib.gg_destroy(h)


def test_errors_are_codes_not_aborts():
    g = Golden(FIXTURES[0])
    lib = L.load()
    rc, _ = _create(g, embedding_dims=30)            # not divisible by 4 heads
    assert rc != 0 and b"n_heads" in lib.gg_last_error()
    rc, _ = _create(g, dropout=1.5)
    assert rc != 0
    rc, h = _create(g)
    assert rc == 0
    # nothing bound yet: a compute entry point must refuse before touching the GPU
    cond = L.GGCond(1, 1, 1, 1, 1, 1, 1)
    rc = lib.gg_forward(h, 0, C.c_void_p(1), C.byref(cond), C.c_void_p(1), 0, None)
    assert rc != 0 and b"workspace" in lib.gg_last_error()
    assert lib.gg_set_dropout(h, C.c_float(0.1)) != 0   # created with dropout 0 -> one replica only
    lib.gg_destroy(h)


@pytest.mark.parametrize("name", FIXTURES)
def test_facade_state_dict_matches_reference_keys(name):
    import gemm_gan_amd as gga
    g = Golden(name)
    d = g.dims
    mod = {"film": gga.film, "img": gga.img_transformer}.get(g.variant, gga)   # the mirror module of the fixture's reference file
    if g.variant == "vanilla":
        gen, disc = gga.vanilla.WGAN_GP_model_nocond(d["L"], d["G"], [], [], [d["H"], d["H"], d["G"]], [d["H"], d["H"], 1], g.slope, False)
    else:
        gen, disc = mod.WGAN_GP_model(d["L"], d["G"], d["E"], [d["H"], d["H"], d["G"]], [d["H"], d["H"], 1],
                                      d["Dt"], d["Dp"], g.slope, False)
    for net, prefix in ((gen, "init_gen"), (disc, "init_disc")):
        ref = g.group(prefix)
        sd = net.state_dict()
        assert list(sd) == list(ref) or set(sd) == set(ref)
        for k in ref:
            assert tuple(sd[k].shape) == tuple(ref[k].shape), k
        net.load_state_dict(g.state(prefix), strict=True)
    with pytest.raises(RuntimeError, match="not bound"):
        disc(*[torch.zeros(1)] * {"xattn_film": 5, "vanilla": 1}.get(g.variant, 4))


@pytest.mark.parametrize("idx", [0, 2, 4])
def test_facade_same_seed_same_init_as_reference(idx):
    """Construction order mirrors the reference, so torch.manual_seed(s) reproduces its initialisation."""
    import gemm_gan_amd as gga
    g = Golden(FIXTURES[idx])
    d = g.dims
    torch.manual_seed(1234 + 100 * idx)          # the seed oracle/make_golden.py used for fixture idx
    mod = {"film": gga.film, "img": gga.img_transformer}.get(g.variant, gga)
    gen, disc = mod.WGAN_GP_model(d["L"], d["G"], d["E"], [d["H"], d["H"], d["G"]], [d["H"], d["H"], 1],
                                  d["Dt"], d["Dp"], g.slope, False)
    for net, prefix in ((gen, "init_gen"), (disc, "init_disc")):
        for k, v in g.group(prefix).items():
            assert np.array_equal(net.state_dict()[k].numpy(), v), k


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gemm_gan_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert "oracle" not in re.sub(r"#.*|\"\"\".*?\"\"\"", "", open(os.path.join(pkg, fn)).read(), flags=re.S), fn


@pytest.mark.parametrize("name", FIXTURES)
def test_backward_stage_ranges_tile_the_flat_gradient_buffer_in_reverse_order(name):
    """gg_cond_stage_range (data-parallel hosts all-reduce a stage's range while the next stage runs): stage 0 ends where the MLP range
    begins, every later stage ends where the previous one begins, the last one starts at offset 0 - the conditioning part of the flat
    buffer exactly once, from the back.  The unconditional variant has no stages."""
    g = Golden(name)
    lib = L.load()
    rc, h = _create(g)
    assert rc == 0, lib.gg_last_error()
    n_st = lib.gg_cond_stage_count(h)
    assert n_st == (0 if g.variant == "vanilla" else 2 + 2)
    off, numel = C.c_int64(), C.c_int64()
    for role in (L.ROLE_GENERATOR, L.ROLE_CRITIC):
        assert lib.gg_mlp_grad_range(h, role, C.byref(off), C.byref(numel)) == 0
        last = 0                                                  # end of the last parameter slot (the buffer itself is padded past it)
        for i in range(lib.gg_param_count(h, role)):
            po, pn, nd = C.c_int64(), C.c_int64(), C.c_int32()
            shape = (C.c_int32 * 3)()
            assert lib.gg_param_info(h, role, i, C.byref(po), C.byref(pn), C.byref(nd), shape) == 0
            last = max(last, po.value + pn.value)
        assert last <= off.value + numel.value <= lib.gg_flat_numel(h, role)
        end = off.value
        for st in range(n_st):
            assert lib.gg_cond_stage_range(h, role, st, C.byref(off), C.byref(numel)) == 0
            # stage 0 (the cross-attention blocks) is empty in the variants without them; hosts skip an empty range
            assert (numel.value > 0 or (st == 0 and g.variant != "xattn_film")) and off.value + numel.value == end, (role, st, off.value, numel.value, end)
            end = off.value
        assert end == 0
        assert lib.gg_cond_stage_range(h, role, n_st, C.byref(off), C.byref(numel)) != 0          # out of range: an error, not a guess
    lib.gg_destroy(h)
