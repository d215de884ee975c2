"""Build-time check of the load pipeline of the weight-stationary Linear kernels (no GPU needed: hipcc cross-compiles gfx950).

Round 3 found that a wave-uniform run-time condition around the tile-pipeline loads (`if (next tile exists) load`) makes the
compiler's wait-count pass emit `s_waitcnt vmcnt(0)` in front of the epilogue: the wave then waits for the rows of the tile AFTER
next that it requested a moment earlier, i.e. the register prefetch does not exist (43 - 57 % of these kernels' wave time was
waits).  The loads are unconditional now; this test keeps them so: in the steady-state loop of every
production instantiation, no `vmcnt(0)` may follow a global load before the loop's stores begin."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# instantiations the headline step launches (NW, RT, KS, XB, EPI, F8, GROUPS, PF as they appear in the mangled name)
PRODUCTION = {
    "out-proj + LayerNorm (bf16 residual one tile ahead)": "ILi4ELi2ELi16ELb1ELi0ELb0ELi1ELb1EE",
    "FFN2 + LayerNorm (bf16 residual one tile ahead)": "ILi8ELi1ELi32ELb1ELi0ELb0ELi1ELb1EE",
    "out-proj + LayerNorm (fp32 residual: layer 0)": "ILi4ELi2ELi16ELb1ELi0ELb0ELi1ELb0EE",
    "FFN1, bf16 X": "ILi8ELi2ELi16ELb1ELi2ELb0ELi1ELb0EE",
    "QKV, bf16 X, three column groups": "ILi4ELi2ELi16ELb1ELi2ELb0ELi3ELb0EE",
    "QKV, fp32 X, three column groups": "ILi4ELi2ELi16ELb0ELi2ELb0ELi3ELb0EE",
    "gate kernel (reference rows one tile ahead)": "ILi8ELi2ELi16ELb1ELi3ELb0ELi1ELb1EE",
    "dctx": "ILi4ELi2ELi16ELb1ELi2ELb0ELi1ELb0EE",
    "dx += dqkv Win, two column groups": "ILi4ELi1ELi48ELb1ELi1ELb0ELi2ELb0EE",
    "dx1 += dh W1 then LayerNorm backward": "ILi8ELi1ELi32ELb1ELi4ELb0ELi1ELb0EE",
}


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "wst.s"
    src = os.path.join(ROOT, "gemm_gan_amd", "csrc", "wst.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, cwd=os.path.dirname(src), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    return out.read_text()


def loop_events(text, tag):
    names = [n for n in re.findall(r"^(_Z\S+):\s*; @", text, re.M) if "wst_ln_kernel" + tag in n]
    assert len(names) == 1, (tag, names)
    body = re.search(r"^" + re.escape(names[0]) + r":(.*?)s_endpgm", text, re.S | re.M).group(1).splitlines()
    first_barrier = next(i for i, l in enumerate(body) if "s_barrier" in l)
    ev = []
    for l in body[first_barrier:]:
        s = l.strip()
        if s.startswith("global_load"): ev.append("L")
        elif s.startswith("global_store"): ev.append("S")
        elif s.startswith("global_atomic"): ev.append("A")
        elif s.startswith("s_waitcnt") and "vmcnt" in s: ev.append("W" + re.search(r"vmcnt\((\d+)\)", s).group(1))
    return ev


@pytest.mark.parametrize("what", list(PRODUCTION))
def test_no_full_drain_between_the_tile_requests_and_the_epilogue(listing, what):
    ev = loop_events(listing, PRODUCTION[what])
    assert "L" in ev and "S" in ev, (what, ev[:40])
    body = ev[: ev.index("S")]            # the loop up to its first store: tile requests, epilogue-operand requests, their waits
    after_load = body[body.index("L"):]
    assert "W0" not in after_load, f"{what}: s_waitcnt vmcnt(0) after a fresh request - the prefetch is drained at once: {' '.join(body)}"
