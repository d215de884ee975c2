"""Times the fused self-attention kernels alone (gg_test_attn_fwd / gg_test_attn_bwd) with and without dropout at a given shape.
usage: python tools/attn_probe.py [N S E nh reps]   (defaults: 768 257 256 4 20 - the cfg3 critic forward)"""
import ctypes as C
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from gemm_gan_amd import _lib as L

N, S, E, nh, reps = (int(a) for a in (sys.argv[1:6] + ["768", "257", "256", "4", "20"][len(sys.argv) - 1:]))
lib = L.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
qkv = (torch.randn(N, S, 3 * E, device=dev, generator=g)).to(torch.bfloat16)
dctx = torch.randn(N, S, E, device=dev, generator=g).to(torch.bfloat16)
mask = torch.zeros(N, S, dtype=torch.uint8, device=dev)
ctx = torch.empty(N, S, E, dtype=torch.bfloat16, device=dev)
lse = torch.empty(N, nh, S, device=dev)
delta = torch.empty(N, nh, S, device=dev)
dqkv = torch.empty(N, S, 3 * E, dtype=torch.bfloat16, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(p):
    def fwd():
        L.check(lib.gg_test_attn_fwd(P(qkv), P(mask), N, P(ctx), P(lse), N, S, E, nh, C.c_float(p), 7, 1000, 1, 1, 0, st))

    def bwd():
        L.check(lib.gg_test_attn_bwd(P(qkv), P(ctx), P(dctx), P(lse), P(delta), P(mask), N, P(dqkv), N, S, E, nh, C.c_float(p), 7, 1000, 1, 1, 0, st))
    out = []
    for f in (fwd, bwd):
        for _ in range(3):
            f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            f()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / reps)
    return out


for p in (0.1, 0.0):
    f, b = run(p)
    print(f"N={N} S={S} E={E} nh={nh} dropout {p}: forward {f:8.1f} us   backward (dQ + dK|dV) {b:8.1f} us")
