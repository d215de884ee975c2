"""Times ffn_fused_kernel alone (through gg_test_ffn_fused) against the two-launch route (FFN1 + FFN2 via gg_test_linear) at the cfg3
replica shape M = 768 * 257; run under rocprofv3 --pmc for the counters (tools/ffn_pmc.sh)."""
import ctypes as C
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemm_gan_amd import _lib as L

lib = L.load()
M, E, F = int(sys.argv[1]) if len(sys.argv) > 1 else 768 * 257, 256, 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
keep = int(sys.argv[3]) if len(sys.argv) > 3 else 2 * M // 3
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, E, device=dev, generator=g)
W1 = (0.06 * torch.randn(F, E, device=dev, generator=g)).bfloat16()
W2 = (0.05 * torch.randn(E, F, device=dev, generator=g)).bfloat16()
W2T = W2.t().contiguous()
b1, b2 = torch.zeros(F, device=dev), torch.zeros(E, device=dev)
gam, bet = torch.ones(E, device=dev), torch.zeros(E, device=dev)
h = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
r2, y, st = torch.empty(M, E, device=dev), torch.empty(M, E, device=dev), torch.empty(M, 2, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def fused():
    L.check(lib.gg_test_ffn_fused(P(x), M, P(W1), P(b1), P(W2T), P(b2), P(h), P(r2), keep, P(gam), P(bet), P(y), P(st), C.c_float(0.1), 7, 2, 3, 1, s))


def two():
    a = L.GGTestLinear()
    a.X, a.ldx, a.M, a.x_bf16, a.W, a.ldw, a.bias = x.data_ptr(), E, M, 0, W1.data_ptr(), E, b1.data_ptr()
    a.Y, a.ldy, a.y_bf16, a.y_rows, a.N, a.K, a.act_relu = h.data_ptr(), F, 1, -1, F, E, 1
    a.drop_p, a.drop_seed, a.drop_site, a.drop_call, a.drop_ld = 0.1, 7, 2, 1, F
    L.check(lib.gg_test_linear(C.byref(a), None, s))
    b = L.GGTestLinear()
    b.X, b.ldx, b.M, b.x_bf16, b.W, b.ldw, b.bias = h.data_ptr(), F, M, 1, W2.data_ptr(), F, b2.data_ptr()
    b.Y, b.ldy, b.y_bf16, b.y_rows, b.N, b.K = r2.data_ptr(), E, 0, keep, E, F
    b.drop_p, b.drop_seed, b.drop_site, b.drop_call, b.drop_ld = 0.1, 7, 3, 1, E
    b.res, b.ldres, b.res_rows = x.data_ptr(), E, M
    b.ln_g, b.ln_b, b.ln_y, b.ln_stats = gam.data_ptr(), bet.data_ptr(), y.data_ptr(), st.data_ptr()
    L.check(lib.gg_test_linear(C.byref(b), None, s))


for name, fn in (("fused", fused), ("two launches", two)):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:14s} M={M} keep={keep}: {e0.elapsed_time(e1) / reps * 1e3:8.1f} us per pass")
