"""Times ffn2_kernel (csrc/enc.hip, through gg_test_ffn2) against the two-launch route (FFN1 + FFN2 via gg_test_linear, bf16-stored
x1 / r2 / x2 as in the default engine) at the cfg3 shapes M = R * 256 * 257; usage: python tools/ffn2_probe.py [M] [reps] [keep_rows]"""
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
x = torch.randn(M, E, device=dev, generator=g).bfloat16()
Wc = torch.cat([(0.06 * torch.randn(F, E, device=dev, generator=g)).reshape(-1), (0.05 * torch.randn(E, F, device=dev, generator=g)).reshape(-1)]).bfloat16().float()
W1f, W2f = Wc[:F * E], Wc[F * E:]
W1, W2 = W1f.reshape(F, E).bfloat16().contiguous(), W2f.reshape(E, F).bfloat16().contiguous()
b1, b2 = torch.zeros(F, device=dev), torch.zeros(E, device=dev)
gam, bet = torch.ones(E, device=dev), torch.zeros(E, device=dev)
h = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
r2, y, st = torch.empty(M, E, dtype=torch.bfloat16, device=dev), torch.empty(M, E, dtype=torch.bfloat16, device=dev), torch.empty(M, 2, device=dev)
wf = torch.zeros(lib.gg_test_ffn2_frag_bytes() + 256 * 8 * 8 * 4, dtype=torch.uint8, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def fused(variant):
    def f():
        L.check(lib.gg_test_ffn2(P(x), M, P(W1f), P(b1), P(W2f), P(b2), P(h), P(r2), 1, keep, P(gam), P(bet), P(y), 1, P(st), C.c_float(0.1), 7, 2, 3, 1,
                                 P(wf), variant, s))
    return f


def two():
    a = L.GGTestLinear()
    a.X, a.ldx, a.M, a.x_bf16, a.W, a.ldw, a.bias = x.data_ptr(), E, M, 1, W1.data_ptr(), E, b1.data_ptr()
    a.Y, a.ldy, a.y_bf16, a.y_rows, a.N, a.K, a.act_relu = h.data_ptr(), F, 1, -1, F, E, 1
    a.drop_p, a.drop_seed, a.drop_site, a.drop_call, a.drop_ld = 0.1, 7, 2, 1, F
    L.check(lib.gg_test_linear(C.byref(a), None, s))
    b = L.GGTestLinear()
    b.X, b.ldx, b.M, b.x_bf16, b.W, b.ldw, b.bias = h.data_ptr(), F, M, 1, W2.data_ptr(), F, b2.data_ptr()
    b.Y, b.ldy, b.y_bf16, b.y_rows, b.N, b.K = r2.data_ptr(), E, 1, keep, E, F
    b.drop_p, b.drop_seed, b.drop_site, b.drop_call, b.drop_ld = 0.1, 7, 3, 1, E
    b.res, b.ldres, b.res_rows, b.res_bf16 = x.data_ptr(), E, M, 1
    b.ln_g, b.ln_b, b.ln_y, b.ln_stats, b.ln_y_bf16 = gam.data_ptr(), bet.data_ptr(), y.data_ptr(), st.data_ptr(), 1
    L.check(lib.gg_test_linear(C.byref(b), None, s))


variants = [int(v) for v in os.environ.get("FFN2_VARIANTS", "0,2,4,6").split(",")]
for name, fn in [(f"ffn2 variant {v}", fused(v)) for v in variants] + [("two launches", two)] + [(f"ffn2 variant {v}", fused(v)) for v in variants] + [("two launches", two)]:
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:14s} M={M} keep={keep}: {e0.elapsed_time(e1) / reps * 1e3:8.1f} us per pass", flush=True)

if os.environ.get("FFN2_STAMPS"):
    names = ["DMA wait A", "barrier A", "W1 products", "DMA wait B", "barrier B", "hidden epilogue", "W2 products", "tile epilogue"]
    for v in [int(x) for x in os.environ["FFN2_STAMPS"].split(",")]:
        wf[lib.gg_test_ffn2_frag_bytes():].zero_()
        fused(64 + v)()
        torch.cuda.synchronize()
        st_ = wf[lib.gg_test_ffn2_frag_bytes():].view(torch.int32).reshape(-1, 8).double()
        nw = 8 if v in (0, 2) else 4
        st_ = st_[: 256 * nw]
        live = st_[st_.sum(1) > 0]
        sweeps = -(-((M + 31) // 32) // 256 // (nw))
        print(f"variant {v}: per-wave cycle sums over the launch, mean over {live.shape[0]} waves (max in brackets); ~{sweeps} sweeps x 16 chunks")
        for k, nm in enumerate(names):
            print(f"   {nm:18s} {live[:, k].mean():10.0f} [{live[:, k].max():9.0f}]   per chunk {live[:, k].mean() / (sweeps * 16):8.0f}")
        print(f"   total              {live.sum(1).mean():10.0f}")
