"""Times encb_kernel (csrc/enc.hip, through gg_test_enc_bwd) at the cfg3 backward shape M = 2 * 256 * 257.  The launches it replaces
(wst MASK, wst LNB, wst ACT; ln_bwd_v4_k stays in front) take 78 + 160 + 54 us live / ~220 us isolated (profiles/r03_kernel_stats.csv).
usage: python tools/encb_probe.py [M] [reps]"""
import ctypes as C
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemm_gan_amd import _lib as L

lib = L.load()
M, E, F = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 257, 256, 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
dx = rn(M, E)
Wcat = torch.cat([(0.06 * rn(F, E)).reshape(-1), (0.05 * rn(E, F)).reshape(-1), (0.06 * rn(E, E)).reshape(-1)]).contiguous()
r2, r1 = rn(M, E).bfloat16(), rn(M, E).bfloat16()
st = torch.stack([torch.zeros(M, device=dev), torch.ones(M, device=dev)], 1).contiguous()
g2 = torch.ones(E, device=dev)
h = (rn(M, F).clamp_min(0) * (torch.rand(M, F, device=dev, generator=g) > 0.1)).bfloat16()
o = lambda n: torch.empty(M, n, dtype=torch.bfloat16, device=dev)
dres2, dh, dres1, dctx = rn(M, E).bfloat16(), o(F), o(E), o(E)
cs = torch.zeros(6, E, device=dev)
wf = torch.zeros(lib.gg_test_enc_bwd_frag_bytes() + 256 * 8 * 8 * 4, dtype=torch.uint8, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(p=0.1, call=1):
    L.check(lib.gg_test_enc_bwd(P(dx), M, P(Wcat), P(dres2), P(h), P(r1), P(st), P(g2), P(dh), P(dres1), P(dctx), P(cs),
                                C.c_float(p), 7, 1, call, P(wf), s))


for name, p in (("encb dropout 0.1", 0.1), ("encb dropout 0", 0.0), ("encb dropout 0.1", 0.1)):
    run(p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(p)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:18s} M={M}: {e0.elapsed_time(e1) / reps * 1e3:8.1f} us per pass (includes the 5 us fragment-image kernel)", flush=True)

if os.environ.get("ENCB_STAMPS"):
    names = ["phase 1: dr2 / dres2 loads", "DMA waits", "barriers", "W2^T products", "gate + dh stores", "W1^T products", "phase 3: LN1 backward + dr1 stores",
             "phase 4: Wo^T products + dctx stores"]
    nb = lib.gg_test_enc_bwd_frag_bytes()
    wf[nb:].zero_()
    run(0.1, 0x40000001)
    torch.cuda.synchronize()
    st_ = wf[nb:].view(torch.int32).reshape(-1, 8).double()[: 256 * 8]
    live = st_[st_.sum(1) > 0]
    print(f"per-wave cycle sums over the launch, mean over {live.shape[0]} waves (max in brackets)")
    for k, nm in enumerate(names):
        print(f"   {nm:40s} {live[:, k].mean():10.0f} [{live[:, k].max():9.0f}]  {100 * live[:, k].mean() / live.sum(1).mean():5.1f} %")
    print(f"   total {live.sum(1).mean():10.0f}")
