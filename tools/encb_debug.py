import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from gemm_gan_amd import _lib as L
lib = L.load()
M, E, F = int(sys.argv[1]) if len(sys.argv) > 1 else 2056, 256, 512
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
dev = "cuda:0"
g = torch.Generator().manual_seed(1)
bf = lambda t: t.float().to(torch.bfloat16).float()
dr2 = torch.randn(M, E, generator=g)
dres2 = bf(torch.randn(M, E, generator=g))
W1, W2, Wo = bf(0.06 * torch.randn(F, E, generator=g)), bf(0.05 * torch.randn(E, F, generator=g)), bf(0.06 * torch.randn(E, E, generator=g))
r1 = bf(torch.randn(M, E, generator=g))
hid = bf(torch.randn(M, F, generator=g).clamp_min(0) * (torch.rand(M, F, generator=g) > 0.1))
mu = r1.double().mean(-1, keepdim=True)
st1 = torch.cat([mu, 1.0 / torch.sqrt(((r1.double() - mu) ** 2).mean(-1, keepdim=True) + 1e-5)], 1).float()
d = lambda t, dt=torch.float32: t.to(dev, dt).contiguous()
dx_d, Wcat = d(dr2), d(torch.cat([W1.reshape(-1), W2.reshape(-1), Wo.reshape(-1)]))
dres2_d, r1_d, h_d, st1_d, g1_d = d(dres2, torch.bfloat16), d(r1, torch.bfloat16), d(hid, torch.bfloat16), d(st1), d(torch.ones(E))
dh_d = torch.full((M, F), float("nan"), dtype=torch.bfloat16, device=dev)
dres1_d, dctx_d = torch.empty(M, E, dtype=torch.bfloat16, device=dev), torch.empty(M, E, dtype=torch.bfloat16, device=dev)
cs_d = torch.zeros(3, E, device=dev)
wf = torch.empty(lib.gg_test_enc_bwd_frag_bytes(), dtype=torch.uint8, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.check(lib.gg_test_enc_bwd(P(dx_d), M, P(Wcat), P(dres2_d), P(h_d), P(r1_d), P(st1_d), P(g1_d), P(dh_d), P(dres1_d), P(dctx_d), P(cs_d), C.c_float(p), 31, 1011, 4, P(wf), s))
torch.cuda.synchronize()
ks = 1.0 / (1.0 - p) if p > 0 else 1.0
want = (dres2.double() @ W2.double()) * (hid.double() > 0) * ks
got = dh_d.float().cpu().double()
err = (got - want).abs()
bad = err > 0.02 * want.abs().max()
print("bad elements", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(got).sum()))
rows, cols = torch.nonzero(bad, as_tuple=True)
print("bad rows (first 20):", rows[:20].tolist())
print("bad cols (first 20):", cols[:20].tolist())
print("rows mod 32 histogram:", np.bincount((rows % 32).numpy(), minlength=32).tolist())
print("cols // 32 histogram:", np.bincount((cols // 32).numpy(), minlength=16).tolist())
print("cols % 32 histogram:", np.bincount((cols % 32).numpy(), minlength=32).tolist())
i = int(torch.argmax(err))
print("worst", i // F, i % F, float(got.flatten()[i]), float(want.flatten()[i]), "ratio", float(got.flatten()[i] / want.flatten()[i]) if want.flatten()[i] != 0 else None, "hid", float(hid.flatten()[i]))
