"""Time the nearest-record kernel (gemm_gan_amd/evaluate.py) and the reference's expression on the same GPU:
python tools/evalnn_probe.py [n_gen n_real dim]"""
import sys, time, pathlib
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from gemm_gan_amd import evaluate

nq, nr, dim = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2000, 8000, 5000)
g = torch.Generator(device="cuda").manual_seed(0)
gen = torch.randn(nq, dim, device="cuda", generator=g)
real = torch.randn(nr, dim, device="cuda", generator=g)
for _ in range(2):
    d1, d2 = evaluate.nearest2(gen, real)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    d1, d2 = evaluate.nearest2(gen, real)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"nearest2 [{nq} x {nr} x {dim}]: {dt * 1e3:.2f} ms  ({3.0 * nq * nr * dim / dt / 1e12:.1f} fp32 TFLOP/s of sub+mul+add)")
torch.cuda.synchronize()
t0 = time.perf_counter()
outs = []
for i in range(0, nq, 128):                      # the reference's batching (src/privacy_evaluator.py:17-24)
    outs.append((gen[i:i + 128, None] - real).pow(2).sum(dim=2).sqrt().min(dim=1).values)
ref = torch.cat(outs)
torch.cuda.synchronize()
print(f"reference expression (torch, same GPU): {(time.perf_counter() - t0) * 1e3:.1f} ms; max |d1 - ref| = {(d1 - ref).abs().max().item():.2e}")
for _ in range(2):
    r = evaluate.compute_prdc(real, gen, 5)
torch.cuda.synchronize()
t0 = time.perf_counter()
r = evaluate.compute_prdc(real, gen, 5)
torch.cuda.synchronize()
print(f"compute_prdc [{nr} real x {nq} fake x {dim}], k = 5 (two k-NN radius passes + one counting pass, L1): "
      f"{(time.perf_counter() - t0) * 1e3:.1f} ms  {r}")
