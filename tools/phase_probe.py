"""Un-profiled timeline of one cfg3 train step: milliseconds between the engine's phase marks (gg_phase_*), averaged over steps.
usage: python tools/phase_probe.py [steps] [precision]   (env switches such as GG_FFN2=1 apply)"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gemm_gan_amd as gga

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
B, G, P, T, Dt = 256, 5000, 256, 1, 512
dev = "cuda:0"
torch.manual_seed(42)
w = gga.WGAN_GP(G, 256, 256, [256, 256, G], [256, 256, 1], text_embedding_dims=Dt, patches_embedding_dims=1024, optimizer="rms_prop",
                n_critic=5, dropout=0.1, seed=1234, device=dev, results_dire="", precision=prec)
w.build_WGAN_GP()
w.init_train()
w.reserve(B, P, T)
eng = w.engine
g = torch.Generator(device=dev).manual_seed(42)
x = torch.randn(B, G, device=dev, generator=g)
patches = torch.randn(B, P, 1024, device=dev, generator=g)
text = torch.randn(B, T, Dt, device=dev, generator=g)
ppad = torch.zeros(B, P, dtype=torch.bool, device=dev)
tpad = torch.zeros(B, T, dtype=torch.bool, device=dev)


def one():
    w.train(x, text, tpad, patches, ppad)


for _ in range(3):
    one()
torch.cuda.synchronize()
eng.phase_enable(True)
acc, order = collections.defaultdict(list), []
tot = []
for _ in range(steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    one()
    e1.record()
    torch.cuda.synchronize()
    tot.append(e0.elapsed_time(e1))
    seen = collections.Counter()
    for name, ms in eng.phase_times():
        seen[name] += 1
        key = (name, seen[name])
        if key not in acc:
            order.append(key)
        acc[key].append(ms)
print(f"step {sum(tot) / len(tot):.3f} ms ({prec}, {steps} steps; the marks cost a few microseconds each)")
by = collections.defaultdict(float)
for key in order:
    m = sum(acc[key]) / len(acc[key])
    by[key[0]] += m
    print(f"  {m:7.3f} ms  {key[0]} #{key[1]}")
print("per phase kind:")
for k, v in by.items():
    print(f"  {v:7.3f} ms  {k}")
