"""How long does the HOST need to enqueue one train() (no synchronisation inside), against the GPU's step time?
usage: python tools/host_enqueue_probe.py [steps]"""
import sys, time
sys.path.insert(0, ".")
import torch
import gemm_gan_amd as gga
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
G, B, P, T = 5000, 256, 256, 1
torch.manual_seed(42)
w = gga.WGAN_GP(G, 256, 256, [256, 256, G], [256, 256, 1], text_embedding_dims=512, patches_embedding_dims=1024, optimizer="rms_prop",
                n_critic=5, dropout=0.1, seed=1, device=dev, results_dire="", precision="bf16")
w.build_WGAN_GP(); w.init_train(); w.reserve(B, P, T)
g = torch.Generator(device=dev).manual_seed(7)
x = torch.randn(B, G, device=dev, generator=g); patches = torch.randn(B, P, 1024, device=dev, generator=g); text = torch.randn(B, T, 512, device=dev, generator=g)
pp = torch.zeros(B, P, dtype=torch.bool, device=dev); tp = torch.zeros(B, T, dtype=torch.bool, device=dev)
for _ in range(3):
    w.train(x, text, tp, patches, pp)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(steps):
    a = time.perf_counter()
    w.train(x, text, tp, patches, pp)
    host.append((time.perf_counter() - a) * 1e3)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue per train(): " + " ".join(f"{h:.1f}" for h in host) + " ms")
print(f"host total {t_enq * 1e3 / steps:.2f} ms/step, wall incl. GPU {t_all * 1e3 / steps:.2f} ms/step")
# split: python-side noise generation vs the C call
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(4):
    w.train(x, text, tp, patches, pp)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
