import sys, os, torch, numpy as np
sys.path.insert(0, '.')
import gemm_gan_amd as gga
dev = torch.device("cuda:0")
G, B, P, T = 5000, 256, 256, 1
def run(prec, steps=25):
    torch.manual_seed(42)
    w = gga.WGAN_GP(G, 256, 256, [256, 256, G], [256, 256, 1], text_embedding_dims=512, patches_embedding_dims=1024,
                    optimizer="rms_prop", n_critic=5, dropout=0.1, seed=1, device=dev, results_dire="", precision=prec)
    w.build_WGAN_GP(); w.init_train(); w.reserve(B, P, T)
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(B, G, device=dev, generator=g); patches = torch.randn(B, P, 1024, device=dev, generator=g); text = torch.randn(B, T, 512, device=dev, generator=g)
    pp = torch.zeros(B, P, dtype=torch.bool, device=dev); tp = torch.zeros(B, T, dtype=torch.bool, device=dev)
    torch.manual_seed(123)
    out = []
    for s in range(steps):
        w.train(x, text, tp, patches, pp)
        if s % 4 == 0 or s == steps - 1:
            out.append((s, float(w.d_batch_loss[0]), float(w.g_batch_loss[0]), float(w.gp_value)))
    return out
a = run("bf16"); b = run("f32")
for (s, d1, g1, p1), (_, d2, g2, p2) in zip(a, b):
    print(f"step {s:3d}  bf16: d {d1:10.3f} g {g1:9.3f} gp {p1:8.4f}   f32: d {d2:10.3f} g {g2:9.3f} gp {p2:8.4f}")
