"""Host enqueue time of one cfg3 train() against its GPU time: is the data-parallel host loop (per-stage calls + collectives) host-bound?
usage: [GG_FORCE_DP_COLLECTIVES=1 [GG_DP_TWO_BUCKETS=1] | GG_FORCE_DP_LOOP=1] python tools/host_time_probe.py [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import gemm_gan_amd as gga

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if os.environ.get("GG_FORCE_DP_COLLECTIVES") == "1":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29656")
    os.environ["GG_FORCE_DP_LOOP"] = "1"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B, G, P, T, Dt = 256, 5000, 256, 1, 512
torch.manual_seed(42)
w = gga.WGAN_GP(G, 256, 256, [256, 256, G], [256, 256, 1], text_embedding_dims=Dt, patches_embedding_dims=1024, optimizer="rms_prop",
                n_critic=5, dropout=0.1, seed=1234, device=dev, results_dire="", precision="bf16")
w.build_WGAN_GP()
w.init_train()
w.reserve(B, P, T)
g = torch.Generator(device=dev).manual_seed(42)
x = torch.randn(B, G, device=dev, generator=g)
patches = torch.randn(B, P, 1024, device=dev, generator=g)
text = torch.randn(B, T, Dt, device=dev, generator=g)
ppad = torch.zeros(B, P, dtype=torch.bool, device=dev)
tpad = torch.zeros(B, T, dtype=torch.bool, device=dev)
for _ in range(3):
    w.train(x, text, tpad, patches, ppad)
torch.cuda.synchronize()
host = 0.0
t0 = time.perf_counter()
for _ in range(steps):
    torch.cuda.synchronize()            # the host starts every step with an empty queue: its enqueue time is then visible by itself
    a = time.perf_counter()
    w.train(x, text, tpad, patches, ppad)
    host += time.perf_counter() - a
torch.cuda.synchronize()
wall = time.perf_counter() - t0
mode = "collectives" if dist.is_initialized() else ("loop" if os.environ.get("GG_FORCE_DP_LOOP") == "1" else "single call")
print(f"{mode:12s} two_buckets={os.environ.get('GG_DP_TWO_BUCKETS', '0')}: host enqueue {host / steps * 1e3:6.2f} ms per train(), step (synchronised each) {wall / steps * 1e3:6.2f} ms")
if dist.is_initialized():
    dist.destroy_process_group()
