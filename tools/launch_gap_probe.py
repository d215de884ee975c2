"""GPU-side cost of a dependent kernel launch when the host is far ahead (no profiler attached): a long blocker kernel
fills the stream, N tiny kernels are enqueued behind it, events bracket the tiny ones.  usage: python tools/launch_gap_probe.py"""
import time
import torch
dev = torch.device("cuda:0")
big = torch.randn(8192, 8192, device=dev)
small = torch.zeros(256, device=dev)
mid = torch.zeros(1 << 20, device=dev)
torch.cuda.synchronize()
def run(n, op, label):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        big @ big                      # ~ 10 ms each: the host enqueues everything below while these run
    e0.record()
    t0 = time.perf_counter()
    for _ in range(n):
        op()
    host = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:40s} n={n}: GPU {e0.elapsed_time(e1) * 1e3 / n:6.2f} us per launch, host enqueue {host * 1e6 / n:6.2f} us per launch")
for n in (500, 2000):
    run(n, lambda: small.add_(1.0), "256-element add_ (1 workgroup)")
    run(n, lambda: mid.add_(1.0), "1M-element add_ (4 MB)")
s1 = torch.cuda.Stream()
def two_streams():
    small.add_(1.0)
run(1000, two_streams, "same again")
