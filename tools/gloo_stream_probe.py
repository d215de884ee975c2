"""What does a gloo all-reduce of a CUDA tensor cost when it is issued from a side stream?  (two ranks on ONE GPU; rehearsal of the
data-parallel host loop's pattern, model.py::_allreduce_stage.)  usage: python -m torch.distributed.run --nproc-per-node 2 tools/gloo_stream_probe.py [backend]"""
import os
import sys
import time
import torch
import torch.distributed as dist

backend = sys.argv[1] if len(sys.argv) > 1 else "gloo"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
rank = int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend)
buf = torch.ones(18 * 2 ** 20 // 4, device=dev)
side = torch.cuda.Stream(dev)
work = torch.zeros(1 << 24, device=dev)


def busy():
    for _ in range(4):
        work.mul_(1.0001)


def run(tag, numel, on_side, n=10):
    t = buf[:numel]
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        busy()
        if on_side:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                w = dist.all_reduce(t, async_op=True)
        else:
            w = dist.all_reduce(t, async_op=True)
        busy()
        w.wait()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    if rank == 0:
        print(f"{backend:5s} {tag:28s} {numel * 4 / 2**20:6.1f} MB  {dt:8.2f} ms per (work + all-reduce + work + wait)", flush=True)


for numel in (2 * 2 ** 20 // 4, 6 * 2 ** 20 // 4, 12 * 2 ** 20 // 4):
    run("current stream", numel, False)
    run("side stream after join", numel, True)
dist.destroy_process_group()
