"""Which hardware queue does each HIP stream land on?  Run under `rocprofv3 --kernel-trace` and read Stream_Id / Queue_Id of the
fill kernels (tools/queue_probe_summary.py prints the map).  usage: python tools/queue_probe.py [n_streams]"""
import sys
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
x = torch.zeros(1 << 20, device=dev)
x.add_(1.0)                                            # the current (default) stream
streams = [torch.cuda.Stream(dev, priority=0) for _ in range(n)] + [torch.cuda.Stream(dev, priority=-1) for _ in range(2)]
bufs = [torch.zeros(1 << 20, device=dev) for _ in streams]
torch.cuda.synchronize()
for rep in range(3):
    for s, b in zip(streams, bufs):
        with torch.cuda.stream(s):
            b.mul_(2.0)
torch.cuda.synchronize()
print("streams:", [hex(s.cuda_stream) for s in streams])
