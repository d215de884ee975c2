"""Micro-benchmark of the GEMM kernels on hot-path shapes (run on the GPU box)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemm_gan_amd import _lib as L

lib = L.load()


def run(M, N, K, la, lb, kernel, sk=1, iters=10):
    A = torch.randn((M, K) if la == 0 else (K, M), device="cuda")
    B = torch.randn((N, K) if lb == 0 else (K, N), device="cuda")
    Cm = torch.zeros(M, N, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fn = getattr(lib, kernel)
    def call():
        rc = fn(C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), M, N, K, A.stride(0), B.stride(0), N,
                la, lb, sk, C.c_float(1.0), None, 0, C.c_float(0.0), 0, st)
        assert rc == 0
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    byt = 4.0 * (M * K + N * K + M * N)
    print(f"{kernel:18s} M={M:7d} N={N:5d} K={K:5d} la{la} lb{lb} sk{sk:3d}: {us:8.1f} us  {byt / us / 1e6:7.2f} TB/s  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    for kern in ("gg_test_gemm_bf16", "gg_test_gemm"):
        run(197376, 768, 256, 0, 0, kern)
        run(197376, 128, 256, 0, 0, kern)
        run(197376, 256, 256, 0, 0, kern)
        run(197376, 256, 512, 0, 0, kern)
        run(131584, 256, 768, 0, 1, kern)
        run(768, 256, 131584, 1, 1, kern, sk=43)
        run(256, 256, 256, 0, 0, kern)
        run(512, 256, 5000, 0, 0, kern, sk=20)
    # plain copy for reference
    x = torch.randn(197376 * 1024, device="cuda"); y = torch.empty_like(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    y.copy_(x); e0.record()
    for _ in range(10): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"torch copy 808 MB r + 808 MB w: {us:.1f} us  {2 * x.numel() * 4 / us / 1e6:.2f} TB/s")
