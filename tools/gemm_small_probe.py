"""Latency of the small-GEMM kernel on the head shapes, host far ahead (queued behind a blocker), vs a trivial kernel.
usage: python tools/gemm_small_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gemm_gan_amd import _lib as L
lib = L.load()
big = torch.randn(8192, 8192, device="cuda")
def run(M, N, K, la, lb, sk=1, n=300):
    A = torch.randn((M, K) if la == 0 else (K, M), device="cuda")
    B = torch.randn((N, K) if lb == 0 else (K, N), device="cuda")
    Cm = torch.zeros(M, N, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def call():
        rc = lib.gg_test_gemm_small(C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), M, N, K, A.stride(0), B.stride(0), N,
                                    la, lb, sk, C.c_float(1.0), None, 0, C.c_float(0.0), 0, st)
        assert rc == 0
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        big @ big
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(f"gemm_small M={M:5d} N={N:5d} K={K:5d} la{la} lb{lb} sk{sk:2d}: {e0.elapsed_time(e1) * 1e3 / n:7.2f} us per dependent launch", flush=True)
for shape in ((768, 256, 256, 0, 0), (768, 256, 256, 0, 1), (256, 256, 768, 1, 1), (768, 512, 256, 0, 0), (192, 256, 1000, 0, 0), (64, 256, 256, 0, 0),
              (768, 256, 512, 0, 0), (768, 256, 1024, 0, 0, 4), (256, 5000, 256, 0, 0)):
    run(*shape)
