"""Unit parity of the MFMA GEMM kernel (all layout combinations, ragged sizes, split-K, epilogues)
through the C ABI (gg_test_gemm) against torch fp64 matmul.  Exact-integer operands with an
ASYMMETRIC pattern catch swapped fragment maps that random data would blur."""
import ctypes as C

import pytest
import torch

from gemm_gan_amd import _lib as L
from gpu_util import Checker

pytestmark = pytest.mark.gpu


def run_gemm(A, B, M, N, K, layA, layB, splitk=1, alpha=1.0, bias=None, act=0, slope=0.0, C0=None, accumulate=0,
             kernel="gg_test_gemm"):
    lib = L.load()
    out = torch.zeros(M, N, device="cuda") if C0 is None else C0.clone()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = getattr(lib, kernel)(C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(out.data_ptr()), M, N, K,
                          A.stride(0), B.stride(0), N, layA, layB, splitk, C.c_float(alpha),
                          None if bias is None else C.c_void_p(bias.data_ptr()), act, C.c_float(slope), accumulate, st)
    assert rc == 0, lib.gg_last_error()
    torch.cuda.synchronize()
    return out


def operands(M, N, K, layA, layB, integer):
    g = torch.Generator().manual_seed(M * 7919 + N * 31 + K)
    if integer:
        Am = torch.randint(-3, 4, (M, K), generator=g).float() + (torch.arange(M)[:, None] % 5 == 0).float()
        Bm = torch.randint(-3, 4, (K, N), generator=g).float() + (torch.arange(N)[None, :] % 7 == 0).float() * 2
    else:
        Am = torch.randn(M, K, generator=g)
        Bm = torch.randn(K, N, generator=g)
    A = (Am if layA == L.LAY_KC else Am.t()).contiguous().cuda()      # KC: [M,K]; KS: [K,M]
    B = (Bm.t() if layB == L.LAY_KC else Bm).contiguous().cuda()      # KC: [N,K]; KS: [K,N]
    return Am, Bm, A, B


@pytest.mark.parametrize("layA", [0, 1])
@pytest.mark.parametrize("layB", [0, 1])
def test_gemm_layouts_exact_integer(layA, layB):
    ck = Checker(f"gemm exact integer layA={layA} layB={layB}", 0.0, metric="max")
    for (M, N, K) in [(32, 32, 8), (128, 128, 32), (200, 136, 72), (257, 64, 257), (5, 37, 69), (300, 260, 100)]:
        Am, Bm, A, B = operands(M, N, K, layA, layB, True)
        out = run_gemm(A, B, M, N, K, layA, layB)
        ck.check(f"{M}x{N}x{K}", out, (Am.double() @ Bm.double()).float())
    ck.done()


@pytest.mark.parametrize("layA,layB", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_random_splitk_and_epilogues(layA, layB):
    ck = Checker(f"gemm random/splitk/epilogue layA={layA} layB={layB}", 2e-5, metric="max")
    M, N, K = 260, 200, 1000
    Am, Bm, A, B = operands(M, N, K, layA, layB, False)
    ref = Am.double() @ Bm.double()
    ck.check("plain", run_gemm(A, B, M, N, K, layA, layB), ref)
    ck.check("splitk7", run_gemm(A, B, M, N, K, layA, layB, splitk=7), ref)
    bias = torch.randn(N, device="cuda")
    C0 = torch.randn(M, N, device="cuda")
    want = 0.5 * ref + bias.double().cpu() + C0.double().cpu()
    want = torch.where(want > 0, want, 0.1 * want)
    ck.check("alpha+bias+accumulate+leaky", run_gemm(A, B, M, N, K, layA, layB, alpha=0.5, bias=bias, act=1, slope=0.1,
                                                     C0=C0, accumulate=1), want)
    ck.check("splitk accumulates onto C", run_gemm(A, B, M, N, K, layA, layB, splitk=4, C0=C0, accumulate=1),
             ref + C0.double().cpu())
    ck.done()


def test_gemm_hot_path_shapes():
    """cfg3 shapes: QKV projection slice, critic first layer (K=5000, split-K), weight-gradient reduction."""
    ck = Checker("gemm hot-path shapes", 3e-5, metric="max")
    for (M, N, K, la, lb, sk) in [(2570, 768, 256, 0, 0, 1), (512, 256, 5000, 0, 0, 20), (768, 256, 4112, 1, 1, 16),
                                  (512, 5000, 256, 0, 1, 1)]:
        Am, Bm, A, B = operands(M, N, K, la, lb, False)
        ck.check(f"{M}x{N}x{K} la{la} lb{lb} sk{sk}", run_gemm(A, B, M, N, K, la, lb, splitk=sk), Am.double() @ Bm.double())
    ck.done()


@pytest.mark.parametrize("layA", [0, 1])
@pytest.mark.parametrize("layB", [0, 1])
def test_gemm_bf16_layouts_exact_integer(layA, layB):
    """bf16 MFMA twin: small integers are exact in bf16 and their sums exact in the fp32 accumulator, so
    the fragment / transpose-staging maps are checked bit for bit."""
    ck = Checker(f"gemm_bf16 exact integer layA={layA} layB={layB}", 0.0, metric="max")
    for (M, N, K) in [(32, 32, 16), (128, 128, 64), (200, 136, 72), (257, 64, 257), (5, 37, 69), (300, 260, 200)]:
        Am, Bm, A, B = operands(M, N, K, layA, layB, True)
        out = run_gemm(A, B, M, N, K, layA, layB, kernel="gg_test_gemm_bf16")
        ck.check(f"{M}x{N}x{K}", out, (Am.double() @ Bm.double()).float())
    ck.done()


@pytest.mark.parametrize("layA,layB", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_bf16_random(layA, layB):
    """Random operands: the result must equal the fp64 product of the bf16-ROUNDED operands to fp32
    accumulation accuracy (this isolates kernel errors from the intended operand rounding)."""
    ck = Checker(f"gemm_bf16 random/splitk/epilogue layA={layA} layB={layB}", 2e-5, metric="max")
    M, N, K = 260, 200, 1000
    Am, Bm, A, B = operands(M, N, K, layA, layB, False)
    ref = Am.bfloat16().double() @ Bm.bfloat16().double()
    ck.check("plain", run_gemm(A, B, M, N, K, layA, layB, kernel="gg_test_gemm_bf16"), ref)
    ck.check("splitk5", run_gemm(A, B, M, N, K, layA, layB, splitk=5, kernel="gg_test_gemm_bf16"), ref)
    bias = torch.randn(N, device="cuda")
    C0 = torch.randn(M, N, device="cuda")
    want = 0.5 * ref + bias.double().cpu() + C0.double().cpu()
    want = torch.where(want > 0, want, 0.1 * want)
    ck.check("alpha+bias+accumulate+leaky", run_gemm(A, B, M, N, K, layA, layB, alpha=0.5, bias=bias, act=1, slope=0.1,
                                                     C0=C0, accumulate=1, kernel="gg_test_gemm_bf16"), want)
    ck.done()


@pytest.mark.parametrize("layA", [0, 1])
@pytest.mark.parametrize("layB", [0, 1])
def test_gemm_small_layouts_exact_integer(layA, layB):
    """64x64-tile / one-shot-K kernel used for the few-tile products: exact-integer check of every layout,
    ragged sizes, K spanning several 256-deep slabs, split-K, epilogue."""
    ck = Checker(f"gemm_small exact integer layA={layA} layB={layB}", 0.0, metric="max")
    for (M, N, K, sk) in [(32, 32, 16, 1), (64, 64, 256, 1), (200, 136, 72, 1), (257, 64, 600, 1), (5, 37, 69, 1), (256, 256, 1000, 4)]:
        Am, Bm, A, B = operands(M, N, K, layA, layB, True)
        out = run_gemm(A, B, M, N, K, layA, layB, splitk=sk, kernel="gg_test_gemm_small")
        ck.check(f"{M}x{N}x{K} sk{sk}", out, (Am.double() @ Bm.double()).float())
    M, N, K = 130, 70, 300
    Am, Bm, A, B = operands(M, N, K, layA, layB, True)
    bias = torch.arange(N, device="cuda").float()
    C0 = torch.ones(M, N, device="cuda")
    want = 2.0 * (Am.double() @ Bm.double()) + bias.double().cpu() + 1.0
    want = torch.where(want > 0, want, 0.5 * want)
    ck.check("alpha+bias+accumulate+leaky", run_gemm(A, B, M, N, K, layA, layB, alpha=2.0, bias=bias, act=1, slope=0.5, C0=C0,
                                                     accumulate=1, kernel="gg_test_gemm_small"), want.float())
    ck.done()


@pytest.mark.parametrize("a_bf16,b_bf16", [(1, 1), (1, 0), (0, 1)])
def test_gemm_bf16_stored_operands_exact_integer(a_bf16, b_bf16):
    """dW = dY^T X with K-strided operands stored as bf16 (engine: weight gradients of bf16-stored branch gradients when the
    reduction is too short for the token-reduction kernel): ragged reduction length, split-K, ragged tiles."""
    lib = L.load()
    ck = Checker(f"gemm bf16-stored operands a={a_bf16} b={b_bf16}", 0.0, metric="max")
    for (M, N, K, sk) in [(128, 128, 64, 1), (768, 256, 1542, 4), (256, 512, 771, 3), (72, 40, 130, 1), (512, 256, 1028, 16)]:
        g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
        Am = torch.randint(-3, 4, (K, M), generator=g).float() + (torch.arange(M)[None, :] % 5 == 0).float()
        Bm = torch.randint(-3, 4, (K, N), generator=g).float() + (torch.arange(N)[None, :] % 7 == 0).float() * 2
        A = (Am.bfloat16() if a_bf16 else Am).contiguous().cuda()
        B = (Bm.bfloat16() if b_bf16 else Bm).contiguous().cuda()
        out = torch.zeros(M, N, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.gg_test_gemm_bf16_stored(C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(out.data_ptr()), M, N, K,
                                          M, N, N, a_bf16, b_bf16, sk, st)
        assert rc == 0, lib.gg_last_error()
        torch.cuda.synchronize()
        ck.check(f"{M}x{N}x{K} sk{sk}", out, (Am.double().t() @ Bm.double()).float())
    ck.done()
