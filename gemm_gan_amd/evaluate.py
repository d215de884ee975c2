"""Device-side privacy metrics: drop-in for the reference's ``dcr`` / ``nndr`` (src/privacy_evaluator.py:9-66, P below) on the
HIP nearest-record kernel (``gg_eval_nn2``, csrc/evalnn.hip).  The reference builds a [128, N, G] difference tensor per batch
and sorts every row; the kernel streams the gene dimension through LDS and keeps the two smallest distances per generated
sample.  Same arguments (numpy arrays or tensors), same return value (a Python float)."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _dev(a, device):
    t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
    return t.to(device, torch.float32).contiguous()


def nearest2(queries: torch.Tensor, refs: torch.Tensor):
    """(d1, d2): Euclidean distance of every row of `queries` to its nearest and second nearest row of `refs`."""
    if queries.device.type != "cuda" or refs.device != queries.device:
        raise RuntimeError("gemm_gan_amd.evaluate needs ROCm GPU tensors (there is no CPU fallback)")
    if queries.dim() != 2 or refs.dim() != 2 or queries.shape[1] != refs.shape[1] or queries.shape[0] == 0 or refs.shape[0] == 0:
        raise ValueError("queries [nq, dim] / refs [nr, dim] expected")
    lib = L.load()
    nq, nr, dim = queries.shape[0], refs.shape[0], queries.shape[1]
    with torch.cuda.device(queries.device):
        d1 = torch.empty(nq, device=queries.device)
        d2 = torch.empty(nq, device=queries.device)
        n = int(lib.gg_eval_nn2_scratch(nq, nr))
        scratch = torch.empty(n, device=queries.device)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(lib.gg_eval_nn2(C.c_void_p(queries.data_ptr()), nq, C.c_void_p(refs.data_ptr()), nr, dim, C.c_void_p(d1.data_ptr()),
                                C.c_void_p(d2.data_ptr()), C.c_void_p(scratch.data_ptr()), n, stream))
    return d1, d2


def dcr(real_data, gen_data, test_data, batch_size=128, device="cuda:0"):
    """Distance to closest record (P:9-33): share of generated samples closer to a training record than to a test record.
    `batch_size` is accepted for signature compatibility (the kernel does not batch)."""
    syn = _dev(gen_data, device)
    d_real, _ = nearest2(syn, _dev(real_data, device))
    d_test, _ = nearest2(syn, _dev(test_data, device))
    return int((d_real < d_test).sum().item()) / syn.shape[0]


def nndr(real_data, gen_data, test_data, batch_size=128, device="cuda:0"):
    """Nearest-neighbour distance ratio (P:35-66): first / second neighbour distance, training set against test set."""
    syn = _dev(gen_data, device)
    r1, r2 = nearest2(syn, _dev(real_data, device))
    t1, t2 = nearest2(syn, _dev(test_data, device))
    return int(((r1 / r2) < (t1 / t2)).sum().item()) / syn.shape[0]
