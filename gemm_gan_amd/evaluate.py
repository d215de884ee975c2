"""Device-side evaluation metrics on the HIP nearest-record kernels (csrc/evalnn.hip): drop-ins for the reference's privacy
metrics ``dcr`` / ``nndr`` (src/privacy_evaluator.py:9-66, P below; ``gg_eval_nn2``) and for ``compute_prdc``
(src/distribution_distances.py:102-142, R below; ``gg_eval_knn`` + ``gg_eval_prdc_counts``).  The reference builds a [128, N, G] difference tensor per batch
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


def kth_smallest(queries: torch.Tensor, refs: torch.Tensor, k: int, l1: bool = True) -> torch.Tensor:
    """[nq, k]: the k smallest distances of every query row to the rows of `refs`, ascending (L1 or Euclidean)."""
    if queries.device.type != "cuda" or refs.device != queries.device:
        raise RuntimeError("gemm_gan_amd.evaluate needs ROCm GPU tensors (there is no CPU fallback)")
    if queries.dim() != 2 or refs.dim() != 2 or queries.shape[1] != refs.shape[1] or not 1 <= k <= 16:
        raise ValueError("queries [nq, dim] / refs [nr, dim], 1 <= k <= 16 expected")
    lib = L.load()
    nq, nr, dim = queries.shape[0], refs.shape[0], queries.shape[1]
    with torch.cuda.device(queries.device):
        width = int(lib.gg_eval_knn_width(k))
        out = torch.empty(nq, width, device=queries.device)
        n = int(lib.gg_eval_knn_scratch(nq, nr, k))
        scratch = torch.empty(n, device=queries.device)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(lib.gg_eval_knn(C.c_void_p(queries.data_ptr()), nq, C.c_void_p(refs.data_ptr()), nr, dim, k, int(l1),
                                C.c_void_p(out.data_ptr()), C.c_void_p(scratch.data_ptr()), n, stream))
    return out[:, :k]


def compute_prdc(real_features, fake_features, nearest_k, device="cuda:0"):
    """Precision, recall, density, coverage (R:102-142) with the reference's L1 distances (R:64); radii = distance to the
    `nearest_k`-th neighbour, the sample itself counted as the zeroth (R:97-98).  The [real x fake] distance matrix is never
    materialised."""
    real, fake = _dev(real_features, device), _dev(fake_features, device)
    if not 1 <= nearest_k <= 15:
        raise ValueError("1 <= nearest_k <= 15")
    lib = L.load()
    rad_real = kth_smallest(real, real, nearest_k + 1)[:, nearest_k].contiguous()
    rad_fake = kth_smallest(fake, fake, nearest_k + 1)[:, nearest_k].contiguous()
    nr, nf = real.shape[0], fake.shape[0]
    with torch.cuda.device(real.device):
        below = torch.empty(nf, dtype=torch.int32, device=real.device)
        anyf = torch.empty(nr, dtype=torch.int32, device=real.device)
        mind = torch.empty(nr, dtype=torch.float32, device=real.device)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(lib.gg_eval_prdc_counts(C.c_void_p(real.data_ptr()), nr, C.c_void_p(fake.data_ptr()), nf, real.shape[1], 1,
                                        C.c_void_p(rad_real.data_ptr()), C.c_void_p(rad_fake.data_ptr()), C.c_void_p(below.data_ptr()),
                                        C.c_void_p(anyf.data_ptr()), C.c_void_p(mind.data_ptr()), stream))
    precision = (below > 0).double().mean().item()
    recall = (anyf > 0).double().mean().item()
    density = below.double().mean().item() / float(nearest_k)
    coverage = (mind < rad_real).double().mean().item()
    return dict(precision=precision, recall=recall, density=density, coverage=coverage)
