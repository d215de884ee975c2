"""Device-side evaluation metrics on the HIP nearest-record kernels (csrc/evalnn.hip): drop-ins for the reference's privacy
metrics ``dcr`` / ``nndr`` (src/privacy_evaluator.py:9-66, P below; ``gg_eval_nn2``) and for ``compute_prdc``
(src/distribution_distances.py:102-142, R below; ``gg_eval_knn`` + ``gg_eval_prdc_counts``), and for the improved
precision / recall of ``ManifoldEstimator`` / ``knn_precision_recall_features`` / ``get_precision_recall``
(src/unsupervised_metrics.py:141-303, U below; the same two kernels with Euclidean distances and inclusive radii).  The reference builds a [128, N, G] difference tensor per batch
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


def _prdc_counts(real, fake, rad_real, rad_fake, mode):
    lib = L.load()
    nr, nf = real.shape[0], fake.shape[0]
    with torch.cuda.device(real.device):
        below = torch.empty(nf, dtype=torch.int32, device=real.device)
        anyf = torch.empty(nr, dtype=torch.int32, device=real.device)
        mind = torch.empty(nr, dtype=torch.float32, device=real.device)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(lib.gg_eval_prdc_counts(C.c_void_p(real.data_ptr()), nr, C.c_void_p(fake.data_ptr()), nf, real.shape[1], mode,
                                        C.c_void_p(rad_real.data_ptr()), C.c_void_p(rad_fake.data_ptr()), C.c_void_p(below.data_ptr()),
                                        C.c_void_p(anyf.data_ptr()), C.c_void_p(mind.data_ptr()), stream))
    return below, anyf, mind


class ManifoldEstimator:
    """U:141-244.  The manifold of `features` = union of hyperspheres around every sample whose radius reaches its
    `nhood_sizes[0]`-th neighbour (the sample itself is the zeroth, U:188-189).  `D` holds the SQUARED radii like the
    reference (its `batch_pairwise_distances` returns squared distances, U:114-138); `evaluate` says for every new vector
    whether it falls inside (<=, U:223) at least one hypersphere.  One neighbourhood size per estimator (the reference's
    callers pass a single one, U:297); row / column batch sizes are accepted and ignored (nothing is materialised)."""

    def __init__(self, features, row_batch_size=25000, col_batch_size=50000, nhood_sizes=(3,), clamp_to_percentile=None, eps=1e-5,
                 device="cuda:0"):
        if len(nhood_sizes) != 1 or not 1 <= int(nhood_sizes[0]) <= 15:
            raise ValueError("one neighbourhood size in [1, 15] per estimator")
        if clamp_to_percentile is not None:
            raise NotImplementedError("clamp_to_percentile is never used by the reference's callers")
        self.nhood_sizes = list(nhood_sizes)
        self.eps = eps
        self._ref = _dev(features, device)
        k = int(nhood_sizes[0])
        self._radius = kth_smallest(self._ref, self._ref, k + 1, l1=False)[:, k].contiguous()      # Euclidean
        self.D = (self._radius * self._radius).reshape(-1, 1).cpu().numpy()

    def evaluate(self, eval_features):
        """[n_eval, 1] int32: 1 where the vector lies inside the manifold."""
        ev = _dev(eval_features, self._ref.device)
        dummy = torch.zeros(ev.shape[0], device=ev.device)          # radii of the evaluated side are not needed here
        below, _, _ = _prdc_counts(self._ref, ev, self._radius, dummy, 2)
        return (below > 0).to(torch.int32).reshape(-1, 1).cpu().numpy()


def knn_precision_recall_features(ref_features, eval_features, nhood_sizes=(3,), row_batch_size=10000, col_batch_size=50000,
                                  num_gpus=1, device="cuda:0"):
    """U:247-297: precision = share of `eval_features` inside the manifold of `ref_features`, recall = the converse.  Both
    come out of ONE counting pass over the never-materialised [ref x eval] distance matrix."""
    if len(nhood_sizes) != 1 or not 1 <= int(nhood_sizes[0]) <= 15:
        raise ValueError("one neighbourhood size in [1, 15]")
    k = int(nhood_sizes[0])
    ref, ev = _dev(ref_features, device), _dev(eval_features, device)
    rad_ref = kth_smallest(ref, ref, k + 1, l1=False)[:, k].contiguous()
    rad_ev = kth_smallest(ev, ev, k + 1, l1=False)[:, k].contiguous()
    below, anyf, _ = _prdc_counts(ref, ev, rad_ref, rad_ev, 2)
    return {"precision": np.array([(below > 0).double().mean().item()]), "recall": np.array([(anyf > 0).double().mean().item()])}


def get_precision_recall(real_data, fake_data, nb_nn=(10,), device="cuda:0"):
    """U:300-323: (precision, recall) of `fake_data` against `real_data`."""
    state = knn_precision_recall_features(real_data, fake_data, nhood_sizes=list(nb_nn), device=device)
    return state["precision"][0], state["recall"][0]
