"""gemm_gan_amd/evaluate.py (HIP nearest-record kernel) against the reference's arithmetic for DCR / NNDR
(src/privacy_evaluator.py:9-66, restated here with the same torch expressions on the CPU)."""
import numpy as np
import pytest
import torch

from gemm_gan_amd import evaluate

pytestmark = pytest.mark.gpu


def _ref_dist(a, b):          # P:22 / P:49: (a[:, None] - b).pow(2).sum(dim=2).sqrt()
    return (a[:, None] - b).pow(2).sum(dim=2).sqrt()


def _ref_scores(real, gen, test):
    dr, dt = _ref_dist(gen, real), _ref_dist(gen, test)
    dcr = (dr.min(dim=1).values < dt.min(dim=1).values).nonzero().shape[0] / gen.shape[0]
    sr, st = torch.sort(dr, dim=1)[0], torch.sort(dt, dim=1)[0]
    nndr = ((sr[:, 0] / sr[:, 1]) < (st[:, 0] / st[:, 1])).nonzero().shape[0] / gen.shape[0]
    return dcr, nndr, sr, st


@pytest.mark.parametrize("shape", [(70, 333, 41, 129), (1, 64, 2, 16), (257, 1000, 513, 300), (130, 65, 1000, 5000)])
def test_nearest_two_and_scores_match_the_reference_arithmetic(shape):
    nq, nr, nt, dim = shape
    g = torch.Generator().manual_seed(nq + dim)
    real = torch.randn(nr, dim, generator=g)
    test = torch.randn(nt, dim, generator=g)
    gen = torch.randn(nq, dim, generator=g)
    gen[0] = real[min(3, nr - 1)]                                     # an exact copy of a training record: distance 0
    if nq > 2:
        gen[2] = real[0] + 1e-3 * torch.randn(dim, generator=g)        # a near copy: the case the metric exists for
    d1, d2 = evaluate.nearest2(gen.cuda(), real.cuda())
    dcr_ref, nndr_ref, sr, st = _ref_scores(real, gen, test)
    assert d1[0].item() == 0.0
    assert torch.allclose(d1.cpu(), sr[:, 0], rtol=1e-5, atol=1e-6) and torch.allclose(d2.cpu(), sr[:, 1], rtol=1e-5, atol=1e-6)
    assert evaluate.dcr(real.numpy(), gen.numpy(), test.numpy()) == pytest.approx(dcr_ref, abs=1.5 / nq)
    assert evaluate.nndr(real.numpy(), gen.numpy(), test.numpy()) == pytest.approx(nndr_ref, abs=1.5 / nq)


def test_single_reference_row_and_bad_arguments():
    q = torch.randn(5, 8).cuda()
    d1, d2 = evaluate.nearest2(q, q[:1].contiguous())
    assert d1[0].item() == 0.0 and torch.isinf(d2).all()
    with pytest.raises(ValueError):
        evaluate.nearest2(q, torch.randn(4, 7).cuda())
    with pytest.raises(RuntimeError):
        evaluate.nearest2(q.cpu(), q.cpu())


def _ref_prdc(real, fake, k):
    """src/distribution_distances.py:87-142 restated with scipy (L1 distances, float64)."""
    from scipy.spatial.distance import cdist
    def radii(x):
        d = cdist(x, x, "cityblock")
        return np.sort(d, axis=1)[:, k]                       # get_kth_value(d, k + 1): the (k+1)-th smallest incl. the 0 of the sample itself
    rr, rf = radii(real), radii(fake)
    d = cdist(real, fake, "cityblock")
    return dict(precision=(d < rr[:, None]).any(axis=0).mean(), recall=(d < rf[None, :]).any(axis=1).mean(),
                density=(1.0 / k) * (d < rr[:, None]).sum(axis=0).mean(), coverage=(d.min(axis=1) < rr).mean()), rr


@pytest.mark.parametrize("shape,k", [((150, 97, 33), 5), ((64, 200, 130), 3), ((300, 310, 700), 10), ((40, 33, 20), 15)])
def test_prdc_matches_the_reference_arithmetic(shape, k):
    nr, nf, dim = shape
    rng = np.random.default_rng(nr + k)
    real = rng.standard_normal((nr, dim))
    fake = 0.8 * rng.standard_normal((nf, dim)) + 0.3          # overlapping, not identical manifolds: every metric strictly inside (0, 1)
    ref, rr = _ref_prdc(real, fake, k)
    radii = evaluate.kth_smallest(torch.tensor(real, dtype=torch.float32).cuda(), torch.tensor(real, dtype=torch.float32).cuda(), k + 1)
    assert radii[:, 0].abs().max().item() < 1e-4                                       # the sample itself
    assert np.allclose(radii[:, k].cpu().numpy(), rr, rtol=2e-5)
    got = evaluate.compute_prdc(real, fake, k)
    tol = 2.5 / min(nr, nf)                                      # a comparison at a float32 / float64 tie may fall either way
    for name in ("precision", "recall", "density", "coverage"):
        assert abs(got[name] - ref[name]) <= tol * max(1.0, ref[name]), (name, got[name], ref[name])


# ---- pinned by outputs of the REAL reference functions (tests/golden/aux_*.npz, oracle/make_golden_aux.py) -------------------
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_prdc_equals_the_recorded_reference_values():
    """src/distribution_distances.py:102-142 compute_prdc run by the reference itself on the stored features."""
    z = np.load(os.path.join(GOLDEN, "aux_prdc.npz"))
    for k in (1, 5, 10):
        got = evaluate.compute_prdc(z["real"], z["fake"], k)
        want = dict(zip(("precision", "recall", "density", "coverage"), z[f"k{k}"]))
        for name, w in want.items():
            assert abs(got[name] - w) <= 1e-9 + 1e-12 * abs(w), (k, name, got[name], w)      # counts: exact


def test_knn_precision_recall_equals_the_recorded_reference_values():
    """src/unsupervised_metrics.py:141-323: ManifoldEstimator radii / membership and get_precision_recall."""
    z = np.load(os.path.join(GOLDEN, "aux_knn_pr.npz"))
    for k in (3, 10):
        est = evaluate.ManifoldEstimator(z["real"], nhood_sizes=[k])
        assert np.allclose(est.D, z[f"k{k}/radii_real"], rtol=2e-4, atol=1e-5)             # squared radii (the reference's
        pred = est.evaluate(z["fake"])                                                      # |u|^2-2uv+|v|^2 form cancels in fp32)
        assert np.array_equal(pred, z[f"k{k}/fake_in_real_manifold"])
        p, r = evaluate.get_precision_recall(z["real"], z["fake"], nb_nn=[k])
        assert (p, r) == pytest.approx(tuple(z[f"k{k}/precision_recall"]), abs=1e-12)
        st = evaluate.knn_precision_recall_features(z["real"], z["fake"], nhood_sizes=[k])
        assert st["precision"].shape == (1,) and st["recall"].shape == (1,)


@pytest.mark.parametrize("case", ["small", "one_batch", "multi_batch"])
def test_dcr_nndr_equal_the_recorded_reference_values(case):
    """src/privacy_evaluator.py:9-66 dcr / nndr run by the reference itself (oracle/make_golden_aux.py::privacy; its `.cuda()`
    calls were no-ops in the GPU-less build container): the scores, and the per-sample nearest / second-nearest distances
    behind them, against the HIP nearest-record kernel."""
    z = np.load(os.path.join(GOLDEN, "aux_privacy.npz"))
    real, test, gen = (z[f"{case}/{k}"] for k in ("real", "test", "gen"))
    want_dcr, want_nndr = z[f"{case}/scores"]
    r1, r2 = evaluate.nearest2(torch.from_numpy(gen).cuda(), torch.from_numpy(real).cuda())
    t1, t2 = evaluate.nearest2(torch.from_numpy(gen).cuda(), torch.from_numpy(test).cuda())
    assert np.allclose(r1.cpu().numpy(), z[f"{case}/dcr_real"], rtol=1e-5, atol=1e-6)
    assert np.allclose(t1.cpu().numpy(), z[f"{case}/dcr_test"], rtol=1e-5, atol=1e-6)
    # ratios: 0 / d2 for the exact copies on both sides; elsewhere fp32 rounding of two square roots
    assert np.allclose((r1 / r2).cpu().numpy(), z[f"{case}/nndr_real"], rtol=2e-5, atol=1e-6)
    assert np.allclose((t1 / t2).cpu().numpy(), z[f"{case}/nndr_test"], rtol=2e-5, atol=1e-6)
    # the scores are counts of strict comparisons; no pair of this fixture sits within fp32 rounding of a tie
    margin = np.abs(z[f"{case}/dcr_real"] - z[f"{case}/dcr_test"]).min()
    assert margin > 1e-4
    assert evaluate.dcr(real, gen, test) == pytest.approx(float(want_dcr), abs=1e-12)
    assert evaluate.nndr(real, gen, test) == pytest.approx(float(want_nndr), abs=1e-12)
