"""CPU check of the evaluation fixtures recorded from the reference's own functions (oracle/make_golden_aux.py): a float64
numpy restatement of src/privacy_evaluator.py:9-66 reproduces the recorded per-sample vectors and scores, i.e. the fixture
is what the GPU test (tests/test_evaluate_gpu.py) believes it is."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _two_nearest(q, r):
    d = np.sqrt(((q[:, None, :].astype(np.float64) - r[None].astype(np.float64)) ** 2).sum(-1))
    d.sort(axis=1)
    return d[:, 0], d[:, 1]


@pytest.mark.parametrize("case", ["small", "one_batch", "multi_batch"])
def test_privacy_fixture_is_the_dcr_nndr_arithmetic(case):
    z = np.load(os.path.join(GOLDEN, "aux_privacy.npz"))
    real, test, gen = (z[f"{case}/{k}"] for k in ("real", "test", "gen"))
    r1, r2 = _two_nearest(gen, real)
    t1, t2 = _two_nearest(gen, test)
    assert np.allclose(r1, z[f"{case}/dcr_real"], rtol=1e-5, atol=1e-6) and np.allclose(t1, z[f"{case}/dcr_test"], rtol=1e-5, atol=1e-6)
    assert np.allclose(r1 / r2, z[f"{case}/nndr_real"], rtol=2e-5, atol=1e-6)
    assert np.allclose(t1 / t2, z[f"{case}/nndr_test"], rtol=2e-5, atol=1e-6)
    dcr, nndr = z[f"{case}/scores"]
    assert (r1 < t1).mean() == pytest.approx(float(dcr), abs=1e-12)
    assert ((r1 / r2) < (t1 / t2)).mean() == pytest.approx(float(nndr), abs=1e-12)
    assert r1[0] == 0.0 and t1[5] == 0.0          # the exact copies of a training / a test record
