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
