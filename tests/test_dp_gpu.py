"""Data parallelism with the REAL engine on the GPU: two fresh child processes share GPU 0, all-reduce over `gloo`
(tests/dp_worker.py), each on its half of a global minibatch; after two train() steps both must hold the parameters one
process gets on the whole minibatch with the same noise (sliced z / alpha).  Exercises the two-phase backward,
the two gradient buckets, *_apply(1/N), the critic conditioning pass computed ahead, and the globally averaged losses.
The parent only starts children (it never replaces itself), the children initialise `gloo` before they touch the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from golden_util import Golden, comparable

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("fixture", ["xattn_film_T3"])
def test_two_ranks_on_one_gpu_equal_one_rank_on_the_global_batch(fixture, tmp_path):
    _two_ranks(fixture, tmp_path, "gloo")


@pytest.mark.timeout(600)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the RCCL (`nccl`) backend, one rank per device")
@pytest.mark.parametrize("fixture", ["xattn_film_T3"])
def test_two_ranks_on_two_gpus_over_rccl_equal_one_rank_on_the_global_batch(fixture, tmp_path):
    """The same assertion over the backend bench.py's N > 1 runs use: RCCL, rank r on GPU r, asynchronous bucket all-reduces on
    RCCL's own stream.  torch.cuda.device_count() above does not initialise the GPU; the children are started before the parent
    touches it and nothing replaces a running process."""
    _two_ranks(fixture, tmp_path, "nccl")


@pytest.mark.timeout(600)
@pytest.mark.parametrize("fixture", ["xattn_film_T3"])
def test_one_rank_over_rccl_runs_the_whole_host_loop(fixture, tmp_path):
    """RCCL itself, on the one GPU this pool's boxes have: a ONE-rank `nccl` process group (gemm_gan_amd.rccl_process_group_options) and
    GG_FORCE_DP_COLLECTIVES=1, under which the data-parallel host loop issues every all-reduce it issues on N ranks - per-stage
    backward calls, collectives from the engine's side stream, *_apply(1/1).  Transport is not exercised; RCCL's launch path, its
    stream and the event pattern between the three streams are.  The result must equal the single-call train() on the same batch."""
    _two_ranks(fixture, tmp_path, "nccl", world=1, extra_env={"GG_FORCE_DP_COLLECTIVES": "1", "GG_FORCE_DP_LOOP": "1"})


def _two_ranks(fixture, tmp_path, backend, world=2, extra_env=None):
    sys.path.insert(0, HERE)
    import dp_worker
    port = str(29600 + os.getpid() % 1500 + (7 if backend == "nccl" else 0))
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(r), str(world), port, fixture, outs[r], backend],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=480)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors="replace")[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)

    g = Golden(fixture)
    inputs, z, alpha = dp_worker.shard_inputs(g, world)
    w = dp_worker.build(g)
    ref_losses = dp_worker.run_steps(w, inputs, z, alpha, slice(0, world * g.dims["B"]))
    torch.cuda.synchronize()
    ref = {"g." + k: v.detach().cpu().numpy() for k, v in w.gen.state_dict().items()}
    ref.update({"d." + k: v.detach().cpu().numpy() for k, v in w.disc.state_dict().items()})
    got = [np.load(o) for o in outs]
    E = g.dims["E"]
    lr, steps = 5e-4, 2 * (g.dims["n_critic"] + 1)
    worst = 0.0
    for k, v in ref.items():
        if "patches_transformer_layer." in k:
            continue
        a, b = got[0][k], got[-1][k]
        assert np.array_equal(a, b), f"ranks diverged on {k}"                   # replicas stay bit-identical
        keep = comparable(k, v, E)
        d = np.abs(a.reshape(-1) - v.reshape(-1))[keep]
        # fp32 summation order differs between one batch of 2B rows and two of B (+ the host all-reduce): entries whose
        # gradient is rounding noise move by up to lr per RMSprop step in either run, everything else agrees to ~1e-4 relative
        bound = 2e-3 * max(float(np.abs(v).max()), 1e-3) + 1e-6
        n_bad = int((d > bound).sum())
        worst = max(worst, n_bad / d.size)
        assert d.max() <= steps * lr * 2.5 + bound, (k, float(d.max()))
        assert n_bad <= max(3, 0.05 * d.size), (k, n_bad, d.size)
    l0, l1 = got[0]["losses"], got[-1]["losses"]
    assert np.allclose(l0, l1, rtol=1e-6, atol=1e-7)                            # reported losses are global means
    assert np.allclose(l0, np.array(ref_losses), rtol=5e-3, atol=5e-4), (l0, ref_losses)
    assert float(got[0]["comm_wait_ms"]) >= 0.0
    assert str(got[0]["backend"]) == backend and int(got[0]["world"]) == world
    # kernel launches of the LAST train() only (the data-parallel loop resets the counter itself): the same on both ranks and
    # of the size of one step, not the running total of two
    la, lb = got[0]["launches"], got[-1]["launches"]
    assert np.array_equal(la, lb) and la[0] > 0 and la[1] <= la[0], (la, lb)     # (step 2 consumes a conditioning pass step 1 ran ahead)
