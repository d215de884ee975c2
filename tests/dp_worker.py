"""One rank of the data-parallel GPU test (tests/test_dp_gpu.py starts two of these as fresh child processes that share
GPU 0 and talk over `gloo`): the REAL engine, split backward / bucketed all-reduce / apply, two train() steps - the second
one consumes the critic conditioning pass computed ahead under the first one's generator all-reduce.

    python tests/dp_worker.py <rank> <world> <port> <fixture> <out.npz> [backend]

backend "gloo" (default): every rank on GPU 0, host all-reduce.  backend "nccl" (= RCCL on ROCm): rank r on GPU r, the
collectives the bench's N > 1 runs use; started only when the box has that many GPUs (tests/test_dp_gpu.py).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def shard_inputs(g, world):
    """Global minibatch = `world` copies of the fixture's batch with sample-dependent perturbations; noise for 2 steps."""
    import torch
    x, text, text_pad, patches, patch_pad = g.inputs()
    gen = torch.Generator().manual_seed(77)
    reps = [t.repeat((world,) + (1,) * (t.dim() - 1)) for t in (x, text, patches)]
    x, text, patches = (t + 0.05 * torch.randn(t.shape, generator=gen) for t in reps)
    text_pad, patch_pad = text_pad.repeat(world, 1), patch_pad.repeat(world, 1)
    n, B, L = g.dims["n_critic"], x.shape[0], g.dims["L"]
    z = torch.randn(2, n + 1, B, L, generator=gen)
    alpha = torch.rand(2, n, B, generator=gen)
    return (x, text, text_pad, patches, patch_pad), z, alpha


def build(g, precision="f32", device="cuda:0"):
    import gemm_gan_amd as gga
    d = g.dims
    w = gga.WGAN_GP(d["G"], d["L"], d["E"], [d["H"], d["H"], d["G"]], [d["H"], d["H"], 1], text_embedding_dims=d["Dt"],
                    patches_embedding_dims=d["Dp"], negative_slope=g.slope, n_critic=d["n_critic"], dropout=0.0,
                    device=device, precision=precision)
    w.build_WGAN_GP()
    w.init_train()
    w.gen.load_state_dict(g.state("init_gen"))
    w.disc.load_state_dict(g.state("init_disc"))
    return w


def run_steps(w, inputs, z, alpha, rows):
    x, text, text_pad, patches, patch_pad = (t[rows] for t in inputs)
    xs, ts, tps, ps, pps = w._prep(x, text, text_pad, patches, patch_pad)
    losses = []
    for step in range(2):
        nxt = (ps, pps, ts, tps) if step == 0 else None
        w.train_with_noise(xs, ts, tps, ps, pps, z[step][:, rows].to(w.device).contiguous(),
                           alpha[step][:, rows].to(w.device).contiguous(), next_cond=nxt)
        losses.append([float(v) for v in w.d_batch_loss] + [float(w.g_batch_loss[0])])
        w.launch_log = getattr(w, "launch_log", []) + [w.engine.launch_count()]
    return losses


def main():
    rank, world, port, fixture, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch
    import torch.distributed as dist
    from golden_util import Golden
    backend = sys.argv[6] if len(sys.argv) > 6 else "gloo"
    device = "cuda:0"
    if backend == "nccl":
        device = f"cuda:{rank}"
        torch.cuda.set_device(rank)
        import gemm_gan_amd as gga
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device),
                                pg_options=gga.rccl_process_group_options())        # as INTEGRATION.md section 3 prescribes
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)          # before anything touches the GPU
    try:
        g = Golden(fixture)
        inputs, z, alpha = shard_inputs(g, world)
        w = build(g, device=device)
        w.measure_comm = True
        B = g.dims["B"]
        losses = run_steps(w, inputs, z, alpha, slice(rank * B, (rank + 1) * B))
        torch.cuda.synchronize()
        sd = {"g." + k: v.detach().cpu().numpy() for k, v in w.gen.state_dict().items()}
        sd.update({"d." + k: v.detach().cpu().numpy() for k, v in w.disc.state_dict().items()})
        np.savez(out, losses=np.array(losses), comm_wait_ms=np.float64(w.comm_wait_ms()), backend=np.array(dist.get_backend()),
                 world=np.int64(dist.get_world_size()), launches=np.array(w.launch_log, dtype=np.int64), **sd)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
