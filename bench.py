#!/usr/bin/env python3
"""Headline benchmark: WGAN-GP samples/sec of the full train() step (n_critic=5, 5 000 genes,
256 patch tokens x 1024 + 1 text token x 512, B=256 per GPU) - BASELINE.json `metric`, config 3 /
config 4 (DP, weak scaling: fixed per-GPU batch).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the env), RCCL ("nccl" backend) all-reduce
of the flat gradient buffer per optimiser step.  Rank 0 prints ONE JSON line.

A "step" = one WGAN_GP.train() (R:463-477): 5 x {G fwd, D(fake), D(real), GP with double backward,
backward, clip 10, RMSprop step} + {G fwd, D fwd, backward, clip 2, RMSprop step}, dropout 0.1 in
the encoder layers exactly as the reference's train() mode, synthetic N(0,1) inputs resident in HBM.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters; dense, no sparsity)
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "fp8": 5000.0}      # fp8: the block-scaled e4m3 rate
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU minibatch (config 3: 256)")
    ap.add_argument("--genes", type=int, default=5000)
    ap.add_argument("--patches", type=int, default=256)
    ap.add_argument("--tokens", type=int, default=1)
    ap.add_argument("--text-dims", type=int, default=512)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--variant", choices=["xattn_film", "film", "img", "vanilla"], default="xattn_film",
                    help="xattn_film: the headline path (conditional_gan_cross_attention_with_film.py); film: the FiLM-only "
                         "sibling (conditional_gan_film.py; BASELINE configs[1] is --variant film --patches 1); img: "
                         "conditional_gan_img_transformer.py (configs[4] per rank: --variant img --batch 128 --genes 18000 --patches 1024); "
                         "vanilla: vanilla_gan_unconditional.py (configs[0]: --variant vanilla --batch 64 --genes 1000 --dropout 0)")
    ap.add_argument("--pad-frac", type=float, default=0.0,
                    help="fraction of samples whose last P/4 patch tokens are padded (SURVEY 8d masking run: 0.25)")
    ap.add_argument("--precision", choices=["f32", "bf16", "fp8", "bf16x3"], default="bf16",
                    help="GEMM arithmetic: bf16 = bf16 MFMA operands, fp32 accumulate (BASELINE north_star; headline), "
                         "f32 = exact fp32-input MFMA (the 1e-3 parity mode)")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the extra f32 parity-mode timing (N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=0,
                    help="minibatch of the CPU baseline (0 = the GPU run's minibatch when ONE timed step of it is estimated to fit in "
                         "--cpu-budget seconds - the estimate comes from a minibatch-8 probe step - else a minibatch-8 sample)")
    ap.add_argument("--cpu-steps", type=int, default=0, help="timed CPU steps (0: one at the full minibatch, or as many minibatch-8 steps as fit in ~20 s)")
    ap.add_argument("--cpu-budget", type=float, default=150.0, help="seconds the full-minibatch CPU step may take in the default run")
    ap.add_argument("--parity-steps", type=int, default=10, help="timed steps of the f32 reference-precision figure")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event per-kernel timing")
    ap.add_argument("--graph", action="store_true",
                    help="replay gg_train_step from its captured hipGraph (N=1) instead of enqueueing it kernel by kernel; "
                         "measured slower on ROCm 7.2, hence opt-in (DESIGN.md section 6)")
    return ap.parse_args()


def host_cores():
    """CPU share of this process: cgroup quota if one is set, else the affinity mask, capped at 16
    (one GPU's share of a box; oversubscribing torch threads beyond the quota stalls for minutes)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


PMC_SUMMARIES = ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")
SIDE_STREAM_FAMILIES = ("wgrad_kernel",)     # parameter-gradient kernels: launched on the engine's side stream, off the main chain


def family(name):
    """Kernel family of a profile class: the kernel's name without its template arguments."""
    return name.split("<")[0]


def fam_sum(rows, fam):
    """One row for a whole family: times, launches and algorithmic work of its instantiations added up."""
    rs = [r for r in rows if family(r["name"]) == fam and r["launches"] > 0]
    if not rs:
        return None
    return dict(name=fam, launches=sum(r["launches"] for r in rs), ms=sum(r["ms"] for r in rs), flops=sum(r["flops"] for r in rs),
                bytes=sum(r["bytes"] for r in rs), instantiations=len(rs))


def pmc_source():
    for f in PMC_SUMMARIES:
        if os.path.exists(os.path.join(ROOT, "profiles", f)):
            return f
    return None


def pmc_traffic(cls):
    """Launch-weighted mean HBM bytes per launch of the kernels in class `cls`, from the committed PMC summary
    (measured offline with rocprofv3 on this same command; bench.py cannot host the profiler itself)."""
    data = None
    for f in PMC_SUMMARIES:                                            # newest committed summary first
        try:
            data = json.load(open(os.path.join(ROOT, "profiles", f)))["kernels"]
            break
        except Exception:
            pass
    if data is None:
        return None
    # classes carry the kernels' own names: a full template name matches that instantiation, a bare name all of them
    want = cls.replace(" ", "")
    n = b = 0.0
    for k, v in data.items():
        kk = k.replace(" ", "").replace("gg::", "")
        # (an instantiation may carry trailing template arguments the class name omits, e.g. wgrad's loader flag)
        if kk == want or ("<" not in want and kk.split("<")[0] == want) or ("<" in want and kk.startswith(want[:-1] + ",")):
            n += v["launches"]
            b += v["launches"] * v["hbm_bytes_per_launch"]
    return round(b / n) if n else None


def algorithmic_step(variant, B, G, P, T, Dt, Dp=1024, E=256, H=256, Lz=256, F=512, n_critic=5):
    """algorithmic FLOPs and HBM bytes of ONE train() step (exact reference semantics, the reference's
    zero-valued backward skipped, backward = 2 x forward); fp32 element sizes for the bytes, every raw tensor once per pass."""
    S = P + 1
    film = Dt * 2 * Dp if variant in ("xattn_film", "film") else 0
    xattn = variant == "xattn_film"
    enc = 2 * (S * E * 3 * E + 2 * S * S * E + S * E * E + 2 * S * E * F)
    cond = film + (T * Dt * E if xattn else 0) + P * Dp * E + enc
    if xattn:
        cond += (2 * E * E + 2 * S * E * E + 2 * S * E) + (2 * E * E + 2 * T * E * E + 2 * T * E)
    if variant == "vanilla":
        cond = 0
    mlp_d = (G + E) * H + H * H + H
    mlp_g = (Lz + E) * H + H * H + H * G
    mac = n_critic * ((cond + mlp_g) + 3 * (cond + mlp_d) + 4 * (cond + mlp_d) + 3 * mlp_d) + \
        ((cond + mlp_g) + (cond + mlp_d) + mlp_d + 2 * (cond + mlp_g))
    flops = 2.0 * mac * B
    gp = n_critic * (5.0 * G * 4 * B + 3.0 * H * G * 4)                       # GP chain: x, x~, x^, grad (w + r) + W1x per GEMM
    passes = 0 if variant == "vanilla" else 6 * n_critic + 3                  # conditioning passes that touch the raw patches
    patch = passes * P * Dp * 4.0 * B
    return flops, gp + patch, {"gp_chain_bytes": gp, "patch_stream_bytes": patch}


def pmc_step_bytes(args):
    """HBM bytes of one step through the PMC counters, from the committed summary of this round (tools/pmc_traffic.sh on this
    same command); None when the summary is of another workload."""
    for f in PMC_SUMMARIES[:2]:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f)))
        except Exception:
            continue
        if not (args.variant == "xattn_film" and args.batch == 256 and args.genes == 5000 and args.patches == 256 and
                args.tokens == 1 and args.precision == "bf16" and args.pad_frac == 0):
            return None, None
        steps = d.get("steps_profiled", 2)                                    # bench.py --steps 1 --warmup 1
        tot = sum(v["launches"] * v["hbm_bytes_per_launch"] for v in d["kernels"].values())
        return tot / steps, f
    return None, None


def rccl_options():
    """gemm_gan_amd.rccl_process_group_options(): RCCL's internal stream from torch's high-priority pool (GG_RCCL_DEFAULT_PRIO=1: the default, A/B)."""
    from gemm_gan_amd.model import rccl_process_group_options
    return None if os.environ.get("GG_RCCL_DEFAULT_PRIO") == "1" else rccl_process_group_options()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(args):
    """Oracle #1 (stock torch modules on the host CPU, the reference's own arithmetic) on this box's host cores.  Default: ONE timed
    train() step at the GPU run's minibatch (BASELINE.md section 2 quotes the CPU path there: per-sample cost is NOT flat in the
    minibatch - 5.0 samples/s at minibatch 8 against 3.2 at 256 on 16 cores), preceded by a minibatch-8 probe that also bounds the
    run: when the full step is estimated beyond --cpu-budget the bounded minibatch-8 sample is reported instead and says so."""
    from oracle.torch_oracle import PathConfig, Trainer, film_config, img_config, synthetic_batch, vanilla_config
    cores = host_cores()
    torch.set_num_threads(cores)
    make = {"film": film_config, "img": img_config, "vanilla": vanilla_config}.get(args.variant, PathConfig)
    cfg = make(n_genes=args.genes, text_dims=args.text_dims, dropout=0.0 if args.variant == "vanilla" else args.dropout)

    def timed(Bc, n_steps, warm_up):
        torch.manual_seed(42)
        tr = Trainer(cfg)
        x, text, text_pad, patches, patch_pad = synthetic_batch(cfg, Bc, args.patches, args.tokens, seed=42)

        def step():
            zs = [torch.randn(Bc, cfg.latent_dims) for _ in range(cfg.n_critic + 1)]
            al = [torch.rand(Bc, 1) for _ in range(cfg.n_critic)]
            tr.train_step(x, text, text_pad, patches, patch_pad, zs, al)
        warm = None
        if warm_up:
            t0 = time.perf_counter()
            step()
            warm = time.perf_counter() - t0
        n = n_steps if n_steps > 0 else max(1, min(12, int(20.0 / max(warm or 1.0, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        return n, time.perf_counter() - t0, warm

    src = f"oracle/torch_oracle.py (stock torch CPU modules, fp32, dropout {args.dropout})"
    small = None
    Bc = args.cpu_batch
    if Bc == 0:         # probe at minibatch 8, then the full minibatch if one step of it fits the budget
        n, dt, warm = timed(min(8, args.batch), 2, True)
        small = dict(value=round(min(8, args.batch) * n / dt, 3), unit="samples/s", minibatch=min(8, args.batch),
                     sample=f"1 warm-up + {n} timed train() steps, {dt:.1f} s")
        est = 1.6 * dt / n * args.batch / min(8, args.batch)          # measured: per-sample cost grows ~1.6x from minibatch 8 to 256
        log(f"cpu baseline: minibatch-{min(8, args.batch)} probe {dt / n:.2f} s per step on {cores} threads; one step at {args.batch} estimated {est:.0f} s")
        Bc = args.batch if est <= args.cpu_budget else min(8, args.batch)
    if Bc == args.batch and args.batch > 8:
        n, dt, _ = timed(Bc, args.cpu_steps if args.cpu_steps > 0 else 1, args.cpu_steps > 1)
        sample = f"{src}, the SAME workload and minibatch as the GPU run ({Bc}): {n} timed train() step(s), {dt:.1f} s" + \
                 ("" if args.cpu_steps > 1 else " (no warm-up step: one step is 10^2 s of dense CPU work, start-up is noise)")
    else:
        n, dt, _ = timed(Bc, args.cpu_steps, True)
        sample = f"{src}, same workload at minibatch {Bc}" + ("" if Bc == args.batch else
                 f" (NOT the GPU run's {args.batch}: the full step was estimated beyond --cpu-budget {args.cpu_budget:.0f} s; "
                 "per-sample CPU cost GROWS with the minibatch, so this figure flatters the CPU)") + \
                 f": 1 warm-up + {n} timed train() steps, {dt:.1f} s"
    out = dict(value=round(Bc * n / dt, 3), unit="samples/s", cores=cores, kind="port", minibatch=Bc, sample=sample)
    if small is not None and Bc != small["minibatch"]:
        out["small_batch_probe"] = small
    try:        # the BASELINE.md section 2 protocol (1 warm-up + 2 timed steps at the full minibatch), measured once per round on a pool box
        full = json.load(open(os.path.join(ROOT, "profiles", "r04_cpu_baseline_B256.json")))
        if args.variant == "xattn_film" and args.batch == 256 and args.genes == 5000 and args.patches == 256:
            out["protocol_1_plus_2_steps"] = full
    except Exception:
        pass
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        try:                                  # host-only, before the GPU is touched
            cpu = cpu_baseline(args)
        except Exception as ex:  # pragma: no cover
            cpu = {"error": repr(ex)}
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP engine has no CPU fallback)")
    ndev = torch.cuda.device_count()
    if os.environ.get("GG_BENCH_BACKEND", "nccl") != "nccl" and ndev > 0:
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  GG_BENCH_BACKEND=gloo lets several ranks share ONE GPU to rehearse the
        # data-parallel path on a single-GPU box (never used for reported numbers).
        backend = os.environ.get("GG_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, pg_options=rccl_options())
        else:
            dist.init_process_group(backend)
    elif os.environ.get("GG_FORCE_DP_COLLECTIVES") == "1":
        # ONE rank, real RCCL: the data-parallel host loop with every all-reduce it issues on N ranks (5 per optimiser step, from the side
        # stream, under the backward stages) - what the loop and RCCL's launch path cost a step on a one-GPU box; not a scaling number
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        os.environ["GG_FORCE_DP_LOOP"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=rccl_options())

    import gemm_gan_amd as gga
    G, B, P, T = args.genes, args.batch, args.patches, args.tokens
    H = E = Lz = 256
    torch.manual_seed(42)                       # identical initial weights on every rank
    film = args.variant in ("film", "img")          # the 4-argument reference files
    if film and T != 1:
        raise SystemExit("--variant film / img take one text vector per sample (--tokens 1)")
    cls = {"film": gga.film.WGAN_GP, "img": gga.img_transformer.WGAN_GP}.get(args.variant, gga.WGAN_GP)
    vanilla = args.variant == "vanilla"
    if vanilla:
        w = gga.vanilla.WGAN_GP_nocond(G, Lz, [], [H, H, G], [H, H, 1], optimizer="rms_prop", n_critic=5, seed=1234 + rank,
                                       device=dev, results_dire="", precision=args.precision)
    else:
        w = cls(G, Lz, E, [H, H, G], [H, H, 1], text_embedding_dims=args.text_dims, patches_embedding_dims=1024,
                optimizer="rms_prop", n_critic=5, dropout=args.dropout, seed=1234 + rank, device=dev, results_dire="",
                precision=args.precision)
    w.build_WGAN_GP()
    w.init_train()
    if args.graph:
        w.use_graph = True
        w.engine.set_graph(True)
    w.reserve(B, P, T)
    g = torch.Generator(device=dev).manual_seed(42 + rank)   # per-rank shard of the global minibatch
    x = torch.randn(B, G, device=dev, generator=g)
    patches = torch.randn(B, P, 1024, device=dev, generator=g)
    text = torch.randn(B, T, args.text_dims, device=dev, generator=g)
    patch_pad = torch.zeros(B, P, dtype=torch.bool, device=dev)
    if args.pad_frac > 0:
        patch_pad[: int(round(B * args.pad_frac)), P - P // 4:] = True
    text_pad = torch.zeros(B, T, dtype=torch.bool, device=dev)

    forced_coll = world == 1 and dist.is_initialized()       # GG_FORCE_DP_COLLECTIVES: the N-rank host loop with its collectives on one rank
    w.measure_comm = world > 1 or forced_coll

    def train_once():
        if vanilla:
            w.train(x)
        elif film:
            w.train(x, text[:, 0, :], patches, patch_pad)
        else:
            # data parallel: the next step's conditioning inputs are known (here: the same synthetic shard), so the critic's
            # first conditioning forward of the next step runs under this step's generator all-reduce
            w.train(x, text, text_pad, patches, patch_pad,
                    next_batch=(x, text, text_pad, patches, patch_pad) if (world > 1 or forced_coll) else None)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    log(f"engine ready: workspace {w.engine.workspace_bytes / 2**30:.1f} GiB; warm-up {args.warmup} step(s)")
    prof = (not args.no_profile) and rank == 0
    rows_all = []
    for i in range(args.warmup):
        last = i == args.warmup - 1
        if prof and last:
            # every GEMM class, untimed, with the engine's side streams serialised: ISOLATED kernel durations (the per-class
            # table, the dominant class and its stand-alone roofline fraction)
            w.engine.set_side_streams(False)
            w.engine.profile(True)
        t1 = time.perf_counter()
        train_once()
        torch.cuda.synchronize(dev)
        log(f"warm-up step {i}: {(time.perf_counter() - t1) * 1e3:.1f} ms")
        if prof and last:
            rows_all = [r for r in w.engine.profile_collect() if r["launches"] > 0]
            w.engine.profile(False)
            w.engine.set_side_streams(True)
    sync()
    if world > 1 or forced_coll:
        w.comm_wait_ms()                          # drop the warm-up's event pairs
    if prof:
        # The timed region carries event pairs for the DOMINANT family and the side-stream family only (live roofline over the
        # timed steps): an event pair around each of the ~600 GEMM-class launches of a step costs 3 ms per step on the host-fed stream.
        # Nominated by FAMILY (all instantiations of one kernel template added up) over the kernels of the main chain; the
        # parameter-gradient family runs on the side stream beside that chain and is reported as `side_stream`.
        fams = {}
        for r in rows_all:
            fams[family(r["name"])] = fams.get(family(r["name"]), 0.0) + r["ms"]
        main_fams = {k: v for k, v in fams.items() if k not in SIDE_STREAM_FAMILIES}
        dom_fam = max(main_fams, key=main_fams.get) if main_fams else (max(fams, key=fams.get) if fams else None)
        live_cls = [r["name"] for r in rows_all if family(r["name"]) == dom_fam or family(r["name"]) in SIDE_STREAM_FAMILIES]
        w.engine.profile(True, live_cls if live_cls else None)
    prof_steps = min(2, args.steps)               # timed steps that carry the event pairs (a host-side flag, no sync)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if prof and i == prof_steps:
            w.engine.profile_pause()
        train_once()
    sync()
    dt = time.perf_counter() - t0
    rows = []
    if prof:
        rows = w.engine.profile_collect()
        w.engine.profile(False)
    launches_per_step = w.engine.launch_count()        # of the last timed step (the parity-mode leg below has its own count)
    phases = None
    if not args.no_profile:
        # one more, untimed step with the engine's phase marks on (an event per phase boundary of train(), ~35 per step, no
        # serialisation): the un-profiled timeline of the step, side streams and all.  Every rank runs it (it holds collectives).
        w.engine.phase_enable(True)
        train_once()
        torch.cuda.synchronize(dev)
        marks = w.engine.phase_times()
        w.engine.phase_enable(False)
        kinds = {}
        for name, pms in marks:
            kinds[name] = kinds.get(name, 0.0) + pms
        phases = {"marks": len(marks), "ms_total": round(sum(kinds.values()), 3),
                  "ms_by_phase": {k: round(v, 3) for k, v in kinds.items()},
                  "note": "milliseconds between the engine's phase marks (include/gemmgan.h gg_phase_*) on the main stream over one "
                          "untimed step after the timed region, summed per phase kind over the 5 critic iterations + the generator "
                          "iteration; a phase's time includes waiting for side-stream work it joins"}
    graph_stats = w.engine.graph_stats() if w.engine.graph else None
    comm_ms = w.comm_wait_ms() / args.steps if (world > 1 or forced_coll) else 0.0
    t = torch.tensor([dt, comm_ms], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, comm_ms = float(t[0].item()), float(t[1].item())
    losses_head = (float(w.d_batch_loss[0]), float(w.g_batch_loss[0]))
    gp_rows = None
    if rank == 0 and prof:
        # north_star: "achieved HBM GB/s on the GP kernel" - the four gradient-penalty kernels of csrc/gpchain.hip on the
        # buffers of the last critic iteration, each timed by its own dispatch timestamps (20 launches)
        gp_rows = [dict(kernel=r["kernel"], avg_launch_us=round(r["us"], 2), algorithmic_bytes=int(r["bytes"]),
                        **{"GB/s": round(r["bytes"] / (r["us"] * 1e-6) / 1e9, 1),
                           "frac_of_hbm_peak": round(r["bytes"] / (r["us"] * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)})
                   for r in w.engine.gp_profile(B)]
    parity = None
    if world == 1 and args.precision == "bf16" and not args.no_parity_mode:
        # The reference-precision figure: the SAME workload in the mode the 1e-3 parity gates run in.  bf16x3 = every fp32 operand
        # as bf16 parts (three in forward products: six MFMAs per tile, fp32-grade; two in backward products: three MFMAs), fp32
        # accumulate and fp32 storage, through the token-on-lane Linear, fused attention and weight-gradient kernels; f32 = the exact
        # fp32-input MFMA on the generic tile GEMMs with unfused attention (the round-1 / round-2 parity mode), timed beside it.
        def timed(mode, steps):
            w.engine.set_precision(mode)
            train_once()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(steps):
                train_once()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t1) / steps
        pdt = timed("bf16x3", args.parity_steps)
        fdt = timed("f32", max(2, args.parity_steps // 2))
        parity = {"dtype": "bf16x3", "ms_per_step": round(pdt * 1e3, 3), "value": round(B / pdt, 2), "unit": "samples/s",
                  "steps": args.parity_steps,
                  "f32_mode": {"dtype": "f32", "ms_per_step": round(fdt * 1e3, 3), "value": round(B / fdt, 2), "steps": max(2, args.parity_steps // 2)},
                  "note": "THE REFERENCE-PRECISION FIGURE: split-operand bf16x3 mode (fp32 operands as bf16 parts, fp32 accumulate, fp32 "
                          "storage) through the fused kernels - the mode tests/test_engine_golden_gpu.py and tests/test_engine_oracle_gpu.py "
                          "hold to <= 1e-3 elementwise against the reference (every test also runs in f32_mode, the exact fp32-input MFMA "
                          "path on the generic kernels); the headline `value` is the bf16-operand mode north_star prescribes, whose kernels "
                          "are held to float64 results on rounded operands in tests/test_kernels_gpu.py and whose gradient error is shown to "
                          "be ReLU gate flips in tests/test_gate_flips_gpu.py"}
        w.engine.set_precision("bf16")
    finite = bool(torch.isfinite(w.engine.flat[0]["w"]).all() and torch.isfinite(w.engine.flat[1]["w"]).all())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        out = {"metric": "WGAN-GP samples/sec (n_critic=5, 5k-gene)", "value": round(value, 2), "unit": "samples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
               "config": {"workload": ("configs[0] family: vanilla_gan_unconditional.py train() (no conditioning inputs), " if vanilla else
                                       "configs[4] family: conditional_gan_img_transformer.py train(), " if args.variant == "img" else
                                       "configs[1] family: conditional_gan_film.py train(), " if film else
                                       "configs[2]/[3]: conditional_gan_cross_attention_with_film.py train(), ")
                                      + f"per-GPU batch {B}, {G} genes, {P} patch tokens x1024, {T} text token x{args.text_dims}, "
                                      f"n_critic=5, rms_prop, dropout {args.dropout}"
                                      + (f", {args.pad_frac:.0%} of samples with the last {P // 4} patches padded" if args.pad_frac > 0 else ""),
                          "global_batch": world * B, "parallelism": f"dp{world}", "kernel_launches_per_step": launches_per_step,
                          "hip_graph": graph_stats,
                          "allreduce_wait_ms_per_step": round(comm_ms, 3) if (world > 1 or forced_coll) else None},
               "finite": finite,
               "losses": {"d": losses_head[0], "g": losses_head[1]}}
        if parity is not None:
            out["parity_mode"] = parity
        if rows:
            rows = [r for r in rows if r["launches"] > 0]
            dom = fam_sum(rows, dom_fam) or max(rows, key=lambda r: r["ms"])
            members = [r for r in rows if family(r["name"]) == dom["name"]] or [dom]
            tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
            # fp8 mode: only the e4m3 Linear instantiations (last template argument `true`) are priced against the fp8 peak;
            # attention, weight gradients and LayerNorm run their MFMAs in bf16 in that mode too
            fp8_kernel = args.precision == "fp8" and dom["name"].startswith(("tlin_", "wst_")) and \
                all(r["name"].rstrip(">").endswith("true") for r in members)
            peak_tf = PEAK_TFLOPS["fp8"] if fp8_kernel else PEAK_TFLOPS["f32" if args.precision == "f32" else "bf16"]
            frac_m, frac_h = tf / peak_tf, gbs / PEAK_HBM_GBS
            # which roof: the family's arithmetic intensity against the machine's ridge point (FLOP per HBM byte)
            intensity = dom["flops"] / max(dom["bytes"], 1.0)
            ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
            bound = "mfma" if intensity >= ridge else "hbm"
            traffic = pmc_traffic(dom["name"])

            def priced(r, b):
                ig = r["bytes"] / (r["ms"] * 1e-3) / 1e9
                it = r["flops"] / (r["ms"] * 1e-3) / 1e12
                return {"achieved": round(it if b == "mfma" else ig, 2),
                        "frac": round(it / peak_tf if b == "mfma" else ig / PEAK_HBM_GBS, 4),
                        "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2), "launches": r["launches"]}
            iso_row = fam_sum(rows_all, dom["name"])
            iso = priced(iso_row, bound) if iso_row else None
            side = []
            for f in SIDE_STREAM_FAMILIES:
                live, ser = fam_sum(rows, f), fam_sum(rows_all, f)
                if live is None:
                    continue
                sb = "mfma" if live["flops"] / max(live["bytes"], 1.0) >= ridge else "hbm"
                side.append({"kernel": f, "instantiations": live["instantiations"], "bound": sb,
                             "unit": "TFLOP/s" if sb == "mfma" else "GB/s", **priced(live, sb),
                             "algorithmic_bytes_per_launch": round(live["bytes"] / live["launches"]), "traffic": pmc_traffic(f),
                             "isolated": priced(ser, sb) if ser else None,
                             "ms_per_step_live": round(live["ms"] / prof_steps, 3),
                             "note": "launched on the engine's side stream beside the main chain: the live duration of a launch includes "
                                     "waiting for the compute units and HBM bandwidth the main chain holds, so its live sum is not time "
                                     "on the step's critical path; `isolated` is the serialised warm-up step"})
            out["roofline"] = {"kernel": dom["name"], "bound": bound,
                               "achieved": round(tf if bound == "mfma" else gbs, 2),
                               "peak": peak_tf if bound == "mfma" else PEAK_HBM_GBS,
                               "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                               "frac": round(frac_m if bound == "mfma" else frac_h, 4), "traffic": traffic,
                               "intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                               "traffic_note": "HBM bytes per launch (launch-weighted mean over the family) from rocprofv3 PMC "
                                               "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate passes, "
                                               f"profiles/{pmc_source()}; algorithmic bytes per launch = "
                                               + str(round(dom["bytes"] / dom["launches"] / 1e6, 1)) + " MB",
                               "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2), "launches": dom["launches"],
                               "share_of_step": round(dom["ms"] / (dt * 1e3 * prof_steps / args.steps), 3),
                               "timed_steps_with_events": prof_steps,
                               "instantiations": [{"name": r["name"], "launches": r["launches"],
                                                   "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2),
                                                   "TFLOP/s": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1),
                                                   "GB/s": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1),
                                                   "traffic": pmc_traffic(r["name"])} for r in members],
                               "side_stream": side,
                               "note": "nominated by kernel FAMILY: the kernel template whose instantiations add up to the largest time on "
                                       "the step's main chain in a step with the engine's streams serialised (classes carry the kernels' "
                                       "own names, as rocprofv3 prints them; `instantiations` lists them, `achieved` = the family's "
                                       "algorithmic bytes or FLOPs / the sum of its launch durations, by HIP events on the launching "
                                       "stream); the parameter-gradient kernels run on the side stream and are `side_stream`, not "
                                       "candidates. `bound` is the side of the ridge the family's algorithmic FLOP/byte falls on - the "
                                       "attention kernels are in fact VALU-bound (exp2, dropout hash, conversions), see "
                                       "profiles/r04_pmc_mfma.json; achieved / frac are live over the timed region, where the family "
                                       "shares the chip with the side streams' kernels; `isolated` is the same family in the serialised "
                                       "warm-up step",
                               "isolated": iso,
                               "gp_chain": gp_rows,
                               "gp_chain_note": "gradient-penalty kernels (R:351-374 closed form + double backward, 6 launches per critic "
                                                "iteration incl. the dW1x weight-gradient and the split-K grad*W1x^T GEMM, which appear "
                                                "under their own classes below); every tensor of the chain is <= B*G*4 = "
                                                f"{B * G * 4 / 1e6:.1f} MB ({B * G * 4 / 8e6:.2f} us at the HBM peak); gp_grad_k = g1 W1x with the "
                                                "row norms in its epilogue (split-operand strip kernel, fp32-grade products; 41 us in "
                                                "round 2, 12 us now)",
                               "time_weighted": (lambda rs: {
                                   "note": "all classes of the table, each priced against the roof its algorithmic intensity falls under "
                                           "(HBM below the ridge, MFMA above), weighted by their time in the serialised step",
                                   "frac": round(sum(r["ms"] * (r["bytes"] / (r["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS
                                                                if r["flops"] / max(r["bytes"], 1.0) < ridge else
                                                                r["flops"] / (r["ms"] * 1e-3) / 1e12 / peak_tf) for r in rs) /
                                                 max(sum(r["ms"] for r in rs), 1e-9), 4),
                                   "ms": round(sum(r["ms"] for r in rs), 2)})([r for r in (rows_all or rows) if r["ms"] > 0]),
                               "kernel_classes_note": ("one untimed warm-up step with event pairs on every class, streams serialised; "
                                                       "algorithmic FLOPs and bytes per class (tensors once at their stored "
                                                       "element sizes)") if rows_all else "timed region",
                               "kernel_classes": [{"name": r["name"], "launches": r["launches"], "ms": round(r["ms"], 2),
                                                     "TFLOP/s": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2),
                                                     "GB/s": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)} for r in (rows_all or rows)]}
        # whole-step roofline: algorithmic work against the step's wall time, and the HBM bytes the step really moves
        fl, by, parts = algorithmic_step(args.variant, B, G, P, T, args.text_dims)
        peak_tf_step = PEAK_TFLOPS["f32" if args.precision == "f32" else "bf16"]    # fp8 mode: most of the step's MFMAs are bf16
        pmc_b, pmc_f = pmc_step_bytes(args)
        step = {"algorithmic_flops": fl, "algorithmic_hbm_bytes": by, **parts,
                "achieved_TFLOP/s": round(fl / (ms * 1e-3) / 1e12, 1), "peak_TFLOP/s": peak_tf_step,
                "frac_mfma": round(fl / (ms * 1e-3) / 1e12 / peak_tf_step, 4),
                "ideal_ms_mfma": round(fl / (peak_tf_step * 1e12) * 1e3, 3), "ideal_ms_hbm": round(by / (PEAK_HBM_GBS * 1e9) * 1e3, 3),
                "pmc_hbm_bytes": pmc_b, "pmc_source": pmc_f,
                "pmc_over_algorithmic": round(pmc_b / by, 2) if pmc_b else None,
                "phases": phases,
                "frac_hbm_on_pmc_bytes": round(pmc_b / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if pmc_b else None,
                "note": "algorithmic = SURVEY.md 8(d) formulas (FLOPs = 2 x MACs of the reference's necessary work; bytes = gradient-"
                        "penalty chain + raw patch stream in fp32); pmc = sum over the heavy kernels of launches x (2*FETCH_SIZE + "
                        "WRITE_SIZE)*1024 from the committed rocprofv3 passes of this command, per step"}
        if "roofline" in out:
            out["roofline"]["step"] = step
        else:
            out["roofline_step"] = step
        if world > 1 or forced_coll:
            out["config"]["collective"] = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                                           "buckets_per_optimizer_step": 1 if vanilla else 1 + getattr(w.engine, "cond_stages", 0),
                                           "optimizer_steps_per_train": 6,
                                           "note": "one all-reduce per backward stage, issued from the side stream the moment that "
                                                   "stage's gradients are final (MLP; then cross-attention, encoder layers last to "
                                                   "first, patch/text front), reduced under the stages still running"}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1 or dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
