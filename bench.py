#!/usr/bin/env python
"""Headline benchmark: clips/sec of one full R(2+1)D-18 training step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts one child process per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is main.py:170-203 of the reference: zero_grad -> forward -> MSE -> backward ->
Adam, on 22 synthetic clips (3x16x112x112, fp32) per GPU with random-init weights
(BASELINE.json configs[1]; per-GPU batch fixed => weak scaling); with N > 1 the gradients are
averaged across ranks by the bucketed RCCL all-reduce of ``ddp.GradientSync``, overlapped
with backward.  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON
line.  Extra objects on that line:

* ``roofline``  : the dominant kernel (the fp32-MFMA convolution on the 64->144 1x3x3 shape,
  41 % of forward FLOPs; kw taps in Winograd F(4,3) form): algorithmic (direct-convolution)
  FLOPs per launch / mean launch duration from HIP events recorded on the launch stream inside
  the timed steps, against the 157.3 TFLOP/s fp32 matrix peak (MI355X_MICROARCH.md); the FLOPs
  the matrix pipe really executes (1/2 of them) are reported next to it;
* ``cpu_baseline``: the CPU oracle (oracle/restatement.py, pinned to the reference) timed on
  this host's cores on a bounded sample (N = 2 clips), rank 0 at N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

T_START = time.perf_counter()
CLIPS_PER_GPU = 22
FRAMES, SIZE = 16, 112
FP32_MFMA_PEAK_TFLOPS = 157.3
# S1 of SURVEY section 2a: Conv3d(64, 144, (1,3,3), stride 1, pad (0,1,1)) on 16x56x56
S1_GEOMETRY = dict(Cin=64, Cout=144, kT=1, kH=3, kW=3, sT=1, sH=1, sW=1, Ti=16, Hi=56, Wi=56)
# the dominant forward kernel per network: geometry (for the HIP-event timer), label, launches per step
DOMINANT = {
    "r2plus1d_18": dict(geometry=S1_GEOMETRY, launches=4, symbol="zsv::conv_wino4_kernel<3, 12>",
                        what="Conv3d(64,144,(1,3,3)) forward @16x56x56 (resnet.py:40-45, layer1 spatial half of Conv2Plus1D)"),
    # network.py:105 conv2 = Conv3d(64,128,3x3x3, pad 1) after pool1 (1,2,2): 22.2 GFLOP/clip, 29 % of C3D's forward FLOPs
    "c3d": dict(geometry=dict(Cin=64, Cout=128, kT=3, kH=3, kW=3, sT=1, sH=1, sW=1, Ti=16, Hi=56, Wi=56), launches=1,
                symbol="zsv::conv_wino4_kernel<4, 0>",
                what="Conv3d(64,128,3x3x3) forward @16x56x56 (network.py:105 conv2)"),
}
BASELINE_CONFIG = {"r2plus1d_18": "BASELINE.json configs[1]; configs[2] when n_gpus > 1", "c3d": "BASELINE.json configs[3]"}


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_commands(n_gpus: int, argv, port: int, script: str = os.path.abspath(__file__)):
    """One ``(command, environment overrides)`` pair per rank: what ``torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1`` would hand each worker.  Pure function (tested on CPU)."""
    jobs = []
    for rank in range(n_gpus):
        env = {"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(n_gpus), "LOCAL_WORLD_SIZE": str(n_gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
               "ZSV_BENCH_CHILD": "1"}
        jobs.append(([sys.executable, script] + list(argv), env))
    return jobs


def launch_children(n_gpus: int, argv) -> int:
    """``python bench.py --gpus N`` without a launcher: start one fresh process per GPU (the reference's
    ``opt.bs *= n_gpu`` + ``nn.DataParallel``, main.py:61-63,126, becomes one rank per device).  The parent
    never touches the GPU (``device_count()`` does not initialise it on this image) and never execs; rank
    0's stdout -- the one JSON line -- is this process's stdout.  Non-zero exit if any rank fails."""
    import subprocess
    have = torch.cuda.device_count()
    if os.environ.get("ZSV_BENCH_SAME_DEVICE"):          # rehearsal on a one-GPU box: every rank on cuda:0, gloo transport
        have = max(have, n_gpus)
    if have < n_gpus:
        print(f"bench.py: --gpus {n_gpus} but only {have} HIP device(s) are visible", file=sys.stderr)
        return 2
    procs = []
    for cmd, extra in child_commands(n_gpus, argv, free_port()):
        env = dict(os.environ)
        env.update(extra)
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if extra["RANK"] == "0" else subprocess.DEVNULL))
    failed = 0
    while procs:
        for p in list(procs):
            rc = p.poll()
            if rc is None:
                continue
            procs.remove(p)
            if rc != 0 and not failed:
                failed = rc
                for q in procs:                       # a dead rank leaves the others waiting in a collective
                    q.terminate()
        time.sleep(0.2)
    return failed


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--network", default="r2plus1d_18")
    ap.add_argument("--batch", type=int, default=CLIPS_PER_GPU, help="clips per GPU")
    ap.add_argument("--optimizer", choices=["fused", "torch"], default="torch",
                    help="fused: zeroshotvideoclassification_amd.optim.FusedAdam (one launch, same update rule as "
                         "torch.optim.Adam); torch: torch.optim.Adam")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=12)
    return ap.parse_args()


def log(msg: str) -> None:
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def usable_cores() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (a GPU box
    hands a 1-GPU job a share of a much larger host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, quota // int(f.read())))
            break
        except Exception:
            continue
    return max(1, min(n, int(os.environ.get("ZSV_CPU_THREADS", "64"))))


def pmc_traffic(n_clips: int):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/r02_s1_hbm_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc runs of
    tools/conv_bench.py at N = 22).  Counters cannot be read inside this process; None when the
    batch differs from the profiled one."""
    path = os.path.join(ROOT, "profiles", "r02_s1_hbm_traffic.json")
    try:
        with open(path) as f:
            k = json.load(f)["kernels"]["conv_wino4_kernel<3, 12>"]
        return round(k["hbm_bytes"]) if n_clips == CLIPS_PER_GPU else None
    except Exception:
        return None


def cpu_baseline(network: str, steps: int):
    """Config A of BASELINE.md on the host cores: oracle train step, N = 2 clips."""
    from oracle import restatement as R
    from zeroshotvideoclassification_amd import synthetic
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads (os.cpu_count()={os.cpu_count()})")
    model = R.oracle_network(R.make_opt(network))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    n = 2
    x = synthetic.synthetic_clips(n, FRAMES, SIZE)
    _, z = synthetic.synthetic_targets(n)
    R.train_step(model, opt, x, z)                       # warm-up (oneDNN primitive creation)
    log("cpu_baseline: warm-up step done")
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        R.train_step(model, opt, x, z)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": n / med, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed train steps (median {med:.3f} s) of the CPU oracle on N=2 clips "
                      f"3x{FRAMES}x{SIZE}x{SIZE} fp32 + Adam after 1 warm-up; oracle is pinned to the reference "
                      "in the build container (tests/golden)"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(launch_children(args.gpus, sys.argv[1:]))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")

    from zeroshotvideoclassification_amd import _lib, ddp, network, ops, optim, synthetic, train
    from types import SimpleNamespace
    _lib.load()
    # (rehearsal knobs for a one-GPU box: ZSV_BENCH_SAME_DEVICE=1 puts every rank on cuda:0, ZSV_BENCH_BACKEND=gloo replaces RCCL,
    # which needs one device per rank; the numbers of such a run mean nothing, the code path is the point)
    if os.environ.get("ZSV_BENCH_SAME_DEVICE"):
        local_rank = 0
    backend = os.environ.get("ZSV_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    torch.manual_seed(0)
    model = network.get_network(SimpleNamespace(network=args.network, fixconvs=False, nopretrained=False))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0))
    model.to(dev).train()
    criterion = torch.nn.MSELoss().to(dev)
    # every rank builds the same name-keyed weights (seed 0): no initial broadcast needed
    sync = ddp.GradientSync(model, broadcast_initial_state=False) if world > 1 else None
    if args.optimizer == "fused":
        if sync is None:
            sync = ddp.GradientSync(model, local=True)        # flat gradient buckets without a collective
        optimizer = optim.FusedAdam(model.parameters(), lr=1e-3, grad_buckets=sync)
    else:
        optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)

    x = synthetic.synthetic_clips(args.batch, FRAMES, SIZE, rank=rank).to(dev)
    _, z = synthetic.synthetic_targets(args.batch, rank=rank)
    z = z.to(dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"model on {dev}, world {world}; warm-up {args.warmup} steps")
    for i in range(args.warmup):
        train.train_step(model, optimizer, criterion, x, z, sync)
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    dom = DOMINANT.get(args.network)
    timer = ops.KernelTimer("conv_fwd", dict(dom["geometry"], N=args.batch)) if dom else None
    ops.KERNEL_TIMER = timer
    barrier()
    t0 = time.perf_counter()
    loss = None
    for _ in range(args.steps):
        _, loss = train.train_step(model, optimizer, criterion, x, z, sync)
    barrier()
    elapsed = time.perf_counter() - t0
    ops.KERNEL_TIMER = None
    log(f"timed {args.steps} steps in {elapsed:.3f}s")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # SURVEY 8d also asks for the forward-only and forward+backward times: measured AFTER the timed
    # region (N = 1 only), never part of `value`
    phases = None
    if world == 1:
        def timed(fn, iters=5):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            return round(1e3 * (time.perf_counter() - t) / iters, 3)

        def fwd_only():
            with torch.no_grad():
                train.embed(model, x)

        def fwd_bwd():
            optimizer.zero_grad(set_to_none=True)
            criterion(train.embed(model, x), z).backward()

        phases = {"forward_ms": timed(fwd_only), "forward_backward_ms": timed(fwd_bwd)}
        optimizer.zero_grad(set_to_none=True)

    if rank == 0:
        total_clips = world * args.batch * args.steps
        value = total_clips / elapsed
        label = {"r2plus1d_18": "R(2+1)D-18", "c3d": "C3D", "r3d_18": "R3D-18"}.get(args.network, args.network)
        which = BASELINE_CONFIG.get(args.network, "not a BASELINE.json config")
        if args.network == "r2plus1d_18":
            which = "BASELINE.json configs[1]" if world == 1 else f"BASELINE.json configs[2] at {world} GPUs"
        if args.batch != CLIPS_PER_GPU:
            which += f", but {args.batch} clips/GPU instead of {CLIPS_PER_GPU}"
        out = {
            "metric": f"clips/sec (fwd+bwd+step) {label} 16x112x112 bs={args.batch}/GPU",
            "value": round(value, 3), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.network} training step (zero_grad+fwd+MSE+bwd+Adam), {args.batch} clips/GPU "
                                   f"3x{FRAMES}x{SIZE}x{SIZE}, random-init, fp32 ({which})",
                       "clips_per_gpu": args.batch, "global_batch": world * args.batch,
                       "optimizer": "Adam lr=1e-3 (" + ("one fused HIP launch over the flat gradient buckets" if args.optimizer == "fused" else "torch.optim.Adam") + ")",
                       "parallelism": f"dp{world}" + (" RCCL bucketed all-reduce overlapped with backward" if world > 1 else ""),
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "launcher": "self (one child process per GPU)" if os.environ.get("ZSV_BENCH_CHILD") else
                                   ("torch.distributed.run" if world > 1 else "single process"),
                       "final_loss": float(loss.item()) if loss is not None else None},
        }
        if sync is not None:
            out["config"]["allreduce_bytes_per_step"] = int(sync.bytes_reduced_last_step)
            out["config"]["allreduce_buckets"] = len(sync.bucket_sizes)
        # whole-step fractions of the two rooflines SURVEY section 8d defines
        per_gpu = value / world
        if args.network.startswith("r2plus1d"):
            out["step_roofline"] = {"fp32_flop_frac": round(per_gpu * 242.5e9 / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4),
                                    "hbm_frac_unfused_bytes": round(per_gpu * 5.03e9 / 8.0e12, 4)}
        if timer is not None and timer.pairs:
            ms = timer.durations_ms()
            mean_ms = sum(ms) / len(ms)
            n = args.batch
            gm = dom["geometry"]
            taps = gm["kT"] * gm["kH"] * gm["kW"]
            voxels = gm["Ti"] * gm["Hi"] * gm["Wi"]                 # stride 1, "same" padding: output voxels = input voxels
            flops = 2.0 * n * gm["Cout"] * gm["Cin"] * taps * voxels   # direct-convolution count (S1: 8.324 GFLOP/clip)
            alg_bytes = 4.0 * (n * gm["Cin"] * voxels + n * gm["Cout"] * voxels + gm["Cout"] * gm["Cin"] * taps)
            achieved = flops / (mean_ms * 1e-3) / 1e12
            # the kernel computes the kw taps in Winograd F(4,3) form (W % 4 == 0): 6 multiplies per 4 outputs instead of
            # 12, so the matrix pipe executes 1/2 of the algorithmic FLOPs (counter-checked: profiles/*_mfma_busy.json)
            executed = flops * 0.5 / (mean_ms * 1e-3) / 1e12
            out["roofline"] = {"kernel": f"{dom['symbol']} = {dom['what']} "
                                         f"(fp32 Winograd F(4,3) along W + its weight-transform launch), {dom['launches']} launch(es)/step",
                               "bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                               "achieved_is": "ALGORITHMIC (direct-convolution) FLOPs / time, as SURVEY 8d counts them; this is not "
                                              "pipe utilisation: the Winograd form executes 1/2 of them -> mfma_executed_frac",
                               "mfma_executed_tflops": round(executed, 2),
                               "mfma_executed_frac": round(executed / FP32_MFMA_PEAK_TFLOPS, 4),
                               "launches_timed": len(ms), "mean_launch_ms": round(mean_ms, 4),
                               "algorithmic_gb_per_s": round(alg_bytes / (mean_ms * 1e-3) / 1e9, 1),
                               "traffic": pmc_traffic(n) if args.network == "r2plus1d_18" else None}
        if phases is not None:
            out["phases"] = phases
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.network, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
