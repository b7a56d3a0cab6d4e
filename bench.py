#!/usr/bin/env python
"""Headline benchmark: clips/sec of one full R(2+1)D-18 training step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts one child process per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is main.py:170-203 of the reference: zero_grad -> forward -> MSE -> backward ->
Adam, on 22 synthetic clips (3x16x112x112, fp32) per GPU with random-init weights
(BASELINE.json configs[1]; per-GPU batch fixed => weak scaling); with N > 1 the gradients are
averaged across ranks by the bucketed RCCL all-reduce of ``ddp.GradientSync``, overlapped
with backward.  Inputs are resident in HBM before the timed region (``--input u8`` instead
streams uint8 frames over PCIe every step and runs the clip transform on the device: the
PCIe-inclusive rate, never the headline).  Rank 0 prints ONE JSON line.  Extra objects on it:

* ``roofline``  : the dominant kernel (the fp32-MFMA convolution on the 64->144 1x3x3 shape,
  41 % of forward FLOPs; kw taps in Winograd F(4,3) form).  ``achieved`` / ``frac`` are the
  FLOPs the matrix pipe EXECUTES (half the direct-convolution count) / mean launch duration
  from HIP events recorded on the launch stream inside the timed steps, against the
  157.3 TFLOP/s fp32 matrix peak (MI355X_MICROARCH.md): a physical fraction <= 1.  The
  direct-convolution (algorithmic) rate SURVEY 8d counts is ``algorithmic_tflops``;
* ``cpu_baseline``: the CPU oracle (oracle/restatement.py, pinned to the reference) timed on
  this host's cores on a bounded sample (N = 2 clips), rank 0 at N = 1 only;
* ``host_enqueue_ms``: host time to queue one step's launches (no device sync inside);
* ``extra``: after everything else, N = 1 only: BASELINE.json configs[3] (C3D training step),
  configs[4]'s per-GPU work (32-frame bf16 ``evaluate()`` protocol) and the mixed-precision (bf16)
  training step of SURVEY row a12 (``extra.train_bf16``: never the headline).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import threading
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
DEFAULT_TIMEOUT_S = 600.0
# S1 of SURVEY section 2a: Conv3d(64, 144, (1,3,3), stride 1, pad (0,1,1)) on 16x56x56
S1_GEOMETRY = dict(Cin=64, Cout=144, kT=1, kH=3, kW=3, sT=1, sH=1, sW=1, Ti=16, Hi=56, Wi=56)
# the dominant forward kernel per network: geometry (for the HIP-event timer), label, launches per step;
# `mfma_saving`: direct-convolution MACs / MACs the Winograd F(4,3)-along-W form executes (12 -> 6 per 4 outputs)
DOMINANT = {
    "r2plus1d_18": dict(geometry=S1_GEOMETRY, launches=4, symbol="zsv::conv_wino4_kernel<3, 12, false, 1>", mfma_saving=2.0,
                        pmc_key="conv_wino4_kernel<3, 12, false, 1>", pmc_key_old="conv_wino4_kernel<3, 12>",
                        what="Conv3d(64,144,(1,3,3)) forward @16x56x56 (resnet.py:40-45, layer1 spatial half of Conv2Plus1D)"),
    # network.py:105 conv2 = Conv3d(64,128,3x3x3, pad 1) after pool1 (1,2,2): 22.2 GFLOP/clip, 29 % of C3D's forward FLOPs
    "c3d": dict(geometry=dict(Cin=64, Cout=128, kT=3, kH=3, kW=3, sT=1, sH=1, sW=1, Ti=16, Hi=56, Wi=56), launches=1,
                symbol="zsv::conv_wino4_kernel<4, 0, false, 4>", mfma_saving=2.0, pmc_key=None,
                what="Conv3d(64,128,3x3x3) forward @16x56x56 (network.py:105 conv2)"),
}
BASELINE_CONFIG = {"r2plus1d_18": "BASELINE.json configs[1]; configs[2] when n_gpus > 1", "c3d": "BASELINE.json configs[3]"}
# committed PMC passes of the dominant kernel, newest first (counters cannot be read inside this process)
PMC_BUSY_FILES = ("r04_s1_mfma_busy.json", "r03_s1_mfma_busy.json", "r02_s1_mfma_busy.json")
PMC_TRAFFIC_FILES = ("r04_s1_hbm_traffic.json", "r03_s1_hbm_traffic.json", "r02_s1_hbm_traffic.json")


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


def supervise(procs, timeout_s: float, poll_s: float = 0.2, grace_s: float = 5.0, label=lambda i: f"rank {i}") -> int:
    """Wait for the started ranks (``procs``: list of ``subprocess.Popen``).  First non-zero exit: the other ranks
    (left waiting in a collective) are terminated.  Past ``timeout_s`` the ranks still running are terminated, after
    ``grace_s`` killed, and the return code is 124 with the stuck ranks named on stderr.  Never blocks forever."""
    deadline = time.monotonic() + float(timeout_s)
    alive = dict(enumerate(procs))
    failed = 0
    timed_out = False
    kill_at = None
    while alive:
        for i, p in list(alive.items()):
            rc = p.poll()
            if rc is None:
                continue
            del alive[i]
            if rc != 0 and not failed and not timed_out:
                failed = rc
                print(f"bench.py: {label(i)} exited with code {rc}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in alive.values():
                    q.terminate()
                kill_at = time.monotonic() + grace_s
        now = time.monotonic()
        if alive and not timed_out and not failed and now > deadline:
            timed_out = True
            stuck = ", ".join(label(i) for i in sorted(alive))
            print(f"bench.py: no result after {timeout_s:.0f} s (--timeout): {stuck} still running "
                  "(hung in a collective or in a kernel?); terminating", file=sys.stderr, flush=True)
            for q in alive.values():
                q.terminate()
            kill_at = now + grace_s
        if alive and kill_at is not None and now > kill_at:
            for q in alive.values():
                q.kill()
            kill_at = now + grace_s
        if alive:
            time.sleep(poll_s)
    return 124 if timed_out else failed


def launch_children(n_gpus: int, argv, timeout_s: float = DEFAULT_TIMEOUT_S, script: str = os.path.abspath(__file__),
                    need_devices: bool = True) -> int:
    """``python bench.py --gpus N`` without a launcher: start one fresh process per GPU (the reference's
    ``opt.bs *= n_gpu`` + ``nn.DataParallel``, main.py:61-63,126, becomes one rank per device).  The parent
    never touches the GPU (``device_count()`` does not initialise it on this image) and never execs; rank
    0's stdout -- the one JSON line -- is this process's stdout.  Non-zero exit if any rank fails or the
    deadline passes (``supervise``)."""
    import subprocess
    if need_devices:
        have = torch.cuda.device_count()
        if os.environ.get("ZSV_BENCH_SAME_DEVICE"):          # rehearsal on a one-GPU box: every rank on cuda:0, gloo transport
            have = max(have, n_gpus)
        if have < n_gpus:
            print(f"bench.py: --gpus {n_gpus} but only {have} HIP device(s) are visible", file=sys.stderr)
            return 2
    procs = []
    for cmd, extra in child_commands(n_gpus, argv, free_port(), script):
        env = dict(os.environ)
        env.update(extra)
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if extra["RANK"] == "0" else subprocess.DEVNULL))
    return supervise(procs, timeout_s)


def rank_core_slice(cores, local_rank: int, local_world: int):
    """The cores rank ``local_rank`` of ``local_world`` keeps: a contiguous slice of the sorted allowed set (ranks
    do not fight for cores; neighbouring GPUs usually share a socket with neighbouring core numbers).  With fewer
    cores than ranks every rank keeps the whole set.  Pure function (tested on CPU)."""
    cores = sorted(cores)
    if local_world <= 1 or len(cores) < 2 * local_world:
        return cores
    per = len(cores) // local_world
    return cores[local_rank * per:(local_rank + 1) * per]


def pin_rank_to_cores(local_rank: int, local_world: int):
    if os.environ.get("ZSV_BENCH_NO_AFFINITY") or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        mine = rank_core_slice(os.sched_getaffinity(0), local_rank, local_world)
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), 8)))
        return mine
    except OSError:
        return None


def start_watchdog(timeout_s: float, rank: int):
    """Under ``torch.distributed.run`` nobody supervises the ranks: a rank that is still alive ``timeout_s`` after
    its start reports and exits with code 124 (``os._exit``: a hung collective cannot be unwound), which makes the
    launcher stop the others."""
    def fire():
        print(f"bench.py: rank {rank} still running after {timeout_s:.0f} s (--timeout): giving up "
              "(hung in a collective or in a kernel?)", file=sys.stderr, flush=True)
        os._exit(124)
    t = threading.Timer(float(timeout_s), fire)
    t.daemon = True
    t.start()
    return t


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--network", default="r2plus1d_18")
    ap.add_argument("--batch", type=int, default=CLIPS_PER_GPU, help="clips per GPU")
    ap.add_argument("--optimizer", choices=["fused", "torch"], default="torch",
                    help="fused: zeroshotvideoclassification_amd.optim.FusedAdam (one launch, same update rule as "
                         "torch.optim.Adam); torch: torch.optim.Adam")
    ap.add_argument("--input", choices=["resident", "u8"], default="resident",
                    help="resident: fp32 clips already in HBM (the headline); u8: pinned uint8 frames (N,T,H,W,3) -> async "
                         "H2D on a copy stream (double-buffered) -> zsv_clip_transform -> step, every step")
    ap.add_argument("--input-hw", default="128x171", help="frame size of the uint8 source clips for --input u8")
    ap.add_argument("--timeout", type=float, default=DEFAULT_TIMEOUT_S,
                    help="seconds after which ranks that are still running are stopped and the run fails")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the C3D / 32-frame bf16 eval legs after the timed region")
    ap.add_argument("--no-phases", action="store_true", help="skip the forward-only / forward+backward legs (counter passes over whole steps)")
    ap.add_argument("--cpu-steps", type=int, default=12)
    return ap.parse_args(argv)


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


def _profile_entry(files, keys):
    for name in files:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                kernels = json.load(f)["kernels"]
        except Exception:
            continue
        for key in keys:
            if key and key in kernels:
                return kernels[key], name
    return None, None


def pmc_traffic(dom, n_clips: int):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    separate rocprofv3 --pmc runs of tools/conv_bench.py at N = 22).  None when the batch differs from the
    profiled one or no pass is committed for this kernel."""
    if not dom.get("pmc_key") or n_clips != CLIPS_PER_GPU:
        return None
    k, _ = _profile_entry(PMC_TRAFFIC_FILES, (dom["pmc_key"], dom.get("pmc_key_old")))
    return round(k["hbm_bytes"]) if k and "hbm_bytes" in k else None


def pmc_busy(dom):
    """(matrix-pipe busy fraction, sustained clock in GHz, file) of the dominant kernel from the committed
    ``--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`` pass."""
    if not dom.get("pmc_key"):
        return None, None, None
    k, name = _profile_entry(PMC_BUSY_FILES, (dom["pmc_key"], dom.get("pmc_key_old")))
    if not k:
        return None, None, None
    return k.get("mfma_pipe_utilisation"), k.get("clock_GHz"), name


def roofline_entry(dom, n_clips: int, durations_ms):
    """The ``roofline`` object of the JSON line for the dominant kernel (pure function of the measured launch
    durations; tested on CPU).  ``frac`` = executed matrix-pipe FLOPs / peak: a physical fraction."""
    mean_ms = sum(durations_ms) / len(durations_ms)
    gm = dom["geometry"]
    taps = gm["kT"] * gm["kH"] * gm["kW"]
    voxels = gm["Ti"] * gm["Hi"] * gm["Wi"]                 # stride 1, "same" padding: output voxels = input voxels
    flops = 2.0 * n_clips * gm["Cout"] * gm["Cin"] * taps * voxels   # direct-convolution count (S1: 8.324 GFLOP/clip)
    alg_bytes = 4.0 * (n_clips * gm["Cin"] * voxels + n_clips * gm["Cout"] * voxels + gm["Cout"] * gm["Cin"] * taps)
    algorithmic = flops / (mean_ms * 1e-3) / 1e12
    executed = algorithmic / dom["mfma_saving"]
    busy, clock, src = pmc_busy(dom)
    return {"kernel": f"{dom['symbol']} = {dom['what']} (fp32 Winograd F(4,3) along W + its weight-transform launch), "
                      f"{dom['launches']} launch(es)/step",
            "bound": "mfma", "achieved": round(executed, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(executed / FP32_MFMA_PEAK_TFLOPS, 4),
            "achieved_is": "FLOPs the matrix pipe executes (direct-convolution FLOPs / speedup_vs_direct; counter-checked: "
                           "SQ_VALU_MFMA_BUSY_CYCLES is exactly 1/2 of the direct kernel's) / mean launch time, against the "
                           "2.4 GHz peak",
            "algorithmic_tflops": round(algorithmic, 2), "speedup_vs_direct": dom["mfma_saving"],
            "algorithmic_flops_per_launch": flops,
            "mfma_busy": busy, "sustained_clock_ghz": clock, "pmc_source": (f"profiles/{src}" if src else None),
            "launches_timed": len(durations_ms), "mean_launch_ms": round(mean_ms, 4),
            "algorithmic_gb_per_s": round(alg_bytes / (mean_ms * 1e-3) / 1e9, 1),
            "traffic": pmc_traffic(dom, n_clips)}


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


def allocator_snapshot(dev=None):
    """What torch's caching allocator has asked the driver for so far (hipMalloc calls, retries after a failed one, bytes
    reserved): a timed region that grows the pool pays for device allocations, not for kernels."""
    st = torch.cuda.memory_stats(dev)
    return {"device_allocs": int(st.get("num_device_alloc", 0)), "device_frees": int(st.get("num_device_free", 0)),
            "alloc_retries": int(st.get("num_alloc_retries", 0)), "reserved_mb": round(st.get("reserved_bytes.all.current", 0) / 2**20, 1)}


def allocator_delta(a, b):
    return {"device_allocs": b["device_allocs"] - a["device_allocs"], "device_frees": b["device_frees"] - a["device_frees"],
            "alloc_retries": b["alloc_retries"] - a["alloc_retries"], "reserved_mb_before": a["reserved_mb"],
            "reserved_mb_after": b["reserved_mb"]}


def run_queued(step_fn, steps):
    """Queue ``steps`` steps without waiting for any of them (how the timed region runs), then wait once.  Returns the wall
    time of the whole region, the host time at which step i was queued, the device time of each step (HIP events on the
    current stream at the step boundaries), how many steps the device still had to finish when the host was done
    queueing, and the last step's result."""
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host = []
    out = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        out = step_fn()
        marks[i + 1].record()
        host.append(time.perf_counter() - t0)
    behind = sum(0 if m.query() else 1 for m in marks[1:])
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    return wall, host, dev_ms, behind, out


def settle_allocator(step_fn, steps, dev, max_rounds=4, label=""):
    """Warm up IN THE MODE THAT IS TIMED: rounds of ``steps`` queued steps until a round asks the driver for no new memory.
    (A step queued while the previous one is still running cannot reuse the blocks the weight-gradient stream still holds
    -- `record_stream` -- so a host that runs n steps ahead needs n sets of activations; a warm-up that waits after every
    step never builds them, and the timed region then pays for the hipMalloc calls: DESIGN section 6.)"""
    rounds = []
    for _ in range(max_rounds):
        a = allocator_snapshot(dev)
        wall, _, _, _, _ = run_queued(step_fn, steps)
        b = allocator_snapshot(dev)
        rounds.append({"ms_per_step": round(1e3 * wall / steps, 3), "device_allocs": b["device_allocs"] - a["device_allocs"],
                       "reserved_mb": b["reserved_mb"]})
        log(f"{label}settle round {len(rounds)}: {rounds[-1]}")
        if rounds[-1]["device_allocs"] == 0:
            break
    return rounds


class U8Feeder:
    """``--input u8``: the input side of main.py:167 on the device.  Two pinned uint8 clip batches ``(N,T,H,W,3)``
    (what the reference's dataset workers produce before ``auxiliary/transforms.py:41-56``) alternate; batch i+1
    crosses PCIe on a copy stream while step i computes; ``preprocess.ClipTransform`` (one HIP kernel: normalise,
    THWC->CTHW, bilinear resize, crop, flip) writes the fp32 model input on the compute stream."""

    def __init__(self, n, frames, h, w, dev, rank=0):
        from zeroshotvideoclassification_amd import preprocess
        g = torch.Generator().manual_seed(4242 + rank)
        self.host = [torch.randint(0, 256, (n, frames, h, w, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(2)]
        self.dev = [torch.empty_like(hst, device=dev) for hst in self.host]
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.copied = [torch.cuda.Event(), torch.cuda.Event()]
        self.consumed = [None, None]
        self.transform = preprocess.get_transform(False, SIZE)
        self.bytes_per_step = self.host[0].numel()
        self.issued = 0
        self.taken = 0

    def prefetch(self):
        slot = self.issued % 2
        with torch.cuda.stream(self.copy_stream):
            if self.consumed[slot] is not None:
                self.copy_stream.wait_event(self.consumed[slot])          # the transform of two steps ago has read the slot
            self.dev[slot].copy_(self.host[slot], non_blocking=True)
            self.copied[slot].record(self.copy_stream)
        self.issued += 1

    def next(self):
        if self.issued == self.taken:
            self.prefetch()
        slot = self.taken % 2
        self.taken += 1
        self.prefetch()                                                  # the next batch travels while this one computes
        main = torch.cuda.current_stream()
        main.wait_event(self.copied[slot])
        x = self.transform(self.dev[slot])                               # (N, 3, T, 112, 112) fp32
        ev = torch.cuda.Event()
        ev.record(main)
        self.consumed[slot] = ev
        return x.unsqueeze(1)


def pacer_depth() -> int:
    """Steps the host may run ahead of the device in every timed loop of this file (``train.StepPacer``); 0 = unbounded (A/B)."""
    return int(os.environ.get("ZSV_BENCH_PACER_DEPTH", "2"))


def extra_c3d(dev, steps=10, warmup=3):
    """BASELINE.json configs[3] for the driver's record: C3D training step at 22 clips (network.py:95-180).  Timed exactly
    like the headline: ``warmup`` queued steps, then ``steps`` queued steps between two device syncs; the allocator's
    requests to the driver inside the timed region are on the record (the round-3 leg paid for them)."""
    from types import SimpleNamespace
    from zeroshotvideoclassification_amd import network, ops, synthetic, train
    model = network.get_network(SimpleNamespace(network="c3d", fixconvs=False, nopretrained=False))   # (network.py:128-131: True would read ./assets/c3d.pickle)
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0))
    model.to(dev).train()
    criterion = torch.nn.MSELoss().to(dev)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    x = synthetic.synthetic_clips(CLIPS_PER_GPU, FRAMES, SIZE).to(dev)
    _, z = synthetic.synthetic_targets(CLIPS_PER_GPU)
    z = z.to(dev)
    pacer = train.StepPacer(pacer_depth()) if pacer_depth() > 0 else None

    def step():
        return train.train_step(model, optimizer, criterion, x, z, pacer=pacer)

    before_warmup = allocator_snapshot(dev)
    step()
    torch.cuda.synchronize()                              # first-use work: weight panels, job tables
    run_queued(step, max(warmup - 1, 1))
    dom = DOMINANT["c3d"]
    timer = ops.KernelTimer("conv_fwd", dict(dom["geometry"], N=CLIPS_PER_GPU))
    ops.KERNEL_TIMER = timer
    a = allocator_snapshot(dev)
    wall, host, dev_ms, behind, (_, loss) = run_queued(step, steps)
    b = allocator_snapshot(dev)
    ops.KERNEL_TIMER = None
    out = {"workload": f"c3d training step (zero_grad+fwd+MSE+bwd+Adam), {CLIPS_PER_GPU} clips 3x{FRAMES}x{SIZE}x{SIZE}, fp32 "
                       "(BASELINE.json configs[3])",
           "value": round(CLIPS_PER_GPU * steps / wall, 2), "unit": "clips/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(1e3 * wall / steps, 3), "median_step_ms_on_device": round(statistics.median(dev_ms), 3),
           "step_ms_on_device": [round(v, 2) for v in dev_ms],
           "host_queued_all_after_ms": round(1e3 * host[-1], 2), "host_lead_steps": behind,
           "pacer_depth": pacer_depth(), "pacer_waits": pacer.waits if pacer else None,
           "allocator_in_timed_region": allocator_delta(a, b), "device_allocs_in_warmup": a["device_allocs"] - before_warmup["device_allocs"],
           "final_loss": float(loss.item())}
    if timer.pairs:
        r = roofline_entry(dom, CLIPS_PER_GPU, timer.durations_ms())
        out["dominant_kernel"] = {k: r[k] for k in ("kernel", "achieved", "frac", "algorithmic_tflops", "mean_launch_ms", "unit")}
    return out


BF16_MFMA_PEAK_TFLOPS = 2500.0       # dense bf16 matrix peak (MI355X_MICROARCH.md); the sparsity figure is never used


def extra_train_bf16(dev, steps=10, warmup=6):
    """SURVEY row a12, never the headline: the reference's mixed-precision step (main.py:172 `with autocast():` +
    main.py:137,195-203 GradScaler) on the bf16 training path (amp.py): R(2+1)D-18, 22 clips, bf16 activations and products,
    fp32 accumulation / statistics / parameters / loss, `optim.LossScaler` driving `optim.FusedAdam`."""
    from types import SimpleNamespace
    from zeroshotvideoclassification_amd import network, ops, optim, synthetic, train
    model = network.get_network(SimpleNamespace(network="r2plus1d_18", fixconvs=False, nopretrained=False))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0))
    model.to(dev).train()
    criterion = torch.nn.MSELoss().to(dev)
    optimizer = optim.FusedAdam(model.parameters(), lr=1e-3)
    scaler = optim.LossScaler(init_scale=2.0 ** 16)
    x = synthetic.synthetic_clips(CLIPS_PER_GPU, FRAMES, SIZE).to(dev)
    _, z = synthetic.synthetic_targets(CLIPS_PER_GPU)
    z = z.to(dev)
    pacer = train.StepPacer(pacer_depth()) if pacer_depth() > 0 else None

    def step():
        return train.train_step(model, optimizer, criterion, x, z, scaler=scaler, pacer=pacer, autocast=True)

    step()
    torch.cuda.synchronize()
    run_queued(step, max(warmup - 1, 1))
    settle = settle_allocator(step, steps, dev, max_rounds=3, label="extra.train_bf16 ")    # (many block sizes on two streams: the pool settles late)
    geometry = dict(S1_GEOMETRY, N=CLIPS_PER_GPU)
    timer = ops.KernelTimer("conv_bf16_fwd", geometry)
    ops.KERNEL_TIMER = timer
    a = allocator_snapshot(dev)
    wall, host, dev_ms, behind, (_, loss) = run_queued(step, steps)
    b = allocator_snapshot(dev)
    ops.KERNEL_TIMER = None
    value = CLIPS_PER_GPU * steps / wall
    out = {"workload": f"r2plus1d_18 MIXED-PRECISION training step (zero_grad + autocast(fwd + MSE) + scaled bwd + unscale / inf check + Adam), "
                       f"{CLIPS_PER_GPU} clips 3x{FRAMES}x{SIZE}x{SIZE}: bf16 activations / products, fp32 accumulation, statistics, parameters, "
                       "loss (main.py:172,137,195-203; SURVEY row a12; not a BASELINE.json config, never the headline)",
           "value": round(value, 2), "unit": "clips/s", "steps": steps, "warmup": warmup, "dtype": "bf16 (fp32 accumulate)",
           "ms_per_step": round(1e3 * wall / steps, 3), "median_step_ms_on_device": round(statistics.median(dev_ms), 3),
           "host_queued_all_after_ms": round(1e3 * host[-1], 2), "host_lead_steps": behind,
           "allocator_in_timed_region": allocator_delta(a, b), "allocator_settle_rounds": settle, "final_loss": float(loss.item()),
           "loss_scale": float(scaler.get_scale()) if hasattr(scaler, "get_scale") else None,
           "step_roofline": {"bf16_flop_frac_algorithmic": round(value * 242.5e9 / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4),
                             "note": "242.5 GFLOP/clip (SURVEY 8d) against the 2.5 PFLOP/s dense bf16 matrix peak; the step is bound by HBM "
                                     "(BatchNorm passes), L2->LDS operand traffic and launches, not by the matrix pipe"}}
    if timer.pairs:
        ms = timer.durations_ms()
        mean_ms = sum(ms) / len(ms)
        gm = S1_GEOMETRY
        flops = 2.0 * CLIPS_PER_GPU * gm["Cout"] * gm["Cin"] * 9 * gm["Ti"] * gm["Hi"] * gm["Wi"]
        tf = flops / (mean_ms * 1e-3) / 1e12
        out["dominant_kernel"] = {"kernel": "zsv::conv_bf16_same_kernel<9, 4, 1, 4> = Conv3d(64,144,(1,3,3)) forward @16x56x56 on channels-last bf16 "
                                            "(v_mfma_f32_16x16x32_bf16), 4 launches/step",
                                  "bound": "mfma", "achieved": round(tf, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(tf / BF16_MFMA_PEAK_TFLOPS, 4), "mean_launch_ms": round(mean_ms, 4),
                                  "launches_timed": len(ms)}
    return out


def extra_train_bf16_c3d(dev, steps=10, warmup=4):
    """The mixed-precision step for BASELINE configs[3]'s network: C3D, 22 clips, amp.Bf16TrainPathC3D (never the headline)."""
    from types import SimpleNamespace
    from zeroshotvideoclassification_amd import network, optim, synthetic, train
    model = network.get_network(SimpleNamespace(network="c3d", fixconvs=False, nopretrained=False))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0))
    model.to(dev).train()
    criterion = torch.nn.MSELoss().to(dev)
    optimizer = optim.FusedAdam(model.parameters(), lr=1e-3)
    scaler = optim.LossScaler(init_scale=2.0 ** 16)
    x = synthetic.synthetic_clips(CLIPS_PER_GPU, FRAMES, SIZE).to(dev)
    _, z = synthetic.synthetic_targets(CLIPS_PER_GPU)
    z = z.to(dev)
    pacer = train.StepPacer(pacer_depth()) if pacer_depth() > 0 else None

    def step():
        return train.train_step(model, optimizer, criterion, x, z, scaler=scaler, pacer=pacer, autocast=True)

    step()
    torch.cuda.synchronize()
    run_queued(step, max(warmup - 1, 1))
    settle = settle_allocator(step, steps, dev, max_rounds=3, label="extra.train_bf16_c3d ")
    a = allocator_snapshot(dev)
    wall, host, dev_ms, behind, (_, loss) = run_queued(step, steps)
    b = allocator_snapshot(dev)
    value = CLIPS_PER_GPU * steps / wall
    return {"workload": f"c3d MIXED-PRECISION training step (autocast(fwd + MSE) + scaled bwd + unscale / inf check + Adam), {CLIPS_PER_GPU} clips "
                        f"3x{FRAMES}x{SIZE}x{SIZE}: convolutions + max-pools forward and backward in bf16, fc6 / regressor / loss fp32 (never the headline)",
            "value": round(value, 2), "unit": "clips/s", "steps": steps, "warmup": warmup, "dtype": "bf16 (fp32 accumulate)",
            "ms_per_step": round(1e3 * wall / steps, 3), "median_step_ms_on_device": round(statistics.median(dev_ms), 3),
            "host_lead_steps": behind, "allocator_in_timed_region": allocator_delta(a, b), "allocator_settle_rounds": settle,
            "final_loss": float(loss.item()),
            "step_roofline": {"bf16_flop_frac_algorithmic": round(value * 3 * 77.06e9 / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4),
                              "note": "3 x 77.06 GFLOP/clip (forward + both gradients; SURVEY section 6) against the 2.5 PFLOP/s dense bf16 peak"}}


def extra_eval_t32_bf16(dev, batches=6, repeats=5):
    """BASELINE.json configs[4] per GPU: the reference's ``evaluate()`` protocol (main.py:224-313) on 32-frame clips
    with the bf16 engine: eval forward + cosine nearest class + the 10 half-class splits, three class tables."""
    from types import SimpleNamespace
    from zeroshotvideoclassification_amd import network, synthetic, train
    model = network.get_network(SimpleNamespace(network="r2plus1d_18", fixconvs=False, nopretrained=False))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0, bn_jitter=True))
    model.to(dev).eval()
    res = {}
    for name, ncls in (("ucf101", 101), ("hmdb51", 51), ("activitynet", 200)):
        table = synthetic.class_table(ncls, seed=1000 + ncls)
        data = []
        for i in range(batches):
            x = synthetic.synthetic_clips(CLIPS_PER_GPU, 32, SIZE, seed=7000 + i).to(dev)
            labels, z = synthetic.synthetic_targets(CLIPS_PER_GPU, ncls, seed=1000 + ncls, rank=i)
            data.append((x, labels, z))
        # warm-up: the whole protocol once (engine build, class table upload, and the allocator's block pool for six batches in flight:
        # with one warm-up batch the first table's timed pass still paid for fresh device allocations, 1.76 k vs 3 k clips/s)
        train.evaluate(model, data, table, device=dev, dtype=torch.bfloat16)
        torch.cuda.synchronize()
        # one pass of the protocol over 132 clips is 40-60 ms of wall time: a single sample of it swung between 2.1 k and 3.2 k clips/s
        # from run to run (round 3 / 4 records); the median of `repeats` passes is what is reported, the samples are on the record
        passes = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            r = train.evaluate(model, data, table, device=dev, dtype=torch.bfloat16)
            torch.cuda.synchronize()
            passes.append(time.perf_counter() - t0)
        res[name] = {"clips_per_s": round(r["n"] / statistics.median(passes), 1), "n": r["n"], "classes": ncls,
                     "passes_ms": [round(1e3 * p, 1) for p in passes]}
    rates = [v["clips_per_s"] for v in res.values()]
    out = {"workload": f"R(2+1)D-18 evaluate() protocol, {batches} batches x {CLIPS_PER_GPU} clips 3x32x{SIZE}x{SIZE}, bf16 engine "
                       "(BatchNorm folded), cosine nearest class + 10 half-class splits (BASELINE.json configs[4], one GPU's share)",
           "value": round(statistics.mean(rates), 1), "unit": "clips/s", "per_table": res}
    # The protocol is host-inclusive (its accuracy bookkeeping runs on the host between the device passes): on a box whose host cores are
    # shared with other jobs the passes of ONE table spread by 2-3x (round-4 record: 48 ... 163 ms where a quiet box reads 46.1 ... 46.8).
    # Say so on the line rather than leave a low median unexplained; the fastest pass is what the device path sustains.
    spread = max(max(v["passes_ms"]) / min(v["passes_ms"]) for v in res.values())
    out["pass_spread"] = round(spread, 2)
    if spread > 1.3:
        best = [v["n"] / (1e-3 * min(v["passes_ms"])) for v in res.values()]
        out["host_noise"] = True
        out["value_fastest_passes"] = round(statistics.mean(best), 1)
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(launch_children(args.gpus, sys.argv[1:], args.timeout))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    watchdog = start_watchdog(args.timeout, rank) if world > 1 else None
    cores = pin_rank_to_cores(local_rank, local_world) if world > 1 else None

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

    feeder = None
    if args.input == "u8":
        h, w = (int(v) for v in args.input_hw.lower().split("x"))
        feeder = U8Feeder(args.batch, FRAMES, h, w, dev, rank)
        x = None
    else:
        x = synthetic.synthetic_clips(args.batch, FRAMES, SIZE, rank=rank).to(dev)
    _, z = synthetic.synthetic_targets(args.batch, rank=rank)
    z = z.to(dev)

    pacer = train.StepPacer(pacer_depth()) if pacer_depth() > 0 else None

    def step():
        return train.train_step(model, optimizer, criterion, feeder.next() if feeder is not None else x, z, sync, pacer=pacer)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"model on {dev}, world {world}; warm-up {args.warmup} steps" + (f"; cores {cores[0]}..{cores[-1]}" if cores else ""))
    # warm-up: the first step is waited for (first-use work: weight panels, job tables, gradient-bucket discovery); the others are
    # queued the way the timed steps are, so the allocator's pool is the one the timed region needs
    for i in range(args.warmup):
        step()
        if i == 0 or os.environ.get("ZSV_BENCH_SYNCED_WARMUP"):
            torch.cuda.synchronize()
        log(f"warm-up step {i} " + ("done" if i == 0 else "queued"))
    dom = DOMINANT.get(args.network)
    timer = ops.KernelTimer("conv_fwd", dict(dom["geometry"], N=args.batch)) if dom else None
    ops.KERNEL_TIMER = timer
    barrier()
    alloc_before = allocator_snapshot(dev)
    step_marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_marks = []
    t0 = time.perf_counter()
    step_marks[0].record()
    loss = None
    for i in range(args.steps):
        _, loss = step()
        step_marks[i + 1].record()
        host_marks.append(time.perf_counter() - t0)
    t_queued = host_marks[-1] if host_marks else 0.0     # every launch of the K steps is queued; nothing was awaited (but the pacer)
    host_lead_steps = sum(0 if m.query() else 1 for m in step_marks[1:])   # steps the device still had to finish at that moment
    barrier()
    elapsed = time.perf_counter() - t0
    ops.KERNEL_TIMER = None
    alloc_timed = allocator_delta(alloc_before, allocator_snapshot(dev))
    step_dev_ms = [step_marks[i].elapsed_time(step_marks[i + 1]) for i in range(args.steps)]
    host_step_ms = [1e3 * (b - a) for a, b in zip([0.0] + host_marks[:-1], host_marks)]
    log(f"timed {args.steps} steps in {elapsed:.3f}s (host had queued them after {t_queued:.3f}s, {host_lead_steps} step(s) ahead of the device; "
        f"allocator in the timed region: {alloc_timed})")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # --input u8: the resident-input rate of the SAME process and device right after (devices of the pool differ by a few percent)
    resident_ref = None
    if feeder is not None:
        xr = synthetic.synthetic_clips(args.batch, FRAMES, SIZE, rank=rank).to(dev)
        for _ in range(2):
            train.train_step(model, optimizer, criterion, xr, z, sync, pacer=pacer)
        barrier()
        tr = time.perf_counter()
        for _ in range(args.steps):
            train.train_step(model, optimizer, criterion, xr, z, sync, pacer=pacer)
        barrier()
        resident_ref = 1e3 * (time.perf_counter() - tr) / args.steps
        del xr

    # host time to queue ONE step on an idle queue (no back-pressure from a full HIP queue): after the timed region
    enqueue = []
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        step()
        enqueue.append(time.perf_counter() - t)
    barrier()
    host_enqueue_ms = 1e3 * statistics.median(enqueue)

    # SURVEY 8d also asks for the forward-only and forward+backward times: measured AFTER the timed
    # region (N = 1 only), never part of `value`
    phases = None
    if world == 1 and feeder is None and not args.no_phases:
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
        ms_per_step = 1e3 * elapsed / args.steps
        label = {"r2plus1d_18": "R(2+1)D-18", "c3d": "C3D", "r3d_18": "R3D-18"}.get(args.network, args.network)
        which = BASELINE_CONFIG.get(args.network, "not a BASELINE.json config")
        if args.network == "r2plus1d_18":
            which = "BASELINE.json configs[1]" if world == 1 else f"BASELINE.json configs[2] at {world} GPUs"
        if args.batch != CLIPS_PER_GPU:
            which += f", but {args.batch} clips/GPU instead of {CLIPS_PER_GPU}"
        inputs = "synthetic" if feeder is None else \
            f"synthetic uint8 frames {args.input_hw} over PCIe every step + on-device clip transform (NOT the resident-input headline)"
        out = {
            "metric": f"clips/sec (fwd+bwd+step) {label} 16x112x112 bs={args.batch}/GPU",
            "value": round(value, 3), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": inputs,
            "config": {"workload": f"{args.network} training step (zero_grad+fwd+MSE+bwd+Adam), {args.batch} clips/GPU "
                                   f"3x{FRAMES}x{SIZE}x{SIZE}, random-init, fp32 ({which})",
                       "clips_per_gpu": args.batch, "global_batch": world * args.batch,
                       "optimizer": "Adam lr=1e-3 (" + ("one fused HIP launch over the flat gradient buckets" if args.optimizer == "fused" else "torch.optim.Adam") + ")",
                       "parallelism": f"dp{world}" + (" RCCL bucketed all-reduce overlapped with backward" if world > 1 else ""),
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "launcher": "self (one child process per GPU)" if os.environ.get("ZSV_BENCH_CHILD") else
                                   ("torch.distributed.run" if world > 1 else "single process"),
                       "input": args.input,
                       "final_loss": float(loss.item()) if loss is not None else None},
            # host side of a step: time until its last launch is queued.  `in_timed_region` includes any back-pressure of a
            # full HIP queue (the host runs ahead of the GPU); `idle_queue` is one step queued right after a device sync.
            "host_enqueue_ms": {"idle_queue": round(host_enqueue_ms, 3),
                                "in_timed_region": round(1e3 * t_queued / args.steps, 3),
                                "frac_of_step": round(host_enqueue_ms / ms_per_step, 3),
                                "cores_of_this_rank": len(cores) if cores else usable_cores(),
                                # the first steps of the timed region show the host's own cost (the queue is empty: nothing holds it
                                # back); once it is `pacer_depth` steps ahead it waits for the device at the head of every step
                                "per_step_first": [round(v, 2) for v in host_step_ms[:6]],
                                "per_step_median": round(statistics.median(host_step_ms), 3) if host_step_ms else None,
                                "host_lead_steps": host_lead_steps, "pacer_depth": pacer_depth(),
                                "pacer_waits": pacer.waits if pacer else None},
            "step_ms_on_device": {"median": round(statistics.median(step_dev_ms), 3), "min": round(min(step_dev_ms), 3),
                                  "max": round(max(step_dev_ms), 3)},
            "allocator_in_timed_region": alloc_timed,
            "degraded": False,                           # True: a secondary leg (extra.*) failed; its error is in the leg and on stderr
        }
        if feeder is not None:
            out["config"]["pcie_bytes_per_step"] = int(feeder.bytes_per_step)
            out["resident_input_same_process"] = {"ms_per_step": round(resident_ref, 3),
                                                  "value": round(world * args.batch / (resident_ref * 1e-3), 3), "unit": "clips/s",
                                                  "u8_over_resident": round(resident_ref / ms_per_step, 4)}
            out["config"]["input_pipeline"] = ("pinned uint8 (N,T,H,W,3) -> hipMemcpyAsync on a copy stream, two slots -> "
                                               "zsv_clip_transform (auxiliary/transforms.py:41-56) -> model")
        if sync is not None:
            out["config"]["allreduce_bytes_per_step"] = int(sync.bytes_reduced_last_step)
            out["config"]["allreduce_buckets"] = len(sync.bucket_sizes)
        # whole-step fractions of the two rooflines SURVEY section 8d defines
        per_gpu = value / world
        if args.network.startswith("r2plus1d"):
            out["step_roofline"] = {"fp32_flop_frac_algorithmic": round(per_gpu * 242.5e9 / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4),
                                    "hbm_frac_unfused_bytes": round(per_gpu * 5.03e9 / 8.0e12, 4)}
        if timer is not None and timer.pairs:
            out["roofline"] = roofline_entry(dom, args.batch, timer.durations_ms())
        if phases is not None:
            out["phases"] = phases
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.network, args.cpu_steps)
        if world == 1 and not args.no_extras and args.network == "r2plus1d_18" and feeder is None:
            del model, optimizer, x
            torch.cuda.empty_cache()
            extra = {}
            for name, fn in (("c3d", extra_c3d), ("eval_t32_bf16", extra_eval_t32_bf16), ("train_bf16", extra_train_bf16),
                             ("train_bf16_c3d", extra_train_bf16_c3d)):
                try:
                    log(f"extra.{name} ...")
                    extra[name] = fn(dev)
                except Exception as e:                      # the headline line must not be lost to a secondary leg -- but say so loudly
                    import traceback
                    traceback.print_exc()
                    print(f"bench.py: extra.{name} FAILED ({type(e).__name__}: {e}); the JSON line is marked degraded",
                          file=sys.stderr, flush=True)
                    extra[name] = {"error": f"{type(e).__name__}: {e}"}
                    out["degraded"] = True
                torch.cuda.empty_cache()
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()


if __name__ == "__main__":
    main()
