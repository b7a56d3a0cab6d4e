"""Host logic of bench.py (no GPU): the rank supervisor's deadline, per-rank core slices, the physical roofline fraction."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _child(code: str):
    return subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def test_supervise_stops_hung_ranks_at_the_deadline(capfd):
    """A rank hung in a collective (here: a child that sleeps) must not hang the run: past --timeout the remaining ranks
    are terminated, the exit code is non-zero and the stuck ranks are named."""
    procs = [_child("print('done')"), _child("import time; time.sleep(600)"), _child("import time; time.sleep(600)")]
    t0 = time.monotonic()
    rc = bench.supervise(procs, timeout_s=2.0, poll_s=0.05, grace_s=2.0)
    assert rc == 124
    assert time.monotonic() - t0 < 15
    assert all(p.poll() is not None for p in procs)
    err = capfd.readouterr().err
    assert "rank 1" in err and "rank 2" in err and "rank 0" not in err.split("still running")[0].split(":")[-1]


def test_supervise_kills_ranks_that_ignore_terminate():
    code = "import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); time.sleep(600)"
    procs = [_child(code)]
    time.sleep(1.0)                      # let the child install its handler
    rc = bench.supervise(procs, timeout_s=0.5, poll_s=0.05, grace_s=0.5)
    assert rc == 124 and procs[0].poll() is not None


def test_supervise_propagates_a_failed_rank_and_stops_the_others(capfd):
    procs = [_child("import time; time.sleep(600)"), _child("import sys; sys.exit(7)")]
    rc = bench.supervise(procs, timeout_s=60.0, poll_s=0.05, grace_s=2.0)
    assert rc == 7 and all(p.poll() is not None for p in procs)
    assert "rank 1 exited with code 7" in capfd.readouterr().err


def test_supervise_all_ranks_fine():
    assert bench.supervise([_child("pass"), _child("pass")], timeout_s=60.0, poll_s=0.05) == 0


def test_launch_children_with_a_sleeping_script(tmp_path):
    """launch_children end to end (fresh child processes, never an exec) on a stand-in script: rank 1 hangs."""
    script = tmp_path / "fake_rank.py"
    script.write_text("import os, time\n"
                      "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "time.sleep(600 if os.environ['RANK'] == '1' else 0)\n")
    t0 = time.monotonic()
    rc = bench.launch_children(2, [], timeout_s=2.0, script=str(script), need_devices=False)
    assert rc == 124 and time.monotonic() - t0 < 20


def test_rank_core_slices_are_disjoint_and_cover():
    cores = list(range(3, 67))                                    # 64 allowed cores, not starting at 0
    slices = [bench.rank_core_slice(cores, r, 8) for r in range(8)]
    assert all(len(s) == 8 for s in slices)
    flat = [c for s in slices for c in s]
    assert sorted(flat) == cores and len(set(flat)) == 64
    assert bench.rank_core_slice(cores, 0, 1) == cores            # one rank keeps everything
    assert bench.rank_core_slice([0, 1, 2], 1, 8) == [0, 1, 2]    # fewer cores than 2 per rank: no pinning


def test_roofline_fraction_is_physical():
    """roofline.frac = EXECUTED matrix-pipe FLOPs / peak (<= 1 for any duration the kernel can reach); the
    direct-convolution rate is reported separately."""
    dom = bench.DOMINANT["r2plus1d_18"]
    r = bench.roofline_entry(dom, 22, [1.0073] * 4)               # BENCH_r02's mean launch time
    assert abs(r["algorithmic_tflops"] - 181.8) < 0.5             # 183.1 GFLOP / 1.0073 ms
    assert abs(r["achieved"] - r["algorithmic_tflops"] / 2) < 0.01 and r["speedup_vs_direct"] == 2.0
    assert abs(r["frac"] - 0.578) < 0.002 and r["frac"] <= 1.0
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    # the fastest the pipe could possibly run the executed FLOPs gives exactly 1
    t_min_ms = r["algorithmic_flops_per_launch"] / 2 / 157.3e12 * 1e3
    assert abs(bench.roofline_entry(dom, 22, [t_min_ms])["frac"] - 1.0) < 1e-3
    # counters from the committed PMC pass travel with the line
    assert r["pmc_source"] and r["pmc_source"].startswith("profiles/") and 0.3 < r["mfma_busy"] < 1.0
    assert 1.5 < r["sustained_clock_ghz"] < 2.5 and r["traffic"] > 9.0e8


def test_parse_defaults_finish_within_minutes():
    a = bench.parse([])
    assert (a.gpus, a.steps, a.warmup, a.input, a.network) == (1, 20, 5, "resident", "r2plus1d_18")
    assert 60 <= a.timeout <= 900
