#!/usr/bin/env python
"""Assemble the committed PMC summaries bench.py reads (profiles/rNN_s1_mfma_busy.json, rNN_s1_hbm_traffic.json) from the per-set
outputs of tools/r03_pmc_s1.sh (tools/pmc_summary.py JSON per `--pmc` pass):
    python tools/pmc_assemble.py gpurun_out/r03_pmc_s1_set 03 [calibration json]"""
import json
import sys

prefix, rnd = sys.argv[1], sys.argv[2]
sets = [json.load(open(f"{prefix}{i}.json")) for i in (1, 2, 3, 4)]
cal = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
WHAT = {
    "conv_wino4_kernel<3, 12, false, 1>": "S1 forward 64->144 1x3x3 with the BatchNorm statistics epilogue, Winograd F(4,3) along W (3 row tiles of 48 channels): the kernel bench.py times",
    "conv_wino4_kernel<4, 0, false, 0>": "S1 dgrad (F(4,3), one 64-row tile, single-buffered U panel, plain epilogue)",
    "conv_wgrad_wino_kernel<9, 1, false>": "S1 wgrad (F(2,3) transpose)",
    "conv_winot4_kernel<4, true, 1>": "T1 forward 144->64 3x1x1, F(4,3) along T, BatchNorm + ReLU folded into the operand read, statistics epilogue",
    "conv_winot4_kernel<3, false, 0>": "T1 dgrad 64->144 (F(4,3) along T, three 48-row tiles)",
    "conv_wgrad_twino_kernel<9, true>": "T1 wgrad (F(2,3) transpose along T, BatchNorm-folded operand)",
}
busy, traffic = {}, {}
for k, what in WHAT.items():
    if k not in sets[0]:
        continue
    a, f, w, l = sets[0][k], sets[1].get(k, {}), sets[2].get(k, {}), sets[3].get(k, {})
    busy[k] = {"what": what, "duration_us": a["duration_us"], "GRBM_GUI_ACTIVE": a["GRBM_GUI_ACTIVE"], "clock_GHz": a["clock_GHz"],
               "SQ_VALU_MFMA_BUSY_CYCLES": a["SQ_VALU_MFMA_BUSY_CYCLES"], "mfma_pipe_utilisation": a["mfma_pipe_utilisation"],
               "SQ_WAVES": a["SQ_WAVES"], "SQ_INSTS_VALU": l.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": l.get("SQ_INSTS_SALU"),
               "SQ_LDS_BANK_CONFLICT": l.get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": l.get("SQ_LDS_IDX_ACTIVE")}
    if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
        rd, wr = f["FETCH_SIZE"] * 2 * 1024, w["WRITE_SIZE"] * 1024
        traffic[k] = {"what": what, "FETCH_SIZE_KiB": f["FETCH_SIZE"], "WRITE_SIZE_KiB": w["WRITE_SIZE"], "read_bytes": rd, "write_bytes": wr,
                      "hbm_bytes": rd + wr, "duration_us_fetch_pass": f["duration_us"], "duration_us_write_pass": w["duration_us"]}
src = ("rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 tools/conv_bench.py --shapes S1,T1 --kinds fwd,dgrad,wgrad --iters 3 --pre --stats "
       f"(round {int(rnd)}, tools/rNN_pmc*.sh: one own pass per counter set, no other trace domains; mean of the last 3 dispatches; tools/pmc_summary.py, tools/pmc_assemble.py)")
json.dump({"source": src,
           "how_to_read": "SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs = 32 cycles per v_mfma_f32_16x16x4_f32; the direct kernel would execute 2,861,236,224 on S1 (89.4 M MFMAs = 183.1 GFLOP), "
                          "F(2,3) 2/3 of that, F(4,3) exactly 1/2; GRBM_GUI_ACTIVE is summed over the 8 XCDs; matrix-pipe utilisation = MFMA_BUSY / (1024 * GUI_ACTIVE / 8); clock = GUI_ACTIVE / 8 / duration",
           "kernels": busy}, open(f"profiles/r{rnd}_s1_mfma_busy.json", "w"), indent=1)
out = {"source": src, "how_to_read": "HBM bytes per launch = FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 bytes: calibration below) + WRITE_SIZE, KiB -> bytes; "
                                     "compulsory: S1 forward 918.7 MB, S1 dgrad 918.7 MB, T1 forward / dgrad 918.7 MB, wgrads 918.4 MB + slabs",
       "kernels": traffic}
if cal:
    c = cal.get("bn_stats_kernel", {})
    out["calibration"] = {"kernel": "bn_stats_kernel reads the 635,830,272-byte layer1 mid tensor once", "FETCH_SIZE_KiB": c.get("FETCH_SIZE"),
                          "ratio_counted_to_read": round(c.get("FETCH_SIZE", 0) * 1024 / 635830272.0, 4)}
json.dump(out, open(f"profiles/r{rnd}_s1_hbm_traffic.json", "w"), indent=1)
print("wrote", len(busy), "kernels")
