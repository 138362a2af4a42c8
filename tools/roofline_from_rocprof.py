#!/usr/bin/env python3
"""Recomputes bench.py's `roofline` object from the committed rocprofv3 kernel statistics alone:
    python tools/roofline_from_rocprof.py [profiles/r2]
achieved = rows x reverse steps in the launch x 172 032 FLOP (2 (40*256 + 256*256 + 256*40) per
row-evaluation) / the average duration of the multi-step k_chain dispatch in <dir>/bench_default_kernel_stats.csv;
peak = dense f16 MFMA peak / 3 products per fp32 product for the split forms, the fp32 MFMA peak for chain_waves = 8."""
import csv
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r2"
F_STEP = 2 * (40 * 256 + 256 * 256 + 256 * 40)
PEAK_F16, PEAK_F32 = 2516.8, 157.3
for stats, line, peak, what in (("bench_default_kernel_stats.csv", "bench_default_under_rocprof.json", PEAK_F16 / 3, "default (split-f16)"),
                                ("bench_fp32_mfma_kernel_stats.csv", "bench_fp32_under_rocprof.json", PEAK_F32, "fp32 MFMA (chain_waves 8)")):
    ps, pl = os.path.join(d, stats), os.path.join(d, line)
    if not (os.path.exists(ps) and os.path.exists(pl)):
        continue
    j = json.load(open(pl))
    rows = j["config"]["rows_per_gpu"]
    # the multi-step launch: the k_chain instantiation with the largest average duration
    ks = [r for r in csv.DictReader(open(ps)) if "k_chain" in r["Name"]]
    k = max(ks, key=lambda r: float(r["AverageNs"]))
    steps_in_launch = int(j["roofline"]["kernel"].split("(")[1].split()[3]) if "reverse steps" in j["roofline"]["kernel"] else 39
    ms, how = float(k["AverageNs"]) / 1e6, "average"
    if ms < 0.5 * float(k["MaxNs"]) / 1e6:   # the fp32 kernel has one instantiation for the multi- and the single-step
        ms, how = float(k["MaxNs"]) / 1e6, "LONGEST call (single-step launches share the row)"   # launches: no average
    ach = rows * steps_in_launch * F_STEP / (ms * 1e-3) / 1e12
    print(("%-28s %s calls, " + how + " %.3f ms  ->  %d rows x %d steps x %d FLOP = %.1f TFLOP/s, peak %.1f, frac %.3f   (bench "
           "line of the same run: %.3f ms, frac %.3f)") % (what, k["Calls"], ms, rows, steps_in_launch, F_STEP, ach, peak,
                                                           ach / peak, j["roofline"]["kernel_ms"], j["roofline"]["frac"]))
