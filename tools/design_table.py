#!/usr/bin/env python3
"""Prints DESIGN.md section 6's table from the bench lines of a profiles/rN directory:  python tools/design_table.py profiles/r2"""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r2"
rows = [("bench_default", "**e7 + guidance** (50 steps, K=2, last 10 steps × 1 Adam iter, 5 candidates, RefineNet, diversity metrics) — default (split-f16 chains)"),
        ("bench_fp32_mfma", "the same with the MLP chains on fp32 MFMA (`--chain_waves 8`)"),
        ("bench_split_bf16", "the same with bfloat16 pieces (`--chain_waves 32`, the round-1 default)"),
        ("bench_e7", "e7 (50 steps, K=2, 5 candidates + RefineNet)"),
        ("bench_e5", "e5 (50 steps, K=2, DDPM only + final STL)"),
        ("bench_big_shard", "the default on config 4's whole batch on ONE GPU (32 768 scenes = 6 291 456 rows, `--scenes 32768`)"),
        ("bench_k8_s100", "e7 + guidance at the reference defaults (100 steps, K=8)"),
        ("bench_e8_train", "e8 training step (config 5, N1: sampling + RefineNet forward/backward + Adam)"),
        ("bench_e7_train", "e7 training step (N1: `--diverse_loss`, DPP diversity objective, merge_net architecture)"),
        ("bench_e8_train_joint", "e8 training step with `--joint` (Adam over the whole net: + the three scene encoders' backward)"),
        ("bench_e7_train_joint", "e7 training step with `--joint` (+ encoders and merge_net backward)"),
        ("bench_trajopt", "traj-opt loop (N4): 50 Adam iterations per batch in one launch")]
print("| workload (786 432 rows = 4096 scenes × 64 × 3, S=64) | ms / batch | trajectories/s | STL-sat rate | chain launch: ms, TFLOP/s, frac |")
print("|---|---|---|---|---|")
for f, label in rows:
    p = os.path.join(d, f + ".json")
    if not os.path.exists(p):
        continue
    j = json.load(open(p))
    r = j.get("roofline", {})
    unit = "%.2f M" % (j["value"] / 1e6) if j["unit"].startswith("traj") else "%.2e row-iterations/s" % j["value"]
    chain = "%.2f, %.0f, %.3f" % (r["kernel_ms"], r["achieved"], r["frac"]) if r else "—"
    sat = "%.3f" % j["stl_sat_rate"] if j.get("stl_sat_rate") is not None and "train" not in f else "—"
    print("| %s | %.1f | %s | %s | %s |" % (label, j["ms_per_step"], unit, sat, chain))
p = os.path.join(d, "bench_default.json")
if os.path.exists(p):
    j = json.load(open(p))
    c = j.get("cpu_baseline")
    if c:
        print("| CPU oracle, %s (`cpu_baseline`, kind \"%s\", %d host threads) | — | %.0f | %.3f | — |" % (
            c["sample"].split(",")[1].strip() + "," + c["sample"].split(",")[2], c["kind"], c["cores"], c["value"], c["stl_sat_rate"]))
        print("\ngpu_same_inputs:", json.dumps(c.get("gpu_same_inputs")))
    print("stl_kernels:", json.dumps(j["roofline"]["stl_kernels"]))
