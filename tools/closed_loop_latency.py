#!/usr/bin/env python3
"""Wall-clock latency per simulation step of the closed loop at the reference settings (192 rows, 100 diffusion steps, K = 8,
maximize guidance on the last 10 steps, 5 candidates + RefineNet), HIP-graph replay against eager launches.
python tools/closed_loop_latency.py [--steps 40]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--steps", type=int, default=40)
    p.add_argument("--K", type=int, default=8)
    a = p.parse_args()
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    from pstl_diffusion_policy_amd.nusc_sim import closed_loop
    sd = init_state_dict(1007)
    out = {}
    for name, graph in (("graph", True), ("eager", False), ("graph_again", True)):
        recs = closed_loop(sd, n_sim_steps=a.steps, K=a.K, S=64, diffusion_steps=100, multi_cands=5, guidance=True,
                           guidance_before=10, guidance_lr=0.04, seed=1, verbose=False, graph=graph)
        lats = np.array(sorted(r["latency_s"] for r in recs[3:])) * 1e3
        out[name] = dict(median_ms=float(np.median(lats)), p10_ms=float(lats[len(lats) // 10]), worst_ms=float(lats[-1]),
                         last_score=recs[-1]["best_score"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
