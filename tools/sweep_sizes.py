#!/usr/bin/env python3
"""Batch-size sweep of the sampling region (engine level, in-kernel noise): ms per batch and trajectories/s at
192 ... 786 432 rows, per workload; with --detail, HIP-event times of the denoiser launches and the STL launches.
python tools/sweep_sizes.py [--workload e7_guid|e7|e5] [--steps 50] [--K 2] [--detail]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--workload", default="e7_guid")
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--K", type=int, default=2)
    p.add_argument("--scenes", default="1,16,128,512,4096")
    p.add_argument("--multi_cands", type=int, default=5)
    p.add_argument("--n_rolls", type=int, default=0)
    p.add_argument("--reps", type=int, default=5)
    p.add_argument("--chain_waves", type=int, default=0)
    p.add_argument("--detail", action="store_true")
    p.add_argument("--diversity", action="store_true")
    p.add_argument("--rollout_only", action="store_true", help="time only the multi-step denoiser launch (HIP events)")
    a = p.parse_args()
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sm = Sampler(PackedWeights(init_state_dict(1007), dev), hp, chain_waves=a.chain_waves)
    rect = a.workload != "e5"
    guid = dict(enabled=True, before=10, niters=1, lr=0.01) if a.workload == "e7_guid" else None
    for bs in [int(v) for v in a.scenes.split(",")]:
        scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=a.K, S=64, seed=1000, invalid_lane_frac=0.2, stlp_mode="wide").items()
                 if k not in ("pre_stlp", "tj_scores_prior")}
        call = [0]

        def step():
            call[0] += 1
            sb = SceneBatch(scene, 64, hp, dev)
            return sm.sampling_region(sb, a.steps, None, None, rect_head=rect, multi_cands=a.multi_cands if rect else None,
                                      guidance=guid, want_scores3=False, seed=1234 + call[0], n_rolls=a.n_rolls,
                                      diversity=a.diversity)

        if a.rollout_only:
            sb = SceneBatch(scene, 64, hp, dev)
            _, base_p, _ = sm.encode(sb, need_rect=False)
            x = torch.randn(sb.N, 40, device=dev)
            sm.trace = []
            for _ in range(a.reps + 1):
                sm.rollout(sb, base_p, x, None, a.steps, n_emit=5, clip=True, seed=7)
            torch.cuda.synchronize()
            ms = [e0.elapsed_time(e1) for (e0, e1, n, _) in sm.trace][1:]
            from pstl_diffusion_policy_amd import ffi      # the library says which layout it picked (pstl_rollout_layout)
            kern, g, rounds = ffi.rollout_layout(sb.cfg(a.steps, ffi.PSTL_FLAG_RNG, sm.chain_waves))
            print(json.dumps(dict(rows=sb.N, kernel=["k_chain latency", "k_chain", "k_chain2", "k_chain exact"][kern],
                                  tiles_per_group=g, rounds=rounds, chain_ms=sum(ms) / len(ms),
                                  us_per_tile_step=1e3 * sum(ms) / len(ms) / ((a.steps - 1) * g * rounds))), flush=True)
            sm.trace = None
            continue
        for _ in range(2):
            step()
        sm.trace, sm.trace_stl = [], {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        N = bs * 192
        rec = dict(workload=a.workload, rows=N, steps=a.steps, K=a.K, ms_per_step=dt * 1e3, value=N / dt)
        if a.detail:
            multi = [e0.elapsed_time(e1) for (e0, e1, n, _) in sm.trace if n > 1]
            rec["chain_multi_ms"] = sum(multi) / a.reps
            rec["chain_multi_steps"] = sm.trace[0][2] if sm.trace else 0
            for kind, evs in sm.trace_stl.items():
                rec[kind + "_ms"] = sum(e0.elapsed_time(e1) for (e0, e1, _) in evs) / a.reps
                rec[kind + "_launches"] = len(evs) // a.reps
        sm.trace, sm.trace_stl = None, None
        assert not sm.check_chain_domain(fallback=False)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
