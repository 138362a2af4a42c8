#!/usr/bin/env python3
"""k_guidance_iter per launch over batch sizes (both layouts: the latency layout runs up to 2 x 256 groups of 64 rows):
    python tools/dbg/guidance_by_size.py [--K 2]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sm = Sampler(PackedWeights(init_state_dict(1007), dev), hp)
    for bs in [int(v) for v in os.environ.get("SIZES", "1,16,43,64,85,86,128,170,171,256,512,1024").split(",")]:
        scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=a.K, S=64, seed=1000, invalid_lane_frac=0.2, stlp_mode="wide").items()
                 if k not in ("params", "pre_stlp", "tj_scores_prior")}
        sb = SceneBatch(scene, 64, hp, dev)
        _, base_p, _ = sm.encode(sb, need_rect=True)
        best = None
        for rep in range(4):
            x = sm.fill_normal(sb, 50, 50, 5)
            sm.trace_stl = {}
            sm.rollout(sb, base_p, x, None, 50, n_emit=5, clip=True, guidance=dict(enabled=True, before=10, niters=1, lr=0.01), seed=5)
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) for e0, e1, n in sm.trace_stl["guidance"])
            t = ts[len(ts) // 2]
            best = t if best is None else min(best, t)
        print("%5d scenes %7d rows %5d groups of 64 rows: %.1f us per launch (median of 10, best of 4 rollouts)" % (bs, sb.N, sb.N // 64, best * 1e3))


if __name__ == "__main__":
    main()
