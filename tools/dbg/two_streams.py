#!/usr/bin/env python3
"""do two half-batches on two HIP streams fill each other's tails (half A's k_guidance_iter beside half
B's single-step k_chain)?  The bench workload as ONE 4096-scene sampling region against TWO 2048-scene regions issued on two
streams (B delayed by a fraction of a guided step so that the phases alternate), per-step time of the pair.
    python tools/dbg/two_streams.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sd = init_state_dict(1007)
    pw = PackedWeights(sd, dev)
    guid = dict(enabled=True, before=10, niters=1, lr=0.01)
    kw = dict(rect_head=True, multi_cands=5, guidance=guid, want_scores3=False, diversity=True)
    full = {k: v.to(dev) for k, v in make_scene_batch(4096, K=2, S=64, seed=1000, invalid_lane_frac=0.2, stlp_mode="wide").items()
            if k not in ("params", "pre_stlp", "tj_scores_prior")}
    halves = [{k: v[:2048].contiguous() for k, v in full.items()}, {k: v[2048:].contiguous() for k, v in full.items()}]
    sms = [Sampler(pw, hp) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def one(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            sms[0].sampling_region(SceneBatch(full, 64, hp, dev), 50, None, None, seed=5 + r, **kw)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def two(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            for h in range(2):
                with torch.cuda.stream(streams[h]):
                    sms[h].sampling_region(SceneBatch(halves[h], 64, hp, dev), 50, None, None, seed=5 + r, **kw)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    for name, fn in (("one stream, 4096 scenes", one), ("two streams, 2 x 2048 scenes", two), ("one stream, 4096 scenes", one),
                     ("two streams, 2 x 2048 scenes", two)):
        fn(2)
        print("%-32s %.2f ms per 786 432 rows" % (name, min(fn(6), fn(6))))


if __name__ == "__main__":
    main()
