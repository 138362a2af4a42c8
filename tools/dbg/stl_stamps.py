#!/usr/bin/env python3
"""Where a wavefront of k_guidance_iter / k_stl_forward spends its cycles (a -DPSTL_STL_STAMP build: tools/dbg/stl_stamps.sh).
Runs the bench workload's guided rollout and its candidate scoring, prints cycles per wavefront and section."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd import ffi  # noqa: E402
from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

NAMES = ["prologue", "dynamics+sincos", "clearance", "lane ranking", "lane distance+heading", "log-sum-exp adds", "forward finish",
         "adjoint head", "adjoint re-derive block", "adjoint clearance", "adjoint lane", "adjoint weights", "adjoint costate",
         "emit (Adam, noise, store)", "zero emit", "tail", "select/gather"]


def stamps():
    L = ctypes.CDLL(ffi.LIB_PATH)
    buf = (ctypes.c_ulonglong * 32)()
    assert L.pstl_debug_stl_stamps(buf) == 0
    return list(buf)


def show(title, st, launches):
    waves = st[31]
    tot = sum(st[:31])
    print("%s: %d wavefronts in %d launch(es), %.0f cycles per wavefront" % (title, waves, launches, tot / max(waves, 1)))
    for i, n in enumerate(NAMES):
        if st[i]:
            print("   %-28s %9.0f cycles per wavefront  %5.1f %%" % (n, st[i] / waves, 100.0 * st[i] / tot))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=4096)
    ap.add_argument("--K", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sm = Sampler(PackedWeights(init_state_dict(1007), dev), hp)
    scene = {k: v.to(dev) for k, v in make_scene_batch(a.bs, K=a.K, S=64, seed=1000, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    sb = SceneBatch(scene, 64, hp, dev)
    _, base_p, _ = sm.encode(sb, need_rect=True)
    for rep in range(2):
        x = sm.fill_normal(sb, 50, 50, 5)
        stamps()
        sm.trace_stl = {}
        emit = sm.rollout(sb, base_p, x, None, 50, n_emit=5, clip=True, guidance=dict(enabled=True, before=10, niters=1, lr=0.01), seed=5)
        st = stamps()
        ts = [e0.elapsed_time(e1) for e0, e1, n in sm.trace_stl["guidance"]]
        sm.trace_stl = {}
        r = sm.score(sb, emit[-5:].contiguous(), select=True)
        st2 = stamps()
        t2 = [e0.elapsed_time(e1) for e0, e1, n in sm.trace_stl["score"]]
    show("k_guidance_iter (ten launches, %.3f ms each)" % (sum(ts) / len(ts)), st, 10)
    show("k_stl_forward, five candidates + select (%.3f ms)" % t2[0], st2, 1)


if __name__ == "__main__":
    main()
