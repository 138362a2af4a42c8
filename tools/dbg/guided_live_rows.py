#!/usr/bin/env python3
"""Who needs the adjoint in the guided phase?  For the bench workload (or --bs scenes of it): at every guided reverse step,
the share of rows whose hinge loss relu(thres - score) is active (non-zero gradient), and the share of 64-row wavefronts
(one (scene, mode) each in the by-mode row mapping) that carry at least one such row -- the adjoint in k_guidance_iter is
skipped per wavefront, so the second number is what the kernel pays for.
    python tools/dbg/guided_live_rows.py [--bs 4096] [--steps 50] [--before 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd import ffi  # noqa: E402
from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--before", type=int, default=10)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--K", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sm = Sampler(PackedWeights(init_state_dict(1007), dev), hp)
    S = 64
    scene = {k: v.to(dev) for k, v in make_scene_batch(a.bs, K=a.K, S=S, seed=1000, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    sb = SceneBatch(scene, S, hp, dev)
    feature, base_p, base_r = sm.encode(sb, need_rect=True)
    x = sm.fill_normal(sb, a.steps, a.steps, 5)
    scale = torch.tensor([hp["mul_w_max"], hp["mul_a_max"]], device=dev).repeat(ffi.T)
    rows = []
    orig = sm._stl_event

    def hook(kind, n):
        if kind == "guidance":
            sc = sm.score(sb, (x * scale).reshape(1, sb.N, ffi.CTRL))["scores"][0]
            act = (sc < hp["stl_nn_thres"]) & (sb.valid.reshape(-1) != 0)
            # rows of a wavefront: sample-major inside (scene, mode)?  by_mode mapping: row = scene*192 + sample*3 + mode
            am = act.reshape(a.bs, S, 3)
            per_wave = am.any(dim=1)                      # (scene, mode)
            rows.append((float(act.float().mean()), float(per_wave.float().mean()),
                         [float(am[:, :, m].float().mean()) for m in range(3)],
                         float(am.float().sum(dim=1)[per_wave].mean()) if per_wave.any() else 0.0))
        return orig(kind, n)

    sm._stl_event = hook
    sm.rollout(sb, base_p, x, None, a.steps, n_emit=5, clip=True,
               guidance=dict(enabled=True, before=a.before, niters=1, lr=a.lr), seed=5)
    torch.cuda.synchronize()
    print("step  active rows   wavefronts with an active row   by mode (0,1,2)          active lanes per such wavefront")
    for i, (r, w, m, l) in enumerate(rows):
        print("%4d   %.3f          %.3f                         %.3f %.3f %.3f        %.1f" % (a.before - i, r, w, m[0], m[1], m[2], l))


if __name__ == "__main__":
    main()
