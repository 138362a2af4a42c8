#!/usr/bin/env python3
"""Fixed and per-tile cost of a single-step (mu_only) k_chain launch, the kind the guided phase issues ten of: time over batch
sizes, least-squares line.   python tools/dbg/single_step_scaling.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd import ffi  # noqa: E402
from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch, diffusion_coeffs  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sm = Sampler(PackedWeights(init_state_dict(1007), dev), hp)
    steps = 50
    beta, alpha, alpha_hat = diffusion_coeffs(steps, dev)
    tb = sm.w.tbias(steps)
    pts = []
    for bs in (512, 1024, 2048, 4096, 8192, 16384):
        scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=2, S=64, seed=1000, stlp_mode="wide").items()
                 if k not in ("params", "pre_stlp", "tj_scores_prior")}
        sb = SceneBatch(scene, 64, hp, dev)
        _, base_p, _ = sm.encode(sb, need_rect=False)
        x = sm.fill_normal(sb, steps, steps, 5)
        cfg = sb.cfg(steps, ffi.PSTL_FLAG_RNG | ffi.PSTL_FLAG_CLIP, 0, 5)
        dbg = torch.empty(1, sb.N, ffi.CTRL, dtype=torch.float32, device=dev)

        def launch(i, mu_only):
            ffi.check(sm.L.pstl_rollout(ctypes.byref(cfg), ffi.ptr(sm.w.packed), ffi.ptr(base_p), ffi.ptr(tb), ffi.ptr(sb.stlp),
                                        ffi.ptr(sb.hl), ffi.ptr(beta), ffi.ptr(alpha), ffi.ptr(alpha_hat), ffi.ptr(None), i, i,
                                        mu_only, ffi.ptr(x), ffi.ptr(dbg), 0, ffi.stream()), "rollout")
        res = {}
        for mu_only in [int(v) for v in os.environ.get("MU_ORDER", "1,0").split(",")]:
            for _ in range(3):
                launch(20, mu_only)
            evs = []
            for rep in range(24):     # queued back to back (a host synchronisation between launches lets the clocks drop)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(20, mu_only)
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ts = [a.elapsed_time(b) for a, b in evs[4:]]
            res[mu_only] = sorted(ts)[len(ts) // 2]
        tiles_per_cu = sb.N / 16 / 256
        pts.append((tiles_per_cu, res[1], res[0]))
        print("%7d rows  %6.1f tiles/CU   mu_only %.4f ms   with noise %.4f ms" % (sb.N, tiles_per_cu, res[1], res[0]))
    t = np.array([p[0] for p in pts])
    for name, col in (("mu_only", 1), ("with noise", 2)):
        y = np.array([p[col] for p in pts])
        a, b = np.polyfit(t[1:], y[1:], 1)
        print("%s: %.2f us fixed + %.3f us per tile-step and CU" % (name, b * 1e3, a * 1e3))


if __name__ == "__main__":
    main()
