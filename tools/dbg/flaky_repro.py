"""Hunts a non-deterministic failure: the same sampling region many times in one process (seeded in-kernel noise), counting the
calls after which the split-f16 overflow flag is set or an output is not finite, and the calls whose outputs differ from the
first call with the same seed.   python tools/dbg/flaky_repro.py [scenes] [calls]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 48
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda:0")
hp = default_hparams()
w = PackedWeights(init_state_dict(1007, rect_head=True, diverse_loss=True), dev)
sm = Sampler(w, hp, chain_waves=int(os.environ.get("PSTL_CHAIN_WAVES", "0")))
scene = {k: v.to(dev) for k, v in make_scene_batch(scenes, K=2, S=64, seed=1, stlp_mode="wide").items()
         if k not in ("params", "pre_stlp", "tj_scores_prior")}
g = dict(enabled=True, before=10, niters=1, lr=0.01)
ref = {}
bad = diff = 0
poison = None
if os.environ.get("PSTL_POISON"):      # NaN patterns in every CU's LDS before each call (tests/ldspoison)
    import ctypes
    poison = ctypes.CDLL(os.path.join(ROOT, "tests", "ldspoison", "liblds_poison.so"))
    poison.lds_poison.argtypes = [ctypes.c_void_p]
for i in range(calls):
    seed = 11 + (i % 4)
    if poison is not None:
        assert poison.lds_poison(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    sb = SceneBatch(scene, 64, hp, dev)
    out = sm.sampling_region(sb, 50, None, None, rect_head=True, multi_cands=5, guidance=g, seed=seed, want_scores3=False)
    torch.cuda.synchronize()
    flag = w.chain_overflowed(clear=True)
    fin = bool(torch.isfinite(out["final_controls"]).all()) and bool(torch.isfinite(out["final_scores"]).all())
    if flag or not fin:
        bad += 1
        print("call %d: overflow flag %s, finite %s" % (i, flag, fin))
    key = (out["final_controls"].clone(), out["final_scores"].clone(), out["sel_controls"].clone())
    if seed in ref:
        if not all(torch.equal(a, b) for a, b in zip(key, ref[seed])):
            diff += 1
            d = (key[0] - ref[seed][0]).abs()
            print("call %d (seed %d) differs from the first call with that seed: %d elements, max %g; sel_controls differ: %s"
                  % (i, seed, int((d > 0).sum()), float(d.max()), not torch.equal(key[2], ref[seed][2])))
    else:
        ref[seed] = key
print("%d calls at %d scenes: %d flagged / non-finite, %d differing from their first run" % (calls, scenes, bad, diff))
