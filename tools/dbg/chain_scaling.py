"""Is the multi-step chain launch limited by the chip's power envelope?  One workgroup (192 rows) per CU at most: the time
of a 39-step launch over 32 ... 256 workgroups (= busy CUs) and then 2, 4, 16 workgroups per CU.  If the clock is held down
under full load, the launch on a fraction of the CUs finishes sooner than the one that fills them all.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd.engine import Sampler, PackedWeights, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

dev = torch.device("cuda:0")
hp = default_hparams()
S, K, steps = 64, 2, 40
w = PackedWeights(init_state_dict(1007), dev)
cw = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for bs in (32, 64, 128, 192, 256, 512, 1024, 4096):
    scene = make_scene_batch(bs, K=K, S=S, seed=3, stlp_mode="wide")
    scene = {k: v.to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
    sb = SceneBatch(scene, S, hp, dev)
    sm = Sampler(w, hp, chain_waves=cw)
    _, base_p, _ = sm.encode(sb, need_rect=False)
    ts = []
    for rep in range(7):
        x = torch.randn(sb.N, 40, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sm.rollout(sb, base_p, x, None, steps, n_emit=0, seed=11 + rep)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2]
    rounds = (bs + 255) // 256
    print("chain_waves %d: %5d workgroups (%4.1f per CU): %.3f ms  = %.3f ms per round of workgroups, %.2f us per tile-step"
          % (cw, bs, bs / 256.0, t, t / rounds, t * 1e3 / rounds / (12 * 39)))
