"""Times pstl_refine (k_merge_pool + the RefineNet chain launch) at the bench shape, 786 432 rows:
    python tools/dbg/refine_time.py      (GPU only; with tools/dbg/variant_run.sh for other builds)"""
import sys, os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd.engine import Sampler, PackedWeights, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

dev = torch.device("cuda:0")
hp = default_hparams()
bs, S, K = 4096, 64, 2
w = PackedWeights(init_state_dict(1007), dev)
scene = make_scene_batch(bs, K=K, S=S, seed=3, stlp_mode="wide")
scene = {k: v.to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
sb = SceneBatch(scene, S, hp, dev)
sm = Sampler(w, hp)
_, _, base_r = sm.encode(sb)
init = (torch.randn(sb.N, 40, device=dev) * 0.3).clamp(-0.5, 0.5)
scores = torch.randn(sb.N, device=dev)
for diverse in (True, False):
    ts = []
    for rep in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = sm.refine(sb, base_r, init, scores, diverse=diverse)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("refine diverse=%s  %.3f ms (min %.3f)  checksum %.6f" % (diverse, sorted(ts)[len(ts) // 2], min(ts), float(out.double().sum())))
