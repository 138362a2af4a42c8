"""Times pstl_diversity at the bench shape (4096 scenes x 64 x 3) on realistic inputs (a sampled batch):
    python tools/dbg/div_time.py        (GPU only; other builds through tools/dbg/with_lib.py)"""
import sys, os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd.engine import Sampler, PackedWeights, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

dev = torch.device("cuda:0")
hp = default_hparams()
bs, S, K = int(os.environ.get("DIV_BS", "4096")), 64, 2
w = PackedWeights(init_state_dict(1007), dev)
scene = make_scene_batch(bs, K=K, S=S, seed=3, stlp_mode="wide")
scene = {k: v.to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
sb = SceneBatch(scene, S, hp, dev)
sm = Sampler(w, hp)
out = sm.sampling_region(sb, 50, None, None, rect_head=True, multi_cands=5, seed=5, want_scores3=False,
                         guidance=dict(enabled=True, before=10, niters=1, lr=0.01))
ctrl, sc = out["final_controls"].reshape(sb.N, 40).contiguous(), out["final_scores"].contiguous()
ts = []
for rep in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    pm, ps, tot = sm.diversity(sb, ctrl, sc)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("diversity %.3f ms (min %.3f)  totals %s" % (sorted(ts)[len(ts) // 2], min(ts), ["%.6g" % v for v in tot.tolist()[:7]]))
