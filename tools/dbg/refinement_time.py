"""Times --refinement (pstl_refinement: 50 Adam iterations over mixing weights) at full size -- 4096 scenes x 64 x 3 rows, a
100-step rollout's list -- and saves the refined controls so that two builds can be compared bit for bit:
    python tools/dbg/refinement_time.py /tmp/out.pt        (GPU only; other builds through tools/dbg/variant_run.sh)"""
import sys, os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd.engine import Sampler, PackedWeights, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

dev = torch.device("cuda:0")
hp = default_hparams()
bs, S, K, steps = int(os.environ.get("SCENES", "4096")), 64, 2, 100
w = PackedWeights(init_state_dict(1007), dev)
scene = make_scene_batch(bs, K=K, S=S, seed=3, stlp_mode="wide")
scene = {k: v.to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
sb = SceneBatch(scene, S, hp, dev)
sm = Sampler(w, hp)
out = sm.sampling_region(sb, steps, None, None, rect_head=True, multi_cands=5, seed=5, want_scores3=False, full_list=True)
clist = out["controls_list"]
ctrl = out["final_controls"].reshape(sb.N, 40).contiguous()
frac = float((out["final_scores"] <= 0).float().mean())
ts = []
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ref = sm.refinement(sb, ctrl, clist, iters=50)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("refinement %.2f ms (min %.2f), %.1f %% of the rows mixed, checksum %.9f" % (sorted(ts)[len(ts) // 2], min(ts), 100 * frac,
                                                                                 float(ref.double().sum())))
if len(sys.argv) > 1:
    torch.save(ref.cpu(), sys.argv[1])
