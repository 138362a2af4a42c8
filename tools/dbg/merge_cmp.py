"""Saves pstl_refine outputs (merge_net architecture, 512 scenes, fixed seeds) to the file named on the command line, so that
two builds of the library can be compared bit for bit (tools/dbg/variant_run.sh for the other build).  GPU only."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd import ffi
from pstl_diffusion_policy_amd.engine import Sampler, PackedWeights, SceneBatch
from pstl_diffusion_policy_amd.nusc_model import init_state_dict
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
dev = torch.device("cuda:0"); hp = default_hparams()
bs, S, K = 512, 64, 2
w = PackedWeights(init_state_dict(1007), dev)
scene = make_scene_batch(bs, K=K, S=S, seed=3, stlp_mode="wide")
scene = {k: v.to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
sb = SceneBatch(scene, S, hp, dev); sm = Sampler(w, hp)
_, _, base_r = sm.encode(sb)
g = torch.Generator(device=dev).manual_seed(1)
init = (torch.randn(sb.N, 40, device=dev, generator=g) * 0.3).clamp(-0.5, 0.5)
scores = torch.randn(sb.N, device=dev, generator=g)
out = sm.refine(sb, base_r, init, scores, diverse=True)
torch.save(out.cpu(), sys.argv[1])
print("saved", float(out.double().sum()))
