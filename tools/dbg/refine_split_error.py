"""How much of the split-bf16 / fp32 difference in the REFINED controls comes from RefineNet's own arithmetic and how much
from its inputs (the sampled controls, which differ by ~3e-5 after 99 steps)?  GPU only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import golden_weights  # noqa: E402
from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

dev = torch.device("cuda:0")
hp = default_hparams()
BS, S, K = 4096, 64, 2
scene = {k: v.to(dev) for k, v in make_scene_batch(BS, K=K, S=S, seed=3, invalid_lane_frac=0.2, stlp_mode="wide").items()
         if k not in ("pre_stlp", "tj_scores_prior")}
w = PackedWeights(golden_weights(), dev)
sb = SceneBatch(scene, S, hp, dev)
sm = {cw: Sampler(w, hp, chain_waves=cw) for cw in (8, 16)}
o = {cw: sm[cw].sampling_region(sb, 100, None, None, rect_head=True, multi_cands=3, seed=77, want_scores3=False) for cw in (8, 16)}
_, _, base_r = sm[8].encode(sb)
same = (o[8]["sel_idx"] == o[16]["sel_idx"]).reshape(BS, 3 * S).all(dim=1).repeat_interleave(3 * S)
clear = same & (o[8]["sel_scores"].abs() > 1e-3) & (o[16]["sel_scores"].abs() > 1e-3)
d = lambda a, b: (a - b).abs().reshape(sb.N, -1).amax(dim=1)[clear].max().item()
print("sampled controls, bf16x3 rollout vs fp32 rollout: %.3e" % d(o[8]["sel_controls"], o[16]["sel_controls"]))
print("refined controls, all-bf16x3 vs all-fp32:          %.3e" % d(o[8]["final_controls"], o[16]["final_controls"]))
r = {(a, b): sm[b].refine(sb, base_r, o[a]["sel_controls"].reshape(sb.N, 40), o[a]["sel_scores"]) for a in (8, 16) for b in (8, 16)}
print("same fp32-rollout inputs, RefineNet bf16x3 vs fp32:  %.3e" % d(r[(8, 8)], r[(8, 16)]))
print("fp32 RefineNet, bf16x3-rollout vs fp32-rollout inputs: %.3e" % d(r[(8, 8)], r[(16, 8)]))
