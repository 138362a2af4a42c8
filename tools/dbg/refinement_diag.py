import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from conftest import *
from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
from pstl_diffusion_policy_amd.synthetic import default_hparams
dev=torch.device("cuda:0")
for name in REFINEMENT_CASES:
    d=load_golden(name); meta=golden_meta(d); hp=default_hparams()
    sb=SceneBatch({k: torch.from_numpy(v) for k,v in scene_from_golden(d).items()}, meta["S"], hp, dev)
    sm=Sampler(PackedWeights(golden_weights(), dev), hp)
    clist=torch.from_numpy(d["controls_list"]).reshape(meta["steps"], sb.N, 40).to(dev)
    cin=torch.from_numpy(d["refinement_in_controls"]).reshape(sb.N,40).to(dev)
    out,tr=sm.refinement(sb,cin,clist,iters=50,trace=True)
    got=out.reshape(sb.N,20,2).cpu().numpy(); want=d["refinement_controls"]
    err=np.abs(got-want).reshape(sb.N,-1).max(1)
    print(name,"err quantiles",np.quantile(err,[0.5,0.75,0.9,0.95,1.0]))
    g=tr.cpu().numpy(); gr=d["refinement_grads"]
    for it in (0,1,2,4,9,19,49):
        sc=np.abs(gr[it]).max(1,keepdims=True)+1e-30
        ok=(np.abs(g[it]-gr[it])<=5e-3*sc+1e-9).all(1)
        print("  it",it,"rows with matching grads",ok.mean())
    s_m=sm.score(sb,out.reshape(1,sb.N,40))["scores"][0].cpu().numpy()
    s_r=sm.score(sb,torch.from_numpy(want).reshape(1,sb.N,40).to(dev))["scores"][0].cpu().numpy()
    s_i=sm.score(sb,cin.reshape(1,sb.N,40))["scores"][0].cpu().numpy()
    v=sb.valid.cpu().numpy()>0
    print("  mean relu(thres-score) over valid: in %.4f  mine %.4f  ref %.4f ; sat in %d mine %d ref %d" % (np.maximum(5e-4-s_i,0)[v].mean(), np.maximum(5e-4-s_m,0)[v].mean(), np.maximum(5e-4-s_r,0)[v].mean(), (s_i[v]>0).sum(), (s_m[v]>0).sum(), (s_r[v]>0).sum()))
