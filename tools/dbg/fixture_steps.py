"""Per-step max |controls_list - reference| of a sampling fixture, per chain variant:  python tools/dbg/fixture_steps.py NAME [cw ...]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import golden_meta, golden_weights, load_golden, region_kwargs, scene_from_golden  # noqa: E402
from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams  # noqa: E402
name = sys.argv[1]
dev = torch.device("cuda:0")
d = load_golden(name); meta = golden_meta(d); hp = default_hparams()
sd = {k: v.copy() for k, v in golden_weights().items()}
if meta["zero_net_out"]:
    sd["policy_net.4.weight"] *= 0; sd["policy_net.4.bias"] *= 0
for cw in [int(v) for v in sys.argv[2:]] or [0, 8]:
    sb = SceneBatch({k: torch.from_numpy(v) for k, v in scene_from_golden(d).items()}, meta["S"], hp, dev)
    sm = Sampler(PackedWeights(sd, dev), hp, chain_waves=cw)
    out = sm.sampling_region(sb, meta["steps"], torch.from_numpy(d["x_T"]).to(dev), torch.from_numpy(d["z"]).to(dev), full_list=True, **region_kwargs(meta))
    cl = out["controls_list"].reshape(meta["steps"], sb.N, 20, 2).cpu().numpy()
    err = np.abs(cl - d["controls_list"])
    print(name, "cw", cw, "N", sb.N, "per-step max err:", " ".join("%.1e" % e for e in err.reshape(meta["steps"], -1).max(1)))
    bad = np.nonzero((err > 1e-4).any(axis=(0, 2, 3)))[0]
    print("   rows with err > 1e-4:", bad[:40])
