"""Role / body / barrier-wait cycles per wave of the split-f16 chain kernel (diagnostic build, chain_waves 816: four
s_memtime reads per iteration).  GPU only, against a -DPSTL_DIAG build:
    tools/dbg/variant_run.sh "-DPSTL_DIAG" tools/dbg/chain_phases.py      (or with_lib.py + a prebuilt _variants library)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd.engine import Sampler, PackedWeights, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

dev = torch.device("cuda:0")
hp = default_hparams()
bs, S, K, steps = 4096, 64, 2, 40
sm = Sampler(PackedWeights(init_state_dict(1007), dev), hp, chain_waves=816)
scene = make_scene_batch(bs, K=K, S=S, seed=3, stlp_mode="wide")
scene = {k: v.to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
sb = SceneBatch(scene, S, hp, dev)
_, base_p, _ = Sampler(sm.w, hp).encode(sb, need_rect=False)
for rep in range(3):
    x = torch.randn(sb.N, 40, device=dev)
    sm.debug_buf = torch.zeros(2 * 8 * 4, dtype=torch.float32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sm.rollout(sb, base_p, x, None, steps, n_emit=0, seed=11)
    e1.record()
    torch.cuda.synchronize()
t = sm.debug_buf.cpu().numpy().view(np.int64).reshape(8, 4).astype(np.float64)
it = t[:, 3]
print("launch %.3f ms; per iteration and wave (cycles): role (epilogue / noise fetch / input split), body (fused block + "
      "partial sums), barrier wait" % e0.elapsed_time(e1))
for w in range(8):
    print("wave %d: role %6.0f  body %6.0f  barrier %6.0f  | total %6.0f" % (
        w, t[w, 0] / it[w], t[w, 1] / it[w], t[w, 2] / it[w], t[w, :3].sum() / it[w]))
