"""Cycle stamps of the MLP-chain kernel's diagnostic build (chain_waves 708 = fp32, 716 = split-bf16): where one
workgroup's waves spend an iteration.  GPU only, against a -DPSTL_DIAG build of mlp_kernels.hip (the shipped library has no
diagnostic instantiations):  tools/dbg/variant_run.sh "-DPSTL_DIAG" tools/dbg/chain_stamps.py [716]"""
import sys, os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd.engine import Sampler, PackedWeights, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

cw = int(sys.argv[1]) if len(sys.argv) > 1 else 716   # 716: the split-f16 kernel
dev = torch.device("cuda:0")
hp = default_hparams()
bs, S, K, steps = 1024, 64, 2, 50
sd = init_state_dict(1007)
sm = Sampler(PackedWeights(sd, dev), hp, chain_waves=cw)
scene = make_scene_batch(bs, K=K, S=S, seed=3, stlp_mode="wide")
scene = {k: v.to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
sb = SceneBatch(scene, S, hp, dev)
_, base_p, _ = Sampler(sm.w, hp).encode(sb, need_rect=False)
x = torch.randn(sb.N, 40, device=dev)
sm.debug_buf = torch.zeros(2 * 32 * 8 * 8, dtype=torch.float32, device=dev)   # 64-bit ticks
sm.rollout(sb, base_p, x, None, steps, n_emit=0, seed=11)
torch.cuda.synchronize()
t = sm.debug_buf.cpu().numpy().view(np.int64).reshape(32, 8, 8)   # [iteration][wave][slot], shader cycles (s_memtime)
per_it = (t[1:, :, 0] - t[:-1, :, 0]).mean()
print("iteration: %.0f cycles" % per_it)
if cw == 716:   # slots: 0 start, 1 before the fused block, 6 after k-block 1, 7 after k-block 4, 2 after k-block 6, 3 end of layer 2,
    order = [0, 1, 6, 7, 2, 3, 4, 5]                                  # 4 layer 3 + partial sums written, 5 behind the barrier
    names = ["epi/noise/split", "kb0-1", "kb2-4", "kb5-6", "kb7", "layer3+part", "barrier"]
else:
    order = [0, 1, 2, 3, 4, 5]
    names = ["epilogue", "layer1", "layer2", "layer3+part", "barrier"]
for w in range(8):
    d = [(t[:, w, order[s + 1]] - t[:, w, order[s]]).mean() for s in range(len(order) - 1)]
    print("wave %d: " % w + "  ".join("%s %.0f" % (n, v) for n, v in zip(names, d)))
# when each wave passes each stamp, relative to the earliest wave's iteration start (one iteration, averaged)
rel = (t[:, :, order] - t[:, :, 0].min(axis=1)[:, None, None]).mean(axis=0)
print("arrival (cycles after the first wave's iteration start), rows = waves, cols = " + " | ".join(["start"] + names))
print(np.round(rel).astype(int))
