"""Does a rollout in the latency layout of k_chain depend on LDS it never wrote?  Poisons every CU's LDS with NaN patterns
(tests/ldspoison), runs ONE multi-step rollout of `scenes` scenes and prints the overflow flag and the state's finiteness.
    python tools/dbg/with_lib.py <lib> tools/dbg/poison_probe.py [scenes]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda:0")
hp = default_hparams()
w = PackedWeights(init_state_dict(1007, rect_head=True, diverse_loss=True), dev)
sm = Sampler(w, hp)
scene = {k: v.to(dev) for k, v in make_scene_batch(scenes, K=2, S=64, seed=1, stlp_mode="wide").items()
         if k not in ("params", "pre_stlp", "tj_scores_prior")}
sb = SceneBatch(scene, 64, hp, dev)
_, base_p, _ = sm.encode(sb, need_rect=False)
x = torch.randn(sb.N, 40, device=dev)
torch.cuda.synchronize()
P = ctypes.CDLL(os.path.join(ROOT, "tests", "ldspoison", "liblds_poison.so"))
P.lds_poison.argtypes = [ctypes.c_void_p]
for trial in range(3):
    assert P.lds_poison(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    xx = x.clone()
    torch.cuda.synchronize()
    sm.rollout(sb, base_p, xx, None, 50, n_emit=0, seed=5 + trial)
    torch.cuda.synchronize()
    print("trial %d: overflow flag %s, state finite %s, status words %s" % (
        trial, w.chain_overflowed(clear=True), bool(torch.isfinite(xx).all()), w.status.tolist() if hasattr(w, "status") else "?"))
