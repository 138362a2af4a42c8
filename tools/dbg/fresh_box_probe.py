"""Run as the FIRST GPU process of a gpurun call (a fresh box): what the LDS holds before anything of ours ran, then the
bench's small workload step by step with the split-f16 overflow flag read after every call."""
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
P = ctypes.CDLL(os.path.join(ROOT, "tests", "ldspoison", "liblds_poison.so"))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = torch.zeros(1024, dtype=torch.int32, device="cuda")
for idx in (0, 9000, 20000, 33000, 36000, 38000, 39500):
    P.lds_peek(idx, 1024, ctypes.c_void_p(out.data_ptr()), st)
    torch.cuda.synchronize()
    c = collections.Counter(("%08x" % (v & 0xffffffff)) for v in out.cpu().tolist())
    print("fresh LDS word %5d: %d distinct values over 1024 workgroups, most common %s" % (idx, len(c), c.most_common(3)))

from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda:0")
hp = default_hparams()
w = PackedWeights(init_state_dict(1007, rect_head=True, diverse_loss=True), dev)
sm = Sampler(w, hp)
scene = {k: v.to(dev) for k, v in make_scene_batch(scenes, K=2, S=64, seed=77, stlp_mode="wide").items()
         if k not in ("params", "pre_stlp", "tj_scores_prior")}
g = dict(enabled=True, before=10, niters=1, lr=0.01)
for i in range(6):
    sb = SceneBatch(scene, 64, hp, dev)
    out = sm.sampling_region(sb, 50, None, None, rect_head=True, multi_cands=5, guidance=g, seed=1000 + i, want_scores3=False,
                             diversity=True)
    torch.cuda.synchronize()
    print("call %d: overflow flag %s, controls finite %s, scores finite %s" % (
        i, w.chain_overflowed(clear=True), bool(torch.isfinite(out["final_controls"]).all()),
        bool(torch.isfinite(out["final_scores"]).all())))
