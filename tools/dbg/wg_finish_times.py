"""When every workgroup of a one-round multi-step k_chain launch finishes, and on which CU: a
-DPSTL_WG_TIMES build of mlp_kernels.hip records s_memrealtime (100 MHz) at each workgroup's first and last instruction and its
HW_ID / XCC_ID.    tools/dbg/build_variants.sh wgtimes:"-DPSTL_WG_TIMES"   (here)
                   python tools/dbg/wg_finish_times.py [scenes=128] [steps=100]   (GPU box)
Prints the histogram of finish times relative to the first start, per-CU statistics, and what the launch would take if every
workgroup took the median time."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pstl_diffusion_policy_amd import ffi  # noqa: E402

L = ctypes.CDLL(os.path.join(ROOT, "tools", "dbg", "_variants", "libpstl_wgtimes.so"))
for n, r, a in ffi.SIGNATURES:
    f = getattr(L, n)
    f.restype, f.argtypes = r, a
ffi._lib = L
from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
hp = default_hparams()
sm = Sampler(PackedWeights(init_state_dict(1007), dev), hp, chain_waves=16)
scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=2, S=64, seed=3, stlp_mode="wide").items() if k not in ("pre_stlp", "tj_scores_prior")}
sb = SceneBatch(scene, 64, hp, dev)
_, base_p, _ = sm.encode(sb, need_rect=False)
kern, tiles, rounds = ffi.rollout_layout(sb.cfg(steps, ffi.PSTL_FLAG_RNG, 16, 0))
n_wg = (sb.N // 16 + tiles - 1) // tiles
print("%d rows, %d steps: kernel %d, %d tiles per workgroup, %d workgroups, %d round(s)" % (sb.N, steps, kern, tiles, n_wg, rounds))
sm.debug_buf = torch.zeros(4 * 4096 * 2, dtype=torch.float32, device=dev)
for rep in range(3):
    x = torch.randn(sb.N, 40, device=dev)
    sm.debug_buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sm.rollout(sb, base_p, x, None, steps, n_emit=0, seed=11)
    e1.record()
    torch.cuda.synchronize()
t = sm.debug_buf.cpu().numpy().view(np.int64).reshape(-1, 4)[:n_wg]
t0, t1, hw, xcc = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
first = t0.min()
start, end, dur = (t0 - first) / 100.0, (t1 - first) / 100.0, (t1 - t0) / 100.0   # microseconds (100 MHz)
cu = (xcc & 0xF) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xF)   # xcc | se | sh | cu
print("launch by HIP events %.1f us; first start -> last end %.1f us" % (e0.elapsed_time(e1) * 1e3, end.max()))
print("workgroup start: median %.1f us, 99th %.1f, max %.1f after the first" % (np.median(start), np.percentile(start, 99), start.max()))
print("workgroup duration: min %.1f  median %.1f  mean %.1f  90th %.1f  99th %.1f  max %.1f us" % (
    dur.min(), np.median(dur), dur.mean(), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("distinct CUs used: %d (workgroups per CU: max %d)" % (len(set(cu.tolist())), np.bincount(np.unique(cu, return_inverse=True)[1]).max()))
edges = np.linspace(np.floor(end.min() / 10) * 10, np.ceil(end.max() / 10) * 10, 16)
h, _ = np.histogram(end, bins=edges)
print("finish times (us after the first start):")
for k in range(len(h)):
    print("  %7.0f .. %7.0f  %4d %s" % (edges[k], edges[k + 1], h[k], "#" * int(60 * h[k] / max(h.max(), 1))))
by_xcc = {}
for x_, d_ in zip((xcc & 0xF).tolist(), dur.tolist()):
    by_xcc.setdefault(x_, []).append(d_)
print("duration by XCC: " + "  ".join("%d: %.1f (n %d)" % (k, np.mean(v), len(v)) for k, v in sorted(by_xcc.items())))
print("if every workgroup took the median: %.1f us + the start spread; the slowest took %.1f %% longer than the median" % (
    np.median(dur), 100 * (dur.max() / np.median(dur) - 1)))
