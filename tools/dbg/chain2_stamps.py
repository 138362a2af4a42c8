"""Section shares of k_chain2's tile-step from a -DPSTL_C2_STAMP build (tools/dbg/build_variants2.sh stamp:"-DPSTL_C2_STAMP"):
    python tools/dbg/chain2_stamps.py [--single] [variant-name, default "stamp"]
Runs the 39-step denoiser launch (--single: the guided phase's mu-only launch of one reverse step, whose workgroups walk tiles) of the default workload (786 432 rows, in-kernel noise) and prints, per wave of workgroup
7, the cycles per tile-step of each section and the in-kernel clock (shader cycles / realtime ticks x 100 MHz)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pstl_diffusion_policy_amd import ffi  # noqa: E402
from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch, diffusion_coeffs  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

single = "--single" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--single"]
name = argv[0] if argv else "stamp"
L = ctypes.CDLL(os.path.join(ROOT, "tools", "dbg", "_variants", "libpstl_%s.so" % name))
for n, restype, argtypes in ffi.SIGNATURES:
    fn = getattr(L, n)
    fn.restype, fn.argtypes = restype, argtypes
dev = torch.device("cuda:0")
hp = default_hparams()
bs, S, K, steps = 4096, 64, 2, 40
w = PackedWeights(init_state_dict(1007), dev)
scene = make_scene_batch(bs, K=K, S=S, seed=3, stlp_mode="wide")
scene = {k: v.to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
sb = SceneBatch(scene, S, hp, dev)
_, base_p, _ = Sampler(w, hp).encode(sb, need_rect=False)
beta, alpha, ah = diffusion_coeffs(steps, dev)
cfg = sb.cfg(steps, ffi.PSTL_FLAG_RNG, 2, 11)
dbg = torch.zeros(48, dtype=torch.int64, device=dev)
names = ["layer 1", "chunk 0", "chunk 1", "A phases of chunks 2-5", "B phases of chunks 2-5", "chunks 6-7", "tail", "epilogue"]
for rep in range(3):
    x = torch.randn(sb.N, 40, device=dev)
    ffi.check(L.pstl_rollout(ctypes.byref(cfg), ffi.ptr(w.packed), ffi.ptr(base_p), ffi.ptr(w.tbias(steps)), ffi.ptr(sb.stlp),
                             ffi.ptr(sb.hl), ffi.ptr(beta), ffi.ptr(alpha), ffi.ptr(ah), ffi.ptr(None),
                             5 if single else steps - 1, 5 if single else 1, 1 if single else 0, ffi.ptr(x), ctypes.c_void_p(dbg.data_ptr()), 0, ffi.stream()))
    torch.cuda.synchronize()
d = dbg.cpu().reshape(-1, 12)[:4]
for wv in range(4):
    r = d[wv].tolist()
    nst = max(r[10], 1)
    clk = r[8] / max(r[9], 1) * 0.1
    print("wave %d: %.0f cycles per tile-step at %.2f GHz (%.2f us): " % (wv, r[8] / nst, clk, r[8] / nst / clk / 1e3) +
          ", ".join("%s %.0f" % (names[k], r[k] / nst) for k in range(8)))
