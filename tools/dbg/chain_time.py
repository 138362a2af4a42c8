"""Times the multi-step denoiser launch (39 reverse steps, 786 432 rows) for a list of chain variants / noise modes:
    python tools/dbg/chain_time.py 0 116 0:nonoise 8
`cw[:nonoise]`; 116 = split-f16 loop without epilogue and noise (timing only; needs a -DPSTL_DIAG build:
tools/dbg/variant_run.sh "-DPSTL_DIAG" tools/dbg/chain_time.py 0 116)."""
import sys, os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pstl_diffusion_policy_amd.engine import Sampler, PackedWeights, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

dev = torch.device("cuda:0")
hp = default_hparams()
bs, S, K, steps = 4096, 64, 2, 40
sd = init_state_dict(1007)
w = PackedWeights(sd, dev)
scene = make_scene_batch(bs, K=K, S=S, seed=3, stlp_mode="wide")
scene = {k: v.to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
sb = SceneBatch(scene, S, hp, dev)
_, base_p, _ = Sampler(w, hp).encode(sb, need_rect=False)
for spec in sys.argv[1:] or ["0"]:
    cw = int(spec.split(":")[0])
    nonoise = spec.endswith(":nonoise")
    sm = Sampler(w, hp, chain_waves=cw)
    sm.debug_buf = torch.zeros(2 * 32 * 8 * 8, dtype=torch.float32, device=dev)
    ts = []
    for rep in range(6):
        x = torch.randn(sb.N, 40, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if nonoise:   # no noise source at all: the kernel adds zeros (timing only)
            import ctypes
            from pstl_diffusion_policy_amd import ffi
            from pstl_diffusion_policy_amd.engine import diffusion_coeffs
            beta, alpha, ah = diffusion_coeffs(steps, dev)
            cfg = sb.cfg(steps, 0, cw, 0)
            ffi.check(sm.L.pstl_rollout(ctypes.byref(cfg), ffi.ptr(w.packed), ffi.ptr(base_p), ffi.ptr(w.tbias(steps)),
                                        ffi.ptr(sb.stlp), ffi.ptr(sb.hl), ffi.ptr(beta), ffi.ptr(alpha), ffi.ptr(ah),
                                        ffi.ptr(None), steps - 1, 1, 0, ffi.ptr(x), ffi.ptr(sm.debug_buf), 0, ffi.stream()))
        else:
            sm.rollout(sb, base_p, x, None, steps, n_emit=0, seed=11 + rep)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("%-12s %.3f ms (min %.3f)" % (spec, sorted(ts)[len(ts) // 2], min(ts)))
