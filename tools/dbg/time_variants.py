"""Interleaved A/B timing of libpstl_hip.so builds in ONE process on the GPU box (cdna_hip_programming.md rule 24):
    python tools/dbg/time_variants.py [--rounds 7] [--single] [name ...]
Every tools/dbg/_variants/libpstl_<name>.so (all of them when no name is given) against the in-tree library ("base"):
the 39-step denoiser launch of the default workload (786 432 rows, in-kernel noise), optionally (--single) the single-step
launch of the guided phase; median / min of the rounds, and a checksum of the resulting state, so that a variant that
changes a single bit of the output shows."""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pstl_diffusion_policy_amd import ffi  # noqa: E402
from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch  # noqa: E402
from pstl_diffusion_policy_amd.nusc_model import init_state_dict  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch  # noqa: E402

args = sys.argv[1:]
rounds = 7
single = False
chain_waves = 0
BS = 4096
while args and args[0].startswith("--"):
    if args[0] == "--rounds":
        rounds = int(args[1]); args = args[2:]
    elif args[0] == "--single":
        single = True; args = args[1:]
    elif args[0] == "--cw":
        chain_waves = int(args[1]); args = args[2:]
    elif args[0] == "--bs":
        BS = int(args[1]); args = args[2:]
    else:
        raise SystemExit("unknown option " + args[0])
vdir = os.path.join(ROOT, "tools", "dbg", "_variants")
names = args or sorted(os.path.basename(p)[len("libpstl_"):-3] for p in glob.glob(os.path.join(vdir, "libpstl_*.so")))


def load(path):
    L = ctypes.CDLL(path)
    for name, restype, argtypes in ffi.SIGNATURES:
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return L


libs = [("base", ffi.lib())] + [(n, load(os.path.join(vdir, "libpstl_%s.so" % n))) for n in names]
dev = torch.device("cuda:0")
hp = default_hparams()
bs, S, K, steps = BS, 64, 2, 40
sd = init_state_dict(1007)
w = PackedWeights(sd, dev)
scene = make_scene_batch(bs, K=K, S=S, seed=3, stlp_mode="wide")
scene = {k: v.to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
sb = SceneBatch(scene, S, hp, dev)
_, base_p, _ = Sampler(w, hp).encode(sb, need_rect=False)
g = torch.Generator(device=dev).manual_seed(7)
x0 = torch.randn(sb.N, 40, device=dev, generator=g)
times = {n: [] for n, _ in libs}
sums = {}
for rnd in range(rounds + 1):          # round 0 warms every library up
    for n, L in libs:
        ffi._lib = L
        sm = Sampler(w, hp, chain_waves=chain_waves)
        x = x0.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if single:      # one reverse step, mu only: what the guided phase launches ten times
            import ctypes as _c
            from pstl_diffusion_policy_amd.engine import diffusion_coeffs
            beta, alpha, ah = diffusion_coeffs(steps, dev)
            cfg = sb.cfg(steps, ffi.PSTL_FLAG_RNG, chain_waves, 11)
            ffi.check(L.pstl_rollout(_c.byref(cfg), ffi.ptr(w.packed), ffi.ptr(base_p), ffi.ptr(w.tbias(steps)),
                                     ffi.ptr(sb.stlp), ffi.ptr(sb.hl), ffi.ptr(beta), ffi.ptr(alpha), ffi.ptr(ah),
                                     ffi.ptr(None), 5, 5, 1, ffi.ptr(x), ffi.ptr(None), 0, ffi.stream()))
        else:
            sm.rollout(sb, base_p, x, None, steps, n_emit=0, seed=11)
        e1.record()
        torch.cuda.synchronize()
        if rnd > 0:
            times[n].append(e0.elapsed_time(e1))
        sums[n] = (float(x.double().sum()), float(x.double().abs().sum()), int(x.view(torch.int32).sum()))
ffi._lib = libs[0][1]
b = sorted(times["base"])[len(times["base"]) // 2]
for n, _ in libs:
    t = sorted(times[n])
    med = t[len(t) // 2]
    print("%-14s median %.3f ms  min %.3f  (%+.1f %% vs base)  state %s%s" % (
        n, med, t[0], 100.0 * (med / b - 1.0), "identical" if sums[n] == sums["base"] else "DIFFERENT", "" if sums[n] == sums["base"] else " %r" % (sums[n],)))
print("overflow flag:", w.chain_overflowed())
