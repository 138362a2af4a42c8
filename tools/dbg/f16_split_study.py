"""Numerical study for the fp32-faithful default chain: the 99-step reverse diffusion of e5_steps100 emulated on the CPU with
every fp32 product formed from two fp16 pieces (hi = f16(v), lo = f16((v - hi) * 2^11)), the three products
hi*hi | (lo*hi + hi*lo) accumulated in two fp32 accumulators and combined as main + 2^-11 * corr.  Compared with the
reference's fp32 result and with variants (no lo scaling, bf16 pieces, 4th product)."""
import sys

import numpy as np
import torch

sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import golden_meta, golden_weights, load_golden, scene_from_golden  # noqa: E402
from oracle import pstl_oracle as orc  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams  # noqa: E402

SC = 2048.0


def pieces(v, dt, scale):
    hi = v.to(dt).to(torch.float32)
    lo = ((v - hi) * scale).to(dt).to(torch.float32)
    return hi, lo


def mm_split(x, w, dt, scale, four=False):
    xh, xl = pieces(x, dt, scale)
    wh, wl = pieces(w, dt, scale)
    corr = xl @ wh.T + xh @ wl.T
    main = xh @ wh.T
    if four:
        corr = corr + (xl @ wl.T) / scale
    return main + corr / scale


def run(name, mode):
    d = load_golden(name); meta = golden_meta(d); hp = default_hparams()
    ref = d["controls_list"][-1]
    ext = list(range(224, 264)) + list(range(296, 303))
    hoisted = list(range(0, 224)) + list(range(264, 296))
    if mode == "fp32":
        mm = lambda x, w: x @ w.T
    elif mode == "f64":
        mm = lambda x, w: (x.double() @ w.double().T).float()
    elif mode == "bf16x2":
        mm = lambda x, w: mm_split(x, w, torch.bfloat16, 1.0)
    elif mode == "f16x2_unscaled":
        mm = lambda x, w: mm_split(x, w, torch.float16, 1.0)
    elif mode == "f16x2":
        mm = lambda x, w: mm_split(x, w, torch.float16, SC)
    elif mode == "f16x2_4p":
        mm = lambda x, w: mm_split(x, w, torch.float16, SC, True)
    orig = orc.relu_mlp

    def mlp(sd, prefix, x):
        if prefix != "policy_net":
            return orig(sd, prefix, x)
        w1, b1 = orc._t(sd[prefix + ".0.weight"]), orc._t(sd[prefix + ".0.bias"])
        h = torch.relu(x[:, hoisted] @ w1[:, hoisted].T + b1 + mm(x[:, ext], w1[:, ext]))
        h = torch.relu(mm(h, orc._t(sd[prefix + ".2.weight"])) + orc._t(sd[prefix + ".2.bias"]))
        return mm(h, orc._t(sd[prefix + ".4.weight"])) + orc._t(sd[prefix + ".4.bias"])
    orc.relu_mlp = mlp
    try:
        out = orc.sampling_region(golden_weights(), scene_from_golden(d), meta["S"], meta["steps"], hp, d["x_T"], d["z"])
    finally:
        orc.relu_mlp = orig
    err = np.abs(out["controls_list"].numpy()[-1] - ref)
    return float(err.max()), float(err.mean())


if __name__ == "__main__":
    for name in ("e5_steps100",):
        for mode in ("fp32", "f64", "bf16x2", "f16x2_unscaled", "f16x2", "f16x2_4p"):
            mx, mean = run(name, mode)
            print("%s %-16s max |d controls| = %.3e  mean = %.3e" % (name, mode, mx, mean))


def mm_ranged(x, w, sx, sw):
    """variant (A): operands pre-scaled by powers of two into fp16's comfortable range, pieces unscaled relative to each
    other, ONE fp32 accumulator (small terms added first here; the kernel interleaves them)."""
    xs, ws = x * sx, w * sw
    xh = xs.to(torch.float16).to(torch.float32); xl = (xs - xh).to(torch.float16).to(torch.float32)
    wh = ws.to(torch.float16).to(torch.float32); wl = (ws - wh).to(torch.float16).to(torch.float32)
    return ((xl @ wh.T + xh @ wl.T) + xh @ wh.T) / (sx * sw)


if __name__ == "__main__":
    import itertools
    _run = run
    for sx, sw in ((64.0, 1024.0), (16.0, 256.0), (1.0, 1024.0), (64.0, 1.0)):
        d = load_golden("e5_steps100")
        globals()["_sx"], globals()["_sw"] = sx, sw
        orig_mm = mm_split
        def run_ranged():
            import types
            d = load_golden("e5_steps100"); meta = golden_meta(d); hp = default_hparams()
            ref = d["controls_list"][-1]
            ext = list(range(224, 264)) + list(range(296, 303))
            hoisted = list(range(0, 224)) + list(range(264, 296))
            mm = lambda x, w: mm_ranged(x, w, sx, sw)
            orig = orc.relu_mlp
            def mlp(sd, prefix, x):
                if prefix != "policy_net":
                    return orig(sd, prefix, x)
                w1, b1 = orc._t(sd[prefix + ".0.weight"]), orc._t(sd[prefix + ".0.bias"])
                h = torch.relu(x[:, hoisted] @ w1[:, hoisted].T + b1 + mm(x[:, ext], w1[:, ext]))
                h = torch.relu(mm(h, orc._t(sd[prefix + ".2.weight"])) + orc._t(sd[prefix + ".2.bias"]))
                return mm(h, orc._t(sd[prefix + ".4.weight"])) + orc._t(sd[prefix + ".4.bias"])
            orc.relu_mlp = mlp
            try:
                out = orc.sampling_region(golden_weights(), scene_from_golden(d), meta["S"], meta["steps"], hp, d["x_T"], d["z"])
            finally:
                orc.relu_mlp = orig
            err = np.abs(out["controls_list"].numpy()[-1] - ref)
            return float(err.max()), float(err.mean())
        mx, mean = run_ranged()
        print("ranged sx=%g sw=%g: max %.3e mean %.3e" % (sx, sw, mx, mean))
