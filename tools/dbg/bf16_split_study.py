"""Numerical feasibility of split-bf16 MFMA for the denoiser chain: emulate, on the CPU, the 99-step
reverse diffusion of e5_steps100 with every fp32 product replaced by products of bf16 pieces accumulated in fp32, and
compare the sampled controls with the plain fp32 chain.  n_pieces=3 keeps 6 of the 9 cross products (hi*hi, hi*mid,
mid*hi, hi*lo, lo*hi, mid*mid); n_pieces=2 keeps 3 (hi*hi, hi*lo, lo*hi)."""
import sys

import numpy as np
import torch

sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import golden_meta, golden_weights, load_golden, scene_from_golden  # noqa: E402
from oracle import pstl_oracle as orc  # noqa: E402
from pstl_diffusion_policy_amd.synthetic import default_hparams  # noqa: E402


def split(x, n):
    parts, r = [], x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        parts.append(p)
        r = r - p
    return parts


def mm_split(x, w, n):   # x (N,K) @ w (O,K)^T with bf16 pieces, fp32 accumulation
    xs, ws = split(x, n), split(w, n)
    pairs = [(0, 0), (0, 1), (1, 0)] if n == 2 else [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
    out = torch.zeros(x.shape[0], w.shape[0])
    for i, j in reversed(pairs):            # small terms first
        out = out + xs[i] @ ws[j].T
    return out


def run(n_pieces):
    d = load_golden("e5_steps100"); meta = golden_meta(d); hp = default_hparams()
    sd = {k: torch.from_numpy(v) for k, v in golden_weights().items()}
    ref = d["controls_list"][-1]
    if n_pieces == 0:
        lin = lambda x, w, b: x @ w.T + b
    else:
        lin = lambda x, w, b: mm_split(x, w, n_pieces) + b
    orig = orc.relu_mlp

    def mlp(sd_, prefix, x):
        if prefix != "policy_net":
            return orig(sd_, prefix, x)
        h = torch.relu(lin(x, orc._t(sd_[prefix + ".0.weight"]), orc._t(sd_[prefix + ".0.bias"])))
        h = torch.relu(lin(h, orc._t(sd_[prefix + ".2.weight"]), orc._t(sd_[prefix + ".2.bias"])))
        return lin(h, orc._t(sd_[prefix + ".4.weight"]), orc._t(sd_[prefix + ".4.bias"]))
    orc.relu_mlp = mlp
    try:
        out = orc.sampling_region(golden_weights(), scene_from_golden(d), meta["S"], meta["steps"], hp, d["x_T"], d["z"])
    finally:
        orc.relu_mlp = orig
    err = np.abs(out["controls_list"].numpy()[-1] - ref)
    return float(err.max()), float(np.mean(err <= 1e-4))


if __name__ == "__main__":
    for n in (0, 3, 2):
        mx, frac = run(n)
        print("pieces=%d: max |d controls| after 99 steps = %.3e, fraction within 1e-4 = %.5f" % (n, mx, frac))
