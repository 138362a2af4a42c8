"""Weight sets with a trained network's dynamic range, derived EXACTLY from tests/golden/weights_seed1007.npz.

TEST INFRASTRUCTURE.  Every sampling / training fixture of rounds 1-2 used nn.Linear's default initialisation (|w| <=
1/sqrt(fan_in), hidden activations O(1)).  The default MLP-chain arithmetic carries operands as IEEE-half pieces of
2^10 w and 2^4 x (csrc/mlp_kernels.hip), exact to 2^-23 only while the pieces stay in half's range, so parity must also be
pinned where weights are heavy-tailed and hidden activations are O(10-100).  The derivation uses integer draws and
power-of-two factors only, so the reference run that records a fixture (tests/golden/make_golden.py --heavy, in the build
container) and the tests on the GPU box see bit-identical weights without the fixture having to carry 2 MB of them:

  * rows of the two hidden layers of policy_net / rect_net scaled by 2^k (k drawn per output row: a discrete log-normal),
    the output layer scaled down by a fixed power of two so that the predicted noise stays O(1);
  * a few outlier weights per matrix, +-(4 .. `out_max`) in steps of 1/2, and outlier biases +-2;
  * the scene encoders' hidden rows scaled by 2^k, k in {0, 1} (the 224-wide feature grows with them).

variant "a": hidden rows 2^0..2^3, outliers up to 8.      variant "b": hidden rows 2^0..2^4, outliers up to 24.
variant "w70": the random-init weights with ONE policy_net weight set to 70 -- outside the split-f16 domain (|w| < 64):
the packer must notice and the chains must fall back to the exact-fp32 kernel.
"""
import numpy as np

CHAINS = ("policy_net", "rect_net")
ENCODERS = ("ego_encoder", "neighbor_encoder", "lane_encoder")

VARIANTS = {
    # k1, k2: exponent ranges [lo, hi] of the per-row factors of layers 1, 2; m3: output layer scaled by 2^-m3
    "a": dict(seed=7001, k1=(0, 3), k2=(-1, 2), m3=5, out_max=8.0, n_out=10, enc_k=(0, 1)),
    "b": dict(seed=7002, k1=(0, 4), k2=(-1, 3), m3=7, out_max=24.0, n_out=16, enc_k=(0, 1)),
}


def _pow2(k):
    return np.ldexp(np.float32(1.0), k).astype(np.float32)


def heavy_weights(sd, variant):
    sd = {k: np.array(v, dtype=np.float32, copy=True) for k, v in sd.items()}
    if variant == "w70":
        sd["policy_net.2.weight"][17, 201] = np.float32(70.0)
        return sd
    v = VARIANTS[variant]
    rng = np.random.Generator(np.random.PCG64(v["seed"]))      # integer draws only: exact on every platform

    def rows(name, lo, hi):
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        k = rng.integers(lo, hi + 1, size=w.shape[0])
        f = _pow2(k)
        w *= f[:, None]
        b *= f

    def outliers(name, vmax, n, wmax_cols=None):
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        nvals = int((vmax - 4.0) * 2) + 1
        for _ in range(n):
            r, c = int(rng.integers(0, w.shape[0])), int(rng.integers(0, w.shape[1] if wmax_cols is None else wmax_cols))
            val = np.float32(4.0 + 0.5 * int(rng.integers(0, nvals)))
            w[r, c] = val if int(rng.integers(0, 2)) else -val
        for _ in range(max(2, n // 2)):
            r = int(rng.integers(0, b.shape[0]))
            b[r] = np.float32(2.0) if int(rng.integers(0, 2)) else np.float32(-2.0)

    for net in CHAINS:
        if net + ".0.weight" not in sd:
            continue
        rows(net + ".0", *v["k1"])
        rows(net + ".2", *v["k2"])
        outliers(net + ".0", v["out_max"], v["n_out"])
        outliers(net + ".2", v["out_max"], v["n_out"])
        sd[net + ".4.weight"] *= _pow2(-v["m3"])
        outliers(net + ".4", 4.0 + 0.5, 2)          # (+-4 or 4.5: an outlier of the output layer moves its control by a lot)
        sd[net + ".4.weight"][np.abs(sd[net + ".4.weight"]) >= 4.0] *= np.float32(0.125)
    for net in ENCODERS:
        rows(net + ".0", *v["enc_k"])
        rows(net + ".2", *v["enc_k"])
    return sd


def describe(sd):
    """max |w| per chain matrix (what the packer's domain check sees)."""
    return {k: float(np.abs(v).max()) for k, v in sd.items() if k.split(".")[0] in CHAINS}
