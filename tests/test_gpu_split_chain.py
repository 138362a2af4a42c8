"""The arithmetic of the MLP chains is what DESIGN.md section 3.1 says it is.

Default (`chain_waves = 0`): every fp32 product of the 47 per-row input columns and of the two hidden layers is formed
from two IEEE-half pieces per operand (weights as pieces of 2^10 w, activations as pieces of 2^4 x; hi = f16(v),
lo = f16(v - hi); hi*hi + lo*hi + hi*lo, fp32 accumulation), the scene/timestep columns of layer 1 in plain fp32.  The
pieces carry an operand to 2^-23, so the chain must sit as close to the reference as the exact-fp32 MFMA kernel does:
<= 3e-6 after 99 chained steps (the fp32 kernel itself: <= 3e-6, a different summation order of the same products).
`chain_waves = 32` is the round-1 form with bfloat16 pieces (operands to 2^-17), kept for comparison: <= 5e-5.
The oracle, patched to form its products the same way on the CPU, must explain the kernel's deviation."""
import numpy as np
import pytest
import torch

from conftest import golden_meta, golden_weights, load_golden, scene_from_golden
from oracle import pstl_oracle as orc
from pstl_diffusion_policy_amd.synthetic import default_hparams

pytestmark = pytest.mark.gpu

SPLIT = {0: (torch.float16, 16.0, 1024.0), 32: (torch.bfloat16, 1.0, 1.0)}   # piece type, activation / weight factor


def _pieces(v, dt):
    hi = v.to(dt).to(torch.float32)
    lo = (v - hi).to(dt).to(torch.float32)
    return hi, lo


def _mm_split(x, w, cw):   # x (N,K) @ w (O,K)^T as the kernel forms it
    dt, sx, sw = SPLIT[cw]
    xh, xl = _pieces(x * sx, dt)
    wh, wl = _pieces(w * sw, dt)
    return ((xl @ wh.T + xh @ wl.T) + xh @ wh.T) / (sx * sw)


def _split_policy_net(orig, cw):
    ext = list(range(224, 264)) + list(range(296, 303))      # x | hl | stlp: the columns that change per row and step
    hoisted = list(range(0, 224)) + list(range(264, 296))    # scene feature | timestep embedding: fp32 in the kernel too

    def mlp(sd, prefix, x):
        if prefix != "policy_net":
            return orig(sd, prefix, x)
        w1, b1 = orc._t(sd[prefix + ".0.weight"]), orc._t(sd[prefix + ".0.bias"])
        h = torch.relu(x[:, hoisted] @ w1[:, hoisted].T + b1 + _mm_split(x[:, ext], w1[:, ext], cw))
        h = torch.relu(_mm_split(h, orc._t(sd[prefix + ".2.weight"]), cw) + orc._t(sd[prefix + ".2.bias"]))
        return _mm_split(h, orc._t(sd[prefix + ".4.weight"]), cw) + orc._t(sd[prefix + ".4.bias"])
    return mlp


def _kernel_controls(dev, d, meta, scene, hp, cw):
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    sb = SceneBatch({k: torch.from_numpy(v) for k, v in scene.items()}, meta["S"], hp, dev)
    sm = Sampler(PackedWeights(golden_weights(), dev), hp, chain_waves=cw)
    out = sm.sampling_region(sb, meta["steps"], torch.from_numpy(d["x_T"]).to(dev), torch.from_numpy(d["z"]).to(dev),
                             full_list=True)
    return out["controls_list"].reshape(meta["steps"], sb.N, 20, 2).cpu().numpy()[-1]


@pytest.mark.parametrize("name", ["e5_steps10", "e5_steps100"])
def test_default_chain_is_fp32_faithful(name, capsys):
    """the default chain is <= 3e-6 from the reference's own fp32 result after 99 chained steps."""
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    dev = torch.device("cuda:0")
    d = load_golden(name)
    meta, hp, scene = golden_meta(d), default_hparams(), scene_from_golden(d)
    ref = d["controls_list"][-1]                      # the reference's own fp32 result
    e_default = np.abs(_kernel_controls(dev, d, meta, scene, hp, 0) - ref).max()
    e_fp32 = np.abs(_kernel_controls(dev, d, meta, scene, hp, 8) - ref).max()
    with capsys.disabled():
        print("\n%s: |default (split-f16) kernel - ref| %.2e   |fp32 MFMA kernel - ref| %.2e" % (name, e_default, e_fp32))
    assert e_default <= 3e-6, e_default
    assert e_fp32 <= 3e-6, e_fp32


@pytest.mark.parametrize("cw", [0, 32])
@pytest.mark.parametrize("name", ["e5_steps10", "e5_steps100"])
def test_kernel_follows_the_documented_split(name, cw, capsys):
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    dev = torch.device("cuda:0")
    d = load_golden(name)
    meta, hp, scene = golden_meta(d), default_hparams(), scene_from_golden(d)
    orig = orc.relu_mlp
    orc.relu_mlp = _split_policy_net(orig, cw)
    try:
        emu = orc.sampling_region(golden_weights(), scene, meta["S"], meta["steps"], hp, d["x_T"], d["z"])
    finally:
        orc.relu_mlp = orig
    emu = emu["controls_list"].numpy()[-1]
    ref = d["controls_list"][-1]
    got = _kernel_controls(dev, d, meta, scene, hp, cw)
    e_split_vs_emu = np.abs(got - emu).max()
    e_split_vs_ref = np.abs(got - ref).max()
    e_emu_vs_ref = np.abs(emu - ref).max()
    with capsys.disabled():
        print("\n%s chain_waves=%d: |split kernel - ref| %.2e  |emulation - ref| %.2e  |split kernel - emulation| %.2e"
              % (name, cw, e_split_vs_ref, e_emu_vs_ref, e_split_vs_emu))
    # (the CPU emulation sums three sgemm calls whose blocking depends on the host: it carries the fp32 oracle's own
    # summation-order noise, up to ~8e-6 after 99 steps on some hosts, on top of the split's error)
    bound = 3e-6 if cw == 0 else 5e-5
    assert e_split_vs_ref <= bound and e_emu_vs_ref <= max(bound, 1e-5), (e_split_vs_ref, e_emu_vs_ref)
    if cw == 32 and meta["steps"] <= 12:
        # over a few steps the emulation tracks the kernel (what is left is the summation order inside the MFMA); over 99
        # steps the two rounding patterns decorrelate and each sits about as far from the other as from the reference.
        # (With half pieces the split's own error is below that summation-order noise, so there is nothing to track.)
        assert e_split_vs_emu <= max(1e-6, 0.5 * e_emu_vs_ref), (e_split_vs_emu, e_emu_vs_ref, e_split_vs_ref)
