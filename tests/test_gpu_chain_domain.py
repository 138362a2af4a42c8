"""GPU tests (-m gpu) of the default (split-f16) chain arithmetic outside the random-init weight regime and at the edge of
its domain:

  * weights with a trained network's dynamic range (tests/heavy_weights.py; hidden activations O(10-100), |w| up to 24):
    reference fixtures e7_heavy_{a,b} -- every chain variant at 1e-4 (test_gpu_parity runs them too), here the default's
    deviation is compared with the exact-fp32 kernel's, both against the reference;
  * a chain weight outside |w| < 63.9: the packer records max |w|, the sampler falls back to the exact-fp32 kernels and
    says so -- no NaN;
  * a layer input outside |x| < 4094: the status word of the packed buffer is set at the conversion that overflowed,
    check_chain_domain() switches the sampler over and the re-run agrees with the CPU oracle."""
import warnings

import numpy as np
import pytest
import torch

from conftest import HEAVY_CASES, golden_meta, golden_weights, load_golden, region_kwargs, scene_from_golden

pytestmark = pytest.mark.gpu


def _hp():
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    return default_hparams()


def _region(d, meta, sd, dev, chain_waves):
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    sb = SceneBatch({k: torch.from_numpy(v) for k, v in scene_from_golden(d).items()}, meta["S"], _hp(), dev)
    sm = Sampler(PackedWeights(sd, dev), _hp(), chain_waves=chain_waves)
    out = sm.sampling_region(sb, meta["steps"], torch.from_numpy(d["x_T"]).to(dev), torch.from_numpy(d["z"]).to(dev),
                             full_list=True, **region_kwargs(meta))
    return sm, sb, out


@pytest.mark.parametrize("name", HEAVY_CASES)
def test_default_chain_is_as_close_to_the_reference_as_exact_fp32_on_heavy_weights(name):
    """Un-guided part of the rollout (the first 39 of 49 reverse steps: no Adam discontinuity involved): the default
    arithmetic's distance from the reference stays within 3x the exact-fp32 kernel's own (+1e-6), and both inside 1e-4."""
    dev = torch.device("cuda:0")
    d = load_golden(name)
    meta = golden_meta(d)
    sd = golden_weights(d)
    assert max(np.abs(sd[k]).max() for k in sd if k.startswith("policy_net") and k.endswith("weight")) >= 8.0
    n_plain = meta["steps"] - meta["guidance_before"]       # list entries 0 .. n_plain - 1 precede the first guided step
    dev_of = {}
    for cw in (0, 8, 32):
        sm, sb, out = _region(d, meta, sd, dev, cw)
        assert sm.chain_waves == cw and not sm.w.chain_overflowed()
        cl = out["controls_list"].reshape(meta["steps"], sb.N, 20, 2).cpu().numpy()
        dev_of[cw] = float(np.abs(cl[:n_plain] - d["controls_list"][:n_plain]).max())
    print("%s: max |controls - reference| over the un-guided steps: split-f16 %.3g, exact fp32 %.3g, split-bf16 %.3g"
          % (name, dev_of[0], dev_of[8], dev_of[32]))
    assert dev_of[8] <= 1e-4 and dev_of[0] <= 1e-4
    assert dev_of[0] <= 3.0 * dev_of[8] + 1e-6


def test_weight_outside_the_half_domain_falls_back_to_exact_fp32():
    from heavy_weights import heavy_weights
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    dev = torch.device("cuda:0")
    hp = _hp()
    sd = heavy_weights(golden_weights(), "w70")
    pw = PackedWeights(sd, dev)
    assert pw.chain_wmax["policy_net"] == 70.0 and not pw.split_f16_ok
    assert abs(pw.chain_wmax["rect_net"] - np.abs(np.concatenate([sd["rect_net.0.weight"][:, 224:].ravel(),
                                                                   sd["rect_net.2.weight"].ravel(),
                                                                   sd["rect_net.4.weight"].ravel()])).max()) == 0.0
    with pytest.warns(RuntimeWarning, match="exact-fp32"):
        sm = Sampler(pw, hp)                       # default arithmetic requested
    assert sm.chain_waves == 8 and "domain" in sm.chain_fallback
    bs, S, K, steps = 2, 8, 3, 10
    scene = make_scene_batch(bs, K=K, S=S, seed=77, stlp_mode="wide")
    g = torch.Generator().manual_seed(3)
    N = bs * S * 3
    x_T = torch.randn(N, 40, generator=g)
    z = torch.randn(steps - 1, N, 40, generator=g)
    ref = orc.sampling_region(sd, {k: v.numpy() for k, v in scene.items()}, S, steps, hp, x_T, z, rect_head=True, multi_cands=3)
    out = sm.sampling_region(SceneBatch(scene, S, hp, dev), steps, x_T.to(dev), z.to(dev), rect_head=True, multi_cands=3,
                             full_list=True)
    cl = out["controls_list"].reshape(steps, N, 20, 2).cpu()
    assert torch.isfinite(cl).all()
    np.testing.assert_allclose(cl.numpy(), ref["controls_list"].numpy(), rtol=0, atol=1e-4)
    # asking for the split-f16 arithmetic through the C ABI regardless sets the status word (the kernel reads the recorded
    # maximum itself); the numbers are undefined then
    sm_forced = Sampler.__new__(Sampler)
    sm_forced.__dict__.update(sm.__dict__)
    sm_forced.chain_waves = 0
    out2 = sm_forced.sampling_region(SceneBatch(scene, S, hp, dev), steps, x_T.to(dev), z.to(dev), rect_head=True, multi_cands=3)
    torch.cuda.synchronize()
    assert out2["final_controls"].shape == (N, 40)
    assert pw.chain_overflowed(clear=True) and not pw.chain_overflowed()


def test_activation_outside_the_half_domain_is_flagged_and_recovered():
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    dev = torch.device("cuda:0")
    hp = _hp()
    sd = {k: v.copy() for k, v in golden_weights().items()}
    # every weight stays inside |w| < 63.9 (1024 / sqrt(303) = 58.8), but layer 1's outputs reach ~1e4: beyond 4094 =
    # 65504 / 2^4.  (ReLU is positively homogeneous: in exact arithmetic the network computes what it did before.)
    sd["policy_net.0.weight"] *= np.float32(1024.0)
    sd["policy_net.0.bias"] *= np.float32(1024.0)
    sd["policy_net.2.weight"] *= np.float32(1.0 / 1024.0)
    pw = PackedWeights(sd, dev)
    assert pw.split_f16_ok and 20.0 < pw.chain_wmax["policy_net"] < 63.9
    sm = Sampler(pw, hp)
    assert sm.chain_waves == 0
    bs, S, K, steps = 2, 8, 2, 8
    scene = make_scene_batch(bs, K=K, S=S, seed=78, stlp_mode="wide")
    g = torch.Generator().manual_seed(4)
    N = bs * S * 3
    x_T = torch.randn(N, 40, generator=g) * 4.0
    z = torch.randn(steps - 1, N, 40, generator=g)
    sb = SceneBatch(scene, S, hp, dev)
    out = sm.sampling_region(sb, steps, x_T.to(dev), z.to(dev), rect_head=True, multi_cands=3)
    torch.cuda.synchronize()
    # (the numbers are undefined now: a NaN does not survive a ReLU -- max(NaN, 0) = 0 -- so the flag is raised where the
    # conversion to half overflows, not where a NaN happens to arrive)
    with pytest.warns(RuntimeWarning, match="exact-fp32"):
        assert sm.check_chain_domain() is True
    assert sm.chain_waves == 8 and not pw.chain_overflowed()
    out = sm.sampling_region(sb, steps, x_T.to(dev), z.to(dev), rect_head=True, multi_cands=3, full_list=True)
    assert sm.check_chain_domain() is False
    ref = orc.sampling_region(sd, {k: v.numpy() for k, v in scene.items()}, S, steps, hp, x_T, z, rect_head=True, multi_cands=3)
    np.testing.assert_allclose(out["controls_list"].reshape(steps, N, 20, 2).cpu().numpy(), ref["controls_list"].numpy(),
                               rtol=0, atol=2e-4)
    with pytest.raises(FloatingPointError):
        sm2 = Sampler(pw, hp, chain_waves=0)
        sm2.sampling_region(sb, steps, x_T.to(dev), z.to(dev), rect_head=True, multi_cands=3)
        sm2.check_chain_domain(fallback=False)


def test_unknown_chain_waves_is_a_shape_error():
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    dev = torch.device("cuda:0")
    hp = _hp()
    sm = Sampler(PackedWeights(golden_weights(), dev), hp, chain_waves=116)
    scene = make_scene_batch(1, K=2, S=8, seed=1, stlp_mode="wide")
    with pytest.raises(RuntimeError, match="shape"):
        sm.sampling_region(SceneBatch(scene, 8, hp, dev), 6, torch.zeros(24, 40, device=dev),
                           torch.zeros(5, 24, 40, device=dev))


@pytest.mark.parametrize("scenes", [1, 48])
def test_empty_pipeline_slots_stay_out_of_the_domain_guard(scenes):
    """Round 3 regression: in the latency layout of k_chain (small batches) the layer 1 woven into an EMPTY pipeline slot read
    a piece buffer nobody had written for that slot.  Its result was never used -- but it passed through the domain guard, and
    on a fresh box (LDS full of a previous tenant's data) one bench process in twenty ended with the overflow flag set.
    tests/ldspoison fills every CU's LDS with large finite garbage (the fixture in conftest.py does it before every -m gpu
    test; here once more right before the launch): the flag must stay clear, one tile per workgroup (192 rows) and three."""
    import conftest
    from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    w = PackedWeights(golden_weights(), dev)
    sm = Sampler(w, hp)
    scene = make_scene_batch(scenes, K=2, S=64, seed=3, stlp_mode="wide")
    sb = SceneBatch(scene, 64, hp, dev)
    _, base_p, _ = sm.encode(sb, need_rect=False)
    x = torch.randn(sb.N, 40, device=dev)
    torch.cuda.synchronize()
    if conftest._LDS_POISON and conftest._LDS_POISON[0] is not None:
        assert conftest._LDS_POISON[0].lds_poison(conftest.ctypes_stream()) == 0
    sm.rollout(sb, base_p, x, None, 20, n_emit=0, seed=9)
    torch.cuda.synchronize()
    assert torch.isfinite(x).all()
    assert not w.chain_overflowed(clear=True)
