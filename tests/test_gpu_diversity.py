"""GPU parity of the post-sampling metrics kernel (pstl_diversity) against
  (1) golden vectors produced by the reference's own measure_diversity / measure_extra_diversity / compute_ade_fde, and
  (2) the CPU oracle (oracle/diversity_oracle.py) on fresh seeded inputs, per (scene, mode).
Tolerances (float metrics, the reference computes them in float32 / Qhull float64 on the CPU): std 5e-5 rel, hull
volume 1e-5 rel, entropies 1e-5 abs (bin counts are integers: exact), ADE/FDE 1e-5 rel, occupancy area 2e-3 rel
(a point within one ulp of a float32 bin edge may fall on the other side: cosf/sinf of the device differ from the host's
in the last bit, and torch.linspace's edge values depend on the host's vector width)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    from pstl_diffusion_policy_amd import ffi
    ffi.lib()
    return torch.device("cuda:0")


def _hp():
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    return default_hparams()


def _run(dev, ego_traj, lane_valid, controls, scores, S):
    """ego_traj (bs,nt,6), lane_valid (bs,3), controls (N,nt,2), scores (N,) -> (per_mode, per_scene, totals) on host."""
    from pstl_diffusion_policy_amd.engine import Sampler, SceneBatch
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    bs = ego_traj.shape[0]
    scene = make_scene_batch(bs, K=2, S=S, seed=1)
    scene["ego_traj"] = ego_traj
    for i, k in enumerate(("curr_id", "left_id", "right_id")):
        scene[k] = lane_valid[:, i:i + 1].clone()
    sb = SceneBatch(scene, S, _hp(), dev)
    sm = Sampler.__new__(Sampler)        # the metric needs no network weights
    from pstl_diffusion_policy_amd import ffi
    sm.L = ffi.lib()
    pm, ps, tot = sm.diversity(sb, controls.reshape(-1, 40).to(dev).contiguous(), scores.to(dev).contiguous())
    torch.cuda.synchronize()
    return pm.cpu().numpy(), ps.cpu().numpy(), tot.cpu().numpy()


@pytest.mark.parametrize("name", ["div_mixed", "div_s64_clip", "div_sparse"])
def test_diversity_matches_reference_golden(dev, name):
    from pstl_diffusion_policy_amd.engine import diversity_from_totals
    g = dict(np.load(os.path.join(GOLD, name + ".npz")))
    S = int(g["S"])
    pm, ps, tot = _run(dev, torch.from_numpy(g["in_ego_traj"]), torch.from_numpy(g["in_valids"]),
                       torch.from_numpy(g["in_controls"]), torch.from_numpy(g["in_scores"]), S)
    d = diversity_from_totals(tot)
    assert d["std"] == pytest.approx(float(g["std"]), rel=5e-5)
    assert d["vol"] == pytest.approx(float(g["vol"]), rel=1e-5)
    assert d["ade"] == pytest.approx(float(g["ade"]), rel=1e-5)
    assert d["fde"] == pytest.approx(float(g["fde"]), rel=1e-5)
    for k in ("ent_s", "ent_w", "ent_a", "ent_wa"):
        assert d[k] == pytest.approx(float(g[k]), abs=1e-5), k
    assert d["area"] == pytest.approx(float(g["area"]), rel=2e-3)
    val = g["in_valids"] > 0
    for m in range(3):
        np.testing.assert_allclose(np.where(val[:, m], pm[:, m, 0], 0), g["std%d" % m], rtol=5e-5, atol=1e-6)
        np.testing.assert_allclose(np.where(val[:, m], pm[:, m, 1], 0), g["vol%d" % m], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("S,bs,seed", [(64, 24, 5), (16, 9, 6), (5, 4, 7)])
def test_diversity_matches_oracle_per_mode(dev, S, bs, seed):
    from oracle import diversity_oracle as dorc
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    scene = make_scene_batch(bs, K=2, S=S, seed=seed, invalid_lane_frac=0.3)
    g = torch.Generator().manual_seed(seed)
    N = bs * S * 3
    ctrl = (torch.randn(N, 20, 2, generator=g) * 0.2 + torch.randn(N, 1, 2, generator=g) * 0.2).clamp(-1, 1)
    ctrl = ctrl * torch.tensor([hp["mul_w_max"], hp["mul_a_max"]])
    scores = torch.randn(N, generator=g) * 0.3 + 0.05
    lane_valid = torch.cat([scene["curr_id"], scene["left_id"], scene["right_id"]], dim=-1)
    valid = lane_valid[:, None, :].repeat(1, S, 1).reshape(N)
    ego = scene["ego_traj"]
    o = dorc.all_metrics(ego[:, 0, :4], ego, ctrl, scores, valid, S, hp, orc.unicycle_rollout)
    pm, ps, tot = _run(dev, ego, lane_valid, ctrl, scores, S)
    np.testing.assert_allclose(pm[:, :, 0], o["std_sm"], rtol=5e-5, atol=1e-6)
    np.testing.assert_allclose(pm[:, :, 1], o["vol_sm"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(pm[:, :, 2], o["ent_s_sm"], atol=2e-6)
    np.testing.assert_allclose(pm[:, :, 3], o["ent_w_sm"].sum(-1), atol=2e-5)
    np.testing.assert_allclose(pm[:, :, 4], o["ent_a_sm"].sum(-1), atol=2e-5)
    np.testing.assert_allclose(pm[:, :, 5], o["area_sm"], rtol=2e-2, atol=1e-6)      # one bin of ~10^2 may flip
    assert np.mean(np.abs(pm[:, :, 5] - o["area_sm"]) <= 1e-5 * np.abs(o["area_sm"]) + 1e-9) >= 0.9
    np.testing.assert_allclose(ps[:, 0], o["ade_s"], rtol=1e-5)
    np.testing.assert_allclose(ps[:, 1], o["fde_s"], rtol=1e-5)
    np.testing.assert_array_equal(pm[:, :, 7], lane_valid.numpy())


def test_diversity_is_reproducible_and_shard_additive(dev):
    """Totals are additive over scene shards (what the multi-GPU all-gather relies on) and bitwise run-to-run stable."""
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    bs, S = 32, 64
    scene = make_scene_batch(bs, K=2, S=S, seed=11, invalid_lane_frac=0.2)
    g = torch.Generator().manual_seed(3)
    N = bs * S * 3
    ctrl = (torch.randn(N, 20, 2, generator=g) * 0.3).clamp(-1, 1) * torch.tensor([hp["mul_w_max"], hp["mul_a_max"]])
    scores = torch.randn(N, generator=g)
    lv = torch.cat([scene["curr_id"], scene["left_id"], scene["right_id"]], dim=-1)
    a = _run(dev, scene["ego_traj"], lv, ctrl, scores, S)
    b = _run(dev, scene["ego_traj"], lv, ctrl, scores, S)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    h = bs // 2
    rows = h * S * 3
    t0 = _run(dev, scene["ego_traj"][:h], lv[:h], ctrl[:rows], scores[:rows], S)
    t1 = _run(dev, scene["ego_traj"][h:], lv[h:], ctrl[rows:], scores[rows:], S)
    np.testing.assert_array_equal(np.concatenate([t0[0], t1[0]]), a[0])
    np.testing.assert_allclose(t0[2] + t1[2], a[2], rtol=1e-12)


@pytest.mark.parametrize("S", [64, 24])
def test_ten_wave_layout_equals_single_wave_layout(dev, S):
    """Few (scene, mode) pairs run ten wavefronts each (k_diversity<true>: the twenty hulls two per wave), many run one: a batch of
    200 scenes (600 pairs: one wave each) against its first 30 scenes evaluated alone (90 pairs: ten waves) -- the eight numbers
    of every (scene, mode) and ADE / FDE of every scene bit for bit."""
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    bs = 200
    scene = make_scene_batch(bs, K=2, S=S, seed=17, invalid_lane_frac=0.2)
    g = torch.Generator().manual_seed(9)
    N = bs * S * 3
    ctrl = (torch.randn(N, 20, 2, generator=g) * 0.25 + torch.randn(N, 1, 2, generator=g) * 0.15).clamp(-1, 1)
    ctrl = ctrl * torch.tensor([hp["mul_w_max"], hp["mul_a_max"]])
    scores = torch.randn(N, generator=g) * 0.3 + 0.1
    lv = torch.cat([scene["curr_id"], scene["left_id"], scene["right_id"]], dim=-1)
    full = _run(dev, scene["ego_traj"], lv, ctrl, scores, S)
    h = 30
    rows = h * S * 3
    part = _run(dev, scene["ego_traj"][:h], lv[:h], ctrl[:rows], scores[:rows], S)
    np.testing.assert_array_equal(part[0], full[0][:h])
    np.testing.assert_array_equal(part[1], full[1][:h])
    assert np.isfinite(part[0]).all() and (part[0][:, :, 1] > 0).any()
