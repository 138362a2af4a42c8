"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
  (1) the golden vectors produced by the reference itself (tests/golden/*.npz), and
  (2) the CPU oracle on fresh seeded inputs.
Tolerances: sampled trajectories 1e-4 abs (north star); STL scores 5e-4 abs / 5e-5 rel; satisfaction masks exact
outside a reported |score| < 1e-4 band (expected empty on these fixtures); indices and counts exact."""
import numpy as np
import pytest
import torch

from conftest import (HEAVY_CASES, SAMPLING_CASES, STL_CASES, golden_meta, hparams_for, golden_weights, guidance_cfg, guided_outlier_rows,
                      load_golden, region_kwargs, scene_from_golden)
from pstl_diffusion_policy_amd.engine import guidance_triggered

pytestmark = pytest.mark.gpu

TRAJ_TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    from pstl_diffusion_policy_amd import ffi
    ffi.lib()   # fails loudly when the HIP library has not been built
    return torch.device("cuda:0")


def _hp():
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    return default_hparams()


def _weights(dev, zero_out=False, d=None):
    from pstl_diffusion_policy_amd.engine import PackedWeights
    sd = {k: v.copy() for k, v in golden_weights(d).items()}
    if zero_out:
        sd["policy_net.4.weight"] *= 0
        sd["policy_net.4.bias"] *= 0
    return PackedWeights(sd, dev), sd


def _scene_batch(d, S, dev):
    from pstl_diffusion_policy_amd.engine import SceneBatch
    return SceneBatch({k: torch.from_numpy(v) for k, v in scene_from_golden(d).items()}, S, hparams_for(d), dev)


def _mask_equal_outside_band(mine, ref, band=1e-4):
    """Satisfaction masks must be identical wherever |score_ref| >= band; returns the number of rows INSIDE the band
    whose mask differs (0 on every fixture here; at full scale a handful of rows can sit within float noise of 0)."""
    mine, ref = np.asarray(mine), np.asarray(ref)
    inband = np.abs(ref) < band
    np.testing.assert_array_equal((mine > 0)[~inband], (ref > 0)[~inband])
    return int(((mine > 0) != (ref > 0))[inband].sum())


@pytest.mark.parametrize("name", STL_CASES)
def test_stl_forward_matches_reference(dev, name):
    from pstl_diffusion_policy_amd.engine import Sampler
    d = load_golden(name)
    bs, S, K, seed = [int(v) for v in d["meta"]]
    w, _ = _weights(dev)
    sb = _scene_batch(d, S, dev)
    sm = Sampler(w, _hp())
    c = torch.from_numpy(d["controls"]).reshape(1, sb.N, 40).to(dev)
    r = sm.score(sb, c, all3=True)
    np.testing.assert_allclose(r["scores3"][:, 0].cpu().numpy(), d["scores3"], rtol=5e-5, atol=5e-4)
    np.testing.assert_allclose(r["scores"][0].cpu().numpy(), d["scores"], rtol=5e-5, atol=5e-4)
    assert _mask_equal_outside_band(r["scores"][0].cpu().numpy(), d["scores"]) == 0
    np.testing.assert_array_equal(r["scores"][0].cpu().numpy() > 0, d["scores"] > 0)
    # selected-formula kernel == all-three kernel, bit for bit
    r1 = sm.score(sb, c, all3=False)
    assert torch.equal(r1["scores"], r["scores"])
    # scoring given trajectories (what compute_stl_dense receives) == scoring controls
    tr = sm.trajs(sb, c[0])
    np.testing.assert_allclose(tr.cpu().numpy(), d["trajs"], rtol=0, atol=2e-4)
    r2 = sm.score(sb, None, states=tr[:, :-1].contiguous().reshape(1, sb.N, 20, 4))
    assert torch.equal(r2["scores"], r["scores"])
    counts, mask = sm.metrics(sb, r["scores"][0], want_mask=True)
    from pstl_diffusion_policy_amd.engine import acc_from_counts
    acc, sacc = acc_from_counts(counts)
    assert acc == float(d["acc"]) and sacc == float(d["scene_acc"])
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), d["scores"] > 0)


@pytest.mark.parametrize("name", STL_CASES)
def test_stl_backward_matches_reference_autograd(dev, name):
    from pstl_diffusion_policy_amd.engine import Sampler
    d = load_golden(name)
    bs, S, K, seed = [int(v) for v in d["meta"]]
    w, _ = _weights(dev)
    sb = _scene_batch(d, S, dev)
    sm = Sampler(w, _hp())
    c = torch.from_numpy(d["controls"]).reshape(sb.N, 40).to(dev)
    sc, g = sm.score_grad(sb, c)
    np.testing.assert_allclose(sc.cpu().numpy(), d["scores"], rtol=5e-5, atol=5e-4)
    ref = d["grad_sum"].reshape(-1, 40)
    scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-20
    np.testing.assert_allclose(g.cpu().numpy() / scale, ref / scale, rtol=5e-3, atol=5e-4)


@pytest.mark.parametrize("name", ["e7_steps12", "e7_steps50_k8"])
def test_scene_encoder_matches_reference(dev, name):
    from pstl_diffusion_policy_amd.engine import Sampler
    d = load_golden(name)
    meta = golden_meta(d)
    w, sd = _weights(dev)
    sb = _scene_batch(d, meta["S"], dev)
    feature, base_p, base_r = Sampler(w, _hp()).encode(sb)
    np.testing.assert_allclose(feature.cpu().numpy(), d["feature_scene"], rtol=0, atol=2e-5)
    W1, b1 = sd["policy_net.0.weight"], sd["policy_net.0.bias"]
    np.testing.assert_allclose(base_p.cpu().numpy(), d["feature_scene"] @ W1[:, :224].T + b1, rtol=0, atol=5e-5)
    W1, b1 = sd["rect_net.0.weight"], sd["rect_net.0.bias"]
    np.testing.assert_allclose(base_r.cpu().numpy(), d["feature_scene"] @ W1[:, :224].T + b1, rtol=0, atol=5e-5)


def _run_region(dev, name, chain_waves=0):
    from pstl_diffusion_policy_amd.engine import Sampler
    d = load_golden(name)
    meta = golden_meta(d)
    w, _ = _weights(dev, zero_out=bool(meta["zero_net_out"]), d=d)
    sb = _scene_batch(d, meta["S"], dev)
    sm = Sampler(w, hparams_for(d), chain_waves=chain_waves)
    assert sm.chain_waves == chain_waves and sm.chain_fallback is None      # every fixture's weights are inside the domain
    out = sm.sampling_region(sb, meta["steps"], torch.from_numpy(d["x_T"]).to(dev), torch.from_numpy(d["z"]).to(dev),
                             full_list=True, **region_kwargs(meta))
    return d, meta, sb, out


@pytest.mark.parametrize("chain_waves", [8, 4, 0, 16, 32, 2])
@pytest.mark.parametrize("name", SAMPLING_CASES)
def test_sampling_region_matches_reference(dev, name, chain_waves):
    from pstl_diffusion_policy_amd.engine import acc_from_counts
    if chain_waves == 32 and name in HEAVY_CASES + ["e7_trained_guid"]:
        # bfloat16 pieces carry an operand to 2^-17: 8e-6 from the reference on random-init weights, but with hidden
        # activations in the hundreds (e7_heavy_b) rows leave 1e-4 within a few un-guided steps -- the round-1 default is not a
        # 1e-4 arithmetic at a trained network's dynamic range.  test_gpu_chain_domain.py records its deviation next to the
        # default's (half pieces, 2^-23) and the exact-fp32 kernel's.
        pytest.skip("split-bf16 (chain_waves 32) is kept for comparison only; not held to 1e-4 on heavy-tailed weights")
    d, meta, sb, out = _run_region(dev, name, chain_waves)
    N = sb.N
    # 1e-4 on every element of every row without guidance.  With guidance: rows that hold an element in Adam's eps regime
    # are excluded (<= 0.1 % of the rows, the regime verified against the reference's recorded gradients -- see
    # conftest.guided_outlier_rows), every other row stays at 1e-4 through selection, RefineNet and the re-rolls.
    cl = out["controls_list"].reshape(meta["steps"], N, 20, 2).cpu().numpy()
    err_all = np.abs(cl - d["controls_list"])
    keep = keep_g = np.ones(N, dtype=bool)
    if meta["guidance"]:
        bad_rows, bad_groups = guided_outlier_rows(err_all, d, meta, TRAJ_TOL)
        keep, keep_g = ~bad_rows, ~bad_groups
        hp = _hp()
        n_guided = sum(1 for i in range(1, meta["steps"]) if guidance_triggered(i, meta["steps"], guidance_cfg(meta)))
        cap = 2.0 * meta["guidance_lr"] * meta["guidance_niters"] * n_guided * np.array([hp["mul_w_max"], hp["mul_a_max"]])
        assert (err_all <= cap + TRAJ_TOL).all(), "a control moved further than its Adam steps allow"
    err = err_all[:, keep].reshape(meta["steps"], -1).max(axis=1)
    assert err.max() <= TRAJ_TOL, "per-step max |delta| of the sampled controls: %s" % err
    merged = bool(meta["diverse"]) and bool(meta["rect_head"])      # merge_net pools over (scene, mode, shard) groups
    keep_r = keep_g if merged else keep

    def close(mine, ref, rows, msg):
        np.testing.assert_allclose(mine.reshape(N, 20, 2).cpu().numpy()[rows], ref[rows], rtol=0, atol=TRAJ_TOL, err_msg=msg)

    has_rect = "rect_controls" in d
    close(out["final_controls"], d["final_controls"], keep_r if has_rect else keep, "final_controls")
    if "sel_idx" in d:
        # (2e-3, not 1e-3: a score is ~20 unicycle steps downstream of the controls; |d score / d control| reaches ~10-20, so
        # controls that agree to 1e-4 give scores that agree to ~2e-3 -- observed: one element of 1920 at 1.04e-3 on e7_guid_c4
        # after ten guided steps, with every control inside 1e-4)
        np.testing.assert_allclose(out["cand_scores"].cpu().numpy()[:, keep], d["cand_scores"][:, keep], rtol=5e-5, atol=2e-3)
        # candidate choice: exact unless two candidates score within the arithmetic noise of each other
        top2 = np.sort(d["cand_scores"], axis=0)[-2:]
        clear = ((top2[1] - top2[0]) > 1e-3) & keep
        np.testing.assert_array_equal(out["sel_idx"].cpu().numpy()[clear], d["sel_idx"][clear])
        close(out["sel_controls"], d["sel_controls"], clear, "sel_controls")
        keep_r = keep_r & (clear | ~keep)      # an ambiguous choice is not RefineNet's error either
        if merged and not clear[keep].all():
            amb = ~clear & keep
            _, amb_g = _groups(amb, meta["S"])
            keep_r = keep_r & ~amb_g
    for k in ["rect_controls", "roll0_controls", "roll1_controls", "roll2_controls"]:
        if k in d:
            close(out[k], d[k], keep_r, k)
    fs, fr = out["final_scores"].cpu().numpy(), d["final_scores"]
    np.testing.assert_allclose(fs[keep_r], fr[keep_r], rtol=1e-4, atol=2e-3)
    assert _mask_equal_outside_band(fs[keep_r], fr[keep_r], band=1e-3) == 0
    acc, sacc = acc_from_counts(out["counts"])
    assert abs(acc - float(d["final_acc"])) <= 0.005 and abs(sacc - float(d["final_scene_acc"])) <= 0.005
    if keep_r.all():
        assert acc == float(d["final_acc"]) and sacc == float(d["final_scene_acc"])
    else:
        # rows were excluded: the counters over the KEPT rows must still be exact -- satisfaction masks equal row by row,
        # and the any-sample-satisfies reduction per (scene, mode) over the kept rows equal too
        np.testing.assert_array_equal((fs > 0)[keep_r], (fr > 0)[keep_r])
        S = meta["S"]
        mine_c = ((fs > 0) & keep_r).reshape(-1, S, 3).any(axis=1)
        ref_c = ((fr > 0) & keep_r).reshape(-1, S, 3).any(axis=1)
        np.testing.assert_array_equal(mine_c, ref_c)


def _groups(rows, S, n_shards=4):
    """rows (N,) bool -> (rows, rows widened to their (scene, mode, shard) max-pool groups)."""
    N = rows.shape[0]
    if S % n_shards:
        return rows, rows
    sps = S // n_shards
    grp = rows.reshape(-1, n_shards, sps, 3).any(axis=2, keepdims=True)
    return rows, np.broadcast_to(grp, (N // (S * 3), n_shards, sps, 3)).reshape(N).copy()


def test_run_to_run_determinism(dev):
    _, _, _, a = _run_region(dev, "e7_wide")
    _, _, _, b = _run_region(dev, "e7_wide")
    for k in ["controls_list", "final_controls", "final_scores", "counts", "sel_idx", "rect_controls"]:
        assert torch.equal(a[k], b[k]), k


def test_against_oracle_on_fresh_scenes(dev):
    """No fixture involved: seeded scenes, weights and noise; oracle on the CPU vs HIP, e7 + guidance."""
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler, acc_from_counts
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    bs, S, K, steps = 6, 16, 5, 14
    scene = make_scene_batch(bs, K=K, S=S, seed=4242, invalid_lane_frac=0.3, stlp_mode="wide")
    sd = golden_weights()
    g = torch.Generator().manual_seed(7)
    N = bs * S * 3
    x_T = torch.randn(N, 40, generator=g)
    z = torch.randn(steps - 1, N, 40, generator=g)
    guid = dict(enabled=True, before=3, niters=2, lr=0.01)
    ref = orc.sampling_region(sd, {k: v.numpy() for k, v in scene.items()}, S, steps, hp, x_T, z, rect_head=True,
                              multi_cands=5, guidance=guid, n_rolls=1)
    sm = Sampler(PackedWeights(sd, dev), hp)
    sb = SceneBatch(scene, S, hp, dev)
    out = sm.sampling_region(sb, steps, x_T.to(dev), z.to(dev), rect_head=True, multi_cands=5, guidance=guid, n_rolls=1,
                             full_list=True)
    cl = out["controls_list"].reshape(steps, N, 20, 2).cpu()
    assert (cl - ref["controls_list"]).abs().max().item() <= TRAJ_TOL
    assert (out["final_controls"].reshape(N, 20, 2).cpu() - ref["final_controls"]).abs().max().item() <= TRAJ_TOL
    np.testing.assert_allclose(out["final_scores"].cpu().numpy(), ref["final_scores"].numpy(), rtol=1e-4, atol=2e-3)
    nband = _mask_equal_outside_band(out["final_scores"].cpu().numpy(), ref["final_scores"].numpy(), band=1e-3)
    acc, sacc = acc_from_counts(out["counts"])
    assert abs(acc - float(ref["final_acc"])) <= 0.005, (acc, float(ref["final_acc"]), nband)
    assert abs(sacc - float(ref["final_scene_acc"])) <= 0.005


def test_shard_invariance_and_outlier_rows(dev):
    """Size-independent properties at a larger size: (a) evaluating a contiguous block of scenes alone gives exactly the
    rows of the full evaluation (what multi-GPU sharding relies on); (b) duplicated scenes with duplicated noise give
    duplicated results; (c) rows whose high-level index is 3 score the constant 1."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    bs, S, K, steps = 96, 64, 4, 6
    scene = make_scene_batch(bs, K=K, S=S, seed=5, invalid_lane_frac=0.2, stlp_mode="wide")
    for k in scene:                      # (b): second half duplicates the first half
        scene[k][bs // 2:] = scene[k][:bs // 2]
    sm = Sampler(PackedWeights(golden_weights(), dev), hp)
    N = bs * S * 3
    g = torch.Generator().manual_seed(11)
    x_T = torch.randn(N // 2, 40, generator=g).repeat(2, 1).to(dev)
    z = torch.randn(steps - 1, N // 2, 40, generator=g).repeat(1, 2, 1).to(dev)
    full = sm.sampling_region(SceneBatch(scene, S, hp, dev), steps, x_T, z, rect_head=True, multi_cands=3)
    h = N // 2
    assert torch.equal(full["final_controls"][:h], full["final_controls"][h:])
    assert torch.equal(full["final_scores"][:h], full["final_scores"][h:])
    lo, hi = 16, 40                      # (a): scenes [16,40) alone
    sub = {k: v[lo:hi].clone() for k, v in scene.items()}
    r0, r1 = lo * S * 3, hi * S * 3
    part = sm.sampling_region(SceneBatch(sub, S, hp, dev), steps, x_T[r0:r1].contiguous(), z[:, r0:r1].contiguous(),
                              rect_head=True, multi_cands=3)
    assert torch.equal(part["final_controls"], full["final_controls"][r0:r1])
    assert torch.equal(part["final_scores"], full["final_scores"][r0:r1])
    sb = SceneBatch(sub, S, hp, dev)     # (c)
    sb.hl = torch.full_like(sb.hl, 3.0)
    sc = sm.score(sb, part["final_controls"].reshape(1, -1, 40), all3=True)["scores"]
    assert torch.equal(sc, torch.ones_like(sc))


def test_in_kernel_noise(dev):
    """PSTL_FLAG_RNG: (1) the draws are standard normal and differ between steps/seeds; (2) a rollout that draws its own
    noise equals, bit for bit, the same rollout fed with pstl_fill_normal's tensors (so the parity-mode tests cover the
    arithmetic of the production mode too), guided steps included; (3) a block of scenes evaluated alone with the right
    row_offset reproduces its rows of the full batch."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    bs, S, K, steps, seed = 48, 64, 3, 9, 20240229
    scene = make_scene_batch(bs, K=K, S=S, seed=8, invalid_lane_frac=0.2, stlp_mode="wide")
    sm = Sampler(PackedWeights(golden_weights(), dev), hp)
    sb = SceneBatch(scene, S, hp, dev)
    z = torch.stack([sm.fill_normal(sb, steps, i, seed) for i in range(steps - 1, 0, -1)])   # z[k] <-> step steps-1-k
    x_T = sm.fill_normal(sb, steps, steps, seed)
    allz = torch.cat([z.reshape(-1), x_T.reshape(-1)]).double()
    n = allz.numel()
    assert abs(allz.mean().item()) < 5 / n ** 0.5 and abs(allz.var().item() - 1) < 10 / n ** 0.5
    assert abs((allz ** 3).mean().item()) < 10 / n ** 0.5 and abs((allz ** 4).mean().item() - 3) < 30 / n ** 0.5
    assert not torch.equal(z[0], z[1]) and not torch.equal(z[0], sm.fill_normal(sb, steps, steps - 1, seed + 1))
    assert torch.equal(z[0], sm.fill_normal(sb, steps, steps - 1, seed))
    guid = dict(enabled=True, before=3, niters=2, lr=0.01)
    a = sm.sampling_region(sb, steps, None, None, rect_head=True, multi_cands=4, guidance=guid, seed=seed, full_list=True)
    b = sm.sampling_region(sb, steps, x_T, z, rect_head=True, multi_cands=4, guidance=guid, full_list=True)
    for k in ["controls_list", "final_controls", "final_scores", "counts"]:
        assert torch.equal(a[k], b[k]), k
    lo, hi = 10, 31
    sub = {k: v[lo:hi].clone() for k, v in scene.items()}
    part = sm.sampling_region(SceneBatch(sub, S, hp, dev, row_offset=lo * S * 3, global_valid_sum=float(sb.valid.sum()),
                                         global_rows=sb.N), steps, None, None, rect_head=True, multi_cands=4,
                              guidance=guid, seed=seed)
    r0, r1 = lo * S * 3, hi * S * 3
    assert torch.equal(part["final_controls"], a["final_controls"][r0:r1])
    assert torch.equal(part["final_scores"], a["final_scores"][r0:r1])


def test_full_size_properties(dev):
    """BASELINE.json's single-GPU size (4096 scenes x 64 samples x 3 modes = 786432 rows, 50 diffusion steps, e7 +
    guidance), checked through size-independent properties: the two halves of the batch evaluated on their own (as two
    GPUs would) reproduce the full run bit for bit and their counters add up; every control respects the clip range;
    the counters agree with the scores; rows of invalid lanes never count."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler, acc_from_counts
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    bs, S, K, steps, seed = 4096, 64, 2, 50, 7
    scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=K, S=S, seed=3, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    sm = Sampler(PackedWeights(golden_weights(), dev), hp)
    guid = dict(enabled=True, before=10, niters=1, lr=0.01)
    sb = SceneBatch(scene, S, hp, dev)
    full = sm.sampling_region(sb, steps, None, None, rect_head=True, multi_cands=5, guidance=guid, seed=seed,
                              want_scores3=False)
    N = sb.N
    assert N == 786432
    c = full["final_controls"].reshape(N, 20, 2)
    assert torch.isfinite(c).all() and torch.isfinite(full["final_scores"]).all()
    assert c[..., 0].abs().max().item() <= hp["mul_w_max"] and c[..., 1].abs().max().item() <= hp["mul_a_max"]
    cnt = full["counts"].tolist()
    sat = (full["final_scores"] > 0) & (sb.valid > 0)
    assert cnt[0] == int(sat.sum()) and cnt[1] == int(sb.valid.sum()) and cnt[2] == N and cnt[5] == 3 * bs
    scene_sat = sat.reshape(bs, S, 3).any(dim=1)
    assert cnt[3] == int(scene_sat.sum()) and cnt[4] == int(sb.valid.reshape(bs, S, 3)[:, 0].sum())
    total = torch.zeros(8, dtype=torch.int64, device=dev)
    vsum = float(sb.valid.sum())
    for half in range(2):
        lo, hi = half * bs // 2, (half + 1) * bs // 2
        sub = {k: v[lo:hi].contiguous() for k, v in scene.items()}
        part = sm.sampling_region(SceneBatch(sub, S, hp, dev, row_offset=lo * S * 3, global_valid_sum=vsum, global_rows=N),
                                  steps, None, None, rect_head=True, multi_cands=5, guidance=guid, seed=seed,
                                  want_scores3=False)
        r0, r1 = lo * S * 3, hi * S * 3
        assert torch.equal(part["final_controls"], full["final_controls"][r0:r1])
        assert torch.equal(part["final_scores"], full["final_scores"][r0:r1])
        total += part["counts"]
    assert torch.equal(total, full["counts"])
    # a shard small enough for 5-tile workgroups: its single-step (guided) denoiser launches walk their groups one by one,
    # while the full batch streams them through the pipeline without draining (CONT, mlp_kernels.hip) -- same bits
    # (the full batch's multi-step segments run on k_chain2, which a 12 288-row shard would not choose for itself: forced, so
    # that both sum in the same order; the single-step launches under test are k_chain's either way)
    lo, hi = 1000, 1064
    sub = {k: v[lo:hi].contiguous() for k, v in scene.items()}
    part = Sampler(sm.w, hp, chain_waves=2).sampling_region(SceneBatch(sub, S, hp, dev, row_offset=lo * S * 3, global_valid_sum=vsum, global_rows=N),
                              steps, None, None, rect_head=True, multi_cands=5, guidance=guid, seed=seed, want_scores3=False)
    assert torch.equal(part["final_controls"], full["final_controls"][lo * S * 3:hi * S * 3])
    assert torch.equal(part["final_scores"], full["final_scores"][lo * S * 3:hi * S * 3])
    # ... and in the DEFAULT mode under the job's plan (pstl_cfg.plan_rows: the shard runs the kernel the batch ran)
    part = sm.sampling_region(SceneBatch(sub, S, hp, dev, row_offset=lo * S * 3, global_valid_sum=vsum, global_rows=N, plan_rows=N),
                              steps, None, None, rect_head=True, multi_cands=5, guidance=guid, seed=seed, want_scores3=False)
    assert torch.equal(part["final_controls"], full["final_controls"][lo * S * 3:hi * S * 3])
    assert torch.equal(part["final_scores"], full["final_scores"][lo * S * 3:hi * S * 3])
    acc, sacc = acc_from_counts(full["counts"])
    assert 0.0 < acc < 1.0 and 0.0 < sacc <= 1.0


def test_full_size_properties_e5(dev):
    """BASELINE.json config 2 at its full size (e5: 4096 scenes x 64 x 3 = 786 432 rows, 50 diffusion steps, DDPM only -- no
    guidance, no RefineNet, no clip): ONE 49-step launch of the denoiser kernel without clip or candidate emission, which only
    bench.py's `also` block ran at this size.  Properties: the two halves evaluated on their own reproduce the
    full run bit for bit and their counters add up; the counters agree with the scores; the un-clipped controls are the
    normalised final state (normalize_diff without clip, nusc_train.py:647-655); a repeat gives the same bits."""
    from pstl_diffusion_policy_amd import ffi
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler, acc_from_counts
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    bs, S, K, steps, seed = 4096, 64, 2, 50, 19
    scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=K, S=S, seed=4, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    sm = Sampler(PackedWeights(golden_weights(), dev), hp)
    sb = SceneBatch(scene, S, hp, dev)
    N = sb.N
    assert N == 786432
    assert ffi.rollout_layout(sb.cfg(steps, ffi.PSTL_FLAG_RNG, 0))[0] == 2        # the row-stationary kernel, one launch
    sm.trace = []
    full = sm.sampling_region(sb, steps, None, None, rect_head=False, seed=seed, want_scores3=False)
    assert [(n, rows) for (_, _, n, rows) in sm.trace] == [(steps - 1, N)], "one 49-step launch over all rows"
    sm.trace = None
    again = sm.sampling_region(sb, steps, None, None, rect_head=False, seed=seed, want_scores3=False)
    c = full["final_controls"]
    assert torch.isfinite(c).all() and torch.isfinite(full["final_scores"]).all()
    assert torch.equal(c, again["final_controls"]) and torch.equal(full["final_scores"], again["final_scores"])
    # no clip: some controls of a random-init denoiser leave the range RefineNet's configs clip to
    c3 = c.reshape(N, 20, 2)
    assert c3[..., 0].abs().max().item() > hp["mul_w_max"] or c3[..., 1].abs().max().item() > hp["mul_a_max"]
    cnt = full["counts"].tolist()
    sat = (full["final_scores"] > 0) & (sb.valid > 0)
    assert cnt[0] == int(sat.sum()) and cnt[1] == int(sb.valid.sum()) and cnt[2] == N and cnt[5] == 3 * bs
    assert cnt[3] == int(sat.reshape(bs, S, 3).any(dim=1).sum())
    # the scores are the scores of exactly these controls (a second scoring pass over them)
    assert torch.equal(sm.score(sb, c.reshape(1, N, 40))["scores"][0], full["final_scores"])
    total = torch.zeros(8, dtype=torch.int64, device=dev)
    for half in range(2):
        lo, hi = half * bs // 2, (half + 1) * bs // 2
        sub = {k: v[lo:hi].contiguous() for k, v in scene.items()}
        part = sm.sampling_region(SceneBatch(sub, S, hp, dev, row_offset=lo * S * 3, plan_rows=N), steps, None, None,
                                  rect_head=False, seed=seed, want_scores3=False)
        r0, r1 = lo * S * 3, hi * S * 3
        assert torch.equal(part["final_controls"], c[r0:r1])
        assert torch.equal(part["final_scores"], full["final_scores"][r0:r1])
        total += part["counts"]
    assert torch.equal(total, full["counts"])
    acc, _ = acc_from_counts(full["counts"])
    assert 0.0 < acc < 1.0


@pytest.mark.parametrize("noise", ["kernel", "tensor"])
def test_streamed_single_step_launches_equal_the_grouped_ones(dev, noise):
    """Single-step denoiser launches of a large batch stream their 12-tile groups through the pipeline without draining
    (CONT, mlp_kernels.hip); a small batch walks 5-tile groups one by one.  With guidance on every other step all launches
    are single steps -- guided ones (mu only) and plain ones (noise drawn in the kernel or read from the caller's tensor,
    candidates emitted by the chain's epilogue): every shard evaluated alone must reproduce the rows of the whole batch
    bit for bit."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    bs, S, K, steps, seed = 256, 64, 3, 14, 11
    scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=K, S=S, seed=9, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    # chain_waves = 16: k_chain for every launch (with 0 the whole batch's mu-only launches go to k_chain2, whose sums run in
    # another order); the shards are evaluated in the throughput layout (16) and in the latency layout (0: 384 tiles) alike
    sm = Sampler(PackedWeights(golden_weights(), dev), hp, chain_waves=16)
    sm_lat = Sampler(sm.w, hp, chain_waves=0)
    guid = dict(enabled=True, freq=2, niters=1, lr=0.01)
    sb = SceneBatch(scene, S, hp, dev)
    N = sb.N
    g = torch.Generator(device=dev).manual_seed(3)
    x_T = torch.randn(N, 40, device=dev, generator=g) if noise == "tensor" else None
    z = torch.randn(steps - 1, N, 40, device=dev, generator=g) if noise == "tensor" else None
    kw = dict(rect_head=True, multi_cands=5, guidance=guid, want_scores3=False, seed=seed if noise == "kernel" else None)
    full = sm.sampling_region(sb, steps, x_T, z, **kw)
    vsum = float(sb.valid.sum())
    for lo in (0, 96, 224):
        hi = lo + 32
        sub = {k: v[lo:hi].contiguous() for k, v in scene.items()}
        r0, r1 = lo * S * 3, hi * S * 3
        for smp in (sm, sm_lat):
            part = smp.sampling_region(SceneBatch(sub, S, hp, dev, row_offset=r0, global_valid_sum=vsum, global_rows=N), steps,
                                       None if x_T is None else x_T[r0:r1].contiguous(),
                                       None if z is None else z[:, r0:r1].contiguous(), **kw)
            for k in ("final_controls", "final_scores", "sel_controls", "cand_scores"):
                a, b = part[k], (full[k][:, r0:r1] if k == "cand_scores" else full[k][r0:r1])
                assert torch.equal(a, b), (k, smp.chain_waves)


@pytest.mark.parametrize("bs", [112, 267])
def test_balanced_workgroup_sizes_do_not_change_results(dev, bs):
    """Multi-step denoiser launches spread a batch's tiles evenly over whole rounds of workgroups (tiles_per_group_balanced,
    mlp_kernels.hip): 112 scenes = 1 344 tiles run as 224 six-tile workgroups (five-tile groups would need a second round for
    13 of them), 267 scenes = 3 204 tiles as two rounds of seven-tile workgroups instead of 267 twelve-tile ones.  Row-wise
    arithmetic, noise keyed by the global row: a 24-scene shard of the batch, evaluated alone in the latency layout, must
    reproduce its rows bit for bit."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    S, steps = 64, 8
    scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=2, S=S, seed=31, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    # (chain_waves = 16 pins the whole batch to k_chain's throughput layout: with 0 the 267-scene batch's launch goes to k_chain2,
    # whose sums run in another order; the shard runs the default, i.e. k_chain's latency layout)
    sm = Sampler(PackedWeights(golden_weights(), dev), hp, chain_waves=16)
    kw = dict(rect_head=True, multi_cands=3, want_scores3=False, seed=5)
    sb = SceneBatch(scene, S, hp, dev)
    full = sm.sampling_region(sb, steps, None, None, **kw)
    lo, hi = bs - 30, bs - 6
    sub = {k: v[lo:hi].contiguous() for k, v in scene.items()}
    r0, r1 = lo * S * 3, hi * S * 3
    part = Sampler(sm.w, hp).sampling_region(SceneBatch(sub, S, hp, dev, row_offset=r0), steps, None, None, **kw)
    for k in ("final_controls", "final_scores", "sel_controls"):
        assert torch.isfinite(full[k]).all() and torch.equal(part[k], full[k][r0:r1]), k


@pytest.mark.parametrize("bs,S,steps", [(1, 64, 100), (3, 16, 12), (7, 32, 20), (40, 64, 9), (100, 64, 6)])
@pytest.mark.parametrize("noise", ["kernel", "tensor"])
def test_latency_layout_equals_throughput_layout(dev, bs, S, steps, noise):
    """`chain_waves = 0` runs the multi-step denoiser launch of a batch with fewer than five tiles per CU in the latency
    layout (1..5 tiles per workgroup, empty pipeline slots skipped: SPARSE in mlp_kernels.hip); `chain_waves = 16` is the
    same arithmetic in the throughput layout whatever the size (what the bench's batch gets, and what every small fixture
    ran before round 3).  Same arithmetic per row, noise keyed by the global row: every output must agree bit for bit --
    192 rows (the closed loop's batch), 144 and 672 rows, 480 tiles (two per workgroup), 1200 tiles (five: nothing skipped)."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=2, S=S, seed=21 + bs, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    w = PackedWeights(golden_weights(), dev)
    N = bs * S * 3
    g = torch.Generator(device=dev).manual_seed(5)
    x_T = torch.randn(N, 40, device=dev, generator=g) if noise == "tensor" else None
    z = torch.randn(steps - 1, N, 40, device=dev, generator=g) if noise == "tensor" else None
    kw = dict(rect_head=True, multi_cands=5, guidance=dict(enabled=True, before=3, niters=1, lr=0.01), want_scores3=False,
              seed=77 if noise == "kernel" else None, full_list=(noise == "tensor"))
    outs = [Sampler(w, hp, chain_waves=cw).sampling_region(SceneBatch(scene, S, hp, dev), steps, x_T, z, **kw) for cw in (0, 16)]
    keys = ["final_controls", "final_scores", "sel_controls", "cand_scores"] + (["controls_list"] if noise == "tensor" else [])
    for k in keys:
        assert torch.isfinite(outs[0][k]).all(), k
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert torch.equal(outs[0]["counts"], outs[1]["counts"])


@pytest.mark.parametrize("K,niters,maximize,norm", [(2, 1, False, False), (8, 1, True, False), (3, 2, False, False),
                                                    (2, 1, False, True), (15, 1, False, False), (1, 3, True, True)])
def test_stl_latency_layout_equals_one_wave_layout(dev, K, niters, maximize, norm):
    """The STL kernels' latency layout (ten wavefronts per 64 rows: geometry two steps per wave, the forward sweep's chains
    on four waves, the adjoint's direct partials two steps per wave, costate recursion and score on wave 0; stl_core.hpp
    stl_pre_chain / adj_pre_*) against the one-wave kernels that run the fused sweeps: a 200-scene batch is 600 groups of 64
    rows (one wave each), its 20-scene shard 60 groups (the latency layout, in the four-candidate scoring launch too: 240
    workgroups) -- every row of the shard must come out bit for bit, through guided steps with one or several Adam iterations, the maximize loss (every row active), --norm_stl, and
    K beyond the winners' record (15 > kRecMaxK: the adjoint ranks again)."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = dict(_hp())
    if norm:
        hp["norm_stl"] = True
    bs, S, steps = 200, 64, 9
    scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=K, S=S, seed=41 + K, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    sm = Sampler(PackedWeights(golden_weights(), dev), hp, chain_waves=16)   # (the MLP chains in ONE layout for both sizes)
    guid = dict(enabled=True, before=5, niters=niters, lr=0.02, maximize=maximize)
    kw = dict(rect_head=True, multi_cands=4, guidance=guid, want_scores3=False, seed=13)
    N = bs * S * 3
    vsum = float(sum(scene[k].sum().item() for k in ("curr_id", "left_id", "right_id"))) * S
    full = sm.sampling_region(SceneBatch(scene, S, hp, dev), steps, None, None, **kw)
    lo, hi = 130, 150
    sub = {k: v[lo:hi].contiguous() for k, v in scene.items()}
    r0, r1 = lo * S * 3, hi * S * 3
    part = sm.sampling_region(SceneBatch(sub, S, hp, dev, row_offset=r0, global_valid_sum=vsum, global_rows=N), steps, None, None, **kw)
    for k in ("final_controls", "final_scores", "sel_controls", "cand_scores"):
        a, b = part[k], (full[k][:, r0:r1] if k == "cand_scores" else full[k][r0:r1])
        assert torch.isfinite(a).all() and torch.equal(a, b), k


def test_stl_masks_at_scale_match_reference(dev):
    """6144 rows (32 scenes x the reference's sampling_size 64 x 3 modes; LDS-staged scene tables): the three formula
    scores and the satisfaction mask against the reference's own compute_stl_dense, plus the loss gradient."""
    from pstl_diffusion_policy_amd.engine import Sampler, acc_from_counts
    d = load_golden("stl_big")
    bs, S, K, seed = [int(v) for v in d["meta"]]
    w, _ = _weights(dev)
    sb = _scene_batch(d, S, dev)
    sm = Sampler(w, _hp())
    c = torch.from_numpy(d["controls"]).reshape(1, sb.N, 40).to(dev)
    r = sm.score(sb, c, all3=True)
    np.testing.assert_allclose(r["scores3"][:, 0].cpu().numpy(), d["scores3"], rtol=5e-5, atol=5e-4)
    got = r["scores"][0].cpu().numpy()
    np.testing.assert_allclose(got, d["scores"], rtol=5e-5, atol=5e-4)
    np.testing.assert_array_equal(got > 0, d["scores"] > 0)                  # 6144 masks, bit for bit
    print("rows with |score| < 1e-3 in the reference: %d of %d" % (int((np.abs(d["scores"]) < 1e-3).sum()), sb.N))
    counts, _ = sm.metrics(sb, r["scores"][0])
    acc, sacc = acc_from_counts(counts)
    assert acc == float(d["acc"]) and sacc == float(d["scene_acc"])
    # d loss / d controls of the guidance loss (what pstl_guidance_step feeds Adam with)
    dscore = torch.where((_hp()["stl_nn_thres"] - r["scores"][0]) > 0, -sb.grad_scale * sb.valid, torch.zeros_like(sb.valid))
    _, g = sm.score_grad(sb, c[0], dscore=dscore.contiguous())
    ref = d["grad_loss"].reshape(-1, 40)
    scale = np.abs(ref).max() + 1e-20
    np.testing.assert_allclose(g.cpu().numpy() / scale, ref / scale, rtol=5e-3, atol=2e-4)
