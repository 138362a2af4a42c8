"""--refinement (reference nusc_train.py:1034-1071, inside the timed region): 50 Adam iterations over per-row mixing weights
of eight control sequences, one launch (k_mixopt), against fixtures recorded from the reference's own run_sampling_test.
The gate (conftest.refinement_gate) is on the mechanism -- the gradient of the first iterations -- and on the population
of rows, because single rows are chaotic under 50 steps of lr 0.3 through a score with hard minima in it."""
import numpy as np
import pytest
import torch

from conftest import REFINEMENT_CASES, golden_meta, golden_weights, load_golden, refinement_gate, region_kwargs, scene_from_golden

pytestmark = pytest.mark.gpu


def _setup(d, dev):
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    meta = golden_meta(d)
    hp = default_hparams()
    sb = SceneBatch({k: torch.from_numpy(v) for k, v in scene_from_golden(d).items()}, meta["S"], hp, dev)
    return meta, sb, Sampler(PackedWeights(golden_weights(), dev), hp)


@pytest.mark.parametrize("name", REFINEMENT_CASES)
def test_refinement_block_matches_the_reference_harness(name):
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    dev = torch.device("cuda:0")
    d = load_golden(name)
    meta, sb, sm = _setup(d, dev)
    clist = torch.from_numpy(d["controls_list"]).reshape(meta["steps"], sb.N, 40).to(dev)
    cin = torch.from_numpy(d["refinement_in_controls"]).reshape(sb.N, 40).to(dev)
    out, tr = sm.refinement(sb, cin, clist, iters=meta["refinement"], trace=True)
    ok = refinement_gate(out.reshape(sb.N, 20, 2).cpu().numpy(), tr.cpu().numpy(), d, tol=1e-4, min_frac=0.5)
    # the population: the refined batch reaches the reference's loss and satisfaction count
    v = sb.valid.cpu().numpy() > 0
    want = torch.from_numpy(d["refinement_controls"]).reshape(1, sb.N, 40).to(dev)
    s_mine = sm.score(sb, out.reshape(1, sb.N, 40))["scores"][0].cpu().numpy()
    s_ref = sm.score(sb, want)["scores"][0].cpu().numpy()
    l_mine, l_ref = np.maximum(5e-4 - s_mine, 0)[v].mean(), np.maximum(5e-4 - s_ref, 0)[v].mean()
    assert abs(l_mine - l_ref) <= 0.01 * l_ref + 1e-4, (l_mine, l_ref)
    assert abs(int((s_mine[v] > 0).sum()) - int((s_ref[v] > 0).sum())) <= max(2, int(0.02 * v.sum()))
    # run to run: bit-identical
    out2 = sm.refinement(sb, cin, clist, iters=meta["refinement"])
    assert torch.equal(out, out2)
    # a list too short for the reference's indices is the reference's IndexError
    with pytest.raises(IndexError):
        sm.refinement(sb, cin, clist[:60].contiguous())


@pytest.mark.parametrize("name", REFINEMENT_CASES)
def test_region_with_refinement_matches_the_reference_harness(name):
    from pstl_diffusion_policy_amd.engine import acc_from_counts
    dev = torch.device("cuda:0")
    d = load_golden(name)
    meta, sb, sm = _setup(d, dev)
    out = sm.sampling_region(sb, meta["steps"], torch.from_numpy(d["x_T"]).to(dev), torch.from_numpy(d["z"]).to(dev),
                             **region_kwargs(meta))
    N = sb.N
    np.testing.assert_allclose(out["controls_list"].reshape(meta["steps"], N, 20, 2).cpu().numpy(), d["controls_list"],
                               rtol=0, atol=1e-4)
    np.testing.assert_allclose(out["rect_controls"].reshape(N, 20, 2).cpu().numpy(), d["rect_controls"], rtol=0, atol=1e-4)
    got = out["refinement_controls"].reshape(N, 20, 2).cpu().numpy()
    ok = np.abs(got - d["refinement_controls"]).reshape(N, -1).max(axis=1) <= 1e-4
    assert ok.mean() >= 0.5, ok.mean()
    np.testing.assert_allclose(out["final_scores"].cpu().numpy()[ok], d["final_scores"][ok], rtol=1e-4, atol=2e-3)
    acc, _ = acc_from_counts(out["counts"])
    assert abs(acc - float(d["final_acc"])) <= 0.03


def test_cli_refinement_runs(capsys):
    """The CLI mirror accepts --refinement (README-style command, 100 diffusion steps) and refuses a list that is too short,
    as the reference's indexing does."""
    from pstl_diffusion_policy_amd import nusc_train as nt
    base = ["-e", "e7_ours", "--diffusion", "--stl_weight", "0.0", "--load_stlp", "--rect_head", "--flex", "--diverse_loss",
            "--multi_cands", "5", "--test", "--run_sampling_test", "--skip_nusc_load", "-b", "4", "--n_trials", "0",
            "--n_neighbors", "3", "--sampling_size", "16", "--n_randoms", "16", "--refinement"]
    md = nt.main(base)
    assert "###[00]" in capsys.readouterr().out and 0.0 <= md("acc") <= 1.0
    with pytest.raises(IndexError):
        nt.main(base + ["--diffusion_steps", "50"])


def test_refinement_is_shard_invariant_where_blocks_are_regrouped_by_xcd():
    """Batches of >= 8 scenes with S % 64 == 0 run their wavefronts by (scene, mode) with the blocks regrouped per XCD
    (stl_kernels.hip, virt_block): the per-scene lists of rows to mix and the staged scene tables must follow the block a
    workgroup STANDS FOR, not blockIdx.x.  20 scenes (two full runs of eight + a partial one that keeps its order) against the
    same scenes in shards of 4 (no regrouping at all): given the batch's loss scale (the reference's loss is a MEAN over the
    batch, so 1/N reaches every gradient and, through Adam's eps, the last bits) a row's arithmetic knows nothing of its batch,
    so bit for bit -- and every iteration's gradient too.  Some rows must actually be mixed, or the test says nothing.
    (Before the lists followed virt_block, rows were evaluated against a neighbouring scene's tables: errors of ~1.0.)"""
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    bs, S, K, n_list, iters = 20, 64, 2, 100, 6
    scene = make_scene_batch(bs, K=K, S=S, seed=41, invalid_lane_frac=0.2, stlp_mode="wide")
    sb = SceneBatch(scene, S, hp, dev)
    sm = Sampler(PackedWeights(golden_weights(), dev), hp)
    g = torch.Generator(device=dev).manual_seed(17)
    clist = torch.rand(n_list, sb.N, 40, device=dev, generator=g) * 2 - 1
    cin = (torch.rand(sb.N, 40, device=dev, generator=g) * 2 - 1).contiguous()
    out, tr = sm.refinement(sb, cin, clist, iters=iters, trace=True)
    mixed = (out != cin).any(dim=1)
    assert 0.05 * sb.N < int(mixed.sum()) < sb.N, int(mixed.sum())
    rps = 3 * S
    for lo in range(0, bs, 4):
        sub = SceneBatch({k: v[lo:lo + 4].clone() for k, v in scene.items()}, S, hp, dev)
        sub.grad_scale = sb.grad_scale
        r0, r1 = lo * rps, (lo + 4) * rps
        o, t = sm.refinement(sub, cin[r0:r1].contiguous(), clist[:, r0:r1].contiguous(), iters=iters, trace=True)
        assert torch.equal(o, out[r0:r1]), lo
        assert torch.equal(t, tr[:, r0:r1]), lo
