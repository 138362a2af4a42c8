"""GPU test (-m gpu) of the closed-loop caller: one scene per simulation step, maximize=True guidance,
fixed STL parameters -- the small-batch use of the same kernels (parity of that mode is covered by the sim_maximize*
golden cases in test_gpu_parity.py)."""
import math

import pytest
import torch

from conftest import golden_weights

pytestmark = pytest.mark.gpu


def test_receding_horizon_loop_runs_and_is_reproducible():
    from pstl_diffusion_policy_amd.nusc_sim import closed_loop
    sd = golden_weights()
    a = closed_loop(sd, n_sim_steps=4, K=4, diffusion_steps=20, guidance_before=5, seed=3, verbose=False)
    b = closed_loop(sd, n_sim_steps=4, K=4, diffusion_steps=20, guidance_before=5, seed=3, verbose=False)
    assert len(a) == 4
    for ra, rb in zip(a, b):
        assert all(math.isfinite(ra[k]) for k in ("best_score", "x", "y", "v", "latency_s"))
        assert (ra["best_score"], ra["x"], ra["y"], ra["v"]) == (rb["best_score"], rb["x"], rb["y"], rb["v"])
    assert a[-1]["x"] > a[0]["x"]                      # the ego moves forward


def test_graph_replay_equals_eager_launches():
    """GraphPlanner: the launches of a planning step captured once and replayed with new observations and seeds (read by the
    kernels from the device-resident pstl_dyn block) give, step for step, the very numbers the eager launches give."""
    from pstl_diffusion_policy_amd.nusc_sim import closed_loop
    sd = golden_weights()
    kw = dict(n_sim_steps=5, K=8, diffusion_steps=30, guidance_before=6, seed=7, verbose=False)
    a = closed_loop(sd, graph=True, **kw)
    b = closed_loop(sd, graph=False, **kw)
    for ra, rb in zip(a, b):
        assert (ra["best_score"], ra["x"], ra["y"], ra["v"]) == (rb["best_score"], rb["x"], rb["y"], rb["v"])
    assert len({r["best_score"] for r in a}) > 1      # the replays did see different inputs


def test_select_plan_is_the_reference_masked_argmax():
    """pstl_select_plan against the reference's own three lines (nusc_sim.py:677-683): scores of modes 1, 2 set to -10000,
    torch.argmax over the flattened (S,3) scores -- first maximum on ties --, that row's controls."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sm = Sampler(PackedWeights(golden_weights(), dev), hp)
    g = torch.Generator().manual_seed(2)
    for S in (8, 64, 100):
        sb = SceneBatch(make_scene_batch(1, K=2, S=S, seed=4, stlp_mode="wide"), S, hp, dev)
        scores = torch.randn(S, 3, generator=g)
        scores[S // 3, 0] = scores[:, 0].max()              # a tie: the first maximum must win
        scores[:, 1:] += 5.0                                # the other modes score higher and must be ignored
        controls = torch.randn(3 * S, 40, generator=g)
        got = sm.select_plan(sb, scores.reshape(-1).contiguous().to(dev), controls.to(dev)).cpu()
        ref = scores.clone()
        ref[:, 1:3] = -10000
        idx = int(torch.argmax(ref))
        assert idx % 3 == 0
        assert got[0] == controls[idx, 0] and got[1] == controls[idx, 1] and got[2] == ref.flatten()[idx]
        assert got[3:4].view(torch.int32).item() == 0       # the domain word: clear


def test_reference_command_line_runs(capsys):
    """README command of the guided closed-loop run, through the mirror's main()."""
    from pstl_diffusion_policy_amd import nusc_sim
    recs = nusc_sim.main("-e e7_ours --diffusion --stl_weight 0.0 --rect_head --flex --diverse_loss --multi_cands 5 --test "
                         "-P e7_ours --filter_traj 0 --test_scenes --viz_last --guidance --guidance_before 10 "
                         "--guidance_niters 1 --guidance_lr 0.04 --suffix sim_guide --n_trials 3 --diffusion_steps 20 "
                         "--n_neighbors 4 --allow_random_init".split())
    assert len(recs) == 3 and "median latency" in capsys.readouterr().out
    with pytest.raises(SystemExit):      # -P names no file: an error unless random init is asked for
        nusc_sim.main("--diffusion --rect_head --diverse_loss --test -P e7_ours --n_trials 1".split())


@pytest.mark.timing
def test_latency_of_a_simulation_step_at_the_reference_settings(capsys):
    """The reference's closed-loop settings (nusc_sim.py: 64 samples x 3 modes = 192 rows, 100 diffusion steps, K = 8
    neighbours, maximize guidance on the last 10 steps, 5 candidates + RefineNet): wall-clock latency per simulation step
    with a device synchronisation on both sides, printed for the record (`pytest -s`, or the captured output of a failure)
    and held below 3 ms (measured on one MI355X: 0.86-0.91 ms, one HIP-graph replay per step)."""
    from pstl_diffusion_policy_amd.nusc_sim import closed_loop
    def run():
        recs = closed_loop(golden_weights(), n_sim_steps=16, K=8, S=64, diffusion_steps=100, multi_cands=5, guidance=True,
                           guidance_before=10, guidance_lr=0.04, seed=1, verbose=False)
        lats = sorted(r["latency_s"] for r in recs[3:])
        return recs, lats[len(lats) // 2] * 1e3, lats[-1] * 1e3

    recs, med, worst = run()
    if med >= 1.5:      # the latency includes the host: one busy spell of a shared box gets a second run, a regression fails both
        recs, med, worst = min(run(), (recs, med, worst), key=lambda r: r[1])
    with capsys.disabled():
        print("\nclosed loop: median %.2f ms, worst %.2f ms per simulation step (192 rows, 100 steps, K=8, guidance)" % (med, worst))
    assert all(math.isfinite(r["best_score"]) for r in recs)
    # wall clock including the host, on a shared box: the measured median is printed (0.84-0.91 ms with the HIP-graph replay;
    # round 3: 1.3, round 2: 2.7).  The default bound, 3 ms, only catches a return to per-launch host latency on a pool whose
    # hosts are shared; PSTL_STRICT_LATENCY=1 (a box of one's own) holds the run to the 1.8 ms of rounds 3-4
    import os
    assert med < (1.8 if os.environ.get("PSTL_STRICT_LATENCY") == "1" else 3.0), med


def test_parameters_in_device_memory_equal_parameters_by_value():
    """ABI 4 (pstl_cfg.dyn): noise seed and guidance-loss scale read by the kernels from a 16-byte device block give, bit for
    bit, what the by-value arguments give -- x_T, a guided rollout and its candidates, in the throughput layout (672 scenes) and
    in the latency layout (2 scenes)."""
    import numpy as np
    from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sm = Sampler(PackedWeights(golden_weights(), dev), hp)
    guid = dict(enabled=True, before=3, niters=1, lr=0.01)
    for bs in (2, 112):
        scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=3, S=64, seed=8, invalid_lane_frac=0.2, stlp_mode="wide").items()
                 if k not in ("params", "pre_stlp", "tj_scores_prior")}
        seed = 0x1234567890abcdef
        ref = sm.sampling_region(SceneBatch(scene, 64, hp, dev), 8, None, None, rect_head=True, multi_cands=3, guidance=guid,
                                 seed=seed, want_scores3=False)
        dyn = torch.zeros(4, dtype=torch.float32, device=dev)
        dyn[:2] = torch.from_numpy(np.array([seed & 0xffffffff, seed >> 32], dtype=np.uint32).view(np.float32)).to(dev)
        sb = SceneBatch(scene, 64, hp, dev, dyn=dyn)          # writes the loss scale into dyn[2] on the device
        got = sm.sampling_region(sb, 8, None, None, rect_head=True, multi_cands=3, guidance=guid, seed=1, want_scores3=False)
        for k in ("final_controls", "final_scores", "sel_controls", "cand_scores"):
            assert torch.equal(got[k], ref[k]), (bs, k)
        assert torch.isfinite(got["final_controls"]).all()


def test_graph_replay_of_the_whole_region_equals_eager_including_counters():
    """engine.GraphCapture over the complete sampling region (what bench.py replays for batches up to 98 304 rows): three
    replays with different seeds against eager calls -- controls, scores, the satisfaction COUNTERS and the diversity totals.
    (Round 4: the counters were zeroed by hipMemsetAsync, whose graph node did not replay: from the second replay on they
    accumulated on top of the previous values.  They are zeroed by a kernel now; the ADE / FDE minima likewise.)"""
    from pstl_diffusion_policy_amd.engine import DynBlock, GraphCapture, PackedWeights, Sampler, SceneBatch
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sm = Sampler(PackedWeights(golden_weights(), dev), hp)
    bs, S, steps = 12, 64, 12
    scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=2, S=S, seed=1000, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    guid = dict(enabled=True, before=4, niters=1, lr=0.01)
    N = bs * S * 3
    vsum = float(sum(scene[k].sum().item() for k in ("curr_id", "left_id", "right_id"))) * S
    kw = dict(rect_head=True, multi_cands=3, guidance=guid, want_scores3=False, diversity=True)
    dyn = DynBlock(dev)
    dyn.set(0, SceneBatch.loss_scale(vsum, N))

    def body():
        sb = SceneBatch(scene, S, hp, dev, global_valid_sum=vsum, global_rows=N, dyn=dyn.dev, scale_in_dyn=True)
        o = sm.sampling_region(sb, steps, None, None, seed=0, **kw)
        return o["counts"], o["div_totals"], o["final_controls"], o["final_scores"]

    g = GraphCapture(body)
    for seed in (5, 6, 5):
        dyn.set(seed)
        got = [t.clone() for t in g.replay()]
        ref = sm.sampling_region(SceneBatch(scene, S, hp, dev, global_valid_sum=vsum, global_rows=N), steps, None, None, seed=seed, **kw)
        for a, k in zip(got, ("counts", "div_totals", "final_controls", "final_scores")):
            assert torch.equal(a, ref[k]), (seed, k)
        assert 0 < int(got[0][0]) <= int(got[0][1]) <= N


def test_back_to_back_replays_each_see_their_own_seed():
    """DynBlock.set() + GraphCapture.replay() twenty times with NO synchronisation in between (what bench.py's timed
    loop does): every replay must run on the seed set for it -- its outputs, cloned on the stream, equal the eager run of that
    seed.  (With one pinned mirror rewritten per call, the queued 16-byte copies read whatever seed the host wrote last.)"""
    from pstl_diffusion_policy_amd.engine import DynBlock, GraphCapture, PackedWeights, Sampler, SceneBatch
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sm = Sampler(PackedWeights(golden_weights(), dev), hp)
    bs, S, steps = 64, 64, 20
    scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=2, S=S, seed=3, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    N = bs * S * 3
    vsum = float(sum(scene[k].sum().item() for k in ("curr_id", "left_id", "right_id"))) * S
    kw = dict(rect_head=False, multi_cands=1, want_scores3=False)
    dyn = DynBlock(dev)
    dyn.set(0, SceneBatch.loss_scale(vsum, N))

    def body():
        sb = SceneBatch(scene, S, hp, dev, global_valid_sum=vsum, global_rows=N, dyn=dyn.dev, scale_in_dyn=True)
        return sm.sampling_region(sb, steps, None, None, seed=0, **kw)["final_controls"]

    g = GraphCapture(body)
    seeds = list(range(100, 120))
    got = []
    for seed in seeds:               # no synchronisation: the copies queue up behind the replays
        dyn.set(seed)
        got.append(g.replay().clone())
    torch.cuda.synchronize()
    sbe = SceneBatch(scene, S, hp, dev, global_valid_sum=vsum, global_rows=N)
    for seed, a in zip(seeds, got):
        ref = sm.sampling_region(sbe, steps, None, None, seed=seed, **kw)["final_controls"]
        assert torch.equal(a, ref), seed
    assert not torch.equal(got[0], got[1])
