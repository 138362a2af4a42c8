"""GPU tests (-m gpu) of the reference-surface mirror (pstl_diffusion_policy_amd.nusc_train / nusc_model): the same
call sequence the reference harness makes (nusc_train.py:957-1105), with torch.randn_like replaying the draws recorded
from the reference run, must reproduce the golden outputs."""
import contextlib

import numpy as np
import pytest
import torch

from conftest import golden_meta, golden_weights, load_golden, scene_from_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


@contextlib.contextmanager
def replay_randn_like(draws):
    orig = torch.randn_like
    it = iter(draws)

    def fake(x, *a, **k):
        return next(it).to(device=x.device, dtype=x.dtype)

    torch.randn_like = fake
    try:
        yield
    finally:
        torch.randn_like = orig


def _setup(name, extra=()):
    from pstl_diffusion_policy_amd import nusc_train as nt
    d = load_golden(name)
    meta = golden_meta(d)
    argv = ["--diffusion", "--load_stlp", "--flex", "--test", "--run_sampling_test", "--diffusion_steps",
            str(meta["steps"]), "--sampling_size", str(meta["S"]), "--n_randoms", str(meta["S"]), "--n_neighbors",
            str(meta["K"])]
    if meta["rect_head"]:
        argv += ["--rect_head", "--diverse_loss", "--multi_cands", str(meta["multi_cands"])]
    if meta["guidance"]:
        argv += ["--guidance", "--guidance_before", str(meta["guidance_before"]), "--guidance_niters",
                 str(meta["guidance_niters"]), "--guidance_lr", str(meta["guidance_lr"])]
    args = nt.generate_parser(argv + list(extra))
    args.diff_full = True
    net = nt.Net(args).cuda()
    sd = {k: torch.from_numpy(v) for k, v in golden_weights().items()}
    net.load_state_dict(sd, strict=False)
    dev = torch.device("cuda:0")
    scene = {k: torch.from_numpy(v).to(dev) for k, v in scene_from_golden(d).items()}
    bs, S = meta["bs"], meta["S"]
    batch = {k: scene[k] for k in ("ego_traj", "neighbors", "currlane_wpts", "leftlane_wpts", "rightlane_wpts",
                                   "curr_id", "left_id", "right_id")}
    batch["neighbor_trajs_aug"] = scene["neighbors_traj"][..., :7]
    batch["gt_high_level"] = torch.zeros(bs, 1, device=dev)
    batch["pre_stlp"] = scene["stlp_modes"].reshape(bs, 1, 3, 1, 6).repeat(1, S, 1, 1, 1)
    return nt, d, meta, args, net, batch, scene, dev


@pytest.mark.parametrize("name,dense", [("e7_wide", False), ("e7_wide", True), ("e7_guid", False), ("e5_steps10", False)])
def test_harness_call_sequence_reproduces_reference(name, dense):
    nt, d, meta, args, net, batch, scene, dev = _setup(name)
    bs, S = meta["bs"], meta["S"]
    N = bs * S * 3
    stls = nt.build_stl_cache(args)
    coeffs = nt.get_diffusion_coeffs(args)
    new_batch = nt.augment_batch_data(batch, scene["stlp_modes"][:, 0], args, n_randoms=S, dense=dense)
    np.testing.assert_array_equal(new_batch["stlp_dense"].cpu().numpy(), d["in_stlp_dense"])
    np.testing.assert_array_equal(new_batch["valids_dense"].cpu().numpy(), d["in_valids_dense"])
    np.testing.assert_array_equal(new_batch["highlevel_dense"].cpu().numpy(), d["in_highlevel_dense"])
    hl = new_batch["highlevel_dense"]
    states_flat = scene["ego_traj"][:, 0, :4].unsqueeze(1).unsqueeze(1).repeat(1, S, 3, 1).reshape(N, 4)
    draws = [torch.from_numpy(d["x_T"])] + [torch.from_numpy(z) for z in d["z"][:-1]]
    noise = torch.empty(N, 40, device=dev)
    gex = (new_batch, states_flat, stls) if args.guidance else None
    with replay_randn_like(draws):
        controls, feature, clist = nt.diffusion_rollout(noise, net, new_batch, hl, None, args, coeffs,
                                                        n_randoms=S, return_feature=True, guidance_extras=gex)
    assert feature.shape == (N, 224)
    np.testing.assert_allclose(feature.reshape(bs, S * 3, 224)[:, 0].cpu().numpy(), d["feature_scene"], rtol=0, atol=2e-5)
    got = torch.stack(clist, 0).cpu().numpy()
    assert np.abs(got - d["controls_list"]).max() <= TOL
    nn_controls = controls
    if args.rect_head:
        mc = args.multi_cands
        states_mul = states_flat.repeat(mc, 1)
        ctrls_mul = torch.cat(clist[-mc:], dim=0)
        trajs_mul = nt.generate_trajs(states_mul, ctrls_mul, args.dt)
        if dense:   # strip the side channel: exactly what the reference passes (row-replicated tensors only)
            prev_in = nt.pre_prepare_stl_cache(new_batch, dense_trajs=trajs_mul[:, :-1], repeat_n=mc)
            prev_in["_pstl"] = None
        else:
            prev_in = nt.pre_prepare_stl_cache(new_batch, dense_trajs=trajs_mul[:, :-1], repeat_n=mc)
        _, sc_hist, _ = nt.compute_stl_dense(prev_in, stls, hl.repeat(mc, 1), prev_in["dense_valids"].reshape(-1), args)
        sc_hist = sc_hist.reshape(mc, N)
        np.testing.assert_allclose(sc_hist.cpu().numpy(), d["cand_scores"], rtol=5e-5, atol=1e-3)
        sc_max, sc_idx = torch.max(sc_hist, dim=0)
        c_max = ctrls_mul.reshape(mc, N, args.nt, 2)[sc_idx, torch.arange(N, device=dev)]
        nn_controls = net.rect_forward(feature, hl, new_batch["stlp_dense"][:, 0], c_max, sc_max)
        top2 = np.sort(d["cand_scores"], axis=0)[-2:]
        clear = (top2[1] - top2[0]) > 1e-3
        np.testing.assert_allclose(nn_controls.cpu().numpy()[clear], d["rect_controls"][clear], rtol=0, atol=TOL)
    trajs = nt.generate_trajs(states_flat, nn_controls, args.dt).reshape(N, args.nt + 1, 4)
    np.testing.assert_allclose(trajs.cpu().numpy(), d["final_trajs"], rtol=1e-5, atol=2e-3)
    stl_in = nt.pre_prepare_stl_cache(new_batch, dense_trajs=trajs[:, :-1])
    scores_list, scores, acc, scene_acc = nt.compute_stl_dense(stl_in, stls, hl, stl_in["dense_valids"], args, scene=True)
    np.testing.assert_allclose(torch.stack(scores_list[:3]).cpu().numpy(), d["final_scores3"], rtol=1e-4, atol=2e-3)
    np.testing.assert_array_equal(scores.cpu().numpy() > 0, d["final_scores"] > 0)
    assert float(acc) == float(d["final_acc"]) and float(scene_acc) == float(d["final_scene_acc"])
    assert torch.equal(scores_list[3], torch.ones_like(scores))


def test_net_forward_is_one_denoiser_evaluation():
    from oracle import pstl_oracle as orc
    nt, d, meta, args, net, batch, scene, dev = _setup("e7_steps12")
    bs, S = meta["bs"], meta["S"]
    N = bs * S * 3
    new_batch = nt.augment_batch_data(batch, scene["stlp_modes"][:, 0], args, n_randoms=S)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, 40, generator=g)
    t = 7
    ext = {"timestep": torch.full((N, 1), t, dtype=torch.long, device=dev), "highlevel": new_batch["highlevel_dense"],
           "noise": x.to(dev), "stlp": new_batch["stlp_dense"]}
    eps, feature = net(new_batch, ext=ext, get_feature=True, n_randoms=S)
    eps2 = net(new_batch, ext=ext, prev_feature=feature, n_randoms=S)
    assert torch.equal(eps, eps2) and eps.shape == (N, 20, 2)
    sd = golden_weights()
    rows = orc.Rows({k: v for k, v in scene_from_golden(d).items()}, S, net.hparams())
    feat_rows = orc.rows_from_scenes(torch.from_numpy(d["feature_scene"]), 3 * S)
    ref = orc.policy_eps(sd, feat_rows, x, t, rows.hl, rows.stlp)
    np.testing.assert_allclose(eps.reshape(N, 40).cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)
    # state_dict surface: the reference's keys and shapes
    keys = sorted(net.state_dict().keys())
    assert keys == sorted(sd.keys())
    assert all(tuple(net.state_dict()[k].shape) == sd[k].shape for k in keys)


def test_cli_runs_the_sampling_test(capsys):
    from pstl_diffusion_policy_amd import nusc_train as nt
    md = nt.main(["-e", "e7_ours", "--diffusion", "--stl_weight", "0.0", "--load_stlp", "--rect_head", "--flex",
                  "--diverse_loss", "--multi_cands", "5", "--test", "--run_sampling_test", "--skip_nusc_load",
                  "--viz_correct", "-b", "16", "--n_trials", "1", "--n_neighbors", "4", "--diffusion_steps", "20",
                  "--guidance", "--guidance_before", "5", "--guidance_niters", "1", "--guidance_lr", "0.01",
                  "--n_rolls", "1", "--time_profile"])
    out = capsys.readouterr().out
    assert "###[00]" in out and "NN acc:" in out and "T:" in out and "end_diffusion-start_diffusion" in out
    assert 0.0 <= md("acc") <= 1.0 and md("time") > 0


def test_infer_gt_stlp_matches_reference():
    """infer_gt_stlp through pstl_stl_signals against the reference's own function (all four high-level labels)."""
    import os
    from pstl_diffusion_policy_amd import nusc_train as nt
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "gt_stlp.npz")))
    dev = torch.device("cuda:0")
    bc = {k[3:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("in_")}
    bc["neighbor_trajs_aug"] = bc["neighbors_traj"][..., :7]
    for flex in (0, 1):
        args = nt.generate_parser(["--diffusion", "--load_stlp"] + (["--flex"] if flex else []))
        got = nt.infer_gt_stlp(bc, bc["ego_traj"][..., :4], args).cpu().numpy()
        np.testing.assert_allclose(got, g["stlp_flex%d" % flex], rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("name", ["sim_maximize", "sim_maximize_b"])
@pytest.mark.parametrize("how", ["inplace", "replace", "no_side_channel"])
def test_closed_loop_calling_sequence_of_the_reference(name, how):
    """The reference's closed-loop caller (nusc_sim.py:429-548), replayed literally through the mirror: augment_batch_data
    with the parameters it happens to have, THEN overwrite new_batch["stlp_dense"] with the fixed values (:467-472) -- in
    place, by assigning a new tensor, or (no_side_channel) with this package's `_pstl` entry removed altogether, as for a
    batch built by the reference's own augment_batch_data -- then diffusion_rollout(maximize=True), candidate scoring through
    generate_trajs / pre_prepare_stl_cache(repeat_n) / compute_stl_dense, torch.max over the candidates, rect_forward and
    the final compute_stl_dense.  Fixtures sim_maximize*: the reference run with those fixed parameters."""
    nt, d, meta, args, net, batch, scene, dev = _setup(name, extra=["--guidance_lr", "0.04"])
    assert meta["maximize"] and args.guidance and args.rect_head
    bs, S = meta["bs"], meta["S"]
    N = bs * S * 3
    stls = nt.build_stl_cache(args)
    coeffs = nt.get_diffusion_coeffs(args)
    fixed = torch.from_numpy(d["in_stlp_dense"]).to(dev)                    # (N,1,6): what the reference run used
    # parameters the caller "happens to have" when it augments: deliberately different from the fixed ones
    batch["pre_stlp"] = batch["pre_stlp"] * 0.5 + 0.3
    new_batch = nt.augment_batch_data(batch, scene["stlp_modes"][:, 0], args, n_randoms=S, dense=(how == "no_side_channel"))
    assert not torch.equal(new_batch["stlp_dense"], fixed)
    if how == "inplace":
        for c in range(6):
            new_batch["stlp_dense"][..., c:c + 1] = fixed[..., c:c + 1]
    else:
        new_batch["stlp_dense"] = fixed.clone()
    if how == "no_side_channel":
        del new_batch["_pstl"]
    hl = new_batch["highlevel_dense"]
    states_flat = scene["ego_traj"][:, 0, :4].unsqueeze(1).unsqueeze(1).repeat(1, S, 3, 1).reshape(N, 4)
    draws = [torch.from_numpy(d["x_T"])] + [torch.from_numpy(z) for z in d["z"][:-1]]
    noise = torch.empty(N, 40, device=dev)
    gex = (new_batch, states_flat.detach(), stls)
    with replay_randn_like(draws):
        controls, feature, clist = nt.diffusion_rollout(noise, net, new_batch, hl, None, args, coeffs, return_feature=True,
                                                        guidance_extras=gex, maximize=True)
    got = torch.stack(clist, 0).cpu().numpy()
    err = np.abs(got - d["controls_list"])
    from conftest import guided_outlier_rows
    bad_rows, bad_groups = guided_outlier_rows(err, d, meta, TOL)
    assert err[:, ~bad_rows].max() <= TOL
    mc = args.multi_cands
    states_mul = states_flat.repeat(mc, 1)
    ctrls_mul = torch.cat(clist[-mc:], dim=0)
    trajs_mul = nt.generate_trajs(states_mul, ctrls_mul, args.dt)
    prev_in = nt.pre_prepare_stl_cache(new_batch, dense_trajs=trajs_mul[:, :-1], repeat_n=mc)
    _, sc_hist, _ = nt.compute_stl_dense(prev_in, stls, hl.repeat((mc, *[1] * (hl.dim() - 1))),
                                         prev_in["dense_valids"].reshape(-1), args)
    sc_hist = sc_hist.reshape(mc, N)
    keep = ~bad_rows
    np.testing.assert_allclose(sc_hist.cpu().numpy()[:, keep], d["cand_scores"][:, keep], rtol=5e-5, atol=1e-3)
    sc_max, sc_idx = torch.max(sc_hist, dim=0)
    c_max = ctrls_mul.reshape(mc, N, args.nt, 2)[sc_idx, torch.arange(N, device=dev)]
    rect = net.rect_forward(feature, hl, new_batch["stlp_dense"][:, 0], c_max.detach(), sc_max.detach(), extras=clist)
    top2 = np.sort(d["cand_scores"], axis=0)[-2:]
    ok = ((top2[1] - top2[0]) > 1e-3) & ~bad_groups
    np.testing.assert_allclose(rect.cpu().numpy()[ok], d["rect_controls"][ok], rtol=0, atol=TOL)
    rect_trajs = nt.generate_trajs(states_flat, rect, args.dt)
    stl_in = nt.pre_prepare_stl_cache(new_batch, dense_trajs=rect_trajs[:, :-1])
    _, scores_all, acc = nt.compute_stl_dense(stl_in, stls, hl, stl_in["dense_valids"], args)
    fs, fr = scores_all.cpu().numpy(), d["final_scores"]
    np.testing.assert_allclose(fs[ok], fr[ok], rtol=1e-4, atol=2e-3)
    np.testing.assert_array_equal((fs > 0)[ok], (fr > 0)[ok])                 # satisfaction masks: exact
    if ok.all():
        assert float(acc) == float(d["final_acc"])
