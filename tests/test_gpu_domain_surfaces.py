"""GPU tests (-m gpu): the split-f16 domain guard on EVERY surface that hands results back.

Outside |layer input| < 4094 the default (split-f16) MLP chains leave plausible-looking garbage, not NaNs, and set a sticky
status word in the packed weight buffer.  Round 3 read that word in run_sampling_test and bench.py only.  Here each other
surface -- the closed loop (nusc_sim.closed_loop), the training step (RectTrainer.train_step, nusc_train.run_training) and the
drop-in API itself (Net.forward, Net.rect_forward, nusc_train.diffusion_rollout) -- is driven with weights that are INSIDE the
weight domain (so nothing falls back up front) but push one hidden activation far outside the half range, and must (i) say so
(RuntimeWarning), (ii) end on the exact-fp32 kernels, (iii) return what the exact-fp32 kernels / the CPU oracle compute.

The trip wire: hidden unit J of a chain's first layer gets bias 5000 (5000 * 2^4 > 65504: its half piece is infinite) and its
outgoing weights in layer 2 are zeroed -- in exact arithmetic the unit contributes nothing, so the network is a perfectly
ordinary one for the exact-fp32 kernels and the oracle, while the split form multiplies 0 by infinity."""
import warnings

import numpy as np
import pytest
import torch

from conftest import golden_meta, golden_weights, load_golden, scene_from_golden
from test_gpu_reference_surface import _setup, replay_randn_like

pytestmark = pytest.mark.gpu
J = 37


def tripwire(sd, net_name):
    sd = {k: np.array(v, copy=True) for k, v in sd.items()}
    sd[net_name + ".0.bias"][J] = np.float32(5000.0)
    sd[net_name + ".2.weight"][:, J] = np.float32(0.0)
    return sd


def _load(net, sdn):
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=False)


def test_closed_loop_plans_again_on_exact_fp32_and_stays_there():
    from pstl_diffusion_policy_amd.nusc_sim import closed_loop
    sd = tripwire(golden_weights(), "policy_net")
    kw = dict(n_sim_steps=3, K=4, diffusion_steps=20, guidance_before=5, seed=3, verbose=False)
    with pytest.warns(RuntimeWarning, match="exact-fp32") as rec:
        got = closed_loop(sd, **kw)
    assert len([w for w in rec if "exact-fp32" in str(w.message)]) == 1      # one switch, at the first step; then it stays
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        want = closed_loop(sd, chain_waves=8, **kw)
        clean = closed_loop(golden_weights(), **kw)                            # ordinary weights: no warning at all
    assert len(clean) == 3
    for a, b in zip(got, want):       # same Philox seeds, same kernels from the re-planned first step on: bit-identical
        assert (a["best_score"], a["x"], a["y"], a["v"]) == (b["best_score"], b["x"], b["y"], b["v"])


def test_net_forward_repeats_the_call_on_exact_fp32():
    from oracle import pstl_oracle as orc
    nt, d, meta, args, net, batch, scene, dev = _setup("e7_steps12")
    sdn = tripwire(golden_weights(), "policy_net")
    _load(net, sdn)
    bs, S = meta["bs"], meta["S"]
    N = bs * S * 3
    new_batch = nt.augment_batch_data(batch, scene["stlp_modes"][:, 0], args, n_randoms=S)
    x = torch.randn(N, 40, generator=torch.Generator().manual_seed(3))
    t = 7
    ext = {"timestep": torch.full((N, 1), t, dtype=torch.long, device=dev), "highlevel": new_batch["highlevel_dense"],
           "noise": x.to(dev), "stlp": new_batch["stlp_dense"]}
    assert net.chain_arith() == 0 and net.domain_check == "eager"
    with pytest.warns(RuntimeWarning, match="exact-fp32"):
        eps, feature = net(new_batch, ext=ext, get_feature=True, n_randoms=S)
    assert net.chain_waves == 8 and not net.packed().chain_overflowed()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        eps2 = net(new_batch, ext=ext, prev_feature=feature, n_randoms=S)
    assert torch.equal(eps, eps2)
    rows = orc.Rows({k: v for k, v in scene_from_golden(d).items()}, S, net.hparams())
    feat_rows = orc.rows_from_scenes(torch.from_numpy(d["feature_scene"]), 3 * S)
    ref = orc.policy_eps(sdn, feat_rows, x, t, rows.hl, rows.stlp)
    np.testing.assert_allclose(eps.reshape(N, 40).cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)
    # deferred mode: nothing is read inside the call; check_domain() reports (and can refuse)
    net.chain_waves, net.domain_check = None, "deferred"
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net(new_batch, ext=ext, prev_feature=feature, n_randoms=S)
    with pytest.raises(FloatingPointError):
        net.check_domain(fallback=False)
    assert net.check_domain() is False       # the flag was cleared by the check that raised


def test_rect_forward_repeats_the_call_on_exact_fp32():
    from oracle import pstl_oracle as orc
    nt, d, meta, args, net, batch, scene, dev = _setup("e7_wide")
    sdn = tripwire(golden_weights(), "rect_net")
    _load(net, sdn)
    bs, S = meta["bs"], meta["S"]
    N = bs * S * 3
    new_batch = nt.augment_batch_data(batch, scene["stlp_modes"][:, 0], args, n_randoms=S)
    hl = new_batch["highlevel_dense"]
    feature = net.scene_feature(new_batch, S * 3)
    g = torch.Generator().manual_seed(5)
    init = (torch.randn(N, 20, 2, generator=g) * torch.tensor([0.1, 1.0])).clamp(-0.5, 0.5)
    prev = torch.randn(N, generator=g)
    with pytest.warns(RuntimeWarning, match="exact-fp32"):
        rect = net.rect_forward(feature, hl, new_batch["stlp_dense"][:, 0], init.to(dev), prev.to(dev))
    assert net.chain_waves == 8
    rows = orc.Rows({k: v for k, v in scene_from_golden(d).items()}, S, net.hparams())
    feat_scene = orc.encode_feat(sdn, {k: v for k, v in scene_from_golden(d).items()})
    ref = orc.rect_forward(sdn, feat_scene, rows, init, prev, net.hparams()["n_shards"], diverse=True)
    np.testing.assert_allclose(rect.cpu().numpy().reshape(N, 40), _np(ref).reshape(N, 40), rtol=0, atol=1e-4)


def _np(t):
    return t.numpy() if hasattr(t, "numpy") else np.asarray(t)


def test_diffusion_rollout_repeats_the_rollout_from_the_same_draws():
    from oracle import pstl_oracle as orc
    nt, d, meta, args, net, batch, scene, dev = _setup("e7_steps12")
    sdn = tripwire(golden_weights(), "policy_net")
    _load(net, sdn)
    bs, S, steps = meta["bs"], meta["S"], meta["steps"]
    N = bs * S * 3
    coeffs = nt.get_diffusion_coeffs(args)
    new_batch = nt.augment_batch_data(batch, scene["stlp_modes"][:, 0], args, n_randoms=S)
    draws = [torch.from_numpy(d["x_T"])] + [torch.from_numpy(z) for z in d["z"][:-1]]
    noise = torch.empty(N, 40, device=dev)
    with replay_randn_like(draws):        # the draws are consumed ONCE: the repeat must reuse them, not draw again
        with pytest.warns(RuntimeWarning, match="exact-fp32"):
            controls, feature, clist = nt.diffusion_rollout(noise, net, new_batch, new_batch["highlevel_dense"], None, args,
                                                            coeffs, n_randoms=S, return_feature=True)
    assert net.chain_waves == 8
    ref = orc.sampling_region(sdn, scene_from_golden(d), S, steps, net.hparams(), torch.from_numpy(d["x_T"]),
                              torch.from_numpy(d["z"]), rect_head=True, multi_cands=meta["multi_cands"])
    got = torch.stack(clist, 0).cpu().numpy()
    np.testing.assert_allclose(got, _np(ref["controls_list"]).reshape(got.shape), rtol=0, atol=1e-4)


@pytest.mark.parametrize("which", ["rect_net", "policy_net"])
def test_train_step_repeats_the_step_before_the_optimiser_sees_it(which):
    """The flag is read BEFORE optimizer.step(): loss, scores and the weights after the step are those of a run that used the
    exact-fp32 kernels from the start (same Philox seed), bit for bit."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, RectTrainer, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sdn = tripwire(golden_weights(), which)
    scene = make_scene_batch(6, K=3, S=64, seed=5, invalid_lane_frac=0.2, stlp_mode="wide")

    def run(chain_waves):
        sd = {k: torch.from_numpy(v).to(dev) for k, v in sdn.items()}
        params = {k: sd[k].clone().requires_grad_() for k in RectTrainer.NAMES}
        opt = torch.optim.Adam([params[k] for k in RectTrainer.NAMES], lr=3e-4)
        sm = Sampler(PackedWeights(dict(sd, **{k: v.detach() for k, v in params.items()}), dev), hp, chain_waves=chain_waves)
        tr = RectTrainer(sm)
        loss, scores = tr.train_step(SceneBatch(scene, 64, hp, dev), params, opt, 12, seed=11, multi_cands=5)
        return sm, float(loss), scores.cpu(), {k: v.detach().cpu() for k, v in params.items()}

    with pytest.warns(RuntimeWarning, match="exact-fp32"):
        sm, loss, scores, after = run(0)
    assert sm.chain_waves == 8 and "domain" in sm.chain_fallback and not sm.w.chain_overflowed()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _, loss8, scores8, after8 = run(8)
    assert loss == loss8 and torch.equal(scores, scores8) and np.isfinite(loss)
    for k in after:
        assert torch.equal(after[k], after8[k]), k


def test_training_cli_stays_on_exact_fp32_after_the_flag(capsys):
    """run_training builds a new sampler (and a new packed buffer) per batch: the fallback must outlive them."""
    from pstl_diffusion_policy_amd import nusc_train as nt
    args = nt.generate_parser(["--diffusion", "--stl_weight", "1.0", "--load_stlp", "--load_tj", "--rect_head", "--flex",
                               "--multi_cands", "3", "--diffusion_steps", "10", "--n_randoms", "16", "--sampling_size", "16",
                               "--n_neighbors", "3", "--lr", "3e-4", "-b", "4", "--epochs", "1", "--n_trials", "2",
                               "--skip_nusc_load"])
    net = nt.Net(args).cuda()
    _load(net, {k: v for k, v in tripwire(golden_weights(), "rect_net").items() if not k.startswith("merge_net")})
    loader = nt.SyntheticLoader(args, n_batches=3)
    with pytest.warns(RuntimeWarning, match="exact-fp32") as rec:
        md = nt.run_training(loader, net, nt.get_diffusion_coeffs(args), args)
    assert net.chain_waves == 8
    assert len([w for w in rec if "exact-fp32" in str(w.message)]) == 1      # batch 0 switches; batches 1 and 2 start exact
    assert np.isfinite(md("loss"))
