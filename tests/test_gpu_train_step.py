"""GPU test (-m gpu) of the RefineNet training step: loss, the gradients of the six rect_net
tensors and the weights after one Adam step, against the reference's own autograd/optimizer (golden) and the oracle."""
import numpy as np
import pytest
import torch

from conftest import golden_weights, hparams_for, load_golden, scene_from_golden

pytestmark = pytest.mark.gpu


def _setup(d, dev, chain_waves=0):
    from pstl_diffusion_policy_amd.engine import PackedWeights, RectTrainer, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    bs, S, K, steps, seed, mc = [int(v) for v in d["meta"]]
    hp = hparams_for(d)
    sd = {k: torch.from_numpy(v).to(dev) for k, v in golden_weights(d).items()}
    sm = Sampler(PackedWeights(sd, dev), hp, chain_waves=chain_waves)
    sb = SceneBatch({k: torch.from_numpy(v) for k, v in scene_from_golden(d).items()}, S, hp, dev)
    return sm, sb, sd, RectTrainer(sm)


# train_e8_heavy: rect_net with a trained network's dynamic range (tests/heavy_weights.py): the training forward pass runs
# on the split-f16 chain too
# chain_waves = 2: the training forward pass on k_chain2's tile-walking form (MODE 3: the saved activations h1, h2, pre written from
# its conversions and its head), which large batches get by default
@pytest.mark.parametrize("chain_waves", [0, 2])
@pytest.mark.parametrize("name", ["train_e8_step", "train_e8_step_b", "train_e8_heavy", "train_e8_norm", "train_e8_trained"])
def test_rect_train_step_matches_reference(name, chain_waves):
    dev = torch.device("cuda:0")
    d = load_golden(name)
    lr = float(d["meta_f"][0])
    sm, sb, sd, tr = _setup(d, dev, chain_waves)
    if chain_waves == 2 and sb.cfg(2).rows_per_scene % 16 != 0:
        pytest.skip("k_chain2 takes scenes of a multiple of 16 rows")
    feature, _, base_r = sm.encode(sb)
    init = torch.from_numpy(d["sel_controls"]).reshape(sb.N, 40).to(dev)
    prev = torch.from_numpy(d["sel_scores"]).to(dev)
    params = {k: sd[k].clone().requires_grad_() for k in tr.NAMES}
    opt = torch.optim.Adam([params[k] for k in tr.NAMES], lr=lr)
    loss, rect, scores, g = tr.loss_and_grads(sb, feature, base_r, params["rect_net.2.weight"], params["rect_net.4.weight"],
                                              init, prev)
    np.testing.assert_allclose(rect.reshape(sb.N, 20, 2).cpu().numpy(), d["rect_controls"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(scores.cpu().numpy(), d["scores"], rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(float(loss), float(d["loss"]), rtol=2e-4)
    for k in tr.NAMES:
        ref = d["grad_" + k]
        np.testing.assert_allclose(g[k].cpu().numpy(), ref, rtol=5e-3, atol=5e-5 * np.abs(ref).max(), err_msg=k)
    # the optimiser step on the device path (pstl_adam_step), with the hyper-parameters of the caller's torch.optim.Adam
    from pstl_diffusion_policy_amd.engine import DeviceAdam
    DeviceAdam.adopt(opt, [params[k] for k in tr.NAMES]).step([g[k] for k in tr.NAMES])
    assert not opt.state, "the caller's torch optimiser is adopted, never stepped"
    for k in tr.NAMES:
        got, want, gref = params[k].detach().cpu().numpy(), d["after_" + k], d["grad_" + k]
        # Adam's first step moves every weight by lr * g/(|g| + 1e-8): where the gradient is at rounding-noise level its
        # sign -- hence the direction of the step -- is not determined, so those entries may differ by up to 2*lr
        solid = np.abs(gref) > 1e-4 * np.abs(gref).max()
        np.testing.assert_allclose(got[solid], want[solid], rtol=0, atol=5e-5, err_msg=k)
        assert np.abs(got - want).max() <= 2 * lr + 1e-6, k
        assert solid.mean() > 0.5, k


def test_rect_train_step_against_oracle_on_fresh_scenes():
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.engine import PackedWeights, RectTrainer, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    bs, S, K = 5, 64, 4
    scene = make_scene_batch(bs, K=K, S=S, seed=77, invalid_lane_frac=0.3, stlp_mode="wide")
    sdn = golden_weights()
    g = torch.Generator().manual_seed(2)
    N = bs * S * 3
    init = (torch.randn(N, 20, 2, generator=g) * torch.tensor([0.1, 1.0])).clamp(-0.5, 0.5)
    prev = torch.randn(N, generator=g)
    ref = orc.rect_train_step(sdn, {k: v.numpy() for k, v in scene.items()}, S, hp, init, prev, 3e-4)
    sd = {k: torch.from_numpy(v).to(dev) for k, v in sdn.items()}
    sm = Sampler(PackedWeights(sd, dev), hp)
    sb = SceneBatch(scene, S, hp, dev)
    tr = RectTrainer(sm)
    feature, _, base_r = sm.encode(sb)
    loss, rect, scores, gr = tr.loss_and_grads(sb, feature, base_r, sd["rect_net.2.weight"], sd["rect_net.4.weight"],
                                               init.reshape(N, 40).to(dev), prev.to(dev))
    np.testing.assert_allclose(float(loss), float(ref["loss"]), rtol=2e-4)
    for k in tr.NAMES:
        r = ref["grads"][k].numpy()
        np.testing.assert_allclose(gr[k].cpu().numpy(), r, rtol=5e-3, atol=5e-5 * np.abs(r).max(), err_msg=k)


def test_train_step_reduces_the_loss_and_repacks_weights():
    """A few optimisation steps on one synthetic batch with frozen noise: the STL loss must go down, and the kernels must
    see the updated weights after re-packing (Net.packed() notices the in-place optimiser update)."""
    from pstl_diffusion_policy_amd import nusc_train as nt
    from pstl_diffusion_policy_amd.engine import RectTrainer, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    dev = torch.device("cuda:0")
    args = nt.generate_parser(["--diffusion", "--stl_weight", "1.0", "--load_stlp", "--load_tj", "--rect_head", "--flex",
                               "--multi_cands", "5", "--diffusion_steps", "12", "--n_randoms", "64", "--sampling_size",
                               "64", "--n_neighbors", "3", "--lr", "3e-4"])
    net = nt.Net(args).cuda()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in golden_weights().items() if not k.startswith("merge_net")})
    opt = torch.optim.Adam(net.rect_net.parameters(), lr=args.lr)
    hp = net.hparams()
    scene = make_scene_batch(6, K=3, S=64, seed=5, invalid_lane_frac=0.2, stlp_mode="wide")
    sb = SceneBatch(scene, 64, hp, dev)
    params = {"rect_net." + k: p for k, p in net.rect_net.named_parameters()}
    losses = []
    for it in range(6):
        sm = Sampler(net.packed(), hp)
        tr = RectTrainer(sm)
        loss, _ = tr.train_step(sb, params, opt, args.diffusion_steps, seed=11, multi_cands=5)
        losses.append(float(loss))
    assert losses[-1] < losses[0], losses
    assert all(np.isfinite(losses))


@pytest.mark.parametrize("name", ["train_e7_step", "train_e7_step_b", "train_e7_step_c", "train_e7_noarch"])
def test_e7_diversity_train_step_matches_reference(name):
    """e7 training objective (--diverse_loss): DPP diversity loss + merge_net architecture, gradients of rect_net;
    train_e7_noarch: the same objective with --no_arch (plain rect_net input, nusc_model.py:185) and --clip_rect."""
    dev = torch.device("cuda:0")
    d = load_golden(name)
    lr = float(d["meta_f"][0])
    stl_w, div_w, scale, reg_w, detach, n_shards, no_arch, clip_rect = [float(v) for v in d["meta_e7"]]
    e7 = dict(stl_weight=stl_w, diversity_weight=div_w, diversity_scale=scale, rect_reg_loss=reg_w, detach=bool(detach))
    sm, sb, sd, tr = _setup(d, dev)
    feature, _, base_r = sm.encode(sb)
    init = torch.from_numpy(d["sel_controls"]).reshape(sb.N, 40).to(dev)
    prev = torch.from_numpy(d["sel_scores"]).to(dev)
    loss, rect, scores, g = tr.loss_and_grads(sb, feature, base_r, sd["rect_net.2.weight"], sd["rect_net.4.weight"], init,
                                              prev, e7=e7, merge=not no_arch, clip_rect=bool(clip_rect))
    np.testing.assert_allclose(rect.reshape(sb.N, 20, 2).cpu().numpy(), d["rect_controls"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(scores.cpu().numpy(), d["scores"], rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(float(tr.last["loss_diversity"]), float(d["loss_diversity"]), rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(float(tr.last["loss_reg"]), float(d["loss_reg"]), rtol=2e-4)
    np.testing.assert_allclose(float(loss), float(d["loss"]), rtol=3e-4, atol=1e-5)
    for k in tr.NAMES:
        ref = d["grad_" + k]
        np.testing.assert_allclose(g[k].cpu().numpy(), ref, rtol=5e-3, atol=3e-4 * np.abs(ref).max(), err_msg=k)


@pytest.mark.parametrize("S,n_shards,detach", [(64, 4, False), (64, 1, False), (16, 4, True), (32, 2, False)])
def test_dpp_kernel_matches_oracle(S, n_shards, detach):
    """pstl_diversity_loss alone: group diversities, d/d rect_controls and d/d scores against torch autograd on the
    oracle's restatement (float32 torch.inverse there, float64 Gauss-Jordan here)."""
    import ctypes
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd import ffi
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    dev = torch.device("cuda:0")
    hp = dict(default_hparams(), n_shards=n_shards)
    bs = 3
    N = bs * S * 3
    g = torch.Generator().manual_seed(S + n_shards)
    rect = (torch.randn(N, 20, 2, generator=g) * torch.tensor([0.1, 1.0])).requires_grad_()
    scores = (torch.randn(N, generator=g) * 0.6).requires_grad_()
    init = rect.detach() + torch.randn(N, 20, 2, generator=g) * 0.01
    ldiv, div = orc.dpp_diversity(rect, scores, bs, S, n_shards, hp, scale=1.5, detach=detach)
    lreg = orc.mask_mean(torch.square(rect - init), (scores[:, None, None] >= 0).float())
    (ldiv * 0.7 + lreg * 0.2).backward()
    cfg = ffi.make_cfg(bs, 3 * S, S, 2, 2, hp)
    G = bs * 3 * n_shards
    out = dict(div=torch.empty(G, device=dev), reg=torch.empty(2, device=dev),
               work=torch.empty(512, dtype=torch.float64, device=dev), dc=torch.empty(N, 40, device=dev),
               ds=torch.empty(N, device=dev))
    # device copies must outlive the asynchronous launch: keep them in named variables
    rect_d = rect.detach().reshape(N, 40).to(dev).contiguous()
    init_d = init.reshape(N, 40).to(dev).contiguous()
    scores_d = scores.detach().to(dev).contiguous()
    ffi.check(ffi.lib().pstl_diversity_loss(ctypes.byref(cfg), ffi.ptr(rect_d), ffi.ptr(init_d), ffi.ptr(scores_d),
                                            ctypes.c_float(1.5),
                                            ctypes.c_float(0.7), int(detach), ctypes.c_float(0.2), ffi.ptr(out["div"]),
                                            ffi.ptr(out["reg"]), ffi.ptr(out["work"], torch.float64), ffi.ptr(out["dc"]),
                                            ffi.ptr(out["ds"]), ffi.stream()), "diversity_loss")
    torch.cuda.synchronize()
    np.testing.assert_allclose(out["div"].cpu().numpy(), div.detach().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(float(out["reg"][0]), float(lreg), rtol=1e-5)
    gd = rect.grad.reshape(N, 40).numpy()
    np.testing.assert_allclose(out["dc"].cpu().numpy(), gd, rtol=2e-3, atol=2e-5 * np.abs(gd).max())
    gs = scores.grad.numpy() if scores.grad is not None else np.zeros(N, dtype=np.float32)
    np.testing.assert_allclose(out["ds"].cpu().numpy(), gs, rtol=2e-3, atol=2e-5 * max(np.abs(gs).max(), 1e-6))


# ---- --joint: gradients into the scene encoders and merge_net -----------------------------------------------------------
def _joint_setup(d, dev):
    sm, sb, sd, tr = _setup(d, dev)
    kw = dict(e7=None, merge=False, clip_rect=False)
    if "meta_e7" in d:
        stl_w, div_w, scale, reg_w, detach, n_shards, no_arch, clip_rect = [float(v) for v in d["meta_e7"]]
        kw = dict(e7=dict(stl_weight=stl_w, diversity_weight=div_w, diversity_scale=scale, rect_reg_loss=reg_w,
                          detach=bool(detach)), merge=not no_arch, clip_rect=bool(clip_rect))
    return sm, sb, sd, tr, kw


@pytest.mark.parametrize("name", ["train_e8_joint", "train_e7_joint", "train_e7_joint_b"])
def test_joint_gradients_match_reference(name):
    """--joint (reference nusc_train.py:1230-1231): d loss / d (three encoders, merge_net, rect_net) against the reference's
    autograd; the saved-token encoder is the plain encoder bit for bit."""
    dev = torch.device("cuda:0")
    d = load_golden(name)
    sm, sb, sd, tr, kw = _joint_setup(d, dev)
    feature, base_p, base_r, saved = sm.encode(sb, save=True)
    f0, p0, r0 = sm.encode(sb)
    assert torch.equal(feature, f0) and torch.equal(base_p, p0) and torch.equal(base_r, r0)
    T = sb.bs * (sb.K + 4)
    assert saved["tok_in"].shape == (T, 48) and saved["tok_out"].shape == (T, 32)
    # the ego token's output is the first 32 feature columns, the lanes' the last 96
    assert torch.equal(saved["tok_out"][:sb.bs], feature[:, :32])
    assert torch.equal(saved["tok_out"][sb.bs * (sb.K + 1):].reshape(sb.bs, 96), feature[:, 128:])
    init = torch.from_numpy(d["sel_controls"]).reshape(sb.N, 40).to(dev)
    prev = torch.from_numpy(d["sel_scores"]).to(dev)
    names = tr.joint_names(kw["merge"])
    assert set(names) == set(str(k) for k in d["joint_names"]) | set(tr.NAMES)
    loss, rect, scores, g = tr.loss_and_grads(sb, feature, base_r, sd["rect_net.2.weight"], sd["rect_net.4.weight"], init,
                                              prev, joint=dict(params=sd, saved=saved), **kw)
    np.testing.assert_allclose(float(loss), float(d["loss"]), rtol=3e-4, atol=1e-5)
    assert set(g) == set(names)
    for k in names:
        ref = d["grad_" + k]
        assert g[k].shape == ref.shape, k
        np.testing.assert_allclose(g[k].cpu().numpy(), ref, rtol=5e-3, atol=3e-4 * np.abs(ref).max(), err_msg=k)


def test_joint_gradients_against_oracle_on_fresh_scenes():
    """Bigger than the fixtures (several workgroups per kernel, invalid lanes, K = 6), e7 architecture."""
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.engine import PackedWeights, RectTrainer, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    bs, S, K = 37, 32, 6
    scene = make_scene_batch(bs, K=K, S=S, seed=91, invalid_lane_frac=0.3, stlp_mode="wide")
    sdn = golden_weights()
    gen = torch.Generator().manual_seed(4)
    N = bs * S * 3
    init = (torch.randn(N, 20, 2, generator=gen) * torch.tensor([0.1, 1.0])).clamp(-0.5, 0.5)
    prev = torch.randn(N, generator=gen)
    e7 = dict(stl_weight=1.0, diversity_weight=0.5, diversity_scale=1.0, rect_reg_loss=0.0, detach=False)
    ref = orc.rect_train_step(sdn, {k: v.numpy() for k, v in scene.items()}, S, hp, init, prev, 3e-4, e7=e7, joint=True)
    sd = {k: torch.from_numpy(v).to(dev) for k, v in sdn.items()}
    sm = Sampler(PackedWeights(sd, dev), hp)
    sb = SceneBatch(scene, S, hp, dev)
    tr = RectTrainer(sm)
    feature, _, base_r, saved = sm.encode(sb, save=True)
    loss, rect, scores, g = tr.loss_and_grads(sb, feature, base_r, sd["rect_net.2.weight"], sd["rect_net.4.weight"],
                                              init.reshape(N, 40).to(dev), prev.to(dev), e7=e7,
                                              joint=dict(params=sd, saved=saved))
    np.testing.assert_allclose(float(loss), float(ref["loss"]), rtol=3e-4, atol=1e-5)
    for k in tr.joint_names(True):
        r = ref["grads"][k].numpy()
        np.testing.assert_allclose(g[k].cpu().numpy(), r, rtol=5e-3, atol=3e-4 * np.abs(r).max(), err_msg=k)


def test_joint_train_steps_move_every_trained_tensor_and_nothing_else():
    """The CLI's --joint path: Adam over net.parameters(); encoders, merge_net and rect_net move, policy_net does not, the
    loss goes down on a frozen batch."""
    from pstl_diffusion_policy_amd import nusc_train as nt
    from pstl_diffusion_policy_amd.engine import RectTrainer, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    dev = torch.device("cuda:0")
    args = nt.generate_parser(["--diffusion", "--stl_weight", "1.0", "--load_stlp", "--rect_head", "--flex", "--diverse_loss",
                               "--diversity_weight", "0.1", "--multi_cands", "5", "--diffusion_steps", "12", "--n_randoms",
                               "64", "--sampling_size", "64", "--n_neighbors", "3", "--lr", "3e-4", "--joint"])
    net = nt.Net(args).cuda()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in golden_weights().items()})
    opt = torch.optim.Adam(net.parameters(), lr=args.lr)
    hp = net.hparams()
    scene = make_scene_batch(6, K=3, S=64, seed=5, invalid_lane_frac=0.2, stlp_mode="wide")
    sb = SceneBatch(scene, 64, hp, dev)
    params = dict(net.named_parameters())
    before = {k: v.detach().clone() for k, v in params.items()}
    e7 = dict(stl_weight=1.0, diversity_weight=0.1, diversity_scale=1.0, rect_reg_loss=0.0, detach=False)
    losses = []
    for it in range(5):
        tr = RectTrainer(Sampler(net.packed(), hp))
        loss, _ = tr.train_step(sb, params, opt, args.diffusion_steps, seed=11, multi_cands=5, e7=e7, joint=True)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    for k, v in params.items():
        moved = not torch.equal(v.detach(), before[k])
        assert moved == (not k.startswith("policy_net.")), k


def test_one_pass_backward_agrees_with_the_exact_fp32_launches():
    """Round 4: k_bwd_l2 / k_bwd_l1 (one pass per layer, split-bf16 weight gradients, per-scene sums in the epilogue) against the
    unfused launches with fp32-MFMA weight gradients that `chain_waves = 8` keeps (the exact request): same inputs, every
    gradient within 2e-3 of its tensor's maximum -- on a batch whose scene count does not divide evenly over the workgroups."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, RectTrainer, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    bs, S, K = 37, 64, 3
    scene = make_scene_batch(bs, K=K, S=S, seed=91, invalid_lane_frac=0.3, stlp_mode="wide")
    sd = {k: torch.from_numpy(v).to(dev) for k, v in golden_weights().items()}
    g = torch.Generator().manual_seed(6)
    N = bs * S * 3
    init = ((torch.randn(N, 20, 2, generator=g) * torch.tensor([0.1, 1.0])).clamp(-0.5, 0.5)).reshape(N, 40).to(dev)
    prev = torch.randn(N, generator=g).to(dev)
    got = {}
    for cw in (0, 8):
        sm = Sampler(PackedWeights(sd, dev), hp, chain_waves=cw)
        sb = SceneBatch(scene, S, hp, dev)
        feature, _, base_r = sm.encode(sb)
        loss, rect, scores, gr = RectTrainer(sm).loss_and_grads(sb, feature, base_r, sd["rect_net.2.weight"],
                                                                sd["rect_net.4.weight"], init, prev, merge=True)
        got[cw] = (float(loss), {k: v.cpu().numpy() for k, v in gr.items()})
    assert abs(got[0][0] - got[8][0]) <= 2e-4 * abs(got[8][0])
    for k in RectTrainer.NAMES:
        a, b = got[0][1][k], got[8][1][k]
        assert np.isfinite(a).all()
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-3 * np.abs(b).max(), err_msg=k)


def test_repacking_one_network_in_place_equals_a_full_pack():
    """pstl_repack_weights / PackedWeights.update: after rect_net (and, second, policy_net + merge_net) changed, the buffer packed
    again in place holds, word for word, what a full pack of the new state_dict holds -- the recomputed max |w| status words
    included --, and the sticky domain word is left alone."""
    from pstl_diffusion_policy_amd.engine import PackedWeights
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(v).to(dev) for k, v in golden_weights().items()}
    pw = PackedWeights(sd, dev)
    pw.status[2:3].view(torch.int32).fill_(1)                    # pretend a launch flagged the domain
    g = torch.Generator(device=dev).manual_seed(3)
    for nets in (("rect_net",), ("policy_net", "merge_net")):
        for k in sd:
            if k.split(".")[0] in nets:
                sd[k] = sd[k] + 0.01 * torch.randn(sd[k].shape, device=dev, generator=g)
        sd["rect_net.2.weight"][3, 5] = 7.5                       # a new maximum for the status word
        pw.update({k: v for k, v in sd.items() if k.split(".")[0] in nets})
        full = PackedWeights(sd, dev)
        so = pw.status.storage_offset()
        a, b = pw.packed.clone(), full.packed.clone()
        assert a[so + 2:so + 3].view(torch.int32).item() == 1 and b[so + 2:so + 3].view(torch.int32).item() == 0
        a[so + 2] = 0.0
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), nets
        assert pw.chain_wmax == full.chain_wmax and pw.chain_wmax["rect_net"] == 7.5
    with pytest.raises(ValueError):
        pw.update({"rect_net.0.weight": sd["rect_net.0.weight"]})  # a network is packed whole


@pytest.mark.parametrize("lr,betas,eps,table,n_steps", [(3e-4, (0.9, 0.999), 1e-8, None, 10), (0.01, (0.8, 0.99), 1e-6, 4, 10),
                                                        (0.02, (0.1, 0.2), 1e-8, None, 20)])
def test_device_adam_equals_torch_adam(lr, betas, eps, table, n_steps, monkeypatch):
    """pstl_adam_step against torch.optim.Adam on the CPU (the reference's optimiser, nusc_train.py:1233) over ten steps on the
    shapes of rect_net's six tensors: both moments bit for bit at every step -- the parameters too wherever torch's vectorised
    CPU square root is the IEEE one (tests/test_adam_core_hostsim.py says why not everywhere), at most three ulps of the increment
    otherwise --, the version counters bumped (what PackedWeights caches key on), the step counter on the device.  table = 4: a
    table of four steps, so that the in-place refresh of the per-step scalars (and the counter's reset) happens twice in the run;
    that run also changes the learning rate half-way, as a scheduler would (the table is rebuilt for the new rate).  The third
    case runs PAST its table: with betas (0.1, 0.2) the per-step scalars reach their limits after 13 steps, the table ends there
    (DeviceAdam.table_len), and steps 14-20 read its last entry -- still torch's update (what lets a captured step be replayed
    any number of times with no host involvement)."""
    from pstl_diffusion_policy_amd.engine import DeviceAdam
    if table:
        monkeypatch.setattr(DeviceAdam, "TABLE", table)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    shapes = [(256, 271), (256,), (256, 256), (256,), (40, 256), (40,)]
    cpu = [(torch.randn(s, generator=g) * 0.05).requires_grad_() for s in shapes]
    gpu = [p.detach().clone().to(dev).requires_grad_() for p in cpu]
    opt = torch.optim.Adam(cpu, lr=lr, betas=betas, eps=eps)
    dopt = DeviceAdam(gpu, lr=lr, betas=betas, eps=eps)
    v0 = [p._version for p in gpu]
    for t in range(n_steps):
        if table and t == 5:
            lr = lr * 0.5
            opt.param_groups[0]["lr"] = lr
            dopt.lr = lr
        grads = [torch.randn(s, generator=g) * (10.0 ** float(torch.randint(-8, 1, (1,), generator=g))) for s in shapes]
        grads[0][::5] = 0.0
        for p, gr in zip(cpu, grads):
            p.grad = gr.clone()
        prev = [p.detach().clone() for p in cpu]
        opt.step()
        dopt.step([gr.to(dev) for gr in grads])
        o = 0
        for i, (pc, pg) in enumerate(zip(cpu, gpu)):
            n = pc.numel()
            st = opt.state[pc]
            m = dopt.exp_avg[o:o + n].cpu().numpy().view(np.uint32)
            v = dopt.exp_avg_sq[o:o + n].cpu().numpy()
            assert np.array_equal(m, st["exp_avg"].numpy().reshape(-1).view(np.uint32)), (t, i)
            assert np.array_equal(v.view(np.uint32), st["exp_avg_sq"].numpy().reshape(-1).view(np.uint32)), (t, i)
            want, got = pc.detach().numpy().reshape(-1), pg.detach().cpu().numpy().reshape(-1)
            ieee = torch.sqrt(st["exp_avg_sq"]).numpy().reshape(-1).view(np.uint32) == np.sqrt(v).view(np.uint32)
            assert np.array_equal(got.view(np.uint32)[ieee], want.view(np.uint32)[ieee]), (t, i)
            step = np.abs(want - prev[i].numpy().reshape(-1))
            assert (np.abs(got - want) <= np.spacing(np.abs(want)) + 4e-7 * step).all(), (t, i)
            with torch.no_grad():
                pg.copy_(pc.detach())           # (every step is checked on its own)
            o += n
    assert dopt.steps_done == n_steps and int(dopt.step_dev.item()) == n_steps - dopt._table_first + 1
    if n_steps > 10:
        assert dopt.table_steps == 13 and dopt._table_final and dopt._table_first == 1
    assert all(p._version > v for p, v in zip(gpu, v0))


def test_training_step_replays_as_one_hip_graph():
    """with the optimiser on the device path (pstl_adam_step: per-step scalars from a device table indexed by
    a device counter) and the in-place re-pack of rect_net (pstl_repack_weights: kernels only), ONE captured HIP graph is a whole
    training step of config 5 -- sampling under in-kernel noise (seed through a pstl_dyn block), RefineNet forward / STL adjoint /
    backward, Adam, re-pack.  Three replays with three seeds must leave the weights that three eager steps leave, bit for bit."""
    from pstl_diffusion_policy_amd.engine import (DeviceAdam, DynBlock, GraphCapture, PackedWeights, RectTrainer, Sampler, SceneBatch)
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    bs, S, K, steps = 24, 64, 2, 12
    scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=K, S=S, seed=21, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    sd0 = init_state_dict(1007)
    seeds = [101, 202, 303]

    def fresh():
        sd = {k: v.to(dev).clone() for k, v in sd0.items()}
        params = {k: sd[k].requires_grad_() for k in RectTrainer.NAMES}
        pw = PackedWeights(sd, dev)
        opt = DeviceAdam([params[k] for k in RectTrainer.NAMES], lr=3e-4)
        return sd, params, pw, opt

    # eager: three steps, re-packing rect_net after each
    sd, params, pw, opt = fresh()
    for s in seeds:
        sb = SceneBatch(scene, S, hp, dev)
        RectTrainer(Sampler(pw, hp)).train_step(sb, params, opt, steps, seed=s, multi_cands=3)
        pw.update({k: v for k, v in sd.items() if k.startswith("rect_net.")}, read_status=False)
    want = {k: params[k].detach().clone() for k in RectTrainer.NAMES}
    assert opt.steps_done == 3

    # one captured graph, replayed with the seed in device memory
    sd, params, pw, opt = fresh()
    dyn = DynBlock(dev)
    vsum = float(sum(scene[k].sum().item() for k in ("curr_id", "left_id", "right_id"))) * S
    dyn.set(seeds[0], SceneBatch.loss_scale(vsum, bs * S * 3))
    sm = Sampler(pw, hp)

    def body():
        sb = SceneBatch(scene, S, hp, dev, dyn=dyn.dev, scale_in_dyn=True)
        sb.grad_scale = dyn.grad_scale            # (the training loss takes its scale by value; the guidance scale sits in dyn)
        loss, scores = RectTrainer(sm).train_step(sb, params, opt, steps, seed=0, multi_cands=3, domain_check="deferred")
        pw.update({k: v for k, v in sd.items() if k.startswith("rect_net.")}, read_status=False)
        return loss

    # GraphCapture's warm-up runs the body eagerly: undo what those steps did to the weights and the optimiser before capturing
    snap = {k: params[k].detach().clone() for k in RectTrainer.NAMES}
    g = GraphCapture(body, warmup=1)
    with torch.no_grad():
        for k in RectTrainer.NAMES:
            params[k].copy_(snap[k])
    opt.exp_avg.zero_(), opt.exp_avg_sq.zero_(), opt.step_dev.zero_()
    opt.steps_done = 0
    pw.update({k: v for k, v in sd.items() if k.startswith("rect_net.")}, read_status=False)
    for s in seeds:
        dyn.set(s)
        g.replay()
        opt.note_replays(1)
    torch.cuda.synchronize()
    assert not sm.check_chain_domain(fallback=False)
    assert int(opt.step_dev.item()) == 3
    for k in RectTrainer.NAMES:
        assert torch.equal(params[k].detach(), want[k]), k
