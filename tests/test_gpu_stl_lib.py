"""GPU parity of the generic STL formula evaluator (pstl_diffusion_policy_amd.stl_d_lib -> pstl_stl_program_*):
values and gradients against (1) golden vectors from the reference's own stl_d_lib and (2) the CPU oracle on fresh
inputs.  Tolerances: robustness 2e-6 abs + 2e-6 rel (float32 logsumexp), gradients 2e-5 abs + 1e-4 rel."""
import os

import numpy as np
import pytest
import torch

import stl_specs

pytestmark = pytest.mark.gpu
G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "stl_lib.npz")))
KEYS = sorted(k[:-2] for k in G if k.endswith("|y"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    from pstl_diffusion_policy_amd import ffi
    ffi.lib()
    return torch.device("cuda:0")


def _run(spec, x, W, tau, hard):
    from pstl_diffusion_policy_amd import stl_d_lib
    f = stl_specs.build(spec, stl_d_lib)
    x = x.clone().requires_grad_()
    y = f(x, tau, {"hard": True} if hard else None)
    (torch.where(torch.isfinite(y), y, torch.zeros_like(y)) * W).sum().backward()
    return y.detach().cpu().numpy(), torch.nan_to_num(x.grad, nan=0.0).cpu().numpy()


@pytest.mark.parametrize("key", KEYS)
def test_formula_matches_reference_golden(dev, key):
    name, T, tau, mode = key.split("|")
    y, g = _run(stl_specs.SPECS[name], torch.from_numpy(G["signals_" + T]).to(dev), torch.from_numpy(G["W_" + T]).to(dev),
                float(tau[3:]), mode == "hard")
    want = G[key + "|y"]
    fin = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(y), fin)
    np.testing.assert_allclose(y[fin], want[fin], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(g, G[key + "|g"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", sorted(stl_specs.SPECS))
def test_formula_matches_oracle_fresh_inputs(dev, name):
    from oracle import stl_lib_oracle as so
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)    # hash() is salted per process: not reproducible
    n, T = 300, 17
    x = torch.randn(4, n, T, generator=g) * 1.5
    W = torch.randn(n, T, generator=g)
    for tau, hard in ((100.0, False), (10.0, False), (100.0, True)):
        xo = x.clone().requires_grad_()
        yo = so.evaluate(stl_specs.SPECS[name], xo, tau, hard)
        (torch.where(torch.isfinite(yo), yo, torch.zeros_like(yo)) * W).sum().backward()
        y, gr = _run(stl_specs.SPECS[name], x.to(dev), W.to(dev), tau, hard)
        fin = np.isfinite(yo.detach().numpy())
        np.testing.assert_array_equal(np.isfinite(y), fin)
        np.testing.assert_allclose(y[fin], yo.detach().numpy()[fin], rtol=2e-6, atol=2e-6)
        # tau*x reaches a few hundred here, where float32 spacing is ~3e-5: every softmax weight exp(tau*(x - max)) carries
        # that much relative noise per nesting level in ANY float32 implementation, hence the wider gate than on the goldens
        np.testing.assert_allclose(gr, torch.nan_to_num(xo.grad, nan=0.0).numpy(), rtol=1e-3, atol=5e-5)


def test_listand_full_returns_children(dev):
    from pstl_diffusion_policy_amd import stl_d_lib
    f = stl_specs.build(stl_specs.SPECS["listand_path"], stl_d_lib)
    s, v = f(torch.from_numpy(G["signals_T20"]).to(dev), 100.0, None, full=True)
    np.testing.assert_allclose(s.cpu().numpy(), G["listand_full_s"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(v.cpu().numpy(), G["listand_full_v"], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("hard", [False, True])
def test_module_level_soft_helpers_run_on_the_kernel(dev, hard):
    """softmax / softmin / softmax_pairs / softmin_pairs of the reference module (stl_d_lib.py:6-26): evaluated by the STL
    program kernel, values and gradients against the reference's torch expressions."""
    from pstl_diffusion_policy_amd import stl_d_lib as sl
    g = torch.Generator().manual_seed(3)
    tau, d = 7.0, {"hard": hard}
    x = (torch.randn(37, 11, generator=g) * 0.5).to(dev).requires_grad_()
    y = (torch.randn(37, 11, generator=g) * 0.5).to(dev).requires_grad_()

    def ref_max(v, dim=1):
        return torch.max(v, dim=dim, keepdim=True)[0] if hard else torch.logsumexp(v * tau, dim=dim, keepdim=True) / tau

    for mine, ref in [(sl.softmax(x, tau, d), ref_max(x)), (sl.softmin(x, tau, d), -ref_max(-x)),
                      (sl.softmax_pairs(x, y, tau, d), ref_max(torch.stack([x, y], dim=1)).squeeze(1)),
                      (sl.softmin_pairs(x, y, tau, d), -ref_max(torch.stack([-x, -y], dim=1)).squeeze(1))]:
        assert mine.shape == ref.shape
        np.testing.assert_allclose(mine.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=0, atol=2e-6)
        gm = torch.autograd.grad(mine.sum(), [x, y], allow_unused=True, retain_graph=True)
        gr = torch.autograd.grad(ref.sum(), [x, y], allow_unused=True, retain_graph=True)
        for a, b in zip(gm, gr):
            if b is not None:
                np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-6)
    assert torch.isinf(sl.softmax(x[:, :0], tau, d)).all() and sl.softmax(x[:, :0], tau, d).shape == (37, 1)


def test_cpu_tensors_are_rejected(dev):
    from pstl_diffusion_policy_amd import stl_d_lib
    f = stl_specs.build(stl_specs.SPECS["alw_ap"], stl_d_lib)
    with pytest.raises(RuntimeError):
        f(torch.zeros(4, 3, 20), 100.0)


@pytest.mark.parametrize("name", ["stl_mixed", "stl_mixed_k8", "stl_wild"])
def test_build_stl_cache_formulas_match_fused_kernel_and_reference(dev, name):
    """The three formulas build_stl_cache returns are real, callable stl_d_lib objects (as in the reference): fed with
    the reference's own signal dict (golden sig_*), they reproduce the reference's scores and the fused kernel's."""
    from conftest import load_golden
    from pstl_diffusion_policy_amd import nusc_train as nt
    d = load_golden(name)
    args = nt.generate_parser(["--diffusion", "--load_stlp"])
    stls = nt.build_stl_cache(args)
    x = {"ego_traj": torch.from_numpy(d["trajs"][:, :-1]).to(dev), "stlp": torch.from_numpy(d["in_stlp_dense"]).to(dev)}
    for k in ("x2curr_d", "x2curr_th", "x2left_d", "x2left_th", "x2right_d", "x2right_th", "min_nei_d"):
        x[k] = torch.from_numpy(d["sig_" + k]).to(dev)
    for m in range(3):
        s = stls[m](x, args.smoothing_factor)[:, 0].cpu().numpy()
        np.testing.assert_allclose(s, d["scores3"][m], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("name", ["stl_mixed", "stl_mixed_k8", "stl_wild", "stl_norm"])
def test_prep_stl_cache_signals_match_reference(dev, name):
    """prep_stl_cache (pstl_stl_signals): the seven (R,T) signals against the reference's compute_t2l_dist /
    compute_shortest_dist_refined outputs, from the dense per-row layout AND the scene-indexed side channel."""
    from conftest import load_golden, scene_from_golden
    from pstl_diffusion_policy_amd import nusc_train as nt
    from pstl_diffusion_policy_amd.engine import SceneBatch
    d = load_golden(name)
    bs, S, K, seed = [int(v) for v in d["meta"]]
    args = nt.generate_parser(["--diffusion", "--load_stlp", "--n_randoms", str(S), "--sampling_size", str(S)])
    traj = torch.from_numpy(d["trajs"][:, :-1]).to(dev)
    m = 3 * S
    dense = {"ego_traj": traj, "stlp": torch.from_numpy(d["in_stlp_dense"]).to(dev),
             "neighbors": nt.dup(torch.from_numpy(d["in_neighbors_traj"]), m).to(dev)}
    for k in ("curr", "left", "right"):
        dense["%slane_wpts" % k] = nt.dup(torch.from_numpy(d["in_%slane_wpts" % k]), m).to(dev)
    sb = SceneBatch({k: torch.from_numpy(v) for k, v in scene_from_golden(d).items()}, S, nt._hp(args), dev)
    side = {"ego_traj": traj, "stlp": dense["stlp"], "_pstl": sb}
    for x in (nt.prep_stl_cache(dense, args), nt.prep_stl_cache(side, args)):
        for k in nt.SIGNAL_KEYS:
            got, want = x[k].cpu().numpy(), d["sig_" + k]
            tol = 2e-4 if k.endswith("_d") else 2e-6     # distances ~1e2 m computed from world coordinates ~1e3
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=tol, err_msg=k)


@pytest.mark.parametrize("generic", [False, True])
def test_norm_stl_variant_matches_reference(dev, generic):
    """--norm_stl (nusc_train.py:88-91,97-113) through compute_stl_dense: the fused kernel (PSTL_FLAG_NORM_STL), and -- generic
    -- signals by pstl_stl_signals + the formula objects on the generic program kernel; both against the reference, the
    fused scores with exact satisfaction masks, and its adjoint against the reference's autograd."""
    from conftest import load_golden, scene_from_golden
    from pstl_diffusion_policy_amd import nusc_train as nt
    from pstl_diffusion_policy_amd.engine import SceneBatch
    d = load_golden("stl_norm")
    bs, S, K, seed = [int(v) for v in d["meta"]]
    args = nt.generate_parser(["--diffusion", "--load_stlp", "--norm_stl", "--n_randoms", str(S), "--sampling_size", str(S)])
    stls = nt.build_stl_cache(args)
    sb = SceneBatch({k: torch.from_numpy(v) for k, v in scene_from_golden(d).items()}, S, nt._hp(args), dev)
    x = {"ego_traj": torch.from_numpy(d["trajs"][:, :-1]).to(dev), "stlp": torch.from_numpy(d["in_stlp_dense"]).to(dev),
         "_pstl": sb}
    hl = torch.from_numpy(d["in_highlevel_dense"]).to(dev)
    valid = torch.from_numpy(d["in_valids_dense"]).to(dev)
    args.generic_stl = generic
    scores_list, scores, acc, scene_acc = nt.compute_stl_dense(x, stls, hl, valid, args, scene=True)
    np.testing.assert_allclose(torch.stack(scores_list[:3]).cpu().numpy(), d["scores3"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(scores.cpu().numpy(), d["scores"], rtol=1e-4, atol=2e-4)
    assert float(acc) == pytest.approx(float(d["acc"]), abs=1e-6) and float(scene_acc) == pytest.approx(float(d["scene_acc"]), abs=1e-6)
    if not generic:
        from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler
        from conftest import golden_weights
        np.testing.assert_array_equal(scores.cpu().numpy() > 0, d["scores"] > 0)
        sm = Sampler(PackedWeights(golden_weights(), dev), nt._hp(args))
        c = torch.from_numpy(d["controls"]).reshape(sb.N, 40).to(dev)
        sc, g = sm.score_grad(sb, c)
        np.testing.assert_allclose(sc.cpu().numpy(), d["scores"], rtol=1e-4, atol=2e-4)
        ref = d["grad_sum"].reshape(-1, 40)
        scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-20
        np.testing.assert_allclose(g.cpu().numpy() / scale, ref / scale, rtol=5e-3, atol=5e-4)
        # the two many-iteration loops are built for the default formulas and say so
        with pytest.raises(RuntimeError, match="shape"):
            sm.trajopt(sb, c.clone(), 2, 0.005)
