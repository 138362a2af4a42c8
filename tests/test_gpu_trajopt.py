"""GPU parity of the traj-opt kernel (pstl_trajopt: all Adam iterations of the data-augmentation loop in one launch)
against the reference's own loop (tests/golden/trajopt_*.npz) and the CPU oracle on fresh inputs.
Gates: controls after k iterations 2e-5 abs for k <= 3; after the full run 2e-4 abs on 99.5 % of the elements and
2*lr*k everywhere (Adam's normalised step lr*m/(sqrt(v)+eps) is discontinuous where a gradient vanishes -- a hinge
switching on/off -- so two float32 implementations can differ by a full step on isolated elements); scores of the last
iterate 2e-3 abs on 97 % of the rows."""
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


def _setup(dev, name):
    from test_oracle_trajopt_golden import load_case
    from pstl_diffusion_policy_amd.engine import Sampler, SceneBatch
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    from pstl_diffusion_policy_amd import ffi
    g, scene, bs, S, iters = load_case(name)
    sb = SceneBatch({k: torch.from_numpy(np.asarray(v)) for k, v in scene.items()}, S, default_hparams(), dev)
    sm = Sampler.__new__(Sampler)
    sm.L = ffi.lib()
    return g, scene, sb, sm, bs, S, iters


def _close_mostly(got, want, atol, frac, hard_cap, rtol=0.0):
    d = np.abs(got - want)
    ok = d <= atol + rtol * np.abs(want)
    assert (d <= hard_cap + rtol * np.abs(want)).all(), d.max()
    assert np.mean(ok) >= frac, (np.mean(ok), d.max())


@pytest.mark.parametrize("name", ["trajopt_a", "trajopt_b"])
def test_trajopt_matches_reference_golden(dev, name):
    g, scene, sb, sm, bs, S, iters = _setup(dev, name)
    lr, thres, reg = [float(v) for v in g["meta_f"]]
    N = bs * S * 3
    init = torch.from_numpy(g["params_init"]).reshape(N, 40)
    for k in (1, 3, iters):
        p = init.clone().to(dev)
        scores, _ = sm.trajopt(sb, p, k, lr, thres, reg)
        torch.cuda.synchronize()
        got, want = p.cpu().numpy(), g["params_after%d" % k].reshape(N, 40)
        if k <= 3:
            _close_mostly(got, want, 2e-5, 0.999, 2 * lr * k)
        else:
            _close_mostly(got, want, 2e-4, 0.995, 2 * lr * k)
    # rows whose controls took an isolated different step (see above) carry that into their score
    _close_mostly(scores.cpu().numpy().reshape(bs * S, 3), g["scores_last"], 2e-3, 0.97, 0.5, rtol=1e-3)
    # a run split over two calls (Adam state handed back) is the same run, bit for bit
    p1 = init.clone().to(dev)
    _, work = sm.trajopt(sb, p1, 3, lr, thres, reg)
    sc2, _ = sm.trajopt(sb, p1, iters - 3, lr, thres, reg, work=work, first_iter=3)
    assert torch.equal(p1, p) and torch.equal(sc2, scores)


def test_trajopt_matches_oracle_and_improves(dev):
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.engine import SceneBatch
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    g, scene0, sb0, sm, *_ = _setup(dev, "trajopt_a")
    hp = default_hparams()
    bs, S, K, iters, lr = 4, 64, 3, 6, 0.02
    scene = make_scene_batch(bs, K=K, S=S, seed=71, invalid_lane_frac=0.3, stlp_mode="wide")
    N = bs * S * 3
    params = scene["params"].reshape(N, 20, 2) * 2.0
    rows = orc.Rows({k: v.numpy() for k, v in scene.items()}, S, hp)
    r = orc.trajopt(rows, params, iters, lr, 0.01, 10.0)
    sb = SceneBatch(scene, S, hp, dev)
    p = params.reshape(N, 40).clone().to(dev)
    scores, _ = sm.trajopt(sb, p, iters, lr, 0.01, 10.0)
    torch.cuda.synchronize()
    _close_mostly(p.cpu().numpy(), r["params"].reshape(N, 40).numpy(), 2e-4, 0.995, 2 * lr * iters)
    _close_mostly(scores.cpu().numpy(), r["scores_last"].numpy(), 3e-3, 0.97, 0.5, rtol=1e-3)
    # and a long run does what the loop is for: the satisfaction rate of valid rows goes up
    before = sm.score(sb, params.reshape(1, N, 40).to(dev).contiguous())["scores"][0] if hasattr(sm, "score") else None
    p2 = params.reshape(N, 40).clone().to(dev)
    sc_long, _ = sm.trajopt(sb, p2, 300, 0.005, 0.01, 10.0)
    valid = sb.valid > 0
    assert ((sc_long > 0) & valid).sum().item() > ((before > 0) & valid).sum().item()
