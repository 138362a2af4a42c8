"""The diversity / ADE-FDE oracle (oracle/diversity_oracle.py) against golden vectors produced by the reference's own
measure_diversity / measure_extra_diversity / compute_ade_fde (tests/golden/make_golden.py --diversity)."""
import os

import numpy as np
import pytest
import torch

from oracle import diversity_oracle as dorc
from oracle import pstl_oracle as orc
from pstl_diffusion_policy_amd.synthetic import default_hparams

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["div_mixed", "div_s64_clip", "div_sparse"]


def run_oracle(g):
    hp = default_hparams()
    S = int(g["S"])
    ego = torch.from_numpy(g["in_ego_traj"])
    bs = ego.shape[0]
    valid = torch.from_numpy(g["in_valids"])[:, None, :].repeat(1, S, 1).reshape(bs * S * 3)
    return dorc.all_metrics(ego[:, 0, :4], ego, torch.from_numpy(g["in_controls"]), torch.from_numpy(g["in_scores"]),
                            valid, S, hp, orc.unicycle_rollout)


@pytest.mark.parametrize("name", CASES)
def test_diversity_oracle_matches_reference(name):
    g = dict(np.load(os.path.join(GOLD, name + ".npz")))
    o = run_oracle(g)
    assert o["std"] == pytest.approx(float(g["std"]), rel=2e-5)
    assert o["vol"] == pytest.approx(float(g["vol"]), rel=1e-6)
    assert o["ade"] == pytest.approx(float(g["ade"]), rel=1e-6)
    assert o["fde"] == pytest.approx(float(g["fde"]), rel=1e-6)
    for k in ("ent_s", "ent_w", "ent_a", "ent_wa", "area"):
        assert o[k] == pytest.approx(float(g[k]), rel=1e-6), k
    val = g["in_valids"] > 0
    for m in range(3):   # the reference's per-mode arrays (filled with 0 where the lane is invalid)
        np.testing.assert_allclose(np.where(val[:, m], o["std_sm"][:, m], 0), g["std%d" % m], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(np.where(val[:, m], o["vol_sm"][:, m], 0), g["vol%d" % m], rtol=1e-6, atol=1e-9)


def test_hull_area_edge_cases():
    assert dorc.hull_area(np.zeros((5, 2))) == 0.0
    assert dorc.hull_area(np.array([[0, 0], [1, 1], [2, 2], [3, 3.0]])) == 0.0
    assert dorc.hull_area(np.array([[0, 0], [1, 0]])) == 0.0
    assert dorc.hull_area(np.array([[0, 0], [2, 0], [2, 2], [0, 2], [1, 1], [2, 1.0]])) == pytest.approx(4.0)
    scipy_spatial = pytest.importorskip("scipy.spatial")
    rng = np.random.default_rng(3)
    for n in (3, 7, 64):
        p = rng.normal(size=(n, 2)) * [3.0, 0.2] + [500.0, -900.0]
        assert dorc.hull_area(p) == pytest.approx(scipy_spatial.ConvexHull(p).volume, rel=1e-9)
