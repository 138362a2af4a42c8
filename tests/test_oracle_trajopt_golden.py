"""The traj-opt restatement of the oracle against the reference's own loop (tests/golden/trajopt_*.npz: the reference's
generate_trajs + compute_trajopt_loss_lite + torch.optim.Adam, make_golden.py --trajopt)."""
import os

import numpy as np
import pytest
import torch

from oracle import pstl_oracle as orc
from pstl_diffusion_policy_amd.synthetic import default_hparams

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_case(name):
    g = dict(np.load(os.path.join(GOLD, name + ".npz")))
    bs, S, K, seed, iters = [int(v) for v in g["meta"]]
    scene = {k[3:]: g[k] for k in g if k.startswith("in_") and k != "in_stlp_dense"}
    scene["stlp_rows"] = g["in_stlp_dense"][:, 0]
    return g, scene, bs, S, iters


@pytest.mark.parametrize("name", ["trajopt_a", "trajopt_b"])
def test_trajopt_oracle_matches_reference(name):
    g, scene, bs, S, iters = load_case(name)
    lr, thres, reg = [float(v) for v in g["meta_f"]]
    rows = orc.Rows(scene, S, default_hparams())
    N = bs * S * 3
    r = orc.trajopt(rows, torch.from_numpy(g["params_init"]).reshape(N, 20, 2), iters, lr, thres, reg,
                    checkpoints=(1, 3, iters))
    np.testing.assert_allclose(r["grad0"].numpy().reshape(g["grad_iter0"].shape), g["grad_iter0"], rtol=2e-4, atol=1e-7)
    for k in (1, 3, iters):
        np.testing.assert_allclose(r["checkpoints"][k].numpy().reshape(g["params_init"].shape), g["params_after%d" % k],
                                   rtol=0, atol=2e-5 if k < iters else 2e-4)
    np.testing.assert_allclose(np.asarray(r["losses"]), g["losses"], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(r["scores_last"].numpy().reshape(bs * S, 3), g["scores_last"], rtol=1e-3, atol=2e-3)
