"""oracle/stl_lib_oracle.py against the values and autograd gradients of the reference's own stl_d_lib classes
(tests/golden/stl_lib.npz, written by tests/golden/make_golden.py --stl-lib)."""
import os

import numpy as np
import pytest
import torch

import stl_specs
from oracle import stl_lib_oracle as so

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "stl_lib.npz")))
KEYS = sorted(k[:-2] for k in G if k.endswith("|y"))


@pytest.mark.parametrize("key", KEYS)
def test_oracle_matches_reference(key):
    name, T, tau, mode = key.split("|")
    x = torch.from_numpy(G["signals_" + T]).clone().requires_grad_()
    W = torch.from_numpy(G["W_" + T])
    y = so.evaluate(stl_specs.SPECS[name], x, float(tau[3:]), hard=(mode == "hard"))
    want = G[key + "|y"]
    fin = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(y.detach().numpy()), fin)
    np.testing.assert_allclose(y.detach().numpy()[fin], want[fin], rtol=1e-6, atol=1e-6)
    (torch.where(torch.isfinite(y), y, torch.zeros_like(y)) * W).sum().backward()
    np.testing.assert_allclose(torch.nan_to_num(x.grad, nan=0.0).numpy(), G[key + "|g"], rtol=1e-5, atol=1e-6)
