"""csrc/adam_core.hpp -- the element update of pstl_adam_step -- compiled for the host (tests/hostsim) and held against
torch.optim.Adam (the optimiser of the reference's training loop, nusc_train.py:1233) BIT FOR BIT over several steps, tiny and
zero gradients included: both moments always, the parameter wherever torch's own vectorised CPU square root is the IEEE one
(it is a 0.5+ ulp routine: ~0.6 % of its results differ from sqrtf by one ulp, and there the parameter may differ by up to
three ulps of its increment too); and the per-step scalars the engine's DeviceAdam puts into its device table against torch's own."""
import ctypes
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib():
    path = os.path.join(HERE, "hostsim", "libpstl_hostsim.so")
    if not os.path.exists(path):
        pytest.skip("tests/hostsim/libpstl_hostsim.so not built (python __graft_entry__.py)")
    L = ctypes.CDLL(path)
    L.hostsim_adam_step.restype = None
    L.hostsim_adam_step.argtypes = [ctypes.c_long] + [ctypes.c_void_p] * 4 + [ctypes.c_float] * 6
    return L


# ((0.1, 0.2): 1 - beta1 >= 0.5 takes the OTHER branch of ATen's lerp)
@pytest.mark.parametrize("lr,betas,eps", [(3e-4, (0.9, 0.999), 1e-8), (1e-3, (0.9, 0.999), 1e-8), (0.01, (0.8, 0.99), 1e-6),
                                          (0.02, (0.1, 0.2), 1e-8), (1e-3, (0.5, 0.9), 1e-8)])
def test_element_update_equals_torch_adam_bit_for_bit(lr, betas, eps):
    from pstl_diffusion_policy_amd.engine import adam_schedule
    L = _lib()
    g = torch.Generator().manual_seed(7)
    n = 4099
    p0 = torch.randn(n, generator=g) * 0.1
    p_t = p0.clone().requires_grad_()
    opt = torch.optim.Adam([p_t], lr=lr, betas=betas, eps=eps)
    p = p0.numpy().copy()
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    sched = adam_schedule(lr, betas, 12)
    for t in range(12):
        grad = torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-9, 1, (1,), generator=g)))
        grad[::7] = 0.0                                   # exact zeros
        grad[1::11] *= 1e-30                              # denormal-range squares
        p_t.grad = grad.clone()
        opt.step()
        gn = grad.numpy().copy()
        p_prev = p.copy()
        L.hostsim_adam_step(n, p.ctypes.data, m.ctypes.data, v.ctypes.data, gn.ctypes.data, float(sched[t, 0]), float(sched[t, 1]),
                            1 - betas[0], betas[1], 1 - betas[1], eps)
        st = opt.state[p_t]
        assert np.array_equal(m.view(np.uint32), st["exp_avg"].numpy().view(np.uint32)), "exp_avg, step %d" % (t + 1)
        assert np.array_equal(v.view(np.uint32), st["exp_avg_sq"].numpy().view(np.uint32)), "exp_avg_sq, step %d" % (t + 1)
        want = p_t.detach().numpy()
        ieee = torch.sqrt(st["exp_avg_sq"]).numpy().view(np.uint32) == np.sqrt(v).view(np.uint32)    # torch's sqrt == sqrtf here
        assert ieee.mean() > 0.97
        assert np.array_equal(p.view(np.uint32)[ieee], want.view(np.uint32)[ieee]), "param, step %d" % (t + 1)
        ulp = np.spacing(np.abs(want).astype(np.float32))
        # (a square root one ulp off moves the denominator by up to two ulps and the quotient by up to three: 3.6e-7 of the increment)
        assert (np.abs(p - want) <= ulp + 4e-7 * np.abs(want - p_prev)).all(), \
            "param off by more than an ulp of itself + three of its increment where torch's sqrt is not the IEEE one"
        p = want.copy()       # (every step is checked on its own)


def test_schedule_matches_torchs_python_scalars():
    from pstl_diffusion_policy_amd.engine import adam_schedule
    s = adam_schedule(3e-4, (0.9, 0.999), 5)
    for t in range(1, 6):
        assert s[t - 1, 0] == np.float32(-(3e-4 / (1 - 0.9 ** t)))
        assert s[t - 1, 1] == np.float32((1 - 0.999 ** t) ** 0.5)


def test_default_table_ends_at_the_scalars_limits():
    """DeviceAdam's table is as long as float32(1 - beta^t) needs to become 1.0 for both betas: its last entry then holds the
    limits (-lr, 1.0), the kernel clamps its step index to that entry, and any number of graph replays stays exact with no
    host involvement.  Checked against the schedule itself: at the limits at the table's end, not yet 2 000 steps earlier."""
    from pstl_diffusion_policy_amd.engine import DeviceAdam, adam_schedule
    lr, betas = 3e-4, (0.9, 0.999)
    n = DeviceAdam.table_len(betas)
    assert 17000 < n < 18000, n
    tail = adam_schedule(lr, betas, 3, first=n - 2)
    assert (tail[:, 0] == np.float32(-lr)).all() and (tail[:, 1] == np.float32(1.0)).all()
    early = adam_schedule(lr, betas, 1, first=n - 2000)
    assert early[0, 1] < np.float32(1.0)
    later = adam_schedule(lr, betas, 2, first=10 * n)           # and it stays there
    assert (later[:, 0] == np.float32(-lr)).all() and (later[:, 1] == np.float32(1.0)).all()
    assert DeviceAdam.table_len((0.9, 0.9999999)) == DeviceAdam.TABLE_MAX      # cut short: extended from the host instead
    assert DeviceAdam.table_len((0.0, 0.5)) >= 2
