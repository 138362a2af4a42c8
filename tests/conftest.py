import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "timing: a test that reports a latency (no poisoned allocations: their fills would be timed)")


_LDS_POISON = []


@pytest.fixture(autouse=True)
def _poisoned_lds(request):
    """Before every -m gpu test: every CU's LDS holds NaN patterns (tests/ldspoison/lds_poison.hip, built by
    __graft_entry__.build()), so that a kernel reading LDS it never wrote fails its test every time instead of depending on
    what the previous kernel left there.  The library is test infrastructure; when it is not built the fixture says so once."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    if not _LDS_POISON:
        import ctypes
        so = os.path.join(ROOT, "tests", "ldspoison", "liblds_poison.so")
        try:
            lib = ctypes.CDLL(so)
            lib.lds_poison.restype = ctypes.c_int
            lib.lds_poison.argtypes = [ctypes.c_void_p]
            _LDS_POISON.append(lib)
        except OSError:
            import warnings
            warnings.warn("tests/ldspoison/liblds_poison.so is not built (run __graft_entry__.build()): LDS not poisoned")
            _LDS_POISON.append(None)
    if _LDS_POISON[0] is not None:
        import torch
        if torch.cuda.is_available():
            assert _LDS_POISON[0].lds_poison(ctypes_stream()) == 0
    yield


@pytest.fixture(autouse=True)
def _poisoned_empty(request, monkeypatch):
    """The same for global memory: during a -m gpu test every floating-point device buffer that comes from torch.empty /
    empty_like is born full of 2.7e36 (large finite, see lds_poison.hip for why not NaN), so that a kernel reading an output or
    work buffer before it wrote it cannot hide behind the zeros a fresh allocation usually holds."""
    if request.node.get_closest_marker("gpu") is None or request.node.get_closest_marker("timing") is not None:
        yield
        return
    import torch
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def poisoned(t):
        if isinstance(t, torch.Tensor) and t.is_cuda and t.is_floating_point() and t.numel() > 0:
            t.fill_(2.66e36)
        return t

    monkeypatch.setattr(torch, "empty", lambda *a, **k: poisoned(real_empty(*a, **k)))
    monkeypatch.setattr(torch, "empty_like", lambda *a, **k: poisoned(real_empty_like(*a, **k)))
    yield


def ctypes_stream():
    import ctypes
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def golden_weights(d=None):
    """The state_dict a fixture was recorded with: weights_seed1007.npz, re-derived by tests/heavy_weights.py when the
    fixture names a `weights_variant` (exact: integer draws and power-of-two factors only)."""
    sd = dict(np.load(os.path.join(GOLDEN, "weights_seed1007.npz")))
    if d is not None and "weights_variant" in d:
        if str(d["weights_variant"]) == "trained":     # the checkpoint the reference itself trained (make_golden.py --trained)
            return dict(np.load(os.path.join(GOLDEN, "weights_trained.npz")))
        from heavy_weights import heavy_weights
        sd = heavy_weights(sd, str(d["weights_variant"]))
    return sd


def hparams_for(d=None):
    """default_hparams(), with norm_stl switched on for fixtures recorded under --norm_stl (`meta_norm`)."""
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    hp = default_hparams()
    if d is not None and "meta_norm" in d and int(d["meta_norm"][0]):
        hp = dict(hp, norm_stl=True)
    return hp


def scene_from_golden(d):
    keys = ["ego_traj", "neighbors", "neighbors_traj", "currlane_wpts", "leftlane_wpts", "rightlane_wpts",
            "curr_id", "left_id", "right_id", "stlp_modes"]
    scene = {k: d["in_" + k] for k in keys}
    if "in_stlp_rows" in d:      # fixtures recorded from the reference's own harness: the STL parameters it inferred
        del scene["stlp_modes"]
        scene["stlp_rows"] = d["in_stlp_rows"]
    return scene


def golden_meta(d):
    names = ["bs", "S", "K", "steps", "seed", "rect_head", "guidance", "multi_cands", "diffusion_clip",
             "force_full", "guidance_before", "guidance_niters", "n_rolls", "zero_net_out", "maximize"]
    m = {k: int(v) for k, v in zip(names, d["meta"])}
    m.setdefault("maximize", 0)
    m["guidance_lr"], m["stl_nn_thres"], m["tau"] = [float(v) for v in d["meta_f"]]
    # flag variants (fl_* fixtures); the older fixtures are all "merge_net architecture, RefineNet on, no clip_rect"
    x = [int(v) for v in d["meta_x"]] if "meta_x" in d else [1, 0, 1, 1, 0, -1]
    m["diverse"], m["clip_rect"], m["refinenet"], m["use_rect"], m["guidance_reverse"], m["guidance_freq"] = x
    m["guidance_sets"] = [int(v) for v in d["guid_sets"]] if ("guid_sets" in d and len(d["guid_sets"])) else None
    m["refinement"] = int(d["meta_refinement"][0]) if "meta_refinement" in d else 0
    return m


def guidance_cfg(meta):
    """The guidance dict engine.Sampler / the oracle take, from a fixture's meta."""
    if not meta["guidance"]:
        return None
    return dict(enabled=True, before=meta["guidance_before"], niters=meta["guidance_niters"], lr=meta["guidance_lr"],
                maximize=bool(meta["maximize"]), reverse=bool(meta["guidance_reverse"]), sets=meta["guidance_sets"],
                freq=None if meta["guidance_freq"] < 0 else meta["guidance_freq"])


def region_kwargs(meta):
    """Keyword arguments of sampling_region (engine and oracle share them) for a fixture."""
    return dict(rect_head=bool(meta["rect_head"]), multi_cands=None if meta["multi_cands"] < 0 else meta["multi_cands"],
                guidance=guidance_cfg(meta), n_rolls=None if meta["n_rolls"] < 0 else meta["n_rolls"],
                refinenet=bool(meta["refinenet"]), diverse=bool(meta["diverse"]), clip_rect=bool(meta["clip_rect"]),
                use_rect=bool(meta["use_rect"]), refinement_iters=meta.get("refinement") or None)


SAMPLING_CASES = ["e5_steps10", "e5_steps100", "e7_steps12", "e7_steps50_k8", "e7_damped", "e7_wide", "e7_guid",
                  "e7_guid_n2_rolls", "e5_guid_all", "sim_maximize", "sim_maximize_b",
                  "fl_e8_clip_rect", "fl_no_arch", "fl_no_refinenet", "fl_not_use_rect", "fl_guid_sets", "fl_guid_freq_rev", "e7_s64_guid", "e7_guid_c4",
                  "e7_heavy_a", "e7_heavy_b", "e7_readme_guidance", "e7_guid_norm", "e7_guid_norm_n2",
                  "e7_trained_guid", "e7_trained_s16"]     # (the last two: on weights the reference itself trained)
HEAVY_CASES = ["e7_heavy_a", "e7_heavy_b"]      # weights with a trained network's dynamic range (tests/heavy_weights.py)
STL_CASES = ["stl_mixed", "stl_mixed_k8", "stl_wild"]
REFINEMENT_CASES = ["e7_refinement", "e7_refinement_b"]


def refinement_gate(mine, grads, d, tol=1e-4, min_frac=0.8):
    """--refinement runs 50 Adam steps (lr 0.3) on eight mixing weights per row through a score with hard minima in it
    (closest neighbour, closest circle pair, closest lane segment): a last-bit difference that flips one of those minima
    changes the gradient's direction, and with steps of 0.3 the two runs part ways for good -- the CPU oracle, which agrees
    with the reference BIT FOR BIT on most rows, is 0.2 away on a few.  So the gate is on the mechanism and on the
    population:  (1) the first gradient (no history) matches everywhere; (2) the gradients of iterations 1..19 match on
    >= 95 % of the rows (observed on the GPU: all rows through iteration 19, 76-98 % at iteration 49); (3) rows the
    reference left untouched are untouched; (4) >= min_frac of the rows end within `tol` of the reference (the callers add:
    the refined batch reaches the reference's loss).  mine (N,20,2); grads (>= 20, N, 8) = d loss / d lambda per iteration."""
    want, g_ref = d["refinement_controls"], d["refinement_grads"]
    N = want.shape[0]
    g = np.asarray(grads)
    np.testing.assert_allclose(g[0], g_ref[0], rtol=2e-3, atol=2e-6 * np.abs(g_ref[0]).max())
    for it in range(1, 20):
        sc = np.abs(g_ref[it]).max(axis=1, keepdims=True) + 1e-30
        row_ok = (np.abs(g[it] - g_ref[it]) <= 5e-3 * sc + 1e-9).all(axis=1)
        assert row_ok.mean() >= 0.95, "iteration %d: gradients agree on %.3f of the rows" % (it, row_ok.mean())
    mine = np.asarray(mine)
    untouched = np.abs(want - d["refinement_in_controls"]).reshape(N, -1).max(axis=1) == 0
    np.testing.assert_array_equal(mine[untouched], want[untouched])
    ok = np.abs(mine - want).reshape(N, -1).max(axis=1) <= tol
    assert ok.mean() >= min_frac, "rows within %g of the reference: %.3f" % (tol, ok.mean())
    return ok


def guided_outlier_rows(err_all, d, meta, tol=1e-4, g_eps=1e-6, n_shards=4):
    """The gate of a GUIDED run.

    Adam's normalised step lr * g / (|g| + 1e-8) is discontinuous in the limit g -> 0: where the reference's own STL
    gradient of an element is below ~1e-6 (a near-cancellation of ~1e-3 terms), a float32-level difference in g moves the
    control by up to lr -- no two float32 implementations agree there.  So: rows in which some element of the per-step
    list leaves `tol` are EXCLUDED from the later comparisons (and counted: at most 0.1 % of the rows, at least one
    allowed), every other row stays at `tol` throughout; and the explanation is CHECKED, not assumed: the first step at
    which a row leaves `tol` must be a guided step, and every element that left it there must have a reference gradient
    |g| < g_eps in one of that step's Adam iterations (`guid_grads`, recorded from the reference's optimizer.step()).

    err_all: (steps, N, 20, 2) = |mine - reference| of controls_list.  Returns (bad_rows (N,), bad_groups (N,)):
    bad_groups also marks the rows that share a merge_net max-pool group (scene, mode, shard) with a bad row -- their
    RefineNet input is touched by the bad row's output."""
    from pstl_diffusion_policy_amd.engine import guidance_triggered
    steps, N = err_all.shape[0], err_all.shape[1]
    cfg = guidance_cfg(meta)
    guided = [i for i in range(steps - 1, 0, -1) if guidance_triggered(i, steps, cfg)]    # rollout order
    nit = meta["guidance_niters"]
    grads = np.abs(d["guid_grads"]).reshape(len(guided), nit, N, 20, 2)
    bad = err_all > tol
    bad_rows = bad.any(axis=(0, 2, 3))
    for r in np.nonzero(bad_rows)[0]:
        k0 = int(np.argmax(bad[:, r].any(axis=(1, 2))))      # list entry k is the state after reverse step steps - k
        i0 = steps - k0
        assert i0 in guided, "row %d leaves %g at reverse step %d, which is not a guided step" % (r, tol, i0)
        gmin = grads[guided.index(i0), :, r].min(axis=0)
        el = bad[k0, r]
        assert (gmin[el] < g_eps).all(), ("row %d, reverse step %d: outlier elements whose reference gradient is NOT in "
                                          "Adam's eps regime: |g| = %s, err = %s" % (r, i0, gmin[el], err_all[k0, r][el]))
    # How many rows may be excluded: at most 0.1 % of the rows, at least one.  Only where the REFERENCE's recorded gradients hold
    # an unusually large population of elements in Adam's eps regime (0 < |g| < g_eps; more than 3 500 of them -- of the committed
    # fixtures that is e7_trained_guid alone: 15 664 such elements in 3 004 active row-steps on the checkpoint the reference
    # trained itself; every fixture on random-init or hand-scaled weights holds <= 3 125 and keeps the allowance of ONE row) does
    # the allowance follow the population: which of those elements flip is a coin toss per float32 implementation -- the CPU
    # oracle, float32 torch like the reference, loses 4 rows there, the HIP builds of rounds 5 and 6 between 4 and 6 --, so the
    # count is held to a two-sigma band around the rate of one row per 3 500 elements: lam + 2 sqrt(lam).
    tiny = int(((grads < g_eps) & (grads > 0)).sum())
    allowed = max(1, int(0.001 * N))
    if tiny > 3500:
        lam = tiny / 3500.0
        allowed = max(allowed, int(np.ceil(lam + 2.0 * np.sqrt(lam))))
    assert bad_rows.sum() <= allowed, "%d of %d rows hold an outlier (allowed %d)" % (bad_rows.sum(), N, allowed)
    S = meta["S"]
    bad_groups = bad_rows.copy()
    if bad_rows.any() and S % n_shards == 0:
        sps = S // n_shards
        grp = bad_rows.reshape(-1, n_shards, sps, 3).any(axis=2, keepdims=True)      # (bs, shard, 1, mode)
        bad_groups = np.broadcast_to(grp, (N // (S * 3), n_shards, sps, 3)).reshape(N).copy()
    return bad_rows, bad_groups
