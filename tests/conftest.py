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


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def golden_weights():
    return dict(np.load(os.path.join(GOLDEN, "weights_seed1007.npz")))


def scene_from_golden(d):
    keys = ["ego_traj", "neighbors", "neighbors_traj", "currlane_wpts", "leftlane_wpts", "rightlane_wpts",
            "curr_id", "left_id", "right_id", "stlp_modes"]
    return {k: d["in_" + k] for k in keys}


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
                use_rect=bool(meta["use_rect"]))


SAMPLING_CASES = ["e5_steps10", "e5_steps100", "e7_steps12", "e7_steps50_k8", "e7_damped", "e7_wide", "e7_guid",
                  "e7_guid_n2_rolls", "e5_guid_all", "sim_maximize", "sim_maximize_b",
                  "fl_e8_clip_rect", "fl_no_arch", "fl_no_refinenet", "fl_not_use_rect", "fl_guid_sets", "fl_guid_freq_rev", "e7_s64_guid"]
STL_CASES = ["stl_mixed", "stl_mixed_k8", "stl_wild"]
