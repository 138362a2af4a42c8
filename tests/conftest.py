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
    return m


SAMPLING_CASES = ["e5_steps10", "e5_steps100", "e7_steps12", "e7_steps50_k8", "e7_damped", "e7_wide", "e7_guid",
                  "e7_guid_n2_rolls", "e5_guid_all", "sim_maximize", "sim_maximize_b"]
STL_CASES = ["stl_mixed", "stl_mixed_k8", "stl_wild"]
