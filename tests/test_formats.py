"""On-disk formats (cache.npz, split files, traj-opt .npy files, checkpoints): round trips, and compatibility with a
cache written by the reference's own save_cache_data + np.savez (tests/golden/ref_cache.npz)."""
import os

import numpy as np
import torch

from pstl_diffusion_policy_amd import nusc_dataset as nd
from pstl_diffusion_policy_amd import nusc_train as nt
from pstl_diffusion_policy_amd.synthetic import make_scene_batch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _args(extra=()):
    return nt.generate_parser(["--diffusion", "--load_stlp", "--n_randoms", "8", "--sampling_size", "8", "--n_neighbors", "3",
                               "--batch_size", "4"] + list(extra))


def test_synthetic_experiment_round_trip(tmp_path):
    args = _args()
    root = str(tmp_path / "exp")
    cache_path, splits, model_dir = nd.write_synthetic_experiment(root, 10, args, seed=3)
    cache, meta = nd.read_cache(cache_path)
    assert sorted(cache.keys()) == [0, 1, 2] and len(meta) == 3
    want = make_scene_batch(10, K=3, S=8, seed=3, invalid_lane_frac=0.2, stlp_mode="wide")
    loader = nd.get_dataloader(args, root, split="val")
    got = [b for b in loader]
    n_val = sum(b["ego_traj"].shape[0] for b in got)
    assert n_val == 3 and len(nd.read_split(splits["train"])) == 7
    b = got[0]
    for k in nd.SCENE_KEYS + ("params", "pre_stlp", "tj_scores_prior"):
        assert torch.equal(b[k], want[k][7:10]), k
    assert b["traj_i"].tolist() == [1, 2, 2] and b["ti"].tolist() == [4, 1, 2]


def test_reads_a_cache_written_by_the_reference(tmp_path):
    """ref_cache.npz was produced by the reference's save_cache_data on a synthetic batch (make_golden.py --formats)."""
    cache, meta = nd.read_cache(os.path.join(GOLD, "ref_cache.npz"))
    want = make_scene_batch(3, K=2, S=4, seed=9)
    args = _args(["--n_randoms", "4", "--sampling_size", "4"])
    ds = nd.MyDataset([(int(t), list(tok)) for t, tok in meta], cache, [(0, 1, "a"), (0, 2, "b"), (1, 1, "c")], args)
    for i in range(3):
        s = ds[i]
        for k in ("ego_traj", "neighbors_traj", "currlane_wpts", "curr_id", "stlp_modes"):
            assert torch.equal(s[k], want[k][i]), k
    # and our writer produces the same bytes-level structure: same keys, dtypes and shapes per sample
    mine = nd.save_cache_data(dict({k: want[k] for k in nd.SCENE_KEYS}, traj_i=torch.tensor([0, 0, 1]),
                                   ti=torch.tensor([1, 2, 1]), len_full=torch.tensor([30, 30, 30])), {})
    for t in cache:
        for ti in cache[t]:
            assert set(cache[t][ti].keys()) == set(mine[t][ti].keys())
            for k in cache[t][ti]:
                a, b = np.asarray(cache[t][ti][k]), np.asarray(mine[t][ti][k])
                assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), k


def test_trajopt_files_and_checkpoint(tmp_path):
    args = _args(["--trajopt_only", "--model_dir", str(tmp_path / "models")])
    args.test = False
    p = torch.randn(2, 8, 3, 20, 2)
    stlp = torch.randn(2 * 8 * 3, 1, 6)
    nt.save_trajopt_params(p, "init", [5, 5], [1, 2], args, save_stlp=stlp)
    nt.save_trajopt_params(p * 2, "final", [5, 5], [1, 2], args)
    nt.save_trajopt_params(torch.ones(2, 8, 3), "scores", [5, 5], [1, 2], args)
    paths = nd.trajopt_paths(args.model_dir, 5, 2)
    assert np.array_equal(np.load(paths["params"]), (p * 2)[1].numpy())
    assert np.array_equal(np.load(paths["params_init"]), p[1].numpy())
    assert np.load(paths["pre_stlp"]).shape == (8, 3, 1, 6)
    assert np.load(paths["tj_scores_prior"]).shape == (8, 3)
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    sd = init_state_dict(1007)
    nd.save_checkpoint(sd, str(tmp_path / "models"))
    back = torch.load(nd.smart_path(str(tmp_path)), map_location="cpu")
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k].cpu()) for k in sd)


def test_get_dense_stlp_matches_reference():
    """Host-side parameter expansion of the traj-opt pass against the reference's get_dense_stlp: exact in the fixed-prior
    branch, and in the --flex branch under the same torch seed (same draws in the same order)."""
    g = dict(np.load(os.path.join(GOLD, "dense_stlp.npz")))
    S = int(g["S"])
    the_stlp = torch.from_numpy(g["in_stlp"])
    batch = {"gt_high_level": torch.from_numpy(g["in_gt_high_level"])}
    for flex in (0, 1):
        args = _args((["--trajopt_only"] if flex else []) + ["--n_randoms", str(S), "--sampling_size", str(S)])
        assert bool(args.flex) == bool(flex)        # --trajopt_only forces --flex, as in the reference's parser
        torch.manual_seed(123)
        got = nt.get_dense_stlp(batch, the_stlp, args).numpy()
        np.testing.assert_array_equal(got, g["dense_flex%d" % flex])
