"""On-disk formats on either side of the hot path, host side only -- no GPU work happens here.

Formats kept byte-compatible with the reference so that its files and these files are interchangeable:
  cache.npz ............ np.savez(data={traj_i: {ti: {key: ndarray}}}, meta_list=[(traj_i, [tokens...]), ...])
                         written by collect_nuscene_data / save_cache_data (reference nusc_train.py:190-208), read by
                         get_dataloader (:155-157) with allow_pickle=True
  <split>_split.txt .... one "traj_i ti token" line per sample (reference nusc_dataset.py:82-90, data/*_split.txt)
  models/params_%05d_%04d{,_init,_stlp}.npy, models/scores_%05d_%04d.npy
                         traj-opt solutions, per-sample STL parameters and scores (reference nusc_train.py:775-797,
                         read by nusc_dataset.py:203-225)
  models/model_last.ckpt torch.save(net.state_dict()) (reference utils.py:81-85)
MyDataset is the offline branch of the reference's dataset (nusc_dataset.py:109-116,202-240); the on-line branch needs the
nuScenes devkit and is out of scope.  `write_synthetic_experiment` lays a complete experiment directory down from the
seeded synthetic scene generator so that `nusc_train.main` can run config 1 end to end from files, as the reference does.
"""
import os

import numpy as np
import torch

from .synthetic import make_scene_batch

SCENE_KEYS = ("ego_traj", "neighbors", "neighbors_traj", "currlane_wpts", "leftlane_wpts", "rightlane_wpts", "curr_id",
              "left_id", "right_id", "gt_high_level", "stlp_modes")


def find_npz_path(path, exp_root="."):
    if ".npz" not in path:
        path = os.path.join(path, "cache.npz")
    return path if path.startswith("/") else os.path.join(exp_root, path)


def save_cache_data(batch, saved_sample_d):
    """Adds the samples of one collated batch to {traj_i: {ti: {key: ndarray}}} (everything but `params`)."""
    batch_np = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in batch.items()}
    for i in range(batch_np["traj_i"].shape[0]):
        traj_i, ti = batch_np["traj_i"][i], batch_np["ti"][i]
        saved_sample_d.setdefault(traj_i, {})[ti] = {k: v[i] for k, v in batch_np.items() if k != "params"}
    return saved_sample_d


def write_cache(path, saved_sample_d, meta_list):
    np.savez(path, data=saved_sample_d, meta_list=np.asarray(meta_list, dtype=object))


def read_cache(path):
    z = np.load(path, allow_pickle=True)
    return z["data"].item(), z["meta_list"]


def write_split(path, indices):
    with open(path, "w") as f:
        for traj_i, ti, token in indices:
            f.write("%d %d %s\n" % (int(traj_i), int(ti), token))


def read_split(path, test_t1=False):
    out = []
    with open(path) as f:
        for line in f:
            traj_i, ti, token = line.strip().split(" ")
            if test_t1 and int(ti) != 1:
                continue
            out.append([int(traj_i), int(ti), token])
    return out


def trajopt_paths(model_dir, traj_i, ti):
    key = (int(traj_i), int(ti))
    return {"params": os.path.join(model_dir, "params_%05d_%04d.npy" % key),
            "params_init": os.path.join(model_dir, "params_%05d_%04d_init.npy" % key),
            "pre_stlp": os.path.join(model_dir, "params_%05d_%04d_stlp.npy" % key),
            "tj_scores_prior": os.path.join(model_dir, "scores_%05d_%04d.npy" % key)}


def save_checkpoint(state_dict, model_dir, name="model_last.ckpt"):
    os.makedirs(model_dir, exist_ok=True)
    torch.save({k: v.detach().cpu() for k, v in state_dict.items()}, os.path.join(model_dir, name))


def smart_path(s):
    return s if ".ckpt" in s else s + "/models/model_last.ckpt"


class MyDataset(torch.utils.data.Dataset):
    """Offline dataset: samples come from cache.npz, traj-opt solutions / STL parameters / scores from .npy files."""

    def __init__(self, meta_list, cache, split_indices, args, params_dir=None):
        self.meta_list, self.cache, self.indices, self.args = meta_list, cache, list(split_indices), args
        self.meta_d = {traj_i: tokens for traj_i, tokens in meta_list}
        self.params_dir = params_dir

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, idx):
        traj_i, ti, _ = self.indices[idx]
        args = self.args
        raw = self.cache[traj_i][ti]
        keep = ("traj_i", "ti", "len_full")
        sample = {k: (v if k in keep else torch.from_numpy(np.asarray(v))) for k, v in raw.items()}
        if self.params_dir is not None:
            p = trajopt_paths(self.params_dir, traj_i, ti)
            if os.path.exists(p["params"]):
                sample["params"] = torch.from_numpy(np.load(p["params"])).float()
                sample["params_init"] = torch.from_numpy(np.load(p["params_init"])).float()
            if getattr(args, "load_stlp", False):
                sample["pre_stlp"] = torch.from_numpy(np.load(p["pre_stlp"])).float()
                sample["tj_scores_prior"] = torch.from_numpy(np.load(p["tj_scores_prior"])).float()
        if "params" not in sample:   # fresh initial guess of the traj-opt pass (nusc_dataset.py:214-218)
            w = (torch.rand(args.n_randoms, 3, args.nt) * 2 - 1) * args.mul_w_max * 0.1
            a = (torch.rand(args.n_randoms, 3, args.nt) * 2 - 1) * args.mul_a_max
            sample["params"] = torch.stack([w, a], dim=-1)
            sample["params_init"] = sample["params"].clone()
        n0 = sample["params_init"].shape[0]
        if n0 != args.n_randoms:     # resample the stored solutions to the requested count (nusc_dataset.py:233-240)
            idx_s = np.random.choice(list(range(n0)), args.n_randoms)
            for k in ("params_init", "params", "pre_stlp", "tj_scores_prior"):
                if k in sample:
                    sample[k] = sample[k][idx_s]
        return sample


def write_synthetic_experiment(root, n_scenes, args, seed=0, val_frac=0.3, with_trajopt=True):
    """<root>/cache.npz, <root>/{train,val}_split.txt and (with_trajopt) <root>/models/params_*/scores_* files, from the
    seeded synthetic scene generator.  Returns (cache_path, split paths, models dir)."""
    os.makedirs(root, exist_ok=True)
    model_dir = os.path.join(root, "models")
    os.makedirs(model_dir, exist_ok=True)
    batch = make_scene_batch(n_scenes, K=args.n_neighbors, nt=args.nt, n_segs=args.n_segs, S=args.n_randoms, seed=seed,
                             dt=args.dt, invalid_lane_frac=0.2, stlp_mode="wide")
    per_traj = 4                                        # synthetic "drives" of 4 consecutive samples each
    batch["traj_i"] = torch.arange(n_scenes) // per_traj
    batch["ti"] = torch.arange(n_scenes) % per_traj + 1
    batch["len_full"] = torch.full((n_scenes,), per_traj + args.nt)
    saved = save_cache_data({k: batch[k] for k in SCENE_KEYS + ("traj_i", "ti", "len_full")}, {})
    n_traj = int(batch["traj_i"].max().item()) + 1
    meta_list = [(t, ["tok_%05d_%04d" % (t, j) for j in range(per_traj + args.nt)]) for t in range(n_traj)]
    cache_path = os.path.join(root, "cache.npz")
    write_cache(cache_path, saved, meta_list)
    idx = [(int(batch["traj_i"][i]), int(batch["ti"][i]), "tok_%05d_%04d" % (int(batch["traj_i"][i]), int(batch["ti"][i])))
           for i in range(n_scenes)]
    n_val = max(1, int(round(n_scenes * val_frac)))
    splits = {"train": os.path.join(root, "train_split.txt"), "val": os.path.join(root, "val_split.txt")}
    write_split(splits["train"], idx[:n_scenes - n_val])
    write_split(splits["val"], idx[n_scenes - n_val:])
    if with_trajopt:
        for i in range(n_scenes):
            p = trajopt_paths(model_dir, batch["traj_i"][i], batch["ti"][i])
            np.save(p["params"], batch["params"][i].numpy())
            np.save(p["params_init"], batch["params"][i].numpy())
            np.save(p["pre_stlp"], batch["pre_stlp"][i].numpy())
            np.save(p["tj_scores_prior"], batch["tj_scores_prior"][i].numpy())
    return cache_path, splits, model_dir


def get_dataloader(args, root, split="val"):
    """DataLoader over an experiment directory laid out as above (the offline branch of the reference's get_dataloader,
    nusc_train.py:152-187)."""
    cache, meta_list = read_cache(find_npz_path(root))
    indices = read_split(os.path.join(root, "%s_split.txt" % split), test_t1=getattr(args, "test_t1", False))
    ds = MyDataset(meta_list, cache, indices, args, params_dir=os.path.join(root, "models"))
    return torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=(split == "train"), num_workers=0,
                                       drop_last=False)
