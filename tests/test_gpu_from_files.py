"""Config 1 end to end FROM FILES on the GPU: an experiment directory in the reference's formats (cache.npz, split file,
models/params_*.npy, model_last.ckpt) -> nusc_train.main(--run_sampling_test) and main(--trajopt_only), which writes the
traj-opt solution files the dataset then reads back."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_sampling_test_and_trajopt_from_files(tmp_path, capsys):
    assert torch.cuda.is_available()
    from pstl_diffusion_policy_amd import nusc_dataset as nd
    from pstl_diffusion_policy_amd import nusc_train as nt
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    root = str(tmp_path / "e1")
    base = ["--diffusion", "--load_stlp", "--n_randoms", "8", "--sampling_size", "8", "--n_neighbors", "3", "--batch_size", "4",
            "--diffusion_steps", "10", "--cache_path", root]
    args = nt.generate_parser(base)
    nd.write_synthetic_experiment(root, 12, args, seed=5)
    nd.save_checkpoint(init_state_dict(1007), os.path.join(root, "models"))
    md = nt.main(base + ["--rect_head", "--multi_cands", "3", "--run_sampling_test", "--test", "-P", nd.smart_path(root)])
    out = capsys.readouterr().out
    line = [l for l in out.splitlines() if l.startswith("###[")][-1]
    assert "TJ acc:" in line and "NN acc:" in line and "nan" not in line, line      # every column of the reference's line
    assert 0.0 <= md("acc") <= 1.0 and np.isfinite(md("std")) and np.isfinite(md("ade"))
    # the data-augmentation pass over the train split rewrites params_*.npy / scores_*.npy
    before = np.load(nd.trajopt_paths(os.path.join(root, "models"), 0, 1)["params"])
    nt.main(base + ["--trajopt_only", "--traj_opt_iters", "25", "--trajopt_lr", "0.01"])
    after = np.load(nd.trajopt_paths(os.path.join(root, "models"), 0, 1)["params"])
    assert after.shape == before.shape == (8, 3, 20, 2) and np.abs(after - before).max() > 1e-3
    sc = np.load(nd.trajopt_paths(os.path.join(root, "models"), 0, 1)["tj_scores_prior"])
    assert sc.shape == (8, 3) and np.isfinite(sc).all()
    ds = nd.get_dataloader(nt.generate_parser(base), root, split="train").dataset
    assert torch.equal(ds[0]["params"], torch.from_numpy(after))


def test_refinenet_training_from_files_updates_the_checkpoint(tmp_path, capsys):
    """README training commands of the RefineNet configs (e7_ours: --diverse_loss --stl_weight 0; e8_ours_ablation:
    --stl_weight 1 --diversity_weight 0) through the CLI mirror: only rect_net moves, model_last.ckpt is rewritten."""
    from pstl_diffusion_policy_amd import nusc_dataset as nd
    from pstl_diffusion_policy_amd import nusc_train as nt
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    root = str(tmp_path / "e7")
    base = ["--diffusion", "--load_stlp", "--n_randoms", "16", "--sampling_size", "16", "--n_neighbors", "3", "--batch_size", "4",
            "--diffusion_steps", "10", "--cache_path", root, "--rect_head", "--flex", "--multi_cands", "3", "--epochs", "2",
            "--lr", "1e-3", "--print_freq", "1"]
    nd.write_synthetic_experiment(root, 12, nt.generate_parser(base), seed=6)
    sd0 = init_state_dict(1007)
    for extra in (["--diverse_loss", "--stl_weight", "0.0"], ["--stl_weight", "1.0", "--diversity_weight", "0.0"],
                  ["--diverse_loss", "--no_arch", "--clip_rect", "--stl_weight", "1.0"]):
        nd.save_checkpoint(sd0, os.path.join(root, "models"))
        md = nt.main(base + extra + ["-P", nd.smart_path(root)])
        assert np.isfinite(md("loss"))
        sd1 = torch.load(nd.smart_path(root), map_location="cpu")
        moved = {k for k in sd1 if k in sd0 and not torch.equal(sd1[k], sd0[k].cpu())}
        assert moved and all(k.startswith("rect_net.") for k in moved), moved
    assert "epoch 001" in capsys.readouterr().out
    # --joint (reference nusc_train.py:1230-1231): the encoders and merge_net move too, the denoiser never does
    for extra in (["--diverse_loss", "--stl_weight", "1.0", "--joint"], ["--stl_weight", "1.0", "--diversity_weight", "0.0", "--joint"]):
        nd.save_checkpoint(sd0, os.path.join(root, "models"))
        md = nt.main(base + extra + ["-P", nd.smart_path(root)])
        assert np.isfinite(md("loss"))
        sd1 = torch.load(nd.smart_path(root), map_location="cpu")
        moved = {k.split(".")[0] for k in sd1 if k in sd0 and not torch.equal(sd1[k], sd0[k].cpu())}
        want = {"rect_net", "ego_encoder", "neighbor_encoder", "lane_encoder"} | ({"merge_net"} if "--diverse_loss" in extra else set())
        assert moved == want, moved
    # a -P that names no file is an error (the reference fails in torch.load), unless random init is asked for
    with pytest.raises(SystemExit):
        nt.main(base + ["--stl_weight", "1.0", "-P", str(tmp_path / "no_such.ckpt")])
    md = nt.main(base + ["--stl_weight", "1.0", "--epochs", "1", "--allow_random_init", "-P", str(tmp_path / "no_such.ckpt")])
    assert np.isfinite(md("loss"))


@pytest.mark.parametrize("extra", [["--rect_head", "--flex", "--diverse_loss", "--multi_cands", "3"],
                                   ["--rect_head", "--flex", "--diverse_loss", "--multi_cands", "3", "--guidance", "--guidance_before",
                                    "4", "--guidance_niters", "1", "--guidance_lr", "0.01", "--n_rolls", "2"],
                                   ["--flex"]])
def test_kernel_noise_graph_replay_equals_the_eager_region(tmp_path, capsys, extra):
    """--kernel_noise: run_sampling_test replays ONE captured HIP graph per batch (nusc_train._GraphRegion); --no_graph keeps
    the eager launches.  Same generator state in, same printed line out -- every batch, every column (the satisfaction rates
    and the diversity / ADE / FDE metrics of the sampled controls are functions of every row)."""
    from pstl_diffusion_policy_amd import nusc_dataset as nd
    from pstl_diffusion_policy_amd import nusc_train as nt
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    root = str(tmp_path / "e7")
    base = ["--diffusion", "--load_stlp", "--n_randoms", "16", "--sampling_size", "16", "--n_neighbors", "3", "--batch_size", "4",
            "--diffusion_steps", "12", "--cache_path", root, "--run_sampling_test", "--test", "--kernel_noise"] + extra
    nd.write_synthetic_experiment(root, 40, nt.generate_parser(base), seed=7)
    sd = init_state_dict(1007)
    if "--rect_head" not in extra:      # an e5 checkpoint holds no RefineNet
        sd = {k: v for k, v in sd.items() if not k.startswith(("rect_net.", "merge_net."))}
    nd.save_checkpoint(sd, os.path.join(root, "models"))
    lines = {}
    for mode in ([], ["--no_graph"]):
        torch.manual_seed(123)
        nt.main(base + mode + ["-P", nd.smart_path(root)])
        out = capsys.readouterr().out
        lines[bool(mode)] = [l.split("||| T:")[0] for l in out.splitlines() if l.startswith("###[")]
    assert len(lines[False]) >= 1 and lines[False] == lines[True], (lines[False], lines[True])
    print("batches compared:", len(lines[False]))
