"""N > 1 on the HIP path: ranks own contiguous scene shards, run the whole
sampling region through libpstl_hip.so with in-kernel noise keyed by the GLOBAL row, and meet only in the two tiny
exchanges of shard.py.  The sharded run must reproduce the single-process run over the whole batch: satisfaction
counters exactly, per-row results bit for bit, diversity totals to summation order.

 * gloo, two ranks sharing GPU 0: runs on the 1-GPU box (RCCL refuses two ranks on one device);
 * nccl (= RCCL), one rank per GPU: runs where >= 2 GPUs are visible, skipped otherwise;
 * `python bench.py --gpus 2` starts its own ranks and reports n_gpus = 2 (or refuses when the GPUs are not there)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu

BS, S, K, STEPS, SEED = 6, 16, 2, 12, 4242
GUID = dict(enabled=True, before=3, niters=1, lr=0.01)


def _region(scene, lo, hi, vsum, vrows, dev):
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    hp = default_hparams()
    sub = {k: v[lo:hi].to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
    sb = SceneBatch(sub, S, hp, dev, global_valid_sum=vsum, global_rows=vrows, row_offset=lo * S * 3)
    sm = Sampler(PackedWeights(init_state_dict(1007), dev), hp)
    return sm.sampling_region(sb, STEPS, None, None, rect_head=True, multi_cands=3, guidance=GUID, seed=SEED,
                              diversity=True, want_scores3=False)


def _scene():
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    return make_scene_batch(BS, K=K, S=S, seed=77, invalid_lane_frac=0.3, stlp_mode="wide")


def _worker(rank, world, port, backend, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pstl_diffusion_policy_amd.shard import gather_final, global_valid_stats, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    local = rank % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = _scene()
    lo, hi = shard_range(BS, rank, world)
    ids = float(sum(scene[k][lo:hi].sum().item() for k in ("curr_id", "left_id", "right_id")))
    vsum, vrows = global_valid_stats(ids * S, (hi - lo) * S * 3, dev if backend == "nccl" else torch.device("cpu"))
    out = _region(scene, lo, hi, vsum, vrows, dev)
    counts, totals = out["counts"], out["div_totals"]
    if backend != "nccl":       # gloo moves host tensors
        counts, totals = counts.cpu(), totals.cpu()
    counts, totals = gather_final(counts, totals)
    torch.save({"vsum": vsum, "vrows": vrows, "counts": counts.cpu(), "totals": totals.cpu(), "lo": lo, "hi": hi,
                "final_controls": out["final_controls"].cpu(), "final_scores": out["final_scores"].cpu()},
               out_path + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


def _check(tmp_path, backend, world):
    out_path = str(tmp_path / "rank")
    port = 23000 + (os.getpid() % 4000)
    mp.spawn(_worker, args=(world, port, backend, out_path), nprocs=world, join=True)
    parts = [torch.load(out_path + ".%d" % r) for r in range(world)]
    dev = torch.device("cuda:0")
    scene = _scene()
    ids = float(sum(scene[k].sum().item() for k in ("curr_id", "left_id", "right_id")))
    assert parts[0]["vsum"] == ids * S and parts[0]["vrows"] == BS * S * 3
    ref = _region(scene, 0, BS, ids * S, BS * S * 3, dev)
    torch.cuda.synchronize()
    for p in parts:       # every rank holds the same global numbers
        assert torch.equal(p["counts"], ref["counts"].cpu())
        assert torch.equal(p["counts"], parts[0]["counts"]) and torch.equal(p["totals"], parts[0]["totals"])
        np.testing.assert_allclose(p["totals"].numpy(), ref["div_totals"].cpu().numpy(), rtol=1e-12)
        rows = slice(p["lo"] * S * 3, p["hi"] * S * 3)    # shard invariance: a shard's rows are the whole batch's rows
        assert torch.equal(p["final_controls"], ref["final_controls"].cpu()[rows])
        assert torch.equal(p["final_scores"], ref["final_scores"].cpu()[rows])


def test_two_gloo_ranks_on_one_gpu_reproduce_the_single_process_run(tmp_path):
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    _check(tmp_path, "gloo", 2)


def test_rccl_ranks_reproduce_the_single_process_run(tmp_path):
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU visible: RCCL needs one device per rank")
    _check(tmp_path, "nccl", 2)


def _bench(args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=900)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2`: two ranks (gloo + shared device here when only one GPU is visible, RCCL otherwise),
    n_gpus = 2 in the line, twice the rows of the one-rank run; with RCCL and too few GPUs it refuses."""
    small = ["--scenes", "48", "--steps", "2", "--warmup", "1", "--no_cpu_baseline"]
    one = _bench(["--gpus", "1"] + small)
    assert one.returncode == 0, one.stderr[-2000:]
    l1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert l1["n_gpus"] == 1
    # the single-GPU line carries the driver-observed extras: the exact-fp32 leg, configs 2 / 3 / 5 and the batch-size sweep
    assert l1["roofline"]["fp32_exact"]["chain_waves"] == 8 and 0.0 < l1["roofline"]["whole_step_frac"] < 1.0
    assert sorted(l1["also"]) == ["e5", "e7", "e8_train"]
    assert all(v["ms_per_step"] > 0 and 0.0 < v["roofline"]["frac"] < 1.0 for v in l1["also"].values())
    assert l1["also"]["e8_train"]["backward"]["achieved_GBps"] > 0
    assert [s["rows"] for s in l1["sweep"]] == [192, 3072, 48 * 192] and all(s["value"] > 0 for s in l1["sweep"])
    assert l1["paper_metric"]["ours"]["time_ms_median"] > 0 and l1["paper_metric"]["ours_guidance"]["paper_time_ms"] == 786.0
    ndev = torch.cuda.device_count()
    two = _bench(["--gpus", "2"] + small, None if ndev >= 2 else {"PSTL_BENCH_BACKEND": "gloo"})
    assert two.returncode == 0, two.stderr[-2000:]
    l2 = json.loads(two.stdout.strip().splitlines()[-1])
    assert l2["n_gpus"] == 2 and l2["scaling"] == "weak" and "also" not in l2 and "sweep" not in l2
    assert l2["config"]["rows_per_gpu"] == l1["config"]["rows_per_gpu"]
    assert abs(l2["stl_sat_rate"] - l1["stl_sat_rate"]) < 0.2
    if ndev < 2:
        refused = _bench(["--gpus", "2"] + small)
        assert refused.returncode != 0 and "refusing" in refused.stderr


def test_bench_with_eight_ranks():
    """`python bench.py --gpus 8` before a real 8-GPU node runs it: eight ranks (gloo + the shared device
    when fewer than eight GPUs are visible, RCCL otherwise), one all-reduce of the valid-row statistics and one all-gather of
    the 20 result words per step; the line says n_gpus = 8 and its counters are exactly those of the eight shards evaluated
    one after the other in this process with the same global statistics, row offsets and seeds."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, Sampler, SceneBatch
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    bs, S, K, steps, world = 8, 64, 2, 50, 8
    ndev = torch.cuda.device_count()
    run = _bench(["--gpus", "8", "--scenes", str(bs), "--steps", "1", "--warmup", "1", "--no_cpu_baseline"],
                 None if ndev >= 8 else {"PSTL_BENCH_BACKEND": "gloo"})
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    N = bs * S * 3
    assert line["config"]["rows_per_gpu"] == N and line["counts"][2] == world * N and line["counts"][5] == world * bs * 3
    # who took part: eight records in the final all-gather, as many distinct devices as the box has
    rk = line["ranks"]
    assert rk["world_size"] == 8 and rk["ranks_seen"] == 8 and rk["distinct_devices"] == min(max(ndev, 1), 8)
    assert rk["backend"].startswith("gloo" if ndev < 8 else "nccl") and rk["host_scalar_exchange_ms_per_step"] >= 0.0
    assert line["config"]["plan_rows"] == N and line["config"]["rows_whole_job"] == world * N
    # the same eight shards, one after the other
    dev = torch.device("cuda:0")
    hp = default_hparams()
    sm = Sampler(PackedWeights(init_state_dict(1007), dev), hp)
    scenes = [make_scene_batch(bs, K=K, S=S, seed=1000 + r, invalid_lane_frac=0.2, stlp_mode="wide") for r in range(world)]
    vsum = sum(float(sum(sc[k].sum().item() for k in ("curr_id", "left_id", "right_id"))) * S for sc in scenes)
    total = torch.zeros(8, dtype=torch.int64, device=dev)
    for r, sc in enumerate(scenes):
        scene = {k: v.to(dev) for k, v in sc.items() if k not in ("pre_stlp", "tj_scores_prior")}
        sb = SceneBatch(scene, S, hp, dev, global_valid_sum=vsum, global_rows=world * N, row_offset=r * N)
        out = sm.sampling_region(sb, steps, None, None, rect_head=True, multi_cands=5,
                                 guidance=dict(enabled=True, before=10, niters=1, lr=0.01), want_scores3=False,
                                 seed=987654321 + 2, diversity=True)      # (the second call of the job: one warm-up, one step)
        total += out["counts"]
    assert line["counts"] == [int(v) for v in total.tolist()]


def test_bench_strong_scaling_splits_a_fixed_job():
    """`--scaling strong --total_scenes T`: a fixed job in contiguous blocks of scenes (BASELINE config 4 as written, at test
    size): three ranks over seven scenes take 3 + 2 + 2, every rank plans for the largest shard, the counters cover the job."""
    S, T, world = 64, 7, 3
    ndev = torch.cuda.device_count()
    run = _bench(["--gpus", str(world), "--scaling", "strong", "--total_scenes", str(T), "--steps", "1", "--warmup", "1",
                  "--no_cpu_baseline"], None if ndev >= world else {"PSTL_BENCH_BACKEND": "gloo"})
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["scaling"] == "strong"
    assert line["config"]["rows_whole_job"] == T * S * 3 and line["config"]["rows_per_gpu"] == 3 * S * 3      # rank 0: three scenes
    assert line["config"]["plan_rows"] == 3 * S * 3 and line["counts"][2] == T * S * 3 and line["counts"][5] == T * 3
    assert line["ranks"]["ranks_seen"] == world and line["value"] > 0


# ---- N > 1 training: RectTrainer.train_step all-reduces the gradients (the loss is a mean over the GLOBAL batch) ---------
T_BS, T_S, T_K, T_STEPS = 6, 16, 3, 10
T_E7 = dict(stl_weight=1.0, diversity_weight=0.5, diversity_scale=1.0, rect_reg_loss=0.0, detach=False)


def _train_grads(scene, lo, hi, vsum, vrows, dev, joint, group_ready):
    from pstl_diffusion_policy_amd.engine import PackedWeights, RectTrainer, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    hp = default_hparams()
    sd = {k: v.to(dev) for k, v in init_state_dict(1007).items()}
    sub = {k: v[lo:hi].to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}
    sb = SceneBatch(sub, T_S, hp, dev, global_valid_sum=vsum, global_rows=vrows, row_offset=lo * T_S * 3)
    names = RectTrainer.joint_names(True) if joint else RectTrainer.NAMES
    params = {k: sd[k].clone().requires_grad_() for k in names}
    opt = torch.optim.Adam([params[k] for k in names], lr=3e-4)
    tr = RectTrainer(Sampler(PackedWeights(sd, dev), hp))
    loss, _ = tr.train_step(sb, params, opt, T_STEPS, seed=SEED, multi_cands=3, e7=T_E7, joint=joint)
    return float(loss), {k: params[k].grad.detach().cpu() for k in names}


def _train_worker(rank, world, port, out_path, joint):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pstl_diffusion_policy_amd.shard import global_valid_stats, shard_range
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = make_scene_batch(T_BS, K=T_K, S=T_S, seed=78, invalid_lane_frac=0.3, stlp_mode="wide")
    lo, hi = shard_range(T_BS, rank, world)
    ids = float(sum(scene[k][lo:hi].sum().item() for k in ("curr_id", "left_id", "right_id")))
    vsum, vrows = global_valid_stats(ids * T_S, (hi - lo) * T_S * 3, torch.device("cpu"))
    loss, grads = _train_grads(scene, lo, hi, vsum, vrows, dev, joint, True)
    torch.save({"loss": loss, "grads": grads}, out_path + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("joint", [False, True])
def test_two_rank_training_step_equals_the_single_process_step(tmp_path, joint):
    """Two gloo ranks sharing GPU 0, three scenes each (e7 objective; joint: the encoders and merge_net too): after the
    all-reduce every rank holds the loss and the gradients of the one-process step over all six scenes."""
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    out_path = str(tmp_path / "train")
    port = 27000 + (os.getpid() % 4000)
    mp.spawn(_train_worker, args=(2, port, out_path, joint), nprocs=2, join=True)
    parts = [torch.load(out_path + ".%d" % r) for r in range(2)]
    scene = make_scene_batch(T_BS, K=T_K, S=T_S, seed=78, invalid_lane_frac=0.3, stlp_mode="wide")
    ids = float(sum(scene[k].sum().item() for k in ("curr_id", "left_id", "right_id")))
    loss, grads = _train_grads(scene, 0, T_BS, ids * T_S, T_BS * T_S * 3, torch.device("cuda:0"), joint, False)
    for p in parts:
        assert p["loss"] == parts[0]["loss"]
        np.testing.assert_allclose(p["loss"], loss, rtol=1e-5, atol=1e-6)
        assert set(p["grads"]) == set(grads)
        for k, g in grads.items():
            assert torch.equal(p["grads"][k], parts[0]["grads"][k]), k
            np.testing.assert_allclose(p["grads"][k].numpy(), g.numpy(), rtol=1e-3, atol=1e-5 * float(g.abs().max()), err_msg=k)
