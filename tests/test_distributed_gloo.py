"""N > 1 path on the CPU: two gloo processes, each owning a contiguous block of scenes.  Checks that the exchanged
quantities (global valid statistics before the rollout, satisfaction counters after the final scoring) reproduce the
single-process result exactly, using the CPU oracle as the per-shard scorer."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _counts_from_scores(scores, valid, bs, S):
    sat = int(((scores > 0) & (valid > 0)).sum())
    cube = scores.reshape(bs, S, 3)
    v0 = valid.reshape(bs, S, 3)[:, 0, :]
    ssat = int(((cube.max(dim=1)[0] > 0) & (v0 > 0)).sum())
    return torch.tensor([sat, int(valid.sum()), bs * S * 3, ssat, int(v0.sum()), bs * 3, 0, 0], dtype=torch.int64)


def _div_totals(o, lane_valid, nt=20):
    """The 12 additive totals of pstl_diversity (include/pstl_hip.h) from the oracle's per-(scene,mode) arrays."""
    v = lane_valid.numpy().astype(np.float64)
    o = {k: np.asarray(x, dtype=np.float64) for k, x in o.items() if np.ndim(x) > 0}
    t = [float((o["std_sm"] * v).sum()), float((o["vol_sm"] * v).sum()), float(v.sum()), float(o["ent_s_sm"].sum()),
         float(o["ent_w_sm"].sum()), float(o["ent_a_sm"].sum()), float(o["area_sm"].sum()), float(v.size),
         float(o["ade_s"].sum()), float(o["fde_s"].sum()), float(v.shape[0]), 0.0]
    return torch.tensor(t, dtype=torch.float64)


def _oracle_div(scene, u, score, valid_rows, S, hp):
    from oracle import diversity_oracle as dorc
    from oracle import pstl_oracle as orc
    ego = torch.as_tensor(scene["ego_traj"])
    lane_valid = torch.cat([torch.as_tensor(scene[k]) for k in ("curr_id", "left_id", "right_id")], dim=-1)
    o = dorc.all_metrics(ego[:, 0, :4], ego, u, score, valid_rows, S, hp, orc.unicycle_rollout)
    return o, _div_totals(o, lane_valid)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.shard import gather_counts, gather_final, global_valid_stats, shard_range
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    hp = default_hparams()
    bs, S, K = 7, 4, 3
    scene = make_scene_batch(bs, K=K, S=S, seed=31, invalid_lane_frac=0.4, stlp_mode="wide")
    lo, hi = shard_range(bs, rank, world)
    sub = {k: v[lo:hi].numpy() for k, v in scene.items()}
    rows = orc.Rows(sub, S, hp)
    g = torch.Generator().manual_seed(9)
    u_all = torch.randn(bs * S * 3, 20, 2, generator=g) * torch.tensor([0.05, 0.5])
    u = u_all[lo * S * 3:hi * S * 3]
    vsum, vrows = global_valid_stats(float(rows.valid.sum()), rows.N, torch.device("cpu"))
    _, score, _ = rows.score(u)
    local_counts = _counts_from_scores(score, rows.valid, hi - lo, S)
    counts = gather_counts(local_counts)
    _, tot = _oracle_div(sub, u, score, rows.valid, S, hp)
    counts2, totals = gather_final(local_counts, tot)
    assert torch.equal(counts, counts2)
    # the same reduction with the rank's identity riding along (bench.py's self-proving line): the sums do not change, every
    # rank sees one record per rank and -- two processes here -- two distinct identities
    from pstl_diffusion_policy_amd.shard import device_identity, distinct_devices
    seen = {}
    counts3, totals3 = gather_final(local_counts, tot, ident=device_identity(None)[0], seen=seen)
    assert torch.equal(counts3, counts2) and torch.equal(totals3, totals)
    assert seen["ranks_seen"] == world and distinct_devices(seen) == world
    if rank == 0:
        torch.save({"vsum": vsum, "vrows": vrows, "counts": counts, "totals": totals}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_shards_reproduce_the_single_process_numbers(tmp_path):
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.engine import acc_from_counts
    from pstl_diffusion_policy_amd.shard import shard_range
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    assert [shard_range(7, r, 2) for r in range(2)] == [(0, 4), (4, 7)]
    assert [shard_range(4096, r, 8) for r in range(8)][-1] == (3584, 4096)
    out_path = str(tmp_path / "r0.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    got = torch.load(out_path)
    hp = default_hparams()
    bs, S, K = 7, 4, 3
    scene = {k: v.numpy() for k, v in make_scene_batch(bs, K=K, S=S, seed=31, invalid_lane_frac=0.4, stlp_mode="wide").items()}
    rows = orc.Rows(scene, S, hp)
    g = torch.Generator().manual_seed(9)
    u = torch.randn(bs * S * 3, 20, 2, generator=g) * torch.tensor([0.05, 0.5])
    _, score, _ = rows.score(u)
    want = _counts_from_scores(score, rows.valid, bs, S)
    assert got["vsum"] == float(rows.valid.sum()) and got["vrows"] == rows.N
    assert torch.equal(got["counts"], want)
    acc, sacc = orc.stl_metrics(score, rows.valid, S)
    a, s = acc_from_counts(got["counts"])
    assert a == float(acc) and s == float(sacc)
    # the diversity half of the final reduction: per-shard totals add up to the single-process numbers
    from pstl_diffusion_policy_amd.engine import diversity_from_totals
    o, want_tot = _oracle_div(scene, u, score, rows.valid, S, hp)
    np.testing.assert_allclose(got["totals"].numpy(), want_tot.numpy(), rtol=1e-12)
    d = diversity_from_totals(got["totals"])
    for k in ("std", "vol", "ade", "fde", "ent_s", "ent_w", "ent_a", "area"):
        assert d[k] == pytest.approx(o[k], rel=1e-5), k


def _worker8(rank, world, port, out_path, bs):
    """A rank of a world-8 job over `bs` scenes (bs < world leaves ranks with an EMPTY shard, which must still join both
    exchanges and contribute zeros)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.shard import gather_final, global_valid_stats, shard_range
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    hp = default_hparams()
    S, K = 4, 3
    scene = make_scene_batch(bs, K=K, S=S, seed=31, invalid_lane_frac=0.4, stlp_mode="wide")
    lo, hi = shard_range(bs, rank, world)
    g = torch.Generator().manual_seed(9)
    u_all = torch.randn(bs * S * 3, 20, 2, generator=g) * torch.tensor([0.05, 0.5])
    if hi > lo:
        sub = {k: v[lo:hi].numpy() for k, v in scene.items()}
        rows = orc.Rows(sub, S, hp)
        u = u_all[lo * S * 3:hi * S * 3]
        vsum, vrows = global_valid_stats(float(rows.valid.sum()), rows.N, torch.device("cpu"))
        _, score, _ = rows.score(u)
        local_counts = _counts_from_scores(score, rows.valid, hi - lo, S)
        _, tot = _oracle_div(sub, u, score, rows.valid, S, hp)
    else:
        vsum, vrows = global_valid_stats(0.0, 0, torch.device("cpu"))
        local_counts = torch.zeros(8, dtype=torch.int64)
        tot = torch.zeros(12, dtype=torch.float64)
    counts, totals = gather_final(local_counts, tot)
    torch.save({"vsum": vsum, "vrows": vrows, "counts": counts, "totals": totals, "range": (lo, hi)}, out_path + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bs", [7, 19])
def test_eight_ranks_including_empty_shards(tmp_path, bs):
    """The N = 8 rank path before a real 8-GPU box runs it: shard_range / global_valid_stats / gather_final under gloo with
    world size 8; with 7 scenes one rank owns nothing (shard_range(7, 7, 8) == (7, 7)) and must still join the all-reduce and
    the all-gather.  Every rank ends with the single-process numbers."""
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.shard import shard_range
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    world = 8
    ranges = [shard_range(bs, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == bs and all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
    if bs == 7:
        assert ranges[7] == (7, 7)
    out_path = str(tmp_path / "r")
    port = 29500 + ((os.getpid() + bs) % 2000)
    mp.spawn(_worker8, args=(world, port, out_path, bs), nprocs=world, join=True)
    hp = default_hparams()
    S, K = 4, 3
    scene = {k: v.numpy() for k, v in make_scene_batch(bs, K=K, S=S, seed=31, invalid_lane_frac=0.4, stlp_mode="wide").items()}
    rows = orc.Rows(scene, S, hp)
    g = torch.Generator().manual_seed(9)
    u = torch.randn(bs * S * 3, 20, 2, generator=g) * torch.tensor([0.05, 0.5])
    _, score, _ = rows.score(u)
    want = _counts_from_scores(score, rows.valid, bs, S)
    _, want_tot = _oracle_div(scene, u, score, rows.valid, S, hp)
    for r in range(world):
        got = torch.load(out_path + ".%d" % r)
        assert got["range"] == ranges[r]
        assert got["vsum"] == float(rows.valid.sum()) and got["vrows"] == rows.N
        assert torch.equal(got["counts"], want), r
        np.testing.assert_allclose(got["totals"].numpy(), want_tot.numpy(), rtol=1e-12)
        if r > 0:      # floats added in rank order: every rank holds bit-identical totals
            assert torch.equal(got["totals"], torch.load(out_path + ".0")["totals"])
