"""Scene sharding across the GPUs of one node (one process per GPU, torch.distributed: RCCL on GPUs, gloo in CPU tests).

Rows interact only inside a scene (merge_net max-pool, scene_acc), so every rank runs the whole
path on a contiguous block of scenes with no data-path collective.  Two tiny exchanges remain:
  * before the rollout: the global sum of valid rows and the global row count (the guidance loss is a mean over the
    WHOLE batch, reference nusc_train.py:23-27,619, so its scale must not depend on how the batch was split);
  * after the final scoring: the 8 integer satisfaction counters and the 12 additive diversity totals
    (pstl_diversity) of each rank -- ONE all-gather of 20 eight-byte words, then summed in rank order.
"""
import torch
import torch.distributed as dist


def shard_range(n_scenes, rank, world):
    """Contiguous block [lo, hi) of scenes for `rank`; blocks differ in size by at most one scene."""
    base, rem = divmod(int(n_scenes), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _active(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def global_valid_stats(valid_sum_local, rows_local, device, group=None):
    """(sum of valid rows, number of rows) over all ranks, as Python floats/ints."""
    t = torch.tensor([float(valid_sum_local), float(rows_local)], dtype=torch.float64, device=device)
    if _active(group):
        dist.all_reduce(t, group=group)
    v = t.tolist()
    return v[0], int(v[1])


def gather_counts(counts, group=None):
    """Sum of the per-rank int64 counter vectors (all-gather + local sum, so every rank holds every shard's numbers)."""
    if not _active(group):
        return counts
    parts = [torch.empty_like(counts) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, counts, group=group)
    return torch.stack(parts).sum(dim=0)


def gather_final(counts, totals, group=None):
    """The final reduction of a sharded run: per-rank satisfaction counters (8 x int64) and diversity totals
    (12 x float64) travel in ONE all-gather (the doubles ride along bit-cast to int64); returns their sums over the
    ranks (floats added in rank order, so every rank gets bit-identical numbers)."""
    if not _active(group):
        return counts, totals
    packed = torch.cat([counts, totals.view(torch.int64)])
    parts = [torch.empty_like(packed) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, packed, group=group)
    allp = torch.stack(parts)
    n = counts.numel()
    tot = allp[:, n:].contiguous().view(torch.float64)
    acc = tot[0].clone()
    for r in range(1, tot.shape[0]):
        acc += tot[r]
    return allp[:, :n].sum(dim=0), acc
