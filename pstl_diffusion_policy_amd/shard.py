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


def plan_rows(n_scenes, world, rows_per_scene):
    """The row count every shard of a job hands to the library as pstl_cfg.plan_rows (SceneBatch(plan_rows=...)): the rows
    of the job's LARGEST shard.  The default arithmetic picks one of two denoiser kernels by batch size (the same products in
    two summation orders); with the job's number instead of each shard's own, every rank -- also a rank that got fewer scenes,
    or none -- runs the kernel the others run, and a row's bits do not depend on which rank holds it."""
    return -(-int(n_scenes) // int(world)) * int(rows_per_scene)


def device_identity(device=None):
    """Two int64 words that tell physical GPUs apart (for the final all-gather of a multi-GPU run: the line it prints can then
    say how many DISTINCT devices took part): the device's UUID where torch reports one, else PCI domain / bus / device, else
    (CPU tests) the process id.  Never initialises a device that is not already in use."""
    import hashlib
    import os
    ident = None
    if device is not None and torch.device(device).type == "cuda":
        p = torch.cuda.get_device_properties(device)
        for attr in ("uuid",):
            if getattr(p, attr, None) is not None:
                ident = "uuid:%s" % getattr(p, attr)
        if ident is None and hasattr(p, "pci_bus_id"):
            ident = "pci:%s:%s:%s" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
        if ident is None:
            ident = "cuda:%d:%s" % (torch.device(device).index or 0, os.environ.get("HIP_VISIBLE_DEVICES", ""))
    if ident is None:
        ident = "host-pid:%d" % os.getpid()
    h = hashlib.sha256(ident.encode()).digest()
    return [int.from_bytes(h[:8], "little", signed=True), int.from_bytes(h[8:16], "little", signed=True)], ident


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


def gather_final(counts, totals, group=None, ident=None, seen=None):
    """The final reduction of a sharded run: per-rank satisfaction counters (8 x int64) and diversity totals
    (12 x float64) travel in ONE all-gather (the doubles ride along bit-cast to int64); returns their sums over the
    ranks (floats added in rank order, so every rank gets bit-identical numbers).
    ident: two int64 words of device_identity() that ride along in the same record (22 words = 176 bytes per rank); `seen`,
    a dict, then receives ranks_seen (records received) and distinct_devices (distinct identities among them) -- device
    tensors until the caller reads them: nothing here synchronises."""
    if not _active(group):
        if seen is not None and ident is not None:
            seen.update(ranks_seen=1, distinct_devices=1)
        return counts, totals
    words = [counts, totals.view(torch.int64)]
    if ident is not None:
        words.append(torch.tensor(ident, dtype=torch.int64, device=counts.device))
    packed = torch.cat(words)
    parts = [torch.empty_like(packed) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, packed, group=group)
    allp = torch.stack(parts)
    n, m = counts.numel(), totals.numel()
    tot = allp[:, n:n + m].contiguous().view(torch.float64)
    acc = tot[0].clone()
    for r in range(1, tot.shape[0]):
        acc += tot[r]
    if seen is not None and ident is not None:
        seen["ranks_seen"] = int(allp.shape[0])
        seen["identities"] = allp[:, n + m:]          # (world, 2) int64: read by the caller after its own synchronisation
    return allp[:, :n].sum(dim=0), acc


def distinct_devices(seen):
    """Number of distinct device identities among the records of the last gather_final(..., seen=seen)."""
    if "identities" not in seen:
        return seen.get("distinct_devices", 1)
    return len({tuple(r) for r in seen["identities"].cpu().tolist()})
