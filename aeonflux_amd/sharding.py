"""Host-side sharding of one batch across the GPUs of a node: one process per GPU, contiguous
item ranges, no collective on the data path — only the 1-byte-per-item status gather at the end."""
import numpy as np


def shard_bounds(count, world, rank):
    """contiguous [lo, hi) of rank's share; shares differ by at most one item"""
    base, rem = divmod(count, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slice_presentation(p, lo, hi):
    """the [lo, hi) items of a presentation dict in the SoA layout ([..., count, 32] arrays)"""
    cut = lambda a: np.ascontiguousarray(a[..., lo:hi, :])
    out = {k: cut(v) for k, v in p.items() if k != "enc"}
    out["enc"] = [{k: cut(v) for k, v in d.items()} for d in p.get("enc", [])]
    return out


def verify_sharded(verify_fn, shape, presentation, count, rank, world, group=None):
    """Every rank holds the host batch, verifies its own range with verify_fn(shape, shard) -> uint8[hi-lo], and
    all ranks receive the full status vector.  With world == 1 no process group is needed."""
    lo, hi = shard_bounds(count, world, rank)
    local = np.asarray(verify_fn(shape, slice_presentation(presentation, lo, hi)), dtype=np.uint8)
    assert local.shape == (hi - lo,)
    if world == 1:
        return local
    import torch
    import torch.distributed as dist
    sizes = [shard_bounds(count, world, r)[1] - shard_bounds(count, world, r)[0] for r in range(world)]
    pad = max(sizes)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.zeros(pad, dtype=torch.uint8, device=dev)
    mine[:hi - lo] = torch.from_numpy(local).to(dev)
    parts = [torch.zeros(pad, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return np.concatenate([parts[r][:sizes[r]].cpu().numpy() for r in range(world)])
