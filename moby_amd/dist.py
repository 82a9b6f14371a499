"""Multi-GPU sharding of a world batch: worlds are independent, so rank r of
an N-rank job owns the contiguous world range [r*B, (r+1)*B) (weak scaling) and the only
communication is one small all-reduce of counters per reporting interval."""
import numpy as np

COUNTER_FIELDS = ("lcp_rows", "lcp_solves", "lcp_pivots", "mini_steps", "stab_iters", "lcp_alg_bytes", "stab_rows")


def shard_range(rank, worlds_per_rank):
    """First world id and count owned by `rank`."""
    return rank * worlds_per_rank, worlds_per_rank


def split_range(rank, world_size, total):
    """Strong scaling: first world id and count of `rank`'s contiguous share of ONE batch of `total` worlds
    ([g B/G, (g+1) B/G) of SURVEY 8e; the remainder goes to the lowest ranks)."""
    base, rem = divmod(total, world_size)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def counter_vector(aux_before, aux_after, bad_mask):
    """Per-rank totals over an interval: deltas of the counter fields + number of failed worlds."""
    d = [float(aux_after[f].astype(np.int64).sum() - aux_before[f].astype(np.int64).sum()) for f in COUNTER_FIELDS]
    d.append(float(int(bad_mask.sum())))
    return np.array(d, dtype=np.float64)


def reduce_interval(elapsed, totals, dist=None, device=None):
    """MAX of the elapsed time and SUM of the counters over all ranks (no-op without a process group)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed, totals
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.from_numpy(np.asarray(totals, dtype=np.float64)).to(device) if device is not None else torch.from_numpy(np.asarray(totals, dtype=np.float64))
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), c.cpu().numpy()
