"""N > 1 path on CPU: world_size 2 over gloo.  Covers the sharding of the world batch
and the per-interval counter reduction that bench.py performs over RCCL on GPUs."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from moby_amd import dist as mdist
from moby_amd import scene as S


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = mdist.shard_range(rank, B)
    st = S.sphere_stack_state_range(first, count)
    # fake one interval of counters: each world "solved" (world id + 1) rows
    a0 = np.zeros(count, dtype=S.AUX_DTYPE); a1 = np.zeros(count, dtype=S.AUX_DTYPE)
    a1["lcp_rows"] = np.arange(first, first + count) + 1
    a1["lcp_solves"] = 2
    tot = mdist.counter_vector(a0, a1, np.zeros(count, dtype=bool))
    elapsed, red = mdist.reduce_interval(0.5 + rank, tot, dist)
    out.put((rank, first, float(st[0, 0]), float(st[-1, 9]), elapsed, red.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    world, B = 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = S.sphere_stack_state_range(0, world * B)
    for rank, first, x0, vlast, elapsed, red in res:
        assert first == rank * B                                        # contiguous shards, no overlap
        assert x0 == full[first, 0] and vlast == full[first + B - 1, 9]  # each rank builds exactly its worlds
        assert elapsed == 1.5                                           # MAX over ranks
        assert red[0] == sum(range(1, world * B + 1))                   # SUM of rows over all worlds
        assert red[1] == 2 * world * B
