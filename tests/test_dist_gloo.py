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


def test_split_range_covers_the_batch_once():
    """Strong scaling (one batch split N ways): contiguous, disjoint, complete; remainder on the lowest ranks."""
    for total in (4096, 4097, 7, 1):
        for n in (1, 2, 3, 8):
            got = [mdist.split_range(r, n, total) for r in range(n)]
            assert got[0][0] == 0 and sum(c for _, c in got) == total
            for (f0, c0), (f1, _) in zip(got, got[1:]):
                assert f1 == f0 + c0
            assert max(c for _, c in got) - min(c for _, c in got) <= 1


def test_bench_launches_its_own_ranks_when_started_as_plain_python():
    """`python bench.py --gpus 2` with no RANK in the environment must start torch.distributed.run as a child (the
    driver's scaling run does exactly this).  Without a GPU every rank stops at "needs a GPU" -- after the rendezvous
    environment was set up, which is what is checked here."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        return      # on a GPU box the scaling run itself is the driver's job
    assert p.returncode != 0
    assert "needs a GPU" in p.stderr and "WORLD_SIZE=1" not in p.stderr, p.stderr[-2000:]
