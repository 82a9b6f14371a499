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


class _FakeBatch:
    """Stand-in for the device batch of bench.py's strong_leg: world w "solves" one LCP of (w + 1) rows per step."""
    def __init__(self, first, count):
        self.ids = np.arange(first, first + count); self.aux = np.zeros(count, dtype=S.AUX_DTYPE)
    def step(self, dt, n, stream=None):
        self.aux["lcp_rows"] += (self.ids + 1).astype(np.uint64) * np.uint64(n); self.aux["lcp_solves"] += np.uint64(n)
    def download(self):
        return None, self.aux.copy()
    def close(self):
        pass


def _strong_worker(rank, world, port, B_total, out):
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r = bench.strong_leg(dist, mdist, S, _FakeBatch, B_total, rank, world, None, steps=3, warmup=2, sync=lambda: None)
    out.put((rank, r))
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_leg_over_gloo():
    """bench.py's strong_leg itself (range split, gathered ranges, barriers, MAX / SUM reduction, the merged object) on two gloo
    ranks: the ranges tile [0, B) exactly once, every rank reports the same merged figures, n_gpus and worlds_per_gpu are right."""
    world, B_total = 2, 4097                      # odd: the remainder goes to rank 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, B_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["rank_ranges"] == res[1]["rank_ranges"] == [(0, 2049), (2049, 2048)]
    covered = np.zeros(B_total, dtype=int)
    for f, c in res[0]["rank_ranges"]:
        covered[f:f + c] += 1
    assert (covered == 1).all()
    for rank in (0, 1):
        r = res[rank]
        assert r["n_gpus"] == 2 and r["worlds_total"] == B_total and r["scaling"] == "strong"
        assert r["worlds_per_gpu"] == (2049 if rank == 0 else 2048)
        assert r["lcp_rows"] == 3 * sum(range(1, B_total + 1)) and r["lcp_solves"] == 3 * B_total      # timed steps only, all worlds
    assert res[0]["value"] == res[1]["value"] and res[0]["ms_per_step"] == res[1]["ms_per_step"]     # MAX / SUM: one answer


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
    import pytest
    if torch.cuda.is_available():
        pytest.skip("on a GPU box the scaling run itself is the driver's job")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "needs a GPU" in p.stderr and "WORLD_SIZE=1" not in p.stderr, p.stderr[-2000:]
