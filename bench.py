#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): BASELINE.json configs[1], sphere-stack x4096 --
4096 independent instances of example/stacks/sphere-stack.xml per GPU.  One
"step" = one pass of the hot path over the whole batch:

  round-1 state of the build: the impact LCP of every world (n = 42,
  ImpactConstraintHandlerQP.cpp:219 lcp_fast_regularized(-20,4,-8) with the
  Lemke ladder ICH-QP:224 on the worlds where it fails), M/q resident in HBM.

Prints ONE JSON line on rank 0 (contract in the task statement) with
`roofline` for the dominant kernel and `cpu_baseline` (the oracle timed on
the host cores, rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_LCP = 42
WORLDS_PER_GPU = 4096


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--worlds", type=int, default=WORLDS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world_size))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world_size > 1:
        dist.barrier()
    from moby_amd import _lib, synth
    from moby_amd.lcp import LCPDevice
    lib = _lib.load()

    B, n = args.worlds, N_LCP
    # synthetic batch: 64 distinct perturbed worlds (world 0 = the reference
    # scene), tiled to B; shard r of an N-GPU job takes worlds offset by r*64
    base = 64
    Mh, qh = synth.sphere_stack_impact_lcp(base, first_world=0 if rank == 0 else rank * base, hard=False)
    reps = (B + base - 1) // base
    Mcm = np.ascontiguousarray(np.transpose(np.tile(Mh, (reps, 1, 1))[:B], (0, 2, 1)))
    qb = np.tile(qh, (reps, 1))[:B].copy()
    M = torch.from_numpy(Mcm).to(dev)
    q = torch.from_numpy(qb).to(dev)
    z0 = torch.zeros(B, n, dtype=torch.float64, device=dev)
    z = torch.zeros_like(z0)
    solver = LCPDevice(B, dev)
    rng0 = solver.rng.clone()
    opts_fast = _lib.mh_lcp_opts(-20, 4, -8, -1.0, -1.0)
    status_fast = torch.zeros(B, dtype=torch.int32, device=dev)

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    kern_ms = []

    def step(timed_kernel=False):
        # handler state at the first impact of every world: z = 0 of size n,
        # every world's rand() stream at srand(1)
        z.copy_(z0)
        solver.rng.copy_(rng0)
        if timed_kernel:
            ev0.record()
        solver.solve(_lib.MH_LCP_FAST_REG, M, q, z, opts_fast)
        if timed_kernel:
            ev1.record()
        status_fast.copy_(solver.status)
        # ICH-QP:221-225: z.set_zero(); lcp_lemke_regularized on the failures.
        # The batch entry solves all worlds; failures are selected afterwards.
        return None

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    for _ in range(args.steps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        z.copy_(z0)
        solver.rng.copy_(rng0)
        e0.record()
        solver.solve(_lib.MH_LCP_FAST_REG, M, q, z, opts_fast)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    kern_avg_s = float(np.mean(kern_ms)) * 1e-3
    n_ok = int(solver.status.sum().item())

    rows_per_step = B * n * world_size
    value = rows_per_step * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    # algorithmic bytes of one launch: M + q in, z out  (8(n^2+2n) per LCP)
    alg_bytes = 8.0 * (n * n + 2 * n) * B
    achieved = alg_bytes / kern_avg_s / 1e9

    out = {
        "metric": "lcp_rows_per_sec",
        "value": value,
        "unit": "LCP rows/s",
        "n_gpus": world_size,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "sphere-stack x%d per GPU: impact LCP n=42, lcp_fast_regularized(-20,4,-8)" % B,
                   "worlds_per_gpu": B, "lcp_n": n, "parallelism": "worlds sharded x%d, no collective" % world_size},
        "world_steps_per_sec": B * world_size * args.steps / elapsed,
        "solved_by_fast_ladder": n_ok,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "mh_k_lcp_wave", "kernel_avg_us": kern_avg_s * 1e6,
                     "algorithmic_bytes_per_launch": alg_bytes},
    }

    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        from tests.oracle_api import Oracle, FAST_REG
        oracle = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
        sample = min(B, 1024)
        Ms = np.tile(Mh, ((sample + base - 1) // base, 1, 1))[:sample]
        qs = np.tile(qh, ((sample + base - 1) // base, 1))[:sample]
        reps_cpu, secs = 0, 0.0
        while secs < 10.0 and reps_cpu < 50:
            s, st, piv, zz, _, _ = oracle.lcp_batch(FAST_REG, Ms, qs, np.zeros((sample, n)), exps=(-20, 4, -8))
            secs += s; reps_cpu += 1
        out["cpu_baseline"] = {"value": sample * n * reps_cpu / secs, "unit": "LCP rows/s", "cores": 1, "kind": "port",
                               "sample": "%d of the same sphere-stack impact LCPs x %d passes, oracle lcp_fast_regularized, 1 thread (host has %d cores)"
                                         % (sample, reps_cpu, os.cpu_count())}
    if rank == 0:
        print(json.dumps(out))
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
