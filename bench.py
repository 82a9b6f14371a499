#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): BASELINE.json configs[1] -- sphere-stack x4096:
4096 independent instances of /root/reference/example/stacks/sphere-stack.xml
per GPU, dt = 1e-3 (world 0 of rank 0 is the reference scene, the others are
perturbed as SURVEY 8d.2 says).  One "step" = TimeSteppingSimulator::step(dt)
on every world of the batch: broad phase, conservative-advancement mini-steps,
forward dynamics, contact generation, the impact QP->LCP (n = 42 / 14) with
lcp_fast_regularized and the Lemke ladder, restitution, constraint
stabilisation.  The K timed steps run inside ONE persistent launch (state never
leaves the GPU between steps); inputs are resident in HBM before the clock starts.

N > 1 started as plain `python bench.py --gpus N ...` (no RANK in the environment) re-launches itself under
`python -m torch.distributed.run` as a CHILD process before anything touches the GPU and returns the child's exit code.
`scaling` is "weak" (`--worlds` per GPU, the headline `value`); for N > 1 the same line also carries `strong_scaling`:
ONE batch of `--worlds` split N ways (4096 / 8 = 512 worlds per GPU = 2 waves per CU), timed the same way.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel) and
`cpu_baseline` (the CPU oracle on the host cores; rank 0, N = 1 only).
Informational legs (rank 0, N = 1, outside the timed region): `config4_impact_handler` -- 4-box stacks x1024 through the
impact-handler entry (include/moby_hip_impact.h), one cold and one warm call; `config4_full_step` -- box stacks as full
simulator steps (include/moby_hip_stack.h); `config5_ur10` -- the ur10 arm x8192 (include/moby_hip_artic.h);
`config2_full_run` -- the headline batch's whole
1000-step run from t = 0; `long_horizon` -- the headline batch at steps 4000-4200.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
WORLDS_PER_GPU = 4096
DT = 1e-3


def pmc_traffic(B, steps):
    """HBM bytes of one launch of B worlds x `steps` steps from the committed rocprofv3 PMC passes of this same command
    (profiles/pmc_traffic.json: FETCH_SIZE + WRITE_SIZE of the timed launch, separate passes).  Counters cannot be read
    in-process, so this is a PROFILE of the build named in the file's `commit`, reported only when this run has the profile's
    batch size and step count (None otherwise: no extrapolation)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if int(t["worlds"]) != int(B) or int(t["steps"]) != int(steps):
            return None
        return (t["fetch_size_kb"] + t["write_size_kb"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def pmc_profile_tag():
    """Which build the committed counter profiles (roofline.traffic / roofline.issue) were cut from."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return {"commit": t.get("commit"), "worlds": t.get("worlds"), "steps": t.get("steps"), "source": t.get("source"),
                "fetch_bytes": t["fetch_size_kb"] * 1024.0, "write_bytes": t["write_size_kb"] * 1024.0}
    except (OSError, KeyError, ValueError):
        return None


def pmc_issue():
    """Issue-slot figures of the same kernel from the committed SQ_* counter passes (profiles/pmc_issue.json): the share of
    SIMD cycles with the vector ALU occupied, the wave-cycle breakdown, instructions per world-step.  The kernel is
    issue / LDS-latency bound, so this -- not the HBM fraction -- is how far it is from its own roofline."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_issue.json")))
        return {"valu_busy_frac": t["valu_busy_frac"], "wave_cycles": t["wave_cycles"], "per_world_step": t["per_world_step"],
                "source": "profiles/pmc_issue.json"}
    except (OSError, KeyError, ValueError):
        return None


def _cpu_worker(arg):
    """One process of the all-cores CPU baseline: `passes` x (nw worlds x ns steps) of the oracle."""
    first, nw, ns, passes = arg
    from moby_amd import scene as S
    from tests.oracle_api import Oracle
    oracle = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    sc = S.sphere_stack_scene()
    rows = 0
    for _ in range(passes):
        st = S.sphere_stack_state_range(first, nw)
        aux = S.new_aux(nw)
        oracle.world_step_batch(sc, st, aux, DT, ns)
        rows += int(aux["lcp_rows"].sum())
    return rows, nw * ns * passes


def usable_cpus():
    """CPUs this process may really use: the affinity mask, cut down to the cgroup CPU quota when
    one is set (os.cpu_count() reports the host's 256 hardware threads inside an 8-CPU container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(B, steps, warmup):
    """The CPU oracle (oracle/world.hpp, `kind: port`) on the host cores, on a bounded sample of the
    same workload: one thread (the reference is single-threaded), and -- for scale -- one process
    per core, each stepping its own worlds.  Runs BEFORE anything touches the GPU (the worker pool
    forks)."""
    import multiprocessing as mp
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    from moby_amd import scene as S
    from tests.oracle_api import Oracle
    oracle = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    sc = S.sphere_stack_scene()
    nw, ns = min(B, 256), min(steps + warmup, 200)
    secs, crow, cstep, passes = 0.0, 0, 0, 0
    while secs < 8.0 and passes < 40:
        st = S.sphere_stack_state_range(0, nw)
        aux = S.new_aux(nw)
        secs += oracle.world_step_batch(sc, st, aux, DT, ns)
        crow += int(aux["lcp_rows"].sum()); cstep += nw * ns; passes += 1
    out = {"value": crow / secs, "unit": "LCP rows/s", "cores": 1, "kind": "port", "world_steps_per_sec": cstep / secs,
           "sample": "%d worlds x %d steps of the same batch x %d passes, CPU oracle (oracle/world.hpp), 1 thread; host has %d cores"
                     % (nw, ns, passes, os.cpu_count())}
    ncores = int(os.environ.get("MH_BENCH_CPU_PROCS", usable_cpus()))
    if ncores > 1:
        per = max(1, min(32, B // ncores if B >= ncores else 1))
        want = max(1, int(4.0 * (cstep / secs) / (per * ns)))           # ~4 s of work per process
        jobs = [(i * per, per, ns, min(want, 64)) for i in range(ncores)]
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(ncores) as pool:
            res = pool.map(_cpu_worker, jobs)
        wall = time.perf_counter() - t0
        out["all_cores"] = {"value": sum(r[0] for r in res) / wall, "unit": "LCP rows/s", "cores": ncores,
                            "world_steps_per_sec": sum(r[1] for r in res) / wall,
                            "sample": "%d processes (usable CPUs of %d hardware threads) x %d worlds x %d steps x %d passes, wall clock incl. process start"
                                      % (ncores, os.cpu_count() or 1, per, ns, jobs[0][3])}
    # for context only (SURVEY 8d iii): the timing footer of the reference's own recording of this scene, ONE world, unknown hardware
    out["reference_recording"] = {"world_steps_per_sec": 1000.0 / 0.408651, "source": "regress/sphere-stack.dat:1001 (0.408651 s per 1000 steps)"}
    return out


def config4_leg(torch, nboxes=4, B=1024):
    """BASELINE config 4 in the small: B stacks of `nboxes` boxes through the impact-handler entry
    (include/moby_hip_impact.h), one cold and one warm-started process_constraints call on every world, timed with
    events on the launch stream.  Not part of `value`; never allowed to break the headline line."""
    try:
        from moby_amd import impact as I
        mass, J, st, cs = I.box_stack(nboxes, B=B)
        ib = I.ImpactBatch(B, nboxes, 4 * nboxes, 4, mass, J)
        ib.upload(st, cs)
        stream = torch.cuda.current_stream().cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        res = {"workload": "box stack of %d (n = %d) x%d worlds, 4 corner contacts per interface" % (nboxes, ib.n, B)}
        for tag in ("cold", "warm"):
            e0.record(); ib.process_async(stream); e1.record(); torch.cuda.synchronize()
            r = ib.download()
            ms = e0.elapsed_time(e1)
            res[tag] = {"ms": ms, "lcp_rows_per_sec": ib.n * float(r["solves"].sum()) / (ms * 1e-3),
                        "pivots_mean": float(r["pivots"].mean()), "pivots_max": int(r["pivots"].max()),
                        "worlds_with_errors": int(((r["status"] & ~2) != 0).sum())}
            w = ib.lu_work(reset=True).sum(axis=0)
            res[tag]["roofline"] = {"bound": "fp64_valu", "achieved": w[0] / (ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": w[0] / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "traffic": None,
                                    "model_flops": float(w[0]), "executed_flops": float(w[2]), "executed_tflops": w[2] / (ms * 1e-3) / 1e12,
                                    "hbm": {"achieved": w[1] / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": w[1] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                    "model": "achieved = MODEL flops (one dense dgesv per pivot: 2/3 k^3 flops, 8 k^2 bytes, SURVEY 8d) over time; executed_flops = what the "
                                             "factorisation routines issue (no MFMA: FP64 vector ALU, unfused), both counted on the device"}
            s2 = r["state"].reshape(B, nboxes, 13); s2[:, :, 8] += -9.81e-3      # gravity acts for another dt
            ib.upload(s2.reshape(B, -1), cs)
        ib.close()
        return res
    except Exception as e:          # noqa: BLE001 -- informational leg
        return {"error": repr(e)}


FP64_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: FP64 vector = FP64 matrix peak (fused multiply-add); this build runs unfused (parity), i.e. against half of it


def _cpu_stack_worker(arg):
    nboxes, w, batch = arg
    from moby_amd import scene as S, stack as K
    from tests.oracle_api import Oracle
    o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    sc = K.box_stack_scene(nboxes); st = K.box_stack_state(nboxes, batch)
    s = st[w].copy(); aux = S.new_aux(1)
    secs = o.big_step(sc, s, aux, DT, 1)["seconds"]
    return secs, int(aux["lcp_rows"][0]), int(aux["lcp_pivots"][0])


def config4_cpu_sample(nboxes, batch, world=2):
    """The CPU side of the config-4 leg, BEFORE the GPU is touched (round 4 ran it beside the GPU leg and cost that ~10 %): one process per usable
    core, each stepping the SAME world of the batch -- world 2, a perturbed stack that walks the whole solver chain with 12.5 k pivots in its cold step (the batch's mean is
    11.5 k) -- through one full cold step with the CPU oracle.  One world, because a cold step of a 16-box world costs the oracle 25 s to 4 min depending on the world
    (measured on worlds 1-9: profiles/r05_f_bench.json and DESIGN 5), and a pool over different worlds lasts as long as its slowest (120 s with worlds 1-4).
    `value`: the mean single-thread rate of those processes (each is one thread on its own core); `all_cores`: all of them over the wall clock."""
    import multiprocessing as mp
    try:
        ncores = int(os.environ.get("MH_BENCH_CPU_PROCS", usable_cpus()))
        w = min(world, max(0, batch - 1))
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(ncores) as pool:
            res = pool.map(_cpu_stack_worker, [(nboxes, w, batch)] * ncores)
        wall = time.perf_counter() - t0
        secs = sum(r[0] for r in res); rows = sum(r[1] for r in res); piv = sum(r[2] for r in res)
        return {"value": rows / secs, "unit": "LCP rows/s", "world_steps_per_sec": len(res) / secs, "cores": 1, "kind": "port",
                "sample": "world %d of the same batch (box stack of %d), one full cold step, CPU oracle (oracle/world.hpp), one thread: %.1f s, %d pivots (mean of %d processes, "
                          "each on its own core, before the GPU legs)" % (w, nboxes, secs / len(res), piv // len(res), len(res)),
                "all_cores": {"value": rows / wall, "unit": "LCP rows/s", "world_steps_per_sec": len(res) / wall, "cores": ncores,
                              "sample": "the same %d processes over the wall clock (%.1f s incl. process start)" % (ncores, wall)}}
    except Exception as e:          # noqa: BLE001
        return {"error": repr(e)}


def config4_cpu_sample_start(nboxes, worlds=2):
    """(round 4's form, kept for tools: the oracle in a CHILD process beside the GPU legs)"""
    import subprocess
    code = ("import sys, json, os; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from tests.oracle_api import Oracle\n"
            "from moby_amd import scene as S, stack as K\n"
            "o = Oracle(os.path.join(%r, 'oracle', 'liboracle.so'))\n"
            "N, W = %d, %d\n"
            "sc = K.box_stack_scene(N); st = K.box_stack_state(N, max(W, 2))\n"
            "secs = 0.0; rows = 0; piv = 0\n"
            "for w in range(W):\n"
            "    s = st[w].copy(); aux = S.new_aux(1)\n"
            "    secs += o.big_step(sc, s, aux, 1e-3, 1)['seconds']; rows += int(aux['lcp_rows'][0]); piv += int(aux['lcp_pivots'][0])\n"
            "print(json.dumps({'secs': secs, 'rows': rows, 'pivots': piv, 'worlds': W}))\n") % (ROOT, ROOT, nboxes, worlds)
    try:
        return subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    except OSError:
        return None


def config4_cpu_sample_collect(proc, nboxes, timeout=400):
    if proc is None:
        return None
    try:
        o, _ = proc.communicate(timeout=timeout)
        r = json.loads(o.strip().splitlines()[-1])
        return {"value": r["rows"] / r["secs"], "unit": "LCP rows/s", "world_steps_per_sec": r["worlds"] / r["secs"], "cores": 1, "kind": "port",
                "sample": "worlds 0..%d of the same batch (box stack of %d), one full step each, CPU oracle (oracle/world.hpp), 1 thread, "
                          "beside the GPU legs: %.1f s, %d pivots" % (r["worlds"] - 1, nboxes, r["secs"], r["pivots"])}
    except Exception as e:          # noqa: BLE001
        proc.kill()
        return {"error": repr(e)}


def config4_full_step_leg(torch, nboxes, B, steps, cpu_proc=None, cpu_part=None, note=None):
    """BASELINE config 4 as FULL simulator steps (include/moby_hip_stack.h) at the bench size (16 boxes, n = 512;
    tests/test_big_gpu.py::test_config4_bench_size_full_batch): B stacks of `nboxes` boxes, each step = conservative advancement
    + contact generation + process_constraints over every island + stabilisation, all on the device.  The first step is cold.
    `roofline`: the solver chain's factorisations priced as SURVEY 8(d) does -- one dgesv per pivot (LCP.cpp:120, :837-838),
    2/3 k^3 flops over 8 k^2 bytes, counted on the device (mh_big_batch_lu_work) -- over the wall time of the step (the block
    solver's kernels are 98 % of it; one step is hundreds of launches, so there is no single launch to time)."""
    try:
        from moby_amd import stack as K
        sc = K.box_stack_scene(nboxes)
        bb = K.BigBatch(sc, K.box_stack_state(nboxes, B))
        res = {"workload": "box stack of %d (impact LCP n = %d) x%d worlds, full TimeSteppingSimulator::step, dt = 1e-3" % (nboxes, 32 * nboxes, B) + (("; " + note) if note else ""),
               "steps": []}
        prev = None; total_s = 0.0
        for k in range(steps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); bb.step(DT, 1); torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
            total_s += dt_s
            _, aux = bb.download()
            rows = float(aux["lcp_rows"].astype(np.int64).sum()) - (prev if prev is not None else 0.0)
            prev = float(aux["lcp_rows"].astype(np.int64).sum())
            res["steps"].append({"s": dt_s, "world_steps_per_sec": B / dt_s, "lcp_rows_per_sec": rows / dt_s})
        wk = bb.lu_work()
        work = wk.sum(axis=0)
        res["worlds_with_errors"] = int(((aux["status"] & ~2) != 0).sum())
        res["worlds_lcp_solver_exception"] = int(((aux["status"] & 1) != 0).sum())     # MH_WORLD_LCP_FAILED: the reference's whole solver chain fails (LCPSolverException, ICH-QP:225)
        # how busy the solver kept the chip: seconds workgroups spent on problems (every ladder attempt) over slots x wall time.  Slots: what the
        # Lemke kinds' kernel holds at once (the lcp_fast kinds' kernel holds one 1024-thread problem per CU: priced against the larger number)
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        slots = (4 if B >= 4 * cus else 2) * cus        # the ladder's tasks: four 128-thread problems per CU from 4 worlds per CU up, else two 256-thread ones (mh_host.h MH_BLK2_MIN_PER_CU)
        res["idle"] = {"solver_workgroup_seconds": float(work[3]), "wall_seconds": total_s, "mean_busy_workgroups": float(work[3]) / total_s, "slots": slots,
                       "idle_frac": 1.0 - float(work[3]) / (total_s * slots),
                       "slowest_world_solver_seconds": float(wk[:, 3].max()), "mean_world_solver_seconds": float(wk[:, 3].mean())}
        res["pivots_per_world_step"] = float(aux["lcp_pivots"].astype(np.int64).mean()) / steps
        res["world_steps_per_sec"] = B * steps / total_s
        res["lcp_rows_per_sec"] = prev / total_s
        tf = work[0] / total_s / 1e12; gbs = work[1] / total_s / 1e9
        res["roofline"] = {"bound": "fp64_valu", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                           "traffic": None, "kernel": "mh::blk::k_lcp_block<1> / <0> (the workgroup-per-problem LCP solver: lcp_lemke / lcp_fast kinds)",
                           "seconds": total_s, "model_flops": float(work[0]), "model_bytes": float(work[1]),
                           "executed_flops": float(work[2]), "executed_tflops": float(work[2]) / total_s / 1e12, "executed_frac_of_unfused_peak": float(work[2]) / total_s / 1e12 / (FP64_PEAK_TFLOPS / 2),
                           "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS},
                           "model": "achieved = MODEL flops: every factorisation (one per pivot) as a dense dgesv, 2/3 k^3 flops, 8 k^2 bytes (SURVEY 8d); the kernels "
                                    "execute no MFMA (FP64 matrix peak = FP64 vector peak on gfx950), run unfused (half of the peak) and perform only the part of a dgesv "
                                    "that Lemke's mostly-slack bases need: executed_flops"}
        # HBM bytes of the block solver's kernels in one such step, from the committed counter passes (not a measurement of this run)
        try:
            import glob
            src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_[a-z]_config4_step_pmc.json")))[-1]     # the latest session's cut
            t = json.load(open(src))
            if ("%d-box" % nboxes) in t["workload"] and ("x %d worlds" % B) in t["workload"]:
                ks = [v for k, v in t["kernels"].items() if "k_lcp_block" in k]
                res["roofline"]["traffic"] = 1e9 * sum(v.get("fetch_GB", 0.0) + v.get("write_GB", 0.0) for v in ks)
                res["roofline"]["counter_profile"] = {"commit": t["commit"], "source": "profiles/" + os.path.basename(src), "seconds_under_profiler": sum(v["seconds"] for v in ks),
                                                      "note": "FETCH_SIZE + WRITE_SIZE of mh::blk / mh::blkw::k_lcp_block over ONE cold step (the first of this leg's)"}
        except (OSError, KeyError, ValueError):
            pass
        bb.close()
        if cpu_proc is not None:
            res["cpu_baseline"] = config4_cpu_sample_collect(cpu_proc, nboxes)
        if cpu_part is not None:
            res["cpu_baseline"] = cpu_part
        return res
    except Exception as e:          # noqa: BLE001 -- informational leg
        return {"error": repr(e)}


def artic_roofline(tflops, ms):
    """The articulated kernel against its real bound, FP64 issue: ~21 kflop per world-step (CRBA + RNEA + Cholesky + the limit LCP of the
    ur10) over the launch time, against the FP64 vector peak; `issue` = the committed SQ_* counter passes of this kernel
    (profiles/r04_a_artic_issue.json)."""
    issue = None
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r04_a_artic_issue.json")))
        k = t["kernels"]["k_artic_step_w4"]
        issue = {"valu_busy_frac": k["valu_busy_frac"], "wave_cycles": k["wave_cycles"], "per_world_step": k["per_world_step"],
                 "source": "profiles/r04_a_artic_issue.json (ur10 x 8192, 200 steps)"}
    except (OSError, KeyError, ValueError):
        pass
    return {"bound": "fp64_valu", "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS, "traffic": None,
            "kernel": "mh::artic::k_artic_step_w4", "kernel_avg_us": ms * 1e3, "launches": 1, "issue": issue,
            "model": "21 kflop per world-step (estimate) / launch time against the FP64 vector = matrix peak; the kernel is bound by LDS round trips "
                     "per wave (6-36 of 64 lanes busy, half of the wave cycles on s_waitcnt), not by arithmetic or HBM"}


def _cpu_artic_worker(arg):
    first, nw, steps, B = arg
    from moby_amd import artic as A, scene as S
    from tests.test_artic_gpu import ur10_states
    from tests.oracle_api import Oracle
    oracle = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    m, _, _ = A.load_sdf(os.path.join(ROOT, "tests", "scenes", "ten_joint_arm.sdf"))
    q0, qd0 = ur10_states(m, B)
    qo, qdo, auxo = q0[first:first + nw].copy(), qd0[first:first + nw].copy(), S.new_aux(nw)
    return oracle.artic_step(m, qo, qdo, auxo, 5e-4, steps), nw * steps


def config5_cpu_sample(B=8192, steps=200, nw=256):
    """BASELINE.md 3 for config 5, before the GPU is touched: the CPU oracle (oracle/artic.hpp) on `nw` arms of the same batch x `steps` steps, one
    thread; then one process per usable core, each with its own `nw` arms."""
    import multiprocessing as mp
    try:
        secs, ws = _cpu_artic_worker((0, nw, steps, B))
        out = {"world_steps_per_sec": ws / secs, "cores": 1, "kind": "port", "sample": "%d worlds x %d steps, CPU oracle (oracle/artic.hpp), 1 thread" % (nw, steps)}
        ncores = int(os.environ.get("MH_BENCH_CPU_PROCS", usable_cpus()))
        if ncores > 1:
            t0 = time.perf_counter()
            with mp.get_context("fork").Pool(ncores) as pool:
                res = pool.map(_cpu_artic_worker, [((k * nw) % max(1, B - nw), nw, steps, B) for k in range(ncores)])
            wall = time.perf_counter() - t0
            out["all_cores"] = {"world_steps_per_sec": sum(r[1] for r in res) / wall, "cores": ncores,
                                "sample": "%d processes x %d worlds x %d steps, wall clock incl. process start" % (ncores, nw, steps)}
        return out
    except Exception as e:          # noqa: BLE001
        return {"error": repr(e)}


def config5_leg(torch, B=8192, steps=200, cpu=True, cpu_part=None):
    """BASELINE config 5: the ur10 arm (tests/scenes/ten_joint_arm.sdf = the numbers of example/ur10/model.sdf) x B random
    states, dt = 5e-4 (ur10.xml:2), `steps` steps in one launch: CRBA + RNEA + Cholesky forward dynamics and the joint-limit
    LCP every step.  With the CPU baseline on, the oracle (oracle/artic.hpp, one thread) is timed on a sample of the same batch --
    beside the GPU run, never inside it."""
    try:
        from moby_amd import artic as A
        from tests.test_artic_gpu import ur10_states
        from moby_amd import scene as S
        m, _, _ = A.load_sdf(os.path.join(ROOT, "tests", "scenes", "ten_joint_arm.sdf"))
        q0, qd0 = ur10_states(m, B)
        if cpu and cpu_part is None:
            from tests.oracle_api import Oracle
            oracle = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
            nw = min(B, 256)
            qo, qdo, auxo = q0[:nw].copy(), qd0[:nw].copy(), S.new_aux(nw)
            secs = oracle.artic_step(m, qo, qdo, auxo, 5e-4, steps)
            cpu_part = {"world_steps_per_sec": nw * steps / secs, "cores": 1, "kind": "port", "sample": "%d worlds x %d steps" % (nw, steps)}
        ab = A.ArticBatch(m, q0, qd0)
        stream = torch.cuda.current_stream().cuda_stream
        ab.step(5e-4, 10, stream); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _, _, a0 = ab.download()
        e0.record(); ab.step(5e-4, steps, stream); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        _, _, a1 = ab.download()
        rows = float(a1["lcp_rows"].astype(np.int64).sum() - a0["lcp_rows"].astype(np.int64).sum())
        ab.close()
        # the same arms with link contacts: a sphere on each finger and on the forearm, a table 15 cm below the lowest of them
        # (kernel k_artic_step_contacts: conservative advancement, mini-steps, contact + limit rows in one no-slip LCP)
        contacts = None
        try:
            ab = A.ArticBatch(m, q0[:64], qd0[:64]); P = ab.link_poses(); ab.close()
            _, links, _ = A.load_sdf(os.path.join(ROOT, "tests", "scenes", "ten_joint_arm.sdf"))
            ids = [links.index(n) for n in ("l_finger", "r_finger", "forearm_link")]
            zmin = float(min(P[:, i, 11].min() for i in ids))
            mc = type(m).from_buffer_copy(m)
            A.add_spheres(mc, [(ids[0], (0.0, 0.0, 0.0), 0.03), (ids[1], (0.0, 0.0, 0.0), 0.03), (ids[2], (0.0, 0.0, 0.0), 0.06)],
                          plane_point=(0.0, 0.0, zmin - 0.15))
            ab = A.ArticBatch(mc, q0, qd0)
            ab.step(5e-4, 10, stream); torch.cuda.synchronize()
            _, _, c0 = ab.download()
            e0.record(); ab.step(5e-4, steps, stream); e1.record(); torch.cuda.synchronize()
            msc = e0.elapsed_time(e1)
            _, _, c1 = ab.download()
            d = lambda a, b, f: float(b[f].astype(np.int64).sum() - a[f].astype(np.int64).sum())
            part = lambda a, b, t: {"ms": t, "world_steps_per_sec": d(a, b, "steps") / (t * 1e-3), "mini_steps_per_sec": d(a, b, "mini_steps") / (t * 1e-3),
                                    "lcp_rows_per_sec": d(a, b, "lcp_rows") / (t * 1e-3), "lcp_solves": d(a, b, "lcp_solves"),
                                    "worlds_that_split_a_step": int((b["mini_steps"] - a["mini_steps"] > b["steps"] - a["steps"]).sum())}
            contacts = {"workload": "the same arms, spheres on both fingers and the forearm over a table 15 cm below the lowest of them",
                        "falling": part(c0, c1, msc),
                        "note": "the first 0.1 s only: once spheres REST on the table the reference's conservative advancement crawls (step = distance / "
                                "(2 rmax |qd| + ...), CCD.cpp:545-583, with distances of 1e-7 inside the contact band) and a lockstep launch lasts as long as its slowest world"}
            ab.close()
            c3 = c1
            contacts["worlds_frozen"] = int(((c3["status"] & (4 | 16)) != 0).sum())
            contacts["worlds_with_errors"] = int(((c3["status"] & ~(2 | 4 | 16)) != 0).sum())
        except Exception as e:      # noqa: BLE001 -- informational
            contacts = {"error": repr(e)}
        # algorithmic bytes of one world-step: q, qd in and out once per LAUNCH (state stays in LDS between steps)
        return {"workload": "ur10 (10 joints) x%d, dt = 5e-4, %d steps in one launch" % (B, steps), "ms": ms, "with_link_contacts": contacts,
                "world_steps_per_sec": B * steps / (ms * 1e-3), "lcp_rows_per_sec": rows / (ms * 1e-3),
                "worlds_with_errors": int(((a1["status"] & ~2) != 0).sum()),
                "flops_per_world_step_est": 21000, "gflops_est": 21000.0 * B * steps / (ms * 1e-3) / 1e9,
                "roofline": artic_roofline(21000.0 * B * steps / (ms * 1e-3) / 1e12, ms),
                "cpu_baseline": cpu_part}
    except Exception as e:          # noqa: BLE001 -- informational leg
        return {"error": repr(e)}


def strong_leg(dist, mdist, S, make_batch, B_total, rank, world_size, dev, steps, warmup, sync, stream=None):
    """Strong scaling: ONE batch of `B_total` worlds split over the ranks (rank r owns worlds [r B/N, (r+1) B/N) of the
    same batch the N = 1 run steps), W warmup + K timed steps, barrier + synchronize on both sides, MAX over ranks.
    `make_batch(first, count)` builds the rank's share (the device batch in bench.py; tests/test_dist_gloo.py runs this very
    function over gloo with a stand-in), `sync()` drains the device.  The ranks' ranges are gathered and must tile
    [0, B_total) exactly once."""
    import torch
    first, count = mdist.split_range(rank, world_size, B_total)
    mine = torch.tensor([first, count], dtype=torch.int64, device=dev)
    allr = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world_size)]
    dist.all_gather(allr, mine)
    ranges = [(int(t[0]), int(t[1])) for t in allr]
    pos = 0
    for f, c in ranges:
        if f != pos or c < 0:
            raise SystemExit("strong scaling: rank ranges %r do not tile [0, %d)" % (ranges, B_total))
        pos += c
    if pos != B_total:
        raise SystemExit("strong scaling: rank ranges %r do not tile [0, %d)" % (ranges, B_total))
    wb = make_batch(first, count)
    if warmup > 0:
        wb.step(DT, warmup, stream)
    sync()
    _, aux0 = wb.download()
    dist.barrier(); sync()
    t0 = time.perf_counter()
    wb.step(DT, steps, stream)
    sync(); dist.barrier(); sync()
    elapsed = time.perf_counter() - t0
    _, aux1 = wb.download()
    tot = mdist.counter_vector(aux0, aux1, (aux1["status"] & ~S.MH_WORLD_IMPACT_TOL) != 0)
    elapsed, tot = mdist.reduce_interval(elapsed, tot, dist, dev)
    wb.close()
    return {"scaling": "strong", "n_gpus": world_size, "worlds_total": B_total, "worlds_per_gpu": count, "rank_ranges": ranges,
            "value": float(tot[0]) / elapsed, "unit": "LCP rows/s", "lcp_rows": float(tot[0]), "lcp_solves": float(tot[1]),
            "world_steps_per_sec": B_total * steps / elapsed, "ms_per_step": elapsed / steps * 1e3,
            "note": "%d worlds per GPU = %.1f waves per CU: below the 16 resident waves per CU the kernel needs to hide LDS latency"
                    % (count, count / 256.0)}


def config2_full_run_leg(torch, S, WorldBatchDevice, B, launches=5, total_steps=1000):
    """BASELINE config 2 as BASELINE.md states it: sphere-stack x B, the whole 1000-step run from t = 0 (regress/sphere-stack.dat is
    1000 rows), in `launches` launches of total_steps / launches steps; rows/s and world-steps/s over the WHOLE run, per-launch
    milliseconds beside them (the run slows down as worlds leave the easy regime: `long_horizon`)."""
    try:
        sc = S.sphere_stack_scene()
        wb = WorldBatchDevice(sc, S.sphere_stack_state_range(0, B))
        stream = torch.cuda.current_stream().cuda_stream
        per = total_steps // launches
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
        torch.cuda.synchronize()
        evs[0].record()
        for k in range(launches):
            wb.step(DT, per, stream); evs[k + 1].record()
        torch.cuda.synchronize()
        _, aux = wb.download()
        ms = [evs[k].elapsed_time(evs[k + 1]) for k in range(launches)]
        tot = evs[0].elapsed_time(evs[launches]) * 1e-3
        rows = float(aux["lcp_rows"].astype(np.int64).sum())
        w0 = wb.download()[0].reshape(B, sc.nb, 13)[0]
        wb.close()
        return {"workload": "sphere-stack x%d, steps 0..%d from t = 0 in %d launches (BASELINE.md 4, config 2)" % (B, per * launches, launches),
                "seconds": tot, "ms_per_launch": ms, "ms_per_step": tot / (per * launches) * 1e3,
                "lcp_rows_per_sec": rows / tot, "world_steps_per_sec": B * per * launches / tot,
                "worlds_with_errors": int(((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) != 0).sum()),
                "world0_heights": [float(w0[b, 2]) for b in range(sc.nb)]}     # regress/sphere-stack.dat:1000: 1, 3, 5
    except Exception as e:          # noqa: BLE001 -- informational leg
        return {"error": repr(e)}


def config3_cpu_sample(steps=6274, seconds=6.0):
    """BASELINE.md 3 (C4) for config 3: the CPU oracle on the recording's own world (theta-dot 0.24), one thread, whole run of `steps` steps when it
    fits `seconds`, else as many steps as do; then one process per usable core, each stepping its own wheel through the same number of steps."""
    import multiprocessing as mp
    from moby_amd import scene as S
    from tests.oracle_api import Oracle
    oracle = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    sc = S.rimless_wheel_scene()
    st = S.rimless_wheel_state((0.24,)); aux = S.new_aux(1)
    secs = oracle.world_step_batch(sc, st, aux, DT, 500)
    ns = int(max(500, min(steps, 500 * seconds / max(secs, 1e-6))))
    st = S.rimless_wheel_state((0.24,)); aux = S.new_aux(1)
    secs = oracle.world_step_batch(sc, st, aux, DT, ns)
    out = {"world_steps_per_sec": ns / secs, "lcp_rows_per_sec": float(aux["lcp_rows"].sum()) / secs, "cores": 1, "kind": "port",
           "sample": "world 0 (theta-dot 0.24) x %d steps, CPU oracle (oracle/world.hpp), 1 thread: %.2f s" % (ns, secs)}
    ncores = int(os.environ.get("MH_BENCH_CPU_PROCS", usable_cpus()))
    if ncores > 1:
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(ncores) as pool:
            res = pool.map(_cpu_wheel_worker, [(w, ns) for w in range(ncores)])
        wall = time.perf_counter() - t0
        out["all_cores"] = {"world_steps_per_sec": ncores * ns / wall, "lcp_rows_per_sec": sum(res) / wall, "cores": ncores,
                            "sample": "%d processes x 1 wheel x %d steps, wall clock incl. process start" % (ncores, ns)}
    return out


def _cpu_wheel_worker(arg):
    w, ns = arg
    from moby_amd import scene as S, synth
    from tests.oracle_api import Oracle
    oracle = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    thd = 0.24 if w == 0 else 0.2 + 0.4 * synth.world_uniforms(w, 1)[0]
    st = S.rimless_wheel_state((thd,)); aux = S.new_aux(1)
    oracle.world_step_batch(S.rimless_wheel_scene(), st, aux, DT, ns)
    return int(aux["lcp_rows"].sum())


def config3_leg(torch, S, WorldBatchDevice, cpu_part, B=2048, total_steps=6274, launches=2):
    """BASELINE config 3 as BASELINE.md 4 states it: rimless wheel x 2048 (example/rimless-wheel/wheel.xml: one rigid body with spokes, mu = 100 => the
    no-slip model, n = 1-2), world 0 at theta-dot 0.24 (regress/regression-test:58-61), the others U(0.2, 0.6); the 6274 steps of
    regress/rimless-wheel.dat from t = 0, device resident, in `launches` launches.  Steps/s over the WHOLE run; world 0's final pose beside the
    recording's last row is the parity tests' business (tests/test_world_gpu.py, tests/test_oracle_wheel.py)."""
    try:
        from moby_amd import synth
        thd = [0.24 if w == 0 else 0.2 + 0.4 * synth.world_uniforms(w, 1)[0] for w in range(B)]
        sc = S.rimless_wheel_scene()
        wb = WorldBatchDevice(sc, S.rimless_wheel_state(thd))
        stream = torch.cuda.current_stream().cuda_stream
        per = [total_steps // launches + (1 if k < total_steps % launches else 0) for k in range(launches)]
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
        torch.cuda.synchronize()
        evs[0].record()
        for k in range(launches):
            wb.step(DT, per[k], stream); evs[k + 1].record()
        torch.cuda.synchronize()
        st, aux = wb.download()
        tot = evs[0].elapsed_time(evs[launches]) * 1e-3
        wb.close()
        rows = float(aux["lcp_rows"].astype(np.int64).sum())
        return {"workload": "rimless wheel x%d (example/rimless-wheel/wheel.xml, no-slip model, n = 1-2), steps 0..%d from t = 0 in %d launches, dt = 1e-3 (BASELINE.md 4, config 3)"
                            % (B, total_steps, launches),
                "seconds": tot, "ms_per_launch": [evs[k].elapsed_time(evs[k + 1]) for k in range(launches)], "ms_per_step": tot / total_steps * 1e3,
                "world_steps_per_sec": B * total_steps / tot, "batch_steps_per_sec": total_steps / tot, "lcp_rows_per_sec": rows / tot,
                "lcp_solves": float(aux["lcp_solves"].astype(np.int64).sum()), "mini_steps": float(aux["mini_steps"].astype(np.int64).sum()),
                "worlds_with_errors": int(((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) != 0).sum()),
                "world0_pose": [float(x) for x in st.reshape(B, sc.nb, 13)[0, 0, :7]],
                "cpu_baseline": cpu_part}
    except Exception as e:          # noqa: BLE001 -- informational leg
        return {"error": repr(e)}


def long_horizon_leg(torch, wb, stream, B, args):
    """The same batch far from t = 0: advance to step `--long-horizon-start` (untimed), then time 200 steps.  After ~3000
    steps a few worlds per thousand cycle lcp_fast to its pivot cap on every rung of the regularisation ladder
    ("Long horizons"); a lockstep launch lasts as long as its slowest world.  Reported besides the plain launch:
    `idle`: from a stamped launch of the next 200 steps (mh_world_batch_profile: per-world cycle totals) the share of wave-time the
    launch leaves idle (1 - mean / slowest) and the share of its duration with under 1 % of the waves alive ((slowest - p99) / slowest);
    `split_streams`: the following 200 steps with the worlds that were slow in the timed interval (over twice the median pivots, or the top 2 %) in a
    launch of their own on a second stream (mh_world_batch_step_ids): when the OTHER worlds are done -- they could start their next
    interval then -- and when everything is."""
    try:
        from moby_amd import _lib
        done = args.warmup + args.steps
        skip = max(0, args.long_horizon_start - done)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        while skip > 0:
            k = min(skip, 1000); wb.step(DT, k, stream); torch.cuda.synchronize(); skip -= k
        _, a0 = wb.download()
        e0.record(); wb.step(DT, 200, stream); e1.record(); torch.cuda.synchronize()
        _, a1 = wb.download()
        ms = e0.elapsed_time(e1)
        rows = float(a1["lcp_rows"].astype(np.int64).sum() - a0["lcp_rows"].astype(np.int64).sum())
        pivw = a1["lcp_pivots"].astype(np.int64) - a0["lcp_pivots"].astype(np.int64)
        res = {"steps_from": max(done, args.long_horizon_start), "steps": 200, "ms_per_step": ms / 200.0,
               "lcp_rows_per_sec": rows / (ms * 1e-3), "world_steps_per_sec": B * 200 / (ms * 1e-3),
               "pivots_per_world_step": float(pivw.sum()) / (B * 200.0),
               "worlds_flagged": int((a1["status"] != 0).sum())}
        # where the launch's time goes: per-world stamped totals of the NEXT 200 steps
        lib = _lib.load()
        phc = lib.mh_world_profile_phase_count()
        ph = np.zeros(phc + 4)
        if phc is not None:
            _lib.check(lib.mh_world_batch_profile(wb.handle, DT, 200, ph.ctypes.data, phc + 4))
            mx, mn, mean, p99 = ph[phc], ph[phc + 1], ph[phc + 2], ph[phc + 3]
            res["idle"] = {"idle_wave_time_frac": 1.0 - mean / mx, "launch_time_with_under_1pct_of_waves_alive_frac": (mx - p99) / mx,
                           "slowest_over_mean_world": mx / mean, "slowest_over_fastest_world": mx / max(mn, 1.0),
                           "source": "mh_world_batch_profile, steps %d..%d (s_memtime totals per world)" % (res["steps_from"] + 201, res["steps_from"] + 400)}
        # the slow worlds of the timed interval in a launch of their own
        thr = min(2.0 * max(1.0, float(np.median(pivw))), float(np.sort(pivw)[int(0.98 * (B - 1))]))     # over twice the median, or the top 2 %
        slow = np.nonzero(pivw > thr)[0].astype(np.int32)
        fast = np.setdiff1d(np.arange(B, dtype=np.int32), slow).astype(np.int32)
        if 0 < len(slow) < B:
            dev = torch.device("cuda", torch.cuda.current_device())
            t_slow = torch.from_numpy(slow).to(dev); t_fast = torch.from_numpy(fast).to(dev)
            s2 = torch.cuda.Stream(device=dev)
            ef, es = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            _, ab = wb.download(); piv_base = ab["lcp_pivots"].astype(np.int64)
            e0.record()                                                     # on the current stream
            s2.wait_event(e0)
            wb.step_ids(DT, 200, t_slow.data_ptr(), len(slow), s2.cuda_stream)
            es.record(s2)
            wb.step_ids(DT, 200, t_fast.data_ptr(), len(fast), stream)
            ef.record()
            torch.cuda.synchronize()
            _, a3 = wb.download()
            # is slowness a property of a world?  the same threshold on the split interval's own pivot counts
            res_p = None
            try:
                pv_before = piv_base
                pv_now = a3["lcp_pivots"].astype(np.int64) - pv_before
                thr2 = min(2.0 * max(1.0, float(np.median(pv_now))), float(np.sort(pv_now)[int(0.98 * (B - 1))]))
                slow_now = set(np.nonzero(pv_now > thr2)[0].tolist())
                res_p = len(slow_now & set(slow.tolist())) / float(len(slow))
            except Exception:       # noqa: BLE001
                pass
            res["split_streams"] = {"slow_worlds": int(len(slow)), "other_worlds": int(len(fast)), "slow_again_in_this_interval_frac": res_p,
                                    "others_done_ms_per_step": e0.elapsed_time(ef) / 200.0, "all_done_ms_per_step": e0.elapsed_time(es) / 200.0,
                                    "others_world_steps_per_sec": len(fast) * 200 / (e0.elapsed_time(ef) * 1e-3),
                                    "note": "worlds are independent, so the slow ones can run on a stream of their own and the rest need not wait for them at the launch "
                                            "boundary -- IF slowness stays with a world: slow_again_in_this_interval_frac says how much of the set picked from the "
                                            "previous interval is slow again; the rest of the tail is other worlds' episodes of lcp_fast failing (its repetitions are skipped, the Lemke ladder that follows is not)"}
        return res
    except Exception as e:          # noqa: BLE001 -- informational leg
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--worlds", type=int, default=WORLDS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true")
    ap.add_argument("--no-config4", action="store_true")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--config4-boxes", type=int, default=16, help="box stack height of the config-4 full-step leg (n = 32 x boxes; 16 = the bench size, n = 512: BASELINE names 64 boxes, which the reference's own solver chain cannot solve -- DESIGN 4.2)")
    ap.add_argument("--config4-worlds", type=int, default=1024)
    ap.add_argument("--config4-steps", type=int, default=3, help="full steps of the config-4 leg: the first cold, the others warm-started from _zlast")
    ap.add_argument("--config4-tall-boxes", type=int, default=28, help="the `config4_tall_stack` leg: the largest stack of which 64 worlds take one full cold step in under 60 s on one MI355X "
                    "(measured, profiles/r05_*: 24 boxes 21.7 s, 28 boxes 36-38 s, 32 boxes 55-60 s; 64 boxes -- BASELINE's size, n = 2048 -- 135 s with 62 of 64 worlds ending in LCPSolverException); 0 = skip")
    ap.add_argument("--config4-tall-worlds", type=int, default=64)
    ap.add_argument("--no-long-horizon", action="store_true")
    ap.add_argument("--long-horizon-start", type=int, default=4000)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  The ranks are CHILD processes started before this
        # process has made any GPU call (never an exec from a process that initialised the GPU).
        import socket
        import subprocess
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    cpu = c3cpu = c4cpu_part = c5cpu = None
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        # every CPU sample runs here, BEFORE the GPU is initialised (the worker pools fork) and before any GPU leg is timed
        cpu = cpu_baseline(args.worlds, args.steps, args.warmup)
        if not args.no_config3:
            c3cpu = config3_cpu_sample()
        if not args.no_config4:
            c4cpu_part = config4_cpu_sample(args.config4_boxes, args.config4_worlds)
        if not args.no_config5:
            c5cpu = config5_cpu_sample()
    import torch
    import torch.distributed as dist
    if world_size != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world_size))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
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
    from moby_amd import scene as S
    from moby_amd import dist as mdist
    from moby_amd.world import WorldBatchDevice

    B = args.worlds
    sc = S.sphere_stack_scene()
    # shard r of an N-GPU job simulates worlds r*B .. r*B+B-1 (no data-path collective)
    first, count = mdist.shard_range(rank, B)
    st0 = S.sphere_stack_state_range(first, count)
    wb = WorldBatchDevice(sc, st0)
    stream = torch.cuda.current_stream(dev).cuda_stream   # HIP events below are recorded on this stream

    # warmup: W untimed steps (one launch)
    if args.warmup > 0:
        wb.step(DT, args.warmup, stream)
    torch.cuda.synchronize()
    _, aux0 = wb.download()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    wb.step(DT, args.steps, stream)  # EXACTLY K steps of every world, one persistent launch
    ev1.record()
    torch.cuda.synchronize()
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_s = ev0.elapsed_time(ev1) * 1e-3
    _, aux1 = wb.download()
    d = lambda f: int(aux1[f].astype(np.int64).sum() - aux0[f].astype(np.int64).sum())
    tot = mdist.counter_vector(aux0, aux1, (aux1["status"] & ~S.MH_WORLD_IMPACT_TOL) != 0)
    # the per-interval reduction of SURVEY 8e: one MAX + one SUM all-reduce (RCCL over xGMI)
    elapsed, tot = mdist.reduce_interval(elapsed, tot, dist if world_size > 1 else None, dev)
    rows, solves, pivots, minis, stabs, alg_bytes_all, stab_rows, bad = [float(x) for x in tot]

    world_steps = B * world_size * args.steps
    value = rows / elapsed
    # roofline of the dominant (only) kernel, per launch, this rank: algorithmic bytes by the
    # LCP-entry model of SURVEY 8(d): 8 (n^2 + 2n) per LCP solved, summed over the launch
    alg_bytes = float(d("lcp_alg_bytes"))
    achieved = alg_bytes / kern_s / 1e9
    fused_bytes = float(B) * (2 * sc.nb * S.MH_BODY_STATE * 8 + 2 * S.AUX_DTYPE.itemsize)
    out = {
        "metric": "lcp_rows_per_sec",
        "value": value,
        "unit": "LCP rows/s",
        "n_gpus": world_size,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "scaling_note": "`value` is WEAK scaling: every GPU steps its own %d worlds (N x %d in all) -- the figure north_star's >= 6x at 8 GPUs is read on.  For N > 1 the same line "
                        "carries `strong_scaling`: ONE batch of %d worlds split N ways (%d / 8 = 512 worlds per GPU = half a wave per SIMD of a kernel whose time is one "
                        "wave's dependent chain): it cannot approach 6x by construction and is reported for completeness" % (B, B, B, B),
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "sphere-stack x%d per GPU (example/stacks/sphere-stack.xml, dt=1e-3): full TimeSteppingSimulator::step per world" % B,
                   "worlds_per_gpu": B, "dt": DT, "parallelism": "worlds sharded x%d, no data-path collective" % world_size},
        "world_steps_per_sec": world_steps / elapsed,
        "batch_steps_per_sec": args.steps / elapsed,
        "lcp_solves": solves, "lcp_rows": rows, "lcp_rows_impact": rows - stab_rows, "lcp_rows_stabilisation": stab_rows,
        "lcp_pivots": pivots, "mini_steps": minis, "stab_iters": stabs,
        "worlds_with_errors": bad,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(B, args.steps),
                     "kernel": "mh_k_world_step", "kernel_avg_us": kern_s * 1e6, "launches": 1,
                     "algorithmic_bytes_per_launch": alg_bytes, "issue": pmc_issue(), "counter_profile": pmc_profile_tag(),
                     "model": "LCP-entry bytes 8(n^2+2n) per solved LCP (SURVEY 8d); the fused kernel itself only moves %d B of state per launch" % int(fused_bytes)},
    }

    if world_size > 1:
        out["strong_scaling"] = strong_leg(dist, mdist, S, lambda f, c: WorldBatchDevice(sc, S.sphere_stack_state_range(f, c)), B, rank, world_size, dev,
                                           args.steps, args.warmup, torch.cuda.synchronize, stream)
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if rank == 0 and world_size == 1 and not args.no_long_horizon:
        out["config2_full_run"] = config2_full_run_leg(torch, S, WorldBatchDevice, B)   # after the timed region; informational
        out["long_horizon"] = long_horizon_leg(torch, wb, stream, B, args)   # after the timed region; informational
    if rank == 0 and world_size == 1 and not args.no_config4:
        out["config4_impact_handler"] = config4_leg(torch)             # after the timed region; informational
        out["config4_full_step"] = config4_full_step_leg(torch, args.config4_boxes, args.config4_worlds, args.config4_steps, None, c4cpu_part)
    if rank == 0 and world_size == 1 and not args.no_config4 and args.config4_tall_boxes > 0:
        out["config4_tall_stack"] = config4_full_step_leg(torch, args.config4_tall_boxes, args.config4_tall_worlds, 1, None, None,
            note="the largest stack of BASELINE config 4's family of which 64 worlds take one full COLD step well inside 60 s on one MI355X (28 boxes: 36-38 s; 32 boxes, "
                 "n = 1024: 55-60 s; BASELINE's own 64 boxes, n = 2048: 135 s for 64 worlds, 62 of them ending in LCPSolverException -- tests/test_big_gpu.py holds 64 boxes x 8 worlds to the oracle's fixture)")
    if rank == 0 and world_size == 1 and not args.no_config3:
        out["config3_rimless_wheel"] = config3_leg(torch, S, WorldBatchDevice, c3cpu)
    if rank == 0 and world_size == 1 and not args.no_config5:
        out["config5_ur10"] = config5_leg(torch, cpu=not args.no_cpu_baseline, cpu_part=c5cpu)
    if rank == 0:
        print(json.dumps(out))
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
