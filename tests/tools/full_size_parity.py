"""Whole-batch parity at BASELINE's full sizes, GPU (C ABI) against the oracle, bit for bit (states / joint states, rand() streams,
flags, every counter); the oracle side runs in worker processes started before this process touches the GPU.
    python tests/tools/full_size_parity.py wheel [worlds=2048] [steps=6274]      config 3: rimless wheel x 2048, the regress run's length
    python tests/tools/full_size_parity.py ur10  [worlds=8192] [steps=2000]      config 5: ur10 x 8192, dt = 5e-4
    python tests/tools/full_size_parity.py ball  [worlds=1]    [steps=1000]      config 1: bouncing ball, dt = 0.01"""
import multiprocessing as mp, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import scene as S

FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "stab_iters", "lcp_alg_bytes", "stab_rows", "zlast_size", "zbuf_size",
          "zbuf_cap", "vns_size")
UR10 = os.path.join(ROOT, "tests", "scenes", "ten_joint_arm.sdf")


def wheel_rates(B):                      # theta_dot of world w: 0.24 for world 0 (regress), then U(0.2, 0.6) from a fixed stream
    r = np.random.default_rng(0x4D4F4259).uniform(0.2, 0.6, B); r[0] = 0.24
    return [float(x) for x in r]


def setup(mode, W):
    if mode == "wheel": return S.rimless_wheel_scene(), S.rimless_wheel_state(wheel_rates(W)), 1e-3
    if mode == "ball": return S.bouncing_ball_scene(), S.bouncing_ball_state(W), 0.01
    raise SystemExit("mode")


def oracle_part(args):
    mode, first, count, W, steps = args
    from tests.oracle_api import Oracle
    o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    if mode == "ur10":
        from moby_amd import artic as A
        from tests.test_artic_gpu import ur10_states
        m, _, _ = A.load_sdf(UR10); q, qd = ur10_states(m, W)
        q = q[first:first + count].copy(); qd = qd[first:first + count].copy(); aux = S.new_aux(count)
        o.artic_step(m, q, qd, aux, 5e-4, steps)
        return first, np.concatenate([q, qd], axis=1), aux
    sc, st, dt = setup(mode, W); st = st[first:first + count].copy(); aux = S.new_aux(count)
    for w in range(count): o.world_step(sc, st[w], aux[w:w + 1], dt, steps, want_traj=False)
    return first, st, aux


if __name__ == "__main__":
    mode = sys.argv[1]
    W = int(sys.argv[2]) if len(sys.argv) > 2 else {"wheel": 2048, "ur10": 8192, "ball": 1}[mode]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else {"wheel": 6274, "ur10": 2000, "ball": 1000}[mode]
    t0 = time.time()
    nproc = max(1, min(12, (os.cpu_count() or 2) - 2, W)); per = (W + nproc - 1) // nproc
    jobs = [(mode, f, min(per, W - f), W, steps) for f in range(0, W, per)]
    with mp.get_context("spawn").Pool(nproc) as pool: parts = sorted(pool.map(oracle_part, jobs), key=lambda p: p[0])
    so = np.concatenate([p[1] for p in parts]); ao = np.concatenate([p[2] for p in parts])
    print("[%5.1f s] oracle done" % (time.time() - t0), flush=True)
    if mode == "ur10":
        from moby_amd import artic as A
        from tests.test_artic_gpu import ur10_states
        m, _, _ = A.load_sdf(UR10); q0, qd0 = ur10_states(m, W)
        ab = A.ArticBatch(m, q0, qd0)
        for _ in range(steps // 200): ab.step(5e-4, 200)
        if steps % 200: ab.step(5e-4, steps % 200)
        q, qd, ag = ab.download(); ab.close(); sg = np.concatenate([q, qd], axis=1)
    else:
        from moby_amd.world import WorldBatch
        sc, st, dt = setup(mode, W)
        wb = WorldBatch(sc, st)
        for _ in range(steps // 200): wb.step(dt, 200)
        if steps % 200: wb.step(dt, steps % 200)
        sg, ag = wb.state, wb.aux
    bad = [f for f in FIELDS if not np.array_equal(ag[f], ao[f])]
    same = np.array_equal(sg, so, equal_nan=True)
    print("full_size_parity %s: %d worlds x %d steps: states %s, counters differing: %s; %d flagged worlds, %d LCP solves, %d pivots"
          % (mode, W, steps, "equal" if same else "DIFFER (max %.3e)" % np.nanmax(np.abs(sg - so)), bad or "none",
             int(((ao["status"] & ~S.MH_WORLD_IMPACT_TOL) != 0).sum()), int(ao["lcp_solves"].sum()), int(ao["lcp_pivots"].sum())))
    sys.exit(0 if same and not bad else 1)
