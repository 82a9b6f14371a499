"""The benchmark scene far past the benchmark window: the first W worlds of the sphere-stack x 4096 batch, `steps` steps, GPU (C ABI)
against the oracle -- states, rand() streams, status flags and every counter bit for bit.  This is the regime where a few worlds are
flagged (ImpactToleranceException, update_q giving up) and their LCPs run the whole regularisation ladder every step.
    python tests/tools/long_horizon_parity.py [worlds] [steps]"""
import multiprocessing as mp, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import scene as S

FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "stab_iters", "lcp_alg_bytes", "stab_rows", "zlast_size", "zbuf_size", "zbuf_cap")


def oracle_part(args):
    first, count, steps = args
    from tests.oracle_api import Oracle
    o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    sc = S.sphere_stack_scene(); st = S.sphere_stack_state_range(first, count); aux = S.new_aux(count)
    for w in range(count): o.world_step(sc, st[w], aux[w:w + 1], 1e-3, steps, want_traj=False)
    return first, st, aux


if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4400
    t0 = time.time()
    nproc = max(1, min(12, (os.cpu_count() or 2) - 2)); per = (W + nproc - 1) // nproc
    jobs = [(f, min(per, W - f), steps) for f in range(0, W, per)]
    with mp.get_context("spawn").Pool(nproc) as pool: parts = pool.map(oracle_part, jobs)          # before this process touches the GPU
    so = np.concatenate([p[1] for p in sorted(parts, key=lambda p: p[0])]); ao = np.concatenate([p[2] for p in sorted(parts, key=lambda p: p[0])])
    print("[%5.1f s] oracle done" % (time.time() - t0), flush=True)
    from moby_amd.world import WorldBatch
    wb = WorldBatch(S.sphere_stack_scene(), S.sphere_stack_state_range(0, W))
    for _ in range(steps // 200): wb.step(1e-3, 200)
    if steps % 200: wb.step(1e-3, steps % 200)
    bad = [f for f in FIELDS if not np.array_equal(wb.aux[f], ao[f])]
    same_state = np.array_equal(wb.state, so, equal_nan=True)
    flagged = int(((ao["status"] & ~S.MH_WORLD_IMPACT_TOL) != 0).sum())
    print("long_horizon_parity: %d worlds x %d steps: states %s, counters differing: %s; %d worlds flagged beyond IMPACT_TOL, %.1f pivots per world-step"
          % (W, steps, "equal" if same_state else "DIFFER (max %.3e)" % np.nanmax(np.abs(wb.state - so)), bad or "none", flagged, ao["lcp_pivots"].sum() / (W * steps)))
    sys.exit(0 if same_state and not bad else 1)
