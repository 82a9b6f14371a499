"""Fuzz of the articulated stepper WITH constraint stabilisation (joint-limit rows, mh_artic_model.cstab_max_iterations > 0) against
the oracle: the random trees of fuzz_artic.py without spheres, random limits (a few infinite), random stabiliser tolerance and iteration
cap, states thrown at and past the first joint's limits (the slacks that open the stabiliser, CStab:117) and the others';
q, qd, the rand() stream, flags and counters bit for bit.     python tests/tools/fuzz_artic_stab.py [seed0] [cases]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import artic as A, scene as S     # noqa: E402
from tests.oracle_api import Oracle             # noqa: E402
from tests.tools.fuzz_artic import make_case    # noqa: E402

FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "lcp_alg_bytes", "vns_size", "stab_iters", "stab_rows")

if __name__ == "__main__":
    o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = 0; iters = 0; t0 = time.time()
    for seed in range(seed0, seed0 + cases):
        m, links, _, q0, qd0, rng = make_case(seed)
        n = m.nj
        for i in range(n):                                           # a few limits removed: rows exist for FINITE limits only
            if rng.random() < 0.15: m.hilimit[i] = np.finfo(float).max
            if rng.random() < 0.15: m.lolimit[i] = -np.finfo(float).max
        m.cstab_max_iterations = int(rng.choice([1, 3, 10, 25])); m.cstab_eps = float(rng.choice([S.NEAR_ZERO, 1e-6, 1e-4]))
        B = 6
        q0 = np.tile(q0[:1], (B, 1)); qd0 = np.tile(qd0[:1], (B, 1))
        for b in range(B):                                           # joint 0 (and sometimes others) at / past a finite limit, moving outwards
            for i in range(n):
                if i == 0 or rng.random() < 0.3:
                    side = rng.random() < 0.5
                    lim = m.hilimit[i] if side else m.lolimit[i]
                    if abs(lim) > 1e300: continue
                    q0[b, i] = lim + (1 if side else -1) * float(rng.uniform(-0.02, 0.05)); qd0[b, i] = (1 if side else -1) * float(rng.uniform(0.0, 6.0))
                else:
                    qd0[b, i] = float(rng.uniform(-2.0, 2.0))
        dt = float(rng.choice([1e-3, 5e-4, 5e-3])); nsteps = int(rng.integers(40, 160))
        ab = A.ArticBatch(m, q0, qd0); ab.step(dt, nsteps); q_g, qd_g, aux_g = ab.download(); ab.close()
        q_o, qd_o, aux_o = q0.copy(), qd0.copy(), S.new_aux(B)
        o.artic_step(m, q_o, qd_o, aux_o, dt, nsteps)
        ok = np.array_equal(q_g, q_o) and np.array_equal(qd_g, qd_o) and all(np.array_equal(aux_g[f], aux_o[f]) for f in FIELDS)
        iters += int(aux_o["stab_iters"].sum())
        if not ok:
            bad += 1
            print("MISMATCH seed %d (nj %d, maxit %d, eps %g): max|dq| %.3e; fields %s" % (seed, n, m.cstab_max_iterations, m.cstab_eps, np.abs(q_g - q_o).max(),
                  [f for f in FIELDS if not np.array_equal(aux_g[f], aux_o[f])]), flush=True)
    print("fuzz_artic_stab: %d cases from seed %d, %d mismatches, %d stabiliser iterations in all, %.0f s" % (cases, seed0, bad, iters, time.time() - t0))
    sys.exit(1 if bad else 0)
