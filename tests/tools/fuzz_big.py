"""Fuzz of the large-world stepper (include/moby_hip_stack.h) against the oracle: random scenes of one or two box stacks
(vertex-face pairs) and loose spheres over the plane, random friction-cone edges, friction, restitution, viscous friction,
compliance, impact model (D-S / A-P), stabiliser cap and LCP capacity, perturbed worlds, a few full steps -- states, counters,
rand() streams and the handlers' _zlast / _z bit for bit.

The oracle runs first, one child process per case (started before this process touches the GPU): a scene whose LCPs keep
failing -- every regularisation level of lcp_lemke_regularized run to MAXITER, 20 000 O(n^3) pivots per solve -- costs minutes
on either side and says nothing new, so a case the oracle cannot finish in SKIP_AFTER seconds is dropped, its child ended.

    python tests/tools/fuzz_big.py [seed0] [cases]"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import scene as S, stack as K  # noqa: E402

FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "stab_iters", "lcp_alg_bytes", "stab_rows",
          "zlast_size", "zbuf_size", "zbuf_cap")
SKIP_AFTER = 8.0
B = 3


def make_case(seed):
    rng = np.random.default_rng(seed)
    gt, dims, mass, J, pos, pairs = [], [], [], [], [], []
    for s in range(int(rng.integers(1, 3))):                        # stacks of 1-3 boxes, side by side
        h = int(rng.integers(1, 4)); base = len(gt)
        for k in range(h):
            d = K.box_dims(k); m = 10.0 * d[0] * d[1] * d[2]
            gt.append(S.MH_GEOM_BOX); dims.append(d); mass.append(m)
            J.append((m / 12 * (d[1] ** 2 + d[2] ** 2), m / 12 * (d[0] ** 2 + d[2] ** 2), m / 12 * (d[0] ** 2 + d[1] ** 2)))
            pos.append((3.0 * s, 0.5 + k + rng.uniform(0.0, 2e-4), 0.0))
            if k > 0: pairs.append((base + k - 1, base + k, K.MH_PAIR_VERTEX_FACE))
    for s in range(int(rng.integers(0, 3))):                        # loose spheres
        r = float(rng.uniform(0.2, 0.5)); m = float(rng.uniform(0.5, 2.0))
        gt.append(S.MH_GEOM_SPHERE); dims.append((r, 0, 0)); mass.append(m); J.append((0.4 * m * r * r,) * 3)
        pos.append((-3.0 - 1.5 * s, r + rng.uniform(0.0, 0.3), 1.0))
    nb = len(gt)
    pairs += [(b, nb, 0) for b in range(nb)]
    sph = [b for b in range(nb) if gt[b] == S.MH_GEOM_SPHERE]
    pairs += [(a, b, 0) for i, a in enumerate(sph) for b in sph[i + 1:]]
    par = dict(nk=int(rng.choice([4, 4, 6, 8])), epsilon=float(rng.choice([0.0, 0.0, 0.4])),
               mu_coulomb=float(rng.choice([0.0, 1e-4, 0.3, 0.8, 150.0])), mu_viscous=float(rng.choice([0.0, 0.0, 0.05])),
               compliance=float(rng.choice([0.0, 0.0, 1e-6])), cstab_max_iterations=int(rng.choice([0, 5, 10])),
               lcp_n_max=int(rng.choice([0, 0, 96])), impact_model=int(rng.random() < 0.25))
    sc = K.BigScene(gt, dims, mass, J, pairs, gravity=(float(rng.uniform(-0.3, 0.3)), -9.81, 0.0), **par)
    st = np.zeros((B, nb, 13)); st[:, :, 6] = 1.0; st[:, :, 0:3] = np.array(pos)
    st[:, :, 8] = -0.05 * rng.random((B, nb))
    st[1:, :, 7] += rng.uniform(-1e-2, 1e-2, (B - 1, nb)); st[1:, :, 9] += rng.uniform(-1e-2, 1e-2, (B - 1, nb))
    st[1:, :, 10:13] += rng.uniform(-1e-2, 1e-2, (B - 1, nb, 3))
    return sc, st.reshape(B, -1), int(rng.integers(2, 9)), par, nb, len(pairs)


def oracle_case(seed, conn):
    from tests.oracle_api import Oracle
    o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    sc, st, nsteps, _, _, _ = make_case(seed)
    cap = sc.lcp_capacity()
    aux = S.new_aux(B); zl = np.zeros((B, cap)); zb = np.zeros((B, cap))
    for w in range(B):
        o.big_step(sc, st[w], aux[w:w + 1], 1e-3, nsteps, zlast=zl[w], zbuf=zb[w], cap=cap)
    conn.send((st, aux, zl, zb)); conn.close()


def oracle_results(seeds, workers):
    """{seed: (state, aux, zlast, zbuf) or None when the case ran past SKIP_AFTER}; ``workers`` children at a time."""
    ctx = mp.get_context("spawn"); out = {}; todo = list(seeds); live = []
    while todo or live:
        while todo and len(live) < workers:
            s = todo.pop(0); a, b = ctx.Pipe(duplex=False); p = ctx.Process(target=oracle_case, args=(s, b)); p.start(); b.close()
            live.append((s, p, a, time.time()))
        time.sleep(0.05)
        for item in list(live):
            s, p, a, t0 = item
            if a.poll():
                out[s] = a.recv(); p.join(); live.remove(item)
            elif not p.is_alive():
                out[s] = None; live.remove(item); print("    oracle child of seed %d died (exit %r)" % (s, p.exitcode), flush=True)
            elif time.time() - t0 > SKIP_AFTER + 3.0:          # + interpreter start
                p.terminate(); p.join(); out[s] = None; live.remove(item)
    return out


if __name__ == "__main__":
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    T0 = time.time()
    ora = oracle_results(range(seed0, seed0 + cases), workers=max(1, min(8, (os.cpu_count() or 2) - 1)))
    print("[%6.1f s] oracle done: %d of %d cases inside %g s" % (time.time() - T0, sum(v is not None for v in ora.values()), cases, SKIP_AFTER), flush=True)
    bad = solves = stab = flagged = skipped = 0
    for seed in range(seed0, seed0 + cases):
        sc, st, nsteps, par, nb, npairs = make_case(seed)
        print("[%6.1f s] seed %d: nb %d pairs %d steps %d %r" % (time.time() - T0, seed, nb, npairs, nsteps, par), flush=True)
        if ora[seed] is None:
            skipped += 1; print("    skipped: the oracle needed more than %g s" % SKIP_AFTER, flush=True); continue
        st_o, aux_o, zl, zb = ora[seed]
        bb = K.BigBatch(sc, st)
        assert bb.cap == sc.lcp_capacity(), (bb.cap, sc.lcp_capacity())
        bb.step(1e-3, nsteps)
        st_g, aux_g = bb.download(); ss = bb.solver_state(); bb.close()
        same = all(np.array_equal(aux_g[f], aux_o[f]) for f in FIELDS) and np.array_equal(st_g, st_o, equal_nan=True)
        for w in range(B):
            n = int(aux_o["zlast_size"][w]); c = int(aux_o["zbuf_cap"][w])
            same = same and np.array_equal(ss["zlast"][w, :n], zl[w, :n]) and np.array_equal(ss["zbuf"][w, :c], zb[w, :c])
        solves += int(aux_g["lcp_solves"].sum()); stab += int(aux_g["stab_iters"].sum())
        flagged += int(((aux_g["status"] & ~S.MH_WORLD_IMPACT_TOL) != 0).sum())
        if not same:
            bad += 1
            print("MISMATCH seed %d: max |dstate| %.3e; status gpu %r oracle %r; %s" % (
                seed, np.nanmax(np.abs(st_g - st_o)), aux_g["status"], aux_o["status"],
                [f for f in FIELDS if not np.array_equal(aux_g[f], aux_o[f])]), flush=True)
    print("fuzz_big: %d cases from seed %d (%d skipped as too slow for the oracle), %d mismatches; %d LCP solves, %d stabilisation iterations, "
          "%d flagged world-runs" % (cases, seed0, skipped, bad, solves, stab, flagged))
    sys.exit(1 if bad else 0)
