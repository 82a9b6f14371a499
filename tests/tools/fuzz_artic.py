"""Fuzz of the articulated stepper with link contacts (mh_artic_model.nspheres > 0) against the oracle: random trees of 1-8
revolute / prismatic joints, random link frames, masses and inertias, 1-4 spheres on random links, a random plane below the body,
random limits (some active), restitution at limits and contacts, both forward-dynamics algorithms, random states; q, qd, the rand()
stream, the warm start and the counters bit for bit.  With `floating` every body rides on a FLOATING base (mh_artic_model.floating_base: six virtual joints under a
base link of random pose and inertia, 0-6 joints of its own, spheres on any link including the base link).     python tests/tools/fuzz_artic.py [seed0] [cases] [floating]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import artic as A, scene as S  # noqa: E402
from tests.oracle_api import Oracle          # noqa: E402

FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "lcp_alg_bytes", "vns_size", "zlast_size", "zbuf_size", "zbuf_cap")


def rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q); w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


FLOATING = False


def make_case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(0, 7)) if FLOATING else int(rng.integers(1, 9))
    links = []
    for i in range(n):
        p = -1 if i == 0 else int(rng.integers(max(0, i - 2), i))           # chains with a few branches
        xp = np.zeros(3) if p < 0 else np.asarray(links[p]["x0"])
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        I = rot(rng); d = rng.uniform(0.01, 0.1, 3); J = I @ np.diag(d) @ I.T; J = 0.5 * (J + J.T)
        pris = rng.random() < 0.2
        lo, hi = (-0.3, 0.3) if pris else (-1.2, 1.2)
        links.append(dict(parent=p, type=A.MH_JOINT_PRISMATIC if pris else A.MH_JOINT_REVOLUTE, R0=rot(rng), x0=xp + rng.uniform(-0.4, 0.4, 3) + np.array([0, 0, -0.3]),
                          axis=ax, com=rng.uniform(-0.15, 0.15, 3), inertia=J, mass=float(rng.uniform(0.5, 3.0)),
                          lo=lo * rng.uniform(0.3, 1.0), hi=hi * rng.uniform(0.3, 1.0), restitution=float(rng.choice([0.0, 0.0, 0.5]))))
    fb = None
    if FLOATING:
        I = rot(rng); J = I @ np.diag(rng.uniform(0.02, 0.2, 3)) @ I.T
        fb = dict(R0=rot(rng), x0=rng.uniform(-0.3, 0.3, 3), mass=float(rng.uniform(1.0, 5.0)), inertia=0.5 * (J + J.T))
    m = A.model_from_links(links, gravity=(float(rng.uniform(-1, 1)), 0.0, -9.81), floating_base=fb)
    m.algorithm = int(rng.random() < 0.4)
    B = 4
    q0 = np.column_stack([rng.uniform(0.8 * L["lo"], 0.8 * L["hi"], B) for L in links]) if n else np.zeros((B, 0))
    qd0 = rng.uniform(-2.0, 2.0, (B, n))
    if FLOATING:        # the base: a small offset and a tilt well inside the middle hinge's +-pi/2, linear and angular rates
        q0 = np.column_stack([rng.uniform(-0.2, 0.2, (B, 3)), rng.uniform(-0.5, 0.5, (B, 3)), q0]); qd0 = np.column_stack([rng.uniform(-1.0, 1.0, (B, 3)), rng.uniform(-2.0, 2.0, (B, 3)), qd0])
        n += 6
    ns = int(rng.integers(1, A.MH_ARTIC_MAX_SPHERES + 1))
    sph = [(int(rng.integers(5 if FLOATING else 0, n)), rng.uniform(-0.2, 0.2, 3), float(rng.uniform(0.03, 0.15))) for _ in range(ns)]
    return m, links, sph, q0, qd0, rng


def complete_case(o, seed):
    """make_case + the plane (just below the lowest sphere of the start states, tilted), the contact parameters and the run length:
    -> (model, q0, qd0, nsteps).  `o`: the oracle (link poses of the start states)."""
    m, links, sph, q0, qd0, rng = make_case(seed)
    B = q0.shape[0]
    P = np.array([o.artic_fwd_dyn(m, q0[b], qd0[b])["poses"] for b in range(B)])
    nrm = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), 1.0]); nrm /= np.linalg.norm(nrm)
    low = min(float(nrm @ (P[b, l, 9:12] + P[b, l, :9].reshape(3, 3) @ c)) - r for b in range(B) for (l, c, r) in sph)
    A.add_spheres(m, sph, plane_normal=nrm, plane_point=nrm * (low - float(rng.uniform(0.0, 0.05))), epsilon=float(rng.choice([0.0, 0.0, 0.4])),
                  mu_coulomb=float(rng.choice([100.0, 100.0, 1e4, 0.5, 0.0, 2.0])), mu_viscous=float(rng.choice([0.0, 0.0, 0.1])),
                  compliance=float(rng.choice([0.0, 1e-6])), nk=int(rng.choice([4, 4, 6, 8])))
    return m, q0, qd0, int(rng.integers(100, 400))


if __name__ == "__main__":
    o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 900
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    FLOATING = len(sys.argv) > 3 and sys.argv[3] == "floating"
    SKIP_AFTER = 5.0
    bad = solves = minis = multi = flagged = skipped = 0
    for case in range(cases):
        m, q0, qd0, nsteps = complete_case(o, seed0 + case)
        B = q0.shape[0]
        # the oracle first, in chunks: a world that keeps hitting the mini-step cap costs minutes on either side -- skip such a case
        q_o, qd_o, aux_o = q0.copy(), qd0.copy(), S.new_aux(B)
        t0 = time.time(); done = 0
        while done < nsteps and time.time() - t0 < SKIP_AFTER:
            k = min(10, nsteps - done); o.artic_step(m, q_o, qd_o, aux_o, 1e-3, k); done += k
        if done < nsteps:
            skipped += 1; print("seed %d skipped: the oracle needed more than %g s" % (seed0 + case, SKIP_AFTER), flush=True); continue
        ab = A.ArticBatch(m, q0, qd0)
        for _ in range(nsteps // 10): ab.step(1e-3, 10)
        if nsteps % 10: ab.step(1e-3, nsteps % 10)
        q_g, qd_g, aux_g = ab.download(); ab.close()
        same = np.array_equal(q_g, q_o, equal_nan=True) and np.array_equal(qd_g, qd_o, equal_nan=True) and all(np.array_equal(aux_g[f], aux_o[f]) for f in FIELDS)
        for w in range(B):
            k = int(aux_o["vns_size"][w]); same = same and np.array_equal(aux_g["vns"][w, :k], aux_o["vns"][w, :k])
            k = int(aux_o["zlast_size"][w]); c = int(aux_o["zbuf_cap"][w])
            same = same and np.array_equal(aux_g["zlast"][w, :k], aux_o["zlast"][w, :k]) and np.array_equal(aux_g["zbuf"][w, :c], aux_o["zbuf"][w, :c])
        solves += int(aux_o["lcp_solves"].sum()); minis += int((aux_o["mini_steps"] - aux_o["steps"]).sum())
        multi += int((aux_o["lcp_rows"] > aux_o["lcp_solves"]).sum()); flagged += int((aux_o["status"] & ~S.MH_WORLD_IMPACT_TOL != 0).sum())
        if not same:
            bad += 1
            print("MISMATCH seed %d: nj %d spheres %d alg %d steps %d; max |dq| %.3e; status gpu %r oracle %r; %s" % (
                seed0 + case, m.nj, m.nspheres, m.algorithm, nsteps, np.nanmax(np.abs(q_g - q_o)), aux_g["status"], aux_o["status"],
                [f for f in FIELDS if not np.array_equal(aux_g[f], aux_o[f])]), flush=True)
        else:                                  # (every case: a run that stops writing for minutes is taken to be hung on the GPU box)
            print("case %d ok (nj %d, %d spheres, %d LCP solves so far)" % (case, m.nj, m.nspheres, solves), flush=True)
    print("fuzz_artic%s: %d cases from seed %d (%d skipped as too slow for the oracle), %d mismatches; %d LCP solves (%d world-runs with multi-row LCPs), %d extra mini-steps, %d flagged world-runs"
          % (" (floating bases)" if FLOATING else "", cases, seed0, skipped, bad, solves, multi, minis, flagged))
    sys.exit(1 if bad else 0)
