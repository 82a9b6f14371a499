"""One-off fuzz: randomised sphere / box scenes, GPU (C ABI) against the oracle, bit for bit."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import scene as S
from moby_amd.world import WorldBatch
from tests.oracle_api import Oracle

def scene_and_state(seed):
    rng = np.random.default_rng(seed)
    with_box = bool(rng.integers(0, 2))
    nb = int(rng.integers(1, 6))
    sc = S.mh_scene(); S._defaults(sc)
    sc.nb = nb; sc.has_ground = 1
    ntot = nb + 1
    for b in range(nb):
        if with_box and b == nb - 1:
            sc.geom_type[b] = S.MH_GEOM_BOX
            e = rng.uniform(0.4, 1.0, 3); m = float(rng.uniform(0.5, 2.0))
            for k in range(3): sc.geom_dim[b][k] = e[k]
            sc.mass[b] = m; M = m / 12.0
            for k, j in enumerate((M * (e[1]**2 + e[2]**2), M * (e[0]**2 + e[2]**2), M * (e[0]**2 + e[1]**2))): sc.inertia[b][k] = j
        else:
            r = float(rng.uniform(0.3, 0.6)); m = float(rng.uniform(0.5, 2.0))
            sc.geom_type[b] = S.MH_GEOM_SPHERE; sc.geom_dim[b][0] = r; sc.mass[b] = m
            for k in range(3): sc.inertia[b][k] = r * r * m * 2.0 / 5.0
    R = S.rpy_to_R(float(rng.uniform(-0.15, 0.15)), 0.0, float(rng.uniform(-0.15, 0.15)))
    for k in range(9): sc.plane_R[k] = R.flat[k]
    for k, g in enumerate((float(rng.uniform(-0.5, 0.5)), -9.81, float(rng.uniform(-0.5, 0.5)))): sc.gravity[k] = g
    for i in range(nb):
        for j in range(i + 1, ntot):
            p = S.pair_index(i, j, ntot)
            sc.cp_epsilon[p] = float(rng.choice([0.0, 0.0, 0.4, 0.9])); sc.cp_mu_coulomb[p] = float(rng.choice([0.0, 0.3, 0.8, 150.0]))
            sc.cp_mu_viscous[p] = float(rng.choice([0.0, 0.0, 0.1])); sc.cp_compliance[p] = float(rng.choice([0.0, 0.0, 1e-5])); sc.cp_nk[p] = int(rng.choice([4, 8]))
            if j < nb and (sc.geom_type[i] == S.MH_GEOM_BOX or sc.geom_type[j] == S.MH_GEOM_BOX): sc.pair_enabled[p] = 0
    sc.cstab_max_iterations = int(rng.choice([0, 10])); sc.lcp_n_max = int(rng.choice([0, 56]))
    B = 4
    st = np.zeros((B, nb, 13)); st[:, :, 6] = 1.0
    for w in range(B):
        for b in range(nb):
            st[w, b, :3] = (1.3 * (b % 3) + rng.uniform(-0.1, 0.1), 0.7 + 1.1 * (b // 3) + rng.uniform(0, 0.3), rng.uniform(-0.1, 0.1))
            q = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 1.0]); st[w, b, 3:7] = q / np.linalg.norm(q)
            st[w, b, 7:10] = rng.uniform(-0.5, 0.5, 3); st[w, b, 10:13] = rng.uniform(-3, 3, 3)
    return sc, st.reshape(B, nb * 13), nb, with_box

if __name__ == "__main__":
    o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    lo, hi, nsteps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 300
    bad = 0
    for seed in range(lo, hi):
        sc, st0, nb, with_box = scene_and_state(seed)
        wb = WorldBatch(sc, st0.copy()); wb.step(1e-3, nsteps)
        so = st0.copy(); ao = S.new_aux(st0.shape[0])
        for w in range(st0.shape[0]): o.world_step(sc, so[w], ao[w:w + 1], 1e-3, nsteps, want_traj=False)
        ok = np.array_equal(wb.state, so) and np.array_equal(wb.aux["rng"], ao["rng"]) and np.array_equal(wb.aux["status"], ao["status"]) and np.array_equal(wb.aux["lcp_pivots"], ao["lcp_pivots"])
        print("seed %3d nb %d box %d cstab %2d nmax %2d: %s  status gpu %s oracle %s  solves %s" % (seed, nb, with_box, sc.cstab_max_iterations, sc.lcp_n_max, "OK" if ok else "MISMATCH",
              wb.aux["status"].tolist(), ao["status"].tolist(), ao["lcp_solves"].tolist()))
        bad += 0 if ok else 1
    print("mismatches:", bad)
