"""Fuzz of the impact-handler entry against the oracle: random single-island contact multigraphs, random poses /
velocities / parameters, two calls each (cold + warm).  python tests/tools/fuzz_impact.py [seed0] [cases] [ds|ap]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import scene as S, impact as I
from tests.oracle_api import Oracle
from tests.test_impact_gpu import random_island, oracle_batch

o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
model = I.MH_IMPACT_MODEL_AP if (len(sys.argv) > 3 and sys.argv[3] == "ap") else I.MH_IMPACT_MODEL_DS
o.set_impact_model(model)
bad = 0
for case in range(cases):
    rng = np.random.default_rng(seed0 + case)
    nb = int(rng.integers(1, 9)); nc = int(rng.integers(max(nb - 1, 1), 4 * nb + 2)); nk = int(rng.choice([4, 4, 6, 8]))
    B = 3; n = I.lcp_size(nc, nk)
    mass = rng.uniform(0.3, 5.0, nb); J = rng.uniform(0.1, 3.0, (nb, 3))
    cs = np.stack([random_island(rng, nb, nc, static_frac=0.4) if nb > 1 else None for _ in range(B)]) if nb > 1 else None
    if cs is None:                                   # one body: every contact against something static
        cs = np.stack([random_island(rng, 2, nc) for _ in range(B)])
        cs["body1"] = 0; cs["body2"] = -1
    cs["nk"] = nk
    st = np.zeros((B, nb, 13)); st[:, :, 0:3] = rng.standard_normal((B, nb, 3))
    q = rng.standard_normal((B, nb, 4)); st[:, :, 3:7] = q / np.linalg.norm(q, axis=2)[:, :, None]
    st[:, :, 7:13] = rng.standard_normal((B, nb, 6)) * rng.choice([0.01, 1.0, 10.0])
    st = st.reshape(B, -1)
    ib = I.ImpactBatch(B, nb, nc, nk, mass, J, model=model)
    aux = S.new_aux(B); zl = np.zeros((B, n)); zb = np.zeros((B, n))
    st_o = st.copy(); st_g = st.copy(); okc = True
    for call in range(2):
        r = ib.process(st_g, cs)
        imp_o, piv_o, sol_o = oracle_batch(o, nb, mass, J, st_o, cs, n, aux, zl, zb)
        okc = okc and np.array_equal(r["state"], st_o) and np.array_equal(r["impulses"], imp_o) and np.array_equal(r["status"], aux["status"]) \
            and np.array_equal(r["pivots"], piv_o) and np.array_equal(r["solves"], sol_o)
        st_g = r["state"].copy(); st_g.reshape(B, nb, 13)[:, :, 7:13] += 0.2 * rng.standard_normal((B, nb, 6)); st_o[:] = st_g
    ib.close()
    bad += (not okc)
    print("%s seed %d nb %d nc %d nk %d n %d  solves %s status %s pivots max %d" % ("ok  " if okc else "FAIL", seed0 + case, nb, nc, nk, n, r["solves"], r["status"], r["pivots"].max()), flush=True)
print("fuzz_impact (%s model): %d cases from seed %d, mismatches: %d" % ("A-P" if model else "D-S", cases, seed0, bad))
sys.exit(1 if bad else 0)
