"""Debug driver for one fuzz_artic case: steps the articulated batch and the oracle one step at a time and prints the first
differences.   python tests/tools/artic_case.py <seed> [max steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_artic as F
from moby_amd import artic as A, scene as S
from tests.oracle_api import Oracle

seed = int(sys.argv[1])
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
m, q0, qd0, nsteps = F.complete_case(o, seed)
B = q0.shape[0]
if len(sys.argv) > 2: nsteps = min(nsteps, int(sys.argv[2]))
print("nj", m.nj, "spheres", m.nspheres, "alg", m.algorithm, "mu", m.cp_mu_coulomb, "eps", m.cp_epsilon, "nk", m.cp_nk, "compl", m.cp_compliance, "visc", m.cp_mu_viscous, "steps", nsteps)
ab = A.ArticBatch(m, q0, qd0)
q_o, qd_o, aux_o = q0.copy(), qd0.copy(), S.new_aux(B)
FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "vns_size", "zlast_size", "zbuf_size", "zbuf_cap")
for s in range(nsteps):
    a_prev = aux_o.copy()
    ab.step(1e-3, 1); q_g, qd_g, aux_g = ab.download()
    o.artic_step(m, q_o, qd_o, aux_o, 1e-3, 1)
    bad = [f for f in FIELDS if not np.array_equal(aux_g[f], aux_o[f])]
    for w in range(B):                                   # the persistent solver vectors, contents
        for name, size in (("vns", int(aux_o["vns_size"][w])), ("zlast", int(aux_o["zlast_size"][w])), ("zbuf", int(aux_o["zbuf_cap"][w]))):
            if not np.array_equal(aux_g[name][w, :size], aux_o[name][w, :size], equal_nan=True):
                bad.append("%s[world %d]" % (name, w))
                idx = np.nonzero(aux_g[name][w, :size] != aux_o[name][w, :size])[0]
                print("  step", s, name, "world", w, "size", size, "differs at", idx[:8], "oracle", aux_o[name][w, idx[:4]], "gpu", aux_g[name][w, idx[:4]],
                      "zbuf_size", aux_o["zbuf_size"][w], "zlast_size", aux_o["zlast_size"][w], "zbuf_cap", aux_o["zbuf_cap"][w])
    if bad or not np.array_equal(q_g, q_o) or not np.array_equal(qd_g, qd_o):
        print("step", s, "differs:", bad, "max |dq|", np.abs(q_g - q_o).max(), "max |dqd|", np.abs(qd_g - qd_o).max())
        for w in range(B):
            if not (np.array_equal(q_g[w], q_o[w]) and np.array_equal(qd_g[w], qd_o[w]) and all(np.array_equal(aux_g[f][w], aux_o[f][w]) for f in FIELDS)):
                print("  world", w, "oracle: mini +%d solves +%d rows +%d pivots +%d status %d | gpu: mini %d solves %d rows %d pivots %d status %d" % (
                    aux_o["mini_steps"][w] - a_prev["mini_steps"][w], aux_o["lcp_solves"][w] - a_prev["lcp_solves"][w], aux_o["lcp_rows"][w] - a_prev["lcp_rows"][w],
                    aux_o["lcp_pivots"][w] - a_prev["lcp_pivots"][w], aux_o["status"][w],
                    aux_g["mini_steps"][w] - a_prev["mini_steps"][w], aux_g["lcp_solves"][w] - a_prev["lcp_solves"][w], aux_g["lcp_rows"][w] - a_prev["lcp_rows"][w],
                    aux_g["lcp_pivots"][w] - a_prev["lcp_pivots"][w], aux_g["status"][w]))
                print("    q oracle", q_o[w], "\n    q gpu   ", q_g[w], "\n    qd oracle", qd_o[w], "\n    qd gpu   ", qd_g[w])
        break
else:
    print("no difference in", nsteps, "steps")
ab.close()
