"""Debug driver for one fuzz_big case: steps the large-world batch and the oracle one step at a time and prints the counters
that differ.   python tests/tools/big_case.py <seed>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_big as F
from moby_amd import scene as S, stack as K
from tests.oracle_api import Oracle

seed = int(sys.argv[1])
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
sc, st, nsteps, par, nb, npairs = F.make_case(seed)
print(par, "nb", nb, "pairs", npairs, "steps", nsteps)
B = st.shape[0]
bb = K.BigBatch(sc, st); cap = bb.cap
st_o = st.copy(); aux_o = S.new_aux(B); zl = np.zeros((B, cap)); zb = np.zeros((B, cap))
for s in range(nsteps):
    bb.step(1e-3, 1); st_g, aux_g = bb.download()
    for w in range(B):
        o.big_step(sc, st_o[w], aux_o[w:w + 1], 1e-3, 1, zlast=zl[w], zbuf=zb[w], cap=cap)
    for f in ("lcp_solves", "lcp_rows", "lcp_pivots", "mini_steps", "stab_iters", "stab_rows", "status", "zlast_size"):
        if not np.array_equal(aux_g[f], aux_o[f]): print("step", s, f, "gpu", aux_g[f], "oracle", aux_o[f])
    print("step", s, "solves", aux_o["lcp_solves"], "rows", aux_o["lcp_rows"], "pivots gpu", aux_g["lcp_pivots"], "oracle", aux_o["lcp_pivots"], "state equal", np.array_equal(st_g, st_o))
bb.close()
