"""Which solver of the chain costs what on a box-stack impact LCP (config 4)?  Takes the _MM / _qq the impact entry
assembles and times lcp_fast_regularized(-20, 4, -8) and lcp_lemke_regularized on them through the LCP entry."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moby_amd import impact as I
from moby_amd.lcp import LCP

for arg in sys.argv[1:]:
    nbx, B = [int(x) for x in arg.split(":")]
    mass, J, st, cs = I.box_stack(nbx, B=B)
    ib = I.ImpactBatch(B, nbx, 4 * nbx, 4, mass, J)
    r = ib.process(st, cs)
    MM, qq = ib.debug_lcp(); ib.close()
    n = qq.shape[1]
    lcp = LCP(B)
    z = np.zeros((B, n))
    t0 = time.perf_counter(); ok1 = lcp.lcp_fast_regularized(MM, qq, z, -20, 4, -8, z_size=np.full(B, n, dtype=np.int32)); t1 = time.perf_counter() - t0
    p1 = lcp.pivots.copy()
    z = np.zeros((B, n))
    t0 = time.perf_counter(); ok2 = lcp.lcp_lemke_regularized(MM, qq, z, z_size=np.full(B, n, dtype=np.int32)); t2 = time.perf_counter() - t0
    p2 = lcp.pivots.copy()
    print(json.dumps({"nboxes": nbx, "n": n, "worlds": B, "fast_reg_s": t1, "fast_reg_ok": int(ok1.sum()), "fast_reg_pivots_mean": float(p1.mean()),
                      "lemke_reg_s": t2, "lemke_reg_ok": int(ok2.sum()), "lemke_pivots_mean": float(p2.mean()), "lemke_pivots_max": int(p2.max()),
                      "chain_pivots_mean": float(r["pivots"].mean())}), flush=True)
