"""The CPU oracle on worlds 1 and 0 of the 64-box batch that tools/impact_bench.py 64:8:0 runs on the GPU (one process_constraints call, cold):
status flags and pivot counts to hold against profiles/r04_a_config4_64_boxes_x8_impact_call.json.  lcp_lemke's bases go through the oracle's
bit-equal structure-exploiting model (a dense dgesv of a 2048 x 2048 basis per pivot would take an hour per rung).  ~7 minutes on one core."""
import sys, time, json, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.oracle_api import Oracle
from moby_amd import scene as S, impact as I
o = Oracle(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'oracle', 'liboracle.so')); o.lib.oracle_dbg_lemke_compact(8)
nbx, B = 64, 8
mass, J, st, cs = I.box_stack(nbx, B=B)
n = I.lcp_size(4 * nbx, 4)
for w in (1, 0):
    so = st[w].copy(); ao = S.new_aux(1)
    t = time.time()
    o.impact_process(nbx, mass, J, so, cs[w], ao, np.zeros(n), np.zeros(n), n)
    print(json.dumps({"world": w, "seconds": time.time() - t, "status": int(ao["status"][0]), "lcp_pivots": int(ao["lcp_pivots"][0]), "lcp_solves": int(ao["lcp_solves"][0])}), flush=True)
