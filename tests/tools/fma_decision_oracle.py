"""DESIGN 8 ("the oracle's dgesv above n = 64"), CPU half of the experiment: the oracle with EVERY multiply-subtract of dgesv fused (oracle_dbg_lu_fma: one
rounding per a - l*u while an LCP of more than 64 rows is solved) against the oracle as defined (unfused), on box-stack worlds of config 4: does the outcome
of a step change -- status flags, pivot counts, the state?  One JSON line per (boxes, world).
python tests/tools/fma_decision_oracle.py BOXES WORLD [WORLD ...]     (batch of 8: the perturbations are those of tests/golden/make_config4_64_boxes.py)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_api import Oracle
from moby_amd import scene as S, stack as K

N = int(sys.argv[1]); worlds = [int(a) for a in sys.argv[2:]]
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
o.lib.oracle_dbg_lemke_compact(8)
sc = K.box_stack_scene(N); cap = sc.lcp_capacity()
for w in worlds:
    res = {}
    for fma in (0, 1):
        o.lib.oracle_dbg_lu_fma(fma)
        so = K.box_stack_state(N, 8)[w].copy(); ao = S.new_aux(1)
        t0 = time.perf_counter()
        o.big_step(sc, so, ao, 1e-3, 1, zlast=np.zeros(cap), zbuf=np.zeros(cap), cap=cap)
        res[fma] = (so, ao.copy(), time.perf_counter() - t0)
    o.lib.oracle_dbg_lu_fma(0)
    a, b = res[0], res[1]
    print(json.dumps({"boxes": N, "n": 32 * N, "world": w,
                      "unfused": {"status": int(a[1]["status"][0]), "lcp_pivots": int(a[1]["lcp_pivots"][0]), "lcp_solves": int(a[1]["lcp_solves"][0]), "seconds": a[2]},
                      "fused": {"status": int(b[1]["status"][0]), "lcp_pivots": int(b[1]["lcp_pivots"][0]), "lcp_solves": int(b[1]["lcp_solves"][0]), "seconds": b[2]},
                      "state_bits_equal": bool(np.array_equal(a[0], b[0])), "max_state_diff": float(np.abs(a[0] - b[0]).max()),
                      "rand_stream_equal": bool(np.array_equal(a[1]["rng"], b[1]["rng"]))}), flush=True)
