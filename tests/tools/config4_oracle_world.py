"""The CPU oracle on ONE world of a config-4 batch that tools/config4_full_size.py stepped on the GPU (--dump-world W FILE.npz): the same
start state through oracle.big_step for the same number of steps, then flags, pivot counts, rand() stream and state side by side.
lcp_lemke's bases go through the oracle's bit-equal structure-exploiting model (oracle_dbg_lemke_compact: a dense dgesv of a
2048 x 2048 basis per pivot would take an hour per rung at 64 boxes).  Prints one JSON line.
python tests/tools/config4_oracle_world.py FILE.npz"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_api import Oracle
from moby_amd import scene as S, stack as K

d = np.load(sys.argv[1])
N, steps, w, cap = int(d["boxes"]), int(d["steps"]), int(d["world"]), int(d["cap"])
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
o.lib.oracle_dbg_lemke_compact(8)
sc = K.box_stack_scene(N)
so = d["st0"].copy(); ao = S.new_aux(1); zl = np.zeros(cap); zb = np.zeros(cap)
t0 = time.perf_counter()
for k in range(steps):
    o.big_step(sc, so, ao, 1e-3, 1, zlast=zl, zbuf=zb, cap=cap)
ag = d["aux"]
print(json.dumps({"boxes": N, "world": w, "steps": steps, "oracle_seconds": time.perf_counter() - t0,
                  "oracle_status": int(ao["status"][0]), "gpu_status": int(ag["status"][0]), "status_equal": bool(int(ao["status"][0]) == int(ag["status"][0])),
                  "oracle_lcp_pivots": int(ao["lcp_pivots"][0]), "gpu_lcp_pivots": int(ag["lcp_pivots"][0]),
                  "pivots_equal": bool(int(ao["lcp_pivots"][0]) == int(ag["lcp_pivots"][0])),
                  "rng_equal": bool(np.array_equal(ao["rng"][0], ag["rng"][0])), "state_equal": bool(np.array_equal(so, d["st"])),
                  "lcp_solves": [int(ao["lcp_solves"][0]), int(ag["lcp_solves"][0])]}))
