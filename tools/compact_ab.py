"""A/B of the block solver's two LU routes on box-stack impact LCPs (config 4): the structure-exploiting LU of Lemke's bases
(mh_lu_compact.inc, mh_debug_set(3, 1)) against the dense dgesv on the assembled basis (mh_debug_set(3, 0)).
Same worlds through both; states, pivots and flags must agree bit for bit.   python tools/compact_ab.py "nboxes:B" ..."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moby_amd import _lib, impact as I

lib = _lib.load()
for a in sys.argv[1:]:
    nbx, B = [int(x) for x in a.split(":")]
    mass, J, st, cs = I.box_stack(nbx, B=B)
    res = {}
    for mode in (1, 0):
        _lib.check(lib.mh_debug_set(3, mode))
        ib = I.ImpactBatch(B, nbx, 4 * nbx, 4, mass, J)
        ib.upload(st, cs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ib.process_async(None); r = ib.download()
        res[mode] = (time.perf_counter() - t0, r); ib.close()
    _lib.check(lib.mh_debug_set(3, 1))
    rc, rd = res[1][1], res[0][1]
    print(json.dumps({"nboxes": nbx, "n": 32 * nbx, "worlds": B, "compact_s": res[1][0], "dense_s": res[0][0],
                      "states_equal": bool(np.array_equal(rc["state"], rd["state"])), "pivots_equal": bool(np.array_equal(rc["pivots"], rd["pivots"])),
                      "worlds_with_other_pivots": int((rc["pivots"] != rd["pivots"]).sum()),
                      "status_equal": bool(np.array_equal(rc["status"], rd["status"])), "pivots_mean": float(rc["pivots"].mean()),
                      "max_state_diff": float(np.abs(rc["state"] - rd["state"]).max())}), flush=True)
