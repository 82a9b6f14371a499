"""A/B of everything this build schedules differently from the reference's loops on full steps of box stacks: the
same worlds with all switches at their defaults and with all of them off (dense LU for Lemke's bases, the ladder in sequence, lcp_fast through the HBM workspace, every
lcp_fast iteration run, every basis factorised from scratch, tasks by block index).  States, rand() streams, counters and flags must
agree bit for bit.     python tools/feature_ab.py "nboxes:B" ..."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moby_amd import _lib, stack as K

FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "stab_iters", "lcp_alg_bytes", "stab_rows",
          "zlast_size", "zbuf_size", "zbuf_cap")
DEFAULTS = {3: 1, 4: 3, 5: 1, 6: 1, 7: 1, 10: 1}
lib = _lib.load()
for a in sys.argv[1:]:
    nbx, B = [int(x) for x in a.split(":")]
    sc = K.box_stack_scene(nbx); st0 = K.box_stack_state(nbx, B)
    res = {}
    for name, keys in (("defaults", DEFAULTS), ("all off", {k: 0 for k in DEFAULTS})):
        for k, v in keys.items():
            _lib.check(lib.mh_debug_set(k, v))
        bb = K.BigBatch(sc, st0)
        t0 = time.perf_counter(); bb.step(1e-3, 1); st, aux = bb.download(); secs = time.perf_counter() - t0
        bb.close()
        res[name] = (secs, st, aux)
        print("%s: %.1f s" % (name, secs), flush=True)
    for k, v in DEFAULTS.items():
        _lib.check(lib.mh_debug_set(k, v))
    a_, b_ = res["defaults"], res["all off"]
    differing = [f for f in FIELDS if not np.array_equal(a_[2][f], b_[2][f])]
    print(json.dumps({"nboxes": nbx, "n": 32 * nbx, "worlds": B, "defaults_s": a_[0], "all_off_s": b_[0], "states_equal": bool(np.array_equal(a_[1], b_[1])),
                      "counters_differing": differing, "pivots_mean": float(a_[2]["lcp_pivots"].mean()), "pivots_max": int(a_[2]["lcp_pivots"].max()),
                      "worlds_with_errors": int(((a_[2]["status"] & ~2) != 0).sum())}), flush=True)
