"""Config 4 at the bench size (16-box stacks x 1024 worlds), `steps` full steps, twice: with the Lemke ladder's tasks launched beside lcp_fast's
kernel (mh_debug_set(4, 3), the default: gate, tasks by verdict, second launch) and after it (4, 2).  States, rand() streams, counters, flags
and warm-start sizes must agree bit for bit; prints one JSON line with the seconds per step of both and a digest of the final state.
  python tools/launch_order_ab.py [boxes] [worlds] [steps]"""
import hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moby_amd import _lib, stack as K

FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "stab_iters", "lcp_alg_bytes", "stab_rows",
          "zlast_size", "zbuf_size", "zbuf_cap")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib = _lib.load()
sc = K.box_stack_scene(N); st0 = K.box_stack_state(N, B)
res = {}
for key in (3, 2):
    _lib.check(lib.mh_debug_set(4, key))
    bb = K.BigBatch(sc, st0); secs = []
    for _ in range(steps):
        t0 = time.perf_counter(); bb.step(1e-3, 1); st, aux = bb.download(); secs.append(time.perf_counter() - t0)
    ss = bb.solver_state(); bb.close()
    res[key] = (secs, st, aux, ss)
    print("key 4 = %d: %s s" % (key, ["%.2f" % s for s in secs]), flush=True)
_lib.check(lib.mh_debug_set(4, 3))
a, b = res[3], res[2]
differing = [f for f in FIELDS if not np.array_equal(a[2][f], b[2][f])]
same = np.array_equal(a[1], b[1]) and not differing and all(np.array_equal(a[3][k], b[3][k]) for k in ("zlast", "zbuf", "sizes"))
print(json.dumps({"workload": "box stack of %d x %d worlds, %d full steps" % (N, B, steps), "seconds_per_step_tasks_beside_lcp_fast": a[0],
                  "seconds_per_step_tasks_after_lcp_fast": b[0], "bit_equal": bool(same), "differing_fields": differing,
                  "state_sha256": hashlib.sha256(np.ascontiguousarray(a[1]).tobytes()).hexdigest(), "lcp_pivots_mean": float(a[2]["lcp_pivots"].mean())}))
sys.exit(0 if same else 1)
