"""One fuzz_throw seed step by step: the first step at which the one-wavefront kernel and the oracle differ, and in what.
    python tests/tools/throw_case.py SEED [steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import scene as S
from moby_amd.world import WorldBatch
from tests.oracle_api import Oracle
import importlib.util
_spec = importlib.util.spec_from_file_location("_twg", os.path.join(ROOT, "tests", "test_world_gpu.py")); _twg = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_twg)
FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "stab_iters", "zlast_size")
seed = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
sc, st0 = _twg._harsh_scene(seed)
print("seed %d: nb %d, masses %s" % (seed, sc.nb, [sc.mass[b] for b in range(sc.nb)]))
wb = WorldBatch(sc, st0.copy()); so = st0.copy(); ao = S.new_aux(st0.shape[0])
done = set()
for k in range(steps):
    wb.step(1e-3, 1)
    for w in range(st0.shape[0]):
        o.world_step(sc, so[w], ao[w:w + 1], 1e-3, 1, want_traj=False)
    for w in range(st0.shape[0]):
        if w in done: continue
        diff = [f for f in FIELDS if not np.array_equal(wb.aux[f][w], ao[f][w])]
        sd = not np.array_equal(wb.state[w], so[w], equal_nan=True)
        if diff or sd:
            done.add(w)
            print("step %d world %d differs: fields %s; state max |diff| %.3e; gpu status %d pivots %d solves %d mini %d | oracle status %d pivots %d solves %d mini %d"
                  % (k, w, diff, np.nanmax(np.abs(wb.state[w] - so[w])) if sd else 0.0, wb.aux["status"][w], wb.aux["lcp_pivots"][w], wb.aux["lcp_solves"][w], wb.aux["mini_steps"][w],
                     ao["status"][w], ao["lcp_pivots"][w], ao["lcp_solves"][w], ao["mini_steps"][w]))
print("worlds that differed:", sorted(done))
