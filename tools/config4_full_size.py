"""BASELINE config 4 at the bench size (16 boxes per world, n = 512 impact LCPs; BASELINE names 64, the reference's chain solves up to 32)
x 1024 worlds, full TimeSteppingSimulator::step calls: the property test of tests/test_big_gpu.py::test_config4_bench_size_properties at the
batch size the configuration names.  Prints one JSON line (kept under profiles/).
python tools/config4_full_size.py [boxes] [worlds]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moby_amd import scene as S, stack as K

from moby_amd import _lib
if os.environ.get("MH_BLK_GEOM"):          # 1 = 256-thread block solver, 2 = 1024-thread (mh_debug_set key 2)
    _lib.check(_lib.load().mh_debug_set(2, int(os.environ["MH_BLK_GEOM"])))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sc = K.box_stack_scene(N)
st0 = K.box_stack_state(N, B)
st0[B // 2:] = st0[:B // 2]                                   # second half = copy of the first: batch-order independence
bb = K.BigBatch(sc, st0)
t0 = time.perf_counter()
bb.step(1e-3, 1)
st, aux = bb.download()
secs = time.perf_counter() - t0
bb.close()
b = st.reshape(B, N, 13); b0 = st0.reshape(B, N, 13)
mass = sc.mass
out = {
    "workload": "box stack of %d (impact LCP n = %d) x %d worlds, one full step, dt = 1e-3" % (N, 32 * N, B),
    "seconds": secs, "world_steps_per_sec": B / secs,
    "worlds_with_errors": int(((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) != 0).sum()),
    "worlds_impact_tolerance_warnings": int(((aux["status"] & S.MH_WORLD_IMPACT_TOL) != 0).sum()),
    "batch_order_independent": bool(np.array_equal(st[B // 2:], st[:B // 2]) and np.array_equal(aux["lcp_pivots"][B // 2:], aux["lcp_pivots"][:B // 2])),
    "max_height_error": float(np.abs(b[:, :, 1] - 0.5 - np.arange(N)).max()),
    "max_speed_after_step": float(np.abs(b[:, :, 7:13]).max()),
    # momentum: gravity adds -m g dt per body per step; what is left of it after the impact is carried by the ground
    "max_abs_vertical_momentum_after_step": float(np.abs((mass[None, :] * b[:, :, 8]).sum(axis=1)).max()),
    "lcp_rows_mean": float(aux["lcp_rows"].mean()), "lcp_pivots_mean": float(aux["lcp_pivots"].mean()), "lcp_pivots_max": int(aux["lcp_pivots"].max()),
    "lcp_solves_mean": float(aux["lcp_solves"].mean()), "stab_rows_mean": float(aux["stab_rows"].mean()), "stab_iters_mean": float(aux["stab_iters"].mean()),
}
print(json.dumps(out))
