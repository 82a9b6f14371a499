"""BASELINE config 4 through the large-world stepper at a size given on the command line: B stacks of N boxes (impact LCP n = 32 N),
`steps` full TimeSteppingSimulator::step calls (the first cold, the others warm-started from _zlast).  16 boxes (n = 512) is the bench size;
BASELINE names 64, which the reference's own solver chain does not solve; 32 is the largest it does.  Prints one JSON line
(kept under profiles/): per-step seconds, properties of the final state; with --dump-world W FILE.npz the start and end state and the
solver record of world W are saved for tests/tools/config4_oracle_world.py, which runs the CPU oracle on the same world and compares.
--states-of B0 takes the start states of the first `worlds` worlds of a batch of B0 (the perturbations are drawn per batch) without mirroring the second
half; --dump-failed PREFIX saves every world that ends with MH_WORLD_LCP_FAILED as PREFIX_w<index>.npz in the same format.
python tools/config4_full_size.py [boxes] [worlds] [steps] [--dump-world W FILE.npz] [--states-of B0] [--dump-failed PREFIX]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moby_amd import scene as S, stack as K

from moby_amd import _lib
if os.environ.get("MH_BLK_GEOM"):          # mh_debug_set key 2 (the block solver's thread geometry)
    _lib.check(_lib.load().mh_debug_set(2, int(os.environ["MH_BLK_GEOM"])))
if os.environ.get("MH_COMPACT_LU"):        # mh_debug_set key 3 (0: the dense LU everywhere)
    _lib.check(_lib.load().mh_debug_set(3, int(os.environ["MH_COMPACT_LU"])))
if os.environ.get("MH_FAST_GEOM"):         # mh_debug_set key 8 (the lcp_fast kinds' thread geometry for n <= 512)
    _lib.check(_lib.load().mh_debug_set(8, int(os.environ["MH_FAST_GEOM"])))
if os.environ.get("MH_TASKS"):             # mh_debug_set key 4 (3: the ladder's tasks behind lcp_fast on a second stream)
    _lib.check(_lib.load().mh_debug_set(4, int(os.environ["MH_TASKS"])))
if os.environ.get("MH_REG_LU"):            # mh_debug_set key 10 (0: lcp_fast's nonbasic systems through the HBM workspace)
    _lib.check(_lib.load().mh_debug_set(10, int(os.environ["MH_REG_LU"])))
if os.environ.get("MH_LPT"):               # mh_debug_set key 11 (0: full-chip launches take the worlds by index, not by the solver time they have used)
    _lib.check(_lib.load().mh_debug_set(11, int(os.environ["MH_LPT"])))
_opt = {"--dump-world": 2, "--states-of": 1, "--dump-failed": 1}
args, _k = [], 1
while _k < len(sys.argv):
    if sys.argv[_k] in _opt: _k += 1 + _opt[sys.argv[_k]]
    else: args.append(sys.argv[_k]); _k += 1
N = int(args[0]) if len(args) > 0 else 16
B = int(args[1]) if len(args) > 1 else 1024
steps = int(args[2]) if len(args) > 2 else 1
ow = int(sys.argv[sys.argv.index("--dump-world") + 1]) if "--dump-world" in sys.argv else None
dump = sys.argv[sys.argv.index("--dump-world") + 2] if ow is not None else None
sc = K.box_stack_scene(N)
B0 = int(sys.argv[sys.argv.index("--states-of") + 1]) if "--states-of" in sys.argv else None
dump_failed = sys.argv[sys.argv.index("--dump-failed") + 1] if "--dump-failed" in sys.argv else None
st0 = K.box_stack_state(N, B) if B0 is None else np.ascontiguousarray(K.box_stack_state(N, B0)[:B])
if B >= 2 and B0 is None:
    st0[B // 2:] = st0[:B // 2]                               # second half = copy of the first: batch-order independence
bb = K.BigBatch(sc, st0)
# a step of a large batch can take many minutes without a line of output: keep a heartbeat file growing (gpurun takes silence for a hang)
import threading
_stop = threading.Event()
def _beat():
    hb = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "config4_heartbeat.txt")
    try:
        os.makedirs(os.path.dirname(hb), exist_ok=True)
        while not _stop.wait(60.0):
            with open(hb, "a") as f:
                f.write("%s boxes %d worlds %d alive\n" % (time.strftime("%H:%M:%S"), N, B))
    except OSError:
        pass
threading.Thread(target=_beat, daemon=True).start()
secs = []
for k in range(steps):
    t0 = time.perf_counter()
    bb.step(1e-3, 1)
    st, aux = bb.download()
    secs.append(time.perf_counter() - t0)
    print("step %d: %.2f s, worlds with errors %d" % (k, secs[-1], int(((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) != 0).sum())), file=sys.stderr, flush=True)
wk = bb.lu_work()
cap = bb.cap
bb.close()
_stop.set()
b = st.reshape(B, N, 13)
mass = sc.mass
h = B // 2 if B >= 2 else B
out = {
    "workload": "box stack of %d (impact LCP n = %d) x %d worlds, %d full step%s, dt = 1e-3" % (N, 32 * N, B, steps, "" if steps == 1 else "s (cold, then warm-started)"),
    "seconds": float(sum(secs)), "seconds_per_step": secs, "world_steps_per_sec": B * steps / float(sum(secs)),
    "worlds_with_errors": int(((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) != 0).sum()),
    "worlds_lcp_failed": int(((aux["status"] & S.MH_WORLD_LCP_FAILED) != 0).sum()),
    "status_flags_per_world": [int(x) for x in aux["status"][:min(B, 16)]],
    "worlds_impact_tolerance_warnings": int(((aux["status"] & S.MH_WORLD_IMPACT_TOL) != 0).sum()),
    "batch_order_independent": None if B0 is not None else bool(B < 2 or (np.array_equal(st[h:2 * h], st[:h]) and np.array_equal(aux["lcp_pivots"][h:2 * h], aux["lcp_pivots"][:h]))),
    "max_height_error": float(np.abs(b[:, :, 1] - 0.5 - np.arange(N)).max()),
    "max_speed_after_last_step": float(np.abs(b[:, :, 7:13]).max()),
    "lcp_rows_mean": float(aux["lcp_rows"].mean()), "lcp_pivots_mean": float(aux["lcp_pivots"].mean()), "lcp_pivots_max": int(aux["lcp_pivots"].max()),
    "lcp_solves_mean": float(aux["lcp_solves"].mean()), "stab_rows_mean": float(aux["stab_rows"].mean()), "stab_iters_mean": float(aux["stab_iters"].mean()),
    "solver_workgroup_seconds": float(wk[:, 3].sum()), "slowest_world_solver_seconds": float(wk[:, 3].max()), "mean_world_solver_seconds": float(wk[:, 3].mean()),
    "model_flops": float(wk[:, 0].sum()), "issued_flops": float(wk[:, 2].sum()),
}
if ow is not None:          # the GPU side of world `ow` for tests/tools/config4_oracle_world.py (the checker that runs the CPU oracle on the same world)
    np.savez(dump, boxes=N, steps=steps, world=ow, cap=cap, st0=st0[ow], st=st[ow], aux=aux[ow:ow + 1])
    out["dumped_world"] = {"world": ow, "file": dump}
if dump_failed is not None:
    bad = [int(w) for w in np.nonzero((aux["status"] & S.MH_WORLD_LCP_FAILED) != 0)[0]]
    for w in bad:
        np.savez("%s_w%d.npz" % (dump_failed, w), boxes=N, steps=steps, world=w, cap=cap, st0=st0[w], st=st[w], aux=aux[w:w + 1])
    out["failed_worlds"] = bad
if B0 is not None:
    out["states_of_batch"] = B0
print(json.dumps(out))
