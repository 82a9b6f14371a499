"""Generates tests/golden/config4_64_boxes.npz: BASELINE config 4 at the size BASELINE states (a stack of 64 boxes, impact LCP
n = 2048; the pattern of /root/reference/example/stacks/stack.xml:36-96 with regress/stacks.setup's dt = 1e-3) stepped by the CPU
ORACLE (oracle/world.hpp big_step: the restatement of TimeSteppingSimulator::step, src/TimeSteppingSimulator.cpp:52-111), in the
build container.  The oracle needs minutes per world here -- far too slow to run on the GPU box beside the device -- so what it
leaves behind is committed as a fixture and tests/test_big_gpu.py holds the device to it bit for bit:

  per world w in WORLDS of moby_amd.stack.box_stack_state(64, BATCH): the start state, the state after STEPS full steps, the whole
  mh_world_aux record (status flags, rand() ring, pivot / solve / row / mini-step / stabiliser counters, time) and the handler's
  _zlast after the last step.

lcp_lemke's bases go through the oracle's bit-equal structure-exploiting model (oracle_dbg_lemke_compact; tests/test_oracle_compact_lu.py
holds it to the dense dgesv): a dense dgesv of a 2048 x 2048 basis per pivot would take an hour per rung.

python tests/golden/make_config4_64_boxes.py [world ...]     (one process per world; merges what is already there)
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_api import Oracle
from moby_amd import scene as S, stack as K

# MH_FIXTURE_BOXES / MH_FIXTURE_STEPS: the same generator for other stack heights (32 boxes x 2 steps -> config4_32_boxes_2steps.npz: the largest stack the
# reference's chain solves, a cold and a warm step; 10-25 minutes of CPU per world)
# MH_FIXTURE_BATCH: the batch the worlds are taken from (the perturbations are drawn per batch): 1024 with 16 boxes and 2 steps -> config4_16_boxes_x1024_2steps.npz, the
# worlds of the bench-size batch that tests/test_big_gpu.py::test_config4_bench_size_full_batch holds the default full-chip schedule to (world 0 and the worlds with the
# most pivots over the two steps: 497, 328)
BOXES, BATCH, STEPS, DT = int(os.environ.get("MH_FIXTURE_BOXES", "64")), int(os.environ.get("MH_FIXTURE_BATCH", "8")), int(os.environ.get("MH_FIXTURE_STEPS", "1")), 1e-3
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config4_%d_boxes%s%s.npz" % (BOXES, "" if BATCH == 8 else "_x%d" % BATCH, "" if STEPS == 1 else "_%dsteps" % STEPS))


def run_world(w):
    o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    o.lib.oracle_dbg_lemke_compact(8)
    sc = K.box_stack_scene(BOXES)
    cap = sc.lcp_capacity()
    st0 = K.box_stack_state(BOXES, BATCH)[w].copy()
    so = st0.copy(); ao = S.new_aux(1); zl = np.zeros(cap); zb = np.zeros(cap)
    t0 = time.perf_counter()
    for _ in range(STEPS):
        o.big_step(sc, so, ao, DT, 1, zlast=zl, zbuf=zb, cap=cap)
    return dict(st0=st0, st=so, aux=ao, zlast=zl, seconds=time.perf_counter() - t0)


def stored():
    have = {}
    if os.path.exists(OUT):
        d = np.load(OUT)
        for k, w in enumerate(d["worlds"]):
            have[int(w)] = dict(st0=d["st0"][k], st=d["st"][k], aux=d["aux"][k:k + 1], zlast=d["zlast"][k], seconds=float(d["oracle_seconds"][k]))
    return have


def main():
    import fcntl
    worlds = [int(a) for a in sys.argv[1:]] or [0, 1]
    for w in worlds:
        r = run_world(w)
        lock = open(OUT + ".lock", "w")
        fcntl.flock(lock, fcntl.LOCK_EX)          # (several processes, one per world, may finish at once)
        have = stored()
        have[w] = r
        print("world %d: %.1f s, status %d, pivots %d, solves %d, mini-steps %d" % (w, r["seconds"], int(r["aux"]["status"][0]), int(r["aux"]["lcp_pivots"][0]),
                                                                                 int(r["aux"]["lcp_solves"][0]), int(r["aux"]["mini_steps"][0])), flush=True)
        ws = sorted(have)
        np.savez_compressed(OUT, boxes=BOXES, batch=BATCH, steps=STEPS, dt=DT, worlds=np.array(ws),
                            st0=np.stack([have[k]["st0"] for k in ws]), st=np.stack([have[k]["st"] for k in ws]),
                            aux=np.concatenate([have[k]["aux"] for k in ws]), zlast=np.stack([have[k]["zlast"] for k in ws]),
                            oracle_seconds=np.array([have[k]["seconds"] for k in ws]))
        fcntl.flock(lock, fcntl.LOCK_UN); lock.close()


if __name__ == "__main__":
    main()
