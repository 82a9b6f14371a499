"""Experiment: BASELINE config 4 at the bench size stepped as G independent batches (worlds g, g + G, g + 2G, ...), each from a host thread of its own on a
stream of its own.  Worlds are independent, so the states must equal the one-batch run's bit for bit; what changes is that the solver rounds of one group -- in
particular the dozen small rounds of the few worlds whose conservative advancement splits a step -- overlap the other groups' big ones instead of following them.
python tools/config4_groups.py [boxes] [worlds] [steps] [groups ...]   (prints one JSON line per group count)"""
import json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moby_amd import scene as S, stack as K

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
groups = [int(a) for a in sys.argv[4:]] or [1, 4]
sc = K.box_stack_scene(N)
st0 = K.box_stack_state(N, B)
ref = None
for G in groups:
    bbs = [K.BigBatch(sc, np.ascontiguousarray(st0[g::G])) for g in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    secs = []
    for k in range(steps):
        errs = []
        def run(g):
            try:
                bbs[g].step(1e-3, 1, stream=streams[g].cuda_stream)
                streams[g].synchronize()
            except Exception as e:      # noqa: BLE001 -- reported below
                errs.append((g, repr(e)))
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(g,)) for g in range(G)]
        for t in th: t.start()
        for t in th: t.join()
        secs.append(time.perf_counter() - t0)
        print("groups %d step %d: %.2f s %s" % (G, k, secs[-1], errs), file=sys.stderr, flush=True)
    st = np.zeros_like(st0); status = np.zeros(B, dtype=np.int64); piv = np.zeros(B, dtype=np.int64)
    for g in range(G):
        s, a = bbs[g].download()
        st[g::G] = s; status[g::G] = a["status"]; piv[g::G] = a["lcp_pivots"]
        bbs[g].close()
    if ref is None: ref = (st, status, piv)
    print(json.dumps({"workload": "box stack of %d x %d worlds, %d full steps" % (N, B, steps), "groups": G, "seconds_per_step": secs,
                      "equal_to_first_run": bool(np.array_equal(st, ref[0], equal_nan=True) and np.array_equal(status, ref[1]) and np.array_equal(piv, ref[2])),
                      "worlds_lcp_failed": int(((status & S.MH_WORLD_LCP_FAILED) != 0).sum())}), flush=True)
