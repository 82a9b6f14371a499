"""BASELINE config 4 through the impact-handler entry (include/moby_hip_impact.h): B box stacks of `nboxes` boxes
(4 corner contacts per interface, NK = 4: n = 32 nboxes), device-resident.  One "step" = one process_constraints
call on every world; the first call is cold (z = 0), the following ones warm-started from _zlast after gravity has
acted for another dt on the resting stack.  Prints one JSON line per (nboxes, B).
  python tools/impact_bench.py "nboxes:B[:warm_calls]" ...
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moby_amd import _lib, impact as I


def case(nbx, B, warm_calls=5):
    lib = _lib.load()
    mass, J, st, cs = I.box_stack(nbx, B=B)
    nc = 4 * nbx
    ib = I.ImpactBatch(B, nbx, nc, 4, mass, J)
    ib.upload(st, cs)
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ib.process_async(stream); e1.record(); torch.cuda.synchronize()
    cold_ms = e0.elapsed_time(e1)
    r = ib.download()
    out = {"case": "config 4 box stack via impact handler", "nboxes": nbx, "nc": nc, "n": ib.n, "worlds": B,
           "cold_ms": cold_ms, "cold_pivots_mean": float(r["pivots"].mean()), "cold_pivots_max": int(r["pivots"].max()),
           "cold_rows_per_s": ib.n * float(r["solves"].sum()) / (cold_ms * 1e-3),
           "alg_bytes_per_solve": 8 * (ib.n * ib.n + 2 * ib.n), "bad_worlds": int((r["status"] & ~2 != 0).sum()),
           "status_flags_per_world": [int(x) for x in r["status"][:64]], "pivots_per_world": [int(x) for x in r["pivots"][:64]]}
    warm = []
    piv = []
    for _ in range(warm_calls):
        s = r["state"].reshape(B, nbx, 13); s[:, :, 8] += -9.81e-3          # gravity for another dt; contacts unchanged
        ib.upload(s.reshape(B, -1), cs)
        e0.record(); ib.process_async(stream); e1.record(); torch.cuda.synchronize()
        warm.append(e0.elapsed_time(e1))
        r = ib.download(); piv.append(float(r["pivots"].mean()))
    if warm:
        w = float(np.median(warm))
        out.update({"warm_ms": w, "warm_pivots_mean": float(np.mean(piv)), "warm_world_steps_per_s": B / (w * 1e-3),
                    "warm_rows_per_s": ib.n * B / (w * 1e-3), "warm_GBps_alg": 8 * (ib.n * ib.n + 2 * ib.n) * B / (w * 1e-3) / 1e9,
                    "bad_worlds_warm": int((r["status"] & ~2 != 0).sum())})
    print(json.dumps(out), flush=True)
    ib.close()


if __name__ == "__main__":
    if os.environ.get("MH_BLK_GEOM"):          # 1 = 256-thread block solver, 2 = 1024-thread (mh_debug_set key 2)
        _lib.check(_lib.load().mh_debug_set(2, int(os.environ["MH_BLK_GEOM"])))
    for a in sys.argv[1:]:
        p = [int(x) for x in a.split(":")]
        case(p[0], p[1], p[2] if len(p) > 2 else 5)
