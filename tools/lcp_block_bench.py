import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moby_amd import synth
from moby_amd.lcp import LCP
n, B = int(sys.argv[1]), int(sys.argv[2])
M, q = synth.random_lcp(min(B, 16), n, "pd", seed=n)
reps = B // M.shape[0]
M = np.tile(M, (reps, 1, 1)); q = np.tile(q, (reps, 1))
lcp = LCP(M.shape[0]); z = np.zeros_like(q)
for _ in range(2):
    t0 = time.perf_counter(); ok = lcp.lcp_fast(M, q, z, z_size=np.zeros(M.shape[0], dtype=np.int32)); t = time.perf_counter() - t0
print(json.dumps({"n": n, "B": int(M.shape[0]), "ms": t * 1e3, "pivots_mean": float(lcp.pivots.mean()), "pivots_max": int(lcp.pivots.max()), "ok": int(ok.sum())}))
