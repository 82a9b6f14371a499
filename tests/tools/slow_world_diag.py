"""Why are the slow worlds of a long run slow?  Steps the headline batch (sphere-stack x B) to step `start` on the GPU, takes the worlds
with the most pivots over the next 200 steps, and steps THOSE on the CPU oracle for 50 steps with lcp_fast's iteration statistics
switched on (oracle_dbg_fast_repeats): iterations run, iterations spent on the index set of the iteration before, on one of the 2..8
before that, draws that decided something, calls that ran into MAX_PIV.
python tests/tools/slow_world_diag.py [start = 4200] [B = 4096] [worlds = 6]"""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from moby_amd import scene as S
from moby_amd.world import WorldBatchDevice
from tests.oracle_api import Oracle

start = int(sys.argv[1]) if len(sys.argv) > 1 else 4200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 6
sc = S.sphere_stack_scene()
wb = WorldBatchDevice(sc, S.sphere_stack_state_range(0, B))
stream = torch.cuda.current_stream().cuda_stream
left = start
while left > 0:
    k = min(left, 1000); wb.step(1e-3, k, stream); torch.cuda.synchronize(); left -= k
st0, a0 = wb.download()
wb.step(1e-3, 200, stream); torch.cuda.synchronize()
_, a1 = wb.download()
piv = a1["lcp_pivots"].astype(np.int64) - a0["lcp_pivots"].astype(np.int64)
order = np.argsort(-piv)
print(json.dumps({"start": start, "pivots_per_world_step_mean": float(piv.mean()) / 200, "median": float(np.median(piv)) / 200,
                  "top": [(int(w), float(piv[w]) / 200) for w in order[:nw]]}))
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
for w in list(order[:nw]) + [int(order[B // 2])]:
    s = st0[w:w + 1].copy(); a = a0[w:w + 1].copy()
    o.lib.oracle_dbg_fast_repeats(1, None)
    p0 = int(a["lcp_pivots"][0])
    o.world_step_batch(sc, s, a, 1e-3, 50)
    out = np.zeros(13, dtype=np.uint64); o.lib.oracle_dbg_fast_repeats(0, out.ctypes.data_as(ctypes.c_void_p))
    print(json.dumps({"world": int(w), "gpu_pivots_per_step": float(piv[w]) / 200, "oracle_pivots_per_step_next_50": (int(a["lcp_pivots"][0]) - p0) / 50.0,
                      "lcp_fast_iterations": int(out[0]), "on_the_previous_set": int(out[1]), "on_one_of_the_2_to_8_before": int(out[2]),
                      "deciding_draws": int(out[3]), "calls_into_MAX_PIV": int(out[4])}))

# where the slow worlds' time goes on the device: a batch of the 64 slowest alone (and one of 64 median worlds) under the stamped launch
from moby_amd import _lib
from tools.world_profile import NAMES
lib = _lib.load()
for label, ids in (("64 slowest", order[:64]), ("64 median", order[B // 2 - 32:B // 2 + 32])):
    ids = np.sort(ids)
    wb2 = WorldBatchDevice(sc, st0[ids].copy(), aux=a0[ids].copy())
    ph = np.zeros(len(NAMES) + 4)
    _lib.check(lib.mh_world_batch_profile(wb2.handle, 1e-3, 200, ph.ctypes.data, len(ph)))
    tot = ph[:10].sum()
    print(label + ": cycles/world-step %.0f  " % (tot / 200) + "  ".join("%s %.0f%%" % (n.strip(), 100 * c / tot) for n, c in zip(NAMES, ph[:len(NAMES)])))
    wb2.close()
