"""The Lemke kinds of the workgroup-per-problem LCP solver alone (mh_lcp_solve_batch_dev, MH_LCP_LEMKE_REG: the whole ladder in sequence
inside ONE workgroup per problem) on the impact LCPs of `nboxes`-box stacks: W different worlds x `copies` copies, device resident,
HIP-event timed.  What it isolates: pivots per second of lcp_lemke's LU under a given load (copies x W workgroups).
  python tools/lemke_bench.py nboxes W copies [geom]      geom: 0 auto, 1 = 256 threads, 2 = 1024 threads per problem"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moby_amd import _lib, impact as I
from moby_amd.lcp import LCPDevice, MH_LCP_LEMKE_REG

nbx, W, copies = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
geom = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if geom:
    _lib.check(_lib.load().mh_debug_set(2, geom))
mass, J, st, cs = I.box_stack(nbx, B=max(W, 2))
ib = I.ImpactBatch(max(W, 2), nbx, 4 * nbx, 4, mass, J)
ib.upload(st, cs); ib.process_async(); ib.download()
MM, qq = ib.debug_lcp()                      # row-major M[b, r, c]
ib.close()
MM, qq = MM[:W], qq[:W]
n = qq.shape[1]; B = W * copies
Mcm = torch.from_numpy(np.ascontiguousarray(np.tile(np.transpose(MM, (0, 2, 1)), (copies, 1, 1)))).cuda()
q = torch.from_numpy(np.tile(qq, (copies, 1))).cuda()
z = torch.zeros((B, n), dtype=torch.float64, device="cuda")
zs = torch.zeros(B, dtype=torch.int32, device="cuda")      # z.size() = 0: cold
out = []
for rep in range(2):
    lcp = LCPDevice(B)
    z.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); st_ = lcp.solve(MH_LCP_LEMKE_REG, Mcm, q, z, None, zs); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    piv = lcp.pivots.cpu().numpy().astype(np.int64)
    out.append({"ms": ms, "pivots_total": int(piv.sum()), "pivots_max": int(piv.max()), "solved": int(st_.cpu().numpy().sum()),
                "us_per_pivot_of_the_longest": 1e3 * ms / max(1, int(piv.max())), "pivots_per_sec": float(piv.sum()) / (ms * 1e-3)})
print(json.dumps({"nboxes": nbx, "n": n, "worlds": W, "copies": copies, "problems": B, "runs": out,
                  "z_checksum": float(z.abs().sum().item())}))
