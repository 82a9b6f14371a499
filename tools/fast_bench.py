"""The lcp_fast kinds of the workgroup-per-problem LCP solver alone (mh_lcp_solve_batch_dev, MH_LCP_FAST_REG with the impact handler's
ladder (-20, 4, -8), ICH-QP:219) on the impact LCPs of `nboxes`-box stacks: W different worlds x `copies` copies, device resident,
HIP-event timed.  With a -DMH_BLK_PROF build (tools/build_variant.sh prof -DMH_BLK_PROF, MOBY_HIP_LIB=...) blocks 0 and 5 print their
per-phase cycles: where an lcp_fast iteration spends its time at config 4's sizes.
  python tools/fast_bench.py nboxes W copies [geom]      geom: 0 auto, 1 = 256 threads, 2 = 1024 threads per problem"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moby_amd import _lib, impact as I
from moby_amd.lcp import LCPDevice, MH_LCP_FAST_REG
from moby_amd._lib import mh_lcp_opts

nbx, W, copies = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
geom = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if geom:
    _lib.check(_lib.load().mh_debug_set(2, geom))
if os.environ.get("MH_REG_LU"):            # mh_debug_set key 10 (0: the nonbasic systems through the HBM workspace)
    _lib.check(_lib.load().mh_debug_set(10, int(os.environ["MH_REG_LU"])))
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
opts = mh_lcp_opts(-20, 4, -8, -1.0, -1.0)
out = []
for rep in range(2):
    lcp = LCPDevice(B)
    z.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); st_ = lcp.solve(MH_LCP_FAST_REG, Mcm, q, z, opts, zs); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    piv = lcp.pivots.cpu().numpy().astype(np.int64)
    out.append({"ms": ms, "pivots_total": int(piv.sum()), "pivots_max": int(piv.max()), "solved": int(st_.cpu().numpy().sum())})
print(json.dumps({"nboxes": nbx, "n": n, "worlds": W, "copies": copies, "problems": B, "runs": out,
                  "z_checksum": float(z.abs().sum().item())}))
