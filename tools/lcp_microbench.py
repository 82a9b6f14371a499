"""GPU micro-benchmark of the LCP kernel on easy / hard / mixed batches."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moby_amd import _lib, synth
from moby_amd.lcp import LCPDevice

def run(name, Mh, qh, B=4096, kind=_lib.MH_LCP_FAST_REG, opts=(-20, 4, -8), iters=20, z0=None):
    dev = torch.device("cuda")
    n = qh.shape[1]
    reps = (B + len(qh) - 1) // len(qh)
    M = torch.from_numpy(np.ascontiguousarray(np.transpose(np.tile(Mh, (reps, 1, 1))[:B], (0, 2, 1)))).to(dev)
    q = torch.from_numpy(np.tile(qh, (reps, 1))[:B].copy()).to(dev)
    z0t = torch.zeros(B, n, dtype=torch.float64, device=dev) if z0 is None else torch.from_numpy(np.tile(z0, (reps, 1))[:B].copy()).to(dev)
    z = z0t.clone()
    s = LCPDevice(B, dev); rng0 = s.rng.clone()
    o = _lib.mh_lcp_opts(opts[0], opts[1], opts[2], -1.0, -1.0)
    ts = []
    for i in range(iters + 3):
        z.copy_(z0t); s.rng.copy_(rng0)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); s.solve(kind, M, q, z, o); e1.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(e0.elapsed_time(e1))
    piv = s.pivots.cpu().numpy()
    print("%-28s n=%2d B=%d  %9.1f us/launch  ok=%d  pivots mean %.1f max %d  -> %.3g rows/s, %.1f GB/s alg" % (
        name, n, B, np.median(ts) * 1e3, int(s.status.sum()), piv.mean(), piv.max(),
        B * n / (np.median(ts) * 1e-3), 8.0 * (n * n + 2 * n) * B / (np.median(ts) * 1e-3) / 1e9))
    return z.cpu().numpy()

if __name__ == "__main__":
    M64, q64 = synth.sphere_stack_impact_lcp(64)
    run("world0 cold(z=0)", M64[:1], q64[:1])
    zsol = run("world0 again", M64[:1], q64[:1])
    run("world0 warm(z=solution)", M64[:1], q64[:1], z0=zsol[:1])
    run("mixed 64 perturbed", M64, q64)
    Mp, qp = synth.random_lcp(64, 42, "pd", seed=1)
    run("random pd n=42 fast", Mp, qp)
    run("random pd n=42 lemke", Mp, qp, kind=_lib.MH_LCP_LEMKE)
    Mp, qp = synth.random_lcp(64, 5, "pd", seed=1)
    run("random pd n=5 fast", Mp, qp, kind=_lib.MH_LCP_FAST)
