"""Phase profile of the sphere-stack batch after `skip` steps (how the per-step cost evolves)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moby_amd import _lib, scene as S
from moby_amd.world import WorldBatchDevice
from tools.world_profile import NAMES
import torch
if __name__ == "__main__":
    B = 4096
    lib = _lib.load()
    wb = WorldBatchDevice(S.sphere_stack_scene(), S.sphere_stack_state_range(0, B))
    done = 0
    for skip in [int(x) for x in sys.argv[1:]]:
        if skip > done:
            wb.step(1e-3, skip - done); done = skip
        torch.cuda.synchronize()
        _, a0 = wb.download()
        t0 = time.perf_counter(); wb.step(1e-3, 200); torch.cuda.synchronize(); t = time.perf_counter() - t0; done += 200
        _, a1 = wb.download()
        d = lambda f: (a1[f].astype(np.int64) - a0[f].astype(np.int64))
        print("steps %5d..%5d: %.2f ms  pivots/step %.2f  solves/step %.2f  rows/solve %.1f  mini/step %.2f  stab iters/step %.3f (max world %.2f)  status %s"
              % (done - 200, done, t * 1e3, d("lcp_pivots").mean() / 200, d("lcp_solves").mean() / 200, d("lcp_rows").sum() / max(1, d("lcp_solves").sum()),
                 d("mini_steps").mean() / 200, d("stab_iters").mean() / 200, d("stab_iters").max() / 200, sorted(set(a1["status"].tolist()))))
        ph = np.zeros(len(NAMES) + 2)
        _lib.check(lib.mh_world_batch_profile(wb.handle, 1e-3, 200, ph.ctypes.data, len(ph))); done += 200
        tot = ph[:10].sum()
        print("   " + "  ".join("%s %.0f%%" % (n.strip(), 100 * c / tot) for n, c in zip(NAMES[:10], ph[:10])) + "  | lemke %.0f%% total %.0f slowest %.0f" % (100 * ph[16] / tot, tot / 200, ph[-2] / 200))
