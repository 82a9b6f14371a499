"""In-kernel phase profile of the world-step kernel (diagnostic launch with s_memtime stamps)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moby_amd import _lib, scene as S
from moby_amd.world import WorldBatchDevice
import torch

NAMES = ["broad+CA", "integrate", "fwd dyn", "contacts", "islands", "problem data", "M build", "LCP solve", "apply/update", "stabilise",
         " lcp:setup", " lcp:gather", " lcp:LU", " lcp:gemv", " lcp:randmin", " lcp:verify", " lcp:lemke"]
if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    lib = _lib.load()
    which = sys.argv[3] if len(sys.argv) > 3 else "stack"
    if which == "wheel":
        from moby_amd import synth
        sc = S.rimless_wheel_scene()
        wb = WorldBatchDevice(sc, S.rimless_wheel_state([0.24 if w == 0 else 0.2 + 0.4 * synth.world_uniforms(w, 1)[0] for w in range(B)]))
    else:
        sc = S.sphere_stack_scene()
        wb = WorldBatchDevice(sc, S.sphere_stack_state_range(0, B))
    print("runtime occupancy query: %d workgroups per CU" % lib.mh_world_batch_occupancy(wb.handle))
    wb.step(1e-3, 20); torch.cuda.synchronize()
    t0 = time.perf_counter(); wb.step(1e-3, nsteps); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("plain launch: %.3f ms for %d steps x %d worlds -> %.1f us per batch step, %.3g world-steps/s" % (t * 1e3, nsteps, B, t / nsteps * 1e6, B * nsteps / t))
    ph = np.zeros(len(NAMES) + 2)
    _lib.check(lib.mh_world_batch_profile(wb.handle, 1e-3, nsteps, ph.ctypes.data, len(ph)))
    tot = ph[:10].sum()
    for n, c in zip(NAMES, ph[:len(NAMES)]):
        print("  %-14s %10.0f cycles/world-step  %5.1f %%" % (n, c / nsteps, 100 * c / tot))
    print("  total stamped  %10.0f cycles/world-step" % (tot / nsteps))
    print("  slowest world  %10.0f   fastest %10.0f cycles/world-step (the launch lasts as long as the slowest)" % (ph[-2] / nsteps, ph[-1] / nsteps))
