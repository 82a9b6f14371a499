"""Diagnostic (GPU): seconds per step of the stabilising articulated kernels on a few bodies, next to the oracle's -- written when the floating torso + foot
(7 joints, 2 spheres) was seen to need 0.1-0.4 s per step with the stabiliser on.  Cause (found by counting Artic::kinematics calls in a scratch copy of the oracle):
not the stabiliser -- it parks a sphere ~2.5e-8 above the floor, and the next steps' conservative advancement inside do_mini_step (TSS:114-222) creeps towards the floor in
steps of min_step_size = 1.5e-8 s: ~20 000 passes per step of 1e-3 s until the body has settled, on the CPU (13 ms per step) as on the device.  The reference's algorithm.  python tests/tools/floating_stab_timing.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from moby_amd import artic as A, scene as S
from tests.oracle_api import Oracle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
sc = lambda f: os.path.join(ROOT, "tests", "scenes", f)


def case(name, m, q0, qd0, dt, steps=20, B=3):
    q = np.tile(q0, (B, 1)); qd = np.tile(qd0, (B, 1))
    ab = A.ArticBatch(m, q, qd)
    qo, qdo, auxo = q.copy(), qd.copy(), S.new_aux(B)
    ab.step(dt, 1); torch.cuda.synchronize()
    t0 = time.perf_counter(); ab.step(dt, steps); torch.cuda.synchronize(); t1 = time.perf_counter()
    o.artic_step(m, qo, qdo, auxo, dt, steps + 1); t2 = time.perf_counter()
    _, _, aux = ab.download(); ab.close()
    print("%-46s gpu %.4f s/step  oracle %.5f s/step  stab_iters %s mini %s status %s" % (name, (t1 - t0) / steps, (t2 - t1) / (steps + 1), aux["stab_iters"][0], aux["mini_steps"][0], aux["status"][0]), flush=True)


m, _, _, q0, qd0, dt = A.load_xml(sc("floating_hinged_pair.xml"))
case("floating pair, stabiliser 10", m, q0, qd0, dt)
m.cstab_max_iterations = 0; case("floating pair, stabiliser off", m, q0, qd0, dt)
m.cstab_max_iterations = 10; m.floating_base = 0; case("same joints, floating_base = 0", m, q0, qd0, dt)
m.floating_base = 1; m.nspheres = 0; case("floating pair, no spheres (limit rows only)", m, q0, qd0, dt)
m, _, _, q0, qd0, dt = A.load_xml(sc("floating_hinged_pair.xml"))
for v in range(6): m.lolimit[v] = -1e6; m.hilimit[v] = 1e6
case("floating pair, finite limits on the virtual joints", m, q0, qd0, dt)
m, _, _, q0, qd0, dt = A.load_xml(sc("floating_spinning_ball.xml")); m.cstab_max_iterations = 10
case("floating ball, stabiliser 10 (dt 0.025)", m, q0, qd0, dt)
m, _, _, q0, qd0, dt = A.load_xml(sc("arm_on_table.xml")); m.cstab_max_iterations = 10
case("arm on table (3 joints), stabiliser 10", m, q0, qd0, dt)

# the perturbed copies of tests/test_artic_floating_gpu.py's first version (seed 77, four worlds), one world per batch
m, _, _, q0, qd0, dt = A.load_xml(sc("floating_hinged_pair.xml"))
B = 4
rng = np.random.default_rng(77)
q = np.tile(q0, (B, 1)); qd = np.tile(qd0, (B, 1))
q[1:, :3] += rng.uniform(-0.05, 0.05, (B - 1, 3)); q[1:, 3:6] += rng.uniform(-0.3, 0.3, (B - 1, 3)); q[1:, 6] = rng.uniform(-0.5, 0.3, B - 1)
qd[1:] += rng.uniform(-0.5, 0.5, (B - 1, 7))
for w in range(B):
    ab = A.ArticBatch(m, q[w:w + 1], qd[w:w + 1])
    for c in range(5):
        t0 = time.perf_counter(); ab.step(dt, 30); torch.cuda.synchronize(); t1 = time.perf_counter()
        _, _, aux = ab.download()
        print("world %d steps %3d: gpu %.4f s/step  stab_iters %d stab_rows %d lcp_solves %d pivots %d mini %d status %d" % (w, 30 * (c + 1), (t1 - t0) / 30, aux["stab_iters"][0], aux["stab_rows"][0], aux["lcp_solves"][0], aux["lcp_pivots"][0], aux["mini_steps"][0], aux["status"][0]), flush=True)
    ab.close()
