"""GPU parity of floating-base articulated bodies (mh_artic_model.floating_base: six virtual joints under the base link, include/moby_hip_artic.h) against
oracle/artic.hpp, bit for bit through the C ABI: the contact kernels (Drumwright-Shell and no-slip models), the stabiliser's contact + limit rows, conservative
advancement with the base's velocity, forward dynamics and calc_jacobian with the six base columns."""
import ctypes
import os

import numpy as np
import pytest

from moby_amd import artic as A
from moby_amd import scene as S
from tests.test_artic_contacts_gpu import run
from tests.test_artic_gpu import assert_parity

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
BALL = os.path.join(HERE, "scenes", "floating_spinning_ball.xml")
PAIR = os.path.join(HERE, "scenes", "floating_hinged_pair.xml")


def test_floating_ball_bounces_like_the_oracle(oracle):
    """the free ball as a floating base of one link (tests/test_artic_floating.py ties the oracle's run to the rigid-body stepper): perturbed drops, restitution 1,
    no friction -> the Drumwright-Shell model on the base link's contact; three seconds of bounces at two step sizes"""
    m, _, _, q0, qd0, dt = A.load_xml(BALL)
    B = 6
    rng = np.random.default_rng(31)
    q = np.tile(q0, (B, 1)); qd = np.tile(qd0, (B, 1))
    q[1:, :3] += rng.uniform(-0.2, 0.2, (B - 1, 3)); qd[1:, :3] += rng.uniform(-0.5, 0.5, (B - 1, 3)); qd[1:, 3] += rng.uniform(-3, 3, B - 1)
    aux = run(oracle, m, q, qd, nsteps=30, chunks=4, dt=dt)
    assert (aux["lcp_solves"] > 0).all() and (aux["mini_steps"] > aux["steps"]).all() and (aux["status"] == 0).all()
    aux = run(oracle, m, q, qd, nsteps=500, chunks=4, dt=1e-3)
    assert (aux["lcp_solves"] > 0).all() and (aux["status"] == 0).all()


@pytest.mark.parametrize("mu,iters", [(100.0, 10), (0.5, 10), (100.0, 0)])
def test_floating_hinged_pair_lands_like_the_oracle(oracle, mu, iters):
    """tests/scenes/floating_hinged_pair.xml: torso + foot on a hinge with limits, thrown onto the floor with spin; contact rows on the base link AND on the hinged link,
    the hinge's limit, the stabiliser's mixed LCP (iters > 0) -- no-slip model (mu = 100) and Drumwright-Shell (mu = 0.5); perturbed copies of the file's state"""
    m, _, _, q0, qd0, dt = A.load_xml(PAIR)
    m.cp_mu_coulomb = mu; m.cstab_max_iterations = iters
    # with the stabiliser on the run is short and starts just above the floor: the stabiliser parks a sphere ~2.5e-8 ABOVE the floor, and do_mini_step's conservative
    # advancement (TSS:114-222) then creeps towards it in steps of min_step_size = 1.5e-8 s -- some twenty thousand kinematics passes per step of 1e-3 s while the body
    # settles, in the oracle (13 ms per step) as on the device (0.1-0.4 s per step on one wavefront; tests/tools/floating_stab_timing.py).  The reference's algorithm, not
    # a defect of either side -- so the long run with the hinge driven into its limit is the iters = 0 one
    B = 3 if iters else 8
    rng = np.random.default_rng(79 if iters else 77)
    q = np.tile(q0, (B, 1)); qd = np.tile(qd0, (B, 1))
    q[1:, :3] += rng.uniform(-0.05, 0.05, (B - 1, 3)); q[1:, 3:6] += rng.uniform(-0.3, 0.3, (B - 1, 3)); q[1:, 6] = rng.uniform(-0.5, 0.3, B - 1)
    qd[1:] += rng.uniform(-0.5, 0.5, (B - 1, 7))
    if iters: q[:, 1] -= 0.07
    aux = run(oracle, m, q, qd, nsteps=20 if iters else 150, chunks=4, dt=dt)
    assert (aux["lcp_solves"] > 0).all() and (aux["mini_steps"] > aux["steps"]).all()
    assert ((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) == 0).all()
    if iters:
        assert (aux["stab_iters"] > 0).any()


def test_floating_forward_dynamics_and_jacobian_seams(oracle):
    """seam B4 and calc_jacobian with the six base columns: qdd, H(q), link poses and J at random states, both algorithms"""
    m, _, _, _, _, _ = A.load_xml(PAIR)
    B = 6
    rng = np.random.default_rng(3)
    q0 = rng.uniform(-0.6, 0.6, (B, 7)); qd0 = rng.uniform(-1, 1, (B, 7)); tau = rng.uniform(-1, 1, (B, 7))
    pts = rng.uniform(-0.5, 0.5, (B, 3))
    for alg in (A.MH_ARTIC_CRB, A.MH_ARTIC_FSAB):
        m.algorithm = alg
        ab = A.ArticBatch(m, q0, qd0)
        qdd, H = ab.fwd_dyn(tau); poses = ab.link_poses()
        for w in range(B):
            r = oracle.artic_fwd_dyn(m, q0[w], qd0[w], tau[w])
            assert np.array_equal(qdd[w], r["qdd"]) and np.array_equal(H[w], r["H"]) and np.array_equal(poses[w], r["poses"])
        for link in (5, 6):
            J = ab.jacobian(link, pts)
            for w in range(B):
                assert np.array_equal(J[w], oracle.artic_jacobian(m, q0[w], link, pts[w]))
        ab.close()
    # a free body falls: the sliders' accelerations are gravity whatever the pose, the hinges' bias is the gyroscopic one
    m.algorithm = A.MH_ARTIC_CRB
    ab = A.ArticBatch(m, q0, np.zeros_like(qd0)); qdd, _ = ab.fwd_dyn(None, want_H=False); ab.close()
    assert np.allclose(qdd[:, :3], [0.0, -9.81, 0.0], atol=1e-12)


def test_create_refuses_a_floating_layout_it_does_not_know():
    """mh_artic_batch_create checks the virtual joints conservative advancement relies on, and that every joint moves mass"""
    from moby_amd import _lib
    lib = _lib.load()
    def create(m):
        h = ctypes.c_void_p()
        rc = lib.mh_artic_batch_create(ctypes.byref(m), 1, ctypes.byref(h))
        if rc == 0: lib.mh_artic_batch_destroy(h)
        return rc
    m = A.load_xml(PAIR)[0]
    assert create(m) == 0
    m.jtype[1] = A.MH_JOINT_REVOLUTE
    assert create(m) != 0 and b"virtual joint" in lib.mh_last_error()
    m = A.load_xml(PAIR)[0]; m.trel[2][0] = 0.1
    assert create(m) != 0
    m = A.load_xml(PAIR)[0]; m.floating_base = 2
    assert create(m) != 0
    m = A.load_xml(PAIR)[0]; m.mass[5] = 0.0; m.mass[6] = 0.0
    assert create(m) != 0 and b"carries no mass" in lib.mh_last_error()
    m = A.load_xml(PAIR)[0]; m.mass[5] = -1.0
    assert create(m) != 0


@pytest.mark.parametrize("iters", [0, 10])
def test_floating_body_without_geometry_keeps_its_hinge_limits(oracle, iters):
    """the torso + foot in free fall, no collision geometry: the kernels of config 5 (one mini-step per step; the stabilising one for iters > 0) with six virtual columns
    in H, the hinge driven into both limits (and started beyond them) -> limit rows through the no-slip path with NC = 0.  The stabiliser itself never acts here, on either
    side: evaluate_unilateral_constraints reads the slacks of joint [body index] = joint 0 (CStab:117, kept), which for a floating body is a virtual slider without limits."""
    m, _, _, q0, qd0, dt = A.load_xml(PAIR)
    m.nspheres = 0; m.cstab_max_iterations = iters
    B = 6
    rng = np.random.default_rng(5)
    q = np.tile(q0, (B, 1)); qd = np.tile(qd0, (B, 1))
    q[:, 6] = [0.0, 0.45, -0.7, 0.39, -0.59, 0.2]; qd[:, 6] = [1.0, 2.0, -2.0, 3.0, -3.0, 0.0]; qd[1:, :6] += rng.uniform(-1, 1, (B - 1, 6))
    ab = A.ArticBatch(m, q, qd)
    aux = assert_parity(ab, oracle, m, q, qd, dt, 75, 4)
    ab.close()
    assert (aux["lcp_solves"][1:5] > 0).all() and (aux["status"] == 0).all() and (aux["stab_iters"] == 0).all()


def test_floating_body_with_welded_links_lands_like_the_oracle(oracle):
    """tests/scenes/floating_welded_pair.xml (FixedJoints welded by the reader: full tensors on the base link and the foot, two spheres on each, a tail hinged to the
    welded hat): perturbed copies thrown onto the floor, no-slip model, four contact rows at most + two hinges' limits"""
    m, _, _, q0, qd0, dt = A.load_xml(os.path.join(HERE, "scenes", "floating_welded_pair.xml"))
    B = 4
    rng = np.random.default_rng(44)
    q = np.tile(q0, (B, 1)); qd = np.tile(qd0, (B, 1))
    q[1:, :3] += rng.uniform(-0.05, 0.05, (B - 1, 3)); q[1:, 3:6] += rng.uniform(-0.3, 0.3, (B - 1, 3)); q[1:, 6:] = rng.uniform(-0.5, 0.3, (B - 1, 2))
    qd[1:] += rng.uniform(-0.5, 0.5, (B - 1, 8))
    aux = run(oracle, m, q, qd, nsteps=60, chunks=4, dt=dt)
    assert (aux["lcp_solves"] > 0).all() and (aux["mini_steps"] > aux["steps"]).all() and ((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) == 0).all()
    assert (aux["lcp_rows"] > aux["lcp_solves"]).any()


@pytest.mark.parametrize("variant", ["config5", "pack", "contacts", "contacts+stab", "stab"])
def test_an_exception_ends_an_articulated_worlds_run(oracle, variant):
    """MH_WORLD_LCP_FAILED on an articulated world is the end of its run, as on every other stepper: in a batch of four, worlds 1 and 3 carry the flag at upload
    -- what an LCPSolverException, a generalized inertia that is not positive definite or a failed compute_X leave behind -- and the launch must not touch them (state,
    time, counters, rand() stream), while worlds 0 and 2 step like the oracle's; through every step kernel (config 5's, two worlds per wavefront, contacts, both stabilising ones)."""
    from moby_amd import _lib
    m, _, _, q0, qd0, dt = A.load_xml(PAIR)
    if variant in ("config5", "pack", "stab"): m.nspheres = 0
    m.cstab_max_iterations = 10 if variant in ("stab", "contacts+stab") else 0
    B = 4
    rng = np.random.default_rng(8)
    q = np.tile(q0, (B, 1)); qd = np.tile(qd0, (B, 1)); q[:, 6] = [0.0, 0.2, 0.39, -0.5]; qd[1:] += rng.uniform(-0.5, 0.5, (B - 1, 7)); q[:, 1] -= 0.08
    aux0 = S.new_aux(B); aux0["status"][[1, 3]] = S.MH_WORLD_LCP_FAILED; aux0["time"][:] = [0.0, 0.5, 0.0, 0.25]; aux0["steps"][:] = [0, 7, 0, 3]
    lib = _lib.load()
    try:
        _lib.check(lib.mh_debug_set(9, 1 if variant == "pack" else 0))
        ab = A.ArticBatch(m, q, qd, aux0)
        ab.step(dt, 25); ab.step(dt, 25)
        q_g, qd_g, aux_g = ab.download(); ab.close()
    finally:
        _lib.check(lib.mh_debug_set(9, 0))
    q_o, qd_o, aux_o = q.copy(), qd.copy(), aux0.copy()
    oracle.artic_step(m, q_o, qd_o, aux_o, dt, 50)
    assert np.array_equal(q_g, q_o) and np.array_equal(qd_g, qd_o)
    for f in ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "stab_iters"):
        assert np.array_equal(aux_g[f], aux_o[f]), f
    for w in (1, 3):
        assert np.array_equal(q_g[w], q[w]) and np.array_equal(qd_g[w], qd[w]) and aux_g["time"][w] == aux0["time"][w] and aux_g["steps"][w] == aux0["steps"][w] and aux_g["mini_steps"][w] == 0
    assert (aux_g["steps"][[0, 2]] == 50).all() and (aux_g["status"][[0, 2]] & S.MH_WORLD_LCP_FAILED == 0).all()
