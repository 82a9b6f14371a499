"""GPU parity of the implicit-joint path of the large-world stepper (include/moby_hip_stack.h: mh_big_scene.njoints ...)
against the oracle: Simulator::solve's KKT forward dynamics (src/Simulator.cpp:608-805) for every jointed island, joint
edges in the constraint islands (src/UnilateralConstraint.cpp:993-1008) -- states, counters and rand() streams bit for bit."""
import numpy as np
import pytest

from moby_amd import scene as S, stack as K
from tests.test_oracle_joints import free_scene, rest_state

pytestmark = pytest.mark.gpu

FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "stab_iters", "lcp_alg_bytes", "stab_rows",
          "zlast_size", "zbuf_size", "zbuf_cap")


def run_both(oracle, sc, st0, dt, nsteps, chunks=1):
    B = st0.shape[0]
    bb = K.BigBatch(sc, st0)
    cap = bb.cap
    st_o = st0.copy(); aux_o = S.new_aux(B); zl = np.zeros((B, cap)); zb = np.zeros((B, cap))
    for _ in range(chunks):
        bb.step(dt, nsteps)
        for w in range(B):
            oracle.big_step(sc, st_o[w], aux_o[w:w + 1], dt, nsteps, zlast=zl[w], zbuf=zb[w], cap=cap)
    st_g, aux_g = bb.download()
    bb.close()
    for f in FIELDS:
        assert np.array_equal(aux_g[f], aux_o[f]), "%s: gpu %r oracle %r" % (f, aux_g[f], aux_o[f])
    assert np.array_equal(st_g, st_o), "max |diff| = %.3e" % np.abs(st_g - st_o).max()
    return st_g, aux_g


def perturbed(st, B, seed, vel=0.3):
    rng = np.random.default_rng(seed)
    out = np.repeat(st.reshape(1, -1, 13), B, axis=0).copy()
    out[1:, :, 7:13] += vel * rng.standard_normal((B - 1, st.shape[0], 6))
    return out.reshape(B, -1)


def test_revolute_pendulums_match_oracle(oracle):
    th0 = 0.4
    st = rest_state([[np.sin(th0), -np.cos(th0), 0.0]])
    st[0, 3:7] = (0.0, 0.0, np.sin(th0 / 2), np.cos(th0 / 2))
    sc = free_scene(1, [K.make_joint(K.MH_IJOINT_REVOLUTE, 1, 0, (0.0, 0.0, 0.0), st, 1, axis=(0, 0, 1))])
    st_g, aux = run_both(oracle, sc, perturbed(st, 6, 1), 1e-3, 100, chunks=2)
    assert (aux["status"] == 0).all() and (aux["lcp_solves"] == 0).all()
    assert np.abs(st_g[0].reshape(1, 13)[0, [2, 9, 10, 11]]).max() < 1e-12


def test_chains_of_spherical_joints_and_a_weld_match_oracle(oracle):
    """World - s0 - s1 - s2 (spherical joints) with s3 welded to s2, plus a free body: one jointed island of four bodies
    (3 + 3 + 3 + 6 equations), one island without joints that keeps the free-body forward dynamics."""
    nb = 5
    st = rest_state([[0.5, 0.0, 0.0], [1.5, 0.0, 0.0], [2.5, 0.0, 0.0], [2.5, 0.6, 0.0], [0.0, 3.0, 0.0]])
    joints = [K.make_joint(K.MH_IJOINT_SPHERICAL, nb, 0, (0.0, 0.0, 0.0), st, nb),
              K.make_joint(K.MH_IJOINT_SPHERICAL, 0, 1, (1.0, 0.0, 0.0), st, nb),
              K.make_joint(K.MH_IJOINT_SPHERICAL, 2, 1, (2.0, 0.0, 0.0), st, nb),      # inboard / outboard swapped on purpose
              K.make_joint(K.MH_IJOINT_FIXED, 2, 3, (2.5, 0.3, 0.0), st, nb)]
    sc = free_scene(nb, joints, mass=[1.0, 2.0, 0.5, 1.5, 1.0])
    st_g, aux = run_both(oracle, sc, perturbed(st, 5, 2), 1e-3, 60, chunks=2)
    assert (aux["status"] == 0).all()
    b = st_g.reshape(5, nb, 13)
    assert abs(b[0, 4, 8] + 9.81 * 0.12) < 1e-12                    # the free body just falls: v = g t


def test_redundant_and_closed_loop_joints_match_oracle(oracle):
    """A closed loop (world - a - b - world) and a doubled joint: J iM J' is rank deficient, the greedy Cholesky of
    Simulator::solve (Sim:728-755) drops the dependent rows -- identically on both sides."""
    nb = 2
    st = rest_state([[0.5, 0.0, 0.0], [1.5, 0.0, 0.0]])
    j1 = K.make_joint(K.MH_IJOINT_REVOLUTE, nb, 0, (0.0, 0.0, 0.0), st, nb, axis=(0, 0, 1))
    j2 = K.make_joint(K.MH_IJOINT_REVOLUTE, 0, 1, (1.0, 0.0, 0.0), st, nb, axis=(0, 0, 1))
    j3 = K.make_joint(K.MH_IJOINT_REVOLUTE, 1, nb, (2.0, 0.0, 0.0), st, nb, axis=(0, 0, 1))
    sc = free_scene(nb, [j1, j2, j3, j2])
    st_g, aux = run_both(oracle, sc, perturbed(st, 4, 3, vel=0.1), 1e-3, 40)
    assert (aux["status"] == 0).all()


def test_joints_merge_contact_islands_and_step_with_impacts(oracle):
    """Spheres dropped on the plane, two of them tied by a spherical joint, a third welded on top of the second: the joint
    edges merge constraint islands (one impact LCP for the tied spheres), the impact handler ignores the joint rows (as the
    reference's does), forward dynamics of the jointed island is the KKT solve.  (The run ends before anything comes to
    rest: with stabilisation off a resting contact sinks, conservative advancement then returns 0 and Simulator::solve
    divides by that h -- in the reference as in the oracle.)"""
    r = 0.2
    nb = 4
    h0 = 0.03
    st = rest_state([[0.0, r + h0, 0.0], [1.0, r + h0, 0.0], [1.0, r + h0 + 0.5, 0.0], [3.0, r + 0.3, 0.0]])
    joints = [K.make_joint(K.MH_IJOINT_SPHERICAL, 0, 1, (0.5, r + h0, 0.0), st, nb),
              K.make_joint(K.MH_IJOINT_FIXED, 1, 2, (1.0, r + h0 + 0.25, 0.0), st, nb)]
    sc = K.BigScene([S.MH_GEOM_SPHERE] * nb, [(r, 0, 0)] * nb, [1.0, 1.0, 0.5, 1.0], [[0.016] * 3] * 2 + [[0.008] * 3] + [[0.016] * 3],
                    [(k, nb, 0) for k in range(nb)], gravity=(0.0, -9.81, 0.0), cstab_max_iterations=0, joints=joints, lcp_n_max=64,
                    mu_coulomb=0.3, epsilon=0.8)
    st_g, aux = run_both(oracle, sc, perturbed(st, 6, 4, vel=0.05), 1e-3, 60, chunks=2)
    assert np.isfinite(st_g).all()
    assert ((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) == 0).all() and (aux["lcp_solves"] > 0).all()
    assert (aux["mini_steps"] > aux["steps"]).any()                # conservative-advancement sub-steps before the impact


def test_scene_checks_for_joints():
    st = rest_state([[0.5, 0.0, 0.0]])
    j = K.make_joint(K.MH_IJOINT_SPHERICAL, 1, 0, (0.0, 0.0, 0.0), st, 1)
    bad = dict(j); bad["type"] = 7
    sc = K.BigScene([S.MH_GEOM_SPHERE], [(0.2, 0, 0)], [1.0], [[0.016] * 3], [], gravity=(0, -9.81, 0), joints=[bad], cstab_max_iterations=0)
    with pytest.raises(Exception, match="MH_IJOINT"):
        K.BigBatch(sc, st.reshape(1, -1))
    # an island beyond the built sizes: 10 welds = 60 equations > MH_IJOINT_MAX_EQNS
    nb = 11
    stn = rest_state([[float(k), 0.0, 0.0] for k in range(nb)])
    welds = [K.make_joint(K.MH_IJOINT_FIXED, k, k + 1, (k + 0.5, 0.0, 0.0), stn, nb) for k in range(nb - 1)]
    with pytest.raises(Exception, match="jointed island"):
        K.BigBatch(free_scene(nb, welds), stn.reshape(1, -1))


# ---- ConstraintStabilization with implicit joints ---------------------------------------------------------------
def stab_scene(nb, joints, iters, pairs=(), **kw):
    J = np.array([[0.4 * 0.2 * 0.2] * 3] * nb)
    return K.BigScene([S.MH_GEOM_SPHERE] * nb, [(0.2, 0, 0)] * nb, np.ones(nb), J, list(pairs), gravity=(0.0, -9.81, 0.0),
                      cstab_max_iterations=iters, joints=joints, lcp_n_max=64, **kw)


def test_stabilised_chains_step_like_the_oracle(oracle):
    """Full steps with the stabiliser on: every step ends with the bilateral Newton iteration of the jointed island
    (greedy full-rank rows of J J' - sqrt(eps) I, (J iM J') lambda = C, Ridders / backtracking on |C|) -- next to a free
    sphere bouncing on the plane whose island keeps the unilateral path."""
    nb = 4
    st = rest_state([[0.5, 2.0, 0.0], [1.5, 2.0, 0.0], [2.5, 2.0, 0.0], [5.0, 0.2 + 0.01, 0.0]])
    joints = [K.make_joint(K.MH_IJOINT_SPHERICAL, nb, 0, (0.0, 2.0, 0.0), st, nb),
              K.make_joint(K.MH_IJOINT_REVOLUTE, 0, 1, (1.0, 2.0, 0.0), st, nb, axis=(0, 0, 1)),
              K.make_joint(K.MH_IJOINT_FIXED, 1, 2, (2.0, 2.0, 0.0), st, nb)]
    sc = stab_scene(nb, joints, 20, pairs=[(3, nb, 0)], epsilon=0.5)
    st_g, aux = run_both(oracle, sc, perturbed(st, 5, 7, vel=0.2), 1e-3, 50, chunks=2)
    assert (aux["status"] == 0).all() and (aux["stab_iters"] > 0).all() and (aux["lcp_solves"] > 0).all()


def test_standalone_stabilize_closes_open_joints_like_the_oracle(oracle):
    """mh_big_batch_stabilize (seam B3) on joints pulled apart by up to 2e-2: same iterations, same poses."""
    nb = 2
    st = rest_state([[1.0, 0.0, 0.0], [2.2, 0.0, 0.0]])
    joints = [K.make_joint(K.MH_IJOINT_REVOLUTE, nb, 0, (0.0, 0.0, 0.0), st, nb, axis=(0, 0, 1)),
              K.make_joint(K.MH_IJOINT_REVOLUTE, 0, 1, (1.6, 0.0, 0.0), st, nb, axis=(0, 1, 0))]
    sc = stab_scene(nb, joints, 30)
    B = 6
    rng = np.random.default_rng(9)
    s = np.repeat(st.reshape(1, nb, 13), B, axis=0).copy()
    s[:, :, 0:3] += 0.02 * rng.uniform(-1, 1, (B, nb, 3))
    q = s[:, :, 3:7] + 0.02 * rng.uniform(-1, 1, (B, nb, 4)); s[:, :, 3:7] = q / np.linalg.norm(q, axis=2)[:, :, None]
    s[:, :, 7:13] = 0.3
    s = s.reshape(B, -1)
    bb = K.BigBatch(sc, s)
    bb.stabilize()
    st_g, aux_g = bb.download()
    bb.close()
    st_o = s.copy(); aux_o = S.new_aux(B)
    for w in range(B):
        oracle.big_step(sc, st_o[w], aux_o[w:w + 1], 1e-3, 1, mode=1)
    for f in FIELDS:
        assert np.array_equal(aux_g[f], aux_o[f]), f
    assert np.array_equal(st_g, st_o), "max |diff| = %.3e" % np.abs(st_g - st_o).max()
    assert (aux_g["status"] == 0).all() and (aux_g["stab_iters"] >= 1).all()
    # the line search stops at the first sign change of any row (CStab:1132-1145), so 30 iterations do not always reach 1e-6
    # from 2e-2: every world must have come most of the way, and the unperturbed-most ones all the way
    err = lambda states: np.array([max(np.abs(oracle.joint_eval(sc, states[w], j)[0]).max() for j in range(2)) for w in range(B)])
    e0, e1 = err(s), err(st_g)
    assert (e1 < 0.1 * e0).all() and e1.min() < 1e-6
    assert np.array_equal(st_g.reshape(B, nb, 13)[:, :, 7:13], s.reshape(B, nb, 13)[:, :, 7:13])     # velocities restored


def test_jointed_bodies_in_contact_stabilise_like_the_oracle(oracle):
    """The stabiliser's contact islands hold implicit joints: compute_X's general case (X = iM - 2G + G'MG, ICH:1590-1695) on
    the device -- a welded pair and a pair tied by a spherical joint sunk into the plane by different amounts per world,
    next to a free sphere; standalone stabilize() and full steps."""
    r = 0.2
    nb = 5
    st0 = rest_state([[0.0, r, 0.0], [0.0, r + 0.5, 0.0], [2.0, r, 0.0], [3.0, r + 0.3, 0.0], [6.0, r, 0.0]])
    joints = [K.make_joint(K.MH_IJOINT_FIXED, 0, 1, (0.0, r + 0.25, 0.0), st0, nb),
              K.make_joint(K.MH_IJOINT_SPHERICAL, 2, 3, (2.5, r + 0.15, 0.0), st0, nb)]
    sc = stab_scene(nb, joints, 20, pairs=[(k, nb, 0) for k in range(nb)])
    B = 5
    rng = np.random.default_rng(12)
    s = np.repeat(st0.reshape(1, nb, 13), B, axis=0).copy()
    sink = rng.uniform(2e-5, 2e-4, (B, 3))
    s[:, 0, 1] -= sink[:, 0]; s[:, 1, 1] -= sink[:, 0]; s[:, 2, 1] -= sink[:, 1]; s[:, 4, 1] -= sink[:, 2]
    s[1:, :, 0] += 1e-3 * rng.uniform(-1, 1, (B - 1, nb))
    s = s.reshape(B, -1)
    bb = K.BigBatch(sc, s)
    bb.stabilize()
    st_g, aux_g = bb.download()
    bb.close()
    st_o = s.copy(); aux_o = S.new_aux(B)
    for w in range(B):
        oracle.big_step(sc, st_o[w], aux_o[w:w + 1], 1e-3, 1, mode=1)
    for f in FIELDS:
        assert np.array_equal(aux_g[f], aux_o[f]), "%s: gpu %r oracle %r" % (f, aux_g[f], aux_o[f])
    assert np.array_equal(st_g, st_o), "max |diff| = %.3e" % np.abs(st_g - st_o).max()
    # worlds whose joints were also pulled sideways may end in the reference's "failed to effectively finish" state (update_q's
    # backtracking gives up, CStab:1196-1198: MH_WORLD_STAB_FAILED) -- identically on both sides; the others are stabilised
    good = aux_g["status"] == 0
    assert good[0] and (aux_g["status"][~good] == S.MH_WORLD_STAB_FAILED).all() and (aux_g["lcp_solves"] >= 2).all()
    o = st_g.reshape(B, nb, 13)[good]
    assert (o[:, [0, 2, 4], 1] >= r - 1e-9).all() and np.abs((o[:, 1, 1] - o[:, 0, 1]) - 0.5).max() < 1e-6
    # and as full steps (gravity pulls everything back into the plane every step)
    run_both(oracle, sc, s, 1e-3, 15)


def test_planar_joint_example_matches_oracle(oracle):
    """The scene after example/planar-joint/constrained.xml (tests/test_oracle_joints.py::planar_box_scene): a joint to the static
    world with a position row along a direction -- KKT forward dynamics with a single body (the first step removes the forbidden
    spin after positions were integrated with it: the stabiliser then turns the box back), bilateral stabilisation."""
    from tests.test_oracle_joints import planar_box_scene
    sc, st = planar_box_scene()
    B = 4
    s0 = np.repeat(st.reshape(1, -1), B, axis=0).copy()
    rng = np.random.default_rng(21)
    s0.reshape(B, 1, 13)[1:, 0, 10:13] += rng.uniform(-2, 2, (B - 1, 3))
    s0.reshape(B, 1, 13)[1:, 0, 7:10] += rng.uniform(-0.2, 0.2, (B - 1, 3))
    st_g, aux = run_both(oracle, sc, s0, 1e-3, 40, chunks=2)
    assert np.isfinite(st_g).all() and (aux["status"] == 0).all() and (aux["stab_iters"] > 0).all()
    assert np.abs(st_g.reshape(B, 13)[:, [10, 12]]).max() < 1e-5 and np.abs(st_g.reshape(B, 13)[:, 1] - 0.5).max() < 1e-6


def test_thousand_jointed_chains_properties():
    """1024 worlds of a four-link chain hung from the world (revolute, spherical, fixed, spherical), random initial spins, 100 steps
    with the stabiliser on: no world fails, every joint stays closed to bilateral_eps, identical worlds give identical results
    wherever they sit in the batch, and the chains really swing."""
    nb, B = 4, 1024
    st = rest_state([[0.5 + k, 0.0, 0.0] for k in range(nb)])
    joints = [K.make_joint(K.MH_IJOINT_REVOLUTE, nb, 0, (0.0, 0.0, 0.0), st, nb, axis=(0, 0, 1)),
              K.make_joint(K.MH_IJOINT_SPHERICAL, 0, 1, (1.0, 0.0, 0.0), st, nb),
              K.make_joint(K.MH_IJOINT_FIXED, 1, 2, (2.0, 0.0, 0.0), st, nb),
              K.make_joint(K.MH_IJOINT_SPHERICAL, 2, 3, (3.0, 0.0, 0.0), st, nb)]
    sc = stab_scene(nb, joints, 20)
    rng = np.random.default_rng(33)
    s0 = np.repeat(st.reshape(1, nb, 13), B, axis=0).copy()
    w = rng.uniform(-1, 1, (B // 2, 3))                           # a rigid rotation about the pivot: consistent with every joint
    for k in range(nb):
        s0[:B // 2, k, 10:13] = w * np.array([0.0, 0.0, 1.0]); s0[:B // 2, k, 7:10] = np.cross(s0[:B // 2, k, 10:13], s0[:B // 2, k, 0:3])
    s0[B // 2:] = s0[:B // 2]                                     # second half = copy of the first
    s0 = s0.reshape(B, -1)
    bb = K.BigBatch(sc, s0)
    bb.step(1e-3, 100)
    st_g, aux = bb.download()
    bb.close()
    assert (aux["status"] == 0).all() and (aux["steps"] == 100).all() and (aux["stab_iters"] > 0).all()
    assert np.array_equal(st_g[B // 2:], st_g[:B // 2]) and np.array_equal(aux["stab_iters"][B // 2:], aux["stab_iters"][:B // 2])
    b = st_g.reshape(B, nb, 13)
    # joint closure, computed from the poses: anchors coincide
    def R(q):
        x, y, z, ww = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)], -1),
                         np.stack([2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)], -1),
                         np.stack([2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)], -1)], -2)
    worst = 0.0
    for j in joints:
        def pt(body, anchor):
            if body >= nb:
                return np.broadcast_to(np.asarray(anchor), (B, 3))
            return b[:, body, 0:3] + np.einsum("bij,j->bi", R(b[:, body, 3:7]), np.asarray(anchor))
        worst = max(worst, np.abs(pt(j["inboard"], j["anchor_in"]) - pt(j["outboard"], j["anchor_out"])).max())
    assert worst < 1e-6
    assert b[:, 3, 1].min() < -0.05 and np.abs(b[:, :, 2]).max() < 1e-6        # they fell, in the plane of the hinge


def test_cpp_implicit_joints_adapter_example():
    """moby_amd/cpp/MobyHipStackSimulator.h::ImplicitJoints (what a binding fills from Moby::Joint objects) through its example:
    a revolute + universal pendulum and a box on a planar joint, 200 stabilised steps."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cpp = os.path.join(root, "moby_amd", "cpp")
    exe = os.path.join(cpp, "example_joints")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", os.path.join(cpp, "example_joints.cpp"), "-L" + os.path.join(root, "moby_amd"),
                           "-lmoby_hip", "-Wl,-rpath," + os.path.join(root, "moby_amd"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, timeout=120)
    assert p.returncode == 0 and b"status=0/0" in p.stdout, p.stdout
