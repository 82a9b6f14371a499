"""CPU tests of the oracle's implicit-joint path: Simulator::solve (the KKT forward dynamics of jointed islands,
src/Simulator.cpp:608-805), Simulator::find_islands (:956-1045) and the joint edges of
UnilateralConstraint::determine_connected_constraints (src/UnilateralConstraint.cpp:993-1008).  The joints' own functions
(evaluate_constraints / calc_constraint_jacobian) are Ravelin's and not in the reference tree: they are pinned here by
consistency (the Jacobian is the derivative of the constraint function) and by physics."""
import numpy as np
import pytest

from moby_amd import scene as S, stack as K


def free_scene(nb, joints, gravity=(0.0, -9.81, 0.0), mass=None, radius=0.2, pairs=()):
    """nb spheres (collision pairs only if listed), stabilisation off as the joint path requires."""
    mass = np.ones(nb) if mass is None else np.asarray(mass, dtype=np.float64)
    J = np.array([[0.4 * m * radius * radius] * 3 for m in mass])
    return K.BigScene([S.MH_GEOM_SPHERE] * nb, [(radius, 0, 0)] * nb, mass, J, list(pairs), gravity=gravity,
                      cstab_max_iterations=0, joints=joints, lcp_n_max=64)


def rest_state(positions):
    st = np.zeros((len(positions), 13)); st[:, 6] = 1.0
    st[:, 0:3] = positions
    return st


def advance(st, v, eps):
    """poses moved by eps along the generalized velocity v (nb x 6: linear, angular)"""
    out = st.copy()
    for b in range(st.shape[0]):
        out[b, 0:3] += eps * v[b, 0:3]
        x, y, z, w = st[b, 3:7]; wx, wy, wz = v[b, 3:6]
        qd = 0.5 * np.array([w * wx + wy * z - wz * y, w * wy + wz * x - wx * z, w * wz + wx * y - wy * x, -wx * x - wy * y - wz * z])
        q = st[b, 3:7] + eps * qd
        out[b, 3:7] = q / np.linalg.norm(q)
    return out


@pytest.mark.parametrize("kind", [K.MH_IJOINT_SPHERICAL, K.MH_IJOINT_REVOLUTE, K.MH_IJOINT_FIXED, K.MH_IJOINT_PLANAR, K.MH_IJOINT_UNIVERSAL, K.MH_IJOINT_PRISMATIC])
def test_jacobian_is_the_derivative_of_the_constraint_function(oracle, kind):
    rng = np.random.default_rng(3 + kind)
    nb = 2
    st0 = rest_state(rng.standard_normal((nb, 3)))
    q = rng.standard_normal((nb, 4)); st0[:, 3:7] = q / np.linalg.norm(q, axis=1)[:, None]
    loc = rng.standard_normal(3); axis = rng.standard_normal(3)
    for inboard, outboard in ((0, 1), (nb, 1), (0, nb)):
        sc = free_scene(nb, [K.make_joint(kind, inboard, outboard, loc, st0, nb, axis=axis)])
        C0, _, _ = oracle.joint_eval(sc, st0.reshape(-1), 0)
        assert np.abs(C0).max() < 1e-14                          # satisfied at the reference poses
        # move the bodies somewhere else: the constraint is violated, the Jacobian is still its derivative
        st = advance(st0, 0.3 * rng.standard_normal((nb, 6)), 1.0)
        C, A, B = oracle.joint_eval(sc, st.reshape(-1), 0)
        v = rng.standard_normal((nb, 6))
        eps = 1e-6
        Cp, _, _ = oracle.joint_eval(sc, advance(st, v, eps).reshape(-1), 0)
        Cm, _, _ = oracle.joint_eval(sc, advance(st, v, -eps).reshape(-1), 0)
        Jv = np.zeros(6)
        if inboard < nb:
            Jv += A @ v[inboard]
        if outboard < nb:
            Jv += B @ v[outboard]
        rows = K.IJOINT_ROWS[kind]
        np.testing.assert_allclose(((Cp - Cm) / (2 * eps))[:rows], Jv[:rows], atol=2e-8)
        assert rows == 6 or np.abs(Jv[rows:]).max() == 0.0


def run(oracle, sc, st, dt, nsteps):
    aux = S.new_aux(1)
    s = st.reshape(-1).copy()
    oracle.big_step(sc, s, aux, dt, nsteps)
    return s.reshape(-1, 13), aux


def test_revolute_pendulum_period_and_drift(oracle):
    """A sphere on a massless arm of length L hinged to the world about z: period of the physical pendulum
    2 pi sqrt((I + m L^2) / (m g L)) for small swings; the hinge point drifts only at the integrator's rate
    (no stabilisation, as in the reference with constraint-stabilization-max-iterations = 0)."""
    L, r, m = 1.0, 0.2, 1.0
    th0 = 0.1
    st = rest_state([[L * np.sin(th0), -L * np.cos(th0), 0.0]])
    st[0, 3:7] = (0.0, 0.0, np.sin(th0 / 2), np.cos(th0 / 2))
    sc = free_scene(1, [K.make_joint(K.MH_IJOINT_REVOLUTE, 1, 0, (0.0, 0.0, 0.0), st, 1, axis=(0, 0, 1))])
    I = 0.4 * m * r * r
    T = 2 * np.pi * np.sqrt((I + m * L * L) / (m * 9.81 * L)) * (1 + th0 * th0 / 16)
    dt = 1e-3
    s, xs = st, []
    nsteps = int(1.25 * T / dt)
    for k in range(nsteps):
        s, aux = run(oracle, sc, s, dt, 1)
        xs.append(s[0, 0])
    assert aux["status"][0] == 0
    xs = np.array(xs)
    down = [k for k in range(1, len(xs)) if xs[k - 1] > 0 >= xs[k]]      # first crossing of the vertical: T / 4
    up = [k for k in range(1, len(xs)) if xs[k - 1] < 0 <= xs[k]]        # second: 3 T / 4
    assert abs(down[0] * dt - T / 4) < 0.01 * T and abs(up[0] * dt - 3 * T / 4) < 0.01 * T
    C, _, _ = oracle.joint_eval(sc, s.reshape(-1), 0)
    assert np.abs(C).max() < 2e-3                                # drift after 2 500 steps, uncorrected
    assert abs(np.linalg.norm(s[0, 0:3]) - L) < 2e-3
    assert np.abs(s[0, [2, 9, 10, 11]]).max() < 1e-12           # planar motion: the two orientation rows hold


def test_fixed_joint_makes_one_rigid_body(oracle):
    """Two spheres welded together, spinning in free fall: the pair falls like one body (v_com = g t), its relative pose
    stays (to the integrator's drift), angular momentum about the centre of mass is conserved."""
    m = np.array([1.0, 2.0])
    st = rest_state([[0.0, 0.0, 0.0], [0.6, 0.0, 0.0]])
    sc = free_scene(2, [K.make_joint(K.MH_IJOINT_FIXED, 0, 1, (0.3, 0.0, 0.0), st, 2)], mass=m)
    w = np.array([0.0, 0.3, 2.0])
    com = (m[:, None] * st[:, 0:3]).sum(axis=0) / m.sum()
    for b in range(2):
        st[b, 10:13] = w; st[b, 7:10] = np.cross(w, st[b, 0:3] - com)

    def ang_mom(s):
        c = (m[:, None] * s[:, 0:3]).sum(axis=0) / m.sum(); vc = (m[:, None] * s[:, 7:10]).sum(axis=0) / m.sum()
        Ib = 0.4 * m * 0.2 * 0.2
        return sum(m[b] * np.cross(s[b, 0:3] - c, s[b, 7:10] - vc) + Ib[b] * s[b, 10:13] for b in range(2))
    L0 = ang_mom(st)
    s, aux = run(oracle, sc, st, 1e-3, 500)
    assert aux["status"][0] == 0
    vc = (m[:, None] * s[:, 7:10]).sum(axis=0) / m.sum()
    np.testing.assert_allclose(vc, [0.0, -9.81 * 0.5, 0.0], atol=1e-9)
    np.testing.assert_allclose(ang_mom(s), L0, rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(s[0, 10:13], s[1, 10:13], atol=2e-3)             # one angular velocity
    assert abs(np.linalg.norm(s[1, 0:3] - s[0, 0:3]) - 0.6) < 2e-3
    C, _, _ = oracle.joint_eval(sc, s.reshape(-1), 0)
    assert np.abs(C).max() < 3e-3


def test_spherical_chain_hangs_from_the_world_and_keeps_its_energy(oracle):
    """Three spheres in a chain of spherical joints, the first hung from the world, released from the horizontal: the
    semi-implicit integrator keeps the energy to a few per cent over one second; every joint point stays put."""
    nb = 3
    st = rest_state([[0.5 + k, 0.0, 0.0] for k in range(nb)])
    joints = [K.make_joint(K.MH_IJOINT_SPHERICAL, nb, 0, (0.0, 0.0, 0.0), st, nb)]
    joints += [K.make_joint(K.MH_IJOINT_SPHERICAL, k, k + 1, (1.0 + k, 0.0, 0.0), st, nb) for k in range(nb - 1)]
    sc = free_scene(nb, joints)

    def energy(s):
        I = 0.4 * 0.2 * 0.2
        return sum(0.5 * (s[b, 7:10] @ s[b, 7:10]) + 0.5 * I * (s[b, 10:13] @ s[b, 10:13]) + 9.81 * s[b, 1] for b in range(nb))
    E0 = energy(st)
    s, aux = run(oracle, sc, st, 1e-3, 1000)
    assert aux["status"][0] == 0 and aux["lcp_solves"][0] == 0
    assert abs(energy(s) - E0) < 0.05 * 9.81 * 4.5
    assert s[:, 1].min() < -0.5                                    # it really fell
    for j in range(nb):
        C, _, _ = oracle.joint_eval(sc, s.reshape(-1), j)
        assert np.abs(C).max() < 2e-2


def test_redundant_joints_are_dropped_by_the_greedy_cholesky(oracle):
    """The same revolute joint listed twice: J iM J' is singular; Simulator::solve keeps the largest leading full-rank set
    (Sim:728-755) and the motion equals the single-joint scene's."""
    th0 = 0.2
    st = rest_state([[np.sin(th0), -np.cos(th0), 0.0]])
    st[0, 3:7] = (0.0, 0.0, np.sin(th0 / 2), np.cos(th0 / 2))
    j = K.make_joint(K.MH_IJOINT_REVOLUTE, 1, 0, (0.0, 0.0, 0.0), st, 1, axis=(0, 0, 1))
    s1, a1 = run(oracle, free_scene(1, [j]), st, 1e-3, 200)
    s2, a2 = run(oracle, free_scene(1, [j, j]), st, 1e-3, 200)
    assert a2["status"][0] == 0
    np.testing.assert_allclose(s2, s1, atol=1e-9)


def test_a_joint_merges_two_contact_islands(oracle):
    """Two spheres resting on the plane: two islands, two impact LCPs per step.  A spherical joint between them makes one
    island (UC:993-1008) -- one LCP of twice the size -- while the impact handler itself ignores the joint's rows (the
    reference never fills island_ijoints there: ImpactConstraintHandler.cpp:1965-2008 sees an empty list)."""
    r = 0.2
    st = rest_state([[0.0, r, 0.0], [1.0, r, 0.0]])
    mk = lambda joints: K.BigScene([S.MH_GEOM_SPHERE] * 2, [(r, 0, 0)] * 2, [1.0, 1.0], [[0.016] * 3] * 2, [(0, 2, 0), (1, 2, 0)],
                                   gravity=(0.0, -9.81, 0.0), cstab_max_iterations=0, joints=joints, lcp_n_max=64, mu_coulomb=0.3)
    s_free, a_free = run(oracle, mk([]), st, 1e-3, 5)
    s_jnt, a_jnt = run(oracle, mk([K.make_joint(K.MH_IJOINT_SPHERICAL, 0, 1, (0.5, r, 0.0), st, 2)]), st, 1e-3, 5)
    assert a_free["status"][0] == 0 and a_jnt["status"][0] == 0
    assert a_free["lcp_solves"][0] == 2 * a_jnt["lcp_solves"][0] and a_free["lcp_rows"][0] == a_jnt["lcp_rows"][0]
    np.testing.assert_allclose(s_jnt[:, 0:3], s_free[:, 0:3], atol=1e-6)         # both just rest


# ---- ConstraintStabilization with implicit joints (bilateral-only islands) ------------------------------------------
def test_stabilisation_keeps_the_joints_closed(oracle):
    """The spherical chain again, with the stabiliser on: after every step the bilateral violation is back under
    bilateral_eps = 1e-6 (CStab:62, 197) where the unstabilised run has drifted to 1e-2; velocities are untouched by it."""
    nb = 3
    st = rest_state([[0.5 + k, 0.0, 0.0] for k in range(nb)])
    joints = [K.make_joint(K.MH_IJOINT_SPHERICAL, nb, 0, (0.0, 0.0, 0.0), st, nb)]
    joints += [K.make_joint(K.MH_IJOINT_SPHERICAL, k, k + 1, (1.0 + k, 0.0, 0.0), st, nb) for k in range(nb - 1)]
    J = np.array([[0.4 * 0.2 * 0.2] * 3] * nb)
    mk = lambda it: K.BigScene([S.MH_GEOM_SPHERE] * nb, [(0.2, 0, 0)] * nb, np.ones(nb), J, [], gravity=(0.0, -9.81, 0.0),
                               cstab_max_iterations=it, joints=joints, lcp_n_max=64)
    s_on, a_on = run(oracle, mk(20), st, 1e-3, 600)
    s_off, a_off = run(oracle, mk(0), st, 1e-3, 600)
    assert a_on["status"][0] == 0 and a_on["stab_iters"][0] > 0 and a_on["lcp_solves"][0] == 0
    worst_on = max(np.abs(oracle.joint_eval(mk(20), s_on.reshape(-1), j)[0]).max() for j in range(nb))
    worst_off = max(np.abs(oracle.joint_eval(mk(0), s_off.reshape(-1), j)[0]).max() for j in range(nb))
    assert worst_on < 1e-6 < 1e-3 < worst_off
    assert s_on[:, 1].min() < -0.3


def test_stabilisation_closes_an_open_joint_in_a_few_iterations(oracle):
    """A revolute joint pulled apart by 1e-2 and twisted: stabilize() alone (seam B3) brings |C| under 1e-6 -- the Newton
    step (J iM J') lambda = C, dq = -iM J' lambda on the largest independent row set, with the Ridders / backtracking line
    search on the constraint norm."""
    st = rest_state([[1.0, 0.0, 0.0], [2.2, 0.0, 0.0]])
    joints = [K.make_joint(K.MH_IJOINT_REVOLUTE, 2, 0, (0.0, 0.0, 0.0), st, 2, axis=(0, 0, 1)),
              K.make_joint(K.MH_IJOINT_REVOLUTE, 0, 1, (1.6, 0.0, 0.0), st, 2, axis=(0, 1, 0))]
    sc = free_scene(2, joints)
    sc.c.cstab_max_iterations = 30
    s = st.copy()
    s[0, 0:3] += (0.01, -0.008, 0.004); s[1, 0:3] += (-0.006, 0.01, 0.0)
    q = np.array([0.01, -0.02, 0.015, 1.0]); s[1, 3:7] = q / np.linalg.norm(q)
    s[:, 7:13] = 0.3                                              # velocities are saved and restored (CStab:181, 246)
    before = max(np.abs(oracle.joint_eval(sc, s.reshape(-1), j)[0]).max() for j in range(2))
    aux = S.new_aux(1)
    out = s.reshape(-1).copy()
    oracle.big_step(sc, out, aux, 1e-3, 1, mode=1)
    after = max(np.abs(oracle.joint_eval(sc, out, j)[0]).max() for j in range(2))
    assert before > 5e-3 and after < 1e-6 and aux["status"][0] == 0 and 1 <= aux["stab_iters"][0] <= 10
    np.testing.assert_array_equal(out.reshape(2, 13)[:, 7:13], s[:, 7:13])


def test_jointed_bodies_in_contact_use_the_general_compute_X(oracle):
    """A sphere sunk 1e-4 into the plane with a second sphere welded on top of it, next to a pair tied by a spherical joint
    of which one has sunk: the stabiliser's contact islands hold implicit joints, X = iM - 2G + G'MG (ICH:1590-1695)
    carries the contact push through the joints -- the welded sphere rises with its partner, the joints stay closed."""
    r = 0.2
    nb = 4
    st0 = rest_state([[0.0, r, 0.0], [0.0, r + 0.5, 0.0], [2.0, r, 0.0], [3.0, r + 0.3, 0.0]])
    joints = [K.make_joint(K.MH_IJOINT_FIXED, 0, 1, (0.0, r + 0.25, 0.0), st0, nb),
              K.make_joint(K.MH_IJOINT_SPHERICAL, 2, 3, (2.5, r + 0.15, 0.0), st0, nb)]
    sc = K.BigScene([S.MH_GEOM_SPHERE] * nb, [(r, 0, 0)] * nb, np.ones(nb), [[0.016] * 3] * nb, [(k, nb, 0) for k in range(nb)],
                    gravity=(0.0, -9.81, 0.0), cstab_max_iterations=20, joints=joints, lcp_n_max=64)
    st = st0.copy()
    st[[0, 1], 1] -= 1e-4                                           # the welded pair sinks as one
    st[2, 1] -= 1e-4                                                # the tied sphere sinks alone: its joint opens by 1e-4 as well
    aux = S.new_aux(1)
    out = st.reshape(-1).copy()
    oracle.big_step(sc, out, aux, 1e-3, 1, mode=1)
    o = out.reshape(nb, 13)
    assert aux["status"][0] == 0 and aux["stab_iters"][0] >= 1 and aux["lcp_solves"][0] >= 1
    assert (o[[0, 2], 1] >= r - 1e-9).all()                         # out of the plane
    assert abs((o[1, 1] - o[0, 1]) - 0.5) < 1e-6                    # the weld held: the upper sphere rose with the lower
    for j in range(2):
        assert np.abs(oracle.joint_eval(sc, out, j)[0]).max() < 1e-6
    assert abs(o[1, 1] - (r + 0.5)) < 2e-5 and np.abs(o[:, [0, 2]] - st0[:, [0, 2]]).max() < 1e-4


def planar_box_scene(iters=10):
    """After example/planar-joint/constrained.xml: a unit box (density 1) on the plane y = 0, tied to the ground by a planar
    joint with normal +y -- it may slide and turn about y only --, spun about x and y, gravity (1, -9.81, 1).  The joint alone
    carries the box here: the file also lists the box-plane contact pair, but with the joint's plane ON the contact plane the
    stabiliser's normal row has no mobility left (Cn X Cn' = 0 with the general X) and its LCP is unsolvable, and the box
    starts exactly on the plane, where conservative advancement returns h = 0 (deviation 9)."""
    st = rest_state([[0.0, 0.5, 0.0]])
    j = K.make_joint(K.MH_IJOINT_PLANAR, 1, 0, (0.0, 0.0, 0.0), st, 1, axis=(0.0, 1.0, 0.0))
    sc = K.BigScene([S.MH_GEOM_BOX], [(1.0, 1.0, 1.0)], [1.0], [[1.0 / 6.0] * 3], [], gravity=(1.0, -9.81, 1.0),
                    cstab_max_iterations=iters, joints=[j], lcp_n_max=64)
    st[0, 10:13] = (10.0, 2.0, 0.0)                               # spin about x (the joint forbids it) and about y (allowed)
    return sc, st


def test_planar_joint_example_slides_and_turns_about_the_normal_only(oracle):
    sc, st = planar_box_scene()
    s, aux = run(oracle, sc, st, 1e-3, 200)
    assert aux["status"][0] == 0 and aux["stab_iters"][0] > 0
    assert abs(s[0, 1] - 0.5) < 1e-6                              # stays on the plane (gravity's normal part is carried by the joint)
    assert abs(s[0, 10]) < 1e-6 and abs(s[0, 12]) < 1e-6           # the forbidden spin is gone after the first step ...
    assert abs(s[0, 11] - 2.0) < 1e-3                              # ... the spin about the normal is untouched
    np.testing.assert_allclose(s[0, [7, 9]], [0.2, 0.2], atol=1e-3)     # g_x = g_z = 1 for 0.2 s
    C, _, _ = oracle.joint_eval(sc, s.reshape(-1), 0)
    assert np.abs(C).max() < 1e-6
