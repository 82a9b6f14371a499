"""CPU: the articulated-body oracle (oracle/artic.hpp).  Ravelin's CRB forward dynamics is not in the reference tree
(SURVEY F2), so the dynamics are PARITY UNPINNED against a reference binary; they are pinned to physics here:
the generalized inertia against the Jacobian form in numpy, the bias against a finite-difference Lagrangian, energy
conservation, the pendulum period, and the loader against the numbers of example/ur10/model.sdf."""
import os

import numpy as np
import pytest

from moby_amd import artic as A
from moby_amd import scene as S

REF_SDF = "/root/reference/example/ur10/model.sdf"
HERE = os.path.dirname(os.path.abspath(__file__))
UR10 = os.path.join(HERE, "scenes", "ten_joint_arm.sdf")


def rodrigues(a, th):
    a = np.asarray(a, dtype=float); K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) * np.cos(th) + (1 - np.cos(th)) * np.outer(a, a) + np.sin(th) * K


def numpy_kinematics(m, q):
    """Independent forward kinematics + Jacobians (plain numpy, link by link)."""
    nj = m.nj
    R = [None] * nj; x = [None] * nj; aw = [None] * nj
    for i in range(nj):
        p = m.parent[i]
        Rrel = np.array(m.Rrel[i][:]).reshape(3, 3); trel = np.array(m.trel[i][:]); ax = np.array(m.axis[i][:])
        if m.jtype[i] == A.MH_JOINT_REVOLUTE:
            Rl = Rrel @ rodrigues(ax, q[i]); tl = trel
        else:
            Rl = Rrel; tl = trel + Rrel @ (ax * q[i])
        Rp = np.eye(3) if p < 0 else R[p]; xp = np.zeros(3) if p < 0 else x[p]
        R[i] = Rp @ Rl; x[i] = xp + Rp @ tl; aw[i] = R[i] @ ax
    return R, x, aw


def numpy_H_and_energy(m, q, qd):
    nj = m.nj
    R, x, aw = numpy_kinematics(m, q)
    H = np.zeros((nj, nj)); pe = 0.0
    g = np.array(m.gravity[:])
    for i in range(nj):
        r = x[i] + R[i] @ np.array(m.com[i][:])
        Jv = np.zeros((3, nj)); Jw = np.zeros((3, nj))
        j = i
        while j >= 0:
            if m.jtype[j] == A.MH_JOINT_REVOLUTE:
                Jw[:, j] = aw[j]; Jv[:, j] = np.cross(aw[j], r - x[j])
            else:
                Jv[:, j] = aw[j]
            j = m.parent[j]
        Iw = R[i] @ np.array(m.inertia[i][:]).reshape(3, 3) @ R[i].T
        H += m.mass[i] * Jv.T @ Jv + Jw.T @ Iw @ Jw
        pe += -m.mass[i] * g @ r
    return H, 0.5 * qd @ H @ qd + pe


def test_sincos_kernel_is_accurate(oracle):
    xs = np.concatenate([np.linspace(-7, 7, 2001), [0.0, 1e-9, 3.141592653589793, 100.0, -250.5]])
    for xv in xs:
        s, c = oracle.sincos(float(xv))
        assert abs(s - np.sin(xv)) < 4e-16 * max(1.0, abs(xv)) and abs(c - np.cos(xv)) < 4e-16 * max(1.0, abs(xv))


@pytest.mark.parametrize("which", ["chain", "branch", "ur10"])
def test_generalized_inertia_and_poses_match_numpy(oracle, which):
    rng = np.random.default_rng(3)
    if which == "chain":
        m = A.chain_model(5, prismatic_last=True)
    elif which == "ur10":
        m, _, _ = A.load_sdf(UR10)
    else:
        links = []
        for i, p in enumerate([-1, 0, 0, 1, 1, 2]):
            Q, _ = np.linalg.qr(rng.standard_normal((3, 3))); Q = Q * np.sign(np.linalg.det(Q))
            J0 = rng.standard_normal((3, 3)); J0 = J0 @ J0.T + np.eye(3)
            links.append(dict(parent=p, type=int(i % 3 == 2), R0=Q, x0=rng.standard_normal(3), axis=rng.standard_normal(3),
                              com=0.2 * rng.standard_normal(3), inertia=0.05 * J0, mass=float(rng.uniform(0.5, 3.0))))
        m = A.model_from_links(links)
    for _ in range(3):
        q = rng.uniform(-1.5, 1.5, m.nj); qd = rng.standard_normal(m.nj)
        r = oracle.artic_fwd_dyn(m, q, qd)
        H, _ = numpy_H_and_energy(m, q, qd)
        R, x, _ = numpy_kinematics(m, q)
        assert r["ok"]
        np.testing.assert_allclose(r["H"], H, rtol=0, atol=1e-12 * max(1.0, np.abs(H).max()))
        for i in range(m.nj):
            np.testing.assert_allclose(r["poses"][i, :9].reshape(3, 3), R[i], atol=1e-13)
            np.testing.assert_allclose(r["poses"][i, 9:], x[i], atol=1e-13)
        # Lagrange: H qdd + C = 0 with C = Hdot qd - dT/dq + dV/dq, by central differences of the numpy energy pieces
        eps = 1e-6
        dLdq = np.zeros(m.nj)
        for k in range(m.nj):
            dq = np.zeros(m.nj); dq[k] = eps
            Hp, Ep = numpy_H_and_energy(m, q + dq, qd); Hm, Em = numpy_H_and_energy(m, q - dq, qd)
            Tp, Tm = 0.5 * qd @ Hp @ qd, 0.5 * qd @ Hm @ qd
            dLdq[k] = ((Tp - (Ep - Tp)) - (Tm - (Em - Tm))) / (2 * eps)        # d(T - V)/dq_k
        Hdot = (numpy_H_and_energy(m, q + eps * qd, qd)[0] - numpy_H_and_energy(m, q - eps * qd, qd)[0]) / (2 * eps)
        C_num = Hdot @ qd - dLdq
        np.testing.assert_allclose(r["C"], C_num, atol=2e-6 * max(1.0, np.abs(C_num).max()))
        np.testing.assert_allclose(H @ r["qdd"], -r["C"], atol=1e-10 * max(1.0, np.abs(r["C"]).max()))


def test_pendulum_period_and_energy(oracle):
    """One rod hinged at its end: small oscillations have T = 2 pi sqrt(I / (m g d)), I = m L^2 / 3, d = L / 2; the
    semi-implicit Euler stepper keeps the energy within O(dt)."""
    L, dt = 0.5, 1e-4
    m = A.chain_model(1, length=L, lo=-10, hi=10)
    q = np.array([[0.05]]); qd = np.zeros((1, 1)); aux = S.new_aux(1)
    T_expected = 2 * np.pi * np.sqrt((L * L / 3.0) / (9.81 * L / 2.0))
    _, e0 = numpy_H_and_energy(m, q[0], qd[0])
    crossings = []; prev = q[0, 0]
    for k in range(int(2.2 * T_expected / dt)):
        oracle.artic_step(m, q, qd, aux, dt, 1)
        if prev > 0.0 >= q[0, 0] and qd[0, 0] < 0:
            crossings.append((k + 1) * dt)
        prev = q[0, 0]
    assert len(crossings) == 2
    assert abs((crossings[1] - crossings[0]) - T_expected) < 2e-3 * T_expected
    _, e1 = numpy_H_and_energy(m, q[0], qd[0])
    assert abs(e1 - e0) < 1e-4 * abs(e0 - (-m.mass[0] * 9.81 * L / 2.0)) + 1e-6
    assert aux["steps"][0] == int(2.2 * T_expected / dt) and aux["lcp_solves"][0] == 0


def test_joint_limit_stops_the_joint_and_follows_the_reference_rules(oracle):
    """A rod falling onto its upper limit: the limit row appears when q >= hi (ArticulatedBody.inl:31), the LCP removes the
    approach velocity (restitution 0) or reverses it (restitution 0.5: l *= eps, ICH:497-525), q never runs away."""
    for eps, expect in ((0.0, 0.0), (0.5, -0.5)):
        m = A.chain_model(1, length=0.5, lo=-0.3, hi=0.3, restitution=eps)
        q = np.array([[0.29]]); qd = np.array([[2.0]]); aux = S.new_aux(1)
        v_before = None
        for k in range(40):
            before = qd[0, 0]
            oracle.artic_step(m, q, qd, aux, 1e-3, 1)
            if aux["lcp_solves"][0] == 1 and v_before is None:
                v_before = before
                assert q[0, 0] >= 0.3
                # the velocity the impact saw is `before` advanced by gravity for one step; afterwards it is -eps times it
                assert qd[0, 0] <= 1e-12 if eps == 0.0 else qd[0, 0] < 0
                if eps > 0:
                    assert abs(qd[0, 0] / before - expect) < 0.02
        assert v_before is not None and aux["status"][0] == 0
        assert q[0, 0] < 0.3 + 3e-3


def test_sdf_loader_reads_the_ur10_model():
    """The numbers of example/ur10/model.sdf: 10 joints -- 8 revolute (two of them +-1e-5 'fixed'), 2
    prismatic fingers --, a chain to the hand, then the two fingers as siblings."""
    path = REF_SDF if os.path.exists(REF_SDF) else UR10
    m, links, joints = A.load_sdf(path)
    assert m.nj == 10
    assert links == ["base_link", "shoulder_link", "upper_arm_link", "forearm_link", "wrist_1_link", "wrist_2_link", "wrist_3_link",
                     "hand", "l_finger", "r_finger"]
    assert joints[0] == "world_joint" and joints[7] == "fixed_hand_to_wrist" and joints[8:] == ["l_finger_actuator", "r_finger_actuator"]
    assert list(m.parent[:10]) == [-1, 0, 1, 2, 3, 4, 5, 6, 7, 7]
    assert list(m.jtype[:10]) == [0] * 8 + [1, 1]
    assert m.mass[1] == 7.778 and m.mass[2] == 12.93 and m.mass[9] == 0.12
    assert (m.lolimit[0], m.hilimit[0]) == (-0.00001, 0.00001) and (m.lolimit[8], m.hilimit[8]) == (-0.00001, 0.014)
    assert m.hilimit[1] == 6.28319
    np.testing.assert_allclose(m.trel[1][:], [0, 0, 0.1273], atol=1e-15)              # shoulder_link above base_link
    np.testing.assert_allclose(m.com[2][:], [0, 0, 0.306], atol=1e-15)
    np.testing.assert_allclose(np.array(m.inertia[2][:]).reshape(3, 3), np.diag([0.421754, 0.421754, 0.0363656]), atol=1e-15)
    # shoulder_lift: xyz = 0 1 0 in the PARENT link's frame (use_parent_model_frame = 1); upper_arm_link is posed with
    # rpy = (pi, pi/2, pi), so in its own frame the axis is still +-y and certainly unit
    ax = np.array(m.axis[2][:]); assert abs(np.linalg.norm(ax) - 1) < 1e-12 and abs(abs(ax[1]) - 1) < 1e-5
    # at q = 0 every link sits where the file puts it
    from tests.oracle_api import Oracle
    o = Oracle(os.path.join(os.path.dirname(HERE), "oracle", "liboracle.so"))
    r = o.artic_fwd_dyn(m, np.zeros(10), np.zeros(10))
    np.testing.assert_allclose(r["poses"][3, 9:], [0.612, 0.049041, 0.1273], atol=1e-12)
    np.testing.assert_allclose(r["poses"][9, 9:], [1.1843, 0.256, 0.0116], atol=1e-12)
    assert r["ok"] and np.linalg.eigvalsh(r["H"]).min() > 0


@pytest.mark.parametrize("which", ["chain", "branch", "ur10"])
def test_articulated_body_algorithm_equals_crb_plus_cholesky(oracle, which):
    """RCArticulatedBody's other algorithm_type, eFeatherstone (FSAB): the O(n) articulated-body recursion gives the accelerations
    of H qdd = tau - C -- with and without joint torques --, and a stepped trajectory stays with the CRB one to round-off."""
    import copy
    rng = np.random.default_rng(8)
    if which == "chain":
        m = A.chain_model(6, prismatic_last=True)
    elif which == "ur10":
        m, _, _ = A.load_sdf(UR10)
    else:
        links = []
        for i, p in enumerate([-1, 0, 0, 1, 1, 2, 4]):
            Q, _ = np.linalg.qr(rng.standard_normal((3, 3))); Q = Q * np.sign(np.linalg.det(Q))
            J0 = rng.standard_normal((3, 3)); J0 = J0 @ J0.T + np.eye(3)
            links.append(dict(parent=p, type=int(i % 3 == 2), R0=Q, x0=rng.standard_normal(3), axis=rng.standard_normal(3),
                              com=0.2 * rng.standard_normal(3), inertia=0.05 * J0, mass=float(rng.uniform(0.5, 3.0))))
        m = A.model_from_links(links)
    ma = copy.deepcopy(m) if not hasattr(m, "_b_needsfree_") else type(m).from_buffer_copy(m)
    ma.algorithm = A.MH_ARTIC_FSAB
    for trial in range(4):
        q = rng.uniform(-1.5, 1.5, m.nj); qd = rng.standard_normal(m.nj)
        tau = None if trial % 2 == 0 else rng.standard_normal(m.nj)
        r_crb = oracle.artic_fwd_dyn(m, q, qd, tau)
        r_aba = oracle.artic_fwd_dyn(ma, q, qd, tau)
        assert r_crb["ok"] and r_aba["ok"]
        scale = max(1.0, np.abs(r_crb["qdd"]).max())
        np.testing.assert_allclose(r_aba["qdd"], r_crb["qdd"], atol=1e-9 * scale)
        assert np.array_equal(r_aba["H"], r_crb["H"])               # the generalized inertia is CRB's either way
    B = 3
    q0 = rng.uniform(-0.3, 0.3, (B, m.nj)); qd0 = rng.uniform(-0.5, 0.5, (B, m.nj))
    qa, qda, auxa = q0.copy(), qd0.copy(), S.new_aux(B)
    qc, qdc, auxc = q0.copy(), qd0.copy(), S.new_aux(B)
    oracle.artic_step(ma, qa, qda, auxa, 5e-4, 200); oracle.artic_step(m, qc, qdc, auxc, 5e-4, 200)
    np.testing.assert_allclose(qa, qc, atol=1e-8); np.testing.assert_allclose(qda, qdc, atol=1e-6)
    assert np.array_equal(auxa["status"], auxc["status"])


@pytest.mark.parametrize("which", ["chain", "ur10"])
def test_link_jacobian_is_the_derivative_of_the_link_pose(oracle, which):
    """calc_jacobian: J qd = the velocity of a point carried by the link (central differences of the numpy kinematics) and its
    angular velocity; columns of joints off the link's path are zero."""
    rng = np.random.default_rng(12)
    m = A.chain_model(5, prismatic_last=True) if which == "chain" else A.load_sdf(UR10)[0]
    for link in (0, m.nj // 2, m.nj - 1):
        q = rng.uniform(-1.0, 1.0, m.nj); qd = rng.standard_normal(m.nj)
        R, x, _ = numpy_kinematics(m, q)
        pl = 0.3 * rng.standard_normal(3)
        p = x[link] + R[link] @ pl
        J = oracle.artic_jacobian(m, q, link, p)
        eps = 1e-6
        Rp, xp, _ = numpy_kinematics(m, q + eps * qd); Rm, xm, _ = numpy_kinematics(m, q - eps * qd)
        v_num = ((xp[link] + Rp[link] @ pl) - (xm[link] + Rm[link] @ pl)) / (2 * eps)
        W = (Rp[link] - Rm[link]) / (2 * eps) @ R[link].T          # [w]x
        w_num = np.array([W[2, 1], W[0, 2], W[1, 0]])
        np.testing.assert_allclose(J[:3] @ qd, v_num, atol=1e-7); np.testing.assert_allclose(J[3:] @ qd, w_num, atol=1e-7)
        on_path = set(); j = link
        while j >= 0:
            on_path.add(j); j = m.parent[j]
        for j in range(m.nj):
            assert (j in on_path) or not J[:, j].any()
