"""GPU parity of contacts on articulated links (mh_artic_model.nspheres > 0: the kernel k_artic_step_contacts through the C ABI)
against oracle/artic.hpp: joint positions / velocities, rand() streams, the no-slip LCP's warm start and the counters bit for
bit, over conservative-advancement mini-steps, landings, resting contacts, contact + limit rows, restitution, both forward
dynamics algorithms; then a batch at scale through size-independent properties."""
import os

import numpy as np
import pytest

from moby_amd import artic as A
from moby_amd import scene as S
from tests.test_artic_gpu import assert_parity, ur10_states
from tests.test_oracle_artic_contacts import tip_height, tip_model

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
UR10 = os.path.join(HERE, "scenes", "ten_joint_arm.sdf")


def run(oracle, m, q0, qd0, nsteps=150, chunks=4, dt=1e-3):
    ab = A.ArticBatch(m, q0, qd0)
    aux = assert_parity(ab, oracle, m, q0, qd0, dt, nsteps, chunks)
    ab.close()
    return aux


def test_ball_on_a_rail_lands_like_the_oracle(oracle):
    links = [dict(parent=-1, type=A.MH_JOINT_PRISMATIC, R0=np.eye(3), x0=(0, 0, 0), axis=(0, 0, 1), com=(0, 0, 0),
                  inertia=np.diag([0.1, 0.1, 0.1]), mass=2.0)]
    for eps in (0.0, 0.5):
        m = A.add_spheres(A.model_from_links(links), [(0, (0, 0, 0), 0.25)], epsilon=eps)
        q0 = 0.25 + np.array([[0.02], [0.001], [1e-7], [0.3]]); qd0 = np.array([[0.0], [-0.5], [0.0], [-2.0]])
        aux = run(oracle, m, q0, qd0, nsteps=60, chunks=4)
        assert (aux["lcp_solves"][:3] > 0).all() and (aux["mini_steps"] > aux["steps"]).any()


@pytest.mark.parametrize("n,alg", [(2, A.MH_ARTIC_CRB), (3, A.MH_ARTIC_CRB), (2, A.MH_ARTIC_FSAB), (4, A.MH_ARTIC_FSAB)])
def test_arm_tip_hits_the_floor_like_the_oracle(oracle, n, alg):
    """planar n-pendulums whose tip sphere lands on a floor: mini-steps at the landing, the no-slip LCP every step afterwards"""
    m = tip_model(n, floor=-0.5 * n + 0.1)
    m.algorithm = alg
    B = 8
    rng = np.random.default_rng(40 + n)
    q0 = np.zeros((B, n)); q0[:, 0] = rng.uniform(0.5, 0.9, B); q0[:, 1:] = rng.uniform(0.0, 0.3, (B, n - 1)); qd0 = rng.uniform(-0.5, 0.5, (B, n))
    aux = run(oracle, m, q0, qd0, nsteps=150, chunks=5)
    assert (aux["lcp_solves"] > 0).all() and (aux["mini_steps"] > aux["steps"]).all()
    assert (aux["status"] & ~S.MH_WORLD_IMPACT_TOL == 0).all(), aux["status"]


def test_contact_and_limit_rows_share_one_lcp(oracle):
    """lower and upper elbow limits with the tip on the floor (the upper one shows the reference's unsigned X L' coupling), with
    restitution at the limits and at the contact"""
    for lo, hi, sgn, eps, er in ((-0.25, 3.0, -1.0, 0.0, 0.0), (-3.0, 0.25, 1.0, 0.0, 0.0), (-0.25, 3.0, -1.0, 0.3, 0.5)):
        m = tip_model(2, floor=-0.9, hi=hi, lo=lo, eps=eps, restitution=er)
        m.lolimit[0] = -3.0; m.hilimit[0] = 3.0
        B = 4
        rng = np.random.default_rng(7)
        q0 = np.column_stack([sgn * rng.uniform(0.7, 0.8, B), sgn * rng.uniform(0.15, 0.22, B)]); qd0 = np.column_stack([np.zeros(B), sgn * rng.uniform(1.0, 2.0, B)])
        aux = run(oracle, m, q0, qd0, nsteps=175, chunks=4)
        assert (aux["lcp_rows"] > aux["lcp_solves"]).all()              # some 2-row LCPs


def test_two_spheres_on_different_links(oracle):
    m = A.chain_model(3, lo=-3.0, hi=3.0)
    A.add_spheres(m, [(1, (0.0, 0.0, -0.5), 0.08), (2, (0.0, 0.0, -0.5), 0.05), (2, (0.1, 0.0, -0.25), 0.04)],
                  plane_normal=(0.1, 0.0, 1.0), plane_point=(0.0, 0.0, -1.0))
    B = 6
    rng = np.random.default_rng(3)
    q0 = np.column_stack([rng.uniform(0.8, 1.2, B), rng.uniform(0.2, 0.6, B), rng.uniform(0.2, 0.6, B)]); qd0 = rng.uniform(-0.5, 0.5, (B, 3))
    aux = run(oracle, m, q0, qd0, nsteps=200, chunks=5)
    assert (aux["lcp_solves"] > 0).all()
    assert (aux["lcp_rows"] > aux["lcp_solves"]).any()                  # two contacts at once somewhere


@pytest.mark.parametrize("mu,nk,eps,compl,visc", [(0.0, 4, 0.0, 0.0, 0.0), (0.5, 4, 0.0, 1e-6, 0.0), (0.8, 8, 0.3, 0.0, 0.05), (5.0, 6, 0.0, 0.0, 0.0)])
def test_drumwright_shell_model_on_link_contacts(oracle, mu, nk, eps, compl, visc):
    """mu_coulomb < 100: the QP -> LCP model over [cn cs ct ncs nct l] with friction polygons of nk edges, compliance, viscous
    friction, restitution (second solve included), warm starts from _zlast -- 3-link arms landing on a floor"""
    m = A.chain_model(3, lo=-3.0, hi=3.0)
    A.add_spheres(m, [(2, (0.0, 0.0, -0.5), 0.05), (1, (0.0, 0.0, -0.5), 0.07)], plane_point=(0.0, 0.0, -1.3), epsilon=eps, mu_coulomb=mu,
                  mu_viscous=visc, compliance=compl, nk=nk)
    B = 8
    rng = np.random.default_rng(int(10 * mu) + nk)
    q0 = np.column_stack([rng.uniform(0.5, 1.0, B), rng.uniform(0.0, 0.4, B), rng.uniform(0.0, 0.4, B)]); qd0 = rng.uniform(-0.5, 0.5, (B, 3))
    aux = run(oracle, m, q0, qd0, nsteps=150, chunks=5)
    assert (aux["lcp_solves"] > 0).all() and (aux["zlast_size"] >= 6 + nk // 2).all()
    assert (aux["status"] & ~(S.MH_WORLD_IMPACT_TOL | S.MH_WORLD_LCP_FAILED) == 0).all(), aux["status"]


def test_drumwright_shell_with_limit_variables(oracle):
    """contact + limit rows in the QP (n = 8 + 2 per limit), lower and upper limits, restitution at the limit"""
    for lo, hi, sgn, er in ((-0.25, 3.0, -1.0, 0.0), (-3.0, 0.25, 1.0, 0.4)):
        m = tip_model(2, floor=-0.9, hi=hi, lo=lo, mu=0.5, restitution=er)
        m.lolimit[0] = -3.0; m.hilimit[0] = 3.0
        B = 4
        rng = np.random.default_rng(17)
        q0 = np.column_stack([sgn * rng.uniform(0.7, 0.8, B), sgn * rng.uniform(0.15, 0.22, B)]); qd0 = np.column_stack([np.zeros(B), sgn * rng.uniform(1.0, 2.0, B)])
        aux = run(oracle, m, q0, qd0, nsteps=175, chunks=4)
        assert (aux["zlast_size"] == 10).any() or (aux["zlast_size"] == 8).all()


def test_ur10_fingers_on_a_table(oracle):
    """the ten-joint arm of config 5 with a sphere on each finger and one on the forearm, over a table: limits and contacts"""
    m, links, _ = A.load_sdf(UR10)
    poses0 = None
    B = 16
    q0, qd0 = ur10_states(m, B, seed=77)
    ab = A.ArticBatch(m, q0, qd0); P = ab.link_poses(); ab.close()
    lf, rf, fa = links.index("l_finger"), links.index("r_finger"), links.index("forearm_link")
    zmin = min(P[:, lf, 11].min(), P[:, rf, 11].min(), P[:, fa, 11].min())
    A.add_spheres(m, [(lf, (0.0, 0.0, 0.0), 0.03), (rf, (0.0, 0.0, 0.0), 0.03), (fa, (0.0, 0.0, 0.0), 0.06)],
                  plane_normal=(0.0, 0.0, 1.0), plane_point=(0.0, 0.0, float(zmin) - 0.15))
    aux = run(oracle, m, q0, qd0, nsteps=100, chunks=4, dt=5e-4)
    assert (aux["lcp_solves"] > 0).all()
    assert ((aux["status"] & S.MH_WORLD_UNSUPPORTED) == 0).all()


def test_many_worlds_properties():
    """2048 three-link arms over a floor, 600 steps: nothing sinks beyond the landing overshoot, worlds are independent of their
    position in the batch, every world ends up resting on the floor or swinging above it."""
    m = tip_model(3, floor=-1.4)
    B = 2048
    rng = np.random.default_rng(9)
    q0 = np.column_stack([rng.uniform(0.4, 1.0, B), rng.uniform(0.0, 0.4, B), rng.uniform(0.0, 0.4, B)]); qd0 = rng.uniform(-0.5, 0.5, (B, 3))
    ab = A.ArticBatch(m, q0, qd0); ab.step(1e-3, 600); q, qd, aux = ab.download(); ab.close()
    assert (aux["status"] & ~S.MH_WORLD_IMPACT_TOL == 0).all() and (aux["steps"] == 600).all()
    h = np.array([tip_height(m, q[b]) for b in range(0, B, 16)])
    assert (h > -3e-3).all()
    assert (aux["lcp_solves"] > 0).mean() > 0.9
    perm = rng.permutation(B)[:256]
    ab = A.ArticBatch(m, q0[perm], qd0[perm]); ab.step(1e-3, 600); q2, qd2, aux2 = ab.download(); ab.close()
    assert np.array_equal(q2, q[perm]) and np.array_equal(qd2, qd[perm]) and np.array_equal(aux2["rng"], aux["rng"][perm])


def test_xml_arm_on_table_steps_like_the_oracle(oracle, tmp_path):
    """tests/scenes/arm_on_table.xml (a Moby XML file with one RCArticulatedBody, link spheres and a table) through
    mh_io_load_xml_artic: perturbed copies of the file's initial state bit for bit against the oracle, and the regress command line
    (moby-hip-regress: rows of time + joint positions) equal to the batch's world 0"""
    import subprocess
    m, links, joints, q0, qd0, dt = A.load_xml(os.path.join(HERE, "scenes", "arm_on_table.xml"))
    B = 8
    rng = np.random.default_rng(21)
    q = np.tile(q0, (B, 1)); qd = np.tile(qd0, (B, 1)); q[1:] += rng.uniform(-0.05, 0.05, (B - 1, m.nj)); qd[1:] += rng.uniform(-0.3, 0.3, (B - 1, m.nj))
    q[:, 2] = np.clip(q[:, 2], -0.04, 0.09)
    aux = run(oracle, m, q, qd, nsteps=300, chunks=4, dt=dt)
    assert (aux["lcp_solves"] > 0).all() and (aux["mini_steps"] > aux["steps"]).all() and (aux["status"] & ~S.MH_WORLD_IMPACT_TOL == 0).all()
    root = os.path.dirname(HERE)
    opts = tmp_path / "arm.setup"; opts.write_text("-s=0.001 -mi=400\n")
    outp = tmp_path / "arm.dat"
    subprocess.check_call([os.path.join(root, "moby_amd", "bin", "moby-hip-regress"), str(opts), os.path.join(HERE, "scenes", "arm_on_table.xml"), str(outp)])
    rows = [l.split() for l in open(outp).read().strip().splitlines()]
    assert len(rows) == 401 and len(rows[0]) == 4 and [float(x) for x in rows[0]] == [0.0, 0.9, 0.3, 0.0]
    ab = A.ArticBatch(m, q0.reshape(1, -1), qd0.reshape(1, -1)); ab.step(1e-3, 399); qg, _, _ = ab.download(); ab.close()
    assert np.allclose([float(x) for x in rows[399][1:]], qg[0], rtol=0, atol=1e-5 * np.abs(qg[0]).max() + 1e-6)      # 6 printed digits


def test_urdf_arm_on_table_steps_like_the_oracle(oracle):
    """tests/scenes/arm_on_table_urdf.xml: the arm read from a URDF file (mh_io_load_xml_artic, urdf-filename; a tool welded to the slider's
    link by a fixed joint carries the tip sphere), perturbed states bit for bit against the oracle through landings and resting contacts"""
    m = A.load_xml(os.path.join(HERE, "scenes", "arm_on_table_urdf.xml"))[0]
    assert m.nspheres == 2 and m.nj == 3
    B = 8
    rng = np.random.default_rng(22)
    q = np.tile([0.9, 0.3, 0.0], (B, 1)); qd = np.tile([0.0, 0.5, 0.0], (B, 1)); q[1:] += rng.uniform(-0.05, 0.05, (B - 1, 3)); qd[1:] += rng.uniform(-0.3, 0.3, (B - 1, 3))
    q[:, 2] = np.clip(q[:, 2], -0.04, 0.09)
    aux = run(oracle, m, q, qd, nsteps=300, chunks=4, dt=1e-3)
    assert (aux["lcp_solves"] > 0).all() and (aux["mini_steps"] > aux["steps"]).all() and (aux["status"] & ~S.MH_WORLD_IMPACT_TOL == 0).all()


@pytest.mark.parametrize("mu,nk,want_flag", [(100.0, 4, False), (0.5, 4, False), (0.5, 16, False), (0.5, 20, True), (0.0, 64, True)])
def test_capacity_edges_four_contacts_and_limits(oracle, mu, nk, want_flag):
    """all MH_ARTIC_MAX_SPHERES spheres of one link land together while a second joint sits on its limit: the no-slip LCP has
    4 + 1 rows, the Drumwright-Shell LCP 24 + 2 nk + 2 rows -- 34 (nk 4), 58 (nk 16), beyond the 64-row wave solver from nk 20 on:
    flagged MH_WORLD_UNSUPPORTED and frozen, identically on both sides"""
    I = np.diag([0.05, 0.05, 0.05])
    links = [dict(parent=-1, type=A.MH_JOINT_PRISMATIC, R0=np.eye(3), x0=(0, 0, 0), axis=(0, 0, 1), com=(0, 0, 0), inertia=I, mass=2.0),
             dict(parent=0, type=A.MH_JOINT_REVOLUTE, R0=np.eye(3), x0=(0, 0, 0.3), axis=(0, 1, 0), com=(0, 0, 0.1), inertia=I, mass=0.5, lo=-0.2, hi=0.2)]
    m = A.model_from_links(links)
    A.add_spheres(m, [(0, (0.3, 0.2, 0.0), 0.1), (0, (-0.3, 0.2, 0.0), 0.1), (0, (0.3, -0.2, 0.0), 0.1), (0, (-0.3, -0.2, 0.0), 0.1)],
                  plane_point=(0.0, 0.0, -0.1), epsilon=0.0, mu_coulomb=mu, nk=nk)
    B = 4
    q0 = np.column_stack([np.array([0.02, 0.01, 0.005, 0.03]), np.full(B, -0.2)]); qd0 = np.column_stack([np.zeros(B), np.array([-0.5, -1.0, -0.2, -0.8])])
    aux = run(oracle, m, q0, qd0, nsteps=60, chunks=3)
    flagged = (aux["status"] & S.MH_WORLD_UNSUPPORTED) != 0
    assert flagged.all() if want_flag else not flagged.any(), aux["status"]
    if not want_flag:
        n_full = 5 if mu >= 100 else 24 + 2 * nk + 2
        assert (aux["lcp_rows"] >= n_full).all() and (aux["steps"] == 180).all()
    else:
        assert (aux["steps"] < 180).all()


def test_lemke_in_a_wide_drumwright_shell_lcp(oracle):
    """tests/tools/fuzz_artic.py's case 20050 (5 joints, 3 spheres, articulated-body algorithm, mu 2, compliance): at step 214 one world's
    18-row LCP (two contacts + a limit) fails lcp_fast_regularized and goes to the Lemke ladder -- whose artificial column needs as many
    entries as the LCP has rows (the first build gave it the no-slip capacity of 16 and the tail clobbered the limit tables)."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "tools"))
    import fuzz_artic as F
    m, q0, qd0, nsteps = F.complete_case(oracle, 20050)
    assert (m.nj, m.nspheres, m.algorithm, m.cp_mu_coulomb) == (5, 3, A.MH_ARTIC_FSAB, 2.0)
    aux = run(oracle, m, q0, qd0, nsteps=87, chunks=3)
    assert (aux["zbuf_cap"] >= 18).any() and aux["lcp_pivots"].max() > 1000
