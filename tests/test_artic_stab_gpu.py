"""GPU parity of ConstraintStabilization with joint-limit rows (mh_artic.hip: k_artic_step_stab / stabilize_limits) against the oracle
(oracle/artic.hpp Artic::stabilize), through the C ABI, on the reference's joint-limit scenes and on the ur10 with
constraint-stabilization-max-iterations > 0: joint positions, velocities, rand() streams, flags and counters bit for bit."""
import os

import numpy as np
import pytest

from moby_amd import artic as A
from moby_amd import scene as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "stab_iters", "stab_rows", "lcp_alg_bytes", "vns_size")


def both(oracle, m, q0, qd0, dt, nsteps, chunks=1):
    B = q0.shape[0]
    ab = A.ArticBatch(m, q0, qd0)
    q_o, qd_o, aux_o = q0.copy(), qd0.copy(), S.new_aux(B)
    for _ in range(chunks):
        ab.step(dt, nsteps)
        oracle.artic_step(m, q_o, qd_o, aux_o, dt, nsteps)
    q_g, qd_g, aux_g = ab.download(); ab.close()
    assert np.array_equal(q_g, q_o), "max |dq| = %.3e" % np.abs(q_g - q_o).max()
    assert np.array_equal(qd_g, qd_o), "max |dqd| = %.3e" % np.abs(qd_g - qd_o).max()
    for f in FIELDS:
        assert np.array_equal(aux_g[f], aux_o[f]), f
    return q_g, qd_g, aux_g


def load(name):
    m, links, joints, q0, qd0, dt = A.load_xml(os.path.join(ROOT, "tests", "scenes", name + ".xml"))
    return m, q0, qd0, (dt or 1e-3)


def test_limit_pendulum_with_the_stabiliser_matches_the_oracle(oracle):
    m, q0, qd0, dt = load("limit_pendulum")
    B = 16
    rng = np.random.default_rng(7)
    q = np.tile(q0, (B, 1)); qd = np.tile(qd0, (B, 1))
    q[1:, 0] = rng.uniform(-0.9, 0.0, B - 1); qd[1:, 0] = rng.uniform(-150.0, 150.0, B - 1)       # both limits get hit
    _, _, aux = both(oracle, m, q, qd, dt, 40, chunks=3)
    assert (aux["stab_iters"] > 0).sum() >= B // 2 and (aux["status"] == 0).all()


@pytest.mark.parametrize("scene", ["limit_double_pendulum", "five_link_chain"])
def test_chains_with_the_stabiliser_match_the_oracle(oracle, scene):
    m, q0, qd0, dt = load(scene)
    B = 12
    rng = np.random.default_rng(11)
    q = np.tile(q0, (B, 1)); qd = np.tile(qd0, (B, 1))
    q[1:, 0] = rng.uniform(-0.95, -0.5, B - 1); qd[1:, 0] = rng.uniform(-4.0, -1.0, B - 1)        # the first joint runs into its lower limit
    q[1:, 1:] = rng.uniform(-0.09, 0.09, (B - 1, m.nj - 1)); qd[1:, 1:] = rng.uniform(-2.0, 2.0, (B - 1, m.nj - 1))
    _, _, aux = both(oracle, m, q, qd, dt, 50, chunks=4)
    assert (aux["stab_iters"] > 0).sum() >= B // 2
    assert (aux["stab_rows"][aux["stab_iters"] > 0] % (2 * m.nj) == 0).all()                        # a row for every finite limit, every iteration


def test_ur10_with_constraint_stabilisation_on_matches_the_oracle(oracle):
    """example/ur10/ur10.xml:11 switches the stabiliser off; with it ON (max-iterations 5) the ten-joint arm's stabiliser LCP has 20 rows."""
    from tests.test_artic_gpu import ur10_states
    m, _, _ = A.load_sdf(os.path.join(ROOT, "tests", "scenes", "ten_joint_arm.sdf"))
    m.cstab_max_iterations = 5
    B = 24
    q0, qd0 = ur10_states(m, B)
    lo0, hi0 = m.lolimit[0], m.hilimit[0]
    q0[: B // 2, 0] = np.linspace(lo0 + 1e-4, lo0 + 5e-3, B // 2); qd0[: B // 2, 0] = -3.0          # half of them cross joint 0's lower limit at once
    q0[B // 2:, 0] = np.linspace(hi0 - 5e-3, hi0 - 1e-4, B - B // 2); qd0[B // 2:, 0] = 3.0
    _, _, aux = both(oracle, m, q0, qd0, 5e-4, 30, chunks=2)
    assert (aux["stab_iters"] > 0).sum() >= B // 2
    nfin = sum(1 for i in range(m.nj) if m.hilimit[i] < 1e300) + sum(1 for i in range(m.nj) if m.lolimit[i] > -1e300)
    assert nfin > 16 and (aux["stab_rows"][aux["stab_iters"] > 0] % nfin == 0).all()                 # beyond the 16-row limit LCP of the impact handler


def test_two_link_chain_with_a_sphere_and_the_stabiliser_matches_the_oracle(oracle):
    """A sphere on the second link of a chain swinging into a plane, stabiliser on (3 iterations; the combination used to be refused):
    a contact row and four limit rows per stabilisation LCP, GPU = oracle."""
    m = A.add_spheres(A.chain_model(2, lo=-3.0, hi=3.0), [(1, (0.0, 0.0, -0.5), 0.05)], plane_point=(0.0, 0.0, -0.9))
    m.cstab_max_iterations = 3
    B = 6
    rng = np.random.default_rng(5)
    q = rng.uniform(-0.4, 0.4, (B, 2)); qd = rng.uniform(-2.0, 2.0, (B, 2))
    _, _, aux = both(oracle, m, q, qd, 1e-3, 200, chunks=3)
    assert (aux["stab_rows"][aux["stab_iters"] > 0] % 5 == 0).all()


@pytest.mark.parametrize("mu", ["100", "0.4"])
def test_arm_on_table_with_the_stabilisers_contact_rows_matches_the_oracle(oracle, tmp_path, mu):
    """Bodies with link spheres AND the stabiliser on (mh_artic_contacts.inc: stabilize_contacts -- the mixed LCP
    [Cn X Cn'  Cn X L'; .  L X L'] of CStab:705-904, 932-970, update_q's line search over the sphere distances): tests/scenes/
    arm_on_table.xml with constraint-stabilization-max-iterations = 10, perturbed arms falling onto the table and resting there, with the
    no-slip model (mu = 100) and the Drumwright-Shell model (mu = 0.4, whose LCP shares the HBM workspace with the stabiliser's):
    joint positions, velocities, rand() streams, flags and counters bit for bit."""
    src = open(os.path.join(ROOT, "tests", "scenes", "arm_on_table.xml")).read()
    p = tmp_path / "arm.xml"
    p.write_text(src.replace('constraint-stabilization-max-iterations="0"', 'constraint-stabilization-max-iterations="10"').replace('mu-coulomb="100"', 'mu-coulomb="%s"' % mu))
    m, links, joints, q0, qd0, dt = A.load_xml(str(p))
    assert m.nspheres == 2 and m.cstab_max_iterations == 10
    B = 8
    rng = np.random.default_rng(3)
    q = np.tile(q0, (B, 1)); qd = np.tile(qd0, (B, 1))
    q[1:, 0] += rng.uniform(-0.3, 0.3, B - 1); q[1:, 1] += rng.uniform(-0.2, 0.2, B - 1); qd[1:] += rng.uniform(-0.5, 0.5, (B - 1, m.nj))
    _, _, aux = both(oracle, m, q, qd, dt, 300, chunks=4)
    assert (aux["stab_iters"] > 0).sum() >= B // 2                              # the stabiliser really ran
    assert (aux["stab_rows"][aux["stab_iters"] > 0] % 8 == 0).all()             # two sphere rows + six limit rows per iteration
    assert ((aux["status"] & ~(S.MH_WORLD_IMPACT_TOL | S.MH_WORLD_UNSUPPORTED | S.MH_WORLD_STALLED)) == 0).all()
