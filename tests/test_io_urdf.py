"""CPU: the URDF reader (include/moby_hip_io.h: mh_io_load_urdf, and urdf-filename on <RCArticulatedBody> in mh_io_load_xml_artic) -- the
numbers it derives from tests/scenes/arm_on_table.urdf against the same arm written as Moby XML and put together by hand, the reference's
own example/urdf files when the reference tree is present, what it rejects, and the loaded model stepped by the oracle."""
import os

import numpy as np
import pytest

from moby_amd import artic as A
from moby_amd import io as mio
from moby_amd import scene as S

HERE = os.path.dirname(os.path.abspath(__file__))
URDF = os.path.join(HERE, "scenes", "arm_on_table.urdf")
URDF_XML = os.path.join(HERE, "scenes", "arm_on_table_urdf.xml")
ARM_XML = os.path.join(HERE, "scenes", "arm_on_table.xml")
REF = "/root/reference/example/urdf"


def welded(parts):
    """[(mass, com, inertia about the com)] -> one rigid body (parallel-axis theorem)"""
    M = sum(p[0] for p in parts); c = sum(p[0] * np.asarray(p[1], float) for p in parts) / M
    I = np.zeros((3, 3))
    for mass, com, J in parts:
        d = np.asarray(com, float) - c
        I += np.asarray(J, float) + mass * (d @ d * np.eye(3) - np.outer(d, d))
    return M, c, I


def test_urdf_arm_is_the_xml_arm():
    """the first two links are arm_on_table.xml's (same frames, masses, tensors); the third carries the welded tool"""
    m, links, joints = A.load_urdf(URDF, gravity=(0.0, -9.81, 0.0))
    x = A.load_xml(ARM_XML)[0]
    assert (m.nj, links, joints) == (3, ["l1", "l2", "l3"], ["shoulder", "elbow", "slider"])
    assert list(m.parent[:3]) == [-1, 0, 1] and list(m.jtype[:3]) == [A.MH_JOINT_REVOLUTE, A.MH_JOINT_REVOLUTE, A.MH_JOINT_PRISMATIC]
    for f in ("lolimit", "hilimit"):
        assert list(getattr(m, f)[:3]) == list(getattr(x, f)[:3]), f
    assert list(m.limit_restitution[:3]) == [0.0, 0.0, 0.0]           # (a URDF joint has no restitution; the XML elbow has 0.2)
    for f in ("Rrel", "trel", "axis"):
        for i in range(3):
            assert list(getattr(m, f)[i]) == pytest.approx(list(getattr(x, f)[i]), abs=1e-15), (f, i)
    for i in range(2):
        assert m.mass[i] == pytest.approx(x.mass[i], rel=1e-15)
        assert list(m.com[i]) == pytest.approx(list(x.com[i]), abs=1e-15) and list(m.inertia[i]) == pytest.approx(list(x.inertia[i]), abs=1e-15)
    # l3 (0.15 kg at (0, -0.1, 0)) + the tool (0.05 kg at (0, -0.2, 0), its tensor turned a quarter about z)
    M, c, I = welded([(0.15, (0, -0.1, 0), np.diag([1e-3] * 3)), (0.05, (0, -0.2, 0), np.diag([1e-4, 2e-4, 3e-4]))])
    assert m.mass[2] == pytest.approx(M, rel=1e-15) and list(m.com[2]) == pytest.approx(list(c), abs=1e-15)
    assert np.array(m.inertia[2]).reshape(3, 3) == pytest.approx(I, abs=1e-18)
    assert list(m.gravity) == [0.0, -9.81, 0.0] and m.nspheres == 0       # (collision geometry is the XML route's business)


def test_urdf_through_the_xml_file():
    m, links, joints, q0, qd0, dt = A.load_xml(URDF_XML)
    u = A.load_urdf(URDF, gravity=(0.0, -9.81, 0.0))[0]
    assert (m.nj, links, joints, dt) == (3, ["l1", "l2", "l3"], ["shoulder", "elbow", "slider"], 1e-3)
    assert np.array_equal(q0, np.zeros(3)) and np.array_equal(qd0, np.zeros(3))
    for f in ("parent", "jtype", "mass", "lolimit", "hilimit", "gravity"):
        assert list(getattr(m, f)) == list(getattr(u, f)), f
    for f in ("Rrel", "trel", "axis", "com", "inertia"):
        assert all(list(getattr(m, f)[i]) == list(getattr(u, f)[i]) for i in range(3)), f
    # l1's cylinder cannot meet the table (pair disabled); l2's ball; the tool's sphere rides on l3 at the weld
    assert m.nspheres == 2 and list(m.sphere_link[:2]) == [1, 2] and list(m.sphere_radius[:2]) == [0.08, 0.05]
    assert list(m.sphere_center[0]) == [0.0, -0.25, 0.0] and list(m.sphere_center[1]) == [0.0, -0.2, 0.0]
    x = A.load_xml(ARM_XML)[0]
    assert list(m.plane_R) == list(x.plane_R) and list(m.plane_o) == list(x.plane_o)
    assert (m.cp_epsilon, m.cp_mu_coulomb, m.cp_nk, m.cstab_max_iterations, m.algorithm) == (0.0, 100.0, 4, 0, A.MH_ARTIC_CRB)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_reference_pendulum_urdf_loads():
    """example/urdf/pendulum.urdf read in place: a continuous joint at (0, 0.18, 0) about z, the rod welded to the pendulum link"""
    m, links, joints = A.load_urdf(os.path.join(REF, "pendulum.urdf"), gravity=(0.0, -9.81, 0.0))
    assert (m.nj, links, joints) == (1, ["pendulum_link"], ["pendulum_joint"]) and m.parent[0] == -1 and m.jtype[0] == A.MH_JOINT_REVOLUTE
    assert list(m.trel[0]) == [0.0, 0.18, 0.0] and list(m.Rrel[0]) == [1, 0, 0, 0, 1, 0, 0, 0, 1] and list(m.axis[0]) == [0.0, 0.0, 1.0]
    assert (m.lolimit[0], m.hilimit[0]) == (-10000.0, 10000.0)          # URDFReader.cpp:334-337 (effort = 0 alone sets neither limit)
    M, c, I = welded([(0.05, (0.25, 0, 0), np.diag([1e-3] * 3)), (0.001, (0, 0, 0), np.diag([1e-3] * 3))])
    assert m.mass[0] == pytest.approx(M, rel=1e-15) and list(m.com[0]) == pytest.approx(list(c), abs=1e-16)
    assert np.array(m.inertia[0]).reshape(3, 3) == pytest.approx(I, abs=1e-18)
    # the scene file next to it: the rod's cylinder could meet the ground plane -- outside the sphere / plane pair this build generates contacts for
    with pytest.raises(mio.SceneError, match="only Sphere"):
        A.load_xml(os.path.join(REF, "pendulum-urdf.xml"))


def test_unsupported_urdf_files_are_rejected(tmp_path):
    src = open(URDF).read()
    cases = {"not supported": src.replace('type="prismatic"', 'type="floating"'),
             "closed chains": src.replace('<child link="tool" />', '<child link="l2" />'),
             "exactly one base": src.replace('<parent link="l3" />', '<parent link="nowhere" />').replace('<link name="tool">', '<link name="nowhere" /><link name="tool">'),
             "unknown link": src.replace('<child link="l1" />', '<child link="l9" />'),
             "no mass": src.replace('<mass value="0.15" />', '<mass value="0" />').replace('<mass value="0.05" />', '<mass value="0" />'),
             "<robot>": src.replace("<robot", "<model").replace("</robot>", "</model>"),
             "zero axis": src.replace('<axis xyz="0 -1 0" />', '<axis xyz="0 0 0" />')}
    for why, text in cases.items():
        p = tmp_path / "x.urdf"; p.write_text(text)
        with pytest.raises(mio.SceneError, match=why):
            A.load_urdf(str(p))
    # a missing URDF file named by the XML file
    x = tmp_path / "a.xml"; x.write_text(open(URDF_XML).read())
    with pytest.raises(mio.SceneError, match="cannot parse"):
        A.load_xml(str(x))


def test_limit_search_and_tensor_check_follow_the_reference(tmp_path):
    """URDFReader::read_limits (URDFReader.cpp:532-566) takes the FIRST <limit> child that carries effort, lower or upper -- one with a velocity only is passed
    over -- and read_inertial (:613-616) disables a link whose tensor is not positive definite just as it disables one without mass (refused here: a disabled
    link inside the tree has no dynamics)."""
    txt = open(URDF).read()
    assert '<limit lower="-2.5" upper="2.5"' in txt or "<limit" in txt
    import re
    m0, _, _ = A.load_urdf(URDF, gravity=(0.0, -9.81, 0.0))
    first = re.search(r"<limit[^>]*/>", txt).group(0)
    two = txt.replace(first, '<limit velocity="3.0"/>\n    ' + first, 1)           # a velocity-only tag in front of the real one
    p = tmp_path / "two_limits.urdf"; p.write_text(two)
    m1, _, _ = A.load_urdf(str(p), gravity=(0.0, -9.81, 0.0))
    assert list(m1.lolimit[:3]) == list(m0.lolimit[:3]) and list(m1.hilimit[:3]) == list(m0.hilimit[:3])
    bad = re.sub(r'ixx="[^"]*"', 'ixx="-1.0"', txt, count=1)
    q = tmp_path / "bad_tensor.urdf"; q.write_text(bad)
    with pytest.raises(Exception) as e:
        A.load_urdf(str(q), gravity=(0.0, -9.81, 0.0))
    assert "positive definite" in str(e.value)


def test_revolute_joints_without_limits_get_a_quarter_turn(tmp_path):
    p = tmp_path / "x.urdf"; p.write_text(open(URDF).read().replace('<limit lower="-2" upper="2" effort="0" />', '<limit velocity="1" />'))
    m = A.load_urdf(str(p))[0]
    assert (m.lolimit[0], m.hilimit[0]) == (-np.pi / 2, np.pi / 2)       # URDFReader.cpp:326-331; a <limit> with velocity alone does not count (:549)


def test_loaded_urdf_arm_steps_in_the_oracle(oracle):
    """from the XML arm's initial joint positions, 1.5 s: the arm swings down, the ball and the welded tool meet the table, limits hold"""
    m = A.load_xml(URDF_XML)[0]
    q = np.array([[0.9, 0.3, 0.0]]); qd = np.array([[0.0, 0.5, 0.0]]); aux = S.new_aux(1)
    oracle.artic_step(m, q, qd, aux, 1e-3, 1500)
    assert aux["status"][0] & ~S.MH_WORLD_IMPACT_TOL == 0 and aux["lcp_solves"][0] > 0 and aux["mini_steps"][0] > 1500
    assert -0.5 - 1e-2 < q[0, 1] < 1.5 + 1e-2 and -0.05 - 1e-2 < q[0, 2] < 0.1 + 1e-2
