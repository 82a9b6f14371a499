"""Host I/O (include/moby_hip_io.h): the Moby XML subset, regress rows, compare-trajs."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from moby_amd import io as mio
from moby_amd import scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "tests", "scenes")
REF = "/root/reference/example"


def scene_bytes(sc, skip=()):
    d = {}
    for name, _ in S.mh_scene._fields_:
        if name in skip:
            continue
        v = getattr(sc, name)
        d[name] = np.array(v).tolist() if hasattr(v, "__len__") else v
    return d


def assert_scene_equal(a, b, skip=("lcp_n_max",)):
    da, db = scene_bytes(a, skip), scene_bytes(b, skip)
    for k in da:
        assert da[k] == db[k], k


def test_sphere_stack_xml_equals_the_builder():
    sc, st, ids, step = mio.load_xml(os.path.join(SCENES, "three_spheres_on_a_plane.xml"))
    assert ids == ["sph1", "sph2", "sph3", "ground"] and step == 0.0
    assert_scene_equal(sc, S.sphere_stack_scene())
    np.testing.assert_array_equal(st, S.sphere_stack_state(1))


def test_bouncing_ball_xml_equals_the_builder():
    sc, st, ids, step = mio.load_xml(os.path.join(SCENES, "dropped_spinning_ball.xml"))
    assert ids == ["ball", "ground"] and step == 0.025
    assert_scene_equal(sc, S.bouncing_ball_scene())
    np.testing.assert_array_equal(st, S.bouncing_ball_state(1))


def test_rimless_wheel_xml_equals_the_builder():
    sc, st, ids, _ = mio.load_xml(os.path.join(SCENES, "six_spoke_wheel.xml"))
    assert ids == ["WHEEL", "GROUND"]
    assert_scene_equal(sc, S.rimless_wheel_scene())
    assert sc.geom_type[0] == S.MH_GEOM_SPOKES
    np.testing.assert_array_equal(st[0, :7], [0, 0, 1.0, 0, 0, 0, 1])     # the pose the XML gives; the init plugin overrides it


def test_sitting_box_xml_equals_the_builder():
    sc, st, ids, step = mio.load_xml(os.path.join(SCENES, "resting_cube.xml"))
    assert ids == ["box", "ground"] and step == 0.1
    assert_scene_equal(sc, S.box_scene())
    np.testing.assert_array_equal(st, S.box_state(pos=(0.0, 0.50001, 0.0)))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_reference_example_files_load_to_the_same_scenes():
    """The reference's own example files (read in place, never copied) give the builders' scenes."""
    sc, st, ids, _ = mio.load_xml(os.path.join(REF, "stacks", "sphere-stack.xml"))
    assert_scene_equal(sc, S.sphere_stack_scene(), skip=("lcp_n_max", "cstab_max_iterations"))
    assert sc.cstab_max_iterations == S.MH_CSTAB_DEFAULT_MAX_ITERATIONS      # one documented default in every entry path
    np.testing.assert_array_equal(st, S.sphere_stack_state(1))
    sc, st, _, _ = mio.load_xml(os.path.join(REF, "bouncing-ball", "bouncing-ball.xml"))
    assert_scene_equal(sc, S.bouncing_ball_scene())
    sc, _, ids, _ = mio.load_xml(os.path.join(REF, "rimless-wheel", "wheel.xml"))
    assert ids == ["WHEEL", "GROUND"]            # the BOX body of wheel.xml is not in the simulator
    assert_scene_equal(sc, S.rimless_wheel_scene())
    sc, st, _, _ = mio.load_xml(os.path.join(REF, "simple-contact", "simplest.xml"))
    assert_scene_equal(sc, S.box_scene())
    np.testing.assert_array_equal(st, S.box_state(pos=(0.0, 0.5, 0.0)))
    sc, st, _, _ = mio.load_xml(os.path.join(REF, "simple-contact", "spinning-box-frictional.xml"))
    assert_scene_equal(sc, S.box_scene(mu_coulomb=0.1))
    np.testing.assert_array_equal(st, S.box_state(w=(0.0, 10.0, 0.0)))


def test_box_tower_file_becomes_a_large_world_scene(oracle):
    """example/stacks/stack.xml (three of its boxes and the ground are in the simulator, stack.xml:84-97) through mh_io_load_xml and
    BigScene.from_scene into the large-world stepper's scene: ground-box pairs as closed forms, the two touching box pairs as
    vertex-face pairs, box1-box3 (never in reach) left out, the file's ContactParameters on their pairs.  The reference's own file,
    read in place when the tree is present, says the same as the fixture written for this repository; 50 steps of the oracle keep
    the tower where it stands."""
    from moby_amd import stack as K
    sc, st, ids, _ = mio.load_xml(os.path.join(SCENES, "three_box_tower.xml"))
    assert ids == ["box1", "box2", "box3", "ground"]
    if os.path.isdir(REF):
        sc_r, st_r, ids_r, _ = mio.load_xml(os.path.join(REF, "stacks", "stack.xml"))
        assert ids_r == ids and bytes(sc_r) == bytes(sc) and np.array_equal(st_r, st)
    big = K.BigScene.from_scene(sc, st)
    assert list(zip(big.pair_a, big.pair_b, big.pair_model)) == [(0, 1, K.MH_PAIR_VERTEX_FACE), (0, 3, 0), (1, 2, K.MH_PAIR_VERTEX_FACE), (1, 3, 0), (2, 3, 0)]
    assert list(big.cp_mu_coulomb) == [1e-4, 1e-4, 1e-4, 0.0, 0.0] and big.c.nk == 4
    np.testing.assert_allclose(big.mass, [10.0, 9.025, 8.1], rtol=1e-15)
    cap = big.lcp_capacity()
    w = st.reshape(-1).copy(); aux = S.new_aux(1)
    oracle.big_step(big, w, aux, 1e-3, 50, zlast=np.zeros(cap), zbuf=np.zeros(cap), cap=cap)
    b = w.reshape(3, 13)
    assert aux["status"][0] == 0 and aux["lcp_rows"][0] >= 50 * 96          # three interfaces x 4 corners x 8 rows per step
    np.testing.assert_allclose(b[:, 1], [0.5, 1.5, 2.5], atol=1e-6)
    assert np.abs(b[:, [0, 2]]).max() < 1e-12 and np.abs(b[:, 7:13]).max() < 1e-2


def test_axis_angle_orientation(tmp_path):
    p = tmp_path / "tilted.xml"
    p.write_text('<XML><MOBY><Sphere id="s" radius="1" mass="1"/><Plane id="p"/><RigidBody id="b" position="0 2 0" aangle="0 0 2 1.5707963267948966">'
                 '<InertiaFromPrimitive primitive-id="s"/><CollisionGeometry primitive-id="s"/></RigidBody>'
                 '<RigidBody id="g" enabled="false"><CollisionGeometry primitive-id="p"/></RigidBody>'
                 '<TimeSteppingSimulator><DynamicBody dynamic-body-id="b"/><DynamicBody dynamic-body-id="g"/></TimeSteppingSimulator></MOBY></XML>')
    sc, st, ids, _ = mio.load_xml(str(p))
    np.testing.assert_allclose(st[0, 3:7], [0.0, 0.0, np.sin(np.pi / 4), np.cos(np.pi / 4)], atol=1e-15)


def test_quat_attribute_is_w_x_y_z(tmp_path):
    """XMLTree.cpp:407-419 reads `quat` as w x y z (and :89-94 writes it so): quat="1 0 0 0" is the identity, and a
    ground body posed with `quat` must rotate the plane (30 degrees about x: the plane's +Y normal tilts toward +Z)."""
    h = np.sqrt(0.5)
    c15, s15 = np.cos(np.pi / 12), np.sin(np.pi / 12)
    p = tmp_path / "quat.xml"
    p.write_text('<XML><MOBY><Sphere id="s" radius="1" mass="1"/><Plane id="p"/>'
                 '<RigidBody id="a" position="0 2 0" quat="1 0 0 0"><InertiaFromPrimitive primitive-id="s"/><CollisionGeometry primitive-id="s"/></RigidBody>'
                 '<RigidBody id="b" position="3 2 0" quat="%r 0 0 %r"><InertiaFromPrimitive primitive-id="s"/><CollisionGeometry primitive-id="s"/></RigidBody>'
                 '<RigidBody id="g" enabled="false" quat="%r %r 0 0"><CollisionGeometry primitive-id="p"/></RigidBody>'
                 '<TimeSteppingSimulator><DynamicBody dynamic-body-id="a"/><DynamicBody dynamic-body-id="b"/><DynamicBody dynamic-body-id="g"/>'
                 '<DisabledPair object1-id="a" object2-id="b"/></TimeSteppingSimulator></MOBY></XML>' % (float(h), float(h), float(c15), float(s15)))
    sc, st, ids, _ = mio.load_xml(str(p))
    np.testing.assert_allclose(st.reshape(-1, 13)[0, 3:7], [0.0, 0.0, 0.0, 1.0], atol=1e-15)          # identity, not a flip about x
    np.testing.assert_allclose(st.reshape(-1, 13)[1, 3:7], [0.0, 0.0, h, h], atol=1e-15)              # 90 degrees about z
    R = np.array(sc.plane_R).reshape(3, 3)
    np.testing.assert_allclose(R[:, 1], [0.0, np.cos(np.pi / 6), np.sin(np.pi / 6)], atol=1e-15)   # the plane normal


def test_unsupported_content_is_an_error(tmp_path):
    p = tmp_path / "joint.xml"
    p.write_text('<XML><MOBY><RCArticulatedBody id="a"/><TimeSteppingSimulator/></MOBY></XML>')
    with pytest.raises(mio.SceneError, match="articulated"):
        mio.load_xml(str(p))
    p.write_text('<XML><MOBY><Sphere id="s" radius="1" mass="1" position="0 1 0"/><RigidBody id="b"><InertiaFromPrimitive primitive-id="s"/>'
                 '<CollisionGeometry primitive-id="s"/></RigidBody><TimeSteppingSimulator><DynamicBody dynamic-body-id="b"/></TimeSteppingSimulator></MOBY></XML>')
    with pytest.raises(mio.SceneError, match="pose of its own"):
        mio.load_xml(str(p))
    with pytest.raises(mio.SceneError):
        mio.load_xml(str(tmp_path / "missing.xml"))


def test_regress_row_format_and_compare_trajs(tmp_path):
    st = np.zeros(13); st[:7] = [0, 0.50001, 0, 0, 0, 0, 1]
    assert mio.format_row(0, st, 1) == "0 0 0.50001 0 0 0 0 1"           # regress/sitting-box.dat:1
    st[:7] = [0.000250481, -1.62086e-14, 0.86617, -3.50766e-15, 0.000144603, 2.22901e-14, 1]
    assert mio.format_row(0.001, st, 1) == "0.001 0.000250481 -1.62086e-14 0.86617 -3.50766e-15 0.000144603 2.22901e-14 1"
    a, b = tmp_path / "a.dat", tmp_path / "b.dat"
    a.write_text("0 1 2\n0.1 1.5 2\n0.5\n")
    b.write_text("0 1 2\n0.1 1.5 2.25\n0.25\n")
    rc, md, tm = mio.compare_trajs(str(a), str(b), 1e-6)
    assert rc == 1 and md == 0.25 and tm == (0.5, 0.25)
    assert mio.compare_trajs(str(a), str(b), 0.3)[0] == 0
    b.write_text("0 1 2\n0.25\n")
    assert mio.compare_trajs(str(a), str(b), 1.0)[0] == -1
    exe = os.path.join(ROOT, "moby_amd", "bin", "moby-hip-compare-trajs")
    assert subprocess.call([exe, str(a), str(a), "1e-9"], stdout=subprocess.DEVNULL) == 0
