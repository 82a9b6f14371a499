"""CPU: the oracle's large-world path (oracle_big_step: scenes of any size behind a SceneView, explicit candidate
pairs, the vertex-face model of stacked boxes) -- pinned to the small-world path it generalises, plus the physics of
BASELINE config 4's box stack."""
import numpy as np

from moby_amd import scene as S
from moby_amd import stack as K


def big_from_small(sc, pairs=None):
    """The BigScene that says the same as an mh_scene of spheres / boxes over a plane (all pairs enabled)."""
    nb = sc.nb
    ntot = nb + sc.has_ground
    if pairs is None:
        pairs = [(i, j) for i in range(ntot) for j in range(i + 1, ntot) if sc.pair_enabled[S.pair_index(i, j, ntot)]]
    idx = [S.pair_index(i, j, ntot) for i, j in pairs]
    return K.BigScene([sc.geom_type[b] for b in range(nb)], [[sc.geom_dim[b][k] for k in range(3)] for b in range(nb)],
                      [sc.mass[b] for b in range(nb)], [[sc.inertia[b][k] for k in range(3)] for b in range(nb)],
                      [(i, j, K.MH_PAIR_CLOSED_FORM) for i, j in pairs], gravity=[sc.gravity[k] for k in range(3)],
                      plane_R=[sc.plane_R[k] for k in range(9)], plane_o=[sc.plane_o[k] for k in range(3)], has_ground=bool(sc.has_ground),
                      nk=sc.cp_nk[idx[0]], epsilon=[sc.cp_epsilon[p] for p in idx], mu_coulomb=[sc.cp_mu_coulomb[p] for p in idx],
                      mu_viscous=[sc.cp_mu_viscous[p] for p in idx], compliance=[sc.cp_compliance[p] for p in idx],
                      cstab_max_iterations=sc.cstab_max_iterations, lcp_n_max=64)


def assert_same_world(st_a, aux_a, st_b, aux_b):
    assert np.array_equal(st_a, st_b)
    for f in ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "stab_iters", "lcp_alg_bytes", "stab_rows"):
        assert np.array_equal(aux_a[f], aux_b[f]), f


def test_big_path_equals_small_path_on_the_sphere_stack(oracle):
    sc = S.sphere_stack_scene()
    big = big_from_small(sc)
    for w in (0, 3):
        st0 = S.sphere_stack_state_range(w, 1)[0]
        st_s, aux_s = st0.copy(), S.new_aux(1)
        oracle.world_step(sc, st_s, aux_s, 1e-3, 60, want_traj=False)
        st_b, aux_b = st0.copy(), S.new_aux(1)
        zl, zb = np.zeros(64), np.zeros(64)
        oracle.big_step(big, st_b, aux_b, 1e-3, 60, zlast=zl, zbuf=zb, cap=64)
        assert_same_world(st_s, aux_s, st_b, aux_b)
        assert aux_s["lcp_rows"][0] > 0
        assert np.array_equal(zl[:aux_s["zlast_size"][0]], aux_s["zlast"][0][:aux_s["zlast_size"][0]])


def test_big_path_equals_small_path_on_a_tumbling_box(oracle):
    sc = S.box_scene(mu_coulomb=0.5, epsilon=0.3, nk=4)
    big = big_from_small(sc)
    st0 = np.zeros(13); st0[1] = 0.8; st0[6] = 1.0; st0[10:13] = (0.3, 0.2, -0.4); st0[7] = 0.1
    q = np.array([0.1, 0.2, 0.05, 1.0]); st0[3:7] = q / np.linalg.norm(q)
    st_s, aux_s = st0.copy(), S.new_aux(1)
    oracle.world_step(sc, st_s, aux_s, 1e-3, 400, want_traj=False)
    st_b, aux_b = st0.copy(), S.new_aux(1)
    oracle.big_step(big, st_b, aux_b, 1e-3, 400, cap=64)
    assert_same_world(st_s, aux_s, st_b, aux_b)
    assert aux_s["lcp_solves"][0] > 0 and aux_s["mini_steps"][0] > 400     # it really hit the plane and sub-stepped


def test_box_stack_rests_and_carries_its_weight(oracle):
    """Config 4 in the small: 3 boxes at rest, exactly touching.  They stay put (to the stabiliser's 3e-8 gaps), the
    velocities after a step are round-off, and one process_constraints per step solved the n = 96 impact LCP."""
    N = 3
    sc = K.box_stack_scene(N)
    st = K.box_stack_state(N, 1)[0].copy(); aux = S.new_aux(1)
    for s in range(4):
        oracle.big_step(sc, st, aux, 1e-3, 1)
    b = st.reshape(N, 13)
    gaps = b[:, 1] - 0.5 - np.arange(N)
    assert (gaps >= 0.0).all() and gaps.max() < 2e-7          # pushed apart by eps + NEAR_ZERO per interface, no drift
    assert np.abs(b[:, 7:13]).max() < 1e-12                   # at rest
    assert np.abs(b[:, [0, 2]]).max() < 1e-12 and np.abs(b[:, 3:6]).max() < 1e-12
    assert aux["status"][0] == 0 and aux["steps"][0] == 4 and aux["mini_steps"][0] == 4
    assert aux["lcp_rows"][0] >= 4 * 96                       # the impact LCP of every step: 12 contacts x 8 rows


def test_vertex_face_pair_reduces_to_box_plane_when_the_support_cannot_move(oracle):
    """A box dropped on a (practically) immovable support box behaves like the same box dropped on the plane at the
    support's top face: same contacts, same conservative-advancement steps -- the documented model of
    include/moby_hip_stack.h is box-plane with a moving plane."""
    dims = [(4.0, 1.0, 4.0), (1.0, 1.0, 1.0)]
    mass = [1e12, 10.0]
    inertia = [[m / 12.0 * (y * y + z * z), m / 12.0 * (x * x + z * z), m / 12.0 * (x * x + y * y)] for m, (x, y, z) in zip(mass, dims)]
    two = K.BigScene([S.MH_GEOM_BOX] * 2, dims, mass, inertia, [(0, 2, 0), (0, 1, K.MH_PAIR_VERTEX_FACE)], gravity=(0, -9.81, 0),
                     mu_coulomb=0.3, epsilon=0.2, lcp_n_max=128)
    one = K.BigScene([S.MH_GEOM_BOX], dims[1:], mass[1:], inertia[1:], [(0, 1, 0)], gravity=(0, -9.81, 0), plane_o=(0.0, 1.0, 0.0),
                     mu_coulomb=0.3, epsilon=0.2, lcp_n_max=128)
    top = np.zeros(13); top[1] = 1.5 + 0.02; top[6] = 1.0; top[8] = -0.5; top[10:13] = (0.2, 0.0, -0.1)
    sup = np.zeros(13); sup[1] = 0.5; sup[6] = 1.0
    st2 = np.concatenate([sup, top]); aux2 = S.new_aux(1)
    st1 = top.copy(); aux1 = S.new_aux(1)
    # up to the first impact the trajectories agree (the support's own mini-steps on the plane shift the sub-step times
    # by microseconds); the impact comes in the same step and leaves the same pose
    first = [None, None]
    for k in range(60):
        prev2, prev1 = st2.copy(), st1.copy()
        oracle.big_step(two, st2, aux2, 1e-3, 1)
        oracle.big_step(one, st1, aux1, 1e-3, 1)
        for i, vy in enumerate((st2[13 + 8], st1[8])):
            if first[i] is None and vy > -0.4:                            # it started at -0.5 and gravity only speeds it up
                first[i] = k
        if first[0] is not None or first[1] is not None:
            break
        np.testing.assert_allclose(st2[13:], st1, atol=2e-4)             # the support itself micro-bounces on the plane (3e-8 gaps)
    assert first[0] is not None and first[0] == first[1] and first[0] > 20
    np.testing.assert_allclose(prev2[13:], prev1, atol=2e-4)
    np.testing.assert_allclose(st2[13:20], st1[:7], atol=2e-4)          # same pose after the impact step
    assert aux1["mini_steps"][0] > first[0] + 1                         # the impact step was cut by conservative advancement
