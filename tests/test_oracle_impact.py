"""CPU tests of the oracle's impact-handler entry (oracle_impact_process: ImpactConstraintHandler::process_constraints
on an explicit contact list, ICH:75-168, 530-626), the checker of include/moby_hip_impact.h."""
import ctypes

import numpy as np
import pytest

from moby_amd import _lib, impact as I, scene as S


def run(oracle, nb, mass, J, state, contacts, n=None, aux=None):
    n = I.lcp_size(len(contacts), int(contacts["nk"][0])) if n is None else n
    aux = S.new_aux(1) if aux is None else aux
    st = state.copy()
    zl = np.zeros(n); zb = np.zeros(n)
    imp, order = oracle.impact_process(nb, mass, J, st, contacts, aux, zl, zb, n)
    return st, imp, order, aux


def test_known_answer_box_stack_normal_impulses(oracle):
    """Frictionless resting stack after one free-fall step (v_y = -g dt): the impact brings every box to rest and the
    normal impulses of interface k carry the momentum of everything above it: sum cn = g dt sum_{j >= k} m_j."""
    nbx = 3
    mass, J, st, cs = I.box_stack(nbx, B=1, mu=0.0)
    s1, imp, order, aux = run(oracle, nbx, mass, J, st[0], cs[0])
    assert aux["status"][0] == 0 and aux["lcp_solves"][0] == 1 and aux["lcp_rows"][0] == I.lcp_size(12, 4)
    v = s1.reshape(nbx, 13)
    assert np.abs(v[:, 7:13]).max() < 1e-12
    for k in range(nbx):
        np.testing.assert_allclose(imp[4 * k:4 * k + 4, 0].sum(), 9.81e-3 * mass[k:].sum(), rtol=1e-10)
    assert np.abs(imp[:, 1:]).max() < 1e-12
    assert (imp[:, 0] > -1e-12).all()
    assert list(order) == list(range(12))          # breadth-first from box 0: its ground contacts, then interface 1, ...


def test_matches_the_scene_based_path_on_a_box_on_the_plane(oracle):
    """The same impact through the two oracle entries: the world stepper's contact generation + handle_impacts
    (pinned by regress/sitting-box.dat) and the explicit contact list (vertex contacts as
    CCD::find_contacts_plane_generic creates them: geom1 = plane, geom2 = box, normal = -plane normal, CCD.inl:866-881)."""
    sc = S.box_scene(mu_coulomb=0.1, nk=4)
    st = S.box_state(pos=(0.0, 0.5, 0.0), v=(0.3, -0.2, 0.1), w=(0.2, 1.0, -0.3))[0]
    a1 = S.new_aux(1); s_scene = st.copy()
    oracle.world_handle_impacts(sc, s_scene, a1)
    cs = np.zeros(4, dtype=I.CONTACT_DTYPE)
    for k, (sx, sz) in enumerate(((1, 1), (1, -1), (-1, 1), (-1, -1))):        # BoxPrimitive.cpp:358-365 order, y = -1/2 vertices
        cs["point"][k] = (0.5 * sx, 0.0, 0.5 * sz)
    cs["normal"] = (0.0, -1.0, 0.0); cs["body1"] = 1; cs["body2"] = 0; cs["mu_coulomb"] = 0.1; cs["nk"] = 4
    mass = np.array([sc.mass[0]]); J = np.array([[sc.inertia[0][k] for k in range(3)]])
    s_list, imp, order, a2 = run(oracle, 1, mass, J, st, cs)
    assert np.array_equal(s_scene, s_list)
    assert not np.array_equal(s_list, st)
    for f in ("status", "lcp_rows", "lcp_pivots", "lcp_solves"):
        assert a1[f][0] == a2[f][0], f
    assert np.array_equal(a1["rng"], a2["rng"])


def reference_bfs(nb, b1, b2):
    """UnilateralConstraint::determine_connected_constraints (UC:1085-1146) as written: neighbours are pushed once per
    multimap edge while not yet processed.  Returns the first island's constraint order."""
    nc = len(b1)
    edges = {}
    nodes = set()
    for i in range(nc):
        a, b = b1[i], b2[i]
        for x in (a, b):
            if 0 <= x < nb:
                nodes.add(x)
        if 0 <= a < nb and 0 <= b < nb:
            edges.setdefault(a, []).append(b); edges.setdefault(b, []).append(a)
    remaining = list(range(nc))
    start = min(nodes)
    q = [start]; processed = set(); order = []
    pops = 0
    while q:
        nd = q.pop(0); pops += 1
        assert pops < 200000
        processed.add(nd)
        for o in edges.get(nd, []):
            if o not in processed:
                q.append(o)
        keep = []
        for i in remaining:
            if b1[i] == nd or b2[i] == nd:
                order.append(i)
            else:
                keep.append(i)
        remaining = keep
    return order


@pytest.mark.parametrize("seed", range(6))
def test_island_order_equals_the_reference_queue_with_duplicates(oracle, seed):
    """The oracle marks nodes when first pushed (linear time); the reference re-pushes a node once per parallel edge
    (exponential in the height of a stack).  Same island order on random connected multigraphs."""
    rng = np.random.default_rng(seed)
    nb = int(rng.integers(2, 7)); nc = int(rng.integers(nb, 3 * nb))
    b1 = np.zeros(nc, dtype=int); b2 = np.zeros(nc, dtype=int)
    for i in range(nc):
        if i < nb - 1:
            a, b = i + 1, int(rng.integers(0, i + 1))          # spanning tree: one island
        else:
            a = int(rng.integers(0, nb)); b = int(rng.integers(0, nb + 1))
            while b == a:
                b = int(rng.integers(0, nb + 1))
        if rng.random() < 0.5 and b < nb:
            a, b = b, a
        b1[i], b2[i] = a, b
    perm = rng.permutation(nc); b1, b2 = b1[perm], b2[perm]
    cs = np.zeros(nc, dtype=I.CONTACT_DTYPE)
    cs["body1"] = b1; cs["body2"] = b2; cs["nk"] = 4; cs["mu_coulomb"] = 0.5
    nrm = rng.standard_normal((nc, 3)); cs["normal"] = nrm / np.linalg.norm(nrm, axis=1)[:, None]
    cs["point"] = rng.standard_normal((nc, 3))
    st = np.zeros((nb, 13)); st[:, 0:3] = rng.standard_normal((nb, 3)); st[:, 6] = 1.0     # at rest: nothing is solved
    _, _, order, aux = run(oracle, nb, np.ones(nb), np.ones((nb, 3)), st.reshape(-1), cs)
    assert aux["lcp_solves"][0] == 0
    assert list(order) == reference_bfs(nb, list(b1), list(b2))


def test_stack_of_sixteen_terminates_and_rests(oracle):
    """4^16 queue entries in the reference's breadth-first search; linear here.  lcp_fast fails on every rung of the
    ladder (the 4 corner contacts of a face are redundant), Lemke solves it."""
    nbx = 16
    mass, J, st, cs = I.box_stack(nbx, B=1, mu=0.0)
    s1, imp, order, aux = run(oracle, nbx, mass, J, st[0], cs[0])
    assert aux["status"][0] == 0
    assert np.abs(s1.reshape(nbx, 13)[:, 7:13]).max() < 1e-9
    np.testing.assert_allclose(imp[:4, 0].sum(), 9.81e-3 * mass.sum(), rtol=1e-8)


def test_impact_never_adds_kinetic_energy_without_restitution(oracle):
    rng = np.random.default_rng(5)
    nbx = 3
    mass, J, st, cs = I.box_stack(nbx, B=1, mu=0.4)
    for trial in range(5):
        s0 = st[0].copy().reshape(nbx, 13)
        s0[:, 7:10] += 0.2 * rng.standard_normal((nbx, 3)); s0[:, 10:13] += 0.2 * rng.standard_normal((nbx, 3))
        s1, imp, _, aux = run(oracle, nbx, mass, J, s0.reshape(-1), cs[0])
        assert aux["status"][0] & ~S.MH_WORLD_IMPACT_TOL == 0
        def ke(s):
            s = s.reshape(nbx, 13)
            return 0.5 * (mass[:, None] * s[:, 7:10] ** 2).sum() + 0.5 * (J * s[:, 10:13] ** 2).sum()     # identity orientation
        assert ke(s1) <= ke(s0.reshape(-1)) + 1e-12


def test_c_abi_argument_errors_need_no_gpu():
    lib = _lib.load()
    h = ctypes.c_void_p()
    m = np.ones(2); J = np.ones((2, 3))
    assert lib.mh_impact_batch_create(1, 2, 4, 5, m.ctypes.data, J.ctypes.data, ctypes.byref(h)) == _lib.MH_ERR_INVALID_ARG      # odd nk
    assert lib.mh_impact_batch_create(1, 2, 4, 2, m.ctypes.data, J.ctypes.data, ctypes.byref(h)) == _lib.MH_ERR_INVALID_ARG      # nk < 4
    assert lib.mh_impact_batch_create(1, 2, 512, 8, m.ctypes.data, J.ctypes.data, ctypes.byref(h)) == _lib.MH_ERR_UNSUPPORTED_N  # n = 5120
    assert lib.mh_impact_batch_create(1, 2, 600, 4, m.ctypes.data, J.ctypes.data, ctypes.byref(h)) == _lib.MH_ERR_INVALID_ARG
    assert lib.mh_impact_batch_create(0, 2, 4, 4, m.ctypes.data, J.ctypes.data, ctypes.byref(h)) == _lib.MH_ERR_INVALID_ARG
    m[1] = -1.0
    assert lib.mh_impact_batch_create(1, 2, 4, 4, m.ctypes.data, J.ctypes.data, ctypes.byref(h)) == _lib.MH_ERR_INVALID_ARG
    assert b"mass" in lib.mh_last_error()
    assert lib.mh_impact_batch_lcp_size(None) == 0
    assert I.lcp_size(256, 4) == 2048 and I.CONTACT_DTYPE.itemsize == 96


# ---- Anitescu-Potra model (ImpactConstraintHandlerLCP.cpp; the reference's -DUSE_AP build) ---------------------------
@pytest.fixture
def ap_oracle(oracle):
    oracle.set_impact_model(I.MH_IMPACT_MODEL_AP)
    yield oracle
    oracle.set_impact_model(I.MH_IMPACT_MODEL_DS)


def test_ap_known_answer_box_stack_normal_impulses(ap_oracle):
    """The frictionless resting stack again: the A-P LCP (5 nc + nc rows at nk = 4, Lemke on a fresh z) brings every box to
    rest, interface k carries the momentum of everything above it, and the velocities come from the accumulated
    contact wrenches (apply_impulses), not from X C^T z."""
    nbx = 3
    mass, J, st, cs = I.box_stack(nbx, B=1, mu=0.0)
    n = I.ap_lcp_size(12, 4)
    assert n == 72
    s1, imp, order, aux = run(ap_oracle, nbx, mass, J, st[0], cs[0], n=I.lcp_size(12, 4))
    assert aux["status"][0] == 0 and aux["lcp_solves"][0] == 1 and aux["lcp_rows"][0] == n
    assert np.abs(s1.reshape(nbx, 13)[:, 7:13]).max() < 1e-12
    for k in range(nbx):
        np.testing.assert_allclose(imp[4 * k:4 * k + 4, 0].sum(), 9.81e-3 * mass[k:].sum(), rtol=1e-10)
    assert (imp[:, 0] > -1e-12).all()


@pytest.mark.parametrize("nk", [4, 8, 16])
def test_ap_lcp_structure(ap_oracle, nk):
    """[UL UR; LL 0]: UL symmetric PSD with the sign pattern of [n, s, -s, t, -t]; LL = [mu, -c, -c, -s, -s] per polygon
    row with c_k = cos(pi k / (2 nk4)); UR = the friction part of -LL^T; q = [Cn v, Cs v, -Cs v, Ct v, -Ct v, 0]."""
    nbx, nc = 2, 8
    mass, J, st, cs = I.box_stack(nbx, B=1, mu=0.3, nk=nk)
    s0 = st[0].copy().reshape(nbx, 13); s0[:, 7:13] += 0.1 * np.random.default_rng(1).standard_normal((nbx, 6))
    n, MM, qq = ap_oracle.impact_lcp(nbx, mass, J, s0.reshape(-1).copy(), cs[0], 400)
    nk4 = (nk + 4) // 4 if nk > 4 else 1
    assert n == I.ap_lcp_size(nc, nk) == 5 * nc + nc * nk4
    UL, UR, LL = MM[:5 * nc, :5 * nc], MM[:5 * nc, 5 * nc:], MM[5 * nc:, :5 * nc]
    assert np.array_equal(UL, UL.T) and np.linalg.eigvalsh(UL).min() > -1e-12
    assert np.array_equal(UL[nc:2 * nc], -UL[2 * nc:3 * nc]) and np.array_equal(UL[3 * nc:4 * nc], -UL[4 * nc:5 * nc])
    assert np.array_equal(MM[5 * nc:, 5 * nc:], np.zeros((nc * nk4, nc * nk4)))
    assert np.array_equal(UR[nc:], -LL[:, nc:].T) and np.array_equal(UR[:nc], np.zeros((nc, nc * nk4)))
    for i in range(nc):
        for k in range(nk4):
            row = LL[i * nk4 + k]
            c, s = (np.cos(np.pi * k / (2.0 * nk4)), np.sin(np.pi * k / (2.0 * nk4))) if nk > 4 else (1.0, 1.0)
            expect = np.zeros(5 * nc); expect[i] = 0.3; expect[nc + i] = expect[2 * nc + i] = -c; expect[3 * nc + i] = expect[4 * nc + i] = -s
            np.testing.assert_allclose(row, expect, atol=1e-16)
    assert np.array_equal(qq[nc:2 * nc], -qq[2 * nc:3 * nc]) and np.array_equal(qq[3 * nc:4 * nc], -qq[4 * nc:5 * nc])
    assert np.array_equal(qq[5 * nc:], np.zeros(nc * nk4))
    # the DS problem of the same island shares the first block row's data
    ap_oracle.set_impact_model(I.MH_IMPACT_MODEL_DS)
    n2, M2, q2 = ap_oracle.impact_lcp(nbx, mass, J, s0.reshape(-1).copy(), cs[0], 400)
    assert np.array_equal(M2[:nc, :nc], UL[:nc, :nc]) and np.array_equal(q2[:nc], qq[:nc])


def test_ap_momentum_balance_and_non_penetration(ap_oracle):
    """Random pre-impact velocities, friction and restitution: the change of every body's momentum equals the sum of the
    reported contact impulses acting on it (apply_impulses through the wrenches == sum of n cn + s cs + t ct), no contact
    approaches afterwards (beyond the impact tolerance), friction impulses stay inside the 4-edge cone's bound."""
    from moby_amd import synth
    rng = np.random.default_rng(11)
    nbx = 3
    for trial, (mu, eps) in enumerate([(0.4, 0.0), (0.2, 0.5), (1.0, 0.0), (0.0, 0.8)]):
        mass, J, st, cs = I.box_stack(nbx, B=1, mu=mu, epsilon=eps)
        s0 = st[0].copy().reshape(nbx, 13)
        s0[:, 7:10] += 0.2 * rng.standard_normal((nbx, 3)); s0[:, 10:13] += 0.2 * rng.standard_normal((nbx, 3))
        s1, imp, _, aux = run(ap_oracle, nbx, mass, J, s0.reshape(-1), cs[0])
        assert aux["status"][0] & ~S.MH_WORLD_IMPACT_TOL == 0
        s1 = s1.reshape(nbx, 13)
        dp = mass[:, None] * (s1[:, 7:10] - s0[:, 7:10])
        acc = np.zeros((nbx, 3))
        for i, c in enumerate(cs[0]):
            n = np.array(c["normal"]); s, t = synth.orthonormal_basis(n)
            j = n * imp[i, 0] + np.array(s) * imp[i, 1] + np.array(t) * imp[i, 2]
            if 0 <= c["body1"] < nbx: acc[c["body1"]] += j
            if 0 <= c["body2"] < nbx: acc[c["body2"]] -= j
        np.testing.assert_allclose(dp, acc, atol=1e-12)
        assert (imp[:, 0] > -1e-10).all()
        if eps == 0.0:
            # A-P's friction rows: mu cn >= |cs| + |ct| for the single nk = 4 row
            assert (mu * imp[:, 0] + 1e-9 >= np.abs(imp[:, 1]) + np.abs(imp[:, 2])).all()
        # on the second-solve branch (ICH-AP:78-82) the reference never propagates the restitution impulses it has already
        # counted in Cn v: such a world ends with approaching contacts and ImpactToleranceException -- reproduced, flagged
        tol_flag = bool(aux["status"][0] & S.MH_WORLD_IMPACT_TOL)
        assert not (tol_flag and eps == 0.0)
        vmin = 0.0
        for i, c in enumerate(cs[0]):
            p = np.array(c["point"]); n = np.array(c["normal"])
            def pv(b):
                return s1[b, 7:10] + np.cross(s1[b, 10:13], p - s1[b, 0:3]) if 0 <= b < nbx else np.zeros(3)
            vmin = min(vmin, n @ (pv(c["body1"]) - pv(c["body2"])))
        assert (vmin < -S.NEAR_ZERO) == tol_flag


def test_ap_and_drumwright_shell_agree_without_friction(oracle):
    """mu = 0: both models solve the same frictionless problem (different LCPs, different impulse application)."""
    nbx = 4
    mass, J, st, cs = I.box_stack(nbx, B=2, mu=0.0)
    s_ds, imp_ds, _, _ = run(oracle, nbx, mass, J, st[1], cs[1])
    oracle.set_impact_model(I.MH_IMPACT_MODEL_AP)
    try:
        s_ap, imp_ap, _, aux = run(oracle, nbx, mass, J, st[1], cs[1])
    finally:
        oracle.set_impact_model(I.MH_IMPACT_MODEL_DS)
    assert aux["status"][0] == 0
    np.testing.assert_allclose(s_ap.reshape(nbx, 13)[:, 8], s_ds.reshape(nbx, 13)[:, 8], atol=1e-10)    # the normal direction is determined
    for k in range(nbx):
        np.testing.assert_allclose(imp_ap[4 * k:4 * k + 4, 0].sum(), imp_ds[4 * k:4 * k + 4, 0].sum(), rtol=1e-8)
