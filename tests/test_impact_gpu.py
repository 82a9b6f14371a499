"""GPU parity of the batched impact handler (include/moby_hip_impact.h, through the C ABI) against the oracle's
restatement of ImpactConstraintHandler::process_constraints: bit-exact states, impulses, pivot counts, status bits,
rand() streams (via the pivot counts of the following call) and assembled _MM / _qq."""
import ctypes

import numpy as np
import pytest

from moby_amd import _lib, impact as I, scene as S

pytestmark = pytest.mark.gpu


def oracle_batch(oracle, nb, mass, J, state, cs, n, aux, zl, zb):
    B = state.shape[0]
    imp = np.zeros((B, cs.shape[1], 3))
    piv0 = aux["lcp_pivots"].copy(); sol0 = aux["lcp_solves"].copy()
    for w in range(B):
        imp[w], _ = oracle.impact_process(nb, mass, J, state[w], cs[w], aux[w:w + 1], zl[w], zb[w], n)
    return imp, (aux["lcp_pivots"] - piv0).astype(np.uint32), (aux["lcp_solves"] - sol0).astype(np.int32)


def assert_same(r, st_o, imp_o, piv_o, sol_o, aux):
    assert np.array_equal(r["state"], st_o), "max |dv| = %.3e" % np.abs(r["state"] - st_o).max()
    assert np.array_equal(r["impulses"], imp_o)
    assert np.array_equal(r["status"], aux["status"])
    assert np.array_equal(r["pivots"], piv_o)
    assert np.array_equal(r["solves"], sol_o)


@pytest.mark.parametrize("nbx,B,eps,mu", [(1, 4, 0.0, 1e-4), (2, 4, 0.0, 1e-4), (3, 4, 0.0, 0.3), (5, 3, 0.0, 1e-4),
                                          (2, 4, 0.5, 0.3), (4, 3, 0.3, 0.5)])
def test_box_stack_calls_match_oracle(oracle, nbx, B, eps, mu):
    """Cold call, then two warm-started calls (gravity acts for another dt on the same contacts).  eps > 0 exercises
    apply_restitution and the second solve (ICH:575-600); n = 32 nbx crosses the wave / block solver boundary at 64."""
    mass, J, st, cs = I.box_stack(nbx, B=B, epsilon=eps, mu=mu)
    nc = 4 * nbx; n = I.lcp_size(nc, 4)
    ib = I.ImpactBatch(B, nbx, nc, 4, mass, J)
    assert ib.n == n
    aux = S.new_aux(B); zl = np.zeros((B, n)); zb = np.zeros((B, n))
    st_g = st.copy(); st_o = st.copy()
    second = 0
    for call in range(3):
        r = ib.process(st_g, cs)
        if call == 0 and eps == 0.0:
            MMg, qqg = ib.debug_lcp()
            for w in range(min(B, 2)):
                nn, MMo, qqo = oracle.impact_lcp(nbx, mass, J, st_o[w].copy(), cs[w], n)
                assert nn == n and np.array_equal(MMo, MMg[w]) and np.array_equal(qqo, qqg[w])
        imp_o, piv_o, sol_o = oracle_batch(oracle, nbx, mass, J, st_o, cs, n, aux, zl, zb)
        assert_same(r, st_o, imp_o, piv_o, sol_o, aux)
        assert (r["status"] & ~S.MH_WORLD_IMPACT_TOL == 0).all()
        second += int((r["solves"] == 2).sum())
        st_g = r["state"].copy()
        for a in (st_g, st_o):
            a.reshape(B, nbx, 13)[:, :, 8] += -9.81e-3
    if eps > 0.0:
        assert second > 0          # the restitution branch with a second solve was really taken
    ib.close()


def random_island(rng, nb, nc, static_frac=0.3):
    cs = np.zeros(nc, dtype=I.CONTACT_DTYPE)
    for i in range(nc):
        if i < nb - 1:
            a, b = i + 1, int(rng.integers(0, i + 1))
        else:
            a = int(rng.integers(0, nb)); b = nb if rng.random() < static_frac else int(rng.integers(0, nb))
            while b == a:
                b = int(rng.integers(0, nb))
        if rng.random() < 0.5:
            a, b = b, a
        cs["body1"][i], cs["body2"][i] = a, b
    cs = cs[rng.permutation(nc)]
    nrm = rng.standard_normal((nc, 3)); cs["normal"] = nrm / np.linalg.norm(nrm, axis=1)[:, None]
    cs["point"] = rng.standard_normal((nc, 3))
    cs["mu_coulomb"] = rng.uniform(0.0, 1.0, nc); cs["mu_viscous"] = rng.uniform(0.0, 0.1, nc) * (rng.random(nc) < 0.3)
    cs["epsilon"] = rng.uniform(0.0, 0.8, nc) * (rng.random(nc) < 0.5); cs["compliance"] = 1e-6 * (rng.random(nc) < 0.2)
    return cs


@pytest.mark.parametrize("seed,nb,nc,nk", [(0, 2, 3, 4), (1, 3, 6, 4), (2, 4, 7, 8), (3, 5, 10, 4), (4, 3, 12, 6), (5, 6, 14, 4)])
def test_random_contact_graphs_match_oracle(oracle, seed, nb, nc, nk):
    """Random single-island multigraphs (permuted contact lists, static and dynamic pairs, arbitrary normals, rotated
    bodies, mixed parameters): island order, Jacobian rows, X blocks and the whole solve chain."""
    rng = np.random.default_rng(seed)
    B = 4
    n = I.lcp_size(nc, nk)
    mass = rng.uniform(0.5, 3.0, nb); J = rng.uniform(0.2, 2.0, (nb, 3))
    cs = np.stack([random_island(rng, nb, nc) for _ in range(B)]); cs["nk"] = nk
    st = np.zeros((B, nb, 13)); st[:, :, 0:3] = rng.standard_normal((B, nb, 3))
    q = rng.standard_normal((B, nb, 4)); st[:, :, 3:7] = q / np.linalg.norm(q, axis=2)[:, :, None]
    st[:, :, 7:13] = rng.standard_normal((B, nb, 6))
    st = st.reshape(B, -1)
    ib = I.ImpactBatch(B, nb, nc, nk, mass, J)
    aux = S.new_aux(B); zl = np.zeros((B, n)); zb = np.zeros((B, n))
    st_o = st.copy(); st_g = st.copy()
    for call in range(2):
        r = ib.process(st_g, cs)
        imp_o, piv_o, sol_o = oracle_batch(oracle, nb, mass, J, st_o, cs, n, aux, zl, zb)
        assert_same(r, st_o, imp_o, piv_o, sol_o, aux)
        st_g = r["state"].copy()
        st_g.reshape(B, nb, 13)[:, :, 7:13] += 0.1 * rng.standard_normal((B, nb, 6)); st_o[:] = st_g
    assert (r["solves"] >= 1).any()
    ib.close()


def test_islands_loop_noslip_model_and_idle_worlds(oracle):
    """World 0: two islands (ICH:105-151 handles them one after the other -- so does this entry); world 1: every
    mu >= 100 (the no-slip model, ICH:134-135, 1009-1417); world 2: nothing impacting; world 3: ordinary."""
    nbx, B = 2, 4
    mass, J, st, cs = I.box_stack(nbx, B=B, perturb=False)
    cs["body2"][0, 4:] = 99                              # box 1 now rests on something static: two islands
    cs["mu_coulomb"][1] = 100.0
    st.reshape(B, nbx, 13)[2, :, 8] = 0.5                # separating
    st.reshape(B, nbx, 13)[1, :, 7] = 0.02               # world 1 slides: the no-slip impulses have something to stop
    r = I.ImpactBatch(B, nbx, 8, 4, mass, J).process(st, cs)
    assert list(r["status"] & ~S.MH_WORLD_IMPACT_TOL) == [0, 0, 0, 0]
    assert np.array_equal(r["state"][2], st[2]) and not np.array_equal(r["state"][3], st[3]) and not np.array_equal(r["state"][1], st[1])
    assert list(r["solves"]) == [2, 1, 0, 1]
    assert np.abs(r["impulses"][2]).max() == 0.0 and r["impulses"][3, :, 0].sum() > 0
    s1 = r["state"].reshape(B, nbx, 13)[1, 0]                                       # no slip: the lowest box's contact points stand still
    for i in range(4):
        p = cs["point"][1, i]
        assert np.abs((s1[7:10] + np.cross(s1[10:13], p - s1[0:3]))[[0, 2]]).max() < 1e-12      # tangential (x, z); the normal one may separate
    n = I.lcp_size(8, 4)
    for w in range(B):
        aux = S.new_aux(1); s = st[w].copy()
        imp, _ = oracle.impact_process(nbx, mass, J, s, cs[w], aux, np.zeros(n), np.zeros(n), n)
        assert np.array_equal(s, r["state"][w]) and aux["status"][0] == r["status"][w], w
        assert np.array_equal(imp, r["impulses"][w]) and aux["lcp_solves"][0] == r["solves"][w] and aux["lcp_pivots"][0] == r["pivots"][w], w


@pytest.mark.parametrize("seed,nb,nc", [(20, 2, 3), (21, 3, 6), (22, 5, 9), (23, 6, 14)])
def test_random_no_slip_islands_match_oracle(oracle, seed, nb, nc):
    """Random contact graphs with mu-coulomb >= 100 on every contact: the greedy tangent-set selection (repeated Cholesky
    tests), the Schur-complement LCP, the warm-started _v, restitution -- two calls, bit for bit."""
    rng = np.random.default_rng(seed)
    B, nk = 4, 4
    n = I.lcp_size(nc, nk)
    mass = rng.uniform(0.5, 3.0, nb); J = rng.uniform(0.2, 2.0, (nb, 3))
    cs = np.stack([random_forest(rng, nb, nc) if nb >= 4 else random_island(rng, nb, nc) for _ in range(B)]); cs["nk"] = nk
    cs["mu_coulomb"] = rng.uniform(100.0, 500.0, (B, nc))
    cs["mu_coulomb"][0, :nc // 2] = rng.uniform(0.0, 1.0, nc // 2)        # world 0: mixed, so some islands take Drumwright-Shell
    st = np.zeros((B, nb, 13)); st[:, :, 0:3] = rng.standard_normal((B, nb, 3))
    q = rng.standard_normal((B, nb, 4)); st[:, :, 3:7] = q / np.linalg.norm(q, axis=2)[:, :, None]
    st[:, :, 7:13] = rng.standard_normal((B, nb, 6))
    st = st.reshape(B, -1)
    ib = I.ImpactBatch(B, nb, nc, nk, mass, J)
    aux = S.new_aux(B); zl = np.zeros((B, n)); zb = np.zeros((B, n))
    st_o = st.copy(); st_g = st.copy()
    for call in range(2):
        r = ib.process(st_g, cs)
        imp_o, piv_o, sol_o = oracle_batch(oracle, nb, mass, J, st_o, cs, n, aux, zl, zb)
        assert_same(r, st_o, imp_o, piv_o, sol_o, aux)
        st_g = r["state"].copy()
        st_g.reshape(B, nb, 13)[:, :, 7:13] += 0.1 * rng.standard_normal((B, nb, 6)); st_o[:] = st_g
    assert (r["solves"] >= 1).any()
    ib.close()


def random_forest(rng, nb, nc):
    """Contacts over nb bodies that fall into several islands: bodies are split into groups, every group gets a random
    connected multigraph, the lists are interleaved."""
    ngroups = int(rng.integers(2, 4))
    groups = np.array_split(rng.permutation(nb), ngroups)
    per = np.maximum(1, np.diff(np.linspace(0, nc, ngroups + 1).astype(int)))
    parts = []
    for g, k in zip(groups, per):
        c = random_island(rng, len(g), int(k), static_frac=1.0 if len(g) == 1 else 0.4)
        for f in ("body1", "body2"):
            c[f] = [int(g[v]) if v < len(g) else nb for v in c[f]]
        parts.append(c)
    cs = np.concatenate(parts)[:nc]
    if len(cs) < nc:
        cs = np.concatenate([cs, cs[:nc - len(cs)]])
    return cs[rng.permutation(nc)]


@pytest.mark.parametrize("seed,nb,nc,nk", [(10, 4, 6, 4), (11, 6, 9, 4), (12, 7, 12, 6), (13, 9, 14, 4)])
def test_random_multi_island_graphs_match_oracle(oracle, seed, nb, nc, nk):
    """Random contact lists over several islands of different sizes (one-body islands against static geometry included),
    two calls: island order, per-island LCP sizes, the _zlast / _z hand-over from one island's solve to the next."""
    rng = np.random.default_rng(seed)
    B = 4
    n = I.lcp_size(nc, nk)
    mass = rng.uniform(0.5, 3.0, nb); J = rng.uniform(0.2, 2.0, (nb, 3))
    cs = np.stack([random_forest(rng, nb, nc) for _ in range(B)]); cs["nk"] = nk
    st = np.zeros((B, nb, 13)); st[:, :, 0:3] = rng.standard_normal((B, nb, 3))
    q = rng.standard_normal((B, nb, 4)); st[:, :, 3:7] = q / np.linalg.norm(q, axis=2)[:, :, None]
    st[:, :, 7:13] = rng.standard_normal((B, nb, 6))
    st = st.reshape(B, -1)
    ib = I.ImpactBatch(B, nb, nc, nk, mass, J)
    aux = S.new_aux(B); zl = np.zeros((B, n)); zb = np.zeros((B, n))
    st_o = st.copy(); st_g = st.copy()
    most = 0
    for call in range(2):
        r = ib.process(st_g, cs)
        imp_o, piv_o, sol_o = oracle_batch(oracle, nb, mass, J, st_o, cs, n, aux, zl, zb)
        assert_same(r, st_o, imp_o, piv_o, sol_o, aux)
        most = max(most, int(r["solves"].max()))
        st_g = r["state"].copy()
        st_g.reshape(B, nb, 13)[:, :, 7:13] += 0.1 * rng.standard_normal((B, nb, 6)); st_o[:] = st_g
    assert most >= 2                                       # some world really had several active islands
    ib.close()


def test_every_batch_type_lives_on_the_device_current_at_create():
    """Device affinity (include/moby_hip.h, Devices): a batch belongs to the device that was current when it was created.  mh_impact_batch_create used to zero the
    field after reading it, so the query is held here for every batch type -- on the one device of the test box (a second GPU is what it takes
    to see the difference; the multi-GPU example checks placement there)."""
    from moby_amd import _lib, scene as S, stack as K, artic as A
    from moby_amd.world import WorldBatchDevice
    import os
    lib = _lib.load()
    dev = lib.mh_device_get()
    assert dev >= 0
    mass, J, st, cs = I.box_stack(2, B=2)
    ib = I.ImpactBatch(2, 2, 8, 4, mass, J)
    assert lib.mh_impact_batch_device(ib.handle) == dev
    ib.close()
    wb = WorldBatchDevice(S.sphere_stack_scene(), S.sphere_stack_state(2))
    assert lib.mh_world_batch_device(wb.handle) == dev
    wb.close()
    bb = K.BigBatch(K.box_stack_scene(2), K.box_stack_state(2, 2))
    assert lib.mh_big_batch_device(bb.handle) == dev
    bb.close()
    m, _, _ = A.load_sdf(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes", "ten_joint_arm.sdf"))
    ab = A.ArticBatch(m, np.zeros((2, m.nj)), np.zeros((2, m.nj)))
    assert lib.mh_artic_batch_device(ab.handle) == dev
    ab.close()


def test_upload_rejects_malformed_contacts():
    lib = _lib.load()
    mass, J, st, cs = I.box_stack(2, B=1)
    ib = I.ImpactBatch(1, 2, 8, 4, mass, J)
    for field, value in (("nk", 6), ("body2", 0)):
        bad = cs.copy(); bad[field][0, 0] = value        # nk differs from the batch's; body1 == body2
        assert lib.mh_impact_batch_upload(ib.handle, st.ctypes.data, bad.ctypes.data) == _lib.MH_ERR_INVALID_ARG
    bad = cs.copy(); bad["normal"][0, 3] = (0.0, 0.0, 0.0)
    assert lib.mh_impact_batch_upload(ib.handle, st.ctypes.data, bad.ctypes.data) == _lib.MH_ERR_INVALID_ARG
    bad = cs.copy(); bad["body1"][0, 0] = -1                # body2 of contact 0 is the ground already: static-static
    assert lib.mh_impact_batch_upload(ib.handle, st.ctypes.data, bad.ctypes.data) == _lib.MH_ERR_INVALID_ARG
    ib.close()


def test_host_convenience_entry_equals_object_api():
    lib = _lib.load()
    B, nbx = 3, 3
    mass, J, st, cs = I.box_stack(nbx, B=B)
    r = I.ImpactBatch(B, nbx, 12, 4, mass, J).process(st, cs)
    s2 = st.copy(); imp = np.zeros((B, 12, 3)); status = np.zeros(B, dtype=np.int32); piv = np.zeros(B, dtype=np.uint32)
    sol = np.zeros(B, dtype=np.int32)
    _lib.check(lib.mh_impact_process_batch(B, nbx, 12, 4, mass.ctypes.data, J.ctypes.data, s2.ctypes.data, cs.ctypes.data,
                                           imp.ctypes.data, status.ctypes.data, piv.ctypes.data, sol.ctypes.data))
    assert np.array_equal(s2, r["state"]) and np.array_equal(imp, r["impulses"]) and np.array_equal(piv, r["pivots"])


def test_tall_stack_properties_at_scale():
    """64 worlds x 8 boxes (n = 256, block solver, Lemke fallback): every box at rest after the impact, normal impulses
    of the ground interface carry the stack's momentum, identical worlds give identical results in any slot."""
    nbx, B = 8, 64
    mass, J, st, cs = I.box_stack(nbx, B=B, mu=0.0, perturb=False)
    r = I.ImpactBatch(B, nbx, 32, 4, mass, J).process(st, cs)
    assert (r["status"] == 0).all() and (r["solves"] == 1).all()
    assert np.abs(r["state"].reshape(B, nbx, 13)[:, :, 7:13]).max() < 1e-9
    np.testing.assert_allclose(r["impulses"][:, :4, 0].sum(axis=1), 9.81e-3 * mass.sum(), rtol=1e-8)
    assert (r["state"] == r["state"][0]).all() and (r["pivots"] == r["pivots"][0]).all()


def test_cpp_impact_handler_adapter_example():
    """moby_amd/cpp/MobyHipImpactHandler.h (the process_constraints-shaped C++ adapter, plain g++): a 3-box stack comes
    to rest, the ground contacts carry the stack's momentum, both worlds of the batch agree."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cpp = os.path.join(root, "moby_amd", "cpp")
    exe = os.path.join(cpp, "example_impact")
    subprocess.check_call(["g++", "-std=c++11", os.path.join(cpp, "example_impact.cpp"), "-L" + os.path.join(root, "moby_amd"),
                           "-lmoby_hip", "-Wl,-rpath," + os.path.join(root, "moby_amd"), "-o", exe])
    out = subprocess.check_output([exe, "3"]).decode()
    assert "n=96 status=0 solves=1" in out and "same=1" in out
    assert float(out.split("vmax=")[1].split()[0]) < 1e-9
    assert abs(float(out.split("weight_dt=")[1].split()[0]) - 1.0) < 1e-9


def test_twelve_box_stack_matches_oracle(oracle):
    """n = 384: the 1024-thread block solver; in the perturbed world lcp_fast fails on all four rungs and the Lemke ladder solves it --
    the regime of BASELINE config 4 -- bit for bit against the oracle (0.8 s per world on the CPU)."""
    nbx, B = 12, 2
    mass, J, st, cs = I.box_stack(nbx, B=B)
    nc = 4 * nbx; n = I.lcp_size(nc, 4)
    ib = I.ImpactBatch(B, nbx, nc, 4, mass, J)
    aux = S.new_aux(B); zl = np.zeros((B, n)); zb = np.zeros((B, n))
    st_o = st.copy()
    r = ib.process(st, cs)
    imp_o, piv_o, sol_o = oracle_batch(oracle, nbx, mass, J, st_o, cs, n, aux, zl, zb)
    assert_same(r, st_o, imp_o, piv_o, sol_o, aux)
    assert r["pivots"].max() > 4 * n and (r["status"] == 0).all()      # world 1: all four lcp_fast rungs fail, Lemke solves
    ib.close()


def test_solver_state_checkpoint_resumes_bit_exact():
    """_zlast and the rand() streams of a handler saved and loaded into a new batch: the next call equals the call an
    uninterrupted handler makes; a fresh handler (what the reference's pickle amounts to) pivots differently."""
    nbx, B = 3, 4
    mass, J, st, cs = I.box_stack(nbx, B=B, mu=0.3)
    a = I.ImpactBatch(B, nbx, 12, 4, mass, J)
    r1 = a.process(st, cs)
    s2 = r1["state"].copy(); s2.reshape(B, nbx, 13)[:, :, 8] += -9.81e-3
    ss = a.solver_state()
    assert (ss["zlast_size"] == a.n).all()
    r2 = a.process(s2, cs)
    b = I.ImpactBatch(B, nbx, 12, 4, mass, J); b.load_solver_state(ss)
    r2b = b.process(s2, cs)
    assert np.array_equal(r2b["state"], r2["state"]) and np.array_equal(r2b["pivots"], r2["pivots"]) and np.array_equal(r2b["impulses"], r2["impulses"])
    sb = b.solver_state(); sa = a.solver_state()
    assert np.array_equal(sb["rng"], sa["rng"]) and np.array_equal(sb["zlast"], sa["zlast"])
    fresh = I.ImpactBatch(B, nbx, 12, 4, mass, J).process(s2, cs)
    assert not np.array_equal(fresh["pivots"], r2["pivots"])
    bad = dict(ss); bad["zlast_size"] = ss["zlast_size"] + 1
    with pytest.raises(_lib.MobyHipError):
        b.load_solver_state(bad)


def test_noslip_state_checkpoint_resumes_bit_exact():
    """The no-slip model's warm start _v (ICH:1239) travels with the checkpoint: a handler restored from it makes the same
    second call (pivots included) as the uninterrupted one; without _v the restored handler starts lcp_fast cold."""
    nbx, B = 3, 4
    mass, J, st, cs = I.box_stack(nbx, B=B, mu=200.0)              # every mu >= 100: the no-slip model
    a = I.ImpactBatch(B, nbx, 12, 4, mass, J)
    r1 = a.process(st, cs)
    s2 = r1["state"].copy(); s2.reshape(B, nbx, 13)[:, :, 8] += -9.81e-3
    ss = a.solver_state()
    assert (ss["v_size"] == 12).all()
    r2 = a.process(s2, cs)
    b = I.ImpactBatch(B, nbx, 12, 4, mass, J); b.load_solver_state(ss)
    r2b = b.process(s2, cs)
    assert np.array_equal(r2b["state"], r2["state"]) and np.array_equal(r2b["pivots"], r2["pivots"]) and np.array_equal(r2b["impulses"], r2["impulses"])
    without = {k: v for k, v in ss.items() if k not in ("v", "v_size")}
    c = I.ImpactBatch(B, nbx, 12, 4, mass, J); c.load_solver_state(without)
    r2c = c.process(s2, cs)
    assert not np.array_equal(r2c["pivots"], r2["pivots"])
    for h in (a, b, c):
        h.close()


# ---- Anitescu-Potra model (a batch in the reference's -DUSE_AP configuration) ------------------------------------------
@pytest.fixture
def ap_oracle(oracle):
    oracle.set_impact_model(I.MH_IMPACT_MODEL_AP)
    yield oracle
    oracle.set_impact_model(I.MH_IMPACT_MODEL_DS)


@pytest.mark.parametrize("nbx,B,eps,mu,nk", [(1, 4, 0.0, 0.3, 4), (2, 4, 0.0, 1e-4, 4), (3, 4, 0.4, 0.5, 4), (2, 4, 0.0, 0.3, 8),
                                             (4, 3, 0.3, 0.5, 6), (3, 3, 0.0, 0.2, 16)])
def test_ap_box_stacks_match_oracle(ap_oracle, nbx, B, eps, mu, nk):
    """apply_ap_model (ICH-AP:94-370) on box stacks: n = 5 nc + nc NK_DIRS rows (wave solver up to 64, block solver above),
    the Lemke ladder (-20, 1, -2) on a fresh z, restitution with and without the second solve, velocities through the
    contacts' accumulated wrenches -- states, impulses, pivot counts and rand() streams bit for bit, three calls."""
    mass, J, st, cs = I.box_stack(nbx, B=B, epsilon=eps, mu=mu, nk=nk)
    nc = 4 * nbx; n = I.lcp_size(nc, nk)
    ib = I.ImpactBatch(B, nbx, nc, nk, mass, J, model=I.MH_IMPACT_MODEL_AP)
    aux = S.new_aux(B); zl = np.zeros((B, n)); zb = np.zeros((B, n))
    st_g = st.copy(); st_o = st.copy()
    for call in range(3):
        r = ib.process(st_g, cs)
        if call == 0:
            MMg, qqg = ib.debug_lcp()
            na = I.ap_lcp_size(nc, nk)
            for w in range(min(B, 2)):
                nn, MMo, qqo = ap_oracle.impact_lcp(nbx, mass, J, st[w].copy(), cs[w], n)
                assert nn == na
                if eps == 0.0:         # (with restitution the device's _qq is the second solve's)
                    Mw = MMg[w].T.reshape(-1)[:na * na].reshape(na, na).T
                    assert np.array_equal(MMo, Mw) and np.array_equal(qqo, qqg[w][:na])
        imp_o, piv_o, sol_o = oracle_batch(ap_oracle, nbx, mass, J, st_o, cs, n, aux, zl, zb)
        assert_same(r, st_o, imp_o, piv_o, sol_o, aux)
        assert (aux["lcp_rows"] > 0).all()
        st_g = r["state"].copy()
        for a in (st_g, st_o):
            a.reshape(B, nbx, 13)[:, :, 8] += -9.81e-3
    assert (aux["zlast_size"] == 0).all()                  # the A-P path never touches the QP solver's _zlast
    ib.close()


@pytest.mark.parametrize("seed,nb,nc,nk", [(20, 4, 6, 4), (21, 6, 9, 8), (22, 7, 12, 4), (23, 3, 10, 12)])
def test_ap_random_multi_island_graphs_match_oracle(ap_oracle, seed, nb, nc, nk):
    """Random multi-island contact lists (mixed friction, restitution, static sides, rotated bodies) through the A-P model."""
    rng = np.random.default_rng(seed)
    B = 4
    n = I.lcp_size(nc, nk)
    mass = rng.uniform(0.5, 3.0, nb); J = rng.uniform(0.2, 2.0, (nb, 3))
    cs = np.stack([random_forest(rng, nb, nc) for _ in range(B)]); cs["nk"] = nk
    st = np.zeros((B, nb, 13)); st[:, :, 0:3] = rng.standard_normal((B, nb, 3))
    q = rng.standard_normal((B, nb, 4)); st[:, :, 3:7] = q / np.linalg.norm(q, axis=2)[:, :, None]
    st[:, :, 7:13] = rng.standard_normal((B, nb, 6))
    st = st.reshape(B, -1)
    ib = I.ImpactBatch(B, nb, nc, nk, mass, J, model=I.MH_IMPACT_MODEL_AP)
    aux = S.new_aux(B); zl = np.zeros((B, n)); zb = np.zeros((B, n))
    st_o = st.copy(); st_g = st.copy()
    for call in range(2):
        r = ib.process(st_g, cs)
        imp_o, piv_o, sol_o = oracle_batch(ap_oracle, nb, mass, J, st_o, cs, n, aux, zl, zb)
        assert_same(r, st_o, imp_o, piv_o, sol_o, aux)
        st_g = r["state"].copy()
        st_g.reshape(B, nb, 13)[:, :, 7:13] += 0.1 * rng.standard_normal((B, nb, 6)); st_o[:] = st_g
    assert (r["solves"] >= 1).any()
    ib.close()


@pytest.mark.parametrize("tasks", [0, 1, 2])
def test_ap_ladder_trivial_exit_draws_nothing(ap_oracle, tasks):
    """lcp_lemke's trivial exit (LCP.cpp:578, min(q) > -zero_tol) returns BEFORE _restart_z0's n rand() draws (:618-620).  Under the
    A-P model the ladder is entered cold (z.size() = 0 != n), so a separating island must leave its world's rand() stream where it
    was -- in sequence (mh_debug_set(4, 0)) and as (world, attempt) tasks (1, 2) alike.  n = 72 (block solver); worlds 0-1
    separate (q >= 0), world 2 falls; rand() streams compared word for word with the oracle's."""
    lib = _lib.load()
    nbx, B, nk = 3, 3, 4
    mass, J, st, cs = I.box_stack(nbx, B=B, epsilon=0.0, mu=0.3, nk=nk)
    st.reshape(B, nbx, 13)[:2, :, 8] = 0.25 + 0.25 * np.arange(nbx)      # separating: every relative normal velocity positive
    nc = 4 * nbx; n = I.lcp_size(nc, nk)
    assert I.ap_lcp_size(nc, nk) > 64
    try:
        _lib.check(lib.mh_debug_set(4, tasks))
        ib = I.ImpactBatch(B, nbx, nc, nk, mass, J, model=I.MH_IMPACT_MODEL_AP)
        aux = S.new_aux(B); zl = np.zeros((B, n)); zb = np.zeros((B, n))
        rng0 = aux["rng"].copy()
        st_o = st.copy()
        for call in range(2):
            r = ib.process(st_o.copy(), cs)
            imp_o, piv_o, sol_o = oracle_batch(ap_oracle, nbx, mass, J, st_o, cs, n, aux, zl, zb)
            assert_same(r, st_o, imp_o, piv_o, sol_o, aux)
            ss = ib.solver_state()
            assert np.array_equal(ss["rng"], aux["rng"])
            assert np.array_equal(aux["rng"][:2], rng0[:2]) and (piv_o[:2] == 0).all()      # the trivial exit: no pivots, no draws
            assert not np.array_equal(aux["rng"][2], rng0[2])
        ib.close()
    finally:
        _lib.check(lib.mh_debug_set(4, 3))


def test_ap_model_argument_check():
    mass, J, st, cs = I.box_stack(1, B=1)
    ib = I.ImpactBatch(1, 1, 4, 4, mass, J)
    lib = _lib.load()
    assert lib.mh_impact_batch_set_model(ib.handle, 7) == _lib.MH_ERR_INVALID_ARG
    assert lib.mh_impact_batch_set_model(ib.handle, I.MH_IMPACT_MODEL_AP) == 0
    ib.close()
