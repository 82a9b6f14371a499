"""The reference's own gtest property tests (test/*.cpp, not wired into its CMake: SURVEY 4), re-run on the HIP path.

  test/TestDie.cpp:34-135          100 random box drops, penetration > -1e-6 until the box has had no kinetic energy for 0.5 s
  test/TestSparseJacobian.cpp:13-223  block-sparse Jacobian products equal dense products within 1e-6 -- here: the
                                   C X C' blocks and the assembled LCP matrix of the impact entry against dense numpy
                                   products J X J' (the role SparseJacobian::mult / mult_transpose play in
                                   ImpactConstraintHandler::compute_problem_data, ICH:2125-2149)
"""
import numpy as np
import pytest

from moby_amd import impact as I
from moby_amd import scene as S
from moby_amd.world import WorldBatch

pytestmark = pytest.mark.gpu


def glibc_rand_stream(n, seed=1):
    st = S.glibc_srand_state(seed); out = []
    idx = int(st[31]); r = [int(x) for x in st[:31]]
    for _ in range(n):
        v = (r[idx] + r[(idx + 28) % 31]) & 0xFFFFFFFF
        r[idx] = v; idx = (idx + 1) % 31
        out.append(v >> 1)
    return np.array(out, dtype=np.float64) / 2147483647.0


def test_die_random_drops_never_penetrate():
    """TestDie (Boxes.ConstraintViolation): test/box.xml -- a unit box of density 1 released at (0, 1, 0) over the plane
    y = -5e-6, mu = 1, NK = 4, epsilon = 0 -- with a random orientation and velocity drawn from rand() as the test
    draws them, dt = 0.01, min_step_size = 0.1 (so a step is never cut: impacts penetrate and stabilisation has to
    repair them), cstab.eps = -NEAR_ZERO.  All 100 drops step as one batch; the minimum pairwise distance after every
    step must stay above -1e-6 until each box has rested (KE < 1e-6) for half a second."""
    B, DT, TOL = 100, 1e-2, 1e-6
    sc = S.box_scene(mu_coulomb=1.0, epsilon=0.0, nk=4, cstab_max_iterations=1000)
    sc.plane_o[1] = -0.000005
    sc.min_step_size = 1e-1
    sc.cstab_eps = -S.NEAR_ZERO
    u = glibc_rand_stream(10 * B).reshape(B, 10) * 2.0 - 1.0          # q.x q.y q.z q.w, then xd[i], w[i] interleaved (TestDie.cpp:73-92)
    st = np.zeros((B, 13)); st[:, 1] = 1.0
    q = u[:, 0:4]; st[:, 3:7] = q / np.linalg.norm(q, axis=1)[:, None]  # state order x y z w = the test's q.x q.y q.z q.w
    st[:, 7:10] = u[:, 4::2]; st[:, 10:13] = u[:, 5::2]
    wb = WorldBatch(sc, st)
    m = sc.mass[0]; J = np.array([sc.inertia[0][k] for k in range(3)])
    half = 0.5 * np.array([sc.geom_dim[0][k] for k in range(3)])
    corners = np.array([[sx, sy, sz] for sx in (1, -1) for sy in (1, -1) for sz in (1, -1)]) * half
    rest_since = np.zeros(B); done = np.zeros(B, dtype=bool); max_vio = np.zeros(B)
    t = 0.0
    for step in range(1500):
        wb.step(DT, 1); t += DT
        s = wb.state
        x, y, z, w = s[:, 3], s[:, 4], s[:, 5], s[:, 6]
        R = np.stack([1 - 2 * (y*y + z*z), 2 * (x*y - z*w), 2 * (x*z + y*w), 2 * (x*y + z*w), 1 - 2 * (x*x + z*z), 2 * (y*z - x*w),
                      2 * (x*z - y*w), 2 * (y*z + x*w), 1 - 2 * (x*x + y*y)], axis=1).reshape(B, 3, 3)
        height = s[:, None, 1] + np.einsum("bj,cj->bc", R[:, 1, :], corners) + 0.000005   # vertices over the plane
        live = ~done
        max_vio[live] = np.minimum(max_vio[live], height.min(axis=1)[live])
        wl = np.einsum("bji,bj->bi", R, s[:, 10:13])                                      # body-frame angular velocity
        ke = 0.5 * m * (s[:, 7:10] ** 2).sum(axis=1) + 0.5 * (J * wl * wl).sum(axis=1)
        rest_since = np.where(ke < 1e-6, rest_since, t)
        done |= (ke < 1e-6) & (t - rest_since > 0.5)
        if done.all():
            break
    assert done.sum() >= 0.95 * B, "%d of %d boxes came to rest within %.1f s" % (done.sum(), B, t)
    assert (max_vio > -TOL).all(), "worst penetration %.3e" % max_vio.min()
    assert ((wb.aux["status"] & ~(S.MH_WORLD_IMPACT_TOL | S.MH_WORLD_STAB_FAILED)) == 0).all()


@pytest.mark.parametrize("seed,nb,nc", [(1, 3, 7), (2, 6, 14), (3, 10, 24)])
def test_sparse_products_equal_dense_products(seed, nb, nc):
    """TestSparseJacobian's property on the product path: the LCP matrix the impact entry assembles from block-sparse rows
    ([d, r x d] per body per direction) equals the dense J X J' of the same contacts within 1e-6 (here: 1e-12)."""
    from tests.test_impact_gpu import random_island
    rng = np.random.default_rng(seed)
    nk = 4
    mass = rng.uniform(0.5, 3.0, nb); Jb = rng.uniform(0.2, 2.0, (nb, 3))
    cs = random_island(rng, nb, nc)[None]; cs["nk"] = nk
    st = np.zeros((1, nb, 13)); st[:, :, 0:3] = rng.standard_normal((1, nb, 3))
    q = rng.standard_normal((1, nb, 4)); st[:, :, 3:7] = q / np.linalg.norm(q, axis=2)[:, :, None]
    st[:, :, 7:13] = rng.standard_normal((1, nb, 6))
    ib = I.ImpactBatch(1, nb, nc, nk, mass, Jb)
    r = ib.process(st.reshape(1, -1).copy(), cs)
    if r["solves"][0] == 0:
        pytest.skip("nothing impacting in this draw")
    MM, qq = ib.debug_lcp()
    ib.close()
    # dense rebuild in the CALLER's contact order; the entry works in island order, so rows are matched by signature first
    X = np.zeros((6 * nb, 6 * nb))
    for b in range(nb):
        x, y, z, w = st[0, b, 3:7]
        R = np.array([[1 - 2 * (y*y + z*z), 2 * (x*y - z*w), 2 * (x*z + y*w)], [2 * (x*y + z*w), 1 - 2 * (x*x + z*z), 2 * (y*z - x*w)],
                      [2 * (x*z - y*w), 2 * (y*z + x*w), 1 - 2 * (x*x + y*y)]])
        X[6*b:6*b+3, 6*b:6*b+3] = np.eye(3) / mass[b]
        X[6*b+3:6*b+6, 6*b+3:6*b+6] = np.linalg.inv(R @ np.diag(Jb[b]) @ R.T)
    Jn = np.zeros((nc, 6 * nb))
    for i in range(nc):
        n = cs["normal"][0, i]; p = cs["point"][0, i]
        for body, sgn in ((cs["body1"][0, i], 1.0), (cs["body2"][0, i], -1.0)):
            if 0 <= body < nb:
                Jn[i, 6*body:6*body+3] = sgn * n; Jn[i, 6*body+3:6*body+6] = np.cross(p - st[0, body, 0:3], sgn * n)
    dense = Jn @ X @ Jn.T
    got = MM[0][:nc, :nc]                                       # Cn X Cn' block (plus compliance on its diagonal), island order
    # match rows by their diagonal + row-sum signature (contacts are distinct with probability 1)
    sig = lambda A: np.round(np.stack([np.diag(A), np.abs(A).sum(axis=1)], axis=1), 9)
    comp = cs["compliance"][0]
    dsig = sig(dense + np.diag(comp)); gsig = sig(got)
    perm = [int(np.argmin(np.abs(dsig - g).sum(axis=1))) for g in gsig]
    assert sorted(perm) == list(range(nc))
    np.testing.assert_allclose(got, (dense + np.diag(comp))[np.ix_(perm, perm)], atol=1e-12 * max(1.0, np.abs(dense).max()))
