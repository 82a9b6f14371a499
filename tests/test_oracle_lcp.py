"""CPU tests of the oracle's LCP restatement (oracle/lcp.hpp) -- pins the
checker before it is trusted (no GPU needed)."""
import ctypes
import numpy as np
import pytest

from moby_amd import synth
from tests.oracle_api import FAST, FAST_REG, LEMKE, LEMKE_REG, DEFAULT_EXPS

KINDS = [FAST, FAST_REG, LEMKE, LEMKE_REG]


def comp_residual(M, q, z):
    w = M @ z + q
    return min(z.min(), w.min()), np.abs(z * w).max()


def test_rand_matches_libc(oracle):
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 2, 12345):
        libc.srand(seed)
        st = oracle.rand_state(seed)
        assert [libc.rand() for _ in range(2000)] == [oracle.rand_next(st) for _ in range(2000)]


def test_lu_matches_lapack(oracle):
    from scipy.linalg import lapack
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 7, 20, 42):
        A = rng.standard_normal((n, n)); b = rng.standard_normal(n)
        info, x = oracle.lu_solve(A, b)
        _, _, xr, info_r = lapack.dgesv(A, b)
        assert info == 0 and info_r == 0
        np.testing.assert_allclose(x, xr, rtol=1e-10, atol=1e-12)
    # exact singularity is reported where LAPACK reports it
    A = np.array([[1.0, 2.0], [2.0, 4.0]])
    info, _ = oracle.lu_solve(A, np.ones(2))
    assert info == lapack.dgesv(A, np.ones(2))[3] == 2
    info, _ = oracle.lu_solve(np.zeros((3, 3)), np.ones(3))
    assert info == 1


@pytest.mark.parametrize("kind", KINDS)
def test_kat_1x1(oracle, kind):
    # w = 2 z - 4  ->  z = 2
    r = oracle.lcp(kind, [[2.0]], [-4.0], z_size=0)
    assert r["ok"] and r["z"][0] == 2.0
    # q >= 0 -> trivial z = 0
    r = oracle.lcp(kind, [[2.0]], [3.0], z_size=0)
    assert r["ok"] and r["z"][0] == 0.0 and r["pivots"] == 0


@pytest.mark.parametrize("kind", KINDS)
def test_kat_2x2_known_basis(oracle, kind):
    # M = [[2,1],[1,2]], q = [-5,-6]: both nonbasic, z = M^-1 (5,6) = (4/3, 7/3)
    r = oracle.lcp(kind, [[2.0, 1.0], [1.0, 2.0]], [-5.0, -6.0], z_size=0)
    assert r["ok"]
    np.testing.assert_allclose(r["z"], [4.0 / 3.0, 7.0 / 3.0], rtol=1e-15)
    # q = [-1, 5]: only z0 = 1/2 active, w1 = 5.5
    r = oracle.lcp(kind, [[2.0, 1.0], [1.0, 2.0]], [-1.0, 5.0], z_size=0)
    assert r["ok"]
    np.testing.assert_allclose(r["z"], [0.5, 0.0], rtol=1e-15)


@pytest.mark.parametrize("kind", KINDS)
def test_kat_sphere_stack_normal_lcp(oracle, kind):
    """SURVEY 8c KAT: Cn X Cn' = (1/m) tridiag, first impact at v = -g dt:
    cn = m g dt [3, 2, 1]."""
    g, dt = 9.81, 1e-3
    M = np.array([[1.0, -1, 0], [-1, 2, -1], [0, -1, 2]])
    q = np.array([-g * dt, 0.0, 0.0])
    r = oracle.lcp(kind, M, q, z_size=0)
    assert r["ok"]
    np.testing.assert_allclose(r["z"], g * dt * np.array([3.0, 2.0, 1.0]), rtol=1e-13)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n", [1, 2, 5, 16, 42, 64])
def test_random_pd_solves(oracle, kind, n):
    Ms, qs = synth.random_lcp(4, n, "pd", seed=n)
    for M, q in zip(Ms, qs):
        r = oracle.lcp(kind, M, q, z_size=0)
        if kind in (FAST, FAST_REG) and not r["ok"]:
            # the principal-pivoting heuristic may give up after 2n pivots
            # (LCP.cpp:107,192-195); the handlers then fall back to Lemke
            assert r["pivots"] >= 2 * n
            continue
        assert r["ok"]
        # a solution found on the regularisation ladder is only a solution of
        # M + lambda I (LCP.cpp:303-312 verifies against _MM, not M)
        marks = [t & 0xFFFF for t in r["trace"] if t & 0x40000000 and t > 0]
        k = marks[-1] if marks else 0
        e = DEFAULT_EXPS[kind]
        lam = 0.0 if k == 0 else 10.0 ** (e[0] + (k - 1) * e[1])
        Mreg = M + lam * np.eye(n)
        lo, comp = comp_residual(Mreg, q, r["z"])
        scale = n * np.abs(Mreg).max()
        assert lo > -1e-7 * scale and comp < 1e-7 * scale


def test_sphere_stack_impact_lcp_42(oracle):
    """The 42x42 _MM/_qq of the reference scene: lcp_fast_regularized(-20,4,-8)
    (ICH-QP:219) must return cn = m g dt [3,2,1] and zero friction."""
    Ms, qs = synth.sphere_stack_impact_lcp(3)
    assert Ms.shape == (3, 42, 42)
    r = oracle.lcp(FAST_REG, Ms[0], qs[0], z_size=42, exps=(-20, 4, -8))
    assert r["ok"]
    np.testing.assert_allclose(r["z"][:3], 9.81e-3 * np.array([3.0, 2.0, 1.0]), rtol=1e-12)
    assert np.abs(r["z"][3:15]).max() < 1e-12
    for w in (1, 2):
        # the handler's chain (ICH-QP:219-225): fast ladder, else Lemke ladder
        r = oracle.lcp(FAST_REG, Ms[w], qs[w], z_size=42, exps=(-20, 4, -8))
        if not r["ok"]:
            r = oracle.lcp(LEMKE_REG, Ms[w], qs[w], z_size=42, rng=r["rng"])
        assert r["ok"]
        lo, comp = comp_residual(Ms[w], qs[w], r["z"])
        assert lo > -1e-9 and comp < 1e-9


def test_warm_start_is_used(oracle):
    Ms, qs = synth.random_lcp(1, 20, "pd", seed=5)
    cold = oracle.lcp(FAST, Ms[0], qs[0], z_size=0)
    warm = oracle.lcp(FAST, Ms[0], qs[0], z=cold["z"], z_size=20)
    assert cold["ok"] and warm["ok"]
    assert warm["pivots"] == 0 and cold["pivots"] > 0
    np.testing.assert_array_equal(cold["z"], warm["z"])


def test_lemke_consumes_rand_iff_size_mismatch(oracle):
    """LCP.cpp:611-621: z.size() != n draws n rand() values."""
    Ms, qs = synth.random_lcp(1, 6, "pd", seed=9)
    st0 = oracle.rand_state(1)
    a = oracle.lcp(LEMKE, Ms[0], qs[0], z_size=6, rng=st0)
    b = oracle.lcp(LEMKE, Ms[0], qs[0], z_size=0, rng=st0)
    np.testing.assert_array_equal(a["z"], b["z"])
    np.testing.assert_array_equal(a["rng"], st0)          # untouched
    adv = st0.copy()
    for _ in range(6):
        oracle.rand_next(adv)
    np.testing.assert_array_equal(b["rng"], adv)


def test_fast_singular_falls_to_ladder(oracle):
    """Structurally singular M (zero block) makes lcp_fast fail on a zero pivot;
    the regularised wrapper recovers (LCP.cpp:122-126, 281-340)."""
    M = np.array([[0.0, 0.0], [0.0, 1.0]]); q = np.array([-1.0, -1.0])
    r = oracle.lcp(FAST, M, q, z_size=0)
    assert not r["ok"]
    r = oracle.lcp(FAST_REG, M, q, z_size=0, exps=(-20, 4, 20))
    # first ladder rung (lambda = 1e-20): z0 = 1/lambda
    assert r["ok"] and r["trace_len"] > 1
    np.testing.assert_allclose(r["z"], [1e20, 1.0], rtol=1e-15)


def test_lemke_keeps_a_nan_ratio_that_comes_first(oracle):
    """LCP.cpp:920 takes theta = *std::min_element(ratios): a NaN that comes FIRST stays the minimum, every `ratio <= theta` test is
    then false and the attempt fails with an empty candidate set (:946-958).  tests/golden/lemke_ladder_n96_case.npz reaches that
    state on the rung lambda = 1e-13 after 453 pivots (the whole ladder: 1518 pivots, solved on a later rung)."""
    import ctypes, os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lemke_ladder_n96_case.npz"))
    MM, qq, rng = d["MM"], d["qq"], d["rng"]; n = len(qq)
    lam = MM.copy(); lam[np.arange(n), np.arange(n)] += 10.0 ** -13
    r = oracle.lcp(LEMKE, lam, qq, z=np.zeros(n), z_size=n, rng=rng)
    assert r["ok"] is False and r["pivots"] == 453 and oracle.lib.oracle_dbg_lemke_exit() == 3
    full = oracle.lcp(LEMKE_REG, MM, qq, z=np.zeros(n), z_size=n, rng=rng, trace_cap=8192)
    assert full["ok"] and full["pivots"] == 1518
