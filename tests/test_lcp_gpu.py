"""GPU parity: the HIP LCP solvers (through the C ABI) against the CPU oracle,
bit-exact on status / pivots / pivot trace / rand() state / z."""
import os

import numpy as np
import pytest

from moby_amd import lcp as L
from moby_amd import synth
from moby_amd.lcp import LCP
from tests.oracle_api import FAST, FAST_REG, LEMKE, LEMKE_REG, DEFAULT_EXPS

pytestmark = pytest.mark.gpu

TRACE_CAP = 2048


def run_gpu(kind, M, q, z0=None, z_size=None, exps=None, seed=1):
    B, n = q.shape
    lcp = LCP(B, seed=seed)
    z = np.zeros((B, n)) if z0 is None else np.array(z0, dtype=np.float64)
    e = exps if exps is not None else DEFAULT_EXPS[kind]
    if kind == FAST:
        ok = lcp.lcp_fast(M, q, z, z_size=z_size, trace_cap=TRACE_CAP)
    elif kind == FAST_REG:
        ok = lcp.lcp_fast_regularized(M, q, z, *e, z_size=z_size, trace_cap=TRACE_CAP)
    elif kind == LEMKE:
        ok = lcp.lcp_lemke(M, q, z, z_size=z_size, trace_cap=TRACE_CAP)
    else:
        ok = lcp.lcp_lemke_regularized(M, q, z, *e, z_size=z_size, trace_cap=TRACE_CAP)
    return ok, z, lcp


def assert_parity(oracle, kind, M, q, z0=None, z_size=None, exps=None, seed=1):
    B, n = q.shape
    ok, z, lcp = run_gpu(kind, M, q, z0, z_size, exps, seed)
    for b in range(B):
        zs = n if z_size is None else int(z_size[b])
        r = oracle.lcp(kind, M[b], q[b], z=None if z0 is None else z0[b], z_size=zs,
                       rng=oracle.rand_state(seed), exps=exps, trace_cap=TRACE_CAP)
        tag = "kind %d n %d problem %d" % (kind, n, b)
        assert bool(ok[b]) == r["ok"], tag
        assert int(lcp.pivots[b]) == r["pivots"], tag
        assert int(lcp.trace_len[b]) == r["trace_len"], tag
        L = min(r["trace_len"], TRACE_CAP)
        np.testing.assert_array_equal(lcp.trace[b, :L], r["trace"][:L], err_msg=tag)
        np.testing.assert_array_equal(lcp.rng[b], r["rng"], err_msg=tag)
        assert int(lcp.z_size[b]) == r["z_size"], tag
        if r["ok"]:
            np.testing.assert_array_equal(z[b], r["z"], err_msg=tag)   # bit-exact (-0 == 0)
    return ok


@pytest.mark.parametrize("kind", [FAST, FAST_REG, LEMKE, LEMKE_REG])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 17, 42, 64])
def test_random_pd_cold(oracle, kind, n):
    M, q = synth.random_lcp(6, n, "pd", seed=100 + n)
    assert_parity(oracle, kind, M, q, z_size=np.zeros(6, dtype=np.int32))


@pytest.mark.parametrize("kind", [FAST, FAST_REG, LEMKE, LEMKE_REG])
@pytest.mark.parametrize("fam", ["psd", "copos"])
@pytest.mark.parametrize("n", [4, 10, 42])
def test_degenerate_families(oracle, kind, fam, n):
    M, q = synth.random_lcp(6, n, fam, seed=7 * n)
    assert_parity(oracle, kind, M, q, z_size=np.zeros(6, dtype=np.int32))


@pytest.mark.parametrize("kind", [FAST, FAST_REG])
def test_warm_start(oracle, kind):
    n = 24
    M, q = synth.random_lcp(8, n, "pd", seed=11)
    ok, z, _ = run_gpu(LEMKE, M, q)
    assert ok.all()
    # perturb q slightly and warm start from the previous solution
    q2 = q + 1e-3 * np.random.default_rng(1).standard_normal(q.shape)
    assert_parity(oracle, kind, M, q2, z0=z)


def test_sphere_stack_bundle(oracle):
    """The config-2 impact LCPs (n = 42): the handler's first call
    lcp_fast_regularized(-20,4,-8) (ICH-QP:219), then the Lemke ladder on the
    failures (ICH-QP:221-225) with the rand() stream carried over."""
    B = 64
    M, q = synth.sphere_stack_impact_lcp(B)
    ok = assert_parity(oracle, FAST_REG, M, q, z0=np.zeros((B, 42)), exps=(-20, 4, -8))
    assert ok[0]
    assert_parity(oracle, LEMKE_REG, M, q, z0=np.zeros((B, 42)))


def test_kats_on_gpu(oracle):
    g, dt = 9.81, 1e-3
    M = np.array([[[1.0, -1, 0], [-1, 2, -1], [0, -1, 2]]])
    q = np.array([[-g * dt, 0.0, 0.0]])
    for kind in (FAST, FAST_REG, LEMKE, LEMKE_REG):
        ok, z, _ = run_gpu(kind, M, q, z_size=np.zeros(1, dtype=np.int32))
        assert ok[0]
        np.testing.assert_allclose(z[0], g * dt * np.array([3.0, 2.0, 1.0]), rtol=1e-13)
    # singular first pivot -> ladder (see tests/test_oracle_lcp.py)
    M = np.array([[[0.0, 0.0], [0.0, 1.0]]]); q = np.array([[-1.0, -1.0]])
    assert_parity(oracle, FAST, M, q, z_size=np.zeros(1, dtype=np.int32))
    assert_parity(oracle, FAST_REG, M, q, z_size=np.zeros(1, dtype=np.int32), exps=(-20, 4, 20))


def test_lemke_rand_consumption(oracle):
    M, q = synth.random_lcp(4, 6, "pd", seed=9)
    assert_parity(oracle, LEMKE, M, q, z_size=np.array([6, 0, 12, 6], dtype=np.int32))


def test_full_batch_properties():
    """4096 worlds (BASELINE config 2 size): size-independent properties --
    complementarity of every accepted solution and batch-order independence."""
    B = 4096
    M64, q64 = synth.sphere_stack_impact_lcp(64)
    reps = B // 64
    M = np.tile(M64, (reps, 1, 1)); q = np.tile(q64, (reps, 1))
    ok, z, lcp = run_gpu(FAST_REG, M, q, z0=np.zeros((B, 42)), exps=(-20, 4, -8))
    # identical problems with identical rand() streams give identical answers
    for r in range(1, reps):
        np.testing.assert_array_equal(ok[:64], ok[r * 64:(r + 1) * 64])
        np.testing.assert_array_equal(z[:64], z[r * 64:(r + 1) * 64])
    w = np.einsum("bij,bj->bi", M, z) + q
    good = ok.astype(bool)
    assert good.sum() > 0
    assert z[good].min() > -1e-9 and w[good].min() > -1e-7 and np.abs(z[good] * w[good]).max() < 1e-7


@pytest.mark.parametrize("kind", [FAST, FAST_REG, LEMKE, LEMKE_REG])
@pytest.mark.parametrize("n,fam", [(65, "pd"), (100, "pd"), (100, "psd"), (130, "copos"), (200, "pd")])
def test_block_solver_parity(oracle, kind, n, fam):
    """n > 64: the workgroup-per-problem solver (mh_lcp_block.h, M read in place from HBM) against
    the oracle, bit for bit like the wave solver."""
    B = 3
    M, q = synth.random_lcp(B, n, fam, seed=31 * n + 1)
    assert_parity(oracle, kind, M, q, z_size=np.zeros(B, dtype=np.int32))


def test_block_solver_warm_start_and_rand_consumption(oracle):
    n = 96
    M, q = synth.random_lcp(4, n, "pd", seed=5)
    ok, z, _ = run_gpu(LEMKE, M, q)
    assert ok.all()
    q2 = q + 1e-3 * np.random.default_rng(2).standard_normal(q.shape)
    assert_parity(oracle, FAST, M, q2, z0=z)
    assert_parity(oracle, FAST_REG, M, q2, z0=z)
    assert_parity(oracle, LEMKE, M, q, z_size=np.array([n, 0, 2 * n, n], dtype=np.int32))


def test_block_solver_large_n(oracle):
    """n = 768 (between BASELINE config 4's trimmed sizes): parity with the oracle (which finishes
    one problem in seconds) plus the LCP conditions on the accepted solutions.  Problem 1 makes
    lcp_fast give up after 2n pivots -- on both sides."""
    B, n = 2, 768
    M, q = synth.random_lcp(B, n, "pd", seed=77)
    ok = assert_parity(oracle, FAST, M, q, z_size=np.zeros(B, dtype=np.int32))
    assert ok[0] and not ok[1]
    _, z, lcp = run_gpu(FAST, M, q, z_size=np.zeros(B, dtype=np.int32))
    w = M[0] @ z[0] + q[0]
    scale = np.abs(M[0]).max() * n
    assert z[0].min() >= 0.0 and w.min() > -1e-10 * scale and np.abs(z[0] * w).max() < 1e-10 * scale
    assert (lcp.pivots <= 2 * n).all()


def _pd_problem(B, n, seed, active):
    """M = A A' / n + I (PD, dense), q chosen so that exactly the first `active` variables are positive at the solution
    (z = 1 there, w = 1 elsewhere): sizes of the nonbasic block are known without solving."""
    rng = np.random.default_rng(seed)
    M = np.empty((B, n, n)); q = np.empty((B, n))
    for b in range(B):
        A = rng.standard_normal((n, n))
        M[b] = A @ A.T / n + np.eye(n)
        zs = np.zeros(n); zs[:active] = 1.0
        ws = np.ones(n); ws[:active] = 0.0
        q[b] = ws - M[b] @ zs
    return M, q


def test_block_solver_lists_beyond_the_lds_caps(oracle):
    """k > 1024: the nonbasic index list and the right-hand side no longer fit the solver's LDS staging (LIST_CAP /
    RHS_CAP in mh_lcp_block.h) and live in the HBM workspace.  n = 1100: a warm-started lcp_fast whose nonbasic block
    is 1060 x 1060, and a Lemke run (its basis is always n x n) that needs a handful of pivots -- both bit for bit."""
    n = 1100
    M, q = _pd_problem(1, n, seed=3, active=1060)
    z0 = np.zeros((1, n)); z0[0, :1060] = 1.0 + 1e-3 * np.random.default_rng(4).standard_normal(1060)
    ok = assert_parity(oracle, FAST, M, q, z0=z0)
    assert ok[0]
    M, q = _pd_problem(1, n, seed=5, active=3)
    ok = assert_parity(oracle, LEMKE, M, q, z_size=np.array([n], dtype=np.int32))
    assert ok[0]


@pytest.mark.parametrize("active", [1, 15, 16, 17, 33, 100, 176, 190, 191, 192, 200])
def test_register_lu_block_edges(oracle, active):
    """mh_lu_reg.inc: in the 1024-thread geometry lcp_fast solves a nonbasic system of up to 191 rows in the registers of its sixteen waves
    (columns by wave, sixteen per register slot).  Warm starts whose nonbasic block has exactly `active` rows -- one wave's worth, one slot's
    worth and one more, the last size the registers take, the first that goes through the workspace -- against the oracle, bit for bit."""
    n = 400                                   # (wide geometry from n = 384 up)
    M, q = _pd_problem(2, n, seed=100 + active, active=active)
    z0 = np.zeros((2, n)); z0[:, :active] = 1.0 + 1e-3 * np.random.default_rng(active).standard_normal((2, active))
    ok = assert_parity(oracle, FAST, M, q, z0=z0)
    assert ok.all()


@pytest.mark.parametrize("n,seed", [(150, 1), (190, 2), (260, 3)])
def test_register_lu_with_row_exchanges(oracle, n, seed):
    """general (unsymmetric, indefinite) matrices in the 1024-thread geometry: every factorisation exchanges rows, lcp_fast wanders through
    nonbasic sets of all sizes (n = 260: on both sides of the 191-row limit) and mostly gives up -- pivots, traces, rand() streams and
    whatever it solves equal the oracle's; the same bits with the register routine off (key 10)"""
    from moby_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    B = 3
    M = rng.standard_normal((B, n, n)) + 0.5 * np.eye(n); q = rng.standard_normal((B, n))
    M[1] = M[1] @ M[1].T / n + 0.05 * rng.standard_normal((n, n))          # nearly symmetric: many pivots before it ends
    _lib.check(lib.mh_debug_set(2, 2))
    try:
        assert_parity(oracle, FAST, M, q, z_size=np.zeros(B, dtype=np.int32))
        a = run_gpu(FAST_REG, M, q, z_size=np.zeros(B, dtype=np.int32), exps=(-20, 4, -8))
        _lib.check(lib.mh_debug_set(10, 0))
        b = run_gpu(FAST_REG, M, q, z_size=np.zeros(B, dtype=np.int32), exps=(-20, 4, -8))
    finally:
        _lib.check(lib.mh_debug_set(2, 0)); _lib.check(lib.mh_debug_set(10, 1))
    np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
    for f in ("pivots", "rng", "trace", "trace_len", "z_size"):
        np.testing.assert_array_equal(getattr(a[2], f), getattr(b[2], f), err_msg=f)


@pytest.mark.parametrize("n", [1024, 2048, 4096])
def test_block_solver_config4_sizes_properties(n):
    """The sizes BASELINE config 4 rests on (n = 1024: 32-box stacks, n = 2048: 64-box stacks) and the entry's advertised maximum
    (MH_LCP_MAX_N_BLOCK = 4096), beyond what the oracle
    finishes in seconds: the LCP conditions on every accepted solution, the known solution recovered, identical
    problems giving identical answers (batch-order independence), for lcp_fast (warm) and lcp_lemke (cold)."""
    B = 2
    M1, q1 = _pd_problem(1, n, seed=n, active=n - 40)
    M = np.repeat(M1, B, axis=0); q = np.repeat(q1, B, axis=0)
    z0 = np.zeros((B, n)); z0[:, :n - 40] = 1.0
    ok, z, lcp = run_gpu(FAST, M, q, z0=z0)
    assert ok.all()
    np.testing.assert_array_equal(z[0], z[1])
    w = M[0] @ z[0] + q[0]
    scale = np.abs(M[0]).max() * n
    assert z[0].min() >= 0.0 and w.min() > -1e-10 * scale and np.abs(z[0] * w).max() < 1e-10 * scale
    np.testing.assert_allclose(z[0, :n - 40], 1.0, atol=1e-9); assert (z[0, n - 40:] == 0.0).all()
    M1, q1 = _pd_problem(1, n, seed=n + 1, active=4)
    M = np.repeat(M1, B, axis=0); q = np.repeat(q1, B, axis=0)
    ok, z, lcp = run_gpu(LEMKE, M, q, z_size=np.full(B, n, dtype=np.int32))
    assert ok.all() and (lcp.pivots <= 50).all()
    np.testing.assert_array_equal(z[0], z[1])
    w = M[0] @ z[0] + q[0]
    assert z[0].min() >= 0.0 and w.min() > -1e-10 * scale and np.abs(z[0] * w).max() < 1e-10 * scale
    np.testing.assert_allclose(z[0, :4], 1.0, atol=1e-9)


@pytest.mark.parametrize("n,active,kind", [(1025, 30, LEMKE), (1536, 60, LEMKE_REG), (2048, 90, LEMKE)])
def test_structure_exploiting_lu_above_1024_rows(oracle, n, active, kind):
    """1024 < n <= 2048: lcp_lemke's bases go through the structure-exploiting LU with TWO rows per lane (mh_lcp_blkx.hip; until round 5 every pivot
    there was a dense dgesv of the assembled basis).  Dense PD problems whose solution has `active` positive variables: Lemke walks at least that many
    pivots, the basis fills with dense columns one at a time, fill-ins and panel boundaries included.  Against the oracle bit for bit (status, pivots,
    trace, rand(), z); the oracle solves its bases with its bit-equal model of the same routine (tests/test_oracle_compact_lu.py holds that to the dense
    dgesv) -- and the device's DENSE route (mh_debug_set(3, 0)) must give the same answer as its structure-exploiting one."""
    from moby_amd import _lib
    M, q = _pd_problem(2, n, seed=7 * n, active=active)
    oracle.lib.oracle_dbg_lemke_compact(8)
    try:
        ok = assert_parity(oracle, kind, M, q, z_size=np.array([n, 0], dtype=np.int32))
    finally:
        oracle.lib.oracle_dbg_lemke_compact(0)
    assert ok.all()
    a = run_gpu(kind, M[:1], q[:1], z_size=np.array([n], dtype=np.int32))
    assert int(a[2].pivots[0]) >= active
    _lib.check(_lib.load().mh_debug_set(3, 0))
    try:
        b = run_gpu(kind, M[:1], q[:1], z_size=np.array([n], dtype=np.int32))
    finally:
        _lib.check(_lib.load().mh_debug_set(3, 1))
    np.testing.assert_array_equal(a[1], b[1])
    for f in ("pivots", "trace_len", "rng"):
        np.testing.assert_array_equal(getattr(a[2], f), getattr(b[2], f), err_msg=f)


@pytest.mark.parametrize("n,active,kind", [(513, 25, LEMKE), (800, 50, LEMKE_REG), (1024, 70, LEMKE)])
def test_four_rows_per_lane_geometry_between_512_and_1024_rows(oracle, n, active, kind):
    """512 < n <= 1024 in the 256-thread geometry (mh_lcp_blky.hip: four rows per lane, two problems per CU -- what the ladder's tasks take when there
    are more of them than CUs; mh_debug_set(2, 5) forces it): against the oracle bit for bit, and equal to the 1024-thread geometry's answer."""
    from moby_amd import _lib
    M, q = _pd_problem(2, n, seed=11 * n, active=active)
    lib = _lib.load()
    _lib.check(lib.mh_debug_set(2, 5))
    try:
        oracle.lib.oracle_dbg_lemke_compact(8)
        try:
            ok = assert_parity(oracle, kind, M, q, z_size=np.array([n, 0], dtype=np.int32))
        finally:
            oracle.lib.oracle_dbg_lemke_compact(0)
        assert ok.all()
        a = run_gpu(kind, M, q, z_size=np.array([n, 0], dtype=np.int32))
        _lib.check(lib.mh_debug_set(2, 2))
        b = run_gpu(kind, M, q, z_size=np.array([n, 0], dtype=np.int32))
    finally:
        _lib.check(lib.mh_debug_set(2, 0))
    assert (a[2].pivots >= active).all()
    np.testing.assert_array_equal(a[1], b[1])
    for f in ("pivots", "trace_len", "rng"):
        np.testing.assert_array_equal(getattr(a[2], f), getattr(b[2], f), err_msg=f)


def test_cpp_adapter_example():
    """The Moby::LCP-shaped C++ adapter (moby_amd/cpp/MobyHipLCP.h) links against
    the C ABI and reproduces the KAT."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cpp = os.path.join(root, "moby_amd", "cpp")
    exe = os.path.join(cpp, "example_lcp")
    subprocess.check_call(["g++", "-std=c++11", os.path.join(cpp, "example_lcp.cpp"), "-L" + os.path.join(root, "moby_amd"),
                           "-lmoby_hip", "-Wl,-rpath," + os.path.join(root, "moby_amd"), "-o", exe])
    out = subprocess.check_output([exe]).decode()
    vals = [float(x) for x in out.splitlines()[0].split("z=")[1].split()]
    np.testing.assert_allclose(vals, 9.81e-3 * np.array([3.0, 2.0, 1.0]), rtol=1e-13)
    assert "lemke ok=1" in out


@pytest.mark.parametrize("geometry", [1, 2, 3, 4])
@pytest.mark.parametrize("kind,n,fam", [(FAST, 100, "pd"), (LEMKE, 100, "psd"), (FAST_REG, 300, "copos"), (LEMKE_REG, 300, "pd"), (LEMKE, 260, "pd")])
def test_block_solver_both_thread_geometries(oracle, geometry, kind, n, fam):
    """The workgroup-per-problem solver exists with 256 and with 1024 threads per problem (chosen by n and B,
    mh_capi.hip) and, for the lcp_lemke kinds, with 64 and 128 threads; mh_debug_set(2, .) forces one: all reproduce the oracle bit
    for bit at sizes on either side of the switch."""
    if geometry >= 3 and kind in (FAST, FAST_REG):
        pytest.skip("the lcp_fast kinds have no 64 / 128-thread geometry")
    from moby_amd import _lib
    lib = _lib.load()
    B = 2
    M, q = synth.random_lcp(B, n, fam, seed=17 * n + geometry)
    _lib.check(lib.mh_debug_set(2, geometry))
    try:
        assert_parity(oracle, kind, M, q, z_size=np.zeros(B, dtype=np.int32))
    finally:
        _lib.check(lib.mh_debug_set(2, 0))


def test_nan_ratio_in_the_lemke_ladder_ends_the_attempt_like_min_element(oracle):
    """tests/golden/lemke_ladder_n96_case.npz (an impact LCP of two box stacks with compliance, found by tests/tools/fuzz_big.py): on
    the rung lambda = 1e-13 the ratio of the FIRST candidate turns NaN after 453 pivots; std::min_element (LCP.cpp:920) keeps a NaN
    that comes first, the candidate set empties and the attempt fails there (:946-958) -- a NaN-ignoring minimum would pivot on.
    Block solver (n = 96): status, pivot counts, the whole pivot trace and z against the oracle, for the ladder and for that rung."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lemke_ladder_n96_case.npz"))
    MM, qq, rng = d["MM"], d["qq"], d["rng"]; n = len(qq)
    lam = MM.copy(); lam[np.arange(n), np.arange(n)] += 10.0 ** -13
    for kind, M, exps in ((LEMKE_REG, MM, None), (LEMKE, lam, None)):
        ro = oracle.lcp(kind, M, qq, z=np.zeros(n), z_size=n, rng=rng, exps=exps, trace_cap=8192)
        g = L.LCP(1); g.rng[0] = rng; z = np.zeros((1, n))
        ok = g._solve(kind, M, qq, z, DEFAULT_EXPS[kind] if kind == LEMKE_REG else None, z_size=np.array([n], dtype=np.int32), trace_cap=8192)
        assert bool(ok[0]) == ro["ok"] and int(g.pivots[0]) == ro["pivots"], (kind, ok, g.pivots, ro["pivots"])
        assert int(g.trace_len[0]) == ro["trace_len"] and np.array_equal(g.trace[0][:len(ro["trace"])], ro["trace"])
        if ro["ok"]:
            assert np.array_equal(z[0], ro["z"])
    assert ro["ok"] is False and ro["pivots"] == 453          # the rung alone: gives up at the NaN
