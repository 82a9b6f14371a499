"""lcp_fast's repeating pivot sequences (tests/test_oracle_fast_repeats.py) are skipped on the device (mh_lcp_wave.h: period 1;
mh_lcp_block.h: periods up to 8).  Nothing observable may change: status, z, z size, pivot counts, the rand() stream and the pivot trace
are those of the oracle, which runs every iteration -- on the impact LCPs of resting box stacks, where most of lcp_fast's iterations are
repetitions, and on the slow worlds of a long sphere-stack run."""
import numpy as np
import pytest

from moby_amd import _lib, scene as S
from moby_amd.lcp import LCP
from tests.boxstack_lcps import dumped_lcps
from tests.oracle_api import FAST_REG

pytestmark = pytest.mark.gpu
TRACE_CAP = 1 << 15
EXPS = (-20, 4, -8)                 # the impact handler's call (ICH:573)


def solve_gpu(probs):
    n = len(probs[0][1]); B = len(probs)
    lcp = LCP(B)
    M = np.stack([p[0] for p in probs]); q = np.stack([p[1] for p in probs]); z = np.stack([p[2] for p in probs])
    for b, p in enumerate(probs):
        lcp.rng[b] = p[3]
    ok = lcp.lcp_fast_regularized(M, q, z, *EXPS, z_size=np.full(B, n, dtype=np.int32), trace_cap=TRACE_CAP)
    return ok, z, lcp


def check_against_oracle(oracle, probs):
    ok, z, lcp = solve_gpu(probs)
    capped = 0
    for b, (M, q, z0, rng, pf, okf) in enumerate(probs):
        n = len(q)
        r = oracle.lcp(FAST_REG, M, q, z=z0, z_size=n, rng=rng, exps=EXPS, trace_cap=TRACE_CAP)
        assert r["pivots"] == pf and r["ok"] == okf
        tag = "problem %d (n %d, %d pivots)" % (b, n, pf)
        assert bool(ok[b]) == r["ok"], tag
        assert int(lcp.pivots[b]) == r["pivots"], tag
        assert int(lcp.trace_len[b]) == r["trace_len"], tag
        L = min(r["trace_len"], TRACE_CAP)
        np.testing.assert_array_equal(lcp.trace[b, :L], r["trace"][:L], err_msg=tag)
        np.testing.assert_array_equal(lcp.rng[b], r["rng"], err_msg=tag)
        assert int(lcp.z_size[b]) == r["z_size"], tag
        if r["ok"]:
            np.testing.assert_array_equal(z[b], r["z"], err_msg=tag)
        capped += pf >= 2 * n
    return capped, lcp


@pytest.mark.parametrize("nboxes,world,steps", [(2, 3, 4), (2, 4, 4), (3, 4, 3), (4, 3, 2), (4, 1, 3), (6, 1, 1)])
def test_box_stack_lcps_with_repeating_pivot_sequences(oracle, nboxes, world, steps):
    """n = 64 (wave solver), 96 .. 192 (block solver, both thread geometries by the launcher's rule)"""
    probs = [p for p in dumped_lcps(oracle, nboxes, world, steps) if len(p[1]) == 32 * nboxes]
    assert probs
    capped, _ = check_against_oracle(oracle, probs)
    assert capped > 0                                       # the case does contain calls that run into MAX_PIV


def test_block_solver_with_and_without_the_skip(oracle):
    probs = [p for p in dumped_lcps(oracle, 4, 3, 2) if len(p[1]) == 128]
    lib = _lib.load()
    outs = []
    for geometry in (1, 2):
        for skip in (1, 0):
            _lib.check(lib.mh_debug_set(2, geometry)); _lib.check(lib.mh_debug_set(5, skip))
            try:
                ok, z, lcp = solve_gpu(probs)
            finally:
                _lib.check(lib.mh_debug_set(2, 0)); _lib.check(lib.mh_debug_set(5, 1))
            outs.append((np.array(ok).copy(), z.copy(), lcp.pivots.copy(), lcp.rng.copy(), lcp.trace.copy(), lcp.trace_len.copy(), lcp.z_size.copy()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            np.testing.assert_array_equal(a, b)


def test_box_stack_lcps_through_the_register_lu(oracle):
    """the 1024-thread geometry's lcp_fast with its nonbasic systems in registers (mh_lu_reg.inc) on box-stack LCPs -- exact zeros, ties in
    the pivot search, singular systems that end an attempt: the oracle's bits, and the same bits with the routine off (key 10)"""
    lib = _lib.load()
    for nboxes, world, steps in ((4, 3, 2), (6, 1, 1)):
        probs = [p for p in dumped_lcps(oracle, nboxes, world, steps) if len(p[1]) == 32 * nboxes]
        outs = []
        _lib.check(lib.mh_debug_set(2, 2))
        try:
            check_against_oracle(oracle, probs)
            for on in (1, 0):
                _lib.check(lib.mh_debug_set(10, on))
                ok, z, lcp = solve_gpu(probs)
                outs.append((np.array(ok).copy(), z.copy(), lcp.pivots.copy(), lcp.rng.copy(), lcp.trace.copy(), lcp.trace_len.copy(), lcp.z_size.copy()))
        finally:
            _lib.check(lib.mh_debug_set(2, 0)); _lib.check(lib.mh_debug_set(10, 1))
        for a, b in zip(outs[0], outs[1]):
            np.testing.assert_array_equal(a, b)


def test_slow_worlds_of_a_long_sphere_stack_run(oracle):
    """Step 4200 of the headline batch: the worlds with the most pivots over the next 100 steps (lcp_fast runs into MAX_PIV about once a
    step there, four fifths of its iterations on one index set) against the oracle from the same states, 40 steps, bit for bit."""
    import torch
    from moby_amd.world import WorldBatchDevice
    B = 1024
    sc = S.sphere_stack_scene()
    wb = WorldBatchDevice(sc, S.sphere_stack_state_range(0, B))
    for k in (1000, 1000, 1000, 1000, 200):
        wb.step(1e-3, k); torch.cuda.synchronize()
    st0, a0 = wb.download()
    wb.step(1e-3, 100); torch.cuda.synchronize()
    _, a1 = wb.download()
    piv = a1["lcp_pivots"].astype(np.int64) - a0["lcp_pivots"].astype(np.int64)
    ids = np.sort(np.argsort(-piv)[:6])
    assert piv[ids].max() > 3 * np.median(piv)              # the slow ones are among these
    wb.close()
    sg = st0[ids].copy(); ag = a0[ids].copy()
    wb2 = WorldBatchDevice(sc, sg.copy(), aux=ag.copy())
    wb2.step(1e-3, 40); torch.cuda.synchronize()
    s_gpu, a_gpu = wb2.download()
    wb2.close()
    s_cpu = sg.copy(); a_cpu = ag.copy()
    oracle.world_step_batch(sc, s_cpu, a_cpu, 1e-3, 40)
    np.testing.assert_array_equal(s_gpu, s_cpu)
    for f in ("lcp_pivots", "lcp_solves", "lcp_rows", "status"):
        np.testing.assert_array_equal(a_gpu[f], a_cpu[f], err_msg=f)
    np.testing.assert_array_equal(a_gpu["rng"], a_cpu["rng"])
