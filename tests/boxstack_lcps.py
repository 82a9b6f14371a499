"""Impact LCPs of resting box stacks, cut from the CPU oracle's own steps (oracle_dbg_lcp_dump: every LCP solve_impact_lcp sees, with the
warm start and the rand() state it was entered with).  These are the problems on which lcp_fast fails by repeating itself
(LCP.cpp:176-187; oracle/lcp.hpp g_fast_stats), used by the tests of the solvers' repeat skipping."""
import os
import tempfile

import numpy as np

from moby_amd import scene as S, stack as K


def dumped_lcps(oracle, nboxes, world, steps):
    """[(M row-major, q, z_in, rng_in, pivots of lcp_fast_regularized(-20, 4, -8), its result)] of `steps` steps of one world."""
    sc = K.box_stack_scene(nboxes)
    st = K.box_stack_state(nboxes, 8)               # (the perturbations depend on the batch size: keep it fixed)
    fd, path = tempfile.mkstemp(suffix=".lcps"); os.close(fd)
    try:
        oracle.lib.oracle_dbg_lcp_dump(path.encode())
        s = st[world].copy(); aux = S.new_aux(1)
        oracle.big_step(sc, s, aux, 1e-3, steps)
        oracle.lib.oracle_dbg_lcp_dump(None)
        raw = open(path, "rb").read()
    finally:
        oracle.lib.oracle_dbg_lcp_dump(None)
        os.unlink(path)
    out = []; off = 0
    while off < len(raw):
        n, okf, pf, _pl, _ok = np.frombuffer(raw, dtype=np.int32, count=5, offset=off); off += 20
        rng = np.frombuffer(raw, dtype=np.uint32, count=32, offset=off).copy(); off += 128
        M = np.frombuffer(raw, dtype=np.float64, count=n * n, offset=off).reshape(n, n).T.copy(); off += 8 * n * n      # column-major in the file
        q = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy(); off += 8 * n
        z = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy(); off += 8 * n
        out.append((M, q, z, rng, int(pf), bool(okf)))
    return out
