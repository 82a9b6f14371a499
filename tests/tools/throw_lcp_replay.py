"""Every impact LCP the oracle solves in a fuzz_throw world (oracle_dbg_lcp_dump: M, q, the warm start and the rand() state it was entered with), replayed through the
HIP LCP entry -- lcp_fast_regularized(-20, 4, -8), then on failure lcp_lemke_regularized from z = 0 (ICH-QP:219-225) -- next to the oracle's own solvers on the same
inputs: the first LCP whose pivot counts / traces / z differ.      python tests/tools/throw_lcp_replay.py SEED WORLD STEPS"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import scene as S
from moby_amd.lcp import LCP
from tests.oracle_api import Oracle, FAST_REG, LEMKE_REG
import importlib.util
_spec = importlib.util.spec_from_file_location("_twg", os.path.join(ROOT, "tests", "test_world_gpu.py")); _twg = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_twg)
seed, world, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
sc, st0 = _twg._harsh_scene(seed)
fd, path = tempfile.mkstemp(suffix=".lcps"); os.close(fd)
o.lib.oracle_dbg_lcp_dump(path.encode())
so = st0[world].copy(); ao = S.new_aux(1)
o.world_step(sc, so, ao, 1e-3, steps, want_traj=False)
o.lib.oracle_dbg_lcp_dump(None)
raw = open(path, "rb").read(); os.unlink(path)
off = 0; k = 0
TC = 4096
while off < len(raw):
    n, okf, pf, pl, ok = np.frombuffer(raw, dtype=np.int32, count=5, offset=off); off += 20
    rng = np.frombuffer(raw, dtype=np.uint32, count=32, offset=off).copy(); off += 128
    M = np.frombuffer(raw, dtype=np.float64, count=n * n, offset=off).reshape(n, n).T.copy(); off += 8 * n * n
    q = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy(); off += 8 * n
    z = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy(); off += 8 * n
    lcp = LCP(1); lcp.rng[0] = rng
    zg = z[None].copy()
    okg = lcp.lcp_fast_regularized(M[None], q[None], zg, -20, 4, -8, z_size=np.array([n], dtype=np.int32), trace_cap=TC)
    r = o.lcp(FAST_REG, M, q, z=z, z_size=n, rng=rng.copy(), exps=(-20, 4, -8), trace_cap=TC)
    same = bool(okg[0]) == r["ok"] and int(lcp.pivots[0]) == r["pivots"] and np.array_equal(lcp.rng[0], r["rng"])
    msg = "lcp %d n %d fast: gpu ok %d piv %d | oracle ok %d piv %d (dump: ok %d piv %d)%s" % (k, n, okg[0], lcp.pivots[0], r["ok"], r["pivots"], okf, pf, "" if same else "   <-- DIFFERS")
    if not same:
        L = min(int(lcp.trace_len[0]), r["trace_len"], TC)
        d = np.nonzero(lcp.trace[0, :L] != r["trace"][:L])[0]
        i0 = int(d[0]) if len(d) else L
        msg += " trace lens %d / %d, first difference at %d: gpu %s oracle %s" % (lcp.trace_len[0], r["trace_len"], i0, lcp.trace[0, max(0, i0 - 3):i0 + 4].tolist(), r["trace"][max(0, i0 - 3):i0 + 4].tolist())
        msg += "; max|M| %.3e min|M!=0| %.3e max|q| %.3e nan/inf in M %d" % (np.abs(M).max(), np.abs(M[M != 0]).min(), np.abs(q).max(), int((~np.isfinite(M)).sum()))
    print(msg)
    if not okf:
        lcp2 = LCP(1); lcp2.rng[0] = r["rng"]
        z2 = np.zeros((1, n))
        ok2 = lcp2.lcp_lemke_regularized(M[None], q[None], z2, z_size=np.array([r["z_size"]], dtype=np.int32), trace_cap=TC)
        r2 = o.lcp(LEMKE_REG, M, q, z=np.zeros(n), z_size=r["z_size"], rng=r["rng"].copy(), trace_cap=TC)
        same2 = bool(ok2[0]) == r2["ok"] and int(lcp2.pivots[0]) == r2["pivots"] and np.array_equal(lcp2.rng[0], r2["rng"])
        print("        lemke ladder: gpu ok %d piv %d | oracle ok %d piv %d (dump piv %d)%s" % (ok2[0], lcp2.pivots[0], r2["ok"], r2["pivots"], pl, "" if same2 else "   <-- DIFFERS"))
        if not same2:
            L = min(int(lcp2.trace_len[0]), r2["trace_len"], TC)
            d = np.nonzero(lcp2.trace[0, :L] != r2["trace"][:L])[0]
            i0 = int(d[0]) if len(d) else L
            print("        trace lens %d / %d, first difference at %d: gpu %s oracle %s" % (lcp2.trace_len[0], r2["trace_len"], i0, lcp2.trace[0, max(0, i0 - 3):i0 + 4].tolist(), r2["trace"][max(0, i0 - 3):i0 + 4].tolist()))
            print("        max|M| %.3e max|q| %.3e" % (np.abs(M).max(), np.abs(q).max()))
    k += 1
