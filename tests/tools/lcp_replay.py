"""Debug driver: dump every impact LCP the oracle solves in one fuzz_big case (oracle_dbg_lcp_dump: inputs, rand() state, pivot
counts of lcp_fast_regularized(-20, 4, -8) and of the Lemke ladder) and replay each through the HIP LCP entry; prints the solves
whose status / pivot counts / solutions differ.     python tests/tools/lcp_replay.py <fuzz_big seed> [world]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_big as F
from moby_amd import scene as S, lcp as L
from tests.oracle_api import Oracle

seed = int(sys.argv[1]); world = int(sys.argv[2]) if len(sys.argv) > 2 else 0
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
sc, st, nsteps, par, nb, npairs = F.make_case(seed)
cap = sc.lcp_capacity()
path = "/tmp/lcp_dump_%d.bin" % seed
o.lib.oracle_dbg_lcp_dump(path.encode())
aux = S.new_aux(1); zl = np.zeros(cap); zb = np.zeros(cap); s0 = st[world].copy()
o.big_step(sc, s0, aux, 1e-3, nsteps, zlast=zl, zbuf=zb, cap=cap)
o.lib.oracle_dbg_lcp_dump(None)
raw = open(path, "rb").read(); off = 0; k = 0; bad = 0
while off < len(raw):
    n, okf, pf, pl, ok = np.frombuffer(raw, dtype=np.int32, count=5, offset=off); off += 20
    rng = np.frombuffer(raw, dtype=np.uint32, count=32, offset=off).copy(); off += 128
    MM = np.frombuffer(raw, dtype=np.float64, count=n * n, offset=off).reshape(n, n).T.copy(); off += 8 * n * n      # column-major in the file
    qq = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy(); off += 8 * n
    z = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy().reshape(1, n); off += 8 * n
    g = L.LCP(1); g.rng[0] = rng
    ok1 = bool(g.lcp_fast_regularized(MM, qq, z, -20, 4, -8)[0]); p1 = int(g.pivots[0]); p2 = 0; ok2 = ok1
    rng_after_fast = g.rng[0].copy()
    if not ok1:
        z[:] = 0.0
        ok2 = bool(g.lcp_lemke_regularized(MM, qq, z)[0]); p2 = int(g.pivots[0])
    if (ok1, p1, p2, ok2) != (bool(okf), int(pf), int(pl), bool(ok)):
        bad += 1
        print("solve %d: n %d  oracle fast ok %d piv %d, lemke piv %d, ok %d | gpu fast ok %d piv %d, lemke piv %d, ok %d" % (k, n, okf, pf, pl, ok, ok1, p1, p2, ok2))
        # where do the pivot traces part?  (per attempt: 0x40000000 | attempt, then entering / leaving ids per pivot)
        from tests.oracle_api import LEMKE_REG
        cap_t = 200000
        ro = o.lcp(LEMKE_REG, MM, qq, z=np.zeros(n), z_size=n, rng=rng_after_fast, trace_cap=cap_t)
        g3 = L.LCP(1); g3.rng[0] = rng_after_fast; z3 = np.zeros((1, n))
        g3._solve(L.MH_LCP_LEMKE_REG, MM, qq, z3, (-20, 1, 1), z_size=np.array([n], dtype=np.int32), trace_cap=cap_t)
        to = ro["trace"]; tg = g3.trace[0][:int(g3.trace_len[0])]
        print("   lemke alone: oracle pivots %d trace %d | gpu pivots %d trace %d" % (ro["pivots"], len(to), int(g3.pivots[0]), len(tg)))
        m = min(len(to), len(tg)); d = np.nonzero(to[:m] != tg[:m])[0]
        first = int(d[0]) if len(d) else m
        print("   traces agree up to entry %d; oracle %s | gpu %s" % (first, to[max(0, first - 6):first + 6], tg[max(0, first - 6):first + 6]))
        marks_o = [(i, int(v) & 0xffff) for i, v in enumerate(to) if v & 0x40000000]; marks_g = [(i, int(v) & 0xffff) for i, v in enumerate(tg) if v & 0x40000000]
        print("   attempt marks oracle", marks_o, "\n   attempt marks gpu   ", marks_g)
    k += 1
print("%d solves replayed, %d differ" % (k, bad))
