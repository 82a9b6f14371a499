"""Debug driver: dump every Drumwright-Shell LCP the oracle solves for ONE world of a fuzz_artic case and replay each through the HIP
LCP entry (lcp_fast_regularized(-20, 4, -8), then the Lemke ladder); prints the solves whose status / pivot counts / z differ.
    python tests/tools/artic_lcp_replay.py <seed> <world>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_artic as F
from moby_amd import artic as A, scene as S, lcp as L
from tests.oracle_api import Oracle, FAST_REG, LEMKE_REG

seed, world = int(sys.argv[1]), int(sys.argv[2])
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
m, q0, qd0, nsteps = F.complete_case(o, seed)
B = q0.shape[0]
path = "/tmp/artic_lcp_dump_%d.bin" % seed
o.lib.oracle_dbg_lcp_dump(path.encode())
q = q0[world:world + 1].copy(); qd = qd0[world:world + 1].copy(); aux = S.new_aux(1)
o.artic_step(m, q, qd, aux, 1e-3, nsteps)
o.lib.oracle_dbg_lcp_dump(None)
raw = open(path, "rb").read(); off = 0; k = 0; bad = 0
while off < len(raw):
    n, okf, pf, pl, ok = np.frombuffer(raw, dtype=np.int32, count=5, offset=off); off += 20
    rs = np.frombuffer(raw, dtype=np.uint32, count=32, offset=off).copy(); off += 128
    MM = np.frombuffer(raw, dtype=np.float64, count=n * n, offset=off).reshape(n, n).T.copy(); off += 8 * n * n
    qq = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy(); off += 8 * n
    z_in = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy(); off += 8 * n
    g = L.LCP(1); g.rng[0] = rs; z = z_in.reshape(1, n).copy()
    ok1 = bool(g.lcp_fast_regularized(MM, qq, z, -20, 4, -8, trace_cap=65536)[0]); p1 = int(g.pivots[0]); p2 = 0; ok2 = ok1
    tg = g.trace[0][:int(g.trace_len[0])].copy(); rng_after = g.rng[0].copy()
    ro = o.lcp(FAST_REG, MM, qq, z=z_in, z_size=n, rng=rs, exps=(-20, 4, -8), trace_cap=65536)
    if not ok1:
        z[:] = 0.0
        ok2 = bool(g.lcp_lemke_regularized(MM, qq, z)[0]); p2 = int(g.pivots[0])
    if (ok1, p1, p2, ok2) != (bool(okf), int(pf), int(pl), bool(ok)):
        bad += 1
        print("solve %d: n %d  oracle fast ok %d piv %d, lemke piv %d, ok %d | gpu fast ok %d piv %d, lemke piv %d, ok %d" % (k, n, okf, pf, pl, ok, ok1, p1, p2, ok2))
        to = ro["trace"]; mlen = min(len(to), len(tg)); d = np.nonzero(to[:mlen] != tg[:mlen])[0]; first = int(d[0]) if len(d) else mlen
        print("   fast_reg alone: oracle ok %s pivots %d trace %d | gpu trace %d; traces agree up to %d: oracle %s | gpu %s" % (ro["ok"], ro["pivots"], len(to), len(tg), first,
              to[max(0, first - 5):first + 5], tg[max(0, first - 5):first + 5]))
        print("   rng equal after fast_reg:", np.array_equal(ro["rng"], rng_after), " z_in nonzero:", int((z_in != 0).sum()), " nan in MM/qq/z_in:", np.isnan(MM).any(), np.isnan(qq).any(), np.isnan(z_in).any())
        np.savez("/tmp/artic_lcp_%d_%d.npz" % (seed, k), MM=MM, qq=qq, rng=rs, z=z_in)
        if bad >= 2: break
    k += 1
print("%d solves replayed, %d differ" % (k, bad))
