"""Generator of tests/golden/lemke_ladder_n96_case.npz: the 4th impact LCP (n = 96) of world 0 of the fuzz_big scene of seed 2053
(two box stacks with compliance), as the ORACLE assembles it -- a problem whose Lemke ladder runs into a NaN ratio on the rung
lambda = 1e-13 (std::min_element keeps a NaN that comes first: the candidate set empties, LCP.cpp:920-958).  CPU only.
    python tests/golden/make_lemke_nan_case.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import fuzz_big as F
from moby_amd import scene as S
from tests.oracle_api import Oracle

o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
sc, st, nsteps, par, nb, npairs = F.make_case(2053)
cap = sc.lcp_capacity(); path = "/tmp/lcp_dump_2053.bin"
o.lib.oracle_dbg_lcp_dump(path.encode())
aux = S.new_aux(1); zl = np.zeros(cap); zb = np.zeros(cap); s0 = st[0].copy()
o.big_step(sc, s0, aux, 1e-3, nsteps, zlast=zl, zbuf=zb, cap=cap)
o.lib.oracle_dbg_lcp_dump(None)
raw = open(path, "rb").read(); off = 0; k = 0
while off < len(raw):
    n, okf, pf, pl, ok = np.frombuffer(raw, dtype=np.int32, count=5, offset=off); off += 20
    rng = np.frombuffer(raw, dtype=np.uint32, count=32, offset=off).copy(); off += 128
    MM = np.frombuffer(raw, dtype=np.float64, count=n * n, offset=off).reshape(n, n).T.copy(); off += 8 * n * n
    qq = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy(); off += 8 * n
    z = np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy(); off += 8 * n
    if k == 3:
        np.savez(os.path.join(ROOT, "tests", "golden", "lemke_ladder_n96_case.npz"), MM=MM, qq=qq, rng=rng, z=z)
        print("saved: n", n, "fast pivots", pf, "lemke pivots", pl)
    k += 1
