"""GPU impact handler vs the oracle on box stacks (debug / parity tool): python tools/impact_check.py nboxes [B] [eps] [mu]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import scene as S, impact as I
from tests.oracle_api import Oracle

def compare(nbx, B, eps, mu, calls=2, verbose=True):
    o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    mass, J, st, cs = I.box_stack(nbx, B=B, epsilon=eps, mu=mu)
    nc = 4 * nbx; n = I.lcp_size(nc, 4)
    ib = I.ImpactBatch(B, nbx, nc, 4, mass, J)
    assert ib.n == n
    aux = S.new_aux(B); zl = np.zeros((B, n)); zb = np.zeros((B, n))
    bad = 0
    st_g = st.copy(); st_o = st.copy()
    for call in range(calls):
        t0 = time.time(); r = ib.process(st_g, cs); tg = time.time() - t0
        if call == 0:
            MMg, qqg = ib.debug_lcp()
        t0 = time.time()
        imp_o = np.zeros((B, nc, 3)); piv0 = aux["lcp_pivots"].copy(); sol0 = aux["lcp_solves"].copy()
        for w in range(B):
            if call == 0 and w < 2:
                nn, MMo, qqo = o.impact_lcp(nbx, mass, J, st_o[w].copy(), cs[w], n)
                if nn == n and not (np.array_equal(MMo, MMg[w]) and np.array_equal(qqo, qqg[w])):
                    bad += 1; print("  world %d: _MM/_qq differ: max |dM| %.3e max |dq| %.3e" % (w, np.abs(MMo - MMg[w]).max(), np.abs(qqo - qqg[w]).max()))
            imp_o[w], _ = o.impact_process(nbx, mass, J, st_o[w], cs[w], aux[w:w + 1], zl[w], zb[w], n)
        to = time.time() - t0
        same = np.array_equal(r["state"], st_o)
        sst = np.array_equal(r["status"], aux["status"]); spv = np.array_equal(r["pivots"], (aux["lcp_pivots"] - piv0).astype(np.uint32))
        ssl = np.array_equal(r["solves"], (aux["lcp_solves"] - sol0).astype(np.int32)); sim = np.array_equal(r["impulses"], imp_o)
        if verbose:
            print("call %d: n %d  GPU %.3f s  oracle %.3f s  state %s status %s pivots %s solves %s impulses %s  (pivots max %d, status or %d, max|dv| %.2e)"
                  % (call, n, tg, to, same, sst, spv, ssl, sim, r["pivots"].max(), np.bitwise_or.reduce(r["status"]), np.abs(r["state"] - st_o).max()))
        bad += (not same) + (not sst) + (not spv) + (not ssl) + (not sim)
        st_g = r["state"].copy()
        # next call: same contacts, gravity applied again (a resting stack, warm-started)
        for a in (st_g, st_o):
            a.reshape(B, nbx, 13)[:, :, 8] += -9.81e-3
    ib.close()
    return bad

if __name__ == "__main__":
    nbx = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    eps = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0; mu = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-4
    print("mismatches:", compare(nbx, B, eps, mu))
