"""Debug driver: one A-P impact batch on the GPU next to the oracle (python tests/tools/ap_case.py nbx B eps mu nk calls)."""
import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from moby_amd import impact as I, scene as S
import oracle_api
nbx, B, eps, mu, nk = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
calls = int(sys.argv[6])
o = oracle_api.Oracle(os.path.join('oracle', 'liboracle.so')); o.set_impact_model(1)
mass, J, st, cs = I.box_stack(nbx, B=B, epsilon=eps, mu=mu, nk=nk)
nc = 4 * nbx; n = I.lcp_size(nc, nk)
ib = I.ImpactBatch(B, nbx, nc, nk, mass, J, model=I.MH_IMPACT_MODEL_AP)
aux = S.new_aux(B); st_o = st.copy()
for call in range(calls):
    r = ib.process(st, cs)
    imp_o = np.zeros((B, nc, 3))
    for w in range(B):
        imp_o[w], _ = o.impact_process(nbx, mass, J, st_o[w], cs[w], aux[w:w + 1], np.zeros(n), np.zeros(n), n)
    print("case", sys.argv[1:], "call", call, "status", r["status"], aux["status"], "piv", r["pivots"], "solves", r["solves"], flush=True)
    for w in range(B):
        dv = np.abs(r["state"][w] - st_o[w]).max(); di = np.abs(r["impulses"][w] - imp_o[w]).max()
        print(" world", w, "max|dstate|", dv, "max|dimp|", di)
        if dv > 0:
            print("  gpu v", r["state"][w].reshape(nbx, 13)[:, 7:13].round(6).tolist()); print("  ora v", st_o[w].reshape(nbx, 13)[:, 7:13].round(6).tolist())
            print("  gpu cn", r["impulses"][w][:, 0].round(6).tolist()); print("  ora cn", imp_o[w][:, 0].round(6).tolist())
    st = r["state"].copy(); st.reshape(B, nbx, 13)[:, :, 8] += -9.81e-3; st_o[:] = st
ib.close()
