"""Does a FUSED trailing update in the block LU (what v_mfma_f64_16x16x4_f64 computes: a - l*u with one rounding) keep
the pivot sequences of the oracle (mul and sub rounded separately, as -ffp-contract=off dgetf2)?  Run against a library
built with -DMH_BLK_FMA_EXPERIMENT (tools/build_variant.sh fma "-DMH_BLK_FMA_EXPERIMENT"; MOBY_HIP_LIB=...).
Counts, over random LCPs of the block-solver range and over box-stack impact LCPs, how many problems change their
status / pivot count / pivot trace / solution bits."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import synth, impact as I
from tests.oracle_api import Oracle, FAST, FAST_REG, LEMKE, LEMKE_REG
import tests.test_lcp_gpu as T

o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
rng = np.random.default_rng(0)
out = {"random": {"problems": 0, "status": 0, "pivots": 0, "trace": 0, "z_bits": 0}, "box_stack": {}}
for it in range(60):
    n = int(rng.integers(65, 260)); fam = str(rng.choice(["pd", "psd", "copos"])); kind = int(rng.integers(0, 4))
    M, q = synth.random_lcp(2, n, fam, seed=int(rng.integers(0, 10**6)))
    zs = np.zeros(2, dtype=np.int32)
    ok, z, lcp = T.run_gpu(kind, M, q, z_size=zs)
    for b in range(2):
        r = o.lcp(kind, M[b], q[b], z_size=0, rng=o.rand_state(1), trace_cap=T.TRACE_CAP)
        d = out["random"]; d["problems"] += 1
        d["status"] += int(bool(ok[b]) != r["ok"]); d["pivots"] += int(int(lcp.pivots[b]) != r["pivots"])
        L = min(r["trace_len"], T.TRACE_CAP, int(lcp.trace_len[b]))
        d["trace"] += int(int(lcp.trace_len[b]) != r["trace_len"] or not np.array_equal(lcp.trace[b, :L], r["trace"][:L]))
        d["z_bits"] += int(r["ok"] and bool(ok[b]) and not np.array_equal(z[b], r["z"]))
# box-stack impact LCPs (degenerate: 4 redundant corner contacts per interface), through the impact entry vs the oracle
for nbx in (3, 5, 8):
    B = 6
    mass, J, st, cs = I.box_stack(nbx, B=B)
    n = I.lcp_size(4 * nbx, 4)
    r = I.ImpactBatch(B, nbx, 4 * nbx, 4, mass, J).process(st, cs)
    from moby_amd import scene as S
    aux = S.new_aux(B); so = st.copy(); piv = np.zeros(B, dtype=np.int64)
    for w in range(B):
        oimp, _ = o.impact_process(nbx, mass, J, so[w], cs[w], aux[w:w + 1], np.zeros(n), np.zeros(n), n)
    out["box_stack"]["%d boxes (n = %d)" % (nbx, n)] = {
        "worlds": B, "pivot_count_differs": int((r["pivots"] != aux["lcp_pivots"].astype(np.uint32)).sum()),
        "state_bits_differ": int((~(r["state"] == so).all(axis=1)).sum()),
        "max_velocity_diff": float(np.abs(r["state"] - so).max())}
print(json.dumps(out))
