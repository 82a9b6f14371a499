"""DESIGN 8 ("the oracle's dgesv above n = 64"), device half of the experiment.  Run with MOBY_HIP_LIB=build/variants/libmoby_hip_fma.so (tools/build_variant.sh fma
"-DMH_BLK_FMA_EXPERIMENT": every multiply-subtract of dgesv in the workgroup-per-problem solver fused) -- or without, for the unfused baseline of the same numbers.
 (1) is there ONE fused definition both sides can hold?  the variant against the oracle with oracle_dbg_lu_fma(1): random LCPs of 65-300 rows, all four kinds, and
     8-box stacks over two full steps -- status, pivots, trace, rand(), z / state, bit for bit;
 (2) what does it buy?  16 boxes x 1024 worlds, two full steps (cold, warm): seconds, pivots, failures.
Prints one JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import synth, scene as S, stack as K
from tests.oracle_api import Oracle
import tests.test_lcp_gpu as T

fused = "fma" in os.environ.get("MOBY_HIP_LIB", "")
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
o.lib.oracle_dbg_lu_fma(1 if fused else 0)
out = {"library": os.environ.get("MOBY_HIP_LIB", "moby_amd/libmoby_hip.so"), "fused": fused}
rng = np.random.default_rng(1)
d = {"problems": 0, "status": 0, "pivots": 0, "trace": 0, "rand": 0, "z_bits": 0}
for it in range(24):
    n = int(rng.integers(65, 300)); fam = str(rng.choice(["pd", "psd", "copos"])); kind = int(rng.integers(0, 4))
    M, q = synth.random_lcp(2, n, fam, seed=int(rng.integers(0, 10**6)))
    zs = np.zeros(2, dtype=np.int32)
    ok, z, lcp = T.run_gpu(kind, M, q, z_size=zs)
    for b in range(2):
        r = o.lcp(kind, M[b], q[b], z_size=0, rng=o.rand_state(1), trace_cap=T.TRACE_CAP)
        d["problems"] += 1
        d["status"] += int(bool(ok[b]) != r["ok"]); d["pivots"] += int(int(lcp.pivots[b]) != r["pivots"])
        L = min(r["trace_len"], T.TRACE_CAP, int(lcp.trace_len[b]))
        d["trace"] += int(int(lcp.trace_len[b]) != r["trace_len"] or not np.array_equal(lcp.trace[b, :L], r["trace"][:L]))
        d["rand"] += int(not np.array_equal(lcp.rng[b], r["rng"]))
        d["z_bits"] += int(r["ok"] and bool(ok[b]) and not np.array_equal(z[b], r["z"]))
out["random_lcps_differing_from_the_oracle"] = d
N, B, steps = 8, 3, 2
sc = K.box_stack_scene(N); st0 = K.box_stack_state(N, B)
bb = K.BigBatch(sc, st0); cap = bb.cap; bb.step(1e-3, steps); st, aux = bb.download(); bb.close()
diff = {"worlds": B, "state": 0, "pivots": 0, "rand": 0, "status": 0}
for w in range(B):
    so = st0[w].copy(); ao = S.new_aux(1)
    o.big_step(sc, so, ao, 1e-3, steps, zlast=np.zeros(cap), zbuf=np.zeros(cap), cap=cap)
    diff["state"] += int(not np.array_equal(so, st[w])); diff["pivots"] += int(ao["lcp_pivots"][0] != aux["lcp_pivots"][w])
    diff["rand"] += int(not np.array_equal(ao["rng"][0], aux["rng"][w])); diff["status"] += int(ao["status"][0] != aux["status"][w])
out["eight_box_stacks_two_steps_differing_from_the_oracle"] = diff
o.lib.oracle_dbg_lu_fma(0)
N, B = 16, 1024
sc = K.box_stack_scene(N); bb = K.BigBatch(sc, K.box_stack_state(N, B))
secs = []
for k in range(2):
    t0 = time.perf_counter(); bb.step(1e-3, 1); st, aux = bb.download(); secs.append(time.perf_counter() - t0)
wk = bb.lu_work(); bb.close()
out["config4_16x1024"] = {"seconds_per_step": secs, "lcp_pivots_mean": float(aux["lcp_pivots"].mean()), "lcp_pivots_max": int(aux["lcp_pivots"].max()),
                          "worlds_lcp_failed": int(((aux["status"] & 1) != 0).sum()), "worlds_impact_tol": int(((aux["status"] & 2) != 0).sum()),
                          "issued_flops": float(wk[:, 2].sum()), "solver_workgroup_seconds": float(wk[:, 3].sum())}
print(json.dumps(out))
