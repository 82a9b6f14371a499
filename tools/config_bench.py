"""Throughput of the other BASELINE configurations (bench.py carries config 2 only):
   config 3  rimless wheel x2048 (spokes + no-slip model, 'wheel' kernel variant)
   box       sitting / tumbling unit boxes x1024 ('large' variant, vertex-plane contacts, n = 40)
   LCP entry random PD problems through the wave (n = 42) and block (n = 128, 256) solvers
Prints one JSON line per case."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moby_amd import _lib, scene as S, synth
from moby_amd.world import WorldBatchDevice
from moby_amd.lcp import LCP
import torch


def world_case(name, sc, st, dt, nsteps, warm=20):
    wb = WorldBatchDevice(sc, st)
    occ = _lib.load().mh_world_batch_occupancy(wb.handle)
    wb.step(dt, warm); torch.cuda.synchronize()
    _, a0 = wb.download()
    t0 = time.perf_counter(); wb.step(dt, nsteps); torch.cuda.synchronize(); t = time.perf_counter() - t0
    _, a1 = wb.download()
    rows = int(a1["lcp_rows"].sum() - a0["lcp_rows"].sum()); solves = int(a1["lcp_solves"].sum() - a0["lcp_solves"].sum())
    B = st.shape[0]
    print(json.dumps({"case": name, "worlds": B, "steps": nsteps, "ms": t * 1e3, "world_steps_per_s": B * nsteps / t, "lcp_rows_per_s": rows / t,
                      "lcp_solves": solves, "mini_steps": int(a1["mini_steps"].sum() - a0["mini_steps"].sum()),
                      "bad_worlds": int(((a1["status"] & ~S.MH_WORLD_IMPACT_TOL) != 0).sum()), "blocks_per_cu": occ}))
    wb.close()


def lcp_case(n, B, kind="pd"):
    M, q = synth.random_lcp(min(B, 16), n, kind, seed=n)
    reps = B // M.shape[0]
    M = np.tile(M, (reps, 1, 1)); q = np.tile(q, (reps, 1))
    lcp = LCP(M.shape[0])
    z = np.zeros_like(q)
    lcp.lcp_fast(M, q, z, z_size=np.zeros(M.shape[0], dtype=np.int32))      # warm-up (includes PCIe)
    t0 = time.perf_counter(); ok = lcp.lcp_fast(M, q, z, z_size=np.zeros(M.shape[0], dtype=np.int32)); t = time.perf_counter() - t0
    print(json.dumps({"case": "lcp_fast entry (host buffers, PCIe included)", "n": n, "B": int(M.shape[0]), "ms": t * 1e3, "rows_per_s": n * M.shape[0] / t,
                      "pivots_mean": float(lcp.pivots.mean()), "ok": int(ok.sum())}))


if __name__ == "__main__":
    thd = [0.24 if w == 0 else 0.2 + 0.4 * synth.world_uniforms(w, 1)[0] for w in range(2048)]
    world_case("config 3: rimless wheel x2048", S.rimless_wheel_scene(), S.rimless_wheel_state(thd), 1e-3, 2000)
    st = np.repeat(S.box_state(pos=(0.0, 0.50001, 0.0)), 1024, axis=0)
    world_case("sitting box x1024 (n = 40)", S.box_scene(), st, 1e-3, 500)
    sts = []
    for w in range(1024):
        u = synth.world_uniforms(w, 10)
        sts.append(S.box_state(pos=(0.0, 0.9 + u[0], 0.0), quat=(u[1] - 0.5, u[2] - 0.5, u[3] - 0.5, 0.5 + u[4]), v=(u[5] - 0.5, 0.0, u[6] - 0.5),
                               w=(4 * u[7] - 2, 4 * u[8] - 2, 4 * u[9] - 2))[0])
    world_case("tumbling dice x1024", S.box_scene(mu_coulomb=0.5, epsilon=0.3, nk=4, cstab_max_iterations=10), np.array(sts), 1e-3, 1000)
    lcp_case(42, 4096); lcp_case(128, 256); lcp_case(256, 256)
