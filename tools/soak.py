"""Long-horizon soak of the two batch configurations: status bits, invariants, throughput."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moby_amd import scene as S, synth
from moby_amd.world import WorldBatchDevice
import torch

if __name__ == "__main__":
    B, n = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    sc = S.sphere_stack_scene()
    wb = WorldBatchDevice(sc, S.sphere_stack_state_range(0, B))
    t0 = time.perf_counter(); wb.step(1e-3, n); torch.cuda.synchronize(); t = time.perf_counter() - t0
    st, aux = wb.download()
    z = st.reshape(B, 3, 13)[:, :, 2]
    print("sphere-stack x%d, %d steps: %.2f s (%.3g world-steps/s); status bits seen: %s; z ranges %s; stab iters/world %.1f; max mini-steps/step %.2f"
          % (B, n, t, B * n / t, sorted(set(aux["status"].tolist())), [(round(z[:, k].min(), 6), round(z[:, k].max(), 6)) for k in range(3)],
             aux["stab_iters"].mean(), aux["mini_steps"].max() / n))
    wb.close()
    B2, n2 = 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    thd = [0.24 if w == 0 else 0.2 + 0.4 * synth.world_uniforms(w, 1)[0] for w in range(B2)]
    wb = WorldBatchDevice(S.rimless_wheel_scene(), S.rimless_wheel_state(thd))
    t0 = time.perf_counter(); wb.step(1e-3, n2); torch.cuda.synchronize(); t = time.perf_counter() - t0
    st, aux = wb.download()
    print("rimless wheel x%d, %d steps: %.2f s (%.3g world-steps/s); status bits seen: %s; |y| max %.2e; x range (%.2f, %.2f); z min %.4f"
          % (B2, n2, t, B2 * n2 / t, sorted(set(aux["status"].tolist())), np.abs(st[:, 1]).max(), st[:, 0].min(), st[:, 0].max(), st[:, 2].min()))
