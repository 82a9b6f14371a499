"""Sizes of the nonbasic blocks lcp_fast factorises (oracle_dbg_lu_hist) in 32 worlds of the sphere-stack batch at several points of a long\nrun: how many of them fit the LDS-resident LU block of the one-wavefront kernel (MHW_KA_V).   python tests/tools/lu_size_hist.py"""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from moby_amd import scene as S
from moby_amd.world import WorldBatchDevice
from tests.oracle_api import Oracle
sc = S.sphere_stack_scene()
B = 1024
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
for start in (20, 1000, 2000, 3000, 4200):
    wb = WorldBatchDevice(sc, S.sphere_stack_state_range(0, B))
    left = start
    while left > 0:
        k = min(left, 1000); wb.step(1e-3, k); torch.cuda.synchronize(); left -= k
    st0, a0 = wb.download(); wb.close()
    h0 = np.zeros(130, dtype=np.uint64); o.lib.oracle_dbg_lu_hist(h0.ctypes.data_as(ctypes.c_void_p))
    s = st0[100:132].copy(); a = a0[100:132].copy()
    o.world_step_batch(sc, s, a, 1e-3, 50)
    h1 = np.zeros(130, dtype=np.uint64); o.lib.oracle_dbg_lu_hist(h1.ctypes.data_as(ctypes.c_void_p))
    h = (h1 - h0).astype(np.int64)
    fast = h[:65]; tot = fast.sum()
    cum = np.cumsum(fast) / max(tot, 1)
    print(start, "lcp_fast LUs %d: k<=8 %.2f, k<=12 %.2f, k<=16 %.2f, k<=20 %.2f, k<=24 %.2f; mean k %.1f; lemke LUs %d" % (tot, cum[8], cum[12], cum[16], cum[20], cum[24], (np.arange(65) * fast).sum() / max(tot, 1), h[65:].sum()))
