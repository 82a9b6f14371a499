import sys,time,numpy as np
sys.path.insert(0,'/root/repo')
from moby_amd import scene as S
from moby_amd.world import WorldBatchDevice
import torch
sc=S.sphere_stack_scene()
B=4096
for name,st in (("perturbed", S.sphere_stack_state_range(0,B)), ("identical (world 7)", np.repeat(S.sphere_stack_state_range(7,1),B,axis=0)), ("identical (world 0)", np.repeat(S.sphere_stack_state_range(0,1),B,axis=0))):
    wb=WorldBatchDevice(sc,st); wb.step(1e-3,20); torch.cuda.synchronize()
    t0=time.perf_counter(); wb.step(1e-3,200); torch.cuda.synchronize(); t=time.perf_counter()-t0
    _,a=wb.download()
    print(name, "%.3f ms"%(t*1e3), "pivots/world", a['lcp_pivots'].mean(), "min/max", a['lcp_pivots'].min(), a['lcp_pivots'].max(), "solves", a['lcp_solves'].mean())
    wb.close()
