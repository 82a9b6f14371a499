"""ur10 x 8192, 200 steps of 5e-4 in one launch: the two forward-dynamics algorithms of RCArticulatedBody (eCRB: H + Cholesky,
eFeatherstone: the articulated-body recursion) on the device.  python tools/artic_algorithms.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moby_amd import artic as A

m, _, _ = A.load_sdf(os.path.join(ROOT, "tests", "scenes", "ten_joint_arm.sdf"))
B, steps = 8192, 200
rng = np.random.default_rng(0x4D4F4259)
lo = np.maximum(np.array(m.lolimit[:m.nj]), -np.pi); hi = np.minimum(np.array(m.hilimit[:m.nj]), np.pi)
q0 = rng.uniform(lo, hi, (B, m.nj)); qd0 = rng.uniform(-1.0, 1.0, (B, m.nj))
for name, alg in (("eCRB", A.MH_ARTIC_CRB), ("eFeatherstone", A.MH_ARTIC_FSAB)):
    mm = type(m).from_buffer_copy(m); mm.algorithm = alg
    ab = A.ArticBatch(mm, q0, qd0)
    s = torch.cuda.current_stream().cuda_stream
    ab.step(5e-4, 10, s); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ab.step(5e-4, steps, s); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    _, _, aux = ab.download(); ab.close()
    print(json.dumps({"algorithm": name, "worlds": B, "steps": steps, "ms": ms, "world_steps_per_sec": B * steps / (ms * 1e-3),
                      "worlds_with_errors": int(((aux["status"] & ~2) != 0).sum()), "limit_lcp_rows": int(aux["lcp_rows"].sum())}))
