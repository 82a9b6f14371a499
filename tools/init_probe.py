import os, sys, ctypes, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
if "make" in mode: subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
if "oracle" in mode:
    from tests.oracle_api import Oracle
    o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
if "run" in mode:
    from moby_amd import scene as S
    sc = S.sphere_stack_scene(); st = S.sphere_stack_state_range(0, 4); aux = S.new_aux(4)
    o.world_step_batch(sc, st, aux, 1e-3, 10)
import torch
print(mode, "avail", torch.cuda.is_available())
torch.cuda.set_device(0)
from moby_amd import _lib
print(mode, "mh_device_count", _lib.load().mh_device_count())
