import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from moby_amd import scene as S
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    sc = S.sphere_stack_scene()
    h = ctypes.c_void_p()
    lib.mh_world_batch_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    rc = lib.mh_world_batch_create(ctypes.addressof(sc), 64, ctypes.byref(h))
    lib.mh_world_batch_occupancy.argtypes = [ctypes.c_void_p]
    print(path, "create rc", rc, "occupancy", lib.mh_world_batch_occupancy(h))
