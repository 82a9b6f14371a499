import multiprocessing as mp, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def w(i): return i
n = int(sys.argv[1])
mode = sys.argv[2] if len(sys.argv) > 2 else "pool"
if mode == "pool":
    with mp.get_context("fork").Pool(n) as p: p.map(w, range(n))
import torch
print("n", n, "torch.cuda.is_available", torch.cuda.is_available(), "count", torch.cuda.device_count())
from moby_amd import _lib
print("mh_device_count", _lib.load().mh_device_count())
