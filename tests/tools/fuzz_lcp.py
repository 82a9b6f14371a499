"""One-off fuzz of the LCP entry (wave and block solvers) against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moby_amd import synth
from tests.oracle_api import Oracle
import tests.test_lcp_gpu as T
o = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 80):
    big = len(sys.argv) > 3 and sys.argv[3] == "big"        # "big": the block solver's sizes up to 420 (the structure-exploiting LU of Lemke's bases)
    n = (int(rng.integers(65, 420)) if big else (int(rng.integers(1, 65)) if rng.random() < 0.7 else int(rng.integers(65, 180))))
    fam = str(rng.choice(["pd", "psd", "copos"])); kind = int(rng.integers(0, 4)); seed = int(rng.integers(0, 10**6))
    if fam == "copos" and n < 2: fam = "pd"
    M, q = synth.random_lcp(2, n, fam, seed=seed)
    zs = np.array([0, n], dtype=np.int32) if rng.random() < 0.5 else np.zeros(2, dtype=np.int32)
    z0 = np.abs(rng.standard_normal((2, n))) * (rng.random((2, n)) < 0.3)
    try:
        T.assert_parity(o, kind, M, q, z0=z0, z_size=zs, seed=int(rng.integers(1, 50)))
        print("ok   kind %d n %3d %s" % (kind, n, fam))
    except AssertionError as e:
        bad += 1; print("FAIL kind %d n %3d %s seed %d: %s" % (kind, n, fam, seed, str(e).split("\n")[0][:100]))
print("failures:", bad)
