"""Host-side mirror of ``Moby::LCP`` (include/Moby/LCP.h:17-58) over the C ABI.

Same method names, argument meaning and return convention as the reference
class -- ``lcp_fast(M, q, z, zero_tol)`` returns ``True``/``False`` and leaves
the solution in ``z`` -- but every method takes a *batch* of independent
problems (leading axis B; a single problem is B = 1).  Like the reference
object (which consumes the process-wide ``rand()``), an ``LCP`` instance owns
the libc random streams of its worlds; they start at ``srand(1)``.

Two flavours:
  * ``LCP``      numpy host arrays in/out (copies through the library),
  * ``LCPDevice`` torch CUDA(HIP) tensors, asynchronous on torch's current
    stream (used by bench.py so inputs are resident in HBM).
"""
import ctypes
import numpy as np

from . import _lib
from ._lib import (MH_LCP_FAST, MH_LCP_FAST_REG, MH_LCP_LEMKE, MH_LCP_LEMKE_REG,
                   MH_RAND_WORDS, mh_lcp_opts)


def rand_states(B, seed=1):
    """B copies of the glibc ``srand(seed)`` state (uint32[B, 32])."""
    lib = _lib.load()
    st = np.zeros(MH_RAND_WORDS, dtype=np.uint32)
    lib.mh_rand_seed(st.ctypes.data, seed)
    return np.tile(st, (B, 1))


def _ptr(a):
    return None if a is None else a.ctypes.data


class LCP:
    """Batch of B worlds, each with its own rand() stream."""

    def __init__(self, B=1, seed=1):
        self.B = B
        self.rng = rand_states(B, seed)
        self.pivots = np.zeros(B, dtype=np.uint32)  # LCP::pivots (LCP.h:30)
        self.z_size = None                          # z.size() after the last call
        self.trace = None
        self.trace_len = None

    # -- the four public solvers of include/Moby/LCP.h:21-27 -----------------
    def lcp_fast(self, M, q, z, zero_tol=-1.0, **kw):
        return self._solve(MH_LCP_FAST, M, q, z, None, zero_tol=zero_tol, **kw)

    def lcp_fast_regularized(self, M, q, z, min_exp=-20, step_exp=4, max_exp=20, piv_tol=-1.0, zero_tol=-1.0, **kw):
        return self._solve(MH_LCP_FAST_REG, M, q, z, (min_exp, step_exp, max_exp), piv_tol, zero_tol, **kw)

    def lcp_lemke(self, M, q, z, piv_tol=-1.0, zero_tol=-1.0, **kw):
        return self._solve(MH_LCP_LEMKE, M, q, z, None, piv_tol, zero_tol, **kw)

    def lcp_lemke_regularized(self, M, q, z, min_exp=-20, step_exp=1, max_exp=1, piv_tol=-1.0, zero_tol=-1.0, **kw):
        return self._solve(MH_LCP_LEMKE_REG, M, q, z, (min_exp, step_exp, max_exp), piv_tol, zero_tol, **kw)

    # ------------------------------------------------------------------------
    def _solve(self, kind, M, q, z, exps, piv_tol=-1.0, zero_tol=-1.0, z_size=None, trace_cap=0):
        """M: (B, n, n) with M[b] holding the matrix in *row-major numpy order*
        (M[b][r, c]); q: (B, n); z: (B, n) float64, overwritten.
        z_size: None (all worlds warm: z.size()==n) or int array (B,).
        Returns a bool array (B,)."""
        lib = _lib.load()
        M = np.asarray(M, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64)
        if M.ndim == 2:
            M = M[None]; q = q.reshape(1, -1)
        B, n = q.shape
        if B != self.B:
            raise ValueError("batch %d != LCP batch %d" % (B, self.B))
        if z.dtype != np.float64 or z.shape != (B, n) or not z.flags.c_contiguous:
            raise ValueError("z must be a C-contiguous float64 array of shape (B, n)")
        if n == 0:  # LCP.cpp:49-54,218-222,557-561
            self.z_size = np.zeros(B, dtype=np.int32)
            return np.ones(B, dtype=bool)
        # column-major per problem, as Ravelin::MatrixNd stores it
        Mcm = np.ascontiguousarray(np.transpose(M, (0, 2, 1)))
        opts = None
        if exps is not None or piv_tol > 0 or zero_tol > 0:
            e = exps if exps is not None else (-20, 1, 1)
            opts = mh_lcp_opts(int(e[0]), int(e[1]), int(e[2]), float(piv_tol), float(zero_tol))
        zs_in = None if z_size is None else np.ascontiguousarray(z_size, dtype=np.int32)
        zs_out = np.zeros(B, dtype=np.int32)
        status = np.zeros(B, dtype=np.int32)
        tr = tl = None
        if trace_cap > 0:
            tr = np.zeros((B, trace_cap), dtype=np.int32)
            tl = np.zeros(B, dtype=np.int32)
        rc = lib.mh_lcp_solve_batch(kind, B, n, Mcm.ctypes.data, n, n * n, q.ctypes.data, z.ctypes.data,
                                    _ptr(zs_in), zs_out.ctypes.data, self.rng.ctypes.data,
                                    status.ctypes.data, self.pivots.ctypes.data,
                                    _ptr(tr), trace_cap, _ptr(tl),
                                    ctypes.byref(opts) if opts is not None else None)
        _lib.check(rc)
        self.z_size = zs_out
        self.trace, self.trace_len = tr, tl
        return status.astype(bool)


class LCPDevice:
    """Same solvers on torch device tensors; nothing is copied or synchronised.

    M: (B, n, n) float64 *column-major per problem* (M[b, c, r] = M_b(r, c)),
    q/z: (B, n) float64, rng: (B, 32) int32/uint32-as-int32, status: (B,) int32.
    """

    def __init__(self, B, device="cuda", seed=1):
        import torch
        self.torch = torch
        self.B = B
        self.device = torch.device(device)
        self.rng = torch.from_numpy(rand_states(B, seed).view(np.int32)).to(self.device)
        self.status = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.pivots = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.z_size = torch.zeros(B, dtype=torch.int32, device=self.device)

    def solve(self, kind, M, q, z, opts=None, z_size_in=None):
        lib = _lib.load()
        torch = self.torch
        B, n = q.shape
        assert M.is_contiguous() and q.is_contiguous() and z.is_contiguous()
        assert M.dtype == torch.float64 and M.shape == (B, n, n)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = lib.mh_lcp_solve_batch_dev(stream, kind, B, n, M.data_ptr(), n, n * n, q.data_ptr(), z.data_ptr(),
                                        None if z_size_in is None else z_size_in.data_ptr(), self.z_size.data_ptr(),
                                        self.rng.data_ptr(), self.status.data_ptr(), self.pivots.data_ptr(),
                                        None, 0, None, ctypes.byref(opts) if opts is not None else None)
        _lib.check(rc)
        return self.status
