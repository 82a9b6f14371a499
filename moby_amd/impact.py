"""Host-side mirror of the batched impact handler (include/moby_hip_impact.h).

``ImpactBatch(B, nb, nc, nk, mass, inertia).process(state, contacts)`` plays the role of
``ImpactConstraintHandler::process_constraints`` (include/Moby/ImpactConstraintHandler.h:47) on every
world's contact list; ``box_stack`` generates BASELINE config 4's scene (the pattern of
/root/reference/example/stacks/stack.xml:6-12,36-96 extended to any height).
"""
import ctypes

import numpy as np

from . import _lib
from . import scene as S

CONTACT_DTYPE = np.dtype([("point", np.float64, 3), ("normal", np.float64, 3), ("body1", np.int32), ("body2", np.int32),
                          ("mu_coulomb", np.float64), ("mu_viscous", np.float64), ("epsilon", np.float64),
                          ("compliance", np.float64), ("nk", np.int32), ("pad", np.int32)], align=True)
assert CONTACT_DTYPE.itemsize == 96


MH_IMPACT_MODEL_DS, MH_IMPACT_MODEL_AP = 0, 1      # moby_hip_impact.h: the reference's default build / its -DUSE_AP build


def ap_lcp_size(nc, nk):
    """Anitescu-Potra LCP: 5 NC variables + NK_DIRS friction rows (ImpactConstraintHandlerLCP.cpp:105-118)."""
    return 5 * nc + nc * ((nk + 4) // 4 if nk > 4 else 1)


def lcp_size(nc, nk):
    """n_imp = 6 NC + NC NK/2 (5 NC variables + NC normal rows + NK/2 friction-polygon rows per contact)."""
    return 6 * nc + nc * (nk // 2)


def box_stack(nboxes, B=1, dt=1e-3, mu=1e-4, epsilon=0.0, nk=4, perturb=True, seed0=0x4D4F4259):
    """nboxes boxes of side (1 - 0.005 k) x 1 x (1 - 0.005 k), density 10, centre y = 0.5 + k, on the plane y = 0;
    contacts = the 4 bottom corners of every box against what lies below (body1 = the box above, normal +y).
    Velocities: one free-fall step (v_y = -g dt); worlds > 0 get small seeded lateral / angular perturbations.
    Returns (mass (nb,), inertia (nb, 3), state (B, nb*13), contacts (B, 4*nboxes) structured)."""
    nb, nc = nboxes, 4 * nboxes
    mass = np.zeros(nb); inertia = np.zeros((nb, 3))
    state = np.zeros((B, nb, S.MH_BODY_STATE)); cs = np.zeros((B, nc), dtype=CONTACT_DTYPE)
    for k in range(nb):
        x = z = 1.0 - 0.005 * k; y = 1.0
        m = 10.0 * (x * y * z); M = m / 12.0       # BoxPrimitive::calc_mass_properties (BoxPrimitive.cpp:692-712)
        mass[k] = m; inertia[k] = (M * (y * y + z * z), M * (x * x + z * z), M * (x * x + y * y))
        state[:, k, 1] = 0.5 + k; state[:, k, 6] = 1.0; state[:, k, 8] = -9.81 * dt
        h = 0.5 * x
        for c, (sx, sz) in enumerate(((1, 1), (1, -1), (-1, 1), (-1, -1))):
            i = 4 * k + c
            cs["point"][:, i] = (sx * h, float(k), sz * h)
            cs["normal"][:, i] = (0.0, 1.0, 0.0)
            cs["body1"][:, i] = k; cs["body2"][:, i] = k - 1 if k > 0 else nb
    cs["mu_coulomb"] = mu; cs["epsilon"] = epsilon; cs["nk"] = nk
    if perturb and B > 1:
        rng = np.random.default_rng(seed0)
        state[1:, :, 7] += rng.uniform(-1e-3, 1e-3, (B - 1, nb))
        state[1:, :, 9] += rng.uniform(-1e-3, 1e-3, (B - 1, nb))
        state[1:, :, 8] += rng.uniform(-1e-3, 0.0, (B - 1, nb))
        state[1:, :, 10:13] += rng.uniform(-1e-3, 1e-3, (B - 1, nb, 3))
    return mass, inertia, state.reshape(B, nb * S.MH_BODY_STATE), cs


class ImpactBatch:
    """B worlds x (nb bodies, nc contacts) behind an ``mh_impact_batch`` handle (persistent handler state)."""

    def __init__(self, B, nb, nc, nk, mass, inertia, model=MH_IMPACT_MODEL_DS):
        lib = _lib.load()
        self.B, self.nb, self.nc, self.nk = int(B), int(nb), int(nc), int(nk)
        m = np.ascontiguousarray(mass, dtype=np.float64); J = np.ascontiguousarray(inertia, dtype=np.float64)
        assert m.shape == (nb,) and J.shape == (nb, 3)
        self.handle = ctypes.c_void_p()
        _lib.check(lib.mh_impact_batch_create(self.B, self.nb, self.nc, self.nk, m.ctypes.data, J.ctypes.data, ctypes.byref(self.handle)))
        self.n = lib.mh_impact_batch_lcp_size(self.handle)
        if model != MH_IMPACT_MODEL_DS:
            _lib.check(lib.mh_impact_batch_set_model(self.handle, int(model)))

    def upload(self, state, contacts):
        st = np.ascontiguousarray(state, dtype=np.float64); cs = np.ascontiguousarray(contacts)
        assert st.shape == (self.B, self.nb * S.MH_BODY_STATE) and cs.shape == (self.B, self.nc) and cs.dtype == CONTACT_DTYPE
        _lib.check(_lib.load().mh_impact_batch_upload(self.handle, st.ctypes.data, cs.ctypes.data))

    def process_async(self, stream=None):
        _lib.check(_lib.load().mh_impact_batch_process(self.handle, stream))

    def download(self):
        st = np.zeros((self.B, self.nb * S.MH_BODY_STATE)); imp = np.zeros((self.B, self.nc, 3))
        status = np.zeros(self.B, dtype=np.int32); piv = np.zeros(self.B, dtype=np.uint32); solves = np.zeros(self.B, dtype=np.int32)
        _lib.check(_lib.load().mh_impact_batch_download(self.handle, st.ctypes.data, imp.ctypes.data, status.ctypes.data,
                                                        piv.ctypes.data, solves.ctypes.data))
        return dict(state=st, impulses=imp, status=status, pivots=piv, solves=solves)

    def process(self, state, contacts):
        self.upload(state, contacts)
        self.process_async()
        return self.download()

    def debug_lcp(self):
        MM = np.zeros((self.B, self.n, self.n)); qq = np.zeros((self.B, self.n))
        _lib.check(_lib.load().mh_impact_batch_debug_lcp(self.handle, MM.ctypes.data, qq.ctypes.data))
        return np.transpose(MM, (0, 2, 1)).copy(), qq          # row-major M[b, r, c]

    def solver_state(self):
        """What a checkpoint must keep besides the body states: _zlast, its size, the rand() streams, status bits."""
        zl = np.zeros((self.B, self.n)); zs = np.zeros(self.B, dtype=np.int32)
        rng = np.zeros((self.B, S.MH_RAND_WORDS), dtype=np.uint32); status = np.zeros(self.B, dtype=np.int32)
        _lib.check(_lib.load().mh_impact_batch_save_solver_state(self.handle, zl.ctypes.data, zs.ctypes.data, rng.ctypes.data, status.ctypes.data))
        v = np.zeros((self.B, S.MH_NOSLIP_MAX)); vs = np.zeros(self.B, dtype=np.int32)
        _lib.check(_lib.load().mh_impact_batch_save_noslip_state(self.handle, v.ctypes.data, vs.ctypes.data))
        return dict(zlast=zl, zlast_size=zs, rng=rng, status=status, v=v, v_size=vs)

    def load_solver_state(self, ss):
        zl = np.ascontiguousarray(ss["zlast"], dtype=np.float64); zs = np.ascontiguousarray(ss["zlast_size"], dtype=np.int32)
        rng = np.ascontiguousarray(ss["rng"], dtype=np.uint32); status = np.ascontiguousarray(ss["status"], dtype=np.int32)
        assert zl.shape == (self.B, self.n) and rng.shape == (self.B, S.MH_RAND_WORDS)
        _lib.check(_lib.load().mh_impact_batch_load_solver_state(self.handle, zl.ctypes.data, zs.ctypes.data, rng.ctypes.data, status.ctypes.data))
        if "v" in ss:
            v = np.ascontiguousarray(ss["v"], dtype=np.float64); vs = np.ascontiguousarray(ss["v_size"], dtype=np.int32)
            assert v.shape == (self.B, S.MH_NOSLIP_MAX)
            _lib.check(_lib.load().mh_impact_batch_load_noslip_state(self.handle, v.ctypes.data, vs.ctypes.data))

    def lu_work(self, reset=False):
        """(B, 4): model flops and model bytes of the block solver's factorisations priced as dense dgesv calls, the flops its
        routines really issue, and the seconds its workgroups spent on the world's problems (mh_impact_batch_lu_work)."""
        w = np.zeros((self.B, 4))
        _lib.check(_lib.load().mh_impact_batch_lu_work(self.handle, w.ctypes.data, int(bool(reset))))
        return w

    def close(self):
        if self.handle:
            _lib.load().mh_impact_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
