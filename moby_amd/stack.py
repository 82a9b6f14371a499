"""Host-side mirror of the large-world stepper (include/moby_hip_stack.h).

``BigScene`` holds the body tables and the candidate-pair list of one scene topology (any number of free
boxes / spheres over one static plane); ``box_stack_scene`` generates BASELINE config 4's scene -- the pattern of
/root/reference/example/stacks/stack.xml:6-12,36-96 extended to any height; ``BigBatch`` plays the
role of B ``TimeSteppingSimulator`` objects (``step``) and of ``ConstraintStabilization::stabilize`` (``stabilize``).
"""
import ctypes

import numpy as np

from . import _lib
from . import scene as S

MH_PAIR_CLOSED_FORM, MH_PAIR_VERTEX_FACE = 0, 1
MH_IJOINT_SPHERICAL, MH_IJOINT_REVOLUTE, MH_IJOINT_FIXED, MH_IJOINT_PLANAR, MH_IJOINT_UNIVERSAL, MH_IJOINT_PRISMATIC = 0, 1, 2, 3, 4, 5        # moby_hip_stack.h
IJOINT_ROWS = {MH_IJOINT_SPHERICAL: 3, MH_IJOINT_REVOLUTE: 5, MH_IJOINT_FIXED: 6, MH_IJOINT_PLANAR: 3, MH_IJOINT_UNIVERSAL: 4, MH_IJOINT_PRISMATIC: 5}
_dp, _ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)


class mh_big_scene(ctypes.Structure):
    _fields_ = [
        ("nb", ctypes.c_int), ("has_ground", ctypes.c_int),
        ("geom_type", _ip), ("geom_dim", _dp), ("mass", _dp), ("inertia", _dp),
        ("plane_R", ctypes.c_double * 9), ("plane_o", ctypes.c_double * 3), ("gravity", ctypes.c_double * 3),
        ("npairs", ctypes.c_int),
        ("pair_a", _ip), ("pair_b", _ip), ("pair_model", _ip),
        ("cp_epsilon", _dp), ("cp_mu_coulomb", _dp), ("cp_mu_viscous", _dp), ("cp_compliance", _dp),
        ("nk", ctypes.c_int),
        ("min_step_size", ctypes.c_double), ("contact_dist_thresh", ctypes.c_double), ("cstab_eps", ctypes.c_double),
        ("cstab_max_iterations", ctypes.c_uint), ("lcp_n_max", ctypes.c_int), ("impact_model", ctypes.c_int),
        ("njoints", ctypes.c_int), ("joint_type", _ip), ("joint_inboard", _ip), ("joint_outboard", _ip),
        ("joint_anchor_in", _dp), ("joint_anchor_out", _dp), ("joint_vec_in", _dp), ("joint_vec_out", _dp),
    ]


class BigScene:
    """numpy tables + the ctypes record pointing into them (keep this object alive while the record is in use)."""

    def __init__(self, geom_type, geom_dim, mass, inertia, pairs, gravity, plane_R=None, plane_o=(0.0, 0.0, 0.0), has_ground=True,
                 nk=4, epsilon=0.0, mu_coulomb=0.0, mu_viscous=0.0, compliance=0.0,
                 cstab_max_iterations=S.MH_CSTAB_DEFAULT_MAX_ITERATIONS, lcp_n_max=0, impact_model=0, joints=()):
        self.geom_type = np.ascontiguousarray(geom_type, dtype=np.int32)
        self.nb = len(self.geom_type)
        self.geom_dim = np.ascontiguousarray(geom_dim, dtype=np.float64).reshape(self.nb, 3)
        self.mass = np.ascontiguousarray(mass, dtype=np.float64)
        self.inertia = np.ascontiguousarray(inertia, dtype=np.float64).reshape(self.nb, 3)
        pairs = sorted(pairs)                                  # (a, b, model), a < b, lexicographic = canonical order
        self.pair_a = np.array([p[0] for p in pairs], dtype=np.int32)
        self.pair_b = np.array([p[1] for p in pairs], dtype=np.int32)
        self.pair_model = np.array([p[2] for p in pairs], dtype=np.int32)
        npairs = len(pairs)
        full = lambda v: np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), (npairs,)))
        self.cp_epsilon, self.cp_mu_coulomb = full(epsilon), full(mu_coulomb)
        self.cp_mu_viscous, self.cp_compliance = full(mu_viscous), full(compliance)
        c = mh_big_scene()
        c.nb, c.has_ground, c.npairs, c.nk = self.nb, 1 if has_ground else 0, npairs, int(nk)
        P = lambda a, t: a.ctypes.data_as(t)
        c.geom_type, c.geom_dim, c.mass, c.inertia = P(self.geom_type, _ip), P(self.geom_dim, _dp), P(self.mass, _dp), P(self.inertia, _dp)
        c.pair_a, c.pair_b, c.pair_model = P(self.pair_a, _ip), P(self.pair_b, _ip), P(self.pair_model, _ip)
        c.cp_epsilon, c.cp_mu_coulomb = P(self.cp_epsilon, _dp), P(self.cp_mu_coulomb, _dp)
        c.cp_mu_viscous, c.cp_compliance = P(self.cp_mu_viscous, _dp), P(self.cp_compliance, _dp)
        R = np.eye(3) if plane_R is None else np.asarray(plane_R, dtype=np.float64).reshape(3, 3)
        for k in range(9):
            c.plane_R[k] = R.flat[k]
        for k in range(3):
            c.plane_o[k] = plane_o[k]; c.gravity[k] = gravity[k]
        c.min_step_size = S.NEAR_ZERO; c.contact_dist_thresh = 1e-6; c.cstab_eps = S.NEAR_ZERO
        c.cstab_max_iterations = int(cstab_max_iterations); c.lcp_n_max = int(lcp_n_max)
        c.impact_model = int(impact_model)                     # MH_IMPACT_MODEL_DS / _AP (moby_hip_impact.h)
        # implicit joints: records of make_joint() (the simulator's <ImplicitConstraint> list, in that order)
        nj = len(joints)
        self.joint_type = np.array([j["type"] for j in joints], dtype=np.int32)
        self.joint_inboard = np.array([j["inboard"] for j in joints], dtype=np.int32)
        self.joint_outboard = np.array([j["outboard"] for j in joints], dtype=np.int32)
        arr = lambda key, w: np.ascontiguousarray(np.array([j[key] for j in joints], dtype=np.float64).reshape(nj, w))
        self.joint_anchor_in, self.joint_anchor_out = arr("anchor_in", 3), arr("anchor_out", 3)
        self.joint_vec_in, self.joint_vec_out = arr("vec_in", 9), arr("vec_out", 9)
        c.njoints = nj
        c.joint_type, c.joint_inboard, c.joint_outboard = P(self.joint_type, _ip), P(self.joint_inboard, _ip), P(self.joint_outboard, _ip)
        c.joint_anchor_in, c.joint_anchor_out = P(self.joint_anchor_in, _dp), P(self.joint_anchor_out, _dp)
        c.joint_vec_in, c.joint_vec_out = P(self.joint_vec_in, _dp), P(self.joint_vec_out, _dp)
        self.c = c

    @classmethod
    def from_scene(cls, sc, state, lcp_n_max=0):
        """The large-world scene that says what an ``mh_scene`` (moby_amd/io.py: ``load_xml``, i.e. XMLReader::read,
        /root/reference/src/XMLReader.cpp:151-204) says, for worlds the one-wavefront stepper does not take: stacked boxes such as
        example/stacks/stack.xml.  Every enabled pair becomes a candidate pair in the canonical order; a box-box pair is modelled as
        MH_PAIR_VERTEX_FACE (moby_hip_stack.h: the higher-id box on the +Y face of the lower-id one) when the two bounding spheres
        of the INITIAL state (``state``: nb x 13) reach each other -- the pairs CCD::broad_phase (src/CCD.cpp:702-988) could ever
        report for a resting stack -- and is left out otherwise (a box two storeys up never meets the box below).  Box-sphere pairs
        are refused: their contact generation is not built."""
        nb = sc.nb; ntot = nb + (1 if sc.has_ground else 0)
        st = np.asarray(state, dtype=np.float64).reshape(-1)[:nb * S.MH_BODY_STATE].reshape(nb, S.MH_BODY_STATE)
        gt = [int(sc.geom_type[b]) for b in range(nb)]
        dims = np.array([[sc.geom_dim[b][k] for k in range(3)] for b in range(nb)])
        rad = [0.5 * float(np.linalg.norm(dims[b])) if gt[b] == S.MH_GEOM_BOX else float(dims[b][0]) for b in range(nb)]
        pairs, idx = [], []
        for i in range(ntot):
            for j in range(i + 1, ntot):
                pi = S.pair_index(i, j, ntot)
                if not sc.pair_enabled[pi]:
                    continue
                model = MH_PAIR_CLOSED_FORM
                if j < nb and (gt[i] == S.MH_GEOM_BOX or gt[j] == S.MH_GEOM_BOX):
                    if gt[i] != gt[j]:
                        raise ValueError("bodies %d, %d: box-sphere contact is not built (disable the pair)" % (i, j))
                    if float(np.linalg.norm(st[i, :3] - st[j, :3])) > rad[i] + rad[j]:
                        continue
                    model = MH_PAIR_VERTEX_FACE
                pairs.append((i, j, model)); idx.append(pi)
        nks = set(int(sc.cp_nk[p]) for p in idx)
        if len(nks) > 1:
            raise ValueError("the large-world stepper takes ONE friction-cone-edges value per scene, got %r" % sorted(nks))
        col = lambda a: [float(a[p]) for p in idx]
        return cls(gt, dims, [sc.mass[b] for b in range(nb)], [[sc.inertia[b][k] for k in range(3)] for b in range(nb)], pairs,
                   gravity=[sc.gravity[k] for k in range(3)], plane_R=[sc.plane_R[k] for k in range(9)], plane_o=[sc.plane_o[k] for k in range(3)],
                   has_ground=bool(sc.has_ground), nk=(nks.pop() if nks else 4), epsilon=col(sc.cp_epsilon), mu_coulomb=col(sc.cp_mu_coulomb),
                   mu_viscous=col(sc.cp_mu_viscous), compliance=col(sc.cp_compliance), cstab_max_iterations=sc.cstab_max_iterations,
                   lcp_n_max=lcp_n_max)

    @property
    def npairs(self):
        return len(self.pair_a)

    def lcp_capacity(self):
        """The capacity mh_big_batch_create gives the handlers' LCPs (mh_big_batch_lcp_capacity): lcp_n_max, or when that is 0
        the rule of mh_big.hip -- 4 contacts per pair with a box, 3 per pin, 1 otherwise, doubled (a box inside the tolerance band
        can show all 8 vertices), at least 8, at most MH_BIG_MAX_CONTACTS; n = 6 nc + nc nk/2, at most MH_LCP_MAX_N_BLOCK."""
        if self.c.lcp_n_max:
            return int(self.c.lcp_n_max)
        nb = len(self.geom_type)
        nc = 0
        for a, b in zip(self.pair_a, self.pair_b):
            box = self.geom_type[a] == S.MH_GEOM_BOX or (b < nb and self.geom_type[b] == S.MH_GEOM_BOX)
            nc += 4 if box else (3 if self.geom_type[a] == S.MH_GEOM_PIN else 1)
        nc = min(512, max(8, 2 * nc))
        return min(4096, 6 * nc + nc * (self.c.nk // 2))


def make_joint(kind, inboard, outboard, location, state, nb, axis=(0.0, 0.0, 1.0), axis2=None):
    """An implicit joint as the XML states it (<RevoluteJoint location= axis= inboard-link-id= outboard-link-id=>: global
    location and axis at the bodies' reference poses) turned into the body-frame data of mh_big_scene.  ``state``: the
    reference poses (nb x 13); a link id of nb (or -1) is the static world."""
    st = np.asarray(state, dtype=np.float64).reshape(nb, S.MH_BODY_STATE)

    def frame(b):
        if b < 0 or b >= nb:
            return np.eye(3), np.zeros(3)
        x, y, z, w = st[b, 3:7]
        return S.quat_to_R((x, y, z, w)), st[b, 0:3]
    inboard = nb if inboard < 0 else inboard
    outboard = nb if outboard < 0 else outboard
    Ri, xi = frame(inboard); Ro, xo = frame(outboard)
    p = np.asarray(location, dtype=np.float64)
    a = np.asarray(axis, dtype=np.float64); a = a / np.linalg.norm(a)
    vin = np.zeros((3, 3)); vout = np.zeros((3, 3))
    if kind == MH_IJOINT_REVOLUTE:
        from .synth import orthonormal_basis
        v1, v2 = orthonormal_basis(a)                               # two directions orthogonal to the axis
        vin[0] = Ri.T @ a; vin[1] = Ri.T @ a
        vout[0] = Ro.T @ np.asarray(v1); vout[1] = Ro.T @ np.asarray(v2)
    elif kind == MH_IJOINT_PRISMATIC:                               # axis = the sliding direction
        from .synth import orthonormal_basis
        t1, t2 = (np.asarray(v) for v in orthonormal_basis(a))
        tri = [t1, t2, a]                                           # a_0, a_1: the two position rows' directions; a_k . b_k = 0 with b_k = a_{k+1}
        for k in range(3):
            vin[k] = Ri.T @ tri[k]; vout[k] = Ro.T @ tri[(k + 1) % 3]
    elif kind == MH_IJOINT_UNIVERSAL:                               # axis (inboard) and axis2 (outboard) stay orthogonal
        from .synth import orthonormal_basis
        a2 = np.asarray(orthonormal_basis(a)[0] if axis2 is None else axis2, dtype=np.float64)
        a2 = a2 - a * (a @ a2); a2 = a2 / np.linalg.norm(a2)
        vin[0] = Ri.T @ a; vout[0] = Ro.T @ a2
    elif kind == MH_IJOINT_PLANAR:                                  # axis = the plane's normal (<PlanarJoint normal=...>)
        from .synth import orthonormal_basis
        t1, t2 = orthonormal_basis(a)
        vin[0] = Ri.T @ np.asarray(t1); vin[1] = Ri.T @ np.asarray(t2); vin[2] = Ri.T @ a
        vout[0] = Ro.T @ a; vout[1] = Ro.T @ a
    elif kind == MH_IJOINT_FIXED:
        e = np.eye(3)
        for k in range(3):                                          # e_k (inboard) stays orthogonal to e_{k+1} (outboard)
            vin[k] = Ri.T @ e[k]; vout[k] = Ro.T @ e[(k + 1) % 3]
    return dict(type=int(kind), inboard=int(inboard), outboard=int(outboard), anchor_in=Ri.T @ (p - xi), anchor_out=Ro.T @ (p - xo),
                vec_in=vin.reshape(9), vec_out=vout.reshape(9))


def pendulum_scene(cstab_max_iterations=25):
    """example/contact-constrained-pendulum/contact-constrained-pendulum.xml: one free body (mass 1, the inertia of a sphere of
    radius 1.5811) whose point (0, 1, 0) the collision plugin pins to the global origin with six frictionless contacts
    (epsilon 0, mu 0, NK 4: a 48-row impact LCP every step), gravity (0, -9.81, 0), 25 stabilisation iterations."""
    r = 1.5811
    J = r * r * 1.0 * 2.0 / 5.0                                      # SpherePrimitive.cpp:149
    return BigScene([S.MH_GEOM_PIN], [(0.0, 1.0, 0.0)], [1.0], [(J, J, J)], [(0, 1, MH_PAIR_CLOSED_FORM)], gravity=(0.0, -9.81, 0.0),
                    nk=4, epsilon=0.0, mu_coulomb=0.0, cstab_max_iterations=cstab_max_iterations, lcp_n_max=64)


def pendulum_state(B=1, seed0=0x4D4F4259):
    """position 1 0 0, rpy 0 0 pi/2 (the XML's numbers), at rest; worlds > 0 start with a small seeded spin about z."""
    st = np.zeros((B, S.MH_BODY_STATE))
    st[:, 0] = 1.0
    h = 1.57079632679490 / 2.0
    st[:, 5] = np.sin(h); st[:, 6] = np.cos(h)
    if B > 1:
        w = np.random.default_rng(seed0).uniform(-0.5, 0.5, B - 1)
        st[1:, 12] = w; st[1:, 8] = w * 1.0                             # v = omega x r with r = (1, 0, 0): rotation about the pin
    return st


def box_dims(k):
    """Box k of the stack: (1 - 0.005 k) x 1 x (1 - 0.005 k) (stack.xml:6-12 extended)."""
    s = 1.0 - 0.005 * k
    return s, 1.0, s


def box_stack_scene(nboxes, mu=1e-4, epsilon=0.0, nk=4, cstab_max_iterations=S.MH_CSTAB_DEFAULT_MAX_ITERATIONS, lcp_n_max=None,
                    impact_model=0):
    """nboxes boxes, density 10, centre y = 0.5 + k, on the plane y = 0, gravity (0, -9.81, 0).  Candidate pairs: the
    ground with every box (its DummyBV is infinite, CCD.cpp:1091-1094) and each box with the one above it (the only
    box pairs whose bounding spheres of radius ~0.87 can overlap at a centre distance of 1), as vertex-face pairs."""
    gt = np.full(nboxes, S.MH_GEOM_BOX, dtype=np.int32)
    dims = np.array([box_dims(k) for k in range(nboxes)])
    mass = 10.0 * dims.prod(axis=1)                                              # BoxPrimitive.cpp:692-712
    M = mass / 12.0
    x, y, z = dims.T
    inertia = np.stack([M * (y * y + z * z), M * (x * x + z * z), M * (x * x + y * y)], axis=1)
    if lcp_n_max is None:
        # every interface (ground-box 0 and box k - box k+1) with its 4 corner contacts: n = nboxes * 4 * (6 + nk/2)
        lcp_n_max = max(64, nboxes * 4 * (6 + nk // 2))
    pairs = [(k, nboxes, MH_PAIR_CLOSED_FORM) for k in range(nboxes)] + [(k, k + 1, MH_PAIR_VERTEX_FACE) for k in range(nboxes - 1)]
    return BigScene(gt, dims, mass, inertia, pairs, gravity=(0.0, -9.81, 0.0), nk=nk, epsilon=epsilon, mu_coulomb=mu,
                    cstab_max_iterations=cstab_max_iterations, lcp_n_max=lcp_n_max, impact_model=impact_model)


def box_stack_state(nboxes, B=1, perturb=True, seed0=0x4D4F4259):
    """Boxes at rest, exactly touching; worlds > 0 get small seeded velocity perturbations."""
    st = np.zeros((B, nboxes, S.MH_BODY_STATE))
    st[:, :, 1] = 0.5 + np.arange(nboxes)
    st[:, :, 6] = 1.0
    if perturb and B > 1:
        rng = np.random.default_rng(seed0)
        st[1:, :, 7] += rng.uniform(-1e-3, 1e-3, (B - 1, nboxes))
        st[1:, :, 9] += rng.uniform(-1e-3, 1e-3, (B - 1, nboxes))
        st[1:, :, 8] += rng.uniform(-1e-3, 0.0, (B - 1, nboxes))
        st[1:, :, 10:13] += rng.uniform(-1e-3, 1e-3, (B - 1, nboxes, 3))
    return st.reshape(B, nboxes * S.MH_BODY_STATE)


class BigBatch:
    """B large worlds behind an ``mh_big_batch`` handle."""

    def __init__(self, scene, state, aux=None):
        lib = _lib.load()
        self.scene = scene
        st = np.ascontiguousarray(state, dtype=np.float64)
        self.B = st.shape[0]
        assert st.shape == (self.B, scene.nb * S.MH_BODY_STATE)
        self.handle = ctypes.c_void_p()
        _lib.check(lib.mh_big_batch_create(ctypes.byref(scene.c), self.B, ctypes.byref(self.handle)))
        self.cap = lib.mh_big_batch_lcp_capacity(self.handle)
        self.upload(st, aux)

    def upload(self, state, aux=None):
        st = np.ascontiguousarray(state, dtype=np.float64)
        a = None if aux is None else np.ascontiguousarray(aux)
        _lib.check(_lib.load().mh_big_batch_upload(self.handle, st.ctypes.data, None if a is None else a.ctypes.data))

    def step(self, dt, nsteps=1, stream=None):
        _lib.check(_lib.load().mh_big_batch_step(self.handle, stream, float(dt), int(nsteps)))

    def stabilize(self, stream=None):
        _lib.check(_lib.load().mh_big_batch_stabilize(self.handle, stream))

    def download(self):
        st = np.zeros((self.B, self.scene.nb * S.MH_BODY_STATE)); aux = np.zeros(self.B, dtype=S.AUX_DTYPE)
        _lib.check(_lib.load().mh_big_batch_download(self.handle, st.ctypes.data, aux.ctypes.data))
        return st, aux

    def solver_state(self):
        zl = np.zeros((self.B, self.cap)); zb = np.zeros((self.B, self.cap)); sz = np.zeros((self.B, 3), dtype=np.int32)
        _lib.check(_lib.load().mh_big_batch_save_solver_state(self.handle, zl.ctypes.data, zb.ctypes.data, sz.ctypes.data))
        return dict(zlast=zl, zbuf=zb, sizes=sz)

    def load_solver_state(self, ss):
        zl = np.ascontiguousarray(ss["zlast"], dtype=np.float64); zb = np.ascontiguousarray(ss["zbuf"], dtype=np.float64)
        sz = np.ascontiguousarray(ss["sizes"], dtype=np.int32)
        assert zl.shape == (self.B, self.cap) and zb.shape == (self.B, self.cap) and sz.shape == (self.B, 3)
        _lib.check(_lib.load().mh_big_batch_load_solver_state(self.handle, zl.ctypes.data, zb.ctypes.data, sz.ctypes.data))

    def lu_work(self, reset=False):
        """(B, 4): model flops and model bytes of the block solver's factorisations priced as dense dgesv calls, the flops its
        routines really issue, and the seconds its workgroups spent on the world's problems (mh_big_batch_lu_work)."""
        w = np.zeros((self.B, 4))
        _lib.check(_lib.load().mh_big_batch_lu_work(self.handle, w.ctypes.data, int(bool(reset))))
        return w

    def close(self):
        if self.handle:
            _lib.load().mh_big_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
