"""Host-side mirror of the articulated-body stepper (include/moby_hip_artic.h).

``ArticBatch(model, q, qd)`` plays the role of B simulators that each hold one fixed-base ``RCArticulatedBody``
(``step``), and of ``RCArticulatedBodyd::calc_fwd_dyn`` / ``get_generalized_inertia`` on their states (``fwd_dyn``);
``load_sdf`` reads example/ur10/model.sdf through the C++ loader of libmoby_hip_io.so; ``model_from_links`` builds a
model from global link poses the way that loader does (synthetic chains for the tests).
"""
import ctypes
import os

import numpy as np

from . import _lib
from . import io as mio
from . import scene as S

MH_ARTIC_MAX_JOINTS = 16
MH_JOINT_REVOLUTE, MH_JOINT_PRISMATIC = 0, 1
MH_ARTIC_MAX_SPHERES = 4
_NJ = MH_ARTIC_MAX_JOINTS
_NS = MH_ARTIC_MAX_SPHERES


class mh_artic_model(ctypes.Structure):
    _fields_ = [("nj", ctypes.c_int), ("parent", ctypes.c_int * _NJ), ("jtype", ctypes.c_int * _NJ),
                ("Rrel", (ctypes.c_double * 9) * _NJ), ("trel", (ctypes.c_double * 3) * _NJ), ("axis", (ctypes.c_double * 3) * _NJ),
                ("com", (ctypes.c_double * 3) * _NJ), ("inertia", (ctypes.c_double * 9) * _NJ), ("mass", ctypes.c_double * _NJ),
                ("lolimit", ctypes.c_double * _NJ), ("hilimit", ctypes.c_double * _NJ), ("limit_restitution", ctypes.c_double * _NJ),
                ("gravity", ctypes.c_double * 3), ("algorithm", ctypes.c_int), ("floating_base", ctypes.c_int),
                ("nspheres", ctypes.c_int), ("sphere_link", ctypes.c_int * _NS), ("sphere_center", (ctypes.c_double * 3) * _NS),
                ("sphere_radius", ctypes.c_double * _NS), ("plane_R", ctypes.c_double * 9), ("plane_o", ctypes.c_double * 3),
                ("cp_epsilon", ctypes.c_double), ("cp_mu_coulomb", ctypes.c_double), ("min_step_size", ctypes.c_double),
                ("contact_dist_thresh", ctypes.c_double), ("cp_mu_viscous", ctypes.c_double), ("cp_compliance", ctypes.c_double),
                ("cp_nk", ctypes.c_int), ("cstab_max_iterations", ctypes.c_int), ("cstab_eps", ctypes.c_double)]


MH_ARTIC_CRB, MH_ARTIC_FSAB = 0, 1      # moby_hip_artic.h: RCArticulatedBody::algorithm_type


class mh_io_artic(ctypes.Structure):
    _fields_ = [("model", mh_artic_model), ("link_id", (ctypes.c_char * mio.MH_IO_ID_LEN) * _NJ),
                ("joint_id", (ctypes.c_char * mio.MH_IO_ID_LEN) * _NJ)]


def load_sdf(path, gravity=(0.0, 0.0, -9.81)):
    """-> (mh_artic_model, link names, joint names): SDFReader::read_model's articulated-body branch (SDFReader.cpp:928-996)."""
    lib = mio.load()
    lib.mh_io_load_sdf.restype = ctypes.c_int
    lib.mh_io_load_sdf.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(mh_io_artic)]
    io = mh_io_artic()
    g = (ctypes.c_double * 3)(*gravity)
    if lib.mh_io_load_sdf(os.fsencode(path), g, ctypes.byref(io)) != 0:
        raise mio.SceneError(lib.mh_io_last_error().decode("utf-8", "replace"))
    m = mh_artic_model()
    ctypes.memmove(ctypes.addressof(m), ctypes.addressof(io.model), ctypes.sizeof(mh_artic_model))
    return m, [io.link_id[i].value.decode() for i in range(m.nj)], [io.joint_id[i].value.decode() for i in range(m.nj)]


def load_urdf(path, gravity=(0.0, 0.0, -9.81)):
    """A URDF robot with a fixed base (src/URDFReader.cpp; include/moby_hip_io.h: mh_io_load_urdf) -> (model, link ids, joint ids)."""
    lib = mio.load()
    lib.mh_io_load_urdf.restype = ctypes.c_int
    lib.mh_io_load_urdf.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(mh_io_artic)]
    out = mh_io_artic()
    g = (ctypes.c_double * 3)(*gravity)
    if lib.mh_io_load_urdf(os.fsencode(path), g, ctypes.byref(out)) != 0:
        raise mio.SceneError(lib.mh_io_last_error().decode("utf-8", "replace"))
    m = mh_artic_model()
    ctypes.memmove(ctypes.addressof(m), ctypes.addressof(out.model), ctypes.sizeof(mh_artic_model))
    return m, [out.link_id[i].value.decode() for i in range(m.nj)], [out.joint_id[i].value.decode() for i in range(m.nj)]


def load_xml(path):
    """-> (mh_artic_model, link names, joint names, q0, qd0, step size): a Moby XML file with one RCArticulatedBody (fixed base, or floating-base="true": six virtual joints first, include/moby_hip_io.h)
    (include/moby_hip_io.h: mh_io_load_xml_artic) -- the model at q = 0, the joints' q / qd attributes as the initial state."""
    lib = mio.load()
    lib.mh_io_load_xml_artic.restype = ctypes.c_int
    lib.mh_io_load_xml_artic.argtypes = [ctypes.c_char_p, ctypes.POINTER(mh_io_artic), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_double)]
    io = mh_io_artic(); q0 = (ctypes.c_double * _NJ)(); qd0 = (ctypes.c_double * _NJ)(); dt = ctypes.c_double(0.0)
    if lib.mh_io_load_xml_artic(os.fsencode(path), ctypes.byref(io), q0, qd0, ctypes.byref(dt)) != 0:
        raise mio.SceneError(lib.mh_io_last_error().decode("utf-8", "replace"))
    m = mh_artic_model()
    ctypes.memmove(ctypes.addressof(m), ctypes.addressof(io.model), ctypes.sizeof(mh_artic_model))
    n = m.nj
    return (m, [io.link_id[i].value.decode() for i in range(n)], [io.joint_id[i].value.decode() for i in range(n)],
            np.array(q0[:n]), np.array(qd0[:n]), dt.value)


def model_from_links(links, gravity=(0.0, 0.0, -9.81), floating_base=None):
    """links: dicts (parents first) with parent (-1 = base), type, R0 (3x3, model frame at q = 0), x0, axis (model frame),
    com (link frame), inertia (3x3 about the COM, link axes), mass, lo, hi, restitution.
    floating_base = dict(R0, x0, mass, inertia): the base link (pose of its COM frame in the global frame, inertia in its own axes) is carried by the six
    virtual joints of mh_artic_model.floating_base (include/moby_hip_artic.h) -- joints 0..5 of the returned model, the base link is link 5, `links` follow
    as joints 6.. with parent -1 meaning the base link; exactly what mh_io_load_xml_artic builds for floating-base="true"."""
    if floating_base is not None:
        Rb = np.asarray(floating_base["R0"], dtype=float); xb = np.asarray(floating_base["x0"], dtype=float)
        virt = [dict(parent=v - 1, type=MH_JOINT_PRISMATIC if v < 3 else MH_JOINT_REVOLUTE, R0=np.eye(3) if v < 3 else Rb, x0=xb,
                     axis=(np.eye(3) if v < 3 else Rb)[:, v % 3], com=(0.0, 0.0, 0.0), inertia=floating_base["inertia"] if v == 5 else np.zeros((3, 3)),
                     mass=float(floating_base["mass"]) if v == 5 else 0.0) for v in range(6)]
        links = virt + [dict(L, parent=5 if L["parent"] < 0 else L["parent"] + 6) for L in links]
    m = mh_artic_model()
    m.nj = len(links)
    for i, L in enumerate(links):
        p = L["parent"]
        Rp = np.eye(3) if p < 0 else np.asarray(links[p]["R0"], dtype=float)
        xp = np.zeros(3) if p < 0 else np.asarray(links[p]["x0"], dtype=float)
        Rc = np.asarray(L["R0"], dtype=float); xc = np.asarray(L["x0"], dtype=float)
        Rrel = Rp.T @ Rc; trel = Rp.T @ (xc - xp)
        al = Rc.T @ np.asarray(L["axis"], dtype=float); al = al / np.linalg.norm(al)
        m.parent[i] = p; m.jtype[i] = L.get("type", MH_JOINT_REVOLUTE)
        for k in range(9):
            m.Rrel[i][k] = Rrel.flat[k]; m.inertia[i][k] = np.asarray(L["inertia"], dtype=float).flat[k]
        for k in range(3):
            m.trel[i][k] = trel[k]; m.axis[i][k] = al[k]; m.com[i][k] = L["com"][k]
        m.mass[i] = L["mass"]
        m.lolimit[i] = L.get("lo", -np.finfo(float).max); m.hilimit[i] = L.get("hi", np.finfo(float).max)
        m.limit_restitution[i] = L.get("restitution", 0.0)
    for k in range(3):
        m.gravity[k] = gravity[k]
    if floating_base is not None:          # the layout mh_artic_batch_create checks, free of the round-off of Rb' Rb
        m.floating_base = 1
        for v in range(6):
            for k in range(3):
                m.axis[v][k] = float(k == v % 3)
                if v > 0: m.trel[v][k] = 0.0
            if v != 3:
                for k in range(9): m.Rrel[v][k] = float(k % 4 == 0)
    m.cstab_eps = S.NEAR_ZERO              # ConstraintStabilization::eps (CStab:59); stabilisation itself off until cstab_max_iterations is set
    return m


def add_spheres(model, spheres, plane_normal=(0.0, 0.0, 1.0), plane_point=(0.0, 0.0, 0.0), epsilon=0.0, mu_coulomb=100.0, mu_viscous=0.0,
                compliance=0.0, nk=4):
    """Sphere primitives on links against one static plane: spheres = [(link, centre in the link frame, radius), ...]; the
    plane through plane_point with the given normal (the +Y axis of the plane frame, as PlanePrimitive has it); the
    ContactParameters of the (robot, plane) pair (ur10.xml:19: epsilon 0, mu-coulomb 100).  Returns the model."""
    assert 0 < len(spheres) <= MH_ARTIC_MAX_SPHERES
    model.nspheres = len(spheres)
    for i, (link, c, r) in enumerate(spheres):
        assert 0 <= link < model.nj and r > 0
        model.sphere_link[i] = int(link); model.sphere_radius[i] = float(r)
        for k in range(3):
            model.sphere_center[i][k] = float(c[k])
    n = np.asarray(plane_normal, dtype=float); n = n / np.linalg.norm(n)
    e = np.eye(3)[int(np.argmin(np.abs(n)))]
    xax = np.cross(n, e); xax = xax / np.linalg.norm(xax); zax = np.cross(xax, n)
    Rp = np.column_stack([xax, n, zax])                       # columns: the plane frame's axes; +Y = normal
    for k in range(9):
        model.plane_R[k] = Rp.flat[k]
    for k in range(3):
        model.plane_o[k] = float(plane_point[k])
    model.cp_epsilon = float(epsilon); model.cp_mu_coulomb = float(mu_coulomb)
    model.cp_mu_viscous = float(mu_viscous); model.cp_compliance = float(compliance); model.cp_nk = int(nk)
    model.min_step_size = S.NEAR_ZERO
    model.contact_dist_thresh = 1e-6
    return model


def chain_model(n, length=0.5, mass=1.0, lo=-1.0, hi=1.0, restitution=0.0, gravity=(0.0, 0.0, -9.81), prismatic_last=False):
    """n rods hanging along -z from the origin, hinged about y (a planar n-pendulum); optionally the last joint slides."""
    links = []
    for i in range(n):
        I = mass * length * length / 12.0
        links.append(dict(parent=i - 1, type=MH_JOINT_PRISMATIC if (prismatic_last and i == n - 1) else MH_JOINT_REVOLUTE,
                          R0=np.eye(3), x0=(0.0, 0.0, -length * i), axis=(0.0, 0.0, 1.0) if (prismatic_last and i == n - 1) else (0.0, 1.0, 0.0),
                          com=(0.0, 0.0, -0.5 * length), inertia=np.diag([I, I, 1e-3 * I]), mass=mass, lo=lo, hi=hi, restitution=restitution))
    return model_from_links(links, gravity)


class ArticBatch:
    def __init__(self, model, q, qd, aux=None):
        lib = _lib.load()
        self.model = model
        self.nj = model.nj
        q = np.ascontiguousarray(q, dtype=np.float64); qd = np.ascontiguousarray(qd, dtype=np.float64)
        self.B = q.shape[0]
        assert q.shape == (self.B, self.nj) and qd.shape == q.shape
        self.handle = ctypes.c_void_p()
        _lib.check(lib.mh_artic_batch_create(ctypes.byref(model), self.B, ctypes.byref(self.handle)))
        self.upload(q, qd, aux)

    def upload(self, q=None, qd=None, aux=None):
        P = lambda a: None if a is None else np.ascontiguousarray(a).ctypes.data
        _lib.check(_lib.load().mh_artic_batch_upload(self.handle, P(q), P(qd), P(aux)))

    def step(self, dt, nsteps=1, stream=None):
        _lib.check(_lib.load().mh_artic_batch_step(self.handle, stream, float(dt), int(nsteps)))

    def fwd_dyn(self, tau=None, want_H=True):
        qdd = np.zeros((self.B, self.nj)); H = np.zeros((self.B, self.nj, self.nj)) if want_H else None
        t = None if tau is None else np.ascontiguousarray(tau, dtype=np.float64)
        _lib.check(_lib.load().mh_artic_batch_fwd_dyn(self.handle, None if t is None else t.ctypes.data, qdd.ctypes.data,
                                                      None if H is None else H.ctypes.data))
        return qdd, H

    def link_poses(self):
        P = np.zeros((self.B, self.nj, 12))
        _lib.check(_lib.load().mh_artic_batch_link_poses(self.handle, P.ctypes.data))
        return P

    def jacobian(self, link, points):
        """calc_jacobian for every resident state: (B, 6, nj); points (B, 3) in the model frame."""
        p = np.ascontiguousarray(points, dtype=np.float64); assert p.shape == (self.B, 3)
        J = np.zeros((self.B, 6, self.nj))
        _lib.check(_lib.load().mh_artic_batch_jacobian(self.handle, int(link), p.ctypes.data, J.ctypes.data))
        return J

    def download(self):
        q = np.zeros((self.B, self.nj)); qd = np.zeros((self.B, self.nj)); aux = np.zeros(self.B, dtype=S.AUX_DTYPE)
        _lib.check(_lib.load().mh_artic_batch_download(self.handle, q.ctypes.data, qd.ctypes.data, aux.ctypes.data))
        return q, qd, aux

    def close(self):
        if self.handle:
            _lib.load().mh_artic_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
