"""Host-side scene description for the many-worlds stepper.

ctypes mirrors of ``mh_scene`` / ``mh_world_aux`` (include/moby_hip.h) and
builders for the reference scenes the BASELINE configs name:

* ``sphere_stack_scene``   /root/reference/example/stacks/sphere-stack.xml
* ``bouncing_ball_scene``  /root/reference/example/bouncing-ball/bouncing-ball.xml

Body order = the order programs/regress.cpp:82-93 prints them (sorted by id);
the static ground is the last id.
"""
import ctypes
import math

import numpy as np

MH_MAX_BODIES = 8
MH_MAX_PAIRS = 36
MH_BODY_STATE = 13
MH_RAND_WORDS = 32
MH_LCP_MAX_N_WAVE = 64
MH_NOSLIP_MAX = 16
MH_CSTAB_DEFAULT_MAX_ITERATIONS = 10   # include/moby_hip.h
MH_MAX_SPOKES = 8
MH_GEOM_SPHERE, MH_GEOM_SPOKES, MH_GEOM_BOX, MH_GEOM_PIN = 0, 1, 2, 3
NEAR_ZERO = math.sqrt(np.finfo(np.float64).eps)   # include/Moby/Constants.h:21

MH_WORLD_OK, MH_WORLD_LCP_FAILED, MH_WORLD_IMPACT_TOL, MH_WORLD_UNSUPPORTED, MH_WORLD_STAB_FAILED, MH_WORLD_STALLED = 0, 1, 2, 4, 8, 16


class mh_scene(ctypes.Structure):
    _fields_ = [
        ("nb", ctypes.c_int), ("has_ground", ctypes.c_int),
        ("geom_type", ctypes.c_int * MH_MAX_BODIES),
        ("geom_dim", (ctypes.c_double * 3) * MH_MAX_BODIES),
        ("mass", ctypes.c_double * MH_MAX_BODIES),
        ("inertia", (ctypes.c_double * 3) * MH_MAX_BODIES),
        ("plane_R", ctypes.c_double * 9), ("plane_o", ctypes.c_double * 3),
        ("gravity", ctypes.c_double * 3),
        ("pair_enabled", ctypes.c_int * MH_MAX_PAIRS),
        ("cp_epsilon", ctypes.c_double * MH_MAX_PAIRS),
        ("cp_mu_coulomb", ctypes.c_double * MH_MAX_PAIRS),
        ("cp_mu_viscous", ctypes.c_double * MH_MAX_PAIRS),
        ("cp_compliance", ctypes.c_double * MH_MAX_PAIRS),
        ("cp_nk", ctypes.c_int * MH_MAX_PAIRS),
        ("min_step_size", ctypes.c_double), ("contact_dist_thresh", ctypes.c_double),
        ("cstab_eps", ctypes.c_double), ("cstab_max_iterations", ctypes.c_uint), ("lcp_n_max", ctypes.c_int),
    ]


class mh_world_aux(ctypes.Structure):
    _fields_ = [
        ("rng", ctypes.c_uint32 * MH_RAND_WORDS), ("time", ctypes.c_double),
        ("zlast", ctypes.c_double * MH_LCP_MAX_N_WAVE), ("zbuf", ctypes.c_double * MH_LCP_MAX_N_WAVE),
        ("zlast_size", ctypes.c_int), ("zbuf_size", ctypes.c_int), ("zbuf_cap", ctypes.c_int), ("status", ctypes.c_int),
        ("vns", ctypes.c_double * MH_NOSLIP_MAX), ("vns_size", ctypes.c_int), ("pad0", ctypes.c_int),
        ("steps", ctypes.c_ulonglong), ("mini_steps", ctypes.c_ulonglong), ("lcp_solves", ctypes.c_ulonglong),
        ("lcp_rows", ctypes.c_ulonglong), ("lcp_pivots", ctypes.c_ulonglong), ("stab_iters", ctypes.c_ulonglong),
        ("lcp_alg_bytes", ctypes.c_ulonglong), ("stab_rows", ctypes.c_ulonglong),
    ]


AUX_DTYPE = np.dtype([
    ("rng", np.uint32, MH_RAND_WORDS), ("time", np.float64),
    ("zlast", np.float64, MH_LCP_MAX_N_WAVE), ("zbuf", np.float64, MH_LCP_MAX_N_WAVE),
    ("zlast_size", np.int32), ("zbuf_size", np.int32), ("zbuf_cap", np.int32), ("status", np.int32),
    ("vns", np.float64, MH_NOSLIP_MAX), ("vns_size", np.int32), ("pad0", np.int32),
    ("steps", np.uint64), ("mini_steps", np.uint64), ("lcp_solves", np.uint64),
    ("lcp_rows", np.uint64), ("lcp_pivots", np.uint64), ("stab_iters", np.uint64),
    ("lcp_alg_bytes", np.uint64), ("stab_rows", np.uint64)], align=True)
assert AUX_DTYPE.itemsize == ctypes.sizeof(mh_world_aux), (AUX_DTYPE.itemsize, ctypes.sizeof(mh_world_aux))


def rpy_to_quat(roll, pitch, yaw):
    """Quatd::rpy as Ravelin forms it (half-angle products, ZYX), x y z w.  The XML readers turn every ``rpy`` attribute
    into a quaternion first (XMLTree.cpp / RigidBody.cpp:201-210, Primitive.cpp:273-279)."""
    cr, sr = math.cos(roll * 0.5), math.sin(roll * 0.5)
    cp, sp = math.cos(pitch * 0.5), math.sin(pitch * 0.5)
    cy, sy = math.cos(yaw * 0.5), math.sin(yaw * 0.5)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])


def quat_to_R(q):
    """Rotation matrix of a unit quaternion (x y z w) in the form Ravelin uses: diagonal 2 (w^2 + q_i^2) - 1.
    PINNED by regress/sphere-stack.dat: the plane of sphere-stack.xml is posed with rpy = (1.5707963267949, 0, 0), 3.4e-15 rad
    past a right angle, and the recording's sphere 1 picks up dv_y = -1.07824e-16 per step = -(3 m g dt) n_y; of the
    algebraically equal forms only 2 w^2 - 1 = -3.66374e-15 gives that n_y (cos(r) = -3.49148e-15, 1 - 2 x^2 = -3.55271e-15,
    w^2 - x^2 = -3.60822e-15) -- tests/test_oracle_world.py::test_sphere_stack_roundoff_fingerprint."""
    x, y, z, w = q
    return np.array([[2 * (w * w + x * x) - 1, 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 2 * (w * w + y * y) - 1, 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 2 * (w * w + z * z) - 1]])


def rpy_to_R(roll, pitch, yaw):
    """Rotation matrix of an XML ``rpy`` attribute (a static pose: the plane), through the quaternion as the reference does."""
    return quat_to_R(rpy_to_quat(roll, pitch, yaw))


def pair_index(i, j, ntot):
    return i * ntot - (i * (i + 1)) // 2 + (j - i - 1)


def _defaults(sc):
    sc.min_step_size = NEAR_ZERO          # TimeSteppingSimulator.cpp:48
    sc.contact_dist_thresh = 1e-6         # ConstraintSimulator.cpp:56
    sc.cstab_eps = NEAR_ZERO              # ConstraintStabilization.cpp:59
    sc.cstab_max_iterations = MH_CSTAB_DEFAULT_MAX_ITERATIONS  # ConstraintStabilization.cpp:56 is UINT_MAX: see moby_hip.h
    for p in range(MH_MAX_PAIRS):
        sc.pair_enabled[p] = 1
        sc.cp_nk[p] = 4                   # ContactParameters.cpp:26


def make_scene(radii, masses, gravity, ground_rpy=None, ground_o=(0.0, 0.0, 0.0), params=None, inertias=None):
    """Spheres 0..nb-1 (+ optional plane with id nb).  params: {(i, j): dict(epsilon, mu_coulomb,
    mu_viscous, compliance, nk)} with i < j."""
    sc = mh_scene()
    _defaults(sc)
    nb = len(radii)
    sc.nb = nb
    sc.has_ground = 1 if ground_rpy is not None else 0
    for b in range(nb):
        sc.geom_type[b] = 0
        sc.geom_dim[b][0] = radii[b]
        sc.mass[b] = masses[b]
        diag = (radii[b] * radii[b] * masses[b] * 2.0 / 5.0) if inertias is None else inertias[b]   # SpherePrimitive.cpp:149
        for k in range(3):
            sc.inertia[b][k] = diag
    R = rpy_to_R(*ground_rpy) if ground_rpy is not None else np.eye(3)
    for k in range(9):
        sc.plane_R[k] = R.flat[k]
    for k in range(3):
        sc.plane_o[k] = ground_o[k]
        sc.gravity[k] = gravity[k]
    ntot = nb + sc.has_ground
    for (i, j), d in (params or {}).items():
        p = pair_index(i, j, ntot)
        sc.cp_epsilon[p] = d.get("epsilon", 0.0)
        sc.cp_mu_coulomb[p] = d.get("mu_coulomb", 0.0)
        sc.cp_mu_viscous[p] = d.get("mu_viscous", 0.0)
        sc.cp_compliance[p] = d.get("compliance", 0.0)
        sc.cp_nk[p] = max(4, d.get("nk", 4))
    return sc


def sphere_stack_scene(cstab_max_iterations=MH_CSTAB_DEFAULT_MAX_ITERATIONS):
    """example/stacks/sphere-stack.xml:11-51.

    ``constraint-stabilization-max-iterations`` is not set in the XML (default
    UINT_MAX, ConstraintStabilization.cpp:56), but with the arithmetic restated
    in oracle/world.hpp the stabilisation loop enters an exact 2-cycle at step 3
    of this scene (tests/test_oracle_world.py::test_stabilisation_cycle), so the
    batch configs run it with a finite cap ("stabilisation cap")."""
    cp = dict(epsilon=0.0, mu_coulomb=0.0, mu_viscous=0.0, nk=16)
    sc = _sphere_stack_scene(cp)
    sc.cstab_max_iterations = cstab_max_iterations
    sc.lcp_n_max = 42          # 3 contacts x (6 + NK/2) rows
    return sc


def _sphere_stack_scene(cp):
    return make_scene([1.0, 1.0, 1.0], [1.0, 1.0, 1.0], (0.0, 0.0, -9.81),
                      ground_rpy=(1.5707963267949, 0.0, 0.0),
                      params={(0, 3): cp, (0, 1): cp, (1, 2): cp})


def sphere_stack_state(B=1, perturb=True):
    """Initial states (B, 3*13): world 0 is the reference scene; world w > 0 shifts
    the stack by U(-1e-3,1e-3) in x,y and gives each sphere v_z U(-0.1,0)."""
    from .synth import world_uniforms
    st = np.zeros((B, 3, MH_BODY_STATE))
    for w in range(B):
        u = world_uniforms(w, 5) if (perturb and w > 0) else None
        for k in range(3):
            st[w, k, 0:3] = (0.0, 0.0, 1.0 + 2.0 * k)
            st[w, k, 6] = 1.0
            if u is not None:
                st[w, k, 0] += (u[0] - 0.5) * 2e-3
                st[w, k, 1] += (u[1] - 0.5) * 2e-3
                st[w, k, 9] = -0.1 * u[2 + k]
    return st.reshape(B, 3 * MH_BODY_STATE)


def sphere_stack_state_range(first_world, B):
    """Worlds first_world .. first_world+B-1 of the infinite perturbed family."""
    from .synth import world_uniforms
    st = np.zeros((B, 3, MH_BODY_STATE))
    for i in range(B):
        w = first_world + i
        u = world_uniforms(w, 5) if w > 0 else None
        for k in range(3):
            st[i, k, 0:3] = (0.0, 0.0, 1.0 + 2.0 * k)
            st[i, k, 6] = 1.0
            if u is not None:
                st[i, k, 0] += (u[0] - 0.5) * 2e-3
                st[i, k, 1] += (u[1] - 0.5) * 2e-3
                st[i, k, 9] = -0.1 * u[2 + k]
    return st.reshape(B, 3 * MH_BODY_STATE)


def bouncing_ball_scene():
    """example/bouncing-ball/bouncing-ball.xml:11-37 (density 1 => m = 4 pi / 3)."""
    r = 1.0
    m = 1.0 * (math.pi * r * r * r * 4.0 / 3.0)     # SpherePrimitive.cpp:144-146
    sc = make_scene([r], [m], (0.0, -9.81, 0.0), ground_rpy=(0.0, 0.0, 0.0),
                    params={(0, 1): dict(epsilon=1.0, mu_coulomb=0.0, mu_viscous=0.0, nk=4)})
    sc.lcp_n_max = 8
    return sc


def bouncing_ball_state(B=1):
    st = np.zeros((B, 1, MH_BODY_STATE))
    st[:, 0, 0:3] = (0.0, 1.5, 0.0)
    st[:, 0, 6] = 1.0
    st[:, 0, 10:13] = (0.0, 10.0, 0.0)
    return st.reshape(B, MH_BODY_STATE)


def rimless_wheel_scene(cstab_max_iterations=MH_CSTAB_DEFAULT_MAX_ITERATIONS):
    """example/rimless-wheel/wheel.xml + coldet-plugin.cpp + params.h: one free body (m = 1,
    J = diag(2,1,2)) whose collision geometry is N_SPOKES = 6 spoke tips at R = 1, a plane with
    rpy = (1.570796326949, 0, 0), gravity (0.099833, 0, -0.995), epsilon 0, mu-coulomb 100
    (=> the no-slip model, ICH:127-135)."""
    sc = mh_scene()
    _defaults(sc)
    sc.nb = 1
    sc.has_ground = 1
    sc.geom_type[0] = MH_GEOM_SPOKES
    sc.geom_dim[0][0] = 1.0       # R  (params.h:4)
    sc.geom_dim[0][1] = 6.0       # N_SPOKES (params.h:6)
    sc.mass[0] = 1.0
    for k, j in enumerate((2.0, 1.0, 2.0)):
        sc.inertia[0][k] = j
    R = rpy_to_R(1.570796326949, 0.0, 0.0)
    for k in range(9):
        sc.plane_R[k] = R.flat[k]
    for k, g in enumerate((0.099833, 0.0, -0.995)):
        sc.gravity[k] = g
    p = pair_index(0, 1, 2)
    sc.cp_epsilon[p] = 0.0
    sc.cp_mu_coulomb[p] = 100.0
    sc.cp_nk[p] = 4
    sc.cstab_max_iterations = cstab_max_iterations
    sc.lcp_n_max = 8
    return sc


def rimless_wheel_regress_scene(cstab_max_iterations=MH_CSTAB_DEFAULT_MAX_ITERATIONS):
    """The scene regress/rimless-wheel.dat was recorded with.  That file predates the wheel.xml in
    the tree: its trajectory (deceleration -0.152 rad/s^2 while pivoting, first post-impact rate
    0.2893 rad/s) is reproduced by the commented-out "alpha = 0.05" gravity line of wheel.xml:13-14
    and an inertia of 2 about the wheel axis, not by the active lines (gravity alpha = 0.1,
    J = diag(2,1,2)) -- see tests/test_oracle_wheel.py."""
    sc = rimless_wheel_scene(cstab_max_iterations)
    for k, g in enumerate((0.049979, 0.0, -0.99875)):
        sc.gravity[k] = g
    sc.inertia[0][1] = 2.0
    return sc


def rimless_wheel_state(theta_dots=(0.24,)):
    """example/rimless-wheel/init.cpp:166-191: theta = 0, z = 0.866025403784439,
    the SVelocityd built there has a null (= GLOBAL) pose, so its linear part 2 pi R * (theta_dot /
    2 pi) is the velocity of the body point at the global ORIGIN; the COM moves with
    v_x = theta_dot R + theta_dot * z.  omega_y = theta_dot (RIMLESS_WHEEL_THETAD, 0.24 in
    regress/regression-test:58-61)."""
    B = len(theta_dots)
    st = np.zeros((B, 1, MH_BODY_STATE))
    for w, thd in enumerate(theta_dots):
        dist_per_rev = 2 * math.pi * 1.0
        rev_per_sec = thd / (math.pi * 2.0)
        st[w, 0, 2] = 0.866025403784439
        st[w, 0, 6] = 1.0
        st[w, 0, 7] = dist_per_rev * rev_per_sec + thd * 0.866025403784439
        st[w, 0, 11] = thd
    return st.reshape(B, MH_BODY_STATE)


def box_scene(dims=(1.0, 1.0, 1.0), density=1.0, gravity=(0.0, -9.81, 0.0), ground_rpy=(0.0, 0.0, 0.0), epsilon=0.0,
              mu_coulomb=0.0, mu_viscous=0.0, nk=8, cstab_max_iterations=MH_CSTAB_DEFAULT_MAX_ITERATIONS):
    """example/simple-contact/simplest.xml (a unit box of density 1 on the default plane, NK = 8) and,
    with mu_coulomb = 0.1, spinning-box-frictional.xml.  Mass properties as BoxPrimitive::
    calc_mass_properties (BoxPrimitive.cpp:692-712)."""
    sc = mh_scene()
    _defaults(sc)
    sc.nb = 1
    sc.has_ground = 1
    sc.geom_type[0] = MH_GEOM_BOX
    x, y, z = dims
    for k in range(3):
        sc.geom_dim[0][k] = dims[k]
    m = density * (x * y * z)
    sc.mass[0] = m
    M = m / 12.0
    for k, j in enumerate((M * (y * y + z * z), M * (x * x + z * z), M * (x * x + y * y))):
        sc.inertia[0][k] = j
    R = rpy_to_R(*ground_rpy)
    for k in range(9):
        sc.plane_R[k] = R.flat[k]
    for k in range(3):
        sc.gravity[k] = gravity[k]
    p = pair_index(0, 1, 2)
    sc.cp_epsilon[p] = epsilon
    sc.cp_mu_coulomb[p] = mu_coulomb
    sc.cp_mu_viscous[p] = mu_viscous
    sc.cp_nk[p] = nk
    sc.cstab_max_iterations = cstab_max_iterations
    sc.lcp_n_max = 64
    return sc


def box_state(pos=(0.0, 0.5, 0.0), quat=(0.0, 0.0, 0.0, 1.0), v=(0.0, 0.0, 0.0), w=(0.0, 0.0, 0.0)):
    st = np.zeros((1, MH_BODY_STATE))
    st[0, 0:3] = pos
    q = np.array(quat, dtype=np.float64)
    st[0, 3:7] = q / np.linalg.norm(q)
    st[0, 7:10] = v
    st[0, 10:13] = w
    return st


def glibc_srand_state(seed=1):
    """The 32-word rand() state (31-word TYPE_3 ring + index) right after srand(seed), in the layout
    mh_rand_seed / oracle/glibc_rand.h use.  Pure Python so that host-side helpers never need the
    HIP library (loading it before torch would bring a second HIP runtime into the process)."""
    seed = int(seed) & 0xFFFFFFFF
    if seed == 0:
        seed = 1
    r = [0] * 31
    r[0] = seed if seed < 2 ** 31 else seed - 2 ** 32
    for i in range(1, 31):
        hi, lo = int(r[i - 1] / 127773), int(math.fmod(r[i - 1], 127773))     # C division / remainder (truncating)
        w = 16807 * lo - 2836 * hi
        if w < 0:
            w += 2147483647
        r[i] = w
    st = [x & 0xFFFFFFFF for x in r]
    idx = 3
    for _ in range(34, 344):
        st[idx] = (st[idx] + st[(idx + 28) % 31]) & 0xFFFFFFFF
        idx = (idx + 1) % 31
    return np.array(st + [idx], dtype=np.uint32)


def new_aux(B, seed=1):
    """B fresh mh_world_aux records (numpy structured array), rand() at srand(seed)."""
    aux = np.zeros(B, dtype=AUX_DTYPE)
    aux["rng"][:] = glibc_srand_state(seed)
    return aux
