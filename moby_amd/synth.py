"""Synthetic inputs for tests and bench.py (numpy only, deterministic).

Per-world perturbations come from splitmix64 keyed by
``seed0 = 0x4D4F4259 ^ world`` so host and device agree.

``sphere_stack_impact_lcp`` assembles, in plain numpy, the impact LCP
``_MM/_qq`` of the reference's sphere-stack scene
(/root/reference/example/stacks/sphere-stack.xml: three unit spheres, m = 1,
J = 0.4 I, at z = 1, 3, 5 on the plane z = 0, mu = 0, NK = 16) following
``ImpactConstraintHandler::compute_problem_data`` (ImpactConstraintHandler.cpp:
1898-2166: rows [d, r x d], X = blockdiag(M_i^-1)) and ``setup_QP`` /
``solve_qp_work`` (ImpactConstraintHandlerQP.cpp:94-263, 271-497).  n = 42.
It is an input generator (and an independent cross-check of the C++/HIP
assembly), not a product path.
"""
import numpy as np

SEED0 = 0x4D4F4259
_MASK = (1 << 64) - 1


def splitmix64(state):
    """One step: returns (new_state, output), both python ints."""
    state = (state + 0x9E3779B97F4A7C15) & _MASK
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return state, z ^ (z >> 31)


def world_uniforms(world, count):
    """`count` doubles in [0,1) for one world."""
    s = (SEED0 ^ world) & _MASK
    out = np.empty(count)
    for i in range(count):
        s, z = splitmix64(s)
        out[i] = (z >> 11) * (1.0 / 9007199254740992.0)
    return out


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])


def orthonormal_basis(n):
    """Two tangents completing `n` (pinned choice; Ravelin's
    Vector3d::determine_orthonormal_basis is not in the tree)."""
    i = int(np.argmin(np.abs(n)))
    e = np.zeros(3); e[i] = 1.0
    s = _cross(n, e); s /= np.linalg.norm(s)
    t = _cross(n, s)
    return s, t


def impact_lcp_from_contacts(bodies, contacts, nk=16, mu=0.0, mu_visc=0.0):
    """bodies: list of dict(x(3), v(6)=[lin;ang], m, J(3)) ; contacts: list of
    dict(p(3), n(3), a, b) with a/b body indices or -1 for the fixed ground.
    Returns (MM (n,n), qq (n,)) of solve_qp_work (ICH-QP:129-216)."""
    nb, nc = len(bodies), len(contacts)
    ngc = 6 * nb
    Xd = np.concatenate([np.concatenate([np.full(3, 1.0 / b["m"]), 1.0 / np.asarray(b["J"], float)]) for b in bodies])
    v = np.concatenate([b["v"] for b in bodies])
    C = {k: np.zeros((nc, ngc)) for k in "nst"}
    for i, c in enumerate(contacts):
        s, t = orthonormal_basis(c["n"])
        for k, d in (("n", c["n"]), ("s", s), ("t", t)):
            for body, sign in ((c["a"], 1.0), (c["b"], -1.0)):
                if body < 0:
                    continue
                r = c["p"] - bodies[body]["x"]
                dd = sign * d
                C[k][i, 6 * body:6 * body + 3] = dd
                C[k][i, 6 * body + 3:6 * body + 6] = _cross(r, dd)
    blk = lambda a, b: (C[a] * Xd) @ C[b].T
    nvars = 5 * nc
    kh = nk // 2
    nineq = nc + nc * kh
    n = nvars + nineq
    H = np.zeros((nvars, nvars))
    rows = [C["n"], C["s"], C["t"], -C["s"], -C["t"]]
    for a in range(5):
        for b in range(5):
            H[a * nc:(a + 1) * nc, b * nc:(b + 1) * nc] = (rows[a] * Xd) @ rows[b].T
    c = np.concatenate([r_ @ v for r_ in rows])
    Mi = np.zeros((nineq, nvars)); qi = np.zeros(nineq)
    Mi[:nc] = H[:nc]; qi[:nc] = c[:nc]
    row = nc
    for i in range(nc):
        vel = np.sqrt(c[nc + i] ** 2 + c[2 * nc + i] ** 2)
        for j in range(kh):
            th = j / (kh - 1) * (np.pi / 2)
            Mi[row, i] = mu
            Mi[row, nc + i] = -np.cos(th); Mi[row, 3 * nc + i] = -np.cos(th)
            Mi[row, 2 * nc + i] = -np.sin(th); Mi[row, 4 * nc + i] = -np.sin(th)
            qi[row] = mu_visc * vel
            row += 1
    MM = np.zeros((n, n)); qq = np.zeros(n)
    MM[:nvars, :nvars] = H; MM[nvars:, :nvars] = Mi; MM[:nvars, nvars:] = -Mi.T
    qq[:nvars] = c; qq[nvars:] = qi
    return MM, qq


def sphere_stack_world(world, dt=1e-3, g=9.81, hard=True):
    """State of world `world` just before its first impact solve: world 0 is the
    reference scene; others get x,y offsets U(-1e-3,1e-3) and initial v_z
    U(-0.1,0) per sphere.  hard=True offsets every sphere
    independently (tilted sphere-sphere normals: degenerate LCPs on which
    lcp_fast cycles -- the parity set); hard=False shifts the whole stack (the
    scene the simulator produces -- the bench set)."""
    u = world_uniforms(world, 9) if world > 0 else np.full(9, 0.5)
    bodies = []
    for k in range(3):
        kk = k if hard else 0
        off = (u[3 * kk:3 * kk + 2] - 0.5) * 2e-3 if world > 0 else np.zeros(2)
        vz0 = -0.1 * u[3 * k + 2] if world > 0 else 0.0
        bodies.append(dict(x=np.array([off[0], off[1], 1.0 + 2.0 * k]),
                           v=np.array([0, 0, vz0 - g * dt, 0, 0, 0.0]), m=1.0, J=[0.4, 0.4, 0.4]))
    contacts = [dict(p=np.array([bodies[0]["x"][0], bodies[0]["x"][1], 0.0]), n=np.array([0, 0, 1.0]), a=0, b=-1)]
    for k in (1, 2):
        d = bodies[k]["x"] - bodies[k - 1]["x"]
        nrm = d / np.linalg.norm(d)
        contacts.append(dict(p=bodies[k - 1]["x"] + nrm * 1.0, n=nrm, a=k, b=k - 1))
    return bodies, contacts


def sphere_stack_impact_lcp(B, first_world=0, hard=True):
    """(M (B,42,42) row-major, q (B,42)) for worlds first_world..first_world+B-1."""
    Ms = np.zeros((B, 42, 42)); qs = np.zeros((B, 42))
    for w in range(B):
        bodies, contacts = sphere_stack_world(first_world + w, hard=hard)
        Ms[w], qs[w] = impact_lcp_from_contacts(bodies, contacts, nk=16, mu=0.0)
    return Ms, qs


def random_lcp(B, n, kind="pd", seed=0):
    """Seeded LCP families: 'pd' (A A' + 0.1 I), 'psd' (rank-deficient A A'),
    'copos' ([H -N'; N 0] saddle like the impact LCP)."""
    rng = np.random.default_rng(seed)
    Ms = np.zeros((B, n, n)); qs = rng.standard_normal((B, n))
    for b in range(B):
        if kind == "pd":
            A = rng.standard_normal((n, n)); Ms[b] = A @ A.T + 0.1 * np.eye(n)
        elif kind == "psd":
            r = max(1, n // 2)
            A = rng.standard_normal((n, r)); Ms[b] = A @ A.T
            qs[b] = Ms[b] @ rng.standard_normal(n) + np.abs(rng.standard_normal(n)) * 0.1
        elif kind == "copos":
            h = n // 2
            A = rng.standard_normal((h, h)); H = A @ A.T + 0.1 * np.eye(h)
            N = np.abs(rng.standard_normal((n - h, h)))
            Ms[b, :h, :h] = H; Ms[b, h:, :h] = N; Ms[b, :h, h:] = -N.T
        else:
            raise ValueError(kind)
    return Ms, qs
