"""CPU: the oracle against the ONE dynamic trajectory among the reference's regression files that a scene in scope
produces -- regress/contact-constrained-pendulum.dat (6500 rows, 6.5 s of a pendulum whose joint is six contact
constraints: example/contact-constrained-pendulum).  Unlike sphere-stack / sitting-box (bodies at rest) this pins the free-
body dynamics, the impact LCP on opposing contacts (rank-deficient), the quaternion integration and the drift of a
velocity-level method over thousands of steps."""
import os

import numpy as np

from moby_amd import scene as S
from moby_amd import stack as K

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pin_error(q7):
    x, y, z, w = q7[3:7]
    Ry = np.array([2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w)])      # R (0, 1, 0)
    return np.linalg.norm(q7[:3] + Ry)


def test_pendulum_follows_the_reference_recording(oracle):
    """The recording was made by the older revision that integrates positions with the UPDATED velocity (row 1 already has
    y = -4.905e-6 = v+ dt with v+ = -(g/2) dt, the post-impact velocity of a rod released horizontally about its end;
    TimeSteppingSimulator.cpp:156-164 integrates with the OLD velocity).  From rest the two schemes produce the same sequence
    one row apart (x_k^old = x_{k+1}^new, v_k^old = v_k^new), so oracle row k + 1 is compared with recording row k.
    Agreement: the file's 6 digits for the first 100 steps, 5e-5 over 0.5 s, 1e-3 over 1 s, 3e-2 over the 6.5 s (the phase
    drifts by 0.3 %); the recording's own constraint drift (the pin has moved 1.5e-2 by the end -- stabilisation does not
    repair it there either) is reproduced to 8 %."""
    rows = np.load(os.path.join(GOLD, "contact_constrained_pendulum_dat.npz"))["rows"]
    assert rows.shape == (6500, 8) and abs(rows[1, 0] - 1e-3) < 1e-12
    sc = K.pendulum_scene()
    st = K.pendulum_state(1)[0].copy(); aux = S.new_aux(1)
    np.testing.assert_allclose(st[:7], rows[0, 1:], atol=5e-7)
    traj = np.zeros((len(rows) + 1, 7)); traj[0] = st[:7]
    for k in range(len(rows)):
        oracle.big_step(sc, st, aux, 1e-3, 1, cap=64)
        traj[k + 1] = st[:7]
    d = np.abs(traj[1:] - rows[:, 1:]).max(axis=1)
    assert d[:100].max() < 6e-7, d[:100].max()                       # the file's resolution
    assert d[:500].max() < 5e-5 and d[:1000].max() < 1e-3 and d.max() < 3e-2, (d[:500].max(), d[:1000].max(), d.max())
    assert np.abs(traj[:-1] - rows[:, 1:])[1:6].max() > 4e-6         # without the one-row shift the first rows are a full step apart
    e_o = np.array([pin_error(q) for q in traj[1:]]); e_g = np.array([pin_error(q) for q in rows[:, 1:]])
    assert 1.2e-2 < e_g[-1] < 1.7e-2 and abs(e_o[-1] / e_g[-1] - 1.0) < 0.1
    late = slice(500, None)
    assert np.abs(e_o[late] / e_g[late] - 1.0).max() < 0.12
    # one 48-row impact LCP per step (6 contacts x (6 + NK/2)); the stabiliser is entered every step (the plugin's distance
    # -|p| is never positive) and gives up in update_q, as the recording's drift says the reference's does
    assert aux["mini_steps"][0] == 6500 and aux["lcp_rows"][0] >= 48 * 6500
    assert aux["status"][0] & ~(S.MH_WORLD_STAB_FAILED | S.MH_WORLD_IMPACT_TOL) == 0


def test_pendulum_first_impact_is_the_analytic_one(oracle):
    """A rod released horizontally about its end: the pin's impulse leaves the COM with acceleration g m d^2 / (J + m d^2);
    J = 0.4 r^2 = 0.99995, d = 1 => v_y after the first step = -9.81e-3 / 1.99995 and the spin about z to match."""
    sc = K.pendulum_scene(cstab_max_iterations=0)
    st = K.pendulum_state(1)[0].copy(); aux = S.new_aux(1)
    oracle.big_step(sc, st, aux, 1e-3, 1, cap=64)
    J = 1.5811 ** 2 * 0.4
    np.testing.assert_allclose(st[8], -9.81e-3 / (1.0 + J), rtol=1e-9)
    np.testing.assert_allclose(st[12], st[8] / 1.0, rtol=1e-9)         # v = omega x r, r = (1, 0, 0)
    assert abs(st[7]) < 1e-15 and abs(st[9]) < 1e-15 and np.abs(st[10:12]).max() < 1e-15
