"""CPU tests of the oracle's rimless-wheel path (spokes geometry of
example/rimless-wheel/coldet-plugin.cpp + the no-slip impact model,
ImpactConstraintHandler.cpp:1009-1417) against the reference's regression data
and closed-form impact maps."""
import math
import os

import numpy as np

from moby_amd import scene as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H = 0.866025403784439


def test_first_no_slip_impact_is_the_pivot_map(oracle):
    """Both rear spokes touch at t = 0; the no-slip model must leave the wheel pivoting about the
    front tip P with the angular momentum about P conserved: w+ = (J w + (r x m v)_y) / (J + m r^2),
    v+ = w+ x (c - P)."""
    for sc, J in ((S.rimless_wheel_scene(), 1.0), (S.rimless_wheel_regress_scene(), 2.0)):
        st = S.rimless_wheel_state((0.24,))[0].copy()
        aux = S.new_aux(1)
        v0 = st[7]
        assert abs(v0 - 0.24 * (1.0 + H)) < 1e-15
        oracle.world_handle_impacts(sc, st, aux)
        assert aux["status"][0] == 0 and aux["lcp_solves"][0] == 1 and aux["lcp_rows"][0] == 2
        w_plus = (J * 0.24 + H * v0) / (J + 1.0)
        np.testing.assert_allclose(st[11], w_plus, rtol=1e-9)
        np.testing.assert_allclose(st[7:10], [H * w_plus, 0.0, 0.5 * w_plus], atol=1e-9)
        assert abs(st[10]) < 1e-9 and abs(st[12]) < 1e-9
        # the rear contact released (cn = 0), the front one carries the impulse
        assert aux["vns_size"][0] == 2
        assert aux["vns"][0, 0] == 0.0 and aux["vns"][0, 1] > 0.0


def test_rimless_wheel_matches_reference_regression_data(oracle):
    """regress/rimless-wheel.dat, 6275 rows at dt = 1e-3.  The recording starts from the POST-impact
    state (its first step already pivots at 0.2892 rad/s), so the run starts with the impact of
    the initial contacts; the .dat was produced by an older revision (see
    scene.rimless_wheel_regress_scene), whose integrator differs in the first step by 2e-4 of the
    step length -- the pivot is an inverted pendulum, so that offset grows: 1e-5 over the first
    100 rows, 2.5e-3 at the end of the 6.27 s (the wheel passes its unstable top at 3.4 s)."""
    g = np.load(os.path.join(GOLD, "rimless_wheel_dat.npz"))
    sc = S.rimless_wheel_regress_scene()
    st = S.rimless_wheel_state((0.24,))[0].copy()
    aux = S.new_aux(1)
    np.testing.assert_allclose(st[:7], g["rows"][0][1:], atol=1e-6)
    oracle.world_handle_impacts(sc, st, aux)
    n = int(g["n_rows"]) - 1
    traj = oracle.world_step(sc, st, aux, 1e-3, n)["traj"][:, 0, :]
    assert aux["status"][0] == 0
    for row, k in zip(g["rows"], g["row_index"]):
        if k == 0:
            continue
        np.testing.assert_allclose(traj[k - 1], row[1:], rtol=0, atol=(1e-5 if k <= 100 else 2.5e-3), err_msg="row %d" % k)
    # one no-slip LCP per step while a spoke is down; no conservative-advancement sub-steps
    assert aux["mini_steps"][0] == n
    assert n - 200 < aux["lcp_solves"][0] <= n + 1


def test_spoke_change_impact_ratio(oracle):
    """Tree scene (J = 1 about the axis, wheel.xml:45): when the next spoke lands the no-slip
    impact maps the pivot rate by (J + m R^2 cos(2 pi / 6)) / (J + m R^2) = 0.75."""
    sc = S.rimless_wheel_scene()
    st = S.rimless_wheel_state((0.5,))[0].copy()
    aux = S.new_aux(1)
    oracle.world_handle_impacts(sc, st, aux)
    w_prev, ratios = st[11], []
    for _ in range(4000):
        oracle.world_step(sc, st, aux, 1e-3, 1, want_traj=False)
        if st[11] < 0.9 * w_prev:
            ratios.append(st[11] / w_prev)
        w_prev = st[11]
    assert aux["status"][0] == 0
    assert len(ratios) >= 1
    np.testing.assert_allclose(ratios, 0.75, rtol=2e-3)


def test_wheel_stays_in_the_plane(oracle):
    sc = S.rimless_wheel_scene()
    st = S.rimless_wheel_state((0.4,))[0].copy()
    aux = S.new_aux(1)
    oracle.world_step(sc, st, aux, 1e-3, 3000, want_traj=False)
    assert aux["status"][0] == 0
    assert abs(st[1]) < 1e-8 and abs(st[3]) < 1e-8 and abs(st[5]) < 1e-8     # y, qx, qz
    assert st[2] > 0.86                                                        # never sinks below the spoke height
