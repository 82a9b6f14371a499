"""The impact handler's exceptions end the run (oracle side; tests/test_big_gpu.py holds the device to the same worlds).

LCPSolverException (/root/reference/src/ImpactConstraintHandlerQP.cpp:225), std::runtime_error("Unable to solve constraint LCP!")
(ImpactConstraintHandler.cpp:1282) and std::exception (ImpactConstraintHandlerLCP.cpp:334) are caught nowhere between apply_model and
main(): ConstraintSimulator.cpp:342-350 catches ImpactToleranceException only, programs/driver.cpp and programs/regress.cpp catch nothing.
The throw therefore unwinds process_constraints, do_mini_step (before current_time += h, TimeSteppingSimulator.cpp:215) and step (before
stabilize, :97), and the process that owned the simulator terminates.  The oracle restates that: MH_WORLD_LCP_FAILED is set, the step is
left where the exception left it, and the world is never stepped again."""
import os

import numpy as np
import pytest

from moby_amd import scene as S, stack as K

N, B, W_FAILS = 4, 256, 82      # world 82 of the 256-world batch of 4-box stacks: the A-P Lemke ladder (ICH-AP:333) fails on every rung in its third step


@pytest.fixture
def ap_oracle(oracle):
    oracle.set_impact_model(1)
    yield oracle
    oracle.set_impact_model(0)


def test_a_failed_impact_lcp_ends_the_step_and_the_run(ap_oracle):
    sc = K.box_stack_scene(N, mu=0.3, impact_model=1)
    st0 = K.box_stack_state(N, B)[W_FAILS]
    so = st0.copy(); ao = S.new_aux(1)
    ap_oracle.big_step(sc, so, ao, 1e-3, 2)
    assert ao["status"][0] & S.MH_WORLD_LCP_FAILED == 0 and ao["steps"][0] == 2
    s2 = so.copy(); t2 = float(ao["time"][0]); m2 = int(ao["mini_steps"][0]); stab2 = int(ao["stab_iters"][0])
    ap_oracle.big_step(sc, so, ao, 1e-3, 1)                       # the step that throws
    assert ao["status"][0] & S.MH_WORLD_LCP_FAILED
    assert ao["steps"][0] == 2                                    # step() did not return
    assert float(ao["time"][0]) == t2 and int(ao["mini_steps"][0]) == m2      # TSS:215 was not reached
    assert int(ao["stab_iters"][0]) == stab2                      # nor was stabilize (TSS:97)
    assert ao["lcp_solves"][0] > 0 and not np.array_equal(so, s2)  # the mini-step's position and velocity updates happened before the throw (TSS:133-192)
    # the run is over: further steps change nothing
    s3 = so.copy(); a3 = ao.copy()
    ap_oracle.big_step(sc, so, ao, 1e-3, 5)
    assert np.array_equal(so, s3) and ao.tobytes() == a3.tobytes()
    # in one call of three steps the outcome is the same
    sb = st0.copy(); ab = S.new_aux(1)
    ap_oracle.big_step(sc, sb, ab, 1e-3, 3)
    assert np.array_equal(sb, s3) and ab.tobytes() == a3.tobytes()
