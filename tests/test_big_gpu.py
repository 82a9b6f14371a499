"""GPU parity of the large-world stepper (include/moby_hip_stack.h, through the C ABI) against the oracle: bit-exact
body states, rand() streams, status bits and counters after full TimeSteppingSimulator::step calls (conservative
advancement, contact generation, the island loop of process_constraints, stabilisation with Ridders), and of
ConstraintStabilization::stabilize alone (seam B3)."""
import numpy as np
import pytest

from moby_amd import scene as S
from moby_amd import stack as K
from tests.test_oracle_big import big_from_small

pytestmark = pytest.mark.gpu

FIELDS = ("rng", "time", "status", "steps", "mini_steps", "lcp_solves", "lcp_rows", "lcp_pivots", "stab_iters", "lcp_alg_bytes", "stab_rows",
          "zlast_size", "zbuf_size", "zbuf_cap")


def run_both(oracle, sc, st0, dt, nsteps, chunks=1, mode=0):
    """GPU batch vs one oracle world at a time; returns (gpu state, gpu aux, oracle state, oracle aux)."""
    B = st0.shape[0]
    bb = K.BigBatch(sc, st0)
    cap = bb.cap
    st_o = st0.copy(); aux_o = S.new_aux(B); zl = np.zeros((B, cap)); zb = np.zeros((B, cap))
    for _ in range(chunks):
        if mode == 0:
            bb.step(dt, nsteps)
        else:
            bb.stabilize()
        for w in range(B):
            oracle.big_step(sc, st_o[w], aux_o[w:w + 1], dt, nsteps, zlast=zl[w], zbuf=zb[w], cap=cap, mode=mode)
    st_g, aux_g = bb.download()
    ss = bb.solver_state()
    bb.close()
    return st_g, aux_g, st_o, aux_o, ss, zl, zb


def assert_parity(st_g, aux_g, st_o, aux_o, ss=None, zl=None, zb=None):
    for f in FIELDS:
        assert np.array_equal(aux_g[f], aux_o[f]), "%s: gpu %r oracle %r" % (f, aux_g[f], aux_o[f])
    assert np.array_equal(st_g, st_o), "max |diff| = %.3e" % np.abs(st_g - st_o).max()
    if ss is not None:
        for w in range(st_g.shape[0]):
            n = int(aux_o["zlast_size"][w]); c = int(aux_o["zbuf_cap"][w])
            assert np.array_equal(ss["zlast"][w, :n], zl[w, :n]) and np.array_equal(ss["zbuf"][w, :c], zb[w, :c])


@pytest.mark.parametrize("nboxes,B,nsteps", [(1, 4, 6), (2, 4, 6), (3, 4, 5), (5, 3, 4)])
def test_box_stacks_step_like_the_oracle(oracle, nboxes, B, nsteps):
    """Config 4 in the small (n = 32 nboxes: wave solver at 1-2 boxes, block solver above), perturbed worlds."""
    sc = K.box_stack_scene(nboxes)
    st0 = K.box_stack_state(nboxes, B)
    r = run_both(oracle, sc, st0, 1e-3, nsteps)
    assert_parity(*r)
    assert (r[1]["lcp_rows"] > 0).all() and (r[1]["steps"] == nsteps).all()


def test_sphere_stack_through_the_large_world_path(oracle):
    """BASELINE config 2's scene through this stepper: the same trajectories, bit for bit, as the oracle -- and hence as
    the one-wavefront kernel of moby_hip.h, which the same oracle checks (tests/test_world_gpu.py)."""
    sc = big_from_small(S.sphere_stack_scene())
    st0 = S.sphere_stack_state_range(0, 6)
    r = run_both(oracle, sc, st0, 1e-3, 20, chunks=2)
    assert_parity(*r)
    assert (r[1]["stab_iters"] > 0).any() and (r[1]["lcp_rows"] > 0).all()


def test_dropped_boxes_conservative_advancement_and_restitution(oracle):
    """A tumbling box dropped on a box lying on the plane (vertex-face pair), mu = 0.4, epsilon = 0.3: conservative-
    advancement sub-steps, intermittent contacts, second solves."""
    dims = [(2.0, 1.0, 2.0), (0.8, 0.6, 0.7)]
    mass = [20.0, 3.0]
    inertia = [[m / 12.0 * (y * y + z * z), m / 12.0 * (x * x + z * z), m / 12.0 * (x * x + y * y)] for m, (x, y, z) in zip(mass, dims)]
    sc = K.BigScene([S.MH_GEOM_BOX] * 2, dims, mass, inertia, [(0, 2, 0), (1, 2, 0), (0, 1, K.MH_PAIR_VERTEX_FACE)], gravity=(0, -9.81, 0),
                    mu_coulomb=0.4, epsilon=0.3, lcp_n_max=192)
    B = 4
    rng = np.random.default_rng(5)
    st = np.zeros((B, 2, 13)); st[:, :, 6] = 1.0
    st[:, 0, 1] = 0.5
    st[:, 1, 1] = 1.0 + 0.3 + 0.05 + 0.02 * rng.random(B); st[:, 1, 8] = -1.0
    q = np.concatenate([0.05 * rng.standard_normal((B, 3)), np.ones((B, 1))], axis=1); st[:, 1, 3:7] = q / np.linalg.norm(q, axis=1)[:, None]
    st[:, 1, 10:13] = 0.5 * rng.standard_normal((B, 3))
    r = run_both(oracle, sc, st.reshape(B, -1), 1e-3, 40, chunks=3)
    assert_parity(*r)
    assert (r[1]["mini_steps"] > r[1]["steps"]).all() and (r[1]["lcp_solves"] > 0).all()


def test_two_stacks_in_one_world_are_two_islands(oracle):
    """The island loop of process_constraints (ICH:105-151) and of the stabiliser: two stacks of two boxes side by side share
    no body, so every step solves two impact LCPs of 64 rows instead of one of 128 -- in the reference's island order."""
    dims = [K.box_dims(0), K.box_dims(1), K.box_dims(0), K.box_dims(1)]
    mass = [10.0 * x * y * z for x, y, z in dims]
    inertia = [[m / 12.0 * (y * y + z * z), m / 12.0 * (x * x + z * z), m / 12.0 * (x * x + y * y)] for m, (x, y, z) in zip(mass, dims)]
    pairs = [(b, 4, 0) for b in range(4)] + [(0, 1, K.MH_PAIR_VERTEX_FACE), (2, 3, K.MH_PAIR_VERTEX_FACE)]
    sc = K.BigScene([S.MH_GEOM_BOX] * 4, dims, mass, inertia, pairs, gravity=(0, -9.81, 0), mu_coulomb=0.2, lcp_n_max=128)
    B = 3
    st = np.zeros((B, 4, 13)); st[:, :, 6] = 1.0
    st[:, :, 1] = (0.5, 1.5, 0.5, 1.5); st[:, 2:, 0] = 5.0
    rng = np.random.default_rng(2)
    st[1:, :, 7:13] += 1e-3 * rng.standard_normal((B - 1, 4, 6))
    r = run_both(oracle, sc, st.reshape(B, -1), 1e-3, 5)
    assert_parity(*r)
    assert (r[1]["lcp_solves"] >= 2 * 5).all()                 # two islands per step (+ the stabiliser's)
    assert (r[1]["lcp_rows"] <= r[1]["lcp_solves"] * 64).all()  # and none of them the 128-row LCP of a single island


def test_stabilize_alone_is_the_b3_seam(oracle):
    """ConstraintStabilization::stabilize (include/Moby/ConstraintStabilization.h:24) on interpenetrating stacks:
    configurations change, velocities come back as they were, the violation is gone -- and every bit equals the oracle."""
    N, B = 3, 4
    sc = K.box_stack_scene(N, cstab_max_iterations=50)
    st = K.box_stack_state(N, B).reshape(B, N, 13)
    rng = np.random.default_rng(9)
    st[:, :, 1] -= 1e-4 * (1 + np.arange(N)) * (1.0 + rng.random((B, 1)))          # every interface penetrates by ~1e-4
    st[:, :, 7:13] = rng.standard_normal((B, N, 6))
    st0 = st.reshape(B, -1).copy()
    r = run_both(oracle, sc, st0, 0.0, 0, mode=1)
    assert_parity(*r)
    g = r[0].reshape(B, N, 13)
    assert np.array_equal(g[:, :, 7:13], st[:, :, 7:13])                                 # velocities restored
    gaps = np.diff(np.concatenate([np.zeros((B, 1)), g[:, :, 1] - 0.5], axis=1), axis=1) - np.array([0.0] + [1.0] * (N - 1))
    assert (gaps > -1e-9).all() and (r[1]["stab_iters"] >= 1).all()


def test_box_tower_scene_file_steps_like_the_oracle(oracle):
    """tests/scenes/three_box_tower.xml (what example/stacks/stack.xml's simulator holds; tests/test_io.py compares the two) through
    mh_io_load_xml -> BigScene.from_scene -> mh_big_batch: n = 96 impact LCPs (block solver), perturbed copies, 12 steps."""
    import os
    from moby_amd import io as mio
    sc1, st1, _, _ = mio.load_xml(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes", "three_box_tower.xml"))
    sc = K.BigScene.from_scene(sc1, st1)
    B = 4
    st0 = np.tile(st1.reshape(1, -1), (B, 1)).reshape(B, 3, 13)
    rng = np.random.default_rng(11)
    st0[1:, :, 7:13] += 1e-3 * rng.standard_normal((B - 1, 3, 6))
    r = run_both(oracle, sc, st0.reshape(B, -1), 1e-3, 6, chunks=2)
    assert_parity(*r)
    assert (r[1]["lcp_rows"] >= 12 * 96).all() and (r[1]["status"] & ~S.MH_WORLD_IMPACT_TOL == 0).all()


# (n = 512 against the oracle: test_config4_bench_size_full_batch below -- two worlds of the 1024-world batch over two steps, the default full-chip schedule;
#  round 4's 16 boxes x 2 worlds x 1 step test is subsumed by it and by the multi-step tests here, which run the small-batch branch)
@pytest.mark.parametrize("nboxes,B,nsteps", [(8, 3, 3), (12, 2, 2)])
def test_tall_stacks_several_steps_match_the_oracle(oracle, nboxes, B, nsteps):
    """The block solver (n = 256 / 384) against the ORACLE over several steps with every scheduling switch at its default: the
    structure-exploiting LU with factor reuse across Lemke pivots, the ladder as tasks handed out by need, lcp_fast's repetitions
    skipped.  Step 2 on enters lcp_fast warm-started from _zlast (ICH-QP:158-162, 233), fails, and walks the ladder again: states,
    rand() streams, every counter and the warm-start vectors themselves equal oracle.big_step bit for bit."""
    sc = K.box_stack_scene(nboxes)
    st0 = K.box_stack_state(nboxes, B)
    r = run_both(oracle, sc, st0, 1e-3, 1, chunks=nsteps)          # one step per launch: the warm start crosses launches too
    assert_parity(*r)
    aux = r[1]
    assert (aux["steps"] == nsteps).all() and (aux["zlast_size"] == 32 * nboxes).all()
    assert (aux["lcp_pivots"][1:] > 2000).all()                     # the perturbed worlds went through the Lemke ladder
    assert (aux["lcp_solves"] >= nsteps).all()


CONFIG4_BOXES = 16        # the bench size, n = 512: what bench.py's config-4 leg runs (NOT BASELINE's 64 boxes)


def test_config4_bench_size_properties():
    """BASELINE config 4 at the size bench.py names, many worlds, one full step: no world fails, the stack stays put,
    and identical worlds give identical results wherever they sit in the batch."""
    N, B = CONFIG4_BOXES, 128
    sc = K.box_stack_scene(N)
    st0 = K.box_stack_state(N, B)
    st0[B // 2:] = st0[:B // 2]                                   # second half = copy of the first
    bb = K.BigBatch(sc, st0)
    bb.step(1e-3, 1)
    st, aux = bb.download()
    bb.close()
    assert np.array_equal(st[B // 2:], st[:B // 2]) and np.array_equal(aux["lcp_pivots"][B // 2:], aux["lcp_pivots"][:B // 2])
    assert ((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) == 0).all()
    assert (aux["steps"] == 1).all() and (aux["lcp_rows"] >= 32 * N).all()
    b = st.reshape(B, N, 13)
    assert np.abs(b[:, :, 1] - 0.5 - np.arange(N)).max() < 1e-4     # nothing sank or flew
    assert np.abs(b[:, :, 7:13]).max() < 5e-2


def test_config4_bench_size_full_batch():
    """BASELINE config 4 at the bench size (16 boxes, impact LCP n = 512) AND the batch size the configuration names: 1024 worlds, TWO full
    TimeSteppingSimulator::step calls, the first cold, the second warm-started from _zlast (what bench.py's `config4_full_step` leg runs first).
    This is the batch on which the DEFAULT full-chip schedule is active (mh_impact.hip core_solve_round: lcp_fast's kernel first, the ladder's tasks by
    verdict behind the gate on a second stream, a second launch for the late verdicts), so the oracle is held against it directly: world 0 and the two
    worlds that need the most pivots, stepped by oracle.big_step for the same two steps (a committed fixture) -- state, rand() stream, every counter, flags,
    the warm-start vector _zlast, bit for bit.  Beside that: no world fails, identical worlds give identical results wherever they sit in the batch, the stacks stay
    put, momentum is what gravity put in, and the solver chain did the work the CPU oracle does on such worlds (thousands of pivots each)."""
    N, B, STEPS = CONFIG4_BOXES, 1024, 2
    sc = K.box_stack_scene(N)
    st0 = K.box_stack_state(N, B)
    st0[B // 2:] = st0[:B // 2]
    bb = K.BigBatch(sc, st0)
    cap = bb.cap
    bb.step(1e-3, 1)
    _, aux1 = bb.download()
    bb.step(1e-3, 1)
    st, aux = bb.download()
    ss = bb.solver_state()
    bb.close()
    assert ((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) == 0).all()                   # no failed world
    assert ((aux["status"] & S.MH_WORLD_IMPACT_TOL) != 0).sum() <= B // 50         # the handler's tolerance warning stays rare
    assert np.array_equal(st[B // 2:], st[:B // 2])
    for f in ("lcp_pivots", "lcp_rows", "lcp_solves", "rng", "stab_iters", "status"):
        assert np.array_equal(aux[f][B // 2:], aux[f][:B // 2]), f
    assert (aux["steps"] == STEPS).all() and (aux["lcp_rows"] >= 32 * N * STEPS).all()
    b = st.reshape(B, N, 13)
    assert np.abs(b[:, :, 1] - 0.5 - np.arange(N)).max() < 1e-5                    # heights (measured 4.8e-7)
    assert np.abs(b[:, :, 7:13]).max() < 5e-3                                       # at rest up to one step of gravity
    mom = (sc.mass[None, :] * b[:, :, 8]).sum(axis=1)                               # what is left is carried by the ground
    assert np.abs(mom).max() < 1e-3 * sc.mass.sum() * 9.81e-3 * 10
    piv = aux1["lcp_pivots"].astype(np.int64)
    assert 2000 < piv.mean() < 40000 and piv.max() < 200000                        # cold step, measured: mean 11 534, max 27 286
    assert aux["lcp_solves"].min() >= STEPS + 1 and aux["stab_rows"].mean() > 100  # impact (+ stabilisation) LCPs were solved in every step
    # the oracle on worlds of THIS batch: the unperturbed one and the two that need the most pivots over both steps (497: 42 328, 328: 38 841), from the fixture the
    # oracle wrote in the build container (MH_FIXTURE_BOXES=16 MH_FIXTURE_BATCH=1024 MH_FIXTURE_STEPS=2 python tests/golden/make_config4_64_boxes.py 0 497 328: a minute
    # and a half of CPU per perturbed world -- round 5 first ran the oracle here, 90 s of the GPU suite)
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_16_boxes_x1024_2steps.npz"))
    assert int(d["boxes"]) == N and int(d["batch"]) == B and int(d["steps"]) == STEPS and len(d["worlds"]) >= 3
    assert int(np.argmax(aux["lcp_pivots"][:B // 2])) in [int(w) for w in d["worlds"]]          # (the fixture does hold the hardest world of the batch)
    for k, w in enumerate(d["worlds"]):
        w = int(w); ao = d["aux"][k:k + 1]
        assert w < B // 2 and np.array_equal(d["st0"][k], st0[w])
        for f in FIELDS:
            assert np.array_equal(aux[f][w], ao[f][0]), "world %d %s: gpu %r oracle %r" % (w, f, aux[f][w], ao[f][0])
        assert np.array_equal(st[w], d["st"][k]), "world %d: max |diff| = %.3e" % (w, np.abs(st[w] - d["st"][k]).max())
        n = int(ao["zlast_size"][0])
        assert np.array_equal(ss["zlast"][w, :n], d["zlast"][k][:n]), "world %d: _zlast" % w


def test_config4_largest_solvable_size_properties():
    """32 boxes per stack (impact LCP n = 1024, the 1024-thread geometry with the left-looking LU): the largest stack of BASELINE config 4's family that the
    reference's own solver chain solves (64 boxes, n = 2048, defeats it: tests/test_oracle_compact_lu.py, profiles/r04_a_config4_64_boxes_x8_impact_call.json).
    One full step of 8 worlds (the batch of 1024: profiles/r04_a_config4_32_boxes_x1024.json): no world fails, identical worlds agree, the stacks stay put."""
    N, B = 32, 8
    sc = K.box_stack_scene(N)
    st0 = K.box_stack_state(N, B)
    st0[B // 2:] = st0[:B // 2]
    bb = K.BigBatch(sc, st0)
    bb.step(1e-3, 1)
    st, aux = bb.download()
    bb.close()
    assert ((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) == 0).all()
    assert np.array_equal(st[B // 2:], st[:B // 2]) and np.array_equal(aux["lcp_pivots"][B // 2:], aux["lcp_pivots"][:B // 2])
    assert (aux["steps"] == 1).all() and (aux["lcp_rows"] >= 32 * N).all()
    b = st.reshape(B, N, 13)
    assert np.abs(b[:, :, 1] - 0.5 - np.arange(N)).max() < 1e-4
    assert np.abs(b[:, :, 7:13]).max() < 5e-2


def test_config4_at_the_size_baseline_states_against_the_oracles_fixture():
    """BASELINE config 4 at the size BASELINE states: stacks of 64 boxes (impact LCP n = 2048; the pattern of example/stacks/stack.xml:36-96,
    regress/stacks.setup's dt), 8 worlds, one full TimeSteppingSimulator::step.  The CPU oracle needs minutes per world here, so what it leaves behind
    was generated in the build container (tests/golden/make_config4_64_boxes.py -> tests/golden/config4_64_boxes.npz) and is held against the device bit
    for bit: state, status flags, rand() ring, every counter, time, and the handler's _zlast, for the unperturbed world (solved: lcp_fast fails on all four
    rungs, the Lemke ladder gets through) and a perturbed one (the reference's whole chain fails: LCPSolverException, ICH-QP:225 -- the step is left
    there and not counted, tests/test_oracle_exception.py).  lcp_lemke's 2048-row bases go through the structure-exploiting LU with two rows per lane
    (mh_lcp_blkx.hip); until round 5 every pivot there was a dense dgesv and this step did not finish inside a 1200-second GPU call."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_64_boxes.npz"))
    N, B = int(d["boxes"]), int(d["batch"])
    assert N == 64 and int(d["steps"]) == 1
    sc = K.box_stack_scene(N)
    st0 = K.box_stack_state(N, B)
    bb = K.BigBatch(sc, st0)
    assert bb.cap == 32 * N
    bb.step(float(d["dt"]), 1)
    st, aux = bb.download()
    ss = bb.solver_state()
    bb.close()
    assert len(d["worlds"]) >= 2
    for k, w in enumerate(d["worlds"]):
        w = int(w); ao = d["aux"][k:k + 1]
        assert np.array_equal(d["st0"][k], st0[w])                       # the fixture's start state is this batch's
        for f in FIELDS:
            assert np.array_equal(aux[f][w], ao[f][0]), "world %d %s: gpu %r oracle %r" % (w, f, aux[f][w], ao[f][0])
        assert np.array_equal(st[w], d["st"][k]), "world %d: max |diff| = %.3e" % (w, np.abs(st[w] - d["st"][k]).max())
        n = int(ao["zlast_size"][0])
        assert np.array_equal(ss["zlast"][w, :n], d["zlast"][k][:n])
    failed = (aux["status"] & S.MH_WORLD_LCP_FAILED) != 0
    assert not failed[0] and aux["steps"][0] == 1 and aux["zlast_size"][0] == 32 * N
    assert failed[1:].all() and (aux["steps"][1:] == 0).all() and (aux["time"][1:] == 0.0).all()       # the perturbed stacks defeat the reference's chain
    assert (aux["lcp_pivots"] > 8000).all()
    b = st.reshape(B, N, 13)
    assert np.abs(b[0, :, 1] - 0.5 - np.arange(N)).max() < 1e-4 and np.abs(b[0, :, 7:13]).max() < 5e-2


def test_config4_largest_solvable_size_against_the_oracles_fixture():
    """32 boxes per stack (impact LCP n = 1024: the largest stack of config 4's family that the reference's own chain solves), 8 worlds, TWO full steps -- a cold one
    and one warm-started from _zlast -- against tests/golden/config4_32_boxes_2steps.npz, which the CPU oracle wrote in the build container (tests/golden/
    make_config4_64_boxes.py with MH_FIXTURE_BOXES=32 MH_FIXTURE_STEPS=2: 10-25 minutes of CPU per world): state, flags, rand() ring, every counter, time and _zlast of
    the worlds the fixture holds, bit for bit.  n = 1024 goes through the 1024-thread geometry for lcp_fast and (176 ladder tasks: fewer than two per CU) for the
    lcp_lemke kinds as well."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_32_boxes_2steps.npz")
    d = np.load(path)
    N, B, steps = int(d["boxes"]), int(d["batch"]), int(d["steps"])
    assert N == 32 and steps == 2
    sc = K.box_stack_scene(N)
    st0 = K.box_stack_state(N, B)
    bb = K.BigBatch(sc, st0)
    bb.step(float(d["dt"]), steps)
    st, aux = bb.download()
    ss = bb.solver_state()
    bb.close()
    assert len(d["worlds"]) >= 2
    for k, w in enumerate(d["worlds"]):
        w = int(w); ao = d["aux"][k:k + 1]
        assert np.array_equal(d["st0"][k], st0[w])
        for f in FIELDS:
            assert np.array_equal(aux[f][w], ao[f][0]), "world %d %s: gpu %r oracle %r" % (w, f, aux[f][w], ao[f][0])
        assert np.array_equal(st[w], d["st"][k]), "world %d: max |diff| = %.3e" % (w, np.abs(st[w] - d["st"][k]).max())
        n = int(ao["zlast_size"][0])
        assert np.array_equal(ss["zlast"][w, :n], d["zlast"][k][:n])
    assert ((aux["status"] & ~S.MH_WORLD_IMPACT_TOL) == 0).all() and (aux["steps"] == steps).all()


def test_every_route_through_the_block_solver_gives_the_same_full_steps():
    """The same six 12-box worlds (LCPs of 384 rows: every geometry's own size range, round 5: was 16 boxes and 64 s of the suite), one full step, through every switch of the workgroup-per-problem solver -- states, rand() streams, pivot counts,
    flags and warm-start sizes equal bit for bit (the oracle checks the default choice: the tests above and test_config4_bench_size_full_batch):
      * the lcp_lemke kinds' thread geometries (mh_debug_set key 2: 256 / 1024 / 64 / 128 threads per problem; panels of 16 / 16 / 8 / 12 columns,
        rounds of 16 / 16 / 4 / 8 steps in the left-looking LU);
      * the two LU routes for Lemke's bases (key 3: the structure-exploiting one / the dense dgesv on the assembled basis);
      * the factors of the columns before the changed one kept from pivot to pivot or not (key 6), with the ladder as tasks beside lcp_fast, as tasks
        after it (handed out by need, key 7, or by block index) and in sequence (key 4).
    (Round 4 had three tests of 6 / 12 / 12 worlds for this: 77 s of the GPU suite.)"""
    from moby_amd import _lib
    N, B = 12, 6
    sc = K.box_stack_scene(N)
    st0 = K.box_stack_state(N, B)
    lib = _lib.load()
    defaults = {2: 0, 3: 1, 4: 3, 6: 1, 7: 1}
    runs = [{}, {2: 1}, {2: 3}, {2: 4}, {3: 0}, {4: 2, 6: 0}, {4: 0}, {4: 1, 7: 0}]
    res = []
    try:
        for sw in runs:
            for k, v in defaults.items():
                _lib.check(lib.mh_debug_set(k, sw.get(k, v)))
            bb = K.BigBatch(sc, st0)
            bb.step(1e-3, 1)
            res.append(bb.download())
            bb.close()
    finally:
        for k, v in defaults.items():
            _lib.check(lib.mh_debug_set(k, v))
    for sw, r in zip(runs[1:], res[1:]):
        assert np.array_equal(res[0][0], r[0]), sw
        for f in FIELDS:
            assert np.array_equal(res[0][1][f], r[1][f]), (sw, f)
    assert (res[0][1]["lcp_pivots"] > 300).all()


def test_ladder_tasks_behind_the_gate_change_nothing(oracle):
    """A batch that fills the chip with LCPs of 384 rows (12-box stacks x 768 worlds: mh_impact.hip core_solve_round's `full_chip`): by default
    (mh_debug_set key 4 = 3) the Lemke ladder's tasks are launched on a second stream behind lcp_fast's kernel and a gate that opens when its
    last workgroup has started, and handed out by lcp_fast's VERDICT per world; with key 4 = 2 they start after lcp_fast has finished.  The batch is
    mixed on purpose: every third world's stack is moving UP (nothing impacts: the world is masked out of the round and no verdict is ever published
    for it), every third world has boxes 1-11 lifted clear (one interface left: an LCP of 32 rows, which belongs to the wave solver of the same call and
    gets no verdict either), the rest are the perturbed resting stacks.  One full step both ways: states, rand() streams, pivot counts, flags and
    the warm-start sizes equal bit for bit -- and one world of each kind against the oracle."""
    from moby_amd import _lib
    N, B = 12, 768
    sc = K.box_stack_scene(N)
    st0 = K.box_stack_state(N, B).reshape(B, N, 13)
    st0[1::3] = K.box_stack_state(N, 1).reshape(1, N, 13)        # the unperturbed stack ...
    st0[1::3, :, 8] = 0.05                                       # ... moving up as one: no impacting contact anywhere
    st0[2::3, 1:, 1] += 0.1                                      # boxes 1.. lifted: only the ground interface is in contact
    st0 = st0.reshape(B, N * 13)
    lib = _lib.load()
    res = {}
    try:
        for tasks in (3, 2):
            _lib.check(lib.mh_debug_set(4, tasks))
            bb = K.BigBatch(sc, st0)
            bb.step(1e-3, 1)
            res[tasks] = bb.download()
            bb.close()
    finally:
        _lib.check(lib.mh_debug_set(4, 3))
    assert np.array_equal(res[3][0], res[2][0])
    for f in FIELDS:
        assert np.array_equal(res[3][1][f], res[2][1][f]), f
    aux = res[3][1]
    assert (aux["lcp_pivots"][0::3] > 500).mean() > 0.9 and (aux["status"] & S.MH_WORLD_LCP_FAILED == 0).mean() > 0.9
    assert (aux["lcp_rows"][1::3] < 384).all() and (aux["lcp_rows"][0::3] >= 384).all()      # (the rising stacks solve their stabilisation LCPs only: no impact LCP, no verdict)
    for w in (3, 4, 5):                                          # one world of each kind through the oracle
        so = st0[w].copy(); ao = S.new_aux(1)
        oracle.big_step(sc, so, ao, 1e-3, 1)
        for f in FIELDS:
            assert np.array_equal(aux[f][w], ao[f][0]), "world %d %s: gpu %r oracle %r" % (w, f, aux[f][w], ao[f][0])
        assert np.array_equal(res[3][0][w], so), w


def test_upload_restores_or_resets_the_handlers_warm_start():
    """mh_big_batch_upload(state, aux) carries the warm-start SIZES (_zlast / _z) like the one-wavefront batch's aux does:
    (i) download + save_solver_state, a NEW batch, upload + load_solver_state, step == the uninterrupted run, bit for bit;
    (ii) a used batch given a fresh state and a zeroed aux behaves like a fresh batch (the next solve is cold)."""
    N, B = 3, 4
    sc = K.box_stack_scene(N)
    st0 = K.box_stack_state(N, B)
    ref = K.BigBatch(sc, st0); ref.step(1e-3, 5); st_ref, aux_ref = ref.download(); ref.close()
    a = K.BigBatch(sc, st0); a.step(1e-3, 3)
    st_mid, aux_mid = a.download(); ss = a.solver_state()
    assert (aux_mid["zlast_size"] > 0).all()                         # there IS a warm start to lose
    b = K.BigBatch(sc, st0)                                          # (i) resume in another batch
    b.upload(st_mid, aux_mid); b.load_solver_state(ss); b.step(1e-3, 2)
    st_b, aux_b = b.download(); b.close()
    assert np.array_equal(st_b, st_ref)
    for f in FIELDS:
        assert np.array_equal(aux_b[f], aux_ref[f]), f
    fresh = K.BigBatch(sc, st0); fresh.step(1e-3, 2); st_f, aux_f = fresh.download(); fresh.close()
    a.upload(st0, S.new_aux(B)); a.step(1e-3, 2)                     # (ii) reuse with a fresh aux
    st_a, aux_a = a.download(); a.close()
    assert np.array_equal(st_a, st_f)
    for f in FIELDS:
        assert np.array_equal(aux_a[f], aux_f[f]), f


def test_cpp_stack_simulator_adapter_example():
    """moby_amd/cpp/MobyHipStackSimulator.h (TimeSteppingSimulator::step and ConstraintStabilization::stabilize for
    large worlds) through its example program."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cpp = os.path.join(root, "moby_amd", "cpp")
    exe = os.path.join(cpp, "example_stack")
    subprocess.check_call(["g++", "-std=c++11", os.path.join(cpp, "example_stack.cpp"), "-L" + os.path.join(root, "moby_amd"),
                           "-lmoby_hip", "-Wl,-rpath," + os.path.join(root, "moby_amd"), "-o", exe])
    out = subprocess.check_output([exe]).decode()
    assert "status=0/0" in out and "time=0.003" in out, out


def test_contact_constrained_pendulum_matches_oracle(oracle):
    """example/contact-constrained-pendulum (the plugin's six-contact pin, MH_GEOM_PIN) on the GPU: 48-row impact LCPs on
    opposing contacts every step, the stabiliser entered and abandoned every step -- bit for bit like the oracle, which
    tests/test_oracle_pendulum.py pins to the reference's 6.5 s recording."""
    sc = K.pendulum_scene()
    st0 = K.pendulum_state(4)
    r = run_both(oracle, sc, st0, 1e-3, 40, chunks=3)
    assert_parity(*r)
    assert (r[1]["lcp_rows"] >= 48 * 120).all()


def test_box_stacks_step_with_the_anitescu_potra_model(oracle):
    """The same full steps with the scene in the reference's -DUSE_AP configuration (impact LCPs [UL UR; LL 0] through the
    Lemke ladder, impulses through the contact wrenches); stabilisation is untouched by the model."""
    nboxes, B = 3, 4
    sc = K.box_stack_scene(nboxes, mu=0.3, impact_model=1)
    st0 = K.box_stack_state(nboxes, B)
    r = run_both(oracle, sc, st0, 1e-3, 5)
    assert_parity(*r)
    assert (r[1]["lcp_rows"] > 0).all() and (r[1]["zlast_size"] == 0).all()
    sc_ds = K.box_stack_scene(nboxes, mu=0.3)
    r_ds = run_both(oracle, sc_ds, st0, 1e-3, 5)
    assert (r_ds[1]["zlast_size"] > 0).all()
    assert np.abs(r[0] - r_ds[0]).max() < 1e-3             # two models of the same resting stack


def test_anitescu_potra_box_stacks_properties_at_scale(oracle):
    """256 worlds of 4-box stacks with the A-P impact model, three full steps: nothing sinks or flies, identical worlds give
    identical results wherever they sit in the batch.  On the redundant corner contacts the A-P Lemke ladder (capped at
    lambda = 1e-3, ICH-AP:333) now and then fails on every rung -- the reference throws there (an exception nothing catches: the run is over,
    tests/test_oracle_exception.py); those worlds are flagged MH_WORLD_LCP_FAILED exactly where the oracle flags them, their failing step is not
    counted, and they are not stepped again."""
    N, B = 4, 256
    sc = K.box_stack_scene(N, mu=0.3, impact_model=1)
    st0 = K.box_stack_state(N, B)
    st0[B // 2:] = st0[:B // 2]
    bb = K.BigBatch(sc, st0)
    bb.step(1e-3, 3)
    st, aux = bb.download()
    bb.close()
    assert np.array_equal(st[B // 2:], st[:B // 2]) and np.array_equal(aux["lcp_pivots"][B // 2:], aux["lcp_pivots"][:B // 2])
    failed = (aux["status"] & S.MH_WORLD_LCP_FAILED) != 0
    assert ((aux["status"] & ~(S.MH_WORLD_IMPACT_TOL | S.MH_WORLD_LCP_FAILED)) == 0).all() and (aux["steps"][~failed] == 3).all() and (aux["steps"][failed] < 3).all() and 1 <= failed.sum() <= B // 16
    assert (aux["zlast_size"] == 0).all() and (aux["lcp_rows"] > 0).all()
    b = st.reshape(B, N, 13)
    assert np.abs(b[:, :, 1] - 0.5 - np.arange(N)).max() < 1e-4 and np.abs(b[:, :, 7:13]).max() < 5e-2
    check = sorted(set(list(np.where(failed)[0][:2]) + [0, 5, 17]))
    for w in check:                                              # the oracle agrees world by world, failures included
        so = st0[w].copy(); ao = S.new_aux(1)
        oracle.big_step(sc, so, ao, 1e-3, 3)
        assert ao["status"][0] == aux["status"][w] and ao["lcp_pivots"][0] == aux["lcp_pivots"][w] and np.array_equal(so, st[w])
        for f in ("steps", "mini_steps", "time", "stab_iters", "lcp_solves", "rng"):
            assert np.array_equal(ao[f][0], aux[f][w]), (w, f)
