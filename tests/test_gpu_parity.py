"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle
on identical, seeded / deterministic inputs.

Tolerances (fp32 positions, per-particle Euclidean distance, scene bounding box
diagonal ~14 for the cloth and ~10 for the bar):
  * known-answer projections and short horizons against the contraction-free float
    build of the reference: <= 1e-6 absolute after one sweep/step (the arithmetic
    is restated in the reference's operation order and compiled with
    -ffp-contract=off, so most cases are bit-exact; libm acos and a few double
    sub-expressions are the exceptions),
  * long horizons against the f64 build: GPU error must stay within a small factor
    of the f32 reference's own error against f64 (the envelope), since
    fp32 trajectories decorrelate on ill-conditioned constraint sets.
"""
import os

import numpy as np
import pytest

from tests import kat, util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pbd():
    import positionbaseddynamics_amd as pbd
    assert pbd.device_count() > 0, "GPU tests need a MI355X: the engine has no CPU fallback"
    return pbd


# ---------------------------------------------------------------------------
# known-answer tests: random disjoint constraints, one and two sweeps
# ---------------------------------------------------------------------------
KAT_TYPES = kat.KAT_TYPES

# types whose projection involves no libm call: must be bit-identical to the float reference
# (dihedral calls acos in double on both sides; device and glibc libm may differ in the last bit)
KAT_EXACT = set(KAT_TYPES) - {"dihedral"}


def _project_both(pbd, ops, x_start, sweeps):
    ref = util.get_oracle("f32")
    util.apply_ref(ref, ops)
    ref.set_time_step_size(0.005)
    ref.set_array(0, x_start)
    for it in range(sweeps):
        ref.solve_position_constraints(it)
    m = util.build_mine(ops)
    m.getParticles().set_array(0, x_start)
    pbd.TimeManager.setCurrent(pbd.TimeManager())
    ts = pbd.TimeStepController()
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
    assert np.array_equal(ref.constraint_types(), m.constraintTypes())
    ts.project(m, sweeps)
    return ref.positions(), m.getParticles().positions()


@pytest.mark.parametrize("type_name", KAT_TYPES)
def test_known_answer_projection(pbd, type_name):
    arrs = kat.kat_arrays(type_name, 257, seed=1234 + KAT_TYPES.index(type_name))
    ops = kat.kat_ops(type_name, arrs)
    worst = 0.0
    for sweeps in (1, 2):
        xr, xg = _project_both(pbd, ops, arrs["x_start"], sweeps)
        assert np.all(np.isfinite(xg))
        assert util.max_err(xr, arrs["x_start"]) > 1e-4, "degenerate test: the oracle did not move anything"
        err = util.max_err(xg, xr)
        worst = max(worst, err)
        if type_name in KAT_EXACT:
            assert util.bitwise_equal(xg, xr.astype(np.float32)), \
                "%s: not bit-identical to the float reference after %d sweep(s): max err %.3e, %d ulp" % (
                    type_name, sweeps, err, util.ulp_diff(xg, xr.astype(np.float32)))
        else:
            assert err <= 2e-6, "%s: max |dx| %.3e after %d sweep(s)" % (type_name, err, sweeps)
    print("KAT %-26s max |dx| vs float reference = %.3e" % (type_name, worst))


@pytest.mark.parametrize("type_name", KAT_TYPES)
def test_known_answer_projection_vs_golden(pbd, type_name):
    """Same, against the committed outputs of the reference (tests/golden/, float and double builds)."""
    g = np.load(os.path.join(util.ROOT, "tests", "golden", "kat_%s.npz" % type_name))
    ops = kat.kat_ops(type_name, g)
    for sweeps in (1, 2):
        m = util.build_mine(ops)
        m.getParticles().set_array(0, g["x_start"])
        pbd.TimeManager.setCurrent(pbd.TimeManager())
        ts = pbd.TimeStepController()
        ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
        ts.project(m, sweeps)
        xg = m.getParticles().positions()
        if type_name in KAT_EXACT:
            assert util.bitwise_equal(xg, g["x%d_f32" % sweeps]), type_name
        else:
            assert util.max_err(xg, g["x%d_f32" % sweeps]) <= 2e-6
        # against the double build: fp32 rounding of O(1..10)-sized data through one or two projections,
        # or -- where the float build of the reference ITSELF leaves that envelope (shape matching on
        # random tets: the polar decomposition flips branch in float) -- no worse than the float reference
        f32_vs_f64 = util.max_err(g["x%d_f32" % sweeps], g["x%d_f64" % sweeps])
        tol = max(1e-3 * max(1.0, util.max_err(g["x%d_f64" % sweeps], g["x_start"])), 1.0001 * f32_vs_f64)
        assert util.max_err(xg, g["x%d_f64" % sweeps]) <= tol


def test_fem_tet_inversion_branch(pbd):
    """Crushed / inverted tets take the SVD branch (MathFunctions.cpp:261-388)."""
    for tname in ("fem_tet", "fem_tet_xpbd"):
        arrs = kat.kat_arrays(tname, 128, seed=99, static_fraction=0.1)
        x_start = kat.invert_tets(arrs, 5)
        ops = kat.kat_ops(tname, arrs)
        xr, xg = _project_both(pbd, ops, x_start, 1)
        assert np.all(np.isfinite(xg))
        assert util.max_err(xr, x_start) > 1e-3
        err = util.max_err(xg, xr)
        exact = util.bitwise_equal(xg, xr.astype(np.float32))
        print("inversion branch %-14s max |dx| = %.3e  bit-exact=%s" % (tname, err, exact))
        # the Jacobi SVD with inversion handling is restated operation for operation (no libm call inside): bitwise
        assert exact, "%s inversion branch: max err %.3e, %d ulp" % (tname, err, util.ulp_diff(xg, xr.astype(np.float32)))


# ---------------------------------------------------------------------------
# whole-step parity on the BASELINE scenes (reduced sizes)
# ---------------------------------------------------------------------------
SCENES = {
    # name: (ops, sub_steps, iters, [(steps, tol_vs_f32_reference)], must_be_bit_exact_at_first_horizon)
    "C2 cloth 50x50 XPBD dist+bend 10it": (util.cloth_spec(50, 50, 4, 3), 1, 10, [(1, 1e-6), (10, 1e-5), (100, 1e-3)]),
    "C2 cloth 64x40 XPBD dist+bend 10it 3 substeps": (util.cloth_spec(64, 40, 4, 3), 3, 10, [(1, 1e-6), (10, 1e-5)]),
    "cloth 50x50 XPBD dist only": (util.cloth_spec(50, 50, 4, 0), 1, 10, [(1, 1e-6), (10, 1e-5)]),
    "cloth 50x50 FEM tri + dihedral": (util.cloth_spec(50, 50, 2, 1), 1, 5, [(1, 1e-5), (10, 1e-4)]),
    "cloth 50x50 strain tri": (util.cloth_spec(50, 50, 3, 0), 1, 5, [(1, 1e-6), (10, 1e-5)]),
    "C3 bar 30x5x5 FEM tet (2) 10it": (util.bar_spec(30, 5, 5, 2), 1, 10, [(1, 1e-6), (10, 1e-5)]),
    "C3 bar 30x5x5 XPBD dist+vol (6) 10it": (util.bar_spec(30, 5, 5, 6), 1, 10, [(1, 1e-6), (10, 1e-5), (100, 1e-3)]),
    "C3 bar 30x5x5 dist+vol (1) 10it": (util.bar_spec(30, 5, 5, 1), 1, 10, [(1, 1e-6), (10, 1e-5)]),
    "C3 bar 30x5x5 strain tet (4) 10it": (util.bar_spec(30, 5, 5, 4), 1, 10, [(1, 1e-6), (10, 1e-5)]),
    "C3 bar 30x5x5 shape matching (5) 10it": (util.bar_spec(30, 5, 5, 5), 1, 10, [(1, 1e-5), (10, 1e-4)]),
    "C3 bar 30x5x5 XPBD FEM tet (3) 1it x 5 substeps": (util.bar_spec(30, 5, 5, 3), 5, 1, [(1, 1e-5), (10, 1e-4)]),
}


SCENES.update({
    "irregular Delaunay cloth XPBD dist+bend": (util.delaunay_cloth_spec(900), 1, 10, [(1, 1e-6), (10, 1e-5)]),
    "irregular Delaunay cloth FEM tri + dihedral": (util.delaunay_cloth_spec(500, cloth_method=2, bending_method=1), 1, 5, [(1, 1e-5), (5, 1e-4)]),
    "irregular Delaunay tets FEM (2)": (util.delaunay_solid_spec(400, solid_method=2), 1, 10, [(1, 1e-6), (10, 1e-5)]),
    "irregular Delaunay tets XPBD dist+vol (6)": (util.delaunay_solid_spec(400, solid_method=6), 1, 10, [(1, 1e-6), (10, 1e-5)]),
    "kitchen sink: all 13 constraint types in one model, mixed-type colours": (util.kitchen_sink_spec(), 2, 3, [(1, 1e-5), (6, 1e-4)]),
})


@pytest.mark.parametrize("name", list(SCENES))
def test_scene_parity_vs_float_reference(pbd, name):
    ops, sub, iters, horizons = SCENES[name]
    for steps, tol in horizons:
        xr = util.oracle_positions(ops, steps, sub, iters, "f32")
        m, ts = util.mine_run(ops, steps, sub, iters)
        xg = m.getParticles().positions()
        assert np.all(np.isfinite(xg))
        err = util.max_err(xg, xr)
        exact = util.bitwise_equal(xg, xr.astype(np.float32))
        print("%-50s steps=%-4d max |dx| vs f32 reference = %.3e%s" % (name, steps, err, "  (bit-exact)" if exact else ""))
        assert err <= tol, "%s after %d steps: %.3e > %.1e" % (name, steps, err, tol)


@pytest.mark.parametrize("name", list(util.GOLDEN_SCENES))
def test_scene_parity_vs_golden(pbd, name):
    """Against the committed outputs of the reference's float and double builds (tests/golden/)."""
    ops, sub, iters, horizons = util.GOLDEN_SCENES[name]
    g = np.load(os.path.join(util.ROOT, "tests", "golden", "scene_%s.npz" % name))
    for steps in horizons:
        m, ts = util.mine_run(ops, steps, sub, iters, resident=(steps > 10))
        xg = m.getParticles().positions()
        e32 = util.max_err(xg, g["x_f32_%d" % steps])
        e64 = util.max_err(xg, g["x_f64_%d" % steps])
        eref = util.max_err(g["x_f32_%d" % steps], g["x_f64_%d" % steps])
        print("%-44s steps=%-4d |gpu-f32ref|=%.3e |gpu-f64ref|=%.3e |f32ref-f64ref|=%.3e" % (name, steps, e32, e64, eref))
        assert e32 <= 1e-5 * steps ** 0.5 + (2e-6 if "dihedral" in name else 0)
        assert e64 <= 3.0 * eref + 1e-6


def test_c1_plumbing_scene_envelope(pbd):
    """C1 (PBD distance + PBD isometric bending, 5 it) is ill-conditioned in float: two
    float builds of the reference differ by 1e-2 after one step.  The GPU must sit inside the
    envelope spanned by the reference's own float builds around the f64 result."""
    ops = util.cloth_spec(50, 50, 1, 2)
    x64 = util.oracle_positions(ops, 1, 1, 5, "f64")
    x32 = util.oracle_positions(ops, 1, 1, 5, "f32")
    m, _ = util.mine_run(ops, 1, 1, 5)
    xg = m.getParticles().positions()
    e_ref = util.max_err(x32, x64)
    e_gpu = util.max_err(xg, x64)
    print("C1: |f32 ref - f64| = %.3e   |gpu - f64| = %.3e   |gpu - f32 ref| = %.3e" % (e_ref, e_gpu, util.max_err(xg, x32)))
    assert e_gpu <= 3.0 * e_ref + 1e-6


def test_long_horizon_envelope_vs_f64(pbd):
    """100 steps of the C2 physics: GPU error against the f64 reference stays within 4x the f32
    reference's own error (trajectory-level statement)."""
    ops = util.cloth_spec(50, 50, 4, 3)
    x64 = util.oracle_positions(ops, 100, 1, 10, "f64")
    x32 = util.oracle_positions(ops, 100, 1, 10, "f32")
    m, _ = util.mine_run(ops, 100, 1, 10, resident=True)
    xg = m.getParticles().positions()
    e_ref = util.max_err(x32, x64)
    e_gpu = util.max_err(xg, x64)
    print("100 steps: |f32 ref - f64| = %.3e   |gpu - f64| = %.3e" % (e_ref, e_gpu))
    assert e_gpu <= 4.0 * e_ref + 1e-5


def test_full_state_and_second_order_velocity(pbd):
    """x, v, oldX, lastX all match, with the second-order velocity update (TimeIntegration.cpp:69-79)."""
    ops = util.cloth_spec(40, 40, 4, 3)
    o = util.oracle_run(ops, 5, 2, 4, "f32", vel_method=1)
    m, ts = util.mine_run(ops, 5, 2, 4, vel_method=1)
    pd = m.getParticles()
    for which, nm in ((0, "x"), (2, "v"), (4, "oldX"), (5, "lastX")):
        err = util.max_err(pd.array(which), o.get_array(which))
        print("second-order: %-5s max err %.3e" % (nm, err))
        assert err <= (1e-5 if which != 2 else 2e-3), nm
    # accelerations: gravity for dynamic particles, untouched (0) for the two pinned ones
    a = pd.array(3)
    assert np.allclose(a[1], [0, -9.81, 0]) and np.all(a[0] == 0)


def test_host_accelerations_follow_gravity_masses_and_host_writes(pbd):
    """TimeStep::clearAccelerations (TimeStep.cpp:28-62) on the host mirror: gravity for every dynamic particle, static ones keep what they
    had.  The engine skips the pass while nothing it depends on has changed -- so change each of those things between resident calls."""
    ops = util.cloth_spec(12, 12, 4, 3)
    m, ts = util.mine_run(ops, 2, 1, 2, resident=True)
    pd = m.getParticles()
    a = pd.array(3)
    assert np.allclose(a[1], [0, -9.81, 0]) and np.all(a[0] == 0)
    # gravity changes
    pbd.Simulation.getCurrent().setVecValueFloat(pbd.Simulation.GRAVITATION, [0.5, -3.0, 0.25])
    ts.stepResident(m, 1); ts.syncToHost(m)
    a = pd.array(3)
    assert np.allclose(a[1], [0.5, -3.0, 0.25]) and np.all(a[0] == 0)
    # the host writes the array: the next step puts gravity back for dynamic particles only
    junk = np.full_like(a, 7.0)
    pd.set_array(3, junk)
    ts.stepResident(m, 1); ts.syncToHost(m)
    a = pd.array(3)
    assert np.allclose(a[1], [0.5, -3.0, 0.25]) and np.all(a[0] == 7.0)
    # a particle becomes static: it keeps its value from now on, also across a change of gravity
    pd.setMass(5, 0.0)
    pbd.Simulation.getCurrent().setVecValueFloat(pbd.Simulation.GRAVITATION, [0.0, -1.0, 0.0])
    ts.stepResident(m, 1); ts.syncToHost(m)
    a = pd.array(3)
    assert np.allclose(a[5], [0.5, -3.0, 0.25]) and np.allclose(a[1], [0, -1.0, 0])
    # another model on the same time step object (possibly at the address of a freed one): filled from scratch
    del m, pd
    m2 = util.build_mine(util.cloth_spec(12, 12, 4, 3))
    ts.stepResident(m2, 1); ts.syncToHost(m2)
    a2 = m2.getParticles().array(3)
    assert np.allclose(a2[1], [0, -1.0, 0]) and np.all(a2[0] == 0)
    pbd.Simulation.getCurrent().setVecValueFloat(pbd.Simulation.GRAVITATION, [0, -9.81, 0])


def test_resident_equals_roundtrip_and_options_are_invariant(pbd):
    """Device-resident stepping == upload/download every step; graph vs eager, XCD remap and
    workgroup size must not change a single bit (they only reorder independent work)."""
    ops = util.cloth_spec(48, 48, 4, 3)
    base, _ = util.mine_run(ops, 6, 2, 5)
    xb = base.getParticles().positions()
    S = pbd.Solver
    for label, kw in (("resident", dict(resident=True)),
                      ("eager", dict(options={S.OPT_USE_GRAPH: 0})),
                      ("xcd remap", dict(options={S.OPT_XCD_REMAP: 1})),
                      ("block 64", dict(options={S.OPT_BLOCK_SIZE: 64})),
                      ("block 128 resident", dict(resident=True, options={S.OPT_BLOCK_SIZE: 128})),
                      ("per-colour schedule", dict(options={S.OPT_FUSE: 0})),
                      ("per-colour, no remap, eager", dict(options={S.OPT_FUSE: 0, S.OPT_XCD_REMAP: 0, S.OPT_USE_GRAPH: 0})),
                      ("fused, 100-particle tiles", dict(options={S.OPT_FUSE: 1, S.OPT_TILE_PARTICLES: 100})),
                      ("fused, 700-particle tiles, 256 threads", dict(options={S.OPT_FUSE: 1, S.OPT_TILE_PARTICLES: 700, S.OPT_FUSE_BLOCK: 256})),
                      ("fused, at most 3 colours per launch", dict(options={S.OPT_FUSE: 1, S.OPT_MAX_SEGMENT_COLOURS: 3})),
                      ("fused, one colour per launch, resident", dict(resident=True, options={S.OPT_FUSE: 1, S.OPT_MAX_SEGMENT_COLOURS: 1})),
                      ("fused, small LDS", dict(options={S.OPT_FUSE: 1, S.OPT_TILE_PARTICLES: 200, S.OPT_LDS_PARTICLES: 500})),
                      ("fused, 1024 threads, no remap", dict(options={S.OPT_FUSE: 1, S.OPT_FUSE_BLOCK: 1024, S.OPT_XCD_REMAP: 0})),
                      ("fused, 512 threads, small tiles", dict(options={S.OPT_FUSE: 1, S.OPT_FUSE_BLOCK: 512, S.OPT_TILE_PARTICLES: 3000}))):
        m, ts = util.mine_run(ops, 6, 2, 5, **kw)
        assert util.bitwise_equal(m.getParticles().positions(), xb), label
        assert ts.solver().plan_info()["active"] == (0 if "per-colour" in label else 1), label


def test_parameter_streams_converted_when_the_predicted_form_misses(pbd):
    """The parameter streams of a plan come in two forms (pbdx_plan.h param_float_index: planes for 1 024-thread workgroups, vector segments
    up to 512); the form is predicted before the plan exists and converted afterwards if the workgroup size turns out otherwise.  A cloth
    with more than 512 particles per CU but tiles forced small: predicted 1 024 threads, runs with <= 512 (and, as the control, the same cloth
    with the tiles the engine picks).  Both must reproduce the per-colour schedule bit for bit (positions, velocities, old / last positions)."""
    S = pbd.Solver
    for label, ops, opts in (("380x380 cloth, 150-particle tiles", util.cloth_spec(380, 380, 4, 3), {S.OPT_TILE_PARTICLES: 150}),
                             ("380x380 cloth, default tiles", util.cloth_spec(380, 380, 4, 3), {})):
        ma, _ = util.mine_run(ops, 2, 1, 3, options={S.OPT_FUSE: 0})
        mb, tsb = util.mine_run(ops, 2, 1, 3, options={S.OPT_FUSE: 1, **opts})
        sol = tsb.solver()
        info = sol.plan_info()
        assert info["active"] == 1
        blocks = sorted({sol.segment_info(g)["block"] for g in range(info["num_segments"])})
        for which in (0, 2, 4, 5):
            assert util.bitwise_equal(ma.getParticles().array(which), mb.getParticles().array(which)), (label, which)
        print("%s: %d tiles, workgroup sizes %s: fused == per-colour" % (label, info["num_tiles"], blocks))
        if opts:
            assert max(blocks) <= 512, "the small tiles were meant to give narrow steps"


@pytest.mark.parametrize("name", list(SCENES))
def test_fused_tiles_equal_per_colour_schedule(pbd, name):
    """Every constraint type through both device schedules: the colour-fused LDS tiles (forced to
    many small tiles so that halos, redundant execution and multi-segment plans are exercised) must
    reproduce the per-colour launches bit for bit -- positions, velocities and XPBD multipliers."""
    ops, sub, iters, _ = SCENES[name]
    S = pbd.Solver
    ma, tsa = util.mine_run(ops, 4, sub, iters, options={S.OPT_FUSE: 0})
    for tile in (48, 160, 0):
        mb, tsb = util.mine_run(ops, 4, sub, iters, options={S.OPT_FUSE: 1, S.OPT_TILE_PARTICLES: tile})
        info = tsb.solver().plan_info()
        assert info["active"] == 1 and info["num_tiles"] >= 1
        for which in (0, 2, 4, 5):
            assert util.bitwise_equal(ma.getParticles().array(which), mb.getParticles().array(which)), (name, tile, which)
    print("%-50s fused == per-colour (last plan: %d segments, %d tiles, redundancy %.2f)" % (
        name, info["num_segments"], info["num_tiles"], info["redundancy"]))


def test_fused_large_cloth_and_lambdas(pbd):
    """300x300 cloth (90 000 particles, 537 606 constraints): fused auto plan vs per-colour launches,
    bit-identical state and multipliers after 3 steps of 10 iterations."""
    ops = util.cloth_spec(300, 300, 4, 3)
    S = pbd.Solver
    ma, tsa = util.mine_run(ops, 3, 1, 10, resident=True, options={S.OPT_FUSE: 0})
    mb, tsb = util.mine_run(ops, 3, 1, 10, resident=True)
    info = tsb.solver().plan_info()
    print("300x300 plan:", info)
    assert info["active"] == 1 and tsa.solver().plan_info()["active"] == 0
    for which in (0, 2, 4, 5):
        assert util.bitwise_equal(ma.getParticles().array(which), mb.getParticles().array(which)), which
    groups = ma.getConstraintGroups()
    types = ma.constraintTypes()
    for batch in (0, 3, len(groups) - 1):
        # batches are added per (group, type); with one type per leading group batch k == group k
        n = sum(1 for c in groups[batch] if types[c] == types[groups[batch][0]])
        if batch < 8:
            la = tsa.solver().get_lambdas(batch, n)
            lb = tsb.solver().get_lambdas(batch, n)
            assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), batch


def test_ensemble_instances_equal_independent_runs(pbd):
    """C4 in small: K instances stacked at the same coordinates inside one model evolve exactly
    like K independent single-instance runs, and the colouring is K x the single colouring."""
    K = 5
    single = util.cloth_spec(30, 30, 4, 3)
    multi = util.cloth_spec(30, 30, 4, 3, instances=K)
    m1, _ = util.mine_run(single, 8, 1, 10)
    mk, _ = util.mine_run(multi, 8, 1, 10)
    x1 = m1.getParticles().positions()
    xk = mk.getParticles().positions().reshape(K, -1, 3)
    for k in range(K):
        assert util.bitwise_equal(xk[k], x1), "instance %d diverged from the single run" % k
    g1 = [len(g) for g in m1.getConstraintGroups()]
    gk = [len(g) for g in mk.getConstraintGroups()]
    assert gk == [K * n for n in g1]
    # and against the reference on the multi-instance model
    xr = util.oracle_positions(multi, 8, 1, 10, "f32")
    assert util.max_err(mk.getParticles().positions(), xr) <= 1e-5


def test_static_particles_and_zero_stiffness(pbd):
    ops = util.cloth_spec(20, 20, 4, 3, cloth_k=0.0, bending_k=0.0)
    xr = util.oracle_positions(ops, 3, 1, 4, "f32")
    m, _ = util.mine_run(ops, 3, 1, 4)
    xg = m.getParticles().positions()
    assert util.max_err(xg, xr) <= 1e-6
    x0 = m.getParticles().array(1)
    assert np.array_equal(xg[0], x0[0]) and np.array_equal(xg[19], x0[19])


def test_schedule_validator_and_lambdas(pbd):
    ops = util.cloth_spec(16, 16, 4, 3)
    m, ts = util.mine_run(ops, 1, 1, 3)
    sol = ts.solver()
    sol.validate_schedule()
    o = util.oracle_run(ops, 1, 1, 3, "f32")
    groups = m.getConstraintGroups()
    types = m.constraintTypes()
    # batch 0 = distance XPBD constraints of colour 0 in creation order
    ids = [c for c in groups[0] if types[c] == pbd.ConstraintType.DISTANCE_XPBD]
    lam = sol.get_lambdas(0, len(ids))
    lam_ref = np.array([o.constraint_lambda(int(c)) for c in ids])
    assert np.allclose(lam, lam_ref, rtol=1e-5, atol=1e-9)


def test_xpbd_multipliers_bitwise_vs_reference_on_a_320x320_cloth(pbd):
    """The XPBD multipliers (m_lambda, Constraints.cpp:1241,1448) of every constraint of the first distance colour, of a
    late distance colour and of the last bending colour after 3 steps x 10 iterations: bit-identical to the reference's."""
    ops = util.cloth_spec(320, 320, 4, 3)
    o = util.oracle_run(ops, 3, 1, 10, "f32", threads=8)
    m, ts = util.mine_run(ops, 3, 1, 10, resident=True)
    sol = ts.solver()
    assert sol.plan_info()["active"] == 1
    assert util.bitwise_equal(m.getParticles().positions(), o.positions().astype(np.float32))
    groups = m.getConstraintGroups()
    types = m.constraintTypes()
    # batches are added per (group, type) in group order: find the batch index of (group, first type of the group)
    batch_of = {}
    b = 0
    for g, members in enumerate(groups):
        for t in sorted(set(int(types[c]) for c in members)):
            batch_of[(g, t)] = b
            b += 1
    checked = 0
    for g in (0, 5, len(groups) - 1):
        t = int(types[groups[g][0]])
        ids = [int(c) for c in groups[g] if types[c] == t]
        lam = sol.get_lambdas(batch_of[(g, t)], len(ids))
        lam_ref = np.array([o.constraint_lambda(c) for c in ids], dtype=np.float32)
        assert np.any(lam_ref != 0)
        assert np.array_equal(lam.view(np.uint32), lam_ref.view(np.uint32)), "group %d: %d of %d multipliers differ" % (
            g, int(np.sum(lam.view(np.uint32) != lam_ref.view(np.uint32))), len(ids))
        checked += len(ids)
    print("multipliers compared bitwise with the reference: %d" % checked)


# ---------------------------------------------------------------------------
# BASELINE.json full sizes
# ---------------------------------------------------------------------------
def _reference_states(ops, iters, horizons, threads=16):
    """positions / velocities of the float reference after each horizon (cumulative steps)."""
    ref = util.get_oracle("f32")
    util.apply_ref(ref, ops)
    ref.set_num_threads(threads); ref.set_time_step_size(0.005); ref.set_gravity(util.GRAVITY); ref.set_params(1, iters, 0)
    out, done = {}, 0
    for h in horizons:
        ref.step(h - done)
        done = h
        out[h] = (ref.positions().astype(np.float32), ref.get_array(2).astype(np.float32))
    ref.reset_all()
    return out


def test_full_size_c2_million_particle_cloth_vs_reference(pbd):
    """configs[1] at full size: 1000x1000 cloth, 5 988 006 constraints, 27 colours, 10 iterations -- every one of the
    3 000 000 coordinates and 3 000 000 velocity components bit-identical to the reference's own float build run on the
    host cores, after 10 steps, on EXACTLY the schedule bench.py times: the persistent one-launch-per-substep form is
    forced and asserted (active, folded, no refusal, no timeout), then one launch per segment, then (2 steps) the
    per-colour schedule."""
    S = pbd.Solver
    ops = util.cloth_spec(1000, 1000, 4, 3)
    want = _reference_states(ops, 10, [2, 10])
    x0 = None
    for label, options, steps in (("persistent, folded", {S.OPT_PERSISTENT: 2}, 10), ("one launch per segment", {S.OPT_PERSISTENT: 0}, 10),
                                  ("default (measured choice)", {}, 10), ("per-colour", {S.OPT_FUSE: 0}, 2)):
        m, ts = util.mine_run(ops, steps, 1, 10, resident=True, options=options)
        sol = ts.solver()
        info, pinfo = sol.plan_info(), sol.persistent_info()
        print("C2 full size [%s]: %s | %s" % (label, info, pinfo))
        if label == "per-colour":
            assert info["active"] == 0
        else:
            assert info["active"] == 1 and info["max_local"] <= 10240
        if label == "persistent, folded":
            assert pinfo["eligible"] == 1 and pinfo["active"] == 1 and pinfo["refusals"] == 0 and pinfo["timeouts"] == 0
            assert pinfo["last_folded"] == 1, "30 passes per substep: integration and velocity update must run inside the launch"
        if label == "one launch per segment":
            assert pinfo["active"] == 0 and pinfo["last_folded"] == 0
        xg, vg = m.getParticles().positions(), m.getParticles().array(2)
        xr, vr = want[steps]
        assert util.bitwise_equal(xg, xr), "%s: max err %.3e" % (label, util.max_err(xg, xr))
        assert util.bitwise_equal(vg, vr), label
        x0 = m.getParticles().array(1)
        # pinned corners never move, state finite
        assert np.array_equal(xg[0], x0[0]) and np.array_equal(xg[999], x0[999]) and np.all(np.isfinite(xg))


def test_full_size_c2_hundred_steps_and_f64_envelope(pbd):
    """configs[1] at full size AND at length: the 1000x1000 sheet after 100 steps x 10 iterations on the
    bench's schedule -- all 6 000 000 position and velocity words bit-identical to the reference's float build (16 host threads,
    about a minute) -- and, after the first 10 steps, the stated fp32 tolerance AT THE BASELINE SIZE: the engine's error against
    the reference's f64 build is the float reference's own error (e_gpu <= e_f32ref, here with equality because the bits agree;
    the per-particle figure is printed)."""
    S = pbd.Solver
    ops = util.cloth_spec(1000, 1000, 4, 3)
    want = _reference_states(ops, 10, [10, 100])
    x64 = util.oracle_positions(ops, 10, 1, 10, "f64", threads=16)
    m, ts = util.mine_run(ops, 10, 1, 10, resident=True, options={S.OPT_PERSISTENT: 2})
    x10 = m.getParticles().positions().copy()
    pinfo = ts.solver().persistent_info()
    assert pinfo["active"] == 1 and pinfo["last_folded"] == 1 and pinfo["refusals"] == 0 and pinfo["timeouts"] == 0
    assert util.bitwise_equal(x10, want[10][0])
    e_ref, e_gpu = util.max_err(want[10][0], x64), util.max_err(x10, x64)
    print("1000x1000, 10 steps: per-particle |f32 reference - f64| = %.3e, |gpu - f64| = %.3e (bounding box diagonal 14)" % (e_ref, e_gpu))
    assert e_gpu <= e_ref + 1e-7
    ts.stepResident(m, 90)
    ts.syncToHost(m)
    pinfo = ts.solver().persistent_info()
    assert pinfo["active"] == 1 and pinfo["refusals"] == 0 and pinfo["timeouts"] == 0
    xg, vg = m.getParticles().positions(), m.getParticles().array(2)
    assert util.bitwise_equal(xg, want[100][0]), "100 steps: max err %.3e" % util.max_err(xg, want[100][0])
    assert util.bitwise_equal(vg, want[100][1])
    print("1000x1000, 100 steps x 10 iterations: 3 000 000 coordinates + 3 000 000 velocity components bit-identical; sheet has dropped to y = %.3f" % float(xg[:, 1].min()))


def test_full_size_c3_bar_reaches_the_inversion_branch_bitwise(pbd):
    """configs[2] at full size and at length: the 100 000-tet FEM bar after 120 steps x 10 iterations.  The bar sags under
    gravity until the tets at its pinned end are crushed below 20 % of their rest volume and the projection takes the
    inversion-handling branch (svdWithInversionHandling, MathFunctions.cpp:261-388; Constraints.cpp:1797) -- asserted on the
    final state by counting those tets -- and every coordinate is still bit-identical to the reference."""
    ops = util.bar_spec(101, 21, 11, 2)
    ref = util.get_oracle("f32")
    util.apply_ref(ref, ops)
    ref.set_num_threads(16); ref.set_time_step_size(0.005); ref.set_gravity(util.GRAVITY); ref.set_params(1, 10, 0)
    ref.step(120)
    xr, vr = ref.positions().astype(np.float32), ref.get_array(2).astype(np.float32)
    ref.reset_all()
    m, ts = util.mine_run(ops, 120, 1, 10, resident=True)
    xg, vg = m.getParticles().positions(), m.getParticles().array(2)
    # tets whose current volume is below 20 % of the rest volume (the reference's handleInversion test) in the final state
    tm = m.getTetModels()[0]
    tets = tm.getParticleMesh().getTets() + tm.getIndexOffset()
    crushed = None
    if tets is not None:
        x0 = m.getParticles().array(1).astype(np.float64)
        t = np.asarray(tets, dtype=np.int64).reshape(-1, 4)

        def vol(x):
            a, b, c, d = x[t[:, 0]], x[t[:, 1]], x[t[:, 2]], x[t[:, 3]]
            return np.einsum("ij,ij->i", d - a, np.cross(c - a, b - a)) / -6.0
        crushed = int(np.sum(vol(xg.astype(np.float64)) / vol(x0) < 0.2))
    print("C3 full size, 120 steps: %s tets below 20 %% of their rest volume at the end; plan %s" % (crushed, ts.solver().plan_info()))
    if crushed is not None:
        assert crushed > 0, "the run never reached the inversion-handling branch: lengthen it"
    assert util.bitwise_equal(xg, xr), "max err %.3e" % util.max_err(xg, xr)
    assert util.bitwise_equal(vg, vr)


def test_full_size_c2_odd_pass_count(pbd):
    """5 iterations x 3 segments = 15 passes per substep (odd): the persistent launch at the 1 M size with the other
    parity of the position double buffer, 4 steps, bit-identical to the reference."""
    S = pbd.Solver
    ops = util.cloth_spec(1000, 1000, 4, 3)
    want = _reference_states(ops, 5, [4])
    m, ts = util.mine_run(ops, 4, 1, 5, resident=True, options={S.OPT_PERSISTENT: 2})
    info, pinfo = ts.solver().plan_info(), ts.solver().persistent_info()
    print("C2 full size, 5 iterations:", info, pinfo)
    assert pinfo["active"] == 1 and pinfo["refusals"] == 0 and pinfo["timeouts"] == 0
    assert pinfo["last_folded"] == 1, "integration and velocity update run inside the launch for any pass count"
    print("passes per substep: %d" % (5 * info["num_segments"]))
    assert util.bitwise_equal(m.getParticles().positions(), want[4][0])
    assert util.bitwise_equal(m.getParticles().array(2), want[4][1])


@pytest.mark.parametrize("method,iters,sub", [(2, 10, 1), (6, 10, 1), (4, 10, 1)])
def test_full_size_c3_100k_tet_bar_vs_reference(pbd, method, iters, sub):
    """configs[2] at full size: 101x21x11 bar, 100 000 tets (FEM tet / XPBD distance+volume / strain tet)."""
    ops = util.bar_spec(101, 21, 11, method)
    xr = util.oracle_positions(ops, 3, sub, iters, "f32", threads=8).astype(np.float32)
    m, ts = util.mine_run(ops, 3, sub, iters, resident=True)
    xg = m.getParticles().positions()
    print("C3 full size method %d plan: %s" % (method, ts.solver().plan_info()))
    assert util.bitwise_equal(xg, xr), "max err %.3e" % util.max_err(xg, xr)


def test_c4_ensemble_block_of_instances_vs_reference(pbd):
    """configs[3], one GPU's FULL share: 64 independent 200x200 sheets in one model (2 560 000 particles, 15 206 784
    constraints) against the reference on the same model, 2 steps x 10 iterations, bit-identical."""
    import time
    ops = util.cloth_spec(200, 200, 4, 3, instances=64, instance_offset=(0.0, 0.0, 12.0), instanced=True)
    want = _reference_states(ops, 10, [2, 10], threads=32)       # the reference builds the 64 sheets one after the other
    t0 = time.perf_counter()
    m = util.build_mine(ops)
    m.initConstraintGroups()
    t_build = time.perf_counter() - t0
    # ... and at 10 steps (the walk -- two tiles per workgroup, alternating order -- was pinned at 2 steps only), on the schedule the bench times
    m10, ts10 = util.mine_run(ops, 10, 1, 10, resident=True, options={pbd.Solver.OPT_PERSISTENT: 2})
    p10 = ts10.solver().persistent_info()
    assert p10["active"] == 1 and p10["refusals"] == 0 and p10["timeouts"] == 0 and ts10.solver().plan_info()["num_tiles"] > p10["grid"], "more tiles than workgroups: the walk"
    assert util.bitwise_equal(m10.getParticles().positions(), want[10][0]) and util.bitwise_equal(m10.getParticles().array(2), want[10][1])
    m, ts = util.mine_run(ops, 2, 1, 10, resident=True)
    d = ts.solver().describe()
    print("C4 block: host build + colouring %.3f s; %s" % (t_build, d))
    print("C4 block plan:", ts.solver().plan_info(), ts.solver().persistent_info())
    assert m.numInstances() == 64 and "one instance planned, replicated" in d
    assert m.getParticles().size() == 64 * 40000
    assert util.bitwise_equal(m.getParticles().positions(), want[2][0])
    assert util.bitwise_equal(m.getParticles().array(2), want[2][1])
    # the same block built the long way round (64 rounds of builder calls, planned as a whole) gives the same bits
    from oracle.scene_ref import expand_instances
    m2, ts2 = util.mine_run(expand_instances(ops), 2, 1, 10, resident=True)
    assert "replicated" not in ts2.solver().describe()
    assert util.bitwise_equal(m2.getParticles().positions(), want[2][0])


def test_walk_of_a_1500x1500_sheet_vs_reference(pbd):
    """What bench.py's size sweep reports above 1 M particles: a 1500x1500 sheet (2 250 000 particles, 13.5 M constraints) needs
    2-3 tiles per workgroup, which the persistent schedule walks in alternating order with the tile at the turn kept in LDS.  2 steps x 10 iterations
    against the reference's float build (16 threads), every position and velocity bit for bit, on the forced and asserted one-launch schedule."""
    S = pbd.Solver
    ops = util.cloth_spec(1500, 1500, 4, 3)
    want = _reference_states(ops, 10, [2], threads=32)
    m, ts = util.mine_run(ops, 2, 1, 10, resident=True, options={S.OPT_PERSISTENT: 2})
    info, pinfo = ts.solver().plan_info(), ts.solver().persistent_info()
    print("1500x1500 sheet: %s | %s" % (info, pinfo))
    assert pinfo["active"] == 1 and pinfo["refusals"] == 0 and pinfo["timeouts"] == 0 and pinfo["last_folded"] == 1
    assert info["num_tiles"] >= 2 * pinfo["grid"], "at least two tiles per workgroup: the walk is what is tested"
    assert util.bitwise_equal(m.getParticles().positions(), want[2][0]), "max err %.3e" % util.max_err(m.getParticles().positions(), want[2][0])
    assert util.bitwise_equal(m.getParticles().array(2), want[2][1])


def test_32_instanced_100k_tet_bars_vs_reference(pbd):
    """bench.py's extra line `c3_x32_fem` (the form in which configs[2] fills the GPU): 32 instanced 101x21x11 FEM bars, 3.2 M tets, 2 steps x 10
    iterations against the reference on the same 32-bar model, bit for bit; the plan is one instance's, replicated."""
    from positionbaseddynamics_amd import scenes
    ops = scenes.bar_spec(101, 21, 11, 2, instances=32, instanced=True)
    want = _reference_states(ops, 10, [2], threads=32)
    m, ts = util.mine_run(ops, 2, 1, 10, resident=True)
    d = ts.solver().describe()
    print("32 bars: %s" % d)
    assert m.numInstances() == 32 and m.getParticles().size() == 32 * 23331 and "one instance planned, replicated" in d
    assert util.bitwise_equal(m.getParticles().positions(), want[2][0]), "max err %.3e" % util.max_err(m.getParticles().positions(), want[2][0])
    assert util.bitwise_equal(m.getParticles().array(2), want[2][1])


# ---------------------------------------------------------------------------
# device-resident ParticleData, dirty tracking, pinned host mirror
# ---------------------------------------------------------------------------
def test_resident_state_dirty_tracking_and_explicit_sync(pbd):
    """Resident stepping keeps the state on the device; a host write through the model API (or through
    the zero-copy getVertices view + markDirty) is picked up by the next resident step; the result
    equals the upload-every-step contract of TimeStep::step."""
    ops = util.cloth_spec(40, 40, 4, 3)

    def fresh():
        m = util.build_mine(ops)
        pbd.TimeManager.setCurrent(pbd.TimeManager())
        ts = pbd.TimeStepController()
        ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
        ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 5)
        return m, ts

    def poke(m, via_view):
        pd = m.getParticles()
        if via_view:
            v = pd.getVertices()
            v[100:110, 1] += np.float32(0.05)
            pd.markDirty()
        else:
            x = pd.positions()
            x[100:110, 1] += np.float32(0.05)
            pd.set_array(0, x)

    # ground truth: host-authoritative stepping
    m0, t0 = fresh()
    for _ in range(3):
        t0.step(m0)
    poke(m0, False)
    for _ in range(3):
        t0.step(m0)
    want = m0.getParticles().positions()

    for via_view in (False, True):
        m, ts = fresh()
        ts.stepResident(m, 3)
        ts.syncToHost(m)                     # host mirror := device state
        poke(m, via_view)                    # host becomes newer than the device
        ts.stepResident(m, 3)                # must re-upload by itself
        ts.syncToHost(m)
        assert util.bitwise_equal(m.getParticles().positions(), want), "dirty host state was not uploaded (via_view=%s)" % via_view

    # explicit syncFromHost + pinned host arrays give the same bits
    m, ts = fresh()
    ts.solver().set_option(pbd.Solver.OPT_PIN_HOST, 1)
    ts.stepResident(m, 3)
    ts.syncToHost(m)
    poke(m, True)
    ts.syncFromHost(m)
    ts.stepResident(m, 3)
    ts.syncToHost(m)
    assert util.bitwise_equal(m.getParticles().positions(), want)
    # and a resident run without any host write does NOT re-upload: device stays ahead of the stale host copy
    m, ts = fresh()
    ts.stepResident(m, 2)
    stale = m.getParticles().positions().copy()
    ts.stepResident(m, 2)
    assert np.array_equal(m.getParticles().positions(), stale)      # host untouched until syncToHost
    ts.syncToHost(m)
    assert not np.array_equal(m.getParticles().positions(), stale)


# ---------------------------------------------------------------------------
# edge cases
# ---------------------------------------------------------------------------
def test_particles_without_constraints_free_fall(pbd):
    """No constraints at all: the step is integrate + velocity update only (no schedule, no plan)."""
    ops = [("vertex", (0.1 * i, 1.0 + 0.01 * i, -0.2 * i)) for i in range(37)] + [("mass", 3, 0.0)]
    xr = util.oracle_positions(ops, 7, 3, 4, "f32")
    m, ts = util.mine_run(ops, 7, 3, 4)
    assert ts.solver().plan_info()["active"] == 0
    assert util.bitwise_equal(m.getParticles().positions(), xr.astype(np.float32))
    m2, _ = util.mine_run(ops, 7, 3, 4, resident=True, vel_method=1)
    xr2 = util.oracle_positions(ops, 7, 3, 4, "f32", vel_method=1)
    assert util.bitwise_equal(m2.getParticles().positions(), xr2.astype(np.float32))


def test_isolated_particles_next_to_constrained_ones(pbd):
    """Particles that no constraint touches are owned by a tile and must be carried through every
    fused segment untouched by the projections (they still integrate)."""
    ops = util.cloth_spec(20, 20, 4, 3) + [("vertex", (3.0 + i, 5.0, 1.0)) for i in range(9)]
    xr = util.oracle_positions(ops, 5, 1, 6, "f32")
    for tile in (0, 50):
        m, ts = util.mine_run(ops, 5, 1, 6, options={pbd.Solver.OPT_TILE_PARTICLES: tile})
        assert ts.solver().plan_info()["active"] == 1
        assert util.bitwise_equal(m.getParticles().positions(), xr.astype(np.float32))


def test_single_constraint_and_single_tile(pbd):
    ops = [("vertex", (0.0, 0.0, 0.0)), ("vertex", (1.0, 0.2, 0.0)), ("constraint", "distance_xpbd", [0, 1], 1000.0)]
    xr = util.oracle_positions(ops, 4, 2, 3, "f32")
    m, ts = util.mine_run(ops, 4, 2, 3)
    assert util.bitwise_equal(m.getParticles().positions(), xr.astype(np.float32))


def test_topology_change_between_steps_rebuilds_the_device_image(pbd):
    """Adding constraints after stepping (every add* clears m_groupsInitialized, SimulationModel.cpp:572)
    must invalidate the schedule and the fused plan."""
    base = util.cloth_spec(24, 24, 4, 0)
    ref = util.get_oracle("f32")
    util.apply_ref(ref, base)
    ref.set_time_step_size(0.005); ref.set_gravity(util.GRAVITY); ref.set_params(1, 5, 0); ref.set_num_threads(1)
    ref.step(3)
    ref.add_bending_constraints(0, 3, 100.0)
    ref.step(3)
    m = util.build_mine(base)
    pbd.TimeManager.setCurrent(pbd.TimeManager())
    ts = pbd.TimeStepController()
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
    ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 5)
    for _ in range(3):
        ts.step(m)
    m.addBendingConstraints(m.getTriangleModels()[0], 3, 100.0)
    for _ in range(3):
        ts.step(m)
    assert util.bitwise_equal(m.getParticles().positions(), ref.positions().astype(np.float32))


def test_auto_schedule_selection_is_measured_and_result_invariant(pbd):
    """PBDX_OPT_FUSE = 2 (default): streaming-bound types run fused without measuring; with compute-heavy
    types (FEM tets here) both schedules are timed once and the faster is kept -- never changing a bit."""
    S = pbd.Solver
    cloth = util.cloth_spec(40, 40, 4, 3)
    m, ts = util.mine_run(cloth, 2, 1, 5)
    # (only the two forms of the fused schedule are timed against each other: PBDX_OPT_PERSISTENT)
    assert ts.solver().plan_info()["active"] == 1 and "autotune(per-colour 0.000 ms" in ts.solver().describe()
    bar = util.bar_spec(30, 5, 5, 2)
    ma, tsa = util.mine_run(bar, 3, 1, 5)                                  # auto
    d = tsa.solver().describe()
    print(d)
    assert "autotune(per-colour" in d
    for forced in (0, 1):
        mb, tsb = util.mine_run(bar, 3, 1, 5, options={S.OPT_FUSE: forced})
        assert tsb.solver().plan_info()["active"] == forced
        assert util.bitwise_equal(ma.getParticles().positions(), mb.getParticles().positions())


def test_raw_solver_abi_like_a_reference_side_binding(pbd):
    """pbdx_solver_* driven directly -- particles, one add_batch per (group, type) taken from the REFERENCE's
    own constraint list and colour groups, step, download -- i.e. exactly what a binding inside the reference
    does, without the product's model / time step mirrors in between."""
    ops = util.cloth_spec(36, 28, 4, 3)
    ref = util.get_oracle("f32")
    util.apply_ref(ref, ops)
    ref.set_num_threads(1); ref.set_time_step_size(0.005); ref.set_gravity(util.GRAVITY); ref.set_params(2, 6, 0)
    x0 = ref.positions().astype(np.float32)
    mass, inv = ref.get_array(6).astype(np.float32), ref.get_array(7).astype(np.float32)
    types = ref.constraint_types()
    groups = ref.groups()
    sol = pbd.Solver()
    sol.set_particles(x0, mass, inv)
    sol.begin_schedule()
    T = pbd.ConstraintType
    for g, members in enumerate(groups):
        for t in sorted(set(types[members].tolist())):
            sel = [int(c) for c in members if types[c] == t]
            idx = np.concatenate([ref.constraint_bodies(c) for c in sel]).astype(np.uint32)
            par = np.concatenate([ref.constraint_params(c) for c in sel]).astype(np.float32)
            sol.add_batch(g, int(t), idx, par)
    sol.end_schedule()
    sol.validate_schedule()
    sol.step(0.005, 2, 6, 0, util.GRAVITY, num_steps=7)
    x, v, old, last = sol.get_particles(len(x0))
    ref.step(7)
    assert util.bitwise_equal(x, ref.positions().astype(np.float32))
    assert util.bitwise_equal(v, ref.get_array(2).astype(np.float32))
    assert util.bitwise_equal(old, ref.get_array(4).astype(np.float32)) and util.bitwise_equal(last, ref.get_array(5).astype(np.float32))
    info = sol.plan_info()
    assert info["active"] == 1 and info["num_colours"] == len(groups)
    # error paths of the raw ABI
    with pytest.raises(pbd.PbdxError):
        sol.add_batch(0, T.DISTANCE, np.array([0, 1], dtype=np.uint32), np.array([1.0, 1.0], dtype=np.float32))   # schedule closed
    sol.begin_schedule()
    with pytest.raises(pbd.PbdxError):
        sol.add_batch(0, T.DISTANCE, np.array([0, 10 ** 6], dtype=np.uint32), np.array([1.0, 1.0], dtype=np.float32))   # index out of range
    with pytest.raises(pbd.PbdxError):
        sol.add_batch(0, 99, np.array([0, 1], dtype=np.uint32), np.array([1.0, 1.0], dtype=np.float32))          # unknown type
    sol.end_schedule()


@pytest.mark.gpu
def test_persistent_schedule_is_bit_identical_and_recovers_from_a_refused_launch(pbd):
    """PBDX_OPT_PERSISTENT: all sweeps of a substep as ONE launch (tile-to-tile hand-off inside the launch) must not
    change a bit against one launch per segment -- cloth, an irregular mesh with every light type, a FEM bar, many
    tiles per workgroup -- and a launch that finds its workgroups not co-resident (self-test, value 3) must leave the
    state untouched so that the engine completes the very same step with the multi-launch schedule."""
    S = pbd.Solver
    scenes = [("xpbd cloth 90x90", util.cloth_spec(90, 90, 4, 3), 4, 2, 6, {}),
              ("pbd cloth, second-order velocities", util.cloth_spec(40, 40, 1, 2), 5, 1, 4, {"vel_method": 1}),
              ("kitchen sink", util.kitchen_sink_spec(), 4, 2, 4, {}),
              ("fem bar", util.bar_spec(30, 6, 6, 2), 4, 1, 5, {}),
              ("xpbd cloth 120x120, 60-particle tiles (more tiles than CUs)", util.cloth_spec(120, 120, 4, 3), 3, 1, 4, {"opts": {S.OPT_TILE_PARTICLES: 48}}),
              # three to four tiles per workgroup, walked forwards and backwards in turn (the tile at the turn stays in LDS), with an odd and an even
              # number of passes per substep
              ("xpbd cloth 120x120, 16-particle tiles (3-4 tiles per workgroup), 4 iterations", util.cloth_spec(120, 120, 4, 3), 3, 1, 4, {"opts": {S.OPT_TILE_PARTICLES: 16}}),
              ("xpbd cloth 120x120, 16-particle tiles (3-4 tiles per workgroup), 5 iterations", util.cloth_spec(120, 120, 4, 3), 2, 2, 5, {"opts": {S.OPT_TILE_PARTICLES: 16, S.OPT_MAX_SEGMENT_COLOURS: 9}}),
              # two tiles resident per CU: every tile still has a workgroup of its own, so owned particles stay in LDS and the passes between the
              # first and the last write back only their boundary particles (FusedTile::wb_begin) -- as in the default one-tile-per-CU scenes above
              ("xpbd cloth 120x120, 40-particle tiles, two resident per CU", util.cloth_spec(120, 120, 4, 3), 3, 1, 4, {"opts": {S.OPT_TILE_PARTICLES: 40, S.OPT_PERSISTENT_WGS_PER_CU: 2}})]
    # both parities of the number of passes per substep (iterations x segments) must be covered: with an odd number the
    # folded launch ends in the other position buffer and the two buffers change roles from substep to substep
    scenes += [("xpbd cloth 70x70, 5 iterations", util.cloth_spec(70, 70, 4, 3), 5, 3, 5, {}),
               ("xpbd cloth 70x70, 5 iterations, at most 5 colours per pass", util.cloth_spec(70, 70, 4, 3), 3, 2, 5, {"opts": {S.OPT_MAX_SEGMENT_COLOURS: 5}}),
               ("xpbd cloth 70x70, 5 iterations, at most 9 colours per pass", util.cloth_spec(70, 70, 4, 3), 3, 2, 5, {"opts": {S.OPT_MAX_SEGMENT_COLOURS: 9}}),
               ("xpbd cloth 70x70, 4 iterations, at most 5 colours per pass", util.cloth_spec(70, 70, 4, 3), 3, 2, 4, {"opts": {S.OPT_MAX_SEGMENT_COLOURS: 5}})]
    parities = set()
    for label, ops, steps, sub, iters, extra in scenes:
        opts = dict(extra.get("opts", {}))
        opts[S.OPT_FUSE] = 1
        kw = dict(vel_method=extra.get("vel_method", 0))
        ref, tsr = util.mine_run(ops, steps, sub, iters, options={**opts, S.OPT_PERSISTENT: 0}, **kw)
        assert tsr.solver().plan_info()["active"] == 1, label
        xr, vr = ref.getParticles().positions(), ref.getParticles().array(2)
        for mode, resident in ((2, False), (2, True), (3, False), (3, True)):
            m, ts = util.mine_run(ops, steps, sub, iters, resident=resident, options={**opts, S.OPT_PERSISTENT: mode}, **kw)
            info = ts.solver().persistent_info()
            # (a plan with more than 8 segments -- the kitchen sink -- is not eligible and runs one launch per segment)
            eligible_before = info["eligible"] == 1 or info["refusals"] >= 1
            if label != "kitchen sink":
                assert eligible_before, (label, info)
            if mode == 2:
                assert info["active"] == info["eligible"] and info["refusals"] == 0, (label, info)
                if info["active"]:
                    assert info["last_folded"] == 1, (label, info)
                    parities.add((iters * ts.solver().plan_info()["num_segments"]) % 2)
            else:
                assert info["active"] == 0 and info["refusals"] == (1 if eligible_before else 0), (label, info)
            assert util.bitwise_equal(m.getParticles().positions(), xr), (label, mode, resident)
            assert util.bitwise_equal(m.getParticles().array(2), vr), (label, mode, resident)
    assert parities == {0, 1}, "both parities of the pass count must have been exercised: %r" % (parities,)


@pytest.mark.gpu
def test_persistent_schedule_auto_mode_reports_its_measurement(pbd):
    """Default (1): the one-launch form is timed once against one launch per segment on scratch positions."""
    m, ts = util.mine_run(util.cloth_spec(64, 64, 4, 3), 2, 1, 5)
    info = ts.solver().persistent_info()
    assert info["eligible"] == 1 and info["autotune_fused_ms"] > 0 and info["autotune_persistent_ms"] > 0
    assert info["active"] == (1 if info["autotune_persistent_ms"] < 1.02 * info["autotune_fused_ms"] else 0)


def test_dictionary_form_of_the_bending_records(pbd):
    """Scenes that run 1 024-thread workgroups keep a tile's DISTINCT bending records in LDS and stream one uint16 per slot
    (pbdx_plan.h FusedStep::dict).  Same values, same arithmetic: bit-identical to the plan without the form (PBDX_NO_DICT), to the
    per-colour schedule, and the streams shrink; a run-time edit of streamed parameters (new rest lengths AND new bending matrices, so that
    the tables change) takes effect as in a model built with the edited values."""
    S = pbd.Solver
    ops = util.cloth_spec(200, 200, 4, 3)

    def run(nodict, edit, steps=2, fuse=1):
        if nodict:
            os.environ["PBDX_NO_DICT"] = "1"
        else:
            os.environ.pop("PBDX_NO_DICT", None)
        try:
            m = util.build_mine(ops)
            pbd.TimeManager.setCurrent(pbd.TimeManager())
            ts = pbd.TimeStepController()
            ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
            ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 4)
            ts.solver().set_option(S.OPT_FUSE, fuse)
            for _ in range(steps):
                ts.step(m)
            if edit:
                nc = m.numConstraints()
                if edit == "scalars":
                    # every constraint's stiffness (a scene-wide scalar of the compact layout): the tables stay valid, the plan is kept
                    for c in range(nc):
                        p = m.constraintParams(c)
                        p[1 if len(p) == 2 else 0] *= np.float32(0.5)
                        m.setConstraintParams(c, p)
                else:
                    for c in list(range(0, 3000, 11)) + list(range(nc - 3000, nc, 13)):       # distance constraints first, bending constraints last
                        p = m.constraintParams(c)
                        if len(p) == 2:
                            p[0] = p[0] * np.float32(1.01)       # rest length
                        else:
                            p[1:] = p[1:] * np.float32(0.5)      # Q
                        m.setConstraintParams(c, p)
                for _ in range(steps):
                    ts.step(m)
            info = ts.solver().plan_info() if fuse else None
            return m, info
        finally:
            os.environ.pop("PBDX_NO_DICT", None)

    for edit in (False, True, "scalars"):
        ma, ia = run(False, edit)
        mb, ib = run(True, edit)
        mc, _ = run(False, edit, fuse=0)
        assert ia["active"] == 1 and ib["active"] == 1
        assert ia["stream_bytes_per_sweep"] < 0.6 * ib["stream_bytes_per_sweep"], (ia, ib)
        for which in (0, 2, 4, 5):
            assert util.bitwise_equal(ma.getParticles().array(which), mb.getParticles().array(which)), (edit, which)
            assert util.bitwise_equal(ma.getParticles().array(which), mc.getParticles().array(which)), (edit, which)
        print("200x200 cloth, %s: dictionary form == streamed == per-colour; streams %.1f vs %.1f MB per sweep" % (
            {False: "as built", True: "rest geometry edited between the steps", "scalars": "all stiffnesses edited between the steps"}[edit], ia["stream_bytes_per_sweep"] / 1e6, ib["stream_bytes_per_sweep"] / 1e6))


def test_parameter_edit_between_resident_steps_keeps_the_device_state(pbd):
    """after stepResident the device is ahead of the host mirror; a constraint-parameter edit, a setMass or a
    write of ONE host array must not roll the simulation back to the stale mirror.  Ground truth: host-authoritative stepping."""
    ops = util.cloth_spec(36, 36, 4, 3)

    def fresh():
        m = util.build_mine(ops)
        pbd.TimeManager.setCurrent(pbd.TimeManager())
        ts = pbd.TimeStepController()
        ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
        ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 5)
        return m, ts

    def edit(m, what):
        if what == "params":
            for c in range(0, 200, 7):
                p = m.constraintParams(c)
                p[1] = p[1] * np.float32(0.25)          # stiffness of XPBD distance constraints
                m.setConstraintParams(c, p)
        elif what == "mass":
            m.getParticles().setMass(500, 0.0)
        else:
            v = np.zeros((m.getParticles().size(), 3), dtype=np.float32)
            v[300:320, 1] = np.float32(0.5)
            m.getParticles().set_array(2, v)            # velocities replaced, nothing else: positions / oldX / lastX stay the device's

    for what in ("params", "mass", "velocity"):
        m0, t0 = fresh()
        for _ in range(3):
            t0.step(m0)
        edit(m0, what)
        for _ in range(3):
            t0.step(m0)
        want_x, want_v = m0.getParticles().positions(), m0.getParticles().array(2)
        m, ts = fresh()
        ts.stepResident(m, 3)                 # no syncToHost: the host mirror still holds the initial state
        edit(m, what)
        ts.stepResident(m, 3)
        ts.syncToHost(m)
        assert util.bitwise_equal(m.getParticles().positions(), want_x), what
        assert util.bitwise_equal(m.getParticles().array(2), want_v), what


def test_persistent_schedule_recovers_from_a_timed_out_tile(pbd):
    """a tile that gives up waiting for a neighbour used to leave garbage in a device-resident state.  Self-test
    (PBDX_OPT_PERSISTENT = 4: tile 0 never publishes its first pass; 5 ms bound): the engine restores the state it saved at
    the start of the call, repeats the call with one launch per segment and reports it -- the result is bit-identical."""
    S = pbd.Solver
    ops = util.cloth_spec(90, 90, 4, 3)
    ref, _ = util.mine_run(ops, 4, 2, 6, options={S.OPT_FUSE: 1, S.OPT_PERSISTENT: 0})
    xr, vr = ref.getParticles().positions(), ref.getParticles().array(2)
    for resident in (False, True):
        m, ts = util.mine_run(ops, 4, 2, 6, resident=resident, options={S.OPT_FUSE: 1, S.OPT_PERSISTENT: 4, S.OPT_PERSISTENT_TIMEOUT_MS: 5})
        info = ts.solver().persistent_info()
        assert info["timeouts"] == 1 and info["active"] == 0 and info["eligible"] == 0, info
        assert util.bitwise_equal(m.getParticles().positions(), xr), resident
        assert util.bitwise_equal(m.getParticles().array(2), vr), resident
        for which in (4, 5):
            assert util.bitwise_equal(m.getParticles().array(which), ref.getParticles().array(which)), which


def test_more_contacts_per_particle_than_the_engine_keeps_is_an_error(pbd):
    """the reference has no per-particle contact limit; the engine keeps 8.  Exceeding it must fail the step
    loudly (PBDX_ERR_UNSUPPORTED from pbdx_solver_step), not skip the particle's contact response silently."""
    ops = util.cloth_spec(8, 8, 4, 3, pin=False)
    eye = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    floor = dict(shape="box", params=[50.0, 0.5, 50.0], com=[0, 0.49, 0], R=eye, v1=[0, 0, 0], v2=[0, 0.49, 0], restitution=0.6, friction=0.2)

    def run(num_colliders):
        m = util.build_mine(ops)
        pbd.TimeManager.setCurrent(pbd.TimeManager())
        ts = pbd.TimeStepController()
        ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
        ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 2)
        sol = ts.solver()
        ts.syncFromHost(m)                         # uploads the particles (collision ranges are checked against them)
        sol.set_collision_ranges([(0, m.getParticles().size(), 0.6, 0.1)])
        sol.set_contact_params(0.05, 100.0, 5)
        sol.set_colliders([floor] * num_colliders)
        ts.stepResident(m, 1)
        return sol.num_contacts()

    assert run(8) == 8 * 64                        # 8 simultaneous contacts per particle: fine
    with pytest.raises(pbd.PbdxError) as e:
        run(9)
    assert e.value.code == 4 and "simultaneous contacts" in str(e.value)


@pytest.mark.gpu
def test_bounds_checked_build_finds_no_out_of_range_access(pbd):
    """SURVEY 5 (sanitizers): the hot path under a sanitizer-grade debug build.  _lib/libpbdx_bounds.so is the product's sources compiled with
    -DPBDX_BOUNDS=1 (csrc/pbdx_bounds.h): every address of the fused / persistent sweep that no buffer descriptor checks in hardware -- particle ids from
    the gid streams, the positions they select, LDS slots of the fill, the gather / scatter and the dictionary tables, chunk / tile descriptor indices,
    the dependency lists -- is compared with the size of what it addresses, violations are recorded and the access suppressed.  A second process runs the
    known-answer tests, every scene, the three schedules' cross-checks, the examples and kitchen-sink scenes (the all-types kernels), the dictionary form
    and the BASELINE workloads at full size (1 M cloth on all schedules, the 1500x1500 walk, the configs[3] block, the 100 k-tet bars) under it; every
    test there must stay bit-identical AND leave the record empty (tests/conftest.py: _bounds_record_stays_empty).  (Until the hazard of
    profiles/HISTORY.md [9] was found -- tests/test_asm_hazards.py -- this build died intermittently on the persistent schedule.)"""
    import subprocess
    import sys
    lib = os.path.join(util.ROOT, "positionbaseddynamics_amd", "_lib", "libpbdx_bounds.so")
    if not os.path.exists(lib):
        pytest.skip("libpbdx_bounds.so not built")
    sel = ("known_answer_projection or fem_tet_inversion_branch or scene_parity_vs_float_reference or fused_tiles_equal_per_colour_schedule or "
           "persistent_schedule_is_bit_identical or full_size_c2_million_particle_cloth_vs_reference or full_size_c2_odd_pass_count or "
           "c4_ensemble_block or example_runs_and_matches_reference or dictionary_form or full_size_c3_100k or walk_of_a_1500")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(util.ROOT, "tests", "test_gpu_parity.py"), os.path.join(util.ROOT, "tests", "test_examples.py"),
                        "-m", "gpu", "-q", "-x", "-k", sel],
                       env=dict(os.environ, PBDX_LIB=lib), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = p.stdout[-2000:]
    assert p.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
    print("bounds-checked build: " + tail.strip().splitlines()[-1])


_FMA_ENVELOPE = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from tests import util
out = {}
for name, ops, steps, sub, iters in (("cloth 50x50 XPBD distance + bending, 100 steps", util.cloth_spec(50, 50, 4, 3), 100, 1, 10),
                                      ("cloth 50x50 PBD distance + bending (C1), 1 step", util.cloth_spec(50, 50, 1, 2), 1, 1, 5),
                                      ("bar 30x5x5 FEM tets, 10 steps", util.bar_spec(30, 5, 5, 2), 10, 1, 10),
                                      ("bar 30x5x5 XPBD distance + volume, 10 steps", util.bar_spec(30, 5, 5, 6), 10, 1, 10),
                                      ("bar 30x5x5 strain tets, 10 steps", util.bar_spec(30, 5, 5, 4), 10, 1, 10),
                                      ("cloth 40x40 FEM triangles + dihedral, 10 steps", util.cloth_spec(40, 40, 2, 1), 10, 1, 5)):
    x64 = util.oracle_positions(ops, steps, sub, iters, "f64")
    x32 = util.oracle_positions(ops, steps, sub, iters, "f32")
    m, ts = util.mine_run(ops, steps, sub, iters, resident=True)
    xg = m.getParticles().positions()
    out[name] = dict(e_ref=util.max_err(x32, x64), e_gpu=util.max_err(xg, x64), d_ref=util.max_err(xg, x32), bitwise=bool(util.bitwise_equal(xg, x32)),
                     lib=__import__("positionbaseddynamics_amd")._ffi.LIB_PATH)
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
def test_fma_build_stays_inside_the_fp32_envelope(pbd):
    """The OPT-IN library built with -ffp-contract=fast (csrc/Makefile: libpbdx_fma.so; 6-10 % faster, the colour steps are VALU-issue-bound)
    is NOT bit-identical to the contraction-free float reference -- and does not claim to be.  Its stated tolerance is the north star's:
    per-particle position error against the reference's f64 build within a small factor of the float reference's OWN error against f64
    (e_gpu <= 4 e_f32ref + 1e-5; fused multiply-adds round once instead of twice, so it is usually the smaller one), on six scenes covering
    the constraint families, up to 100 steps."""
    import json
    import subprocess
    import sys
    lib = os.path.join(util.ROOT, "positionbaseddynamics_amd", "_lib", "libpbdx_fma.so")
    if not os.path.exists(lib):
        pytest.skip("libpbdx_fma.so not built")
    p = subprocess.run([sys.executable, "-c", _FMA_ENVELOPE % {"root": util.ROOT}], env=dict(os.environ, PBDX_LIB=lib), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    differs = 0
    for name, r in res.items():
        print("fma build, %-48s |f32 ref - f64| = %.3e  |gpu - f64| = %.3e  |gpu - f32 ref| = %.3e%s" % (name, r["e_ref"], r["e_gpu"], r["d_ref"], "  (bit-identical)" if r["bitwise"] else ""))
        assert r["lib"].endswith("libpbdx_fma.so")
        assert r["e_gpu"] <= 4.0 * r["e_ref"] + 1e-5, name
        differs += 0 if r["bitwise"] else 1
    assert differs > 0, "the contracted build is bit-identical everywhere: is it really built with -ffp-contract=fast?"


@pytest.mark.gpu
def test_substep_events_time_every_substep_of_a_call(pbd):
    """PBDX_OPT_SUBSTEP_EVENTS (the bench's median device time): one interval per substep of the last call, all positive, adding up to the
    call's own event time; the option changes no result and is empty when switched off."""
    S = pbd.Solver
    ops = util.cloth_spec(120, 90, 4, 3)
    base, _ = util.mine_run(ops, 6, 2, 5, resident=True)
    m = util.build_mine(ops)
    ts = pbd.TimeStepController()
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 2)
    ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 5)
    sol = ts.solver()
    ts.stepResident(m, 2)
    assert sol.substep_times() == []
    sol.set_option(S.OPT_SUBSTEP_EVENTS, 1)
    ts.stepResident(m, 4)
    t = sol.substep_times()
    total = sol.stats()["total_ms"]
    print("substep events: %d intervals, median %.4f ms, sum %.4f ms, call %.4f ms" % (len(t), sorted(t)[len(t) // 2], sum(t), total))
    assert len(t) == 8 and all(x > 0.0 for x in t)
    assert sum(t) <= total * 1.001 + 1e-3 and sum(t) >= 0.8 * total
    sol.set_option(S.OPT_SUBSTEP_EVENTS, 0)
    ts.syncToHost(m)
    assert util.bitwise_equal(m.getParticles().positions(), base.getParticles().positions())
