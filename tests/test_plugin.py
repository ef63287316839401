"""The reference-side binding (positionbaseddynamics_amd/plugin/TimeStepControllerHIP): the
reference's own SimulationModel + Simulation (oracle/_ref build of the unmodified sources) with
OUR TimeStep plug-in installed exactly as the reference installs a custom time step
(Demos/PositionBasedElasticRodsDemo/PositionBasedElasticRodsDemo.cpp:51-54).

CPU part: the plug-in loads into the reference, and without a GPU every step() is REFUSED
(no silent CPU path).  GPU part: the reference scene stepped through the plug-in equals the
reference's CPU TimeStepController (bit-exact for a float host, fp32 envelope for a double host).
"""
import ctypes as C
import os

import numpy as np
import pytest

from tests import util

PLUGIN_DIR = os.path.join(util.ROOT, "positionbaseddynamics_amd", "plugin", "_build")


def _plugin(variant):
    from oracle import refdrv
    path = os.path.join(PLUGIN_DIR, "libpbd_timestep_hip_%s.so" % variant)
    if not (refdrv.available(variant) and os.path.exists(path)):
        pytest.skip("reference build / plug-in for %s not present (built by __graft_entry__.build() where /root/reference exists)" % variant)
    return refdrv, path


def _counters(path):
    lib = C.CDLL(path)
    out = {}
    for name in ("gpu_steps", "fallback_steps", "failed_steps", "param_refreshes", "schedule_builds", "uploads"):
        f = getattr(lib, "pbdx_timestep_hip_" + name)
        f.argtypes = [C.c_void_p]
        f.restype = C.c_uint
        out[name] = f
    for name, args in (("step_resident", [C.c_void_p, C.c_void_p, C.c_uint]), ("sync_to_host", [C.c_void_p, C.c_void_p]),
                       ("sync_from_host", [C.c_void_p, C.c_void_p])):
        f = getattr(lib, "pbdx_timestep_hip_" + name)
        f.argtypes = args
        f.restype = C.c_int
        out[name] = f
    lib.pbdx_timestep_hip_mark_host_dirty.argtypes = [C.c_void_p]
    lib.pbdx_timestep_hip_mark_host_dirty.restype = None
    out["mark_host_dirty"] = lib.pbdx_timestep_hip_mark_host_dirty
    return lib, out


def _setup(ref, ops, sub_steps, iters):
    util.apply_ref(ref, ops)
    ref.set_num_threads(1)
    ref.set_time_step_size(0.005)
    ref.set_gravity(util.GRAVITY)


def test_plugin_refuses_without_gpu(have_gpu):
    """No device -> step() must not run anywhere (certainly not on the CPU) and must say so."""
    if have_gpu:
        pytest.skip("GPU present")
    refdrv, path = _plugin("f32")
    ref = refdrv.Ref("f32")
    _setup(ref, util.cloth_spec(12, 12, 4, 3), 1, 5)
    assert ref.install_timestep_plugin(path) == 0
    ref.set_params(1, 5, 0)
    x0 = ref.positions().copy()
    ref.step(2)
    assert np.array_equal(ref.positions(), x0), "the plug-in moved particles without a GPU: a CPU path is hiding somewhere"
    lib, cnt = _counters(path)
    ts = C.c_void_p(ref.lib.refdrv_get_timestep())
    assert cnt["gpu_steps"](ts) == 0 and cnt["fallback_steps"](ts) == 0 and cnt["failed_steps"](ts) == 2
    ref.reset_all()   # drops the plug-in; a fresh reference TimeStepController is installed


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["cloth", "bar"])
def test_plugin_f32_host_bit_exact(scene):
    refdrv, path = _plugin("f32")
    ops, sub, iters = {"cloth": (util.cloth_spec(40, 30, 4, 3), 1, 10), "bar": (util.bar_spec(20, 5, 5, 2), 2, 5)}[scene]
    want = util.oracle_run(ops, 8, sub, iters, "f32")
    x_cpu, v_cpu = want.positions().copy(), want.get_array(2).copy()
    ref = refdrv.Ref("f32")
    _setup(ref, ops, sub, iters)
    assert ref.install_timestep_plugin(path) == 0
    ref.set_params(sub, iters, 0)
    ref.step(8)
    lib, cnt = _counters(path)
    ts = C.c_void_p(ref.lib.refdrv_get_timestep())
    assert cnt["gpu_steps"](ts) == 8 and cnt["fallback_steps"](ts) == 0 and cnt["failed_steps"](ts) == 0
    x_gpu, v_gpu = ref.positions().copy(), ref.get_array(2).copy()
    ref.reset_all()
    assert util.bitwise_equal(x_gpu, x_cpu), "max err %.3e" % util.max_err(x_gpu, x_cpu)
    assert util.bitwise_equal(v_gpu, v_cpu)


@pytest.mark.gpu
def test_plugin_f64_host_envelope():
    """Default reference build is double: the plug-in converts at the boundary; result within the
    fp32 envelope of the double CPU path (XPBD cloth 10 steps f32-vs-f64 ~3e-5)."""
    refdrv, path = _plugin("f64")
    ops = util.cloth_spec(40, 30, 4, 3)
    x_cpu = util.oracle_positions(ops, 10, 1, 10, "f64").copy()
    ref = refdrv.Ref("f64")
    _setup(ref, ops, 1, 10)
    assert ref.install_timestep_plugin(path) == 0
    ref.set_params(1, 10, 0)
    ref.step(10)
    x_gpu = ref.positions().copy()
    ref.reset_all()
    err = util.max_err(x_gpu, x_cpu)
    print("plug-in in a double host: max |dx| vs double CPU path after 10 steps = %.3e" % err)
    assert err <= 2e-4


@pytest.mark.gpu
def test_plugin_with_static_colliders_and_contacts():
    """ClothCollisionDemo restated: the reference owns the rigid bodies and the DistanceFieldCollisionDetection;
    the plug-in reads the colliders from it and runs detection + contact velocity solve on the GPU."""
    refdrv, path = _plugin("f32")
    ops = util.cloth_spec(30, 30, 4, 3, T=(-5, 4, -5), pin=False)

    def scene(ref):
        _setup(ref, ops, 1, 5)
        ref.add_static_collider("box", (0, -2.5, 0), (1, 0, 0, 0), (100, 1, 100), (100, 1, 100), 0.6, 0.2)
        ref.add_static_collider("torus", (0, 1.5, 0), (1, 0, 0, 0), (6, 2, 6), (2, 1), 0.6, 0.1)
        ref.enable_collisions(0.05, 0.6, 0.1)

    ref = refdrv.Ref("f32")
    scene(ref)
    ref.set_params(1, 5, 0)
    ref.step(200)
    assert len(ref.contacts()) > 0
    x_cpu, v_cpu = ref.positions().copy(), ref.get_array(2).copy()
    scene(ref)
    assert ref.install_timestep_plugin(path) == 0
    # the collision detection was attached to the previous time step object: attach it to the plug-in
    ref.lib.refdrv_attach_collision_detection()
    ref.set_params(1, 5, 0)
    ref.step(200)
    lib, cnt = _counters(path)
    ts = C.c_void_p(ref.lib.refdrv_get_timestep())
    assert cnt["gpu_steps"](ts) == 200 and cnt["failed_steps"](ts) == 0 and cnt["fallback_steps"](ts) == 0
    x_gpu, v_gpu = ref.positions().copy(), ref.get_array(2).copy()
    ref.reset_all()
    assert util.bitwise_equal(x_gpu, x_cpu), "max err %.3e" % util.max_err(x_gpu, x_cpu)
    assert util.bitwise_equal(v_gpu, v_cpu)


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["sphere", "floor", "two bodies", "sphere on 200x200"])
def test_plugin_with_a_dynamic_rigid_body_as_impulse_sink(what):
    """SURVEY 8f rank 2, the remainder: rigid bodies of FINITE mass in the particle contacts.  A 16 kg sphere dropped onto a cloth held at its
    four corners: ParticleRigidBodyContactConstraint::solveVelocityConstraint (Constraints.cpp:2148-2189) changes the sphere's velocity and
    angular velocity with every contact, and the next contact of the sphere reads them -- the device solves the list sequentially in the
    reference's contact order (pair order, then the cloth's point hierarchy left to right), the bodies' own time integration stays on the host
    (plug-in: integrateBodies).  Particles, body positions, rotations and velocities equal the CPU TimeStepController's bit for bit.
    floor: the cloth is free and lands on a STATIC box (whose mesh is not tested against the sphere: no contact between rigid bodies), the sphere
    lands on the cloth -- particles squeezed between the two have a contact with each, both in the sequential list, in the reference's order.
    sphere on 200x200: the first scene on a 40 000-particle sheet.
    two bodies: a tumbling 9 kg box (rotated, spinning: the full world inertia tensor in computeMatrixK and in the angular impulse) and a sphere,
    neither testing its mesh, on the held cloth."""
    refdrv, path = _plugin("f32")
    n = 200 if what.endswith("200x200") else 30          # (200x200: several hundred contacts per step in the sequential list)
    floor, big = what == "floor", what.endswith("200x200")
    calls = [1] * 20 + [20] * 5 if big else [40] * 8          # (steps per call; contacts are counted after every call)
    ops = util.cloth_spec(n, n, 4, 3, T=(-5, 4, -5), pin=not floor) + ([] if floor else [("mass", (n - 1) * n, 0.0), ("mass", n * n - 1, 0.0)])

    def scene(ref):
        _setup(ref, ops, 2, 5)
        bodies = []
        if floor:
            f = ref.add_dynamic_collider("box", (0, 2.0, 0), (1, 0, 0, 0), (100, 1, 100), (100, 1, 100), density=1.0, restitution=0.6, friction=0.2)
            ref.set_rigid_body_mass(f, 0.0)
        if what == "two bodies":
            q = np.array([0.9, 0.2, 0.3, 0.1])
            b = ref.add_dynamic_collider("box", (1.0, 5.5, 0.5), tuple(q / np.linalg.norm(q)), (2, 1, 3), (2, 1, 3), density=1.5, restitution=0.5, friction=0.4)
            ref.set_rigid_body_velocity(b, (0.5, 0, -0.3), (1.0, 2.0, -1.5))
            bodies.append(b)
            bodies.append(ref.add_dynamic_collider("sphere", (-2.0, 6.0, -1.5), (1, 0, 0, 0), (1.6, 1.6, 1.6), (0.8,), density=3.0, restitution=0.6, friction=0.3))
        else:
            bodies.append(ref.add_dynamic_collider("sphere", (0.3, 6.5 if floor else 5.2 if big else 5.5, -0.2), (1, 0, 0, 0), (2, 2, 2), (1.0,), density=2.0, restitution=0.6, friction=0.3))
            if big:
                ref.set_rigid_body_velocity(bodies[-1], (0.4, -12.0, 0.2), (0.5, 0, 1.0))      # (thrown at the sheet: a large soft sheet falls almost as fast as the sphere)
        ref.enable_collisions(0.05, 0.6, 0.1)
        return bodies

    ref = refdrv.Ref("f32")
    bodies = scene(ref)
    ref.set_params(2, 5, 0)
    seen = two = 0
    for k in calls:
        ref.step(k)
        c = ref.contacts()
        seen += len(c)
        if len(c):
            two += int((np.unique(c[:, 0].astype(np.int64), return_counts=True)[1] > 1).sum())
    assert seen > 0 and (two > 0 or not floor)            # (floor: some particle touched the floor and the sphere in the same step)
    x_cpu, v_cpu, body_cpu = ref.positions().copy(), ref.get_array(2).copy(), np.array([ref.rigid_body_state(b) for b in bodies])
    assert abs(body_cpu[:, 10:13]).max() > 1e-3 and abs(body_cpu[:, 7]).max() > 1e-3          # the contacts pushed the bodies sideways and set them spinning
    bodies = scene(ref)
    assert ref.install_timestep_plugin(path) == 0
    ref.lib.refdrv_attach_collision_detection()
    ref.set_params(2, 5, 0)
    for k in calls:
        ref.step(k)
    lib, cnt = _counters(path)
    ts = C.c_void_p(ref.lib.refdrv_get_timestep())
    assert cnt["gpu_steps"](ts) == sum(calls) and cnt["failed_steps"](ts) == 0 and cnt["fallback_steps"](ts) == 0
    x_gpu, v_gpu, body_gpu = ref.positions().copy(), ref.get_array(2).copy(), np.array([ref.rigid_body_state(b) for b in bodies])
    ref.reset_all()
    print("contacts seen at the call boundaries of the CPU run:", seen, "; bodies after %d steps" % sum(calls), " (position, rotation, velocity, angular velocity, mass): CPU", body_cpu, "GPU", body_gpu)
    assert np.array_equal(body_gpu, body_cpu), "body state differs by %.3e" % abs(body_gpu - body_cpu).max()
    assert util.bitwise_equal(x_gpu, x_cpu), "max err %.3e" % util.max_err(x_gpu, x_cpu)
    assert util.bitwise_equal(v_gpu, v_cpu)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["f32", "f64"])
def test_plugin_resident_mode_and_dirty_tracking(variant):
    """SURVEY 8f rank 1 on the reference side: TimeStepControllerHIP::stepResident keeps the state in HBM (one upload, no
    download), syncToHost brings ParticleData up to date, a host write in between is seen (sampled hash of the arrays) and
    merged -- the arrays the host did not write are taken from the device, not rolled back."""
    refdrv, path = _plugin(variant)
    ops = util.cloth_spec(40, 30, 4, 3)

    def run(resident):
        ref = refdrv.Ref(variant)
        _setup(ref, ops, 1, 10)
        assert ref.install_timestep_plugin(path) == 0
        ref.set_params(1, 10, 0)
        lib, cnt = _counters(path)
        ts, model = ref.timestep_ptr(), ref.model_ptr()
        if resident:
            assert cnt["step_resident"](ts, model, 4) == 0
            assert cnt["uploads"](ts) == 1 and cnt["gpu_steps"](ts) == 4
            x_before = ref.positions().copy()
            assert np.array_equal(x_before, ref.get_array(1))             # host mirror untouched: still the initial state
            # host edit of the velocities only, while the device is ahead
            v = np.zeros_like(ref.get_array(2)); v[100:120, 1] = 0.25
            ref.set_array(2, v)
            assert cnt["step_resident"](ts, model, 3) == 0
            assert cnt["uploads"](ts) == 2                                 # the edit was noticed
            assert cnt["step_resident"](ts, model, 2) == 0
            assert cnt["uploads"](ts) == 2                                 # ... and nothing else was re-uploaded
            assert cnt["sync_to_host"](ts, model) == 0
        else:
            ref.step(4)
            v = np.zeros_like(ref.get_array(2)); v[100:120, 1] = 0.25
            ref.set_array(2, v)
            ref.step(5)
        assert cnt["failed_steps"](ts) == 0 and cnt["fallback_steps"](ts) == 0 and cnt["schedule_builds"](ts) == 1
        out = [ref.get_array(k).copy() for k in (0, 2, 4, 5)]
        ref.reset_all()
        return out

    a, b = run(False), run(True)
    for k in range(4):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.gpu
def test_plugin_picks_up_runtime_parameter_edits_without_replanning():
    """SimulationModel::setClothStiffness / setClothBendingStiffness between steps (what the demos' GUI does)
    change neither the constraint count nor m_groupsInitialized.  The plug-in notices (parameter hash), refreshes only the
    parameter streams -- no schedule rebuild -- and stays bit-identical to the CPU reference making the same edit."""
    refdrv, path = _plugin("f32")
    ops = util.cloth_spec(40, 30, 4, 3)

    def run(gpu):
        ref = refdrv.Ref("f32")
        _setup(ref, ops, 1, 8)
        if gpu:
            assert ref.install_timestep_plugin(path) == 0
        ref.set_params(1, 8, 0)
        ref.step(3)
        ref.set_cloth_stiffness(2500.0)
        ref.step(3)
        ref.set_cloth_bending_stiffness(7.0)
        ref.set_constraint_stiffness(11, 123.0)       # one constraint differs from the rest: the shared stiffness is no longer uniform
        ref.step(3)
        x, v = ref.positions().copy(), ref.get_array(2).copy()
        if gpu:
            lib, cnt = _counters(path)
            ts = ref.timestep_ptr()
            assert cnt["gpu_steps"](ts) == 9 and cnt["failed_steps"](ts) == 0
            assert cnt["schedule_builds"](ts) == 1 and cnt["param_refreshes"](ts) == 2
        ref.reset_all()
        return x, v

    xc, vc = run(False)
    xg, vg = run(True)
    assert util.bitwise_equal(xg, xc), "max err %.3e" % util.max_err(xg, xc)
    assert util.bitwise_equal(vg, vc)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["f32", "f64"])
def test_plugin_runs_a_mixed_model_group_by_group(variant):
    """a model with constraint classes the engine does not know -- the 50x50 ClothDemo sheet (XPBD distance + XPBD
    isometric bending) plus two of the reference's GenericDistanceConstraints (Demos/GenericConstraintsDemos: a stitch between two
    distant particles and a doubled edge) and one GenericIsometricBendingConstraint -- is NOT refused: the known (group, type) buckets run
    on the GPU, the generic ones through the reference's own solvePositionConstraint on the host inside the same colour groups.  Float
    host: bit-identical to the CPU TimeStepController over 60 steps x 2 substeps; double host: inside the usual fp32 envelope."""
    refdrv, path = _plugin(variant)
    ops = util.cloth_spec(50, 50, 4, 3)

    def run(gpu):
        ref = refdrv.Ref(variant)
        _setup(ref, ops, 2, 5)
        ref.add_generic_distance_constraint(60, 1890, 0.5)                   # a stitch across the sheet
        ref.add_generic_distance_constraint(777, 778, 1.0)                   # on top of an existing edge
        ref.add_generic_isometric_bending_constraint(1200, 1251, 1201, 1250, 0.3)
        if gpu:
            assert ref.install_timestep_plugin(path) == 0
        ref.set_params(2, 5, 0)
        ref.step(60)
        out = (ref.positions().copy(), ref.get_array(2).copy())
        if gpu:
            lib, cnt = _counters(path)
            ts = ref.timestep_ptr()
            lib.pbdx_timestep_hip_mixed_groups.argtypes = [C.c_void_p]; lib.pbdx_timestep_hip_mixed_groups.restype = C.c_uint
            mixed = lib.pbdx_timestep_hip_mixed_groups(ts)
            print("mixed model (%s host): %d colour groups hold host constraints; gpu steps %d" % (variant, mixed, cnt["gpu_steps"](ts)))
            assert cnt["gpu_steps"](ts) == 60 and cnt["failed_steps"](ts) == 0 and cnt["fallback_steps"](ts) == 0
            assert 1 <= mixed <= 3
        ref.reset_all()
        return out

    (xc, vc), (xg, vg) = run(False), run(True)
    plain = util.oracle_positions(ops, 60, 2, 5, variant)
    assert util.max_err(xc, plain) > 1e-3, "the generic constraints change nothing: the test would prove nothing"
    if variant == "f32":
        assert util.bitwise_equal(xg, xc), "max err %.3e" % util.max_err(xg, xc)
        assert util.bitwise_equal(vg, vc)
    else:
        print("mixed model, double host: max |dx| vs the double CPU path = %.3e" % util.max_err(xg, xc))
        assert util.max_err(xg, xc) <= 2e-3


@pytest.mark.gpu
def test_plugin_mixed_model_calls_the_per_substep_hook_of_user_subclasses():
    """the reference calls every constraint's initConstraintBeforeProjection once per substep, after the integration
    (TimeStepController.cpp:264-268).  A user subclass that overrides it (oracle/ref_driver.cpp: HookedDistanceConstraint, whose rest length follows
    the INTEGRATED velocity and x - oldX of a particle) lives on the host side of a mixed model: the plug-in must call the hook there, as often as the
    CPU controller does, and on the same state -- bit-identical positions and velocities after 40 steps x 2 substeps are the proof."""
    refdrv, path = _plugin("f32")
    ops = util.cloth_spec(50, 50, 4, 3)

    def run(gpu):
        ref = refdrv.Ref("f32")
        _setup(ref, ops, 2, 5)
        ref.add_hooked_distance_constraint(60, 1890, 0.5)
        ref.add_hooked_distance_constraint(1300, 1301, 1.0)
        if gpu:
            assert ref.install_timestep_plugin(path) == 0
        ref.set_params(2, 5, 0)
        ref.hooked_calls(reset=True)
        ref.step(40)
        calls = ref.hooked_calls(reset=True)
        out = (ref.positions().copy(), ref.get_array(2).copy(), calls)
        if gpu:
            lib, cnt = _counters(path)
            ts = ref.timestep_ptr()
            assert cnt["gpu_steps"](ts) == 40 and cnt["failed_steps"](ts) == 0 and cnt["fallback_steps"](ts) == 0
        ref.reset_all()
        return out

    (xc, vc, nc), (xg, vg, ng) = run(False), run(True)
    assert nc == 2 * 40 * 2 and ng == nc, "hook calls: CPU controller %d, plug-in %d" % (nc, ng)
    plain = util.oracle_positions(ops, 40, 2, 5, "f32")
    assert util.max_err(xc, plain) > 1e-3, "the hooked constraints change nothing: the test would prove nothing"
    assert util.bitwise_equal(xg, xc), "max err %.3e" % util.max_err(xg, xc)
    assert util.bitwise_equal(vg, vc)


@pytest.mark.gpu
def test_plugin_full_size_c2_reference_model():
    """The 1000x1000 REFERENCE model (the reference's own SimulationModel, 5 988 006 heap constraints) stepped through the
    plug-in: bit-identical to the CPU path after 2 steps; plug-in cost per step printed for the round trip
    (TimeStep::step contract) and for resident stepping."""
    import time
    refdrv, path = _plugin("f32")
    ops = util.cloth_spec(1000, 1000, 4, 3)
    ref = refdrv.Ref("f32")
    _setup(ref, ops, 1, 10)
    ref.set_num_threads(32)
    ref.set_params(1, 10, 0)
    ref.step(2)
    x_cpu, v_cpu = ref.positions().copy(), ref.get_array(2).copy()
    _setup(ref, ops, 1, 10)
    assert ref.install_timestep_plugin(path) == 0
    ref.set_params(1, 10, 0)
    t0 = time.perf_counter()
    ref.step(1)
    t_first = time.perf_counter() - t0
    ref.step(1)
    lib, cnt = _counters(path)
    ts, model = ref.timestep_ptr(), ref.model_ptr()
    assert cnt["gpu_steps"](ts) == 2 and cnt["failed_steps"](ts) == 0
    assert util.bitwise_equal(ref.positions(), x_cpu) and util.bitwise_equal(ref.get_array(2), v_cpu)
    lib.pbdx_timestep_hip_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    lap = (C.c_double * 7)()
    lib.pbdx_timestep_hip_timing(ts, lap, 1)
    t0 = time.perf_counter()
    ref.step(10)
    t_round_exact = (time.perf_counter() - t0) / 10
    lib.pbdx_timestep_hip_timing(ts, lap, 1)
    print("plug-in round trip at 1000x1000 (default: exact parameter scan of all 5 988 006 records), ms per step: host-array hashes %.3f, uploads %.3f, parameter check %.3f, colliders %.3f, engine step %.3f (device events %.3f), download %.3f" % (
        lap[0] / 10, lap[1] / 10, lap[2] / 10, lap[3] / 10, lap[4] / 10, lap[6] / 10, lap[5] / 10))
    # opt-in sampled parameter check (what round 3 did by default)
    _extra(path).pbdx_timestep_hip_set_full_parameter_scan(ts, 0)
    ref.step(1)
    refreshes_after_switch = cnt["param_refreshes"](ts)
    lib.pbdx_timestep_hip_timing(ts, lap, 1)
    t0 = time.perf_counter()
    ref.step(10)
    t_round = (time.perf_counter() - t0) / 10
    lib.pbdx_timestep_hip_timing(ts, lap, 1)
    print("plug-in round trip at 1000x1000 (opt-in: sampled parameter check), ms per step: host-array hashes %.3f, uploads %.3f, parameter check %.3f, colliders %.3f, engine step %.3f (device events %.3f), download %.3f" % (
        lap[0] / 10, lap[1] / 10, lap[2] / 10, lap[3] / 10, lap[4] / 10, lap[6] / 10, lap[5] / 10))
    # and the exact scan once more (the first leg above also carries whatever the process did before: thread pools, first touches)
    _extra(path).pbdx_timestep_hip_set_full_parameter_scan(ts, 1)
    ref.step(1)
    refreshes_after_switch = cnt["param_refreshes"](ts)
    lib.pbdx_timestep_hip_timing(ts, lap, 1)
    t0 = time.perf_counter()
    ref.step(10)
    t_round_exact2 = (time.perf_counter() - t0) / 10
    lib.pbdx_timestep_hip_timing(ts, lap, 1)
    print("plug-in round trip at 1000x1000 (exact parameter scan, second leg), ms per step: host-array hashes %.3f, uploads %.3f, parameter check %.3f, colliders %.3f, engine step %.3f (device events %.3f), download %.3f" % (
        lap[0] / 10, lap[1] / 10, lap[2] / 10, lap[3] / 10, lap[4] / 10, lap[6] / 10, lap[5] / 10))
    t_round_exact = min(t_round_exact, t_round_exact2)
    print("round trip per step: exact scan %.3f ms, sampled check %.3f ms" % (1e3 * t_round_exact, 1e3 * t_round))
    assert cnt["step_resident"](ts, model, 5) == 0
    t0 = time.perf_counter()
    assert cnt["step_resident"](ts, model, 50) == 0
    t_res = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    assert cnt["sync_to_host"](ts, model) == 0
    t_sync = time.perf_counter() - t0
    assert cnt["schedule_builds"](ts) == 1 and cnt["param_refreshes"](ts) == refreshes_after_switch
    print("plug-in at 1000x1000: first step (schedule build + plan + autotune) %.2f s; round trip %.3f ms/step; resident %.3f ms/step; syncToHost %.2f ms" % (
        t_first, 1e3 * t_round, 1e3 * t_res, 1e3 * t_sync))
    try:      # (diagnostic only: a refused or timed-out one-launch schedule would explain a slow resident figure)
        import positionbaseddynamics_amd as pbd
        lib.pbdx_timestep_hip_solver.argtypes = [C.c_void_p]; lib.pbdx_timestep_hip_solver.restype = C.c_void_p
        pi = pbd.Solver(handle=lib.pbdx_timestep_hip_solver(ts)).persistent_info()
        print("  engine of the plug-in: one-launch schedule active %s, refusals %s, time-outs %s" % (pi.get("active"), pi.get("refusals"), pi.get("timeouts")))
    except Exception as e:
        print("  (no schedule information: %r)" % (e,))
    ref.reset_all()
    assert t_res < 2.2e-3 and t_round < 8e-3 and t_round_exact < 30e-3      # (resident: well below a round trip; 0.66 measured, 1.2 once on a throttled host)


# ---------------------------------------------------------------------------
# the compiled python side: the reference's OWN pypbd (pyPBD/*.cpp compiled unmodified, plugin/Makefile) with the one
# added class TimeStepControllerHIP (plugin/TimeStepHIPModule.cpp)
# ---------------------------------------------------------------------------
_PYPBD_RUN = r"""
import sys, json, importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location("ex", %(example)r)
ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
pbd = ex.pbd
out = {"classes": sorted(n for n in dir(pbd) if not n.startswith("_"))}
for cfg in %(configs)r:
    xs = {}
    key = "".join(str(c) for c in cfg)
    for gpu in %(modes)r:
        # (Simulation.getCurrent() creates AND initialises the singleton, Simulation.cpp:30-38; every leg installs its own time step)
        pbd.TimeManager.getCurrent().setTime(0.0)
        ex.buildModel(*cfg, gpu=gpu)
        model = pbd.Simulation.getCurrent().getModel()
        x0 = np.array(model.getParticles().getVertices(), copy=True)
        for frame in range(2):
            ex.timeStep()
            if %(edit)r and frame == 0:
                # pyPBD/ConstraintsModule.cpp:62-116: python writes ONE constraint's fields in place -- nothing is announced
                cs = model.getConstraints()
                c = cs[len(cs) // 3]
                for field in ("stiffness", "xxStiffness", "stretchStiffness"):
                    if hasattr(c, field):
                        setattr(c, field, 0.25 * getattr(c, field))
        x = np.array(model.getParticles().getVertices(), copy=True)
        ts = pbd.Simulation.getCurrent().getTimeStep()
        xs[gpu] = x
        out[key + ("_gpu" if gpu else "_cpu")] = dict(
            type=type(ts).__name__, moved=bool(np.abs(x - x0).max() > 0), finite=bool(np.isfinite(x).all()),
            gpu_steps=ts.numGpuSteps() if gpu else None, failed=ts.numFailedSteps() if gpu else None,
            refreshes=ts.numParameterRefreshes() if gpu else None, builds=ts.numScheduleBuilds() if gpu else None,
            is_controller=isinstance(ts, pbd.TimeStepController), constraints=len(model.getConstraints()))
    if True in xs and False in xs:
        a, b = xs[True].astype(np.float32), xs[False].astype(np.float32)
        out[key + "_bitwise"] = bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
        out[key + "_maxabs"] = float(np.abs(a - b).max())
print("RESULT " + json.dumps(out))
"""


def _run_pypbd(modes, example="cloth_model_pypbd.py", configs=((2, 2), (4, 3)), edit=False):
    import glob
    import json
    import subprocess
    import sys
    if not glob.glob(os.path.join(PLUGIN_DIR, "pypbd*.so")):
        pytest.skip("pypbd module not built (needs /root/reference at build time)")
    code = _PYPBD_RUN % {"example": os.path.join(util.ROOT, "examples", example), "modes": modes, "configs": [list(c) for c in configs], "edit": edit}
    # own process: pypbd carries its own copy of the reference's singletons (Simulation::current ...), like the reference's build
    # (OMP_NUM_THREADS=1: the reference forks / joins one parallel region per colour group; on a 256-thread host that costs more
    # than the 2 500-particle sheet it distributes)
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_compiled_pypbd_is_the_references_own_module_plus_one_class(have_gpu):
    """The module is the reference's pyPBD/*.cpp (every class its ten *Module.cpp files register is there: constraints with
    their fields, tet models, rigid bodies, joints, collision detection, loaders) + `TimeStepControllerHIP`, which is a
    TimeStepController, is accepted by Simulation.setTimeStep, inherits the parameter ids (setValueUInt(NUM_SUB_STEPS, 3) in the
    example) -- and without a GPU refuses every step instead of computing anything on the host."""
    if have_gpu:
        pytest.skip("GPU present: covered by the gpu tests")
    res = _run_pypbd([True])
    for name in ("TimeStepControllerHIP", "TimeStepController", "TetModel", "DistanceConstraint_XPBD", "IsometricBendingConstraint_XPBD", "FEMTetConstraint",
                 "RigidBody", "BallJoint", "CubicSDFCollisionDetection", "DistanceFieldCollisionDetection", "OBJLoader", "TetGenLoader", "VecConstraints", "Timing"):
        assert name in res["classes"], name
    for key in ("22_gpu", "43_gpu"):
        r = res[key]
        assert r["type"] == "TimeStepControllerHIP" and r["is_controller"]
        assert r["gpu_steps"] == 0 and r["failed"] == 16 and not r["moved"] and r["finite"]
    res = _run_pypbd([True], example="beam_model_pypbd.py", configs=((2,),))
    assert res["2_gpu"]["failed"] == 16 and not res["2_gpu"]["moved"]


def test_discregrid_shim_fails_loudly_instead_of_inventing_distance_fields():
    """oracle/shim/Discregrid/All is COMPILE-ONLY: the reference's cubic-SDF entry points exist in the module (unmodified
    pyPBD/CollisionDetectionModule.cpp:155-186) and raise when called -- no stand-in distance field produces numbers nobody can check."""
    import glob
    import subprocess
    import sys
    if not glob.glob(os.path.join(PLUGIN_DIR, "pypbd*.so")):
        pytest.skip("pypbd module not built (needs /root/reference at build time)")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, pypbd as pbd\n"
            "sim = pbd.Simulation.getCurrent(); sim.initDefault()\n"
            "cd = sim.getTimeStep().getCollisionDetection()\n"
            "print('CD', type(cd).__name__)\n"
            "vd = pbd.VertexData(); mesh = pbd.IndexedFaceMesh()\n"
            "for p in ((0,0,0),(1,0,0),(0,1,0),(0,0,1)): vd.addVertex(np.array(p, dtype=np.float32))\n"
            "mesh.initMesh(4, 6, 4)\n"
            "for f in ((0,2,1),(0,1,3),(0,3,2),(1,2,3)): mesh.addFace(list(f))\n"
            "mesh.buildNeighbors()\n"
            "try:\n    pbd.CubicSDFCollisionDetection.generateSDF(vd, mesh, np.array([4,4,4], dtype=np.uint32))\n    print('NO ERROR')\n"
            "except RuntimeError as e:\n    print('RAISED', e)\n") % PLUGIN_DIR
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "CD CubicSDFCollisionDetection" in p.stdout, p.stdout
    assert "RAISED" in p.stdout and "Discregrid is not part of this build" in p.stdout, p.stdout


@pytest.mark.gpu
def test_reference_python_example_through_compiled_pypbd_on_the_gpu():
    """pyPBD/examples/cloth_model.py's logic, `import pypbd as pbd` = the reference's own module, the marked lines added to install
    the GPU time step: 16 steps x 3 substeps for the example's default (FEM triangles + isometric bending) and for the XPBD variant,
    bit-identical to the same script with the reference's own TimeStepController (same float host)."""
    res = _run_pypbd([True, False])
    for key in ("22", "43"):
        g, c = res[key + "_gpu"], res[key + "_cpu"]
        assert g["type"] == "TimeStepControllerHIP" and c["type"] == "TimeStepController"
        assert g["gpu_steps"] == 16 and g["failed"] == 0 and g["moved"] and c["moved"]
        print("compiled pypbd, cloth model %s: bitwise %s, max abs %.3e" % (key, res[key + "_bitwise"], res[key + "_maxabs"]))
        assert res[key + "_bitwise"], res[key + "_maxabs"]


@pytest.mark.gpu
def test_reference_python_beam_example_through_compiled_pypbd_on_the_gpu():
    """pyPBD/examples/beam_model.py's logic (30x5x5 tet beam through TetModel / addSolidConstraints of the reference's own bindings,
    pyPBD/SimulationModelModule.cpp:206-299): FEM tets, XPBD FEM tets (the example's default), strain tets, shape matching, XPBD
    distance + volume -- 16 steps x 3 substeps bit-identical to pypbd's own TimeStepController."""
    res = _run_pypbd([True, False], example="beam_model_pypbd.py", configs=((2,), (3,), (4,), (5,), (6,)))
    for key in ("2", "3", "4", "5", "6"):
        g, c = res[key + "_gpu"], res[key + "_cpu"]
        assert g["type"] == "TimeStepControllerHIP" and c["type"] == "TimeStepController"
        assert g["gpu_steps"] == 16 and g["failed"] == 0 and g["moved"] and c["moved"]
        print("compiled pypbd, beam model %s: bitwise %s, max abs %.3e" % (key, res[key + "_bitwise"], res[key + "_maxabs"]))
        assert res[key + "_bitwise"], res[key + "_maxabs"]


@pytest.mark.gpu
def test_python_edit_of_one_constraint_field_is_seen_without_announcement():
    """pyPBD/ConstraintsModule.cpp:62-116 makes every constraint's fields python-writable.  ONE constraint of the 50x50 sheet
    (14 406 constraints) and of the beam gets a quarter of its stiffness between two frames, nothing is announced: the plug-in's
    exact parameter scan finds it (one parameter refresh, no new schedule) and the run stays bit-identical to the CPU time step."""
    res = _run_pypbd([True, False], edit=True)
    for key in ("22", "43"):
        g = res[key + "_gpu"]
        print("edit, cloth %s: constraints %d, refreshes %d, builds %d, bitwise %s" % (key, g["constraints"], g["refreshes"], g["builds"], res[key + "_bitwise"]))
        assert g["constraints"] > 4096 and g["refreshes"] == 1 and g["builds"] == 1 and g["failed"] == 0
        assert res[key + "_bitwise"], res[key + "_maxabs"]
    res = _run_pypbd([True, False], example="beam_model_pypbd.py", configs=((2,), (4,)), edit=True)
    for key in ("2", "4"):
        g = res[key + "_gpu"]
        assert g["refreshes"] == 1 and g["builds"] == 1 and g["failed"] == 0
        assert res[key + "_bitwise"], res[key + "_maxabs"]


# ---------------------------------------------------------------------------
# dirty tracking at a size where a sampled hash would miss single-element edits: the plug-in looks at
# EVERY word of the host arrays (block hashes, include/pbdx.h) before every step
# ---------------------------------------------------------------------------
def _np_block_hashes(a):
    """pbdx_hash_block (include/pbdx.h) restated in numpy for a packed array."""
    raw = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    elem_bytes = raw.size // len(a)
    if raw.size % 8:
        raw = np.concatenate([raw, np.zeros(4, dtype=np.uint8)])          # an odd 32-bit word at the end counts as a 64-bit word with a zero upper half
    w = raw.view(np.uint64)
    i = np.arange(w.size, dtype=np.uint64)
    with np.errstate(over="ignore"):
        m = (w ^ (i * np.uint64(0x9E3779B97F4A7C15))) * np.uint64(0xD1B54A32D192ED03)
    m ^= m >> np.uint64(29)
    per = 1024 * elem_bytes // 8
    nb = (len(a) + 1023) // 1024
    return np.array([np.bitwise_xor.reduce(m[b * per:(b + 1) * per]) for b in range(nb)], dtype=np.uint64)


@pytest.mark.gpu
@pytest.mark.parametrize("n,mirror", [(5 * 1024 + 137, 0), (5 * 1024 + 137, 1), (700 * 1024 + 13, 1), (1500 * 1024 + 7, 0)])
def test_device_block_hashes_equal_the_host_definition(n, mirror):
    """pbdx_solver_get_particles_hashed: the hashes the device returns are the host definition applied to the delivered bytes
    (float and double hosts; a particle count that is not a multiple of the block size), and pbdx_solver_update_particle_ranges
    replaces exactly the given ranges.  mirror = 1: the same through the engine's page-locked mirror (PBDX_OPT_PIN_HOST; the large
    case is copied by several host threads per array); mirror = 0 goes through the library's bounce buffer (pbdx_hostio.hip: 8 MiB
    halves -- the 1.5 M case takes three per array)."""
    import positionbaseddynamics_amd as pbd
    from positionbaseddynamics_amd import _ffi
    rng = np.random.default_rng(7)
    x = rng.standard_normal((n, 3)).astype(np.float32)
    v = rng.standard_normal((n, 3)).astype(np.float32)
    mass = np.ones(n, dtype=np.float32)
    sol = pbd.Solver()
    sol.set_option(pbd.Solver.OPT_PIN_HOST, mirror)
    sol.set_particles(x, mass, v=v)
    pu64 = C.POINTER(C.c_uint64)
    nb = (n + 1023) // 1024
    for dt, fn, cptr in ((np.float32, _ffi.lib.pbdx_solver_get_particles_hashed, _ffi.pf), (np.float64, _ffi.lib.pbdx_solver_get_particles_hashed_f64, C.POINTER(C.c_double))):
        out = [np.zeros((n, 3), dtype=dt) for _ in range(4)]
        hs = [np.zeros(nb, dtype=np.uint64) for _ in range(4)]
        _ffi.check(fn(sol._h, n, *[o.ctypes.data_as(cptr) for o in out], *[h.ctypes.data_as(pu64) for h in hs]), "get_particles_hashed")
        assert np.array_equal(out[0], x.astype(dt)) and np.array_equal(out[1], v.astype(dt))
        for o, h in zip(out, hs):
            assert np.array_equal(h, _np_block_hashes(o))
    # partial upload: two ranges of x, one of v; everything else stays
    x2 = x.copy(); x2[10:20] += 1.0; x2[4000:4100] -= 2.0; x2[3000] = 99.0          # 3000 is NOT in a range: must not arrive
    rl = [10, 10, 4000, 100]
    if n > 600000:
        x2[100000:550000] *= np.float32(3.0); rl += [100000, 450000]                 # (a range several host threads copy)
    ranges = np.array(rl, dtype=np.uint32)
    _ffi.check(_ffi.lib.pbdx_solver_update_particle_ranges(sol._h, 0, x2.ctypes.data_as(_ffi.pf), len(rl) // 2, ranges.ctypes.data_as(_ffi.pu)), "update_ranges")
    want = x.copy()
    for q in range(0, len(rl), 2):
        want[rl[q]:rl[q] + rl[q + 1]] = x2[rl[q]:rl[q] + rl[q + 1]]
    got = sol.get_particles()
    assert np.array_equal(got[0], want) and np.array_equal(got[1], v)
    bad = np.array([n - 5, 10], dtype=np.uint32)
    assert _ffi.lib.pbdx_solver_update_particle_ranges(sol._h, 0, x2.ctypes.data_as(_ffi.pf), 1, bad.ctypes.data_as(_ffi.pu)) != 0
    # a double host: the whole image again, every array given, and back (values are floats: the conversion is exact both ways)
    pd_ = C.POINTER(C.c_double)
    arrs = [rng.standard_normal((n, 3)).astype(np.float32).astype(np.float64) for _ in range(4)]
    m64 = np.full(n, 2.0); w64 = np.full(n, 0.5)
    _ffi.check(_ffi.lib.pbdx_solver_set_particles_f64(sol._h, n, *[a.ctypes.data_as(pd_) for a in arrs], m64.ctypes.data_as(pd_), w64.ctypes.data_as(pd_)), "set_particles_f64")
    back = [np.zeros((n, 3)) for _ in range(4)]
    _ffi.check(_ffi.lib.pbdx_solver_get_particles_f64(sol._h, n, *[b.ctypes.data_as(pd_) for b in back]), "get_particles_f64")
    for a, b in zip(arrs, back):
        assert np.array_equal(a, b)
    sol.set_option(pbd.Solver.OPT_PIN_HOST, 0)            # (the mirror goes; transfers keep working)
    got = sol.get_particles()
    assert np.array_equal(got[0], arrs[0].astype(np.float32)) and np.array_equal(got[3], arrs[3].astype(np.float32))


def _extra(path):
    lib = C.CDLL(path)
    lib.pbdx_timestep_hip_partial_uploads.argtypes = [C.c_void_p]; lib.pbdx_timestep_hip_partial_uploads.restype = C.c_uint
    lib.pbdx_timestep_hip_refresh_parameters.argtypes = [C.c_void_p]; lib.pbdx_timestep_hip_refresh_parameters.restype = None
    lib.pbdx_timestep_hip_set_full_parameter_scan.argtypes = [C.c_void_p, C.c_int]; lib.pbdx_timestep_hip_set_full_parameter_scan.restype = None
    lib.pbdx_timestep_hip_speculative_steps.argtypes = [C.c_void_p]; lib.pbdx_timestep_hip_speculative_steps.restype = C.c_uint
    lib.pbdx_timestep_hip_repeated_steps.argtypes = [C.c_void_p]; lib.pbdx_timestep_hip_repeated_steps.restype = C.c_uint
    lib.pbdx_timestep_hip_set_speculative_step.argtypes = [C.c_void_p, C.c_int]; lib.pbdx_timestep_hip_set_speculative_step.restype = None
    lib.pbdx_timestep_hip_allow_reference_fallback.argtypes = [C.c_void_p, C.c_int]; lib.pbdx_timestep_hip_allow_reference_fallback.restype = None
    return lib


BIG = (320, 320)      # 102 400 particles, 611 522 constraints: a 4 096-element sample would see 1 particle in 25


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["f32", "f64"])
def test_plugin_sees_one_pinned_particle_in_a_large_model(variant):
    """ParticleData::setMass(i, 0) on ONE particle between step() calls (pinning / mouse attach, ParticleData.h:239-246): the
    reference reads pd.getInvMass(i) fresh in every step; so must the device image.  102 400 particles, the pinned particle
    sits in the middle of a hash block.  Bitwise vs the CPU TimeStepController making the same edit (float host; fp32
    envelope for the double host)."""
    refdrv, path = _plugin(variant)
    ops = util.cloth_spec(BIG[0], BIG[1], 4, 3)
    victim = 160 * 320 + 171

    def run(gpu):
        ref = refdrv.Ref(variant)
        _setup(ref, ops, 1, 5)
        ref.set_num_threads(8)
        if gpu:
            assert ref.install_timestep_plugin(path) == 0
        ref.set_params(1, 5, 0)
        ref.step(3)
        ref.set_mass(victim, 0.0)
        ref.step(3)
        x_pinned = ref.positions()[victim].copy()
        ref.step(2)
        assert np.array_equal(ref.positions()[victim], x_pinned), "the pinned particle moved"
        ref.set_mass(victim, 1.0)              # ... and released again
        ref.step(2)
        out = (ref.positions().copy(), ref.get_array(2).copy())
        if gpu:
            lib, cnt = _counters(path)
            ts = ref.timestep_ptr()
            assert cnt["gpu_steps"](ts) == 10 and cnt["failed_steps"](ts) == 0
            # one full upload (the first step); the two mass edits went up as changed blocks, and no step re-sent what the device
            # itself had delivered
            assert cnt["uploads"](ts) == 1 and _extra(path).pbdx_timestep_hip_partial_uploads(ts) == 2
        ref.reset_all()
        return out

    (xc, vc), (xg, vg) = run(False), run(True)
    if variant == "f32":
        assert util.bitwise_equal(xg, xc), "max err %.3e" % util.max_err(xg, xc)
        assert util.bitwise_equal(vg, vc)
    else:
        # pinning and releasing a particle of a stiff falling sheet is a violent event: float and double runs drift apart
        # quickly.  The bar for a double host is the envelope: the fp32 engine is no further from the double CPU
        # path than 3x what the reference's OWN float build is
        refdrv32, _ = _plugin("f32")
        ref = refdrv32.Ref("f32")
        _setup(ref, ops, 1, 5)
        ref.set_num_threads(8)
        ref.set_params(1, 5, 0)
        ref.step(3); ref.set_mass(victim, 0.0); ref.step(5); ref.set_mass(victim, 1.0); ref.step(2)
        x32 = ref.positions().copy()
        ref.reset_all()
        e_gpu, e_f32 = util.max_err(xg, xc), util.max_err(x32, xc)
        print("double host, one particle pinned / released: |gpu - f64| = %.3e, |f32 reference - f64| = %.3e" % (e_gpu, e_f32))
        assert e_gpu <= 3.0 * e_f32 + 1e-6


@pytest.mark.gpu
def test_plugin_sees_one_moved_particle_between_resident_steps():
    """Resident mode contract (TimeStepControllerHIP.h): host arrays are stale until syncToHost; after it, host writes are
    found by full-coverage block hashes and only the changed blocks are uploaded.  ONE particle of 102 400 is moved (and given
    a velocity) between resident steps; bitwise vs the CPU TimeStepController making the same edit."""
    refdrv, path = _plugin("f32")
    ops = util.cloth_spec(BIG[0], BIG[1], 4, 3)
    victim = 200 * 320 + 57

    def edit(ref):
        x = ref.positions().copy(); x[victim] += np.array([0.0, 0.05, 0.02], dtype=x.dtype); ref.set_array(0, x)
        v = ref.get_array(2).copy(); v[victim, 1] = 1.5; ref.set_array(2, v)

    def run(gpu):
        ref = refdrv.Ref("f32")
        _setup(ref, ops, 1, 5)
        ref.set_num_threads(8)
        ref.set_params(1, 5, 0)
        if gpu:
            assert ref.install_timestep_plugin(path) == 0
            ref.set_params(1, 5, 0)
            lib, cnt = _counters(path)
            ts, model = ref.timestep_ptr(), ref.model_ptr()
            assert cnt["step_resident"](ts, model, 4) == 0
            assert cnt["sync_to_host"](ts, model) == 0
            edit(ref)
            assert cnt["step_resident"](ts, model, 3) == 0
            assert cnt["uploads"](ts) == 1 and _extra(path).pbdx_timestep_hip_partial_uploads(ts) == 1
            assert cnt["step_resident"](ts, model, 2) == 0
            assert cnt["uploads"](ts) == 1 and _extra(path).pbdx_timestep_hip_partial_uploads(ts) == 1
            assert cnt["sync_to_host"](ts, model) == 0
            assert cnt["failed_steps"](ts) == 0
        else:
            ref.step(4)
            edit(ref)
            ref.step(5)
        out = [ref.get_array(k).copy() for k in (0, 2, 4, 5)]
        ref.reset_all()
        return out

    a, b = run(False), run(True)
    for k in range(4):
        assert util.bitwise_equal(a[k], b[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["unannounced", "sampled+refreshParameters"])
@pytest.mark.parametrize("size", ["clothdemo50", "big"])
def test_plugin_sees_one_edited_constraint(size, how):
    """`constraint.stiffness = ...` on ONE constraint (pyPBD/ConstraintsModule.cpp:62-116), in the 50x50 ClothDemo sheet (14 406
    constraints: already more than the 4 096 the sampled check looks at) and in a 611 522-constraint sheet.  Default: NOTHING is
    announced -- the exact parameter scan (every record of every constraint, every step) finds it.  Opt-in sampled check: the host
    announces the edit with refreshParameters().  Either way: parameter streams refreshed once, no replanning, bitwise vs the CPU
    TimeStepController making the same edit."""
    refdrv, path = _plugin("f32")
    ops = util.cloth_spec(50, 50, 1, 2) if size == "clothdemo50" else util.cloth_spec(BIG[0], BIG[1], 4, 3)
    victim = 9001 if size == "clothdemo50" else 123457

    def run(gpu):
        ref = refdrv.Ref("f32")
        _setup(ref, ops, 1, 5)
        ref.set_num_threads(8 if size == "big" else 1)
        if gpu:
            assert ref.install_timestep_plugin(path) == 0
        ref.set_params(1, 5, 0)
        if gpu and how != "unannounced":
            _extra(path).pbdx_timestep_hip_set_full_parameter_scan(ref.timestep_ptr(), 0)
        ref.step(3)
        ref.set_constraint_stiffness(victim, 3.0 if size == "big" else 0.37)
        if gpu and how != "unannounced":
            _extra(path).pbdx_timestep_hip_refresh_parameters(ref.timestep_ptr())
        ref.step(4)
        out = (ref.positions().copy(), ref.get_array(2).copy())
        if gpu:
            lib, cnt = _counters(path)
            ts = ref.timestep_ptr()
            assert cnt["gpu_steps"](ts) == 7 and cnt["failed_steps"](ts) == 0
            assert cnt["schedule_builds"](ts) == 1 and cnt["param_refreshes"](ts) == 1
            ex = _extra(path)
            ex.pbdx_timestep_hip_speculative_steps.restype = C.c_uint; ex.pbdx_timestep_hip_repeated_steps.restype = C.c_uint
            spec, rep = ex.pbdx_timestep_hip_speculative_steps(ts), ex.pbdx_timestep_hip_repeated_steps(ts)
            print("speculative steps %d, repeated %d (%s, %s)" % (spec, rep, size, how))
            if how == "unannounced":
                # the exact scan runs WHILE the device steps: every step but the first (which builds the schedule) is speculative, and exactly the one
                # step that followed the edit had to be undone on the device and repeated
                assert spec == 6 and rep == 1
            else:
                assert rep == 0
        ref.reset_all()
        return out

    (xc, vc), (xg, vg) = run(False), run(True)
    assert not np.array_equal(xc, util.oracle_positions(ops, 7, 1, 5, "f32")), "the edit changes nothing: the test would prove nothing"
    assert util.bitwise_equal(xg, xc), "max err %.3e" % util.max_err(xg, xc)
    assert util.bitwise_equal(vg, vc)


@pytest.mark.gpu
@pytest.mark.parametrize("fallback", [False, True])
def test_plugin_failed_repeat_of_a_speculative_step_leaves_the_pre_step_state(fallback):
    """the speculative step() downloads its result before the parameter scan has finished.  If the scan then finds an edit and the REPEAT
    fails, the host must not keep the stale result: without a fallback the step is reported as not executed and ParticleData holds the pre-step
    state (the next step continues from it, bitwise like a CPU run that made the same edit); with the reference fallback opted in the CPU
    controller steps ONCE from that pre-step state (no double step)."""
    refdrv, path = _plugin("f32")
    ops = util.cloth_spec(50, 50, 1, 2)
    victim = 9001
    ex = _extra(path)
    ex.pbdx_timestep_hip_set_fail_repeat_for_test.argtypes = [C.c_void_p, C.c_int]; ex.pbdx_timestep_hip_set_fail_repeat_for_test.restype = None
    lib, cnt = _counters(path)

    def run(gpu):
        ref = refdrv.Ref("f32")
        _setup(ref, ops, 1, 5)
        ref.set_num_threads(1)
        if gpu:
            assert ref.install_timestep_plugin(path) == 0
        ref.set_params(1, 5, 0)
        ref.step(3)
        before = [ref.get_array(k).copy() for k in (0, 2, 4, 5)]
        ref.set_constraint_stiffness(victim, 0.37)
        if gpu:
            ts = ref.timestep_ptr()
            ex.pbdx_timestep_hip_allow_reference_fallback(ts, 1 if fallback else 0)
            ex.pbdx_timestep_hip_set_fail_repeat_for_test(ts, 1)
            ref.step(1)                      # speculative step, edit found, repeat made to fail
            ex.pbdx_timestep_hip_set_fail_repeat_for_test(ts, 0)
            after = [ref.get_array(k).copy() for k in (0, 2, 4, 5)]
            if fallback:
                assert cnt["fallback_steps"](ts) == 1 and cnt["failed_steps"](ts) == 0
            else:
                assert cnt["failed_steps"](ts) == 1
                for k in range(4):
                    assert util.bitwise_equal(after[k], before[k]), "array %d: the host holds a state other than the pre-step one" % k
                ref.step(1)                  # the step the host asked for, now executed
            ref.step(2)
            assert cnt["failed_steps"](ts) == (0 if fallback else 1)
        else:
            ref.step(3)
        out = [ref.get_array(k).copy() for k in (0, 2, 4, 5)]
        ref.reset_all()
        return out

    a, b = run(False), run(True)
    for k in range(4):
        assert util.bitwise_equal(a[k], b[k]), "array %d" % k
