"""particle vs static rigid body contacts with analytic distance fields on the device.

Oracle = the reference's own DistanceFieldCollisionDetection + ParticleRigidBodyContactConstraint
(oracle/_ref, unmodified sources) on a cloth / solid falling onto static colliders
(Demos/DistanceFieldDemos/ClothCollisionDemo.cpp restated with procedural collider meshes).  The product
receives the colliders through the raw C ABI (pbdx_solver_set_colliders, fed with the transformation the
reference computed for each rigid body) and must reproduce positions AND velocities bit for bit: contacts
only act on velocities, so any deviation in detection, contact initialisation or the velocity sweeps shows
up in the next step's positions."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

SCENES = {
    # name: (particle ops, colliders [(shape, pos, quat(w,x,y,z), bbox, params, restitution, friction, invert)], sub, iters, steps)
    "cloth on torus + floor": (util.cloth_spec(30, 30, 4, 3, T=(-5, 4, -5), pin=False),
                               [("box", (0, -2.5, 0), (1, 0, 0, 0), (100, 1, 100), (100, 1, 100), 0.6, 0.2, False),
                                ("torus", (0, 1.5, 0), (1, 0, 0, 0), (6, 2, 6), (2, 1), 0.6, 0.1, False)], 1, 5, 320),
    "cloth on rotated box + sphere": (util.cloth_spec(24, 24, 4, 3, T=(-4, 3, -4), pin=False, width=8.0, height=8.0),
                                      [("box", (0.5, 0.0, 0.3), (0.9238795, 0.0, 0.0, 0.3826834), (3, 1, 2), (3, 1, 2), 0.5, 0.3, False),
                                       ("sphere", (-2.0, 0.5, 1.0), (1, 0, 0, 0), (2.4, 2.4, 2.4), (1.2,), 0.7, 0.05, False),
                                       ("cylinder", (2.5, -0.5, -2.0), (1, 0, 0, 0), (1.6, 3.0, 1.6), (0.8, 3.0), 0.6, 0.2, False)], 2, 4, 240),
    "config-5-like: three FEM tet solids on a static floor, 5 substeps x 1 iteration": (
        util.config5_like_spec(), [("box", (0, 0, 0), (1, 0, 0, 0), (100, 1, 100), (100, 1, 100), 0.6, 0.0, False)], 5, 1, 280),
    "bar inside hollow sphere": (util.bar_spec(8, 3, 3, 6, T=(-1.5, 0.5, -0.3), scale=(3.0, 0.6, 0.6))[:1] + [("solid", 0, 6, 100000.0, 0.3, 100000.0, False, False)],
                                 [("hollow_sphere", (0, 0, 0), (1, 0, 0, 0), (6.2, 6.2, 6.2), (3.0, 0.1), 0.6, 0.2, False)], 1, 5, 200),
}


def _setup_ref(ref, ops, colliders, sub, iters, tolerance):
    util.apply_ref(ref, ops)
    ref.set_num_threads(1)
    ref.set_time_step_size(0.005)
    ref.set_gravity(util.GRAVITY)
    ref.set_params(sub, iters, 0)
    for shape, pos, quat, bbox, params, rest, fric, inv in colliders:
        ref.add_static_collider(shape, pos, quat, bbox, params, rest, fric, inv)
    ref.enable_collisions(tolerance, 0.1, 0.3)


@pytest.mark.parametrize("name", list(SCENES))
def test_contacts_vs_reference(name):
    import positionbaseddynamics_amd as pbd
    ops, colliders, sub, iters, steps = SCENES[name]
    ref = util.get_oracle("f32")
    _setup_ref(ref, ops, colliders, sub, iters, 0.05)
    cols, ranges, tol, stiff = ref.collision_objects()

    m = util.build_mine(ops)
    sim = pbd.Simulation(); pbd.Simulation.setCurrent(sim); sim.setVecValueFloat(pbd.Simulation.GRAVITATION, util.GRAVITY)
    pbd.TimeManager.setCurrent(pbd.TimeManager()); pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
    ts = pbd.TimeStepController()
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, sub)
    ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, iters)
    ts.syncFromHost(m)              # uploads the particles (collision ranges are validated against them)
    sol = ts.solver()
    sol.set_colliders(cols)
    sol.set_collision_ranges(ranges)
    sol.set_contact_params(tol, stiff, 5)
    seen = 0
    chunk = 40
    for k in range(0, steps, chunk):
        ref.step(chunk)
        for _ in range(chunk):
            ts.step(m)
        nref = len(ref.contacts())
        ngpu = sol.num_contacts()
        seen = max(seen, nref)
        xr, vr = ref.positions().astype(np.float32), ref.get_array(2).astype(np.float32)
        xg, vg = m.getParticles().positions(), m.getParticles().array(2)
        ex, ev = util.max_err(xg, xr), util.max_err(vg, vr)
        print("%-32s step %4d  contacts ref %4d gpu %4d  max|dx| %.3e max|dv| %.3e" % (name, k + chunk, nref, ngpu, ex, ev))
        assert ngpu == nref, "contact count differs"
        assert util.bitwise_equal(xg, xr) and util.bitwise_equal(vg, vr), "state differs from the reference with contacts"
    assert seen > 0, "scene never produced a contact: the test would be vacuous"


def test_contact_with_a_body_of_finite_mass_host_functions_vs_reference():
    """The arithmetic of a contact between a particle and a rigid body of ANY mass (csrc/pbdx_contact.h: compute_matrix_k, dyn_contact_init,
    dyn_contact_velocity_solve), run on the HOST, against the reference's own init_ParticleRigidBodyContactConstraint /
    velocitySolve_ParticleRigidBodyContactConstraint applied as ParticleRigidBodyContactConstraint::solveVelocityConstraint applies them
    (PositionBasedRigidBodyDynamics.cpp:11-45,2385-2539, Constraints.cpp:2148-2189): 400 random contacts -- static and dynamic bodies, pinned and
    free particles, full symmetric world inertia tensors, penetrating and separating -- three sweeps each, bit for bit."""
    import ctypes as C
    from oracle import refdrv
    from positionbaseddynamics_amd import _ffi
    if not refdrv.available("f32"):
        pytest.skip("reference build not present")
    ref = refdrv.Ref("f32")
    ref.lib.refdrv_dyn_contact_kat.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    ref.lib.refdrv_dyn_contact_kat.restype = None
    rng = np.random.default_rng(20260930)
    worst = 0
    for case in range(400):
        a = rng.standard_normal((3, 3))
        ji = (a @ a.T * 0.3 + np.eye(3) * 0.2).astype(np.float32)          # symmetric positive definite
        ji = ((ji + ji.T) * np.float32(0.5)).astype(np.float32)
        n = rng.standard_normal(3); n /= np.linalg.norm(n)
        cp1 = rng.standard_normal(3) * 2.0
        cp0 = cp1 + n * rng.uniform(-0.05, 0.05) + rng.standard_normal(3) * 0.01
        w0 = 0.0 if case % 11 == 0 else rng.uniform(0.2, 3.0)
        w1 = 0.0 if case % 5 == 0 else rng.uniform(0.01, 0.5)
        row = np.concatenate([[w0, 0.0 if w0 == 0.0 else 1.0 / w0], rng.standard_normal(3) * 3.0, [w1], rng.standard_normal(3), rng.standard_normal(3), ji.reshape(-1),
                              rng.standard_normal(3), cp0, cp1, n, [rng.uniform(0.0, 0.9), 100.0, rng.uniform(0.0, 0.6), 3.0, 0.0]]).astype(np.float32)
        assert row.size == 38
        mine = np.zeros(20, dtype=np.float32)
        _ffi.check(_ffi.lib.pbdx_debug_dyn_contact_kat(row.ctypes.data_as(_ffi.pf), mine.ctypes.data_as(_ffi.pf)), "dyn_contact_kat")
        rin = row.astype(np.float64)
        rout = np.zeros(20, dtype=np.float64)
        ref.lib.refdrv_dyn_contact_kat(rin.ctypes.data_as(C.POINTER(C.c_double)), rout.ctypes.data_as(C.POINTER(C.c_double)))
        theirs = rout.astype(np.float32)
        if not util.bitwise_equal(mine, theirs):
            worst += 1
            print("case", case, "mine", mine[:16], "reference", theirs[:16])
    assert worst == 0, "%d of 400 contacts differ from the reference" % worst
