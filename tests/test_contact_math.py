"""Host-side known-answer tests of the contact arithmetic (csrc/pbdx_contact.h is host + device code): no GPU needed."""
import numpy as np
import pytest

from tests import util


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
