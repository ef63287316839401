"""Contact arithmetic of the engine (pbdx_contact.h) pinned against the reference WITHOUT a GPU: the
header is host+device code, so a tiny helper (tests/helpers/host_contact.cpp) runs it on the CPU.  For every
particle that the reference's DistanceFieldCollisionDetection puts in contact, the engine's detection +
contact initialisation + velocity sweeps must return the reference's post-step velocity bit for bit, and no
other particle may change."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from tests import util


def _helper():
    src = os.path.join(util.ROOT, "tests", "helpers", "host_contact.cpp")
    out = os.path.join(tempfile.gettempdir(), "pbdx_host_contact_%d.so" % os.getuid())
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(
            os.path.join(util.ROOT, "positionbaseddynamics_amd", "csrc", "pbdx_contact.h"))):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", out, src])
    return C.CDLL(out)


SCENES = {
    "cloth on torus + floor": (util.cloth_spec(26, 26, 4, 3, T=(-5, 4, -5), pin=False),
                               [("box", (0, -2.5, 0), (1, 0, 0, 0), (100, 1, 100), (100, 1, 100), 0.6, 0.2, False),
                                ("torus", (0, 1.5, 0), (1, 0, 0, 0), (6, 2, 6), (2, 1), 0.6, 0.1, False)], 140),
    "cloth on rotated box, sphere, cylinder, hollow box": (
        util.cloth_spec(24, 24, 4, 3, T=(-4, 3, -4), pin=False, width=8.0, height=8.0),
        [("box", (0.5, 0.0, 0.3), (0.9238795, 0.0, 0.0, 0.3826834), (3, 1, 2), (3, 1, 2), 0.5, 0.3, False),
         ("sphere", (-2.0, 0.5, 1.0), (1, 0, 0, 0), (2.4, 2.4, 2.4), (1.2,), 0.7, 0.05, False),
         ("cylinder", (2.5, -0.5, -2.0), (1, 0, 0, 0), (1.6, 3.0, 1.6), (0.8, 3.0), 0.6, 0.2, False),
         ("hollow_box", (0.0, -3.0, 0.0), (1, 0, 0, 0), (12.4, 2.4, 12.4), (12, 2, 12, 0.2), 0.6, 0.2, False)], 200),
    "bar in hollow sphere": (util.bar_spec(8, 3, 3, 6, T=(-1.5, 0.5, -0.3), scale=(3.0, 0.6, 0.6))[:1] + [("solid", 0, 6, 100000.0, 0.3, 100000.0, False, False)],
                             [("hollow_sphere", (0, 0, 0), (1, 0, 0, 0), (6.2, 6.2, 6.2), (3.0, 0.1), 0.6, 0.2, False)], 30),
}


def _prepare(ref, ops, colliders, collide):
    util.apply_ref(ref, ops)
    ref.set_num_threads(1)
    ref.set_time_step_size(0.005)
    ref.set_gravity(util.GRAVITY)
    ref.set_params(1, 5, 0)
    if collide:
        for shape, pos, quat, bbox, params, rest, fric, inv in colliders:
            ref.add_static_collider(shape, pos, quat, bbox, params, rest, fric, inv)
        ref.enable_collisions(0.05, 0.6, 0.1)


@pytest.mark.parametrize("name", list(SCENES))
def test_contact_math_matches_reference_on_the_host(name):
    from oracle import refdrv
    from positionbaseddynamics_amd import _ffi
    if not refdrv.available("f32"):
        pytest.skip("reference build not present")
    lib = _helper()
    ops, colliders, steps = SCENES[name]
    ref = refdrv.Ref("f32")
    _prepare(ref, ops, colliders, True)
    cols, ranges, tol, stiff = ref.collision_objects()
    arr = (_ffi.Collider * len(cols))()
    for i, col in enumerate(cols):
        k = arr[i]
        k.shape, k.invert = col["shape"], int(col["invert"])
        for j in range(4):
            k.params[j] = col["params"][j]
        for nm, n in (("com", 3), ("R", 9), ("v1", 3), ("v2", 3)):
            for j in range(n):
                getattr(k, nm)[j] = np.float32(col[nm][j])
        k.restitution, k.friction = col["restitution"], col["friction"]
    first, count, mrest, mfric = ranges[0]
    pf = C.POINTER(C.c_float)
    checked = 0
    # step the colliding reference; before every step remember the state, then replay that ONE step without
    # collisions to obtain the pre-contact velocities (contacts act after the velocity update)
    for s in range(steps):
        state = [ref.get_array(w).copy() for w in (0, 2, 4, 5)]
        ref.step(1)
        contacts = ref.contacts()
        if not len(contacts):
            continue
        x_after, v_after = ref.positions().astype(np.float32), ref.get_array(2).astype(np.float32)
        mass, inv = ref.get_array(6).astype(np.float32), ref.get_array(7).astype(np.float32)
        keep = [ref.get_array(w).copy() for w in (0, 2, 4, 5)]
        # replay without contacts: same model, velocity iterations switched off
        for w, a in zip((0, 2, 4, 5), state):
            ref.set_array(w, a)
        ref.set_max_iterations_v(0)
        ref.step(1)
        ref.set_max_iterations_v(5)
        x_pre, v_pre = ref.positions().astype(np.float32), ref.get_array(2).astype(np.float32)
        assert np.array_equal(x_pre, x_after)              # contacts never move positions within the step
        touched = sorted(set(int(r[0]) for r in contacts))
        for p in range(first, first + count):
            v = v_pre[p].copy()
            nc = lib.host_particle_contacts(x_pre[p].ctypes.data_as(pf), v.ctypes.data_as(pf), C.c_float(inv[p]), C.c_float(mass[p]),
                                            arr, len(cols), C.c_float(tol), C.c_float(stiff), C.c_float(mrest), C.c_float(mfric), 5)
            assert nc == sum(1 for r in contacts if int(r[0]) == p), (name, s, p)
            assert np.array_equal(v.view(np.uint32), v_after[p].view(np.uint32)), (name, s, p, v, v_after[p])
        checked += len(touched)
        for w, a in zip((0, 2, 4, 5), keep):
            ref.set_array(w, a)
        if checked > 400:
            break
    assert checked > 0, "scene never produced a contact"
    print("%s: %d particle-contact chains verified bit-exact on the host" % (name, checked))
