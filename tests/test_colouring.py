"""Constraint colouring on the device (positionbaseddynamics_amd/csrc/pbdx_colour.hip).

The reference colours greedily in creation order (SimulationModel::initConstraintGroups, SimulationModel.cpp:1033-1094); the engine
consumes that colouring verbatim, so the device form must reproduce it group for group.  The host form (pbdx_model.cpp) is the
checker here: it is itself compared integer for integer with the reference in tests/test_model_vs_reference.py."""
import time

import numpy as np
import pytest

from tests import util


@pytest.fixture(scope="module")
def pbd():
    import positionbaseddynamics_amd as pbd
    assert pbd.device_count() > 0, "GPU tests need a MI355X: the engine has no CPU fallback"
    return pbd


@pytest.fixture(scope="module")
def pbd_cpu():
    import positionbaseddynamics_amd as pbd
    return pbd


def reference_first_fit(num_bodies, body_off, bodies):
    """The recurrence of SimulationModel.cpp:1046-1083 restated directly (small inputs only)."""
    groups_of_body = [set() for _ in range(num_bodies)]
    out = []
    for c in range(len(body_off) - 1):
        bs = bodies[body_off[c]:body_off[c + 1]]
        used = set().union(*[groups_of_body[b] for b in bs])
        g = 0
        while g in used:
            g += 1
        out.append(g)
        for b in bs:
            groups_of_body[b].add(g)
    return np.array(out, dtype=np.uint32)


def test_entry_points_without_a_device_fail_loudly(pbd_cpu):
    """No GPU: PBDX_ERR_NO_DEVICE, never a host fallback behind the caller's back (malformed input is refused before the device is looked for)."""
    pbd = pbd_cpu
    import ctypes as C
    from positionbaseddynamics_amd import _ffi
    off = np.array([0, 2, 4], dtype=np.uint32)
    bodies = np.array([0, 1, 1, 2], dtype=np.uint32)
    if pbd.device_count() == 0:
        with pytest.raises(Exception):
            pbd.colour_constraints(3, off, bodies)
        m = util.build_mine(util.cloth_spec(6, 6, 4, 3))
        with pytest.raises(Exception):
            m.initConstraintGroups(device=0)
        assert not _ffi.lib.pbdx_model_groups_initialized(m._h)
        m.initConstraintGroups()                      # the host form still works afterwards
        assert _ffi.lib.pbdx_model_groups_initialized(m._h)
    bad = np.array([0, 5], dtype=np.uint32)           # five bodies
    g = np.zeros(1, dtype=np.uint32)
    r = _ffi.lib.pbdx_colour_constraints(0, 8, 1, bad.ctypes.data_as(C.POINTER(C.c_uint32)), np.arange(5, dtype=np.uint32).ctypes.data_as(C.POINTER(C.c_uint32)),
                                         g.ctypes.data_as(C.POINTER(C.c_uint32)), None, None)
    assert r == 4                                     # PBDX_ERR_UNSUPPORTED


def test_host_colouring_on_raw_arrays_is_the_reference_recurrence(pbd_cpu):
    """pbdx_colour_constraints_host (what the model mirror and the reference-side plug-in colour with) against the recurrence restated in
    python: random constraints of 1..6 bodies, also beyond 64 and 128 groups (the bit masks grow)."""
    pbd = pbd_cpu
    rng = np.random.default_rng(3)
    for trial in range(8):
        nb = int(rng.integers(5, 300))
        nc = int(rng.integers(1, 2500))
        sizes = [int(min(s, nb)) for s in rng.integers(1, 7, size=nc)]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
        bodies = np.concatenate([rng.choice(nb, size=s, replace=False) for s in sizes]).astype(np.uint32)
        got, ng = pbd.colour_constraints_host(nb, off, bodies)
        want = reference_first_fit(nb, off, bodies)
        assert np.array_equal(got, want) and ng == want.max() + 1, trial
    off = np.arange(0, 2 * 200 + 1, 2, dtype=np.uint32)                  # a star: 200 groups
    bodies = np.stack([np.zeros(200, dtype=np.uint32), np.arange(1, 201, dtype=np.uint32)], axis=1).reshape(-1)
    got, ng = pbd.colour_constraints_host(201, off, bodies)
    assert ng == 200 and np.array_equal(got, np.arange(200))
    with pytest.raises(Exception):
        pbd.colour_constraints_host(3, np.array([0, 2], dtype=np.uint32), np.array([1, 7], dtype=np.uint32))


SCENES = {
    "cloth 40x30, XPBD distance + bending": lambda: util.cloth_spec(40, 30, 4, 3),
    "cloth 37x91, PBD distance + isometric bending": lambda: util.cloth_spec(37, 91, 1, 2),
    "irregular triangle mesh": lambda: util.delaunay_cloth_spec(),
    "bar 20x6x5 FEM tets": lambda: util.bar_spec(20, 6, 5, 2),
    "bar 20x6x5 XPBD distance + volume (57 colours)": lambda: util.bar_spec(20, 6, 5, 6),
    "irregular tet mesh": lambda: util.delaunay_solid_spec(),
    "three cloth instances": lambda: util.cloth_spec(24, 24, 4, 3, instances=3, instanced=True),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENES))
def test_device_colouring_equals_host_colouring(pbd, name):
    ops = SCENES[name]()
    host = util.build_mine(ops)
    dev = util.build_mine(ops)
    hg = host.getConstraintGroups()
    dev.initConstraintGroups(device=0)
    dg = dev.getConstraintGroups()
    assert len(hg) == len(dg), name
    for a, b in zip(hg, dg):
        assert np.array_equal(a, b), name
    print("%-50s %d groups, %d constraints: device == host" % (name, len(hg), sum(len(g) for g in hg)))


@pytest.mark.gpu
def test_device_colouring_on_raw_arrays_random_hypergraphs(pbd):
    """Random constraints of 1..4 bodies (no structure at all) against the recurrence restated in python; and the refusals."""
    rng = np.random.default_rng(11)
    for trial in range(6):
        nb = int(rng.integers(5, 400))
        nc = int(rng.integers(1, 3000))
        sizes = rng.integers(1, 5, size=nc)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
        bodies = np.concatenate([rng.choice(nb, size=min(s, nb), replace=False) if s <= nb else rng.choice(nb, size=nb, replace=False) for s in sizes]).astype(np.uint32)
        off = np.concatenate([[0], np.cumsum([min(s, nb) for s in sizes])]).astype(np.uint32)
        want = reference_first_fit(nb, off, bodies)
        if want.max() >= 128:
            with pytest.raises(Exception):
                pbd.colour_constraints(nb, off, bodies)
            continue
        got, ng, rounds = pbd.colour_constraints(nb, off, bodies)
        assert np.array_equal(got, want), trial
        assert ng == want.max() + 1 and 1 <= rounds <= nc
    # a dense clique needs more than 128 groups: refused, not truncated
    off = np.arange(0, 2 * 200 + 1, 2, dtype=np.uint32)
    bodies = np.stack([np.zeros(200, dtype=np.uint32), np.arange(1, 201, dtype=np.uint32)], axis=1).reshape(-1)
    with pytest.raises(Exception):
        pbd.colour_constraints(201, off, bodies)
    # the same body twice in one constraint: refused
    with pytest.raises(Exception):
        pbd.colour_constraints(4, np.array([0, 2], dtype=np.uint32), np.array([1, 1], dtype=np.uint32))


@pytest.mark.gpu
def test_device_colouring_full_size_c2(pbd):
    """BASELINE configs[1]: the 1000x1000 cloth (5 988 006 constraints, 27 groups): group for group identical; timing printed."""
    ops = util.cloth_spec(1000, 1000, 4, 3)
    host = util.build_mine(ops)
    dev = util.build_mine(ops)
    t0 = time.perf_counter(); host.initConstraintGroups(); t_host = time.perf_counter() - t0
    dev.initConstraintGroups(device=0)              # (first call: device context, allocation)
    dev2 = util.build_mine(ops)
    t0 = time.perf_counter(); dev2.initConstraintGroups(device=0); t_dev = time.perf_counter() - t0
    hg, dg = host.getConstraintGroups(), dev2.getConstraintGroups()
    assert len(hg) == len(dg) == 27
    for a, b in zip(hg, dg):
        assert np.array_equal(a, b)
    print("1000x1000 cloth: host colouring %.3f s, device colouring %.3f s (upload, sort, propagation, download, group lists)" % (t_host, t_dev))
