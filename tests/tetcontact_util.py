"""Helpers of the particle-tet contact tests: the two-bar collision scene on the oracle and the conversion of the oracle's
collision objects (box distance fields on tet models + the bounding-sphere hierarchies the REFERENCE constructed) into the
engine's pbdx_tet_collider records."""
import ctypes as C

import numpy as np

from positionbaseddynamics_amd import _ffi

DIMS = (8, 3, 3)
SCALE = (2.0, 0.5, 0.5)
T_LOWER = (0.0, 0.0, 0.0)
T_UPPER = (0.3, 0.62, 0.05)


def two_bar_scene(ref, solid_method=6, tolerance=0.01, friction=0.0, dims=DIMS, t_upper=T_UPPER):
    """Lower bar clamped at both ends, upper bar falling onto it; both tet models carry a box distance field in their rest frame
    and test their particles against the other solid (DistanceFieldCollisionDetection.cpp:160-177)."""
    ref.reset_all()
    ref.set_num_threads(1)
    ref.set_time_step_size(0.005)
    ref.set_gravity((0, -9.81, 0))
    ref.add_regular_tet_model(*dims, T_LOWER, None, SCALE)
    ref.add_regular_tet_model(*dims, t_upper, None, SCALE)
    w, h, d = dims
    for j in range(h):
        for k in range(d):
            ref.set_mass(j * d + k, 0.0)
            ref.set_mass(((w - 1) * h + j) * d + k, 0.0)
    for tm in (0, 1):
        ref.add_solid_constraints(tm, solid_method, 1e5 if solid_method in (3, 6) else 1.0, 0.3, 1e5 if solid_method == 6 else 1.0, False, False)
    ref.set_collision_tolerance(tolerance)
    ref.set_tet_model_initial_transform(0, T_LOWER)
    ref.set_tet_model_initial_transform(1, t_upper)
    objs = [ref.add_tet_collision_box(tm, SCALE, True, 0.6, friction) for tm in (0, 1)]
    ref.attach_collision_detection()
    return objs


def stacked_bars_scene(ref, n_bars=3, solid_method=6, tolerance=0.01, dims=DIMS):
    """n bars stacked with gaps, the lowest clamped at both ends: every ordered pair (i, k) of solids whose boxes meet is tested,
    i outer (DistanceFieldCollisionDetection.cpp:33-46) -- the contact list interleaves several pairs."""
    ref.reset_all()
    ref.set_num_threads(1)
    ref.set_time_step_size(0.005)
    ref.set_gravity((0, -9.81, 0))
    offsets = [(0.3 * (q % 2), 0.62 * q, 0.05 * (q % 2)) for q in range(n_bars)]
    for t in offsets:
        ref.add_regular_tet_model(*dims, t, None, SCALE)
    w, h, d = dims
    for j in range(h):
        for k in range(d):
            ref.set_mass(j * d + k, 0.0)
            ref.set_mass(((w - 1) * h + j) * d + k, 0.0)
    for tm in range(n_bars):
        ref.add_solid_constraints(tm, solid_method, 1e5 if solid_method in (3, 6) else 1.0, 0.3, 1e5 if solid_method == 6 else 1.0, False, False)
    ref.set_collision_tolerance(tolerance)
    for tm, t in enumerate(offsets):
        ref.set_tet_model_initial_transform(tm, t)
    objs = [ref.add_tet_collision_box(tm, SCALE, True, 0.6, 0.0) for tm in range(n_bars)]
    ref.attach_collision_detection()
    return objs


def two_bar_scene_shapes(ref, shapes, solid_method=6, tolerance=0.01, dims=DIMS, t_upper=T_UPPER):
    """two_bar_scene with other analytic distance fields in the bars' rest frames: shapes = [(shape id, params, invert)] * 2.  The field
    need not look like the mesh: a contact is "a particle of the other solid, inside one of my tets, at a point of my rest shape where
    my field is negative"."""
    ref.reset_all()
    ref.set_num_threads(1)
    ref.set_time_step_size(0.005)
    ref.set_gravity((0, -9.81, 0))
    ref.add_regular_tet_model(*dims, T_LOWER, None, SCALE)
    ref.add_regular_tet_model(*dims, t_upper, None, SCALE)
    w, h, d = dims
    for j in range(h):
        for k in range(d):
            ref.set_mass(j * d + k, 0.0)
            ref.set_mass(((w - 1) * h + j) * d + k, 0.0)
    for tm in (0, 1):
        ref.add_solid_constraints(tm, solid_method, 1e5 if solid_method in (3, 6) else 1.0, 0.3, 1e5 if solid_method == 6 else 1.0, False, False)
    ref.set_collision_tolerance(tolerance)
    ref.set_tet_model_initial_transform(0, T_LOWER)
    ref.set_tet_model_initial_transform(1, t_upper)
    objs = [ref.add_tet_collision_shape(tm, sh[0], sh[1], True, sh[2], 0.6, 0.0) for tm, sh in zip((0, 1), shapes)]
    ref.attach_collision_detection()
    return objs


class TetColliders:
    """pbdx_tet_collider array built from the oracle's collision objects; keeps the numpy arrays alive."""

    def __init__(self, ref, objs, tet_models, tolerance, friction=0.0):
        self.keep = []
        self.n = len(objs)
        self.arr = (_ffi.TetCollider * self.n)()
        self.tolerance = tolerance
        n_per = []
        for q, (co, tm) in enumerate(zip(objs, tet_models)):
            info = ref.tet_model_info(tm)
            c = self.arr[q]
            shape, invert, params = ref.collision_object_shape(co)     # as stored: m_box = 0.5 * box (DistanceFieldCollisionDetection.cpp:502), ...
            assert shape >= 0
            c.shape, c.invert = shape, invert
            for k in range(4):
                c.params[k] = params[k]
            c.first_particle, c.num_vertices, c.num_tets = info["offset"], info["num_vertices"], info["num_tets"]
            tets = np.ascontiguousarray(info["tets"], dtype=np.uint32)
            self.keep.append(tets)
            c.tets = tets.ctypes.data_as(C.POINTER(C.c_uint32))
            for k in range(3):
                c.initial_x[k] = info["initial_x"][k]
            for k in range(9):
                c.initial_R[k] = info["initial_R"][k]
            c.restitution, c.friction, c.test_mesh, c.body_index = 0.6, friction, 1, tm
            for which, field in ((0, "points"), (1, "tets_bvh"), (2, "tets_rest")):
                b = ref.bvh(co, which)
                lst = np.ascontiguousarray(b["lst"], dtype=np.uint32)
                nodes = np.ascontiguousarray(b["nodes"], dtype=np.int32)
                hulls = np.ascontiguousarray(b["hulls"], dtype=np.float32)
                self.keep += [lst, nodes, hulls]
                f = getattr(c, field)
                f.num_nodes, f.num_entities = len(nodes), len(lst)
                f.entities = lst.ctypes.data_as(C.POINTER(C.c_uint32))
                f.nodes = nodes.ctypes.data_as(C.POINTER(C.c_int32))
                f.hulls = hulls.ctypes.data_as(C.POINTER(C.c_float))


class GoldenTetColliders:
    """the same records from tests/golden/tetcontact_two_bars.npz (made by tests/golden/make_golden.py from the reference)"""

    def __init__(self, g):
        self.keep = []
        self.n = 2
        self.arr = (_ffi.TetCollider * self.n)()
        self.tolerance = float(g["tolerance"])
        for q in range(self.n):
            shape, invert, offset, nv, nt, body = (int(v) for v in g["c%d_meta" % q])
            c = self.arr[q]
            c.shape, c.invert = shape, invert
            for k in range(4):
                c.params[k] = g["c%d_params" % q][k]
            c.first_particle, c.num_vertices, c.num_tets = offset, nv, nt
            tets = np.ascontiguousarray(g["c%d_tets" % q], dtype=np.uint32)
            self.keep.append(tets)
            c.tets = tets.ctypes.data_as(C.POINTER(C.c_uint32))
            for k in range(3):
                c.initial_x[k] = g["c%d_initial_x" % q][k]
            for k in range(9):
                c.initial_R[k] = g["c%d_initial_R" % q].reshape(-1)[k]
            c.restitution, c.friction, c.test_mesh, c.body_index = 0.6, 0.0, 1, body
            for name, field in (("points", "points"), ("tets", "tets_bvh"), ("rest", "tets_rest")):
                lst = np.ascontiguousarray(g["c%d_%s_lst" % (q, name)], dtype=np.uint32)
                nodes = np.ascontiguousarray(g["c%d_%s_nodes" % (q, name)], dtype=np.int32)
                hulls = np.ascontiguousarray(g["c%d_%s_hulls" % (q, name)], dtype=np.float32)
                self.keep += [lst, nodes, hulls]
                f = getattr(c, field)
                f.num_nodes, f.num_entities = len(nodes), len(lst)
                f.entities = lst.ctypes.data_as(C.POINTER(C.c_uint32))
                f.nodes = nodes.ctypes.data_as(C.POINTER(C.c_int32))
                f.hulls = hulls.ctypes.data_as(C.POINTER(C.c_float))


def host_contacts_of_state(x, x0, w, colliders, capacity=4096, v=None, mass=None):
    """pbdx_debug_tet_contacts on a given state (v: velocities or None = at rest; they only feed the tangent / maximal tangent impulse columns)"""
    pos4 = np.ascontiguousarray(np.concatenate([x, w[:, None]], axis=1), dtype=np.float32)
    rest4 = np.ascontiguousarray(np.concatenate([x0, w[:, None]], axis=1), dtype=np.float32)
    vel4 = None
    if v is not None:
        vel4 = np.ascontiguousarray(np.concatenate([v, (mass if mass is not None else np.ones(len(v)))[:, None]], axis=1), dtype=np.float32)
    out = np.zeros((capacity, _ffi.TET_CONTACT_FLOATS), dtype=np.float32)
    count = C.c_uint32(0)
    _ffi.check(_ffi.lib.pbdx_debug_tet_contacts(len(x), pos4.ctypes.data_as(_ffi.pf), rest4.ctypes.data_as(_ffi.pf), vel4.ctypes.data_as(_ffi.pf) if vel4 is not None else None, colliders.n, colliders.arr,
                                                float(colliders.tolerance), capacity, C.byref(count), out.ctypes.data_as(_ffi.pf)), "debug_tet_contacts")
    return out[:count.value]


def host_contacts(ref, colliders, capacity=4096):
    """pbdx_debug_tet_contacts on the oracle's current state: the engine's detection code evaluated on the host."""
    x = ref.positions().astype(np.float32)
    x0 = ref.get_array(1).astype(np.float32)
    w = ref.get_array(7).astype(np.float32)
    pos4 = np.ascontiguousarray(np.concatenate([x, w[:, None]], axis=1), dtype=np.float32)
    rest4 = np.ascontiguousarray(np.concatenate([x0, w[:, None]], axis=1), dtype=np.float32)
    vel4 = np.ascontiguousarray(np.concatenate([ref.get_array(2), ref.get_array(6)[:, None]], axis=1), dtype=np.float32)
    out = np.zeros((capacity, _ffi.TET_CONTACT_FLOATS), dtype=np.float32)
    count = C.c_uint32(0)
    _ffi.check(_ffi.lib.pbdx_debug_tet_contacts(len(x), pos4.ctypes.data_as(_ffi.pf), rest4.ctypes.data_as(_ffi.pf), vel4.ctypes.data_as(_ffi.pf), colliders.n, colliders.arr,
                                                float(colliders.tolerance), capacity, C.byref(count), out.ctypes.data_as(_ffi.pf)), "debug_tet_contacts")
    return out[:count.value]


def oracle_contacts_as_engine_records(ref):
    """The oracle's ParticleTetContactConstraints in the column layout of the engine's 30-float record (vertex ids left out)."""
    c = ref.particle_solid_contacts()
    if not len(c):
        return np.zeros((0, 26), dtype=np.float32)
    out = np.zeros((len(c), 26), dtype=np.float32)
    out[:, 0:3] = c[:, 0:3]
    out[:, 3:6] = c[:, 3:6]
    out[:, 6:9] = c[:, 6:9]          # constraintInfo.col(0) = normal
    out[:, 9] = c[:, 6 + 6]          # constraintInfo(0, 2): column-major -> index 6
    out[:, 10:22] = c[:, 16:28]
    out[:, 22:26] = c[:, 28:32]
    return out


def oracle_contact_velocity_columns(ref):
    """tangent (constraintInfo.col(1)) and maximal tangent impulse (constraintInfo(1, 2)) of the oracle's contacts: columns 30..33 of the engine's record"""
    c = ref.particle_solid_contacts()
    out = np.zeros((len(c), 4), dtype=np.float32)
    if len(c):
        out[:, 0:3] = c[:, 9:12]          # column 1 of the column-major 3x3
        out[:, 3] = c[:, 6 + 7]           # (1, 2): column 2, row 1
    return out


# ---- velocity part of ONE contact: adversarial known-answer inputs ----------------------------------------------------------------------
def velocity_kat_inputs(n=4000, seed=11):
    """26 floats per case (pbdx_debug_tet_velocity_kat): invMass0, v0[3], invMass[4], v[4][3], bary[3], normal[3].  Four families: generic;
    relative velocity along the normal plus a tangential part of 1e-9 .. 1e-3; relative velocity exactly along the normal (what is left
    of the tangent is rounding residue); a particle falling onto a resting tet with a slightly tilted normal.  In the last three the tangent
    stays un-normalised (|t|^2 <= 1e-6) and the maximal tangent impulse u_rel . t comes out NEGATIVE for about half of the cases: the
    branch `frictionCoeff * lambda > pMax` of velocitySolve_ParticleTetContactConstraint that a friction coefficient of 0 does not switch off."""
    rng = np.random.default_rng(seed)
    rows = []
    for trial in range(n):
        nrm = rng.standard_normal(3); nrm /= np.linalg.norm(nrm)
        bary = rng.dirichlet(np.ones(4))[:3]
        w = np.where(rng.random(5) < 0.15, 0.0, rng.uniform(0.5, 2.0, 5))
        vt = rng.standard_normal((4, 3)) * 0.3
        b0 = 1 - bary.sum()
        v1 = b0 * vt[0] + bary[0] * vt[1] + bary[1] * vt[2] + bary[2] * vt[3]
        mode = trial % 4
        if mode == 0:
            v0 = rng.standard_normal(3)
        elif mode == 1:
            v0 = v1 + nrm * rng.uniform(-3, 3) + rng.standard_normal(3) * 10.0 ** rng.uniform(-9, -3)
        elif mode == 2:
            v0 = v1 + nrm * rng.uniform(-3, 3)
        else:
            vt[:] = 0
            v0 = np.array([0, -rng.uniform(0, 2), 0])
            nrm = np.array([1e-4 * rng.standard_normal(), 1.0, 1e-4 * rng.standard_normal()]); nrm /= np.linalg.norm(nrm)
        rows.append(np.concatenate([[w[0]], v0, w[1:], vt.reshape(-1), bary, nrm]))
    return np.asarray(rows, dtype=np.float32)


def engine_velocity_kat(inputs):
    out = np.zeros((len(inputs), 20), dtype=np.float32)
    for i, row in enumerate(np.ascontiguousarray(inputs, dtype=np.float32)):
        _ffi.check(_ffi.lib.pbdx_debug_tet_velocity_kat(row.ctypes.data_as(_ffi.pf), out[i].ctypes.data_as(_ffi.pf)), "tet_velocity_kat")
    return out


def velocity_kat_equal(got, want):
    """tangent and pMax bit for bit; corrections bit for bit up to the sign of a zero (the reference's impulse of a contact WITHOUT the pMax < 0
    branch is (-0 * garbage) * tangent, a zero whose sign belongs to the garbage)"""
    z = lambda a: np.where(a == 0, np.float32(0), a).view(np.uint32)
    return bool(np.array_equal(got[:, :4].view(np.uint32), want[:, :4].view(np.uint32)) and np.array_equal(z(got[:, 5:]), z(want[:, 5:])))


# ---- BASELINE configs[4] in the shape that can be pinned (data/scenes/ArmadilloCollisionScene.json) ---------------------------------------
ARMADILLO_PLACEMENT = [((0, 10, 0), (0, 1, 0), 1.57, 0.1), ((0, 4, 0), (0, 0, 1), 0.1, 0.0), ((0, 18, 0), (0, 0, 1), 0.5, 0.0)]   # translation, axis, angle, restitution
ARMADILLO_BOX = (3.0, 4.0, 2.6)      # analytic stand-in (full side lengths, rest frame) for the scene's Discregrid SDF of armadillo.obj (neither is in the tree)


def armadillo_scene(ref, sub_steps, models=None):
    """The scene file's three armadillo_4k tet models (scale 2, its translations / rotations, FEM tets with stiffness 1 and Poisson ratio 0.2,
    maxIterations 1, maxIterationsV 5, time step 0.01, contact tolerance 0, contact stiffness 100) above its static floor box (100 x 1 x 100,
    restitution 0.6, friction 0), on the reference.  Deviations, all forced: the tet models carry an analytic box in their rest frame instead of
    the cubic SDF of armadillo.obj (Discregrid and the surface mesh are not in the tree), their friction coefficient is 0 instead of 0.3 (the
    reference's friction impulse between deformables reads an unset multiplier).  models: (vertices, tets) per model to feed addTetModel with
    (a box without /root/reference: the fixture holds what the reference's own TetGenLoader + placement produced); None: load the files."""
    ref.reset_all()
    ref.set_num_threads(1)
    ref.set_time_step_size(0.01)
    ref.set_gravity((0, -9.81, 0))
    base = "/root/reference/data/models/armadillo_4k"
    for q, (x, axis, angle, _) in enumerate(ARMADILLO_PLACEMENT):
        if models is None:
            ref.add_tetgen_model(base + ".node", base + ".ele", x, axis, angle, (2, 2, 2))
        else:
            ref.add_tet_model(models[q][0], models[q][1])
            ref.set_tet_model_initial_transform(q, models[q][2], models[q][3])
    for tm in range(3):
        ref.add_solid_constraints(tm, 2, 1.0, 0.2, 1.0, False, False)
    ref.set_collision_tolerance(0.0)
    ref.add_static_collider("box", (0, 0, 0), (1, 0, 0, 0), (100, 1, 100), (100, 1, 100), 0.6, 0.0)
    objs = [ref.add_tet_collision_shape(tm, 0, ARMADILLO_BOX, True, False, ARMADILLO_PLACEMENT[tm][3], 0.0) for tm in range(3)]
    ref.attach_collision_detection()
    ref.set_params(sub_steps, 1, 0)
    ref.set_max_iterations_v(5)
    return objs
