"""positionbaseddynamics_amd -- MI355X-native PBD/XPBD constraint-projection engine.

Host-side mirror of the `pypbd` surface that sits on the hot path of
InteractiveComputerGraphics/PositionBasedDynamics (reference pyPBD/*Module.cpp):
`Simulation`, `SimulationModel`, `ParticleData`, `TimeManager`, `TimeStepController`
with the reference's method names, argument meaning and defaults, so a script
written for `pypbd` cloth / solid scenes runs unchanged apart from the import:

    import positionbaseddynamics_amd as pbd
    sim = pbd.Simulation.getCurrent(); sim.initDefault()
    model = sim.getModel()
    model.addRegularTriangleModel(50, 50, (0, 1, 0), R, (10, 10))
    model.getParticles().setMass(0, 0.0)
    model.addClothConstraints(model.getTriangleModels()[0], 4, 1e5, ...)
    ts = sim.getTimeStep(); ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 10)
    ts.step(model)

Everything below is a thin shim over the C ABI (include/pbdx.h, libpbdx.so);
all simulation arithmetic runs in hand-written HIP kernels on gfx950.  There is
no CPU path: stepping without a GPU raises.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import lib, check, PbdxError, StepStats  # noqa: F401

__all__ = ["Simulation", "SimulationModel", "ParticleData", "TimeManager", "TimeStepController",
           "TriangleModel", "TetModel", "Solver", "ConstraintType", "PbdxError", "device_count", "Logger", "LogLevel"]


def device_count():
    return lib.pbdx_device_count()


def bounds_report(device=0, reset=False):
    """Record of the range checks of a sanitizer-grade debug build (csrc/pbdx_bounds.h; PBDX_LIB=.../libpbdx_bounds.so).  Returns a dict:
    `checked` (is the loaded library such a build), `violations` since the last reset, and the first one (kind, workgroup, thread, index,
    limit, tile).  The product build checks nothing and reports checked = False."""
    out = (C.c_uint32 * 8)()
    check(lib.pbdx_debug_bounds_report(int(device), out, 1 if reset else 0), "pbdx_debug_bounds_report")
    kinds = {1: "gid stream", 2: "particle id", 3: "LDS fill slot", 4: "chunk descriptor", 5: "dictionary table", 6: "LDS gather/scatter slot",
             7: "tile index", 8: "dependency list", 9: "table source", 10: "chunk range"}
    return {"checked": bool(out[7]), "violations": int(out[0]), "kind": kinds.get(int(out[1]), int(out[1])), "workgroup": int(out[2]), "thread": int(out[3]),
            "index": int(out[4]), "limit": int(out[5]), "tile": int(out[6])}


def copy_bandwidth(device=0, nbytes=1 << 30, reps=3):
    """Measured device-to-device float4 copy bandwidth (GB/s, read + written): the practical HBM roof."""
    out = C.c_double(0.0)
    check(lib.pbdx_debug_copy_bandwidth(int(device), int(nbytes), int(reps), C.byref(out)), "pbdx_debug_copy_bandwidth")
    return float(out.value)


def valu_issue_interval(device=0, threads=1024):
    """Measured shader cycles per wave64 v_mul_f32 / v_add_f32 and SIMD with `threads` threads per workgroup, one workgroup per CU."""
    out = C.c_double(0.0)
    check(lib.pbdx_debug_valu_issue(int(device), int(threads), C.byref(out)), "pbdx_debug_valu_issue")
    return float(out.value)


def colour_constraints(num_bodies, body_off, bodies, device=0):
    """pbdx_colour_constraints: the reference's greedy colouring (SimulationModel.cpp:1033-1094) on the device, on raw arrays.
    Returns (group_of, num_groups, rounds)."""
    body_off = np.ascontiguousarray(body_off, dtype=np.uint32)
    bodies = np.ascontiguousarray(bodies, dtype=np.uint32)
    nc = len(body_off) - 1
    group_of = np.zeros(max(nc, 1), dtype=np.uint32)
    ng, rounds = C.c_uint32(0), C.c_uint32(0)
    check(lib.pbdx_colour_constraints(int(device), int(num_bodies), nc, _u(body_off), _u(bodies), _u(group_of), C.byref(ng), C.byref(rounds)),
          "pbdx_colour_constraints")
    return group_of[:nc], ng.value, rounds.value


def colour_constraints_host(num_bodies, body_off, bodies):
    """pbdx_colour_constraints_host: the same colouring on the host (bit masks per body).  Returns (group_of, num_groups)."""
    body_off = np.ascontiguousarray(body_off, dtype=np.uint32)
    bodies = np.ascontiguousarray(bodies, dtype=np.uint32)
    nc = len(body_off) - 1
    group_of = np.zeros(max(nc, 1), dtype=np.uint32)
    ng = C.c_uint32(0)
    check(lib.pbdx_colour_constraints_host(int(num_bodies), nc, _u(body_off), _u(bodies), _u(group_of), C.byref(ng)), "pbdx_colour_constraints_host")
    return group_of[:nc], ng.value


def _f(a):
    return a.ctypes.data_as(_ffi.pf)


def _u(a):
    return a.ctypes.data_as(_ffi.pu)


def _vec(v, n):
    a = np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
    if a.size != n:
        raise ValueError("expected %d values" % n)
    return a


class ConstraintType:
    """pbdx_constraint_type (include/pbdx.h)."""
    DISTANCE, DISTANCE_XPBD, DIHEDRAL, ISOMETRIC_BENDING, ISOMETRIC_BENDING_XPBD, FEM_TRIANGLE, STRAIN_TRIANGLE, \
        VOLUME, VOLUME_XPBD, FEM_TET, FEM_TET_XPBD, STRAIN_TET, SHAPE_MATCHING = range(13)
    COUNT = 13

    @staticmethod
    def name(t):
        return lib.pbdx_type_name(t).decode()

    @staticmethod
    def num_bodies(t):
        return lib.pbdx_type_num_bodies(t)

    @staticmethod
    def param_stride(t):
        return lib.pbdx_type_param_stride(t)

    @staticmethod
    def algorithmic_bytes(t):
        return lib.pbdx_type_algorithmic_bytes(t)


# --------------------------------------------------------------------------
class TimeManager:
    """PBD::TimeManager (Simulation/TimeManager.cpp): current time and step size h (default 0.005)."""
    _current = None

    def __init__(self):
        self._h = np.float32(0.005)
        self._time = np.float32(0.0)

    @staticmethod
    def getCurrent():
        if TimeManager._current is None:
            TimeManager._current = TimeManager()
        return TimeManager._current

    @staticmethod
    def setCurrent(tm):
        TimeManager._current = tm

    @staticmethod
    def hasCurrent():
        return TimeManager._current is not None

    def getTime(self):
        return float(self._time)

    def setTime(self, t):
        self._time = np.float32(t)

    def getTimeStepSize(self):
        return float(self._h)

    def setTimeStepSize(self, h):
        self._h = np.float32(h)


class ParticleData:
    """PBD::ParticleData accessors (Simulation/ParticleData.h:139-260) over the model's host arrays."""

    def __init__(self, model):
        self._m = model

    def _arr(self, which):
        n = self.size()
        out = np.empty((n, 3) if which < 6 else (n,), dtype=np.float32)
        check(lib.pbdx_model_get_array(self._m._h, which, _f(out)), "pbdx_model_get_array")
        return out

    def _set(self, which, i, v):
        a = self._arr(which)
        a[i] = v
        check(lib.pbdx_model_set_array(self._m._h, which, _f(np.ascontiguousarray(a))), "pbdx_model_set_array")

    def size(self):
        return lib.pbdx_model_num_particles(self._m._h)

    getNumberOfParticles = size

    def addVertex(self, x):
        return lib.pbdx_model_add_vertex(self._m._h, _f(_vec(x, 3)))

    def getPosition(self, i):
        return self._arr(0)[i]

    def setPosition(self, i, x):
        self._set(0, i, x)

    def getPosition0(self, i):
        return self._arr(1)[i]

    def setPosition0(self, i, x):
        self._set(1, i, x)

    def getVelocity(self, i):
        return self._arr(2)[i]

    def setVelocity(self, i, v):
        self._set(2, i, v)

    def getAcceleration(self, i):
        return self._arr(3)[i]

    def setAcceleration(self, i, a):
        self._set(3, i, a)

    def getMass(self, i):
        return float(self._arr(6)[i])

    def getInvMass(self, i):
        return float(self._arr(7)[i])

    def setMass(self, i, m):
        check(lib.pbdx_model_set_mass(self._m._h, int(i), float(m)), "pbdx_model_set_mass")

    def getVertices(self):
        """Zero-copy view of the packed position array (pyPBD/ParticleDataModule.cpp:54-58)."""
        n = self.size()
        if n == 0:
            return np.empty((0, 3), dtype=np.float32)
        ptr = lib.pbdx_model_positions_ptr(self._m._h)
        return np.ctypeslib.as_array(ptr, shape=(n, 3))

    def markDirty(self):
        """Tell the engine that the host state was written through a zero-copy view (getVertices)."""
        check(lib.pbdx_model_mark_state_dirty(self._m._h), "pbdx_model_mark_state_dirty")

    # bulk helpers (not in pypbd; used by tests / bench)
    def positions(self):
        return self._arr(0)

    def velocities(self):
        return self._arr(2)

    def array(self, which):
        return self._arr(which)

    def set_array(self, which, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        check(lib.pbdx_model_set_array(self._m._h, which, _f(a)), "pbdx_model_set_array")


class _MeshModel:
    def __init__(self, model, index):
        self._m = model
        self._i = index

    def __index__(self):                 # usable wherever the reference API takes the model (or its index)
        return self._i

    def updateMeshNormals(self, pd):     # rendering only in the reference (TriangleModel::updateMeshNormals)
        pass


class _FaceMeshView:
    """The part of Utilities::IndexedFaceMesh a pypbd script reads (numFaces, numVertices, getFaces, getEdges)."""

    def __init__(self, tm):
        self._tm = tm

    def numVertices(self):
        return lib.pbdx_model_triangle_model_num_vertices(self._tm._m._h, self._tm._i)

    def numFaces(self):
        return lib.pbdx_model_triangle_model_num_faces(self._tm._m._h, self._tm._i)

    def numEdges(self):
        return lib.pbdx_model_triangle_model_num_edges(self._tm._m._h, self._tm._i)

    def getFaces(self):
        out = np.empty(3 * self.numFaces(), dtype=np.uint32)
        check(lib.pbdx_model_triangle_model_get_faces(self._tm._m._h, self._tm._i, _u(out)), "get_faces")
        return out

    def getEdges(self):
        return self._tm.getEdges()


class _TetMeshView:
    def __init__(self, tm):
        self._tm = tm

    def numVertices(self):
        return lib.pbdx_model_tet_model_num_vertices(self._tm._m._h, self._tm._i)

    def numTets(self):
        return lib.pbdx_model_tet_model_num_tets(self._tm._m._h, self._tm._i)

    def numEdges(self):
        return lib.pbdx_model_tet_model_num_edges(self._tm._m._h, self._tm._i)

    def getTets(self):
        out = np.empty(4 * self.numTets(), dtype=np.uint32)
        check(lib.pbdx_model_tet_model_get_tets(self._tm._m._h, self._tm._i, _u(out)), "get_tets")
        return out


class TriangleModel(_MeshModel):
    def getIndexOffset(self):
        return lib.pbdx_model_triangle_model_index_offset(self._m._h, self._i)

    def getParticleMesh(self):
        return _FaceMeshView(self)

    def getEdges(self):
        n = lib.pbdx_model_triangle_model_num_edges(self._m._h, self._i)
        out = np.empty((n, 4), dtype=np.uint32)
        check(lib.pbdx_model_triangle_model_get_edges(self._m._h, self._i, _u(out)), "get_edges")
        return out


class TetModel(_MeshModel):
    def getIndexOffset(self):
        return lib.pbdx_model_tet_model_index_offset(self._m._h, self._i)

    def getParticleMesh(self):
        return _TetMeshView(self)

    def getEdges(self):
        n = lib.pbdx_model_tet_model_num_edges(self._m._h, self._i)
        out = np.empty((n, 2), dtype=np.uint32)
        check(lib.pbdx_model_tet_model_get_edges(self._m._h, self._i, _u(out)), "get_edges")
        return out


class LogLevel:
    DEBUG, INFO, WARN, ERR = range(4)


class Logger:
    """pypbd.Logger stand-in: the engine reports through status codes / pbdx_last_error()."""

    @staticmethod
    def addConsoleSink(level):
        pass

    @staticmethod
    def addFileSink(level, path):
        pass


def _idx(tm):
    return tm._i if isinstance(tm, _MeshModel) else int(tm)


class SimulationModel:
    """PBD::SimulationModel for particle scenes (pyPBD/SimulationModelModule.cpp:90-383)."""

    def __init__(self):
        h = C.c_void_p()
        check(lib.pbdx_model_create(C.byref(h)), "pbdx_model_create")
        self._h = h
        self._pd = ParticleData(self)

    def __del__(self):
        try:
            if self._h:
                lib.pbdx_model_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def init(self):
        pass

    def reset(self):
        check(lib.pbdx_model_reset(self._h), "pbdx_model_reset")

    def cleanup(self):
        check(lib.pbdx_model_cleanup(self._h), "pbdx_model_cleanup")

    def getParticles(self):
        return self._pd

    def getTriangleModels(self):
        return [TriangleModel(self, i) for i in range(lib.pbdx_model_num_triangle_models(self._h))]

    def getTetModels(self):
        return [TetModel(self, i) for i in range(lib.pbdx_model_num_tet_models(self._h))]

    # -- meshes --
    def addRegularTriangleModel(self, width, height, translation=(0, 0, 0), rotation=None, scale=(1, 1), testMesh=False):
        R = np.eye(3, dtype=np.float32) if rotation is None else np.ascontiguousarray(rotation, dtype=np.float32)
        r = lib.pbdx_model_add_regular_triangle_model(self._h, int(width), int(height), _f(_vec(translation, 3)), _f(_vec(R, 9)), _f(_vec(scale, 2)))
        if r < 0:
            raise PbdxError(r, "addRegularTriangleModel")
        return TriangleModel(self, r)

    def addTriangleModel(self, points, indices, testMesh=False):
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        f = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        r = lib.pbdx_model_add_triangle_model(self._h, len(p), len(f), _f(p), _u(f))
        if r < 0:
            raise PbdxError(r, "addTriangleModel")
        return TriangleModel(self, r)

    def addRegularTetModel(self, width, height, depth, translation=(0, 0, 0), rotation=None, scale=(1, 1, 1), testMesh=False):
        R = np.eye(3, dtype=np.float32) if rotation is None else np.ascontiguousarray(rotation, dtype=np.float32)
        r = lib.pbdx_model_add_regular_tet_model(self._h, int(width), int(height), int(depth), _f(_vec(translation, 3)), _f(_vec(R, 9)), _f(_vec(scale, 3)))
        if r < 0:
            raise PbdxError(r, "addRegularTetModel")
        return TetModel(self, r)

    def addTetModel(self, points, indices, testMesh=False):
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 4)
        r = lib.pbdx_model_add_tet_model(self._h, len(p), len(t), _f(p), _u(t))
        if r < 0:
            raise PbdxError(r, "addTetModel")
        return TetModel(self, r)

    def addInstances(self, offsets):
        """Append len(offsets) congruent copies of everything the model holds (an extension over pypbd for the ensemble
        workloads): copy k is what calling the same builders again with translation T + offsets[k] would build --
        same particle / constraint / group numbering, same colouring -- stored once.  The model is sealed afterwards."""
        o = np.ascontiguousarray(offsets, dtype=np.float32).reshape(-1, 3)
        check(lib.pbdx_model_add_instances(self._h, len(o), _f(o)), "add_instances")

    def numInstances(self):
        return lib.pbdx_model_num_instances(self._h)

    # -- per-constraint builders (return bool like the reference) --
    def addDistanceConstraint(self, p1, p2, stiffness):
        return bool(lib.pbdx_model_add_distance_constraint(self._h, p1, p2, stiffness))

    def addDistanceConstraint_XPBD(self, p1, p2, stiffness):
        return bool(lib.pbdx_model_add_distance_constraint_xpbd(self._h, p1, p2, stiffness))

    def addDihedralConstraint(self, p1, p2, p3, p4, stiffness):
        return bool(lib.pbdx_model_add_dihedral_constraint(self._h, p1, p2, p3, p4, stiffness))

    def addIsometricBendingConstraint(self, p1, p2, p3, p4, stiffness):
        return bool(lib.pbdx_model_add_isometric_bending_constraint(self._h, p1, p2, p3, p4, stiffness))

    def addIsometricBendingConstraint_XPBD(self, p1, p2, p3, p4, stiffness):
        return bool(lib.pbdx_model_add_isometric_bending_constraint_xpbd(self._h, p1, p2, p3, p4, stiffness))

    def addFEMTriangleConstraint(self, p1, p2, p3, xx, yy, xy, xyPoisson, yxPoisson):
        return bool(lib.pbdx_model_add_fem_triangle_constraint(self._h, p1, p2, p3, xx, yy, xy, xyPoisson, yxPoisson))

    def addStrainTriangleConstraint(self, p1, p2, p3, xx, yy, xy, normalizeStretch, normalizeShear):
        return bool(lib.pbdx_model_add_strain_triangle_constraint(self._h, p1, p2, p3, xx, yy, xy, int(normalizeStretch), int(normalizeShear)))

    def addVolumeConstraint(self, p1, p2, p3, p4, stiffness):
        return bool(lib.pbdx_model_add_volume_constraint(self._h, p1, p2, p3, p4, stiffness))

    def addVolumeConstraint_XPBD(self, p1, p2, p3, p4, stiffness):
        return bool(lib.pbdx_model_add_volume_constraint_xpbd(self._h, p1, p2, p3, p4, stiffness))

    def addFEMTetConstraint(self, p1, p2, p3, p4, stiffness, poissonRatio):
        return bool(lib.pbdx_model_add_fem_tet_constraint(self._h, p1, p2, p3, p4, stiffness, poissonRatio))

    def addFEMTetConstraint_XPBD(self, p1, p2, p3, p4, stiffness, poissonRatio):
        return bool(lib.pbdx_model_add_fem_tet_constraint_xpbd(self._h, p1, p2, p3, p4, stiffness, poissonRatio))

    def addStrainTetConstraint(self, p1, p2, p3, p4, stretch, shear, normalizeStretch, normalizeShear):
        return bool(lib.pbdx_model_add_strain_tet_constraint(self._h, p1, p2, p3, p4, stretch, shear, int(normalizeStretch), int(normalizeShear)))

    def addShapeMatchingConstraint(self, numberOfParticles, particleIndices, numClusters, stiffness):
        p = np.ascontiguousarray(particleIndices, dtype=np.uint32)
        nc = np.ascontiguousarray(numClusters, dtype=np.uint32)
        return bool(lib.pbdx_model_add_shape_matching_constraint(self._h, int(numberOfParticles), _u(p), _u(nc), stiffness))

    # -- bulk builders --
    def addClothConstraints(self, tm, clothMethod, distanceStiffness=1.0, xxStiffness=1.0, yyStiffness=1.0, xyStiffness=1.0,
                            xyPoissonRatio=0.3, yxPoissonRatio=0.3, normalizeStretch=False, normalizeShear=False):
        check(lib.pbdx_model_add_cloth_constraints(self._h, _idx(tm), clothMethod, distanceStiffness, xxStiffness, yyStiffness, xyStiffness,
                                                   xyPoissonRatio, yxPoissonRatio, int(normalizeStretch), int(normalizeShear)), "addClothConstraints")

    def addBendingConstraints(self, tm, bendingMethod, stiffness):
        check(lib.pbdx_model_add_bending_constraints(self._h, _idx(tm), bendingMethod, stiffness), "addBendingConstraints")

    def addSolidConstraints(self, tm, solidMethod, stiffness=1.0, poissonRatio=0.3, volumeStiffness=1.0,
                            normalizeStretch=False, normalizeShear=False):
        check(lib.pbdx_model_add_solid_constraints(self._h, _idx(tm), solidMethod, stiffness, poissonRatio, volumeStiffness,
                                                   int(normalizeStretch), int(normalizeShear)), "addSolidConstraints")

    # -- constraints / colouring --
    def numConstraints(self):
        return lib.pbdx_model_num_constraints(self._h)

    def constraintTypes(self):
        n = self.numConstraints()
        return np.array([lib.pbdx_model_constraint_type(self._h, i) for i in range(n)], dtype=np.int32)

    def constraintBodies(self, c):
        nb = ConstraintType.num_bodies(lib.pbdx_model_constraint_type(self._h, c))
        out = np.empty(nb, dtype=np.uint32)
        check(lib.pbdx_model_constraint_bodies(self._h, c, _u(out)), "constraint_bodies")
        return out

    def constraintParams(self, c):
        ns = ConstraintType.param_stride(lib.pbdx_model_constraint_type(self._h, c))
        out = np.empty(ns, dtype=np.float32)
        check(lib.pbdx_model_constraint_params(self._h, c, _f(out)), "constraint_params")
        return out

    def setConstraintParams(self, c, params):
        p = np.ascontiguousarray(params, dtype=np.float32)
        check(lib.pbdx_model_set_constraint_params(self._h, c, _f(p)), "set_constraint_params")

    def planCheck(self, tile_particles=0, lds_particles=0, max_segment_colours=0):
        """Plan the colour-fused tile schedule on the host and prove it equivalent to the
        colour-sequential sweep by symbolic execution (no GPU needed).  Returns the plan summary."""
        pi = _ffi.PlanInfo()
        check(lib.pbdx_model_plan_check(self._h, int(tile_particles), int(lds_particles), int(max_segment_colours), C.byref(pi)),
              "pbdx_model_plan_check")
        return {k: getattr(pi, k) for k, _ in _ffi.PlanInfo._fields_}

    def initConstraintGroups(self, device=None):
        """SimulationModel::initConstraintGroups.  device=None: on the host; device=k: the same colouring computed on GPU k
        (group for group identical; raises for what the device form does not take -- call again without `device` then)."""
        if device is None:
            check(lib.pbdx_model_init_constraint_groups(self._h), "initConstraintGroups")
        else:
            check(lib.pbdx_model_init_constraint_groups_device(self._h, int(device)), "initConstraintGroups(device)")

    def getConstraintGroups(self):
        self.initConstraintGroups()
        res = []
        for g in range(lib.pbdx_model_num_groups(self._h)):
            out = np.empty(lib.pbdx_model_group_size(self._h, g), dtype=np.uint32)
            check(lib.pbdx_model_get_group(self._h, g, _u(out)), "get_group")
            res.append(out)
        return res


class TimeStepController:
    """PBD::TimeStepController (Simulation/TimeStepController.cpp) running on the GPU engine.

    Parameter ids are the class attributes below; `setValueUInt/Int`, `getValueUInt/Int`
    follow GenParam::ParameterObject as exposed by pyPBD/ParameterObjectModule.cpp:17-30.
    """
    NUM_SUB_STEPS = 0
    MAX_ITERATIONS = 1
    MAX_ITERATIONS_V = 2
    VELOCITY_UPDATE_METHOD = 3
    ENUM_VUPDATE_FIRST_ORDER = 0
    ENUM_VUPDATE_SECOND_ORDER = 1

    def __init__(self, device=0):
        h = C.c_void_p()
        check(lib.pbdx_timestep_create(C.byref(h), int(device)), "pbdx_timestep_create")
        self._h = h
        self._sim = None

    def __del__(self):
        try:
            if self._h:
                lib.pbdx_timestep_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def init(self):
        pass

    def reset(self):
        check(lib.pbdx_timestep_reset(self._h), "pbdx_timestep_reset")

    def setValueUInt(self, pid, v):
        check(lib.pbdx_timestep_set_param(self._h, pid, int(v)), "setValue")

    setValueInt = setValueUInt

    def getValueUInt(self, pid):
        return int(lib.pbdx_timestep_get_param(self._h, pid))

    getValueInt = getValueUInt

    def _sync_globals(self):
        tm = TimeManager.getCurrent()
        check(lib.pbdx_timestep_set_time_step_size(self._h, float(tm._h)), "set_time_step_size")
        g = Simulation.getCurrent()._gravity if Simulation.hasCurrent() else np.array([0, -9.81, 0], dtype=np.float32)
        check(lib.pbdx_timestep_set_gravity(self._h, _f(_vec(g, 3))), "set_gravity")

    def step(self, model):
        """TimeStep::step(model): host state in, one full time step on the GPU, host state out."""
        self._sync_globals()
        check(lib.pbdx_timestep_step(self._h, model._h), "TimeStepController.step")
        tm = TimeManager.getCurrent()
        tm._time = np.float32(tm._time + tm._h)

    def stepResident(self, model, numSteps=1):
        """Device-resident stepping: no host transfers until syncToHost()."""
        self._sync_globals()
        check(lib.pbdx_timestep_step_resident(self._h, model._h, int(numSteps)), "TimeStepController.stepResident")
        tm = TimeManager.getCurrent()
        for _ in range(int(numSteps)):
            tm._time = np.float32(tm._time + tm._h)

    def syncToHost(self, model):
        check(lib.pbdx_timestep_sync_to_host(self._h, model._h), "syncToHost")

    def syncFromHost(self, model):
        check(lib.pbdx_timestep_sync_from_host(self._h, model._h), "TimeStepController.syncFromHost")

    def invalidate(self):
        check(lib.pbdx_timestep_invalidate(self._h), "invalidate")

    def project(self, model, iterations=1):
        """Projection loop only (no integrate / velocity update): known-answer tests."""
        self._sync_globals()
        check(lib.pbdx_timestep_project(self._h, model._h, int(iterations)), "TimeStepController.project")

    def solver(self):
        h = lib.pbdx_timestep_solver(self._h)
        if not h:
            raise PbdxError(2, "TimeStepController.solver")      # PBDX_ERR_NO_DEVICE: no CPU path
        return Solver(handle=h, owner=self)


class DeviceEnsemble:
    """pbdx_ensemble_*: independent instances of one model over several HIP devices of THIS process -- contiguous blocks of instances, one
    engine per entry of `devices` (a device may be listed twice), all devices stepping at once, no exchange on the data path.  The multi-process form
    (one rank per GPU, RCCL for the bookkeeping) is positionbaseddynamics_amd.ensemble / bench.py --gpus N."""

    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        check(lib.pbdx_ensemble_create(C.byref(h), devs, len(devices)), "pbdx_ensemble_create")
        self._h = h
        self._model = None

    def __del__(self):
        try:
            if self._h:
                lib.pbdx_ensemble_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def setValueUInt(self, pid, v):
        check(lib.pbdx_ensemble_set_param(self._h, pid, int(v)), "pbdx_ensemble_set_param")

    def setModel(self, model):
        tm = TimeManager.getCurrent()
        check(lib.pbdx_ensemble_set_time_step_size(self._h, float(tm._h)), "pbdx_ensemble_set_time_step_size")
        g = Simulation.getCurrent()._gravity if Simulation.hasCurrent() else np.array([0, -9.81, 0], dtype=np.float32)
        check(lib.pbdx_ensemble_set_gravity(self._h, _f(_vec(g, 3))), "pbdx_ensemble_set_gravity")
        check(lib.pbdx_ensemble_set_model(self._h, model._h), "pbdx_ensemble_set_model")
        self._model = model

    def step(self, numSteps=1):
        check(lib.pbdx_ensemble_step(self._h, int(numSteps)), "pbdx_ensemble_step")

    def gather(self):
        check(lib.pbdx_ensemble_gather(self._h, self._model._h), "pbdx_ensemble_gather")

    def numShards(self):
        return int(lib.pbdx_ensemble_num_shards(self._h))

    def shard(self, i):
        dev, b, e, ms = C.c_int(0), C.c_uint64(0), C.c_uint64(0), C.c_double(0.0)
        check(lib.pbdx_ensemble_get_shard(self._h, int(i), C.byref(dev), C.byref(b), C.byref(e), C.byref(ms)), "pbdx_ensemble_get_shard")
        return {"device": int(dev.value), "begin": int(b.value), "end": int(e.value), "last_step_ms": float(ms.value)}

    def shardSolver(self, i):
        ts = lib.pbdx_ensemble_timestep(self._h, int(i))
        h = lib.pbdx_timestep_solver(ts) if ts else None
        if not h:
            raise PbdxError(2, "DeviceEnsemble.shardSolver")
        return Solver(handle=h, owner=self)

    def lastStepMs(self):
        return float(lib.pbdx_ensemble_last_step_ms(self._h))


class Solver:
    """The raw device engine (pbdx_solver_*): what a reference-side TimeStep plug-in binds to."""

    def __init__(self, device=0, handle=None, owner=None):
        self._owner = owner
        if handle is not None:
            self._h = C.c_void_p(handle)
            self._own = False
        else:
            h = C.c_void_p()
            check(lib.pbdx_solver_create(C.byref(h), int(device)), "pbdx_solver_create")
            self._h = h
            self._own = True
        self._n = 0

    def __del__(self):
        try:
            if self._own and self._h:
                lib.pbdx_solver_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_particles(self, x, mass, inv_mass=None, v=None, old_x=None, last_x=None):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3)
        mass = np.ascontiguousarray(mass, dtype=np.float32)
        if inv_mass is None:
            inv_mass = np.where(mass != 0, np.float32(1.0) / np.where(mass != 0, mass, 1).astype(np.float32), np.float32(0)).astype(np.float32)
        inv_mass = np.ascontiguousarray(inv_mass, dtype=np.float32)
        opt = [None if a is None else np.ascontiguousarray(a, dtype=np.float32) for a in (v, old_x, last_x)]
        self._n = len(x)
        check(lib.pbdx_solver_set_particles(self._h, len(x), _f(x), *[None if a is None else _f(a) for a in opt], _f(mass), _f(inv_mass)),
              "pbdx_solver_set_particles")

    def set_positions(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3)
        check(lib.pbdx_solver_set_positions(self._h, len(x), _f(x)), "pbdx_solver_set_positions")

    def get_particles(self, n=None):
        n = self._n if n is None else n
        out = [np.empty((n, 3), dtype=np.float32) for _ in range(4)]
        check(lib.pbdx_solver_get_particles(self._h, n, *[_f(a) for a in out]), "pbdx_solver_get_particles")
        return out

    def begin_schedule(self):
        check(lib.pbdx_solver_begin_schedule(self._h), "begin_schedule")

    def add_batch(self, group, ctype, indices, params):
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        par = np.ascontiguousarray(params, dtype=np.float32)
        nb = ConstraintType.num_bodies(ctype)
        count = idx.size // nb if nb else 0
        check(lib.pbdx_solver_add_batch(self._h, int(group), int(ctype), count, _u(idx), _f(par), par.size // max(count, 1)), "add_batch")

    def end_schedule(self):
        check(lib.pbdx_solver_end_schedule(self._h), "end_schedule")

    def validate_schedule(self):
        check(lib.pbdx_solver_validate_schedule(self._h), "validate_schedule")

    def step(self, h, sub_steps, max_iterations, velocity_update_method=0, gravity=(0, -9.81, 0), num_steps=1):
        check(lib.pbdx_solver_step(self._h, float(h), int(sub_steps), int(max_iterations), int(velocity_update_method),
                                   _f(_vec(gravity, 3)), int(num_steps)), "pbdx_solver_step")

    def project(self, h_sub, iterations):
        check(lib.pbdx_solver_project(self._h, float(h_sub), int(iterations)), "pbdx_solver_project")

    def get_lambdas(self, batch_index, count):
        out = np.empty(count, dtype=np.float32)
        check(lib.pbdx_solver_get_lambdas(self._h, batch_index, count, _f(out)), "get_lambdas")
        return out

    def set_option(self, option, value):
        check(lib.pbdx_solver_set_option(self._h, int(option), int(value)), "set_option")

    def set_profiling(self, on):
        check(lib.pbdx_solver_set_profiling(self._h, int(bool(on))), "set_profiling")

    def stats(self):
        st = StepStats()
        check(lib.pbdx_solver_get_stats(self._h, C.byref(st)), "get_stats")
        return {k: getattr(st, k) for k, _ in StepStats._fields_}

    def substep_times(self):
        """Device milliseconds of every substep of the last step call (set_option(OPT_SUBSTEP_EVENTS, 1) beforehand)."""
        n = C.c_uint32(0)
        check(lib.pbdx_solver_get_substep_times(self._h, None, 0, C.byref(n)), "get_substep_times")
        buf = (C.c_float * max(n.value, 1))()
        check(lib.pbdx_solver_get_substep_times(self._h, buf, n.value, C.byref(n)), "get_substep_times")
        return [float(buf[i]) for i in range(n.value)]

    def type_stats(self, ctype):
        ms = C.c_double()
        launches = C.c_uint64()
        proj = C.c_uint64()
        check(lib.pbdx_solver_get_type_stats(self._h, int(ctype), C.byref(ms), C.byref(launches), C.byref(proj)), "get_type_stats")
        return ms.value, launches.value, proj.value

    def describe(self):
        buf = C.create_string_buffer(512)
        check(lib.pbdx_solver_describe(self._h, buf, 512), "describe")
        return buf.value.decode()

    SHAPES = {"box": 0, "sphere": 1, "torus": 2, "cylinder": 3, "hollow_sphere": 4, "hollow_box": 5}

    def set_colliders(self, colliders):
        """colliders: list of dicts with the fields of pbdx_collider (shape by name or number)."""
        arr = (_ffi.Collider * max(len(colliders), 1))()
        for i, c in enumerate(colliders):
            k = arr[i]
            k.shape = self.SHAPES.get(c["shape"], c["shape"]) if isinstance(c["shape"], str) else int(c["shape"])
            k.invert = int(bool(c.get("invert", False)))
            par = list(c["params"]) + [0.0] * (4 - len(c["params"]))
            for j in range(4):
                k.params[j] = par[j]
            for name, n in (("com", 3), ("R", 9), ("v1", 3), ("v2", 3), ("body_v", 3), ("body_omega", 3)):
                vals = np.asarray(c.get(name, np.zeros(n)), dtype=np.float32).reshape(-1)
                for j in range(n):
                    getattr(k, name)[j] = vals[j]
            k.restitution = c.get("restitution", 0.6)
            k.friction = c.get("friction", 0.2)
            k.body_index = int(c.get("body_index", i))
        check(lib.pbdx_solver_set_colliders(self._h, len(colliders), arr), "set_colliders")

    def set_collider_dynamics(self, dynamics):
        """dynamics: per collider (inv_mass, inertia_inv_w 3x3, object_index), or [] (every body static): pbdx_solver_set_collider_dynamics."""
        arr = (_ffi.ColliderDynamics * max(len(dynamics), 1))()
        for i, (inv_mass, ji, obj) in enumerate(dynamics):
            arr[i].inv_mass = float(inv_mass)
            vals = np.asarray(ji, dtype=np.float32).reshape(-1)
            for j in range(9):
                arr[i].inertia_inv_w[j] = vals[j]
            arr[i].object_index = int(obj)
        check(lib.pbdx_solver_set_collider_dynamics(self._h, len(dynamics), arr), "set_collider_dynamics")

    def set_contact_order(self, range_object_index, rank):
        ro = np.ascontiguousarray(range_object_index, dtype=np.uint32)
        rk = np.ascontiguousarray(rank, dtype=np.uint32)
        check(lib.pbdx_solver_set_contact_order(self._h, len(ro), _u(ro), len(rk), _u(rk)), "set_contact_order")

    def get_body_velocities(self, n):
        v = np.zeros((n, 3), dtype=np.float32)
        w = np.zeros((n, 3), dtype=np.float32)
        check(lib.pbdx_solver_get_body_velocities(self._h, n, _f(v), _f(w)), "get_body_velocities")
        return v, w

    def set_collision_ranges(self, ranges):
        """ranges: list of (first, count, restitution, friction)."""
        arr = (_ffi.CollisionRange * max(len(ranges), 1))()
        for i, r in enumerate(ranges):
            arr[i].first, arr[i].count, arr[i].restitution, arr[i].friction = int(r[0]), int(r[1]), float(r[2]), float(r[3])
        check(lib.pbdx_solver_set_collision_ranges(self._h, len(ranges), arr), "set_collision_ranges")

    def set_contact_params(self, tolerance=0.01, contact_stiffness=100.0, max_iterations_v=5):
        check(lib.pbdx_solver_set_contact_params(self._h, float(tolerance), float(contact_stiffness), int(max_iterations_v)), "set_contact_params")

    def set_rest_positions(self, x0):
        """ParticleData::m_x0 (what findRefTetAt of the deformable contacts reads); defaults to the first uploaded positions."""
        x0 = np.ascontiguousarray(x0, dtype=np.float32).reshape(-1, 3)
        check(lib.pbdx_solver_set_rest_positions(self._h, len(x0), x0.ctypes.data_as(_ffi.pf)), "set_rest_positions")

    def set_tet_colliders(self, records, count, tolerance):
        """records: ctypes array of _ffi.TetCollider (distance field in the rest frame + the three bounding-sphere hierarchies
        the host application built for the tet model); friction must be 0."""
        check(lib.pbdx_solver_set_tet_colliders(self._h, int(count), records, float(tolerance)), "set_tet_colliders")

    def tet_contacts(self, capacity=1 << 16):
        """Contact list of the last detection between deformable solids: (count, 34) float records (include/pbdx.h)."""
        out = np.zeros((capacity, _ffi.TET_CONTACT_FLOATS), dtype=np.float32)
        n = C.c_uint32()
        check(lib.pbdx_solver_get_tet_contacts(self._h, capacity, C.byref(n), out.ctypes.data_as(_ffi.pf)), "get_tet_contacts")
        return out[:min(n.value, capacity)].copy()

    def num_contacts(self):
        n = C.c_uint32()
        check(lib.pbdx_solver_get_num_contacts(self._h, C.byref(n)), "get_num_contacts")
        return n.value

    def num_tet_contacts(self):
        """number of contacts between deformable solids found by the last detection"""
        c = (C.c_uint32 * 8)()
        check(lib.pbdx_debug_tet_counters(self._h, c), "tet_counters")
        return int(c[0])

    def tet_impulses(self):
        """(contacts of the last detection that carried a velocity impulse, total since the colliders were set)"""
        last, total = C.c_uint32(), C.c_uint64()
        check(lib.pbdx_debug_tet_impulses(self._h, C.byref(last), C.byref(total)), "tet_impulses")
        return last.value, total.value

    def plan_info(self):
        """The colour-fused tile schedule planned for the current constraint schedule."""
        pi = _ffi.PlanInfo()
        check(lib.pbdx_solver_get_plan_info(self._h, C.byref(pi)), "get_plan_info")
        return {k: getattr(pi, k) for k, _ in _ffi.PlanInfo._fields_}

    def segment_info(self, segment):
        si = _ffi.SegmentInfo()
        check(lib.pbdx_solver_get_segment_info(self._h, int(segment), C.byref(si)), "get_segment_info")
        return {k: getattr(si, k) for k, _ in _ffi.SegmentInfo._fields_}

    def persistent_info(self):
        """The one-launch form of the fused schedule (OPT_PERSISTENT): eligibility, use, launch geometry, measurements."""
        pi = _ffi.PersistentInfo()
        check(lib.pbdx_solver_get_persistent_info(self._h, C.byref(pi)), "get_persistent_info")
        return {k: getattr(pi, k) for k, _ in _ffi.PersistentInfo._fields_}

    def trace(self, segment):
        """(num_tiles, stride) array of 100 MHz wall-clock stamps of the last launch of `segment`."""
        si = self.segment_info(segment)
        cap = si["num_tiles"] * 128
        buf = np.zeros(cap, dtype=np.uint64)
        stride = C.c_uint32()
        check(lib.pbdx_solver_get_trace(self._h, int(segment), buf.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(stride)), "get_trace")
        return buf[: si["num_tiles"] * stride.value].reshape(si["num_tiles"], stride.value)

    OPT_TRACE = 9
    OPT_PAIRS = 10
    OPT_PIN_HOST = 11
    OPT_PERSISTENT = 12
    OPT_PERSISTENT_TIMEOUT_MS = 13
    OPT_PERSISTENT_WGS_PER_CU = 14
    OPT_TET_CONTACTS_SERIAL = 15
    OPT_TET_FORCE_IMPULSES = 16
    OPT_SUBSTEP_EVENTS = 17
    OPT_USE_GRAPH = 1
    OPT_BLOCK_SIZE = 2
    OPT_XCD_REMAP = 3
    OPT_FUSE = 4
    OPT_TILE_PARTICLES = 5
    OPT_FUSE_BLOCK = 6
    OPT_MAX_SEGMENT_COLOURS = 7
    OPT_LDS_PARTICLES = 8


class Simulation:
    """PBD::Simulation singleton (Simulation/Simulation.cpp): model + time step + gravity."""
    _current = None
    GRAVITATION = 0

    def __init__(self):
        self._gravity = np.array([0.0, -9.81, 0.0], dtype=np.float32)   # Simulation.cpp:16
        self._model = None
        self._timeStep = None

    @staticmethod
    def getCurrent():
        if Simulation._current is None:
            Simulation._current = Simulation()
        return Simulation._current

    @staticmethod
    def setCurrent(s):
        Simulation._current = s

    @staticmethod
    def hasCurrent():
        return Simulation._current is not None

    def init(self):
        """Simulation::init (Simulation.cpp:50-57): default TimeStepController, h = 0.005."""
        self._timeStep = TimeStepController()
        TimeManager.getCurrent().setTimeStepSize(0.005)

    def initDefault(self):
        """pyPBD/SimulationModule.cpp:26-32: a fresh SimulationModel + default time step."""
        self._model = SimulationModel()
        self.init()

    def reset(self):
        if self._model is not None:
            self._model.reset()
        if self._timeStep is not None:
            self._timeStep.reset()
        TimeManager.getCurrent().setTime(0.0)

    def getModel(self):
        return self._model

    def setModel(self, m):
        self._model = m

    def getTimeStep(self):
        return self._timeStep

    def setTimeStep(self, ts):
        self._timeStep = ts

    def getVecValueFloat(self, pid):
        return self._gravity.copy()

    def setVecValueFloat(self, pid, v):
        self._gravity = np.asarray(v, dtype=np.float32).copy()
