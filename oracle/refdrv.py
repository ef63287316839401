"""oracle/refdrv.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_ref/libpbdref_{f32,f64,fast}.so: the *unmodified*
reference (InteractiveComputerGraphics/PositionBasedDynamics) compiled from
/root/reference by oracle/Makefile plus the headless driver oracle/ref_driver.cpp.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Each variant is loaded with RTLD_LOCAL into its own namespace so
f32 and f64 can coexist in one process (the reference uses singletons).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")

_u = C.c_uint
_d = C.c_double
_i = C.c_int
_pd = C.POINTER(C.c_double)
_pu = C.POINTER(C.c_uint)


def available(variant="f32"):
    return os.path.exists(os.path.join(REF_DIR, "libpbdref_%s.so" % variant))


def best_timing_variant():
    """The release-like reference build for CPU timing that THIS host can execute: 'v4' (-O3 -march=x86-64-v4, AVX-512) if
    /proc/cpuinfo lists the level's features, else 'fast' (-O3 -march=x86-64-v3), else None.  The reference's own flags
    are -O3 -march=native (CMake/Common.cmake:66); a native build of the build container cannot travel to another host."""
    need = {"avx512f", "avx512bw", "avx512cd", "avx512dq", "avx512vl"}
    flags = set()
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("flags"):
                    flags = set(line.split(":", 1)[1].split())
                    break
    except OSError:
        pass
    if need <= flags and available("v4"):
        return "v4"
    return "fast" if available("fast") else None


def _dp(a):
    return a.ctypes.data_as(_pd)


def _up(a):
    return a.ctypes.data_as(_pu)


class Ref:
    """One loaded build of the reference (variant in {'f32','f64','fast'})."""

    _cache = {}

    def __new__(cls, variant="f32"):
        if variant in cls._cache:
            return cls._cache[variant]
        self = super().__new__(cls)
        path = os.path.join(REF_DIR, "libpbdref_%s.so" % variant)
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle ref` where /root/reference exists)")
        self.variant = variant
        self.lib = lib = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
        lib.refdrv_get_time_step_size.restype = _d
        lib.refdrv_get_time.restype = _d
        lib.refdrv_time_steps.restype = _d
        lib.refdrv_constraint_lambda.restype = _d
        lib.refdrv_set_time_step_size.argtypes = [_d]
        lib.refdrv_set_gravity.argtypes = [_d, _d, _d]
        lib.refdrv_set_params.argtypes = [_u, _u, _i]
        lib.refdrv_set_mass.argtypes = [_u, _d]
        lib.refdrv_add_regular_triangle_model.argtypes = [_i, _i, _pd, _pd, _pd]
        lib.refdrv_add_regular_tet_model.argtypes = [_i, _i, _i, _pd, _pd, _pd]
        lib.refdrv_add_triangle_model.argtypes = [_u, _u, _pd, _pu]
        lib.refdrv_add_tet_model.argtypes = [_u, _u, _pd, _pu]
        lib.refdrv_add_vertex.argtypes = [_pd]
        lib.refdrv_add_cloth_constraints.argtypes = [_u, _u, _d, _d, _d, _d, _d, _d, _i, _i]
        lib.refdrv_add_bending_constraints.argtypes = [_u, _u, _d]
        lib.refdrv_add_solid_constraints.argtypes = [_u, _u, _d, _d, _d, _i, _i]
        lib.refdrv_add_distance_constraint.argtypes = [_u, _u, _d]
        lib.refdrv_add_distance_constraint_xpbd.argtypes = [_u, _u, _d]
        lib.refdrv_add_dihedral_constraint.argtypes = [_u, _u, _u, _u, _d]
        lib.refdrv_add_isometric_bending_constraint.argtypes = [_u, _u, _u, _u, _d]
        lib.refdrv_add_isometric_bending_constraint_xpbd.argtypes = [_u, _u, _u, _u, _d]
        lib.refdrv_add_fem_triangle_constraint.argtypes = [_u, _u, _u, _d, _d, _d, _d, _d]
        lib.refdrv_add_strain_triangle_constraint.argtypes = [_u, _u, _u, _d, _d, _d, _i, _i]
        lib.refdrv_add_volume_constraint.argtypes = [_u, _u, _u, _u, _d]
        lib.refdrv_add_volume_constraint_xpbd.argtypes = [_u, _u, _u, _u, _d]
        lib.refdrv_add_fem_tet_constraint.argtypes = [_u, _u, _u, _u, _d, _d]
        lib.refdrv_add_fem_tet_constraint_xpbd.argtypes = [_u, _u, _u, _u, _d, _d]
        lib.refdrv_add_strain_tet_constraint.argtypes = [_u, _u, _u, _u, _d, _d, _i, _i]
        lib.refdrv_add_shape_matching_constraint.argtypes = [_u, _pu, _pu, _d]
        lib.refdrv_get_array.argtypes = [_i, _pd]
        lib.refdrv_set_array.argtypes = [_i, _pd]
        lib.refdrv_triangle_model_get_edges.argtypes = [_u, _pu]
        lib.refdrv_tet_model_get_edges.argtypes = [_u, _pu]
        lib.refdrv_constraint_bodies.argtypes = [_u, _pu]
        lib.refdrv_constraint_params.argtypes = [_u, _pd]
        lib.refdrv_get_group.argtypes = [_u, _pu]
        lib.refdrv_install_timestep_plugin.argtypes = [C.c_char_p, C.c_char_p]
        lib.refdrv_get_timestep.restype = C.c_void_p
        lib.refdrv_add_static_collider.argtypes = [_i, _pd, _pd, _pd, _pd, _d, _d, _i]
        lib.refdrv_enable_collisions.argtypes = [_d, _d, _d]
        lib.refdrv_get_particle_rigid_body_contact.argtypes = [_u, _pd]
        lib.refdrv_set_max_iterations_v.argtypes = [_u]
        lib.refdrv_get_collision_object.argtypes = [_u, _pd]
        lib.refdrv_contact_stiffness_particle_rigid_body.restype = _d
        cls._cache[variant] = self
        return self

    # -- lifecycle -----------------------------------------------------------
    _tet_boxes = {}

    def reset_all(self):
        self._tet_boxes = {}
        self._reset_all()

    def _reset_all(self):
        self.lib.refdrv_reset_all()

    @property
    def real_size(self):
        return self.lib.refdrv_real_size()

    def set_num_threads(self, n):
        self.lib.refdrv_set_num_threads(int(n))

    def max_threads(self):
        return self.lib.refdrv_max_threads()

    def set_time_step_size(self, h):
        self.lib.refdrv_set_time_step_size(float(h))

    def set_gravity(self, g):
        self.lib.refdrv_set_gravity(float(g[0]), float(g[1]), float(g[2]))

    def set_params(self, sub_steps, max_iter, vel_method=0):
        self.lib.refdrv_set_params(int(sub_steps), int(max_iter), int(vel_method))

    # -- meshes ---------------------------------------------------------------
    def add_regular_triangle_model(self, w, h, T=(0, 0, 0), R=None, scale=(1, 1)):
        T = np.asarray(T, dtype=np.float64)
        R = np.eye(3) if R is None else np.ascontiguousarray(R, dtype=np.float64)
        s = np.asarray(scale, dtype=np.float64)
        return self.lib.refdrv_add_regular_triangle_model(int(w), int(h), _dp(T), _dp(R), _dp(s))

    def add_regular_tet_model(self, w, h, d, T=(0, 0, 0), R=None, scale=(1, 1, 1)):
        T = np.asarray(T, dtype=np.float64)
        R = np.eye(3) if R is None else np.ascontiguousarray(R, dtype=np.float64)
        s = np.asarray(scale, dtype=np.float64)
        return self.lib.refdrv_add_regular_tet_model(int(w), int(h), int(d), _dp(T), _dp(R), _dp(s))

    def add_triangle_model(self, points, faces):
        p = np.ascontiguousarray(points, dtype=np.float64)
        f = np.ascontiguousarray(faces, dtype=np.uint32)
        return self.lib.refdrv_add_triangle_model(len(p), len(f), _dp(p), _up(f))

    def add_tet_model(self, points, tets):
        p = np.ascontiguousarray(points, dtype=np.float64)
        t = np.ascontiguousarray(tets, dtype=np.uint32)
        return self.lib.refdrv_add_tet_model(len(p), len(t), _dp(p), _up(t))

    def add_vertex(self, x):
        x = np.asarray(x, dtype=np.float64)
        return self.lib.refdrv_add_vertex(_dp(x))

    def set_mass(self, i, m):
        self.lib.refdrv_set_mass(int(i), float(m))

    def add_cloth_constraints(self, tm, method, k=1.0, xx=1.0, yy=1.0, xy=1.0, xyP=0.3, yxP=0.3, ns=False, nsh=False):
        self.lib.refdrv_add_cloth_constraints(tm, method, k, xx, yy, xy, xyP, yxP, int(ns), int(nsh))

    def add_bending_constraints(self, tm, method, k):
        self.lib.refdrv_add_bending_constraints(tm, method, k)

    def add_solid_constraints(self, tm, method, k=1.0, poisson=0.3, kv=1.0, ns=False, nsh=False):
        self.lib.refdrv_add_solid_constraints(tm, method, k, poisson, kv, int(ns), int(nsh))

    def add_constraint(self, type_name, bodies, *args):
        """type_name as in the reference's add*: 'distance', 'distance_xpbd', 'dihedral', ..."""
        fn = getattr(self.lib, "refdrv_add_%s_constraint" % type_name if not type_name.endswith("_xpbd")
                     else "refdrv_add_%s_constraint_xpbd" % type_name[:-5])
        if type_name == "shape_matching":
            b = np.asarray(bodies, dtype=np.uint32)
            nc = np.asarray(args[0], dtype=np.uint32)
            return fn(len(b), _up(b), _up(nc), float(args[1]))
        return fn(*[int(b) for b in bodies], *args)

    # -- state -----------------------------------------------------------------
    def num_particles(self):
        return self.lib.refdrv_num_particles()

    def get_array(self, which):
        n = self.num_particles()
        out = np.empty((n, 3) if which < 6 else (n,), dtype=np.float64)
        self.lib.refdrv_get_array(which, _dp(out))
        return out

    def set_array(self, which, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.lib.refdrv_set_array(which, _dp(a))

    def positions(self):
        return self.get_array(0)

    # -- topology ---------------------------------------------------------------
    def triangle_model_edges(self, tm):
        n = self.lib.refdrv_triangle_model_num_edges(tm)
        out = np.empty((n, 4), dtype=np.uint32)
        self.lib.refdrv_triangle_model_get_edges(tm, _up(out))
        return out

    def tet_model_edges(self, tm):
        n = self.lib.refdrv_tet_model_num_edges(tm)
        out = np.empty((n, 2), dtype=np.uint32)
        self.lib.refdrv_tet_model_get_edges(tm, _up(out))
        return out

    # -- constraints --------------------------------------------------------------
    def num_constraints(self):
        return self.lib.refdrv_num_constraints()

    def constraint_types(self):
        n = self.num_constraints()
        return np.array([self.lib.refdrv_constraint_type(i) for i in range(n)], dtype=np.int32)

    def constraint_bodies(self, c):
        nb = self.lib.refdrv_constraint_num_bodies(c)
        out = np.empty(nb, dtype=np.uint32)
        self.lib.refdrv_constraint_bodies(c, _up(out))
        return out

    def constraint_params(self, c):
        out = np.empty(32, dtype=np.float64)
        n = self.lib.refdrv_constraint_params(c, _dp(out))
        return out[:n].copy()

    def constraint_lambda(self, c):
        return self.lib.refdrv_constraint_lambda(c)

    def groups(self):
        ng = self.lib.refdrv_num_groups()
        res = []
        for g in range(ng):
            n = self.lib.refdrv_group_size(g)
            out = np.empty(n, dtype=np.uint32)
            self.lib.refdrv_get_group(g, _up(out))
            res.append(out)
        return res

    # -- stepping -------------------------------------------------------------------
    def step(self, n=1):
        self.lib.refdrv_step(int(n))

    def time_steps(self, n=1):
        return self.lib.refdrv_time_steps(int(n))

    def solve_position_constraints(self, it=0, grouped=False):
        if grouped:
            self.lib.refdrv_solve_position_constraints_grouped(int(it))
        else:
            self.lib.refdrv_solve_position_constraints(int(it))

    def model_reset(self):
        self.lib.refdrv_model_reset()

    # -- static colliders / contacts --------------------------------------------------------
    SHAPES = {"box": 0, "sphere": 1, "torus": 2, "cylinder": 3, "hollow_sphere": 4, "hollow_box": 5}

    def add_static_collider(self, shape, pos, quat, bbox, params, restitution=0.6, friction=0.2, invert=False):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (pos, quat, bbox, list(params) + [0.0] * (4 - len(params)))]
        return self.lib.refdrv_add_static_collider(self.SHAPES[shape], _dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]),
                                                   float(restitution), float(friction), int(bool(invert)))

    def add_dynamic_collider(self, shape, pos, quat, bbox, params, density=100.0, restitution=0.6, friction=0.2, test_mesh=False):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (pos, quat, bbox, list(params) + [0.0] * (4 - len(params)))]
        self.lib.refdrv_add_dynamic_collider.argtypes = [_i, _pd, _pd, _pd, _pd, _d, _d, _d, _i]
        return self.lib.refdrv_add_dynamic_collider(self.SHAPES[shape], _dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]), float(density),
                                                    float(restitution), float(friction), int(bool(test_mesh)))

    def rigid_body_state(self, index):
        """position (3), rotation w x y z (4), velocity (3), angular velocity (3), mass."""
        out = np.zeros(14, dtype=np.float64)
        self.lib.refdrv_get_rigid_body_state.argtypes = [_u, _pd]
        assert self.lib.refdrv_get_rigid_body_state(int(index), _dp(out)) == 0
        return out

    def set_rigid_body_mass(self, index, mass):
        self.lib.refdrv_set_rigid_body_mass.argtypes = [_u, _d]
        self.lib.refdrv_set_rigid_body_mass(int(index), float(mass))

    def set_rigid_body_velocity(self, index, v, omega=(0, 0, 0)):
        a = [np.ascontiguousarray(q, dtype=np.float64) for q in (v, omega)]
        self.lib.refdrv_set_rigid_body_velocity.argtypes = [_u, _pd, _pd]
        self.lib.refdrv_set_rigid_body_velocity(int(index), _dp(a[0]), _dp(a[1]))

    def enable_collisions(self, tolerance=0.05, restitution=0.6, friction=0.1):
        self.lib.refdrv_enable_collisions(float(tolerance), float(restitution), float(friction))

    def set_max_iterations_v(self, n):
        self.lib.refdrv_set_max_iterations_v(int(n))

    def collision_objects(self):
        """(colliders, ranges, tolerance, contact stiffness) in the form the product's raw API takes."""
        colliders, ranges, tol = [], [], 0.01
        for i in range(self.lib.refdrv_num_collision_objects()):
            o = np.zeros(32, dtype=np.float64)
            self.lib.refdrv_get_collision_object(i, _dp(o))
            tol = o[31]
            if o[0] == 0:
                assert o[30] == 0.0, "only static rigid bodies"
                colliders.append(dict(shape=int(o[1]), invert=bool(o[2]), params=list(o[3:7]), com=o[7:10], R=o[10:19], v1=o[19:22], v2=o[22:25],
                                      restitution=o[25], friction=o[26], body_index=int(o[27])))
            else:
                ranges.append((int(o[28]), int(o[29]), o[25], o[26]))
        return colliders, ranges, tol, self.lib.refdrv_contact_stiffness_particle_rigid_body()

    def contacts(self):
        n = self.lib.refdrv_num_particle_rigid_body_contacts()
        out = np.empty((n, 18), dtype=np.float64)
        for i in range(n):
            self.lib.refdrv_get_particle_rigid_body_contact(i, _dp(out[i]))
        return out

    def set_cloth_stiffness(self, k):
        self.lib.refdrv_set_cloth_stiffness.argtypes = [_d]
        self.lib.refdrv_set_cloth_stiffness(float(k))

    def set_cloth_bending_stiffness(self, k):
        self.lib.refdrv_set_cloth_bending_stiffness.argtypes = [_d]
        self.lib.refdrv_set_cloth_bending_stiffness(float(k))

    def set_constraint_stiffness(self, c, k):
        self.lib.refdrv_set_constraint_stiffness.argtypes = [_u, _d]
        self.lib.refdrv_set_constraint_stiffness(int(c), float(k))

    def add_hooked_distance_constraint(self, p1, p2, stiffness):
        """A user subclass of the reference's GenericDistanceConstraint that overrides initConstraintBeforeProjection (oracle/ref_driver.cpp)."""
        self.lib.refdrv_add_hooked_distance_constraint.argtypes = [_u, _u, _d]
        self.lib.refdrv_add_hooked_distance_constraint.restype = C.c_int
        assert self.lib.refdrv_add_hooked_distance_constraint(int(p1), int(p2), float(stiffness)) == 0

    def hooked_calls(self, reset=False):
        self.lib.refdrv_hooked_calls.restype = C.c_uint
        n = int(self.lib.refdrv_hooked_calls())
        if reset:
            self.lib.refdrv_reset_hooked_calls()
        return n

    def add_generic_distance_constraint(self, p1, p2, stiffness):
        """A constraint class outside the engine's scope (Demos/GenericConstraintsDemos/GenericConstraints.cpp): mixed-model tests."""
        self.lib.refdrv_add_generic_distance_constraint.argtypes = [_u, _u, _d]
        self.lib.refdrv_add_generic_distance_constraint.restype = C.c_int
        assert self.lib.refdrv_add_generic_distance_constraint(int(p1), int(p2), float(stiffness)) == 0

    def add_generic_isometric_bending_constraint(self, p1, p2, p3, p4, stiffness):
        self.lib.refdrv_add_generic_isometric_bending_constraint.argtypes = [_u, _u, _u, _u, _d]
        self.lib.refdrv_add_generic_isometric_bending_constraint.restype = C.c_int
        assert self.lib.refdrv_add_generic_isometric_bending_constraint(int(p1), int(p2), int(p3), int(p4), float(stiffness)) == 0

    def model_ptr(self):
        self.lib.refdrv_get_model.restype = C.c_void_p
        return C.c_void_p(self.lib.refdrv_get_model())

    def timestep_ptr(self):
        self.lib.refdrv_get_timestep.restype = C.c_void_p
        return C.c_void_p(self.lib.refdrv_get_timestep())

    # ---- deformable vs deformable contacts (tet models with an analytic box in their rest frame) ----
    def add_tet_collision_box(self, tet_model, box, test_mesh=True, restitution=0.6, friction=0.0):
        self.lib.refdrv_add_tet_collision_box.argtypes = [_u, _pd, _i, _d, _d]
        b = np.ascontiguousarray(box, dtype=np.float64)
        if not hasattr(self, "_tet_boxes"):
            self._tet_boxes = {}
        self._tet_boxes[int(tet_model)] = b.copy()
        return self.lib.refdrv_add_tet_collision_box(int(tet_model), _dp(b), int(bool(test_mesh)), float(restitution), float(friction))

    def add_tet_collision_shape(self, tet_model, shape, params, test_mesh=True, invert=False, restitution=0.6, friction=0.0):
        """shape: 0 box (full side lengths), 1 sphere (radius), 2 torus (radii), 3 cylinder (radius, height), 4 hollow sphere (radius,
        thickness), 5 hollow box (full side lengths, thickness) -- the arguments of DistanceFieldCollisionDetection::addCollision*."""
        self.lib.refdrv_add_tet_collision_shape.argtypes = [_u, _i, _pd, _i, _i, _d, _d]
        p = np.zeros(4, dtype=np.float64)
        p[:len(params)] = params
        return self.lib.refdrv_add_tet_collision_shape(int(tet_model), int(shape), _dp(p), int(bool(test_mesh)), int(bool(invert)), float(restitution), float(friction))

    def collision_object_shape(self, co):
        """(shape id, invert, params[4]) as the collision object STORES them (half extents for boxes, ...): what a binding hands to the engine"""
        self.lib.refdrv_get_collision_object_shape.argtypes = [_u, C.POINTER(C.c_int), _pd]
        inv = C.c_int(0)
        p = np.zeros(4, dtype=np.float64)
        shape = self.lib.refdrv_get_collision_object_shape(int(co), C.byref(inv), _dp(p))
        return shape, inv.value, p

    def set_tet_model_initial_transform(self, tet_model, x, R=None):
        self.lib.refdrv_set_tet_model_initial_transform.argtypes = [_u, _pd, _pd]
        xx = np.ascontiguousarray(x, dtype=np.float64)
        rr = np.ascontiguousarray(np.eye(3) if R is None else R, dtype=np.float64)
        self.lib.refdrv_set_tet_model_initial_transform(int(tet_model), _dp(xx), _dp(rr))

    def tet_model_info(self, tm):
        self.lib.refdrv_tet_model_info.argtypes = [_u, _pd, _pu, _u]
        out = np.zeros(16, dtype=np.float64)
        self.lib.refdrv_tet_model_info(int(tm), _dp(out), None, 0)
        tets = np.zeros(4 * int(out[2]), dtype=np.uint32)
        self.lib.refdrv_tet_model_info(int(tm), _dp(out), _up(tets), len(tets))
        return {"offset": int(out[0]), "num_vertices": int(out[1]), "num_tets": int(out[2]), "initial_x": out[3:6].copy(),
                "initial_R": out[6:15].copy(), "tets": tets, "box": self._tet_boxes.get(int(tm))}

    def set_collision_tolerance(self, t):
        self.lib.refdrv_set_collision_tolerance.argtypes = [_d]
        self.lib.refdrv_set_collision_tolerance(float(t))

    def attach_collision_detection(self):
        self.lib.refdrv_attach_collision_detection()

    def add_tetgen_model(self, node_file, ele_file, x, axis, angle, scale):
        """a tet model loaded by the reference's TetGenLoader and placed like SceneLoaderDemo places the models of a scene file"""
        self.lib.refdrv_add_tetgen_model.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
        self.lib.refdrv_add_tetgen_model.restype = C.c_int
        X, A, S = (np.ascontiguousarray(v, dtype=np.float64) for v in (x, axis, scale))
        r = self.lib.refdrv_add_tetgen_model(str(node_file).encode(), str(ele_file).encode(), _dp(X), _dp(A), float(angle), _dp(S))
        if r < 0:
            raise RuntimeError("reference TetGenLoader could not read %s / %s" % (node_file, ele_file))
        return r

    def kat_tet_contact_velocity(self, inputs, friction=0.0, lam=0.0):
        """the reference's init_ParticleTetContactConstraint + velocitySolve_ParticleTetContactConstraint on one contact (26 inputs -> 20 outputs)"""
        a = np.ascontiguousarray(inputs, dtype=np.float64)
        out = np.zeros(20, dtype=np.float64)
        self.lib.refdrv_kat_tet_contact_velocity.argtypes = [C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_double)]
        self.lib.refdrv_kat_tet_contact_velocity.restype = None
        self.lib.refdrv_kat_tet_contact_velocity(_dp(a), float(friction), float(lam), _dp(out))
        return out

    def collision_detection_only(self):
        self.lib.refdrv_collision_detection_only()

    def num_particle_solid_contacts(self):
        return int(self.lib.refdrv_num_particle_solid_contacts())

    def particle_solid_contacts(self):
        """(n, 33) array: particle, solid, tet, bary[3], constraintInfo[9] (column-major), friction, m_x[12], m_invMasses[4], m_lambda"""
        n = self.lib.refdrv_num_particle_solid_contacts()
        out = np.zeros((n, 33), dtype=np.float64)
        self.lib.refdrv_get_particle_solid_contact.argtypes = [_u, _pd]
        for i in range(n):
            self.lib.refdrv_get_particle_solid_contact(i, _dp(out[i]))
        return out

    def bvh(self, co, which):
        """The reference's own bounding-sphere hierarchy of collision object `co` (which: 0 points, 1 tets, 2 tets at rest):
        dict(lst, nodes (n, 4: child0, child1, begin, count), hulls (n, 4: centre, radius))."""
        f = self.lib.refdrv_get_bvh
        f.argtypes = [_u, _i, _pu, _u, C.POINTER(C.c_int), _pd, _u, _pu]
        f.restype = _u
        ne = C.c_uint(0)
        nn = f(int(co), int(which), None, 0, None, None, 0, C.byref(ne))
        lst = np.zeros(max(ne.value, 1), dtype=np.uint32)
        nodes = np.zeros((max(nn, 1), 4), dtype=np.int32)
        hulls = np.zeros((max(nn, 1), 4), dtype=np.float64)
        f(int(co), int(which), _up(lst), ne.value, nodes.ctypes.data_as(C.POINTER(C.c_int)), _dp(hulls), nn, C.byref(ne))
        return {"lst": lst[:ne.value], "nodes": nodes[:nn], "hulls": hulls[:nn]}

    def install_timestep_plugin(self, path, symbol="pbdx_create_timestep_hip"):
        return self.lib.refdrv_install_timestep_plugin(path.encode(), symbol.encode())


# ---------------------------------------------------------------------------
# Scene definitions restated from the reference demos (deterministic, no RNG).
# ---------------------------------------------------------------------------
def rot_x_half_pi(dtype=np.float64):
    """AngleAxisr(M_PI*0.5, (1,0,0)).matrix() (Demos/ClothDemo/main.cpp:136).

    Eigen evaluates sin/cos of the angle in Real, so the float build has
    cos = float(cos(float(pi/2))) = -4.371139e-08, not 0."""
    a = dtype(np.pi * 0.5)
    c = np.cos(a).astype(dtype) if hasattr(np.cos(a), "astype") else dtype(np.cos(a))
    s = dtype(np.sin(a))
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def build_cloth(sim, n_cols, n_rows, cloth_method, bending_method, cloth_k=None, bending_k=None,
                width=10.0, height=10.0, T=(0, 1, 0), pin=True, R=None):
    """Demos/ClothDemo/main.cpp:132-162 with nCols x nRows particles.

    `sim` is any object with the Ref-like builder API (Ref or the product's
    SimulationModel adapter in tests)."""
    if R is None:
        R = rot_x_half_pi(np.float32 if getattr(sim, "real_size", 4) == 4 else np.float64)
    off = sim.num_particles()
    tm = sim.add_regular_triangle_model(n_cols, n_rows, T, R, (width, height))
    if pin:
        sim.set_mass(off, 0.0)
        sim.set_mass(off + n_rows - 1, 0.0)
    if cloth_k is None:
        cloth_k = 100000.0 if cloth_method == 4 else 1.0
    if bending_k is None:
        bending_k = 100.0 if bending_method == 3 else 0.01
    if cloth_method:
        sim.add_cloth_constraints(tm, cloth_method, cloth_k)
    if bending_method:
        sim.add_bending_constraints(tm, bending_method, bending_k)
    return tm


def build_bar(sim, width, height, depth, solid_method, k=None, kv=None, poisson=0.3,
              T=(5, 0, 0), scale=(10.0, 1.5, 1.5), ns=False, nsh=False):
    """Demos/BarDemo/main.cpp:130-166 with width x height x depth particles."""
    off = sim.num_particles()
    tm = sim.add_regular_tet_model(width, height, depth, T, None, scale)
    for j in range(height):
        for k_ in range(depth):
            sim.set_mass(off + j * depth + k_, 0.0)
    if k is None:
        k = {3: 1000000.0, 6: 100000.0}.get(solid_method, 1.0)
    if kv is None:
        kv = 100000.0 if solid_method == 6 else 1.0
    sim.add_solid_constraints(tm, solid_method, k, poisson, kv, ns, nsh)
    return tm
