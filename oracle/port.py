"""oracle/port.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_ref/libpbd_oracle.so (oracle/pbd_oracle.c, the plain-C
restatement of the reference's hot path).  `Port('f32')` / `Port('f64')` expose
the same methods as `oracle.refdrv.Ref`, so the tests can run against either.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_ref", "libpbd_oracle.so")

_pd = C.POINTER(C.c_double)
_pu = C.POINTER(C.c_uint)
_u = C.c_uint
_d = C.c_double
_i = C.c_int

TYPE_BY_NAME = {"distance": 0, "distance_xpbd": 1, "dihedral": 2, "isometric_bending": 3, "isometric_bending_xpbd": 4,
                "fem_triangle": 5, "strain_triangle": 6, "volume": 7, "volume_xpbd": 8, "fem_tet": 9, "fem_tet_xpbd": 10,
                "strain_tet": 11, "shape_matching": 12}
NUM_BODIES = [2, 2, 4, 4, 4, 3, 3, 4, 4, 4, 4, 4, 4]


def available():
    return os.path.exists(LIB)


def build():
    subprocess.check_call(["make", "port"], cwd=_HERE)


def _dp(a):
    return a.ctypes.data_as(_pd)


def _up(a):
    return a.ctypes.data_as(_pu)


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not available():
            build()
        _lib = C.CDLL(LIB)
    return _lib


class Port:
    def __init__(self, variant="f32"):
        assert variant in ("f32", "f64"), "the C port has float and double instantiations only"
        self.variant = variant
        self.lib = _load()
        self.p = "po32_" if variant == "f32" else "po64_"
        self._s = None
        f = self._f
        f("create").restype = C.c_void_p
        f("get_time").restype = _d
        f("constraint_lambda").restype = _d
        for name, args in {
            "destroy": [C.c_void_p], "add_vertex": [C.c_void_p, _pd], "set_mass": [C.c_void_p, _u, _d],
            "add_triangle_model": [C.c_void_p, _u, _u, _pd, _pu], "add_regular_triangle_model": [C.c_void_p, _i, _i, _pd, _pd, _pd],
            "add_tet_model": [C.c_void_p, _u, _u, _pd, _pu], "add_regular_tet_model": [C.c_void_p, _i, _i, _i, _pd, _pd, _pd],
            "add_constraint": [C.c_void_p, _i, _pu, _pd, _pu],
            "add_cloth_constraints": [C.c_void_p, _u, _u, _d, _d, _d, _d, _d, _d, _i, _i],
            "add_bending_constraints": [C.c_void_p, _u, _u, _d], "add_solid_constraints": [C.c_void_p, _u, _u, _d, _d, _d, _i, _i],
            "init_constraint_groups": [C.c_void_p], "solve_position_constraints": [C.c_void_p, _u], "step": [C.c_void_p, _u],
            "num_particles": [C.c_void_p], "num_constraints": [C.c_void_p], "constraint_type": [C.c_void_p, _u],
            "constraint_bodies": [C.c_void_p, _u, _pu], "constraint_params": [C.c_void_p, _u, _pd], "constraint_lambda": [C.c_void_p, _u],
            "num_groups": [C.c_void_p], "group_size": [C.c_void_p, _u], "get_group": [C.c_void_p, _u, _pu],
            "set_params": [C.c_void_p, _u, _u, _i], "set_time_step_size": [C.c_void_p, _d], "set_gravity": [C.c_void_p, _d, _d, _d],
            "get_time": [C.c_void_p], "get_array": [C.c_void_p, _i, _pd], "set_array": [C.c_void_p, _i, _pd],
            "tri_num_edges": [C.c_void_p, _u], "tri_get_edges": [C.c_void_p, _u, _pu],
            "tet_num_edges": [C.c_void_p, _u], "tet_get_edges": [C.c_void_p, _u, _pu],
        }.items():
            f(name).argtypes = args
        self.reset_all()

    def _f(self, name):
        return getattr(self.lib, self.p + name)

    def __del__(self):
        try:
            if self._s:
                self._f("destroy")(self._s)
                self._s = None
        except Exception:
            pass

    # -- lifecycle -----------------------------------------------------------
    def reset_all(self):
        if self._s:
            self._f("destroy")(self._s)
        self._s = C.c_void_p(self._f("create")())

    @property
    def real_size(self):
        return self._f("real_size")()

    def set_num_threads(self, n):
        pass   # the port is a scalar single-thread restatement

    def max_threads(self):
        return 1

    def set_time_step_size(self, h):
        self._f("set_time_step_size")(self._s, float(h))

    def set_gravity(self, g):
        self._f("set_gravity")(self._s, float(g[0]), float(g[1]), float(g[2]))

    def set_params(self, sub_steps, max_iter, vel_method=0):
        self._f("set_params")(self._s, int(sub_steps), int(max_iter), int(vel_method))

    # -- meshes ---------------------------------------------------------------
    def add_regular_triangle_model(self, w, h, T=(0, 0, 0), R=None, scale=(1, 1)):
        T = np.asarray(T, dtype=np.float64)
        R = np.eye(3) if R is None else np.ascontiguousarray(R, dtype=np.float64)
        s = np.asarray(scale, dtype=np.float64)
        return self._f("add_regular_triangle_model")(self._s, int(w), int(h), _dp(T), _dp(R), _dp(s))

    def add_regular_tet_model(self, w, h, d, T=(0, 0, 0), R=None, scale=(1, 1, 1)):
        T = np.asarray(T, dtype=np.float64)
        R = np.eye(3) if R is None else np.ascontiguousarray(R, dtype=np.float64)
        s = np.asarray(scale, dtype=np.float64)
        return self._f("add_regular_tet_model")(self._s, int(w), int(h), int(d), _dp(T), _dp(R), _dp(s))

    def add_triangle_model(self, points, faces):
        p = np.ascontiguousarray(points, dtype=np.float64)
        f = np.ascontiguousarray(faces, dtype=np.uint32)
        return self._f("add_triangle_model")(self._s, len(p), len(f), _dp(p), _up(f))

    def add_tet_model(self, points, tets):
        p = np.ascontiguousarray(points, dtype=np.float64)
        t = np.ascontiguousarray(tets, dtype=np.uint32)
        return self._f("add_tet_model")(self._s, len(p), len(t), _dp(p), _up(t))

    def add_vertex(self, x):
        x = np.asarray(x, dtype=np.float64)
        return self._f("add_vertex")(self._s, _dp(x))

    def set_mass(self, i, m):
        self._f("set_mass")(self._s, int(i), float(m))

    def add_cloth_constraints(self, tm, method, k=1.0, xx=1.0, yy=1.0, xy=1.0, xyP=0.3, yxP=0.3, ns=False, nsh=False):
        self._f("add_cloth_constraints")(self._s, tm, method, k, xx, yy, xy, xyP, yxP, int(ns), int(nsh))

    def add_bending_constraints(self, tm, method, k):
        self._f("add_bending_constraints")(self._s, tm, method, k)

    def add_solid_constraints(self, tm, method, k=1.0, poisson=0.3, kv=1.0, ns=False, nsh=False):
        self._f("add_solid_constraints")(self._s, tm, method, k, poisson, kv, int(ns), int(nsh))

    def add_constraint(self, type_name, bodies, *args):
        t = TYPE_BY_NAME[type_name]
        b = np.asarray(bodies, dtype=np.uint32)
        nc = None
        if type_name == "shape_matching":
            nc = np.asarray(args[0], dtype=np.uint32)
            a = np.asarray([args[1]], dtype=np.float64)
        else:
            a = np.asarray([float(v) for v in args], dtype=np.float64)
        return self._f("add_constraint")(self._s, t, _up(b), _dp(a), None if nc is None else _up(nc))

    # -- state -----------------------------------------------------------------
    def num_particles(self):
        return self._f("num_particles")(self._s)

    def get_array(self, which):
        n = self.num_particles()
        out = np.empty((n, 3) if which < 6 else (n,), dtype=np.float64)
        self._f("get_array")(self._s, which, _dp(out))
        return out

    def set_array(self, which, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        self._f("set_array")(self._s, which, _dp(a))

    def positions(self):
        return self.get_array(0)

    def triangle_model_edges(self, tm):
        n = self._f("tri_num_edges")(self._s, tm)
        out = np.empty((n, 4), dtype=np.uint32)
        self._f("tri_get_edges")(self._s, tm, _up(out))
        return out

    def tet_model_edges(self, tm):
        n = self._f("tet_num_edges")(self._s, tm)
        out = np.empty((n, 2), dtype=np.uint32)
        self._f("tet_get_edges")(self._s, tm, _up(out))
        return out

    def num_constraints(self):
        return self._f("num_constraints")(self._s)

    def constraint_types(self):
        return np.array([self._f("constraint_type")(self._s, i) for i in range(self.num_constraints())], dtype=np.int32)

    def constraint_bodies(self, c):
        out = np.empty(NUM_BODIES[self._f("constraint_type")(self._s, c)], dtype=np.uint32)
        self._f("constraint_bodies")(self._s, c, _up(out))
        return out

    def constraint_params(self, c):
        out = np.empty(32, dtype=np.float64)
        n = self._f("constraint_params")(self._s, c, _dp(out))
        return out[:n].copy()

    def constraint_lambda(self, c):
        return self._f("constraint_lambda")(self._s, c)

    def groups(self):
        res = []
        for g in range(self._f("num_groups")(self._s)):
            out = np.empty(self._f("group_size")(self._s, g), dtype=np.uint32)
            self._f("get_group")(self._s, g, _up(out))
            res.append(out)
        return res

    def step(self, n=1):
        self._f("step")(self._s, int(n))

    def time_steps(self, n=1):
        t0 = time.perf_counter()
        self.step(n)
        return time.perf_counter() - t0

    def solve_position_constraints(self, it=0, grouped=False):
        assert not grouped
        self._f("solve_position_constraints")(self._s, int(it))
