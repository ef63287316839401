"""Developer probe: throughput of the library's host copies (pbdx_hostio.hip) on this host, and the cost of moving a 1 M particle
state to the device and back through the bounce buffer (PBDX_OPT_PIN_HOST 0) and through an engine's mirror (1)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import positionbaseddynamics_amd as pbd
from positionbaseddynamics_amd import _ffi

lib = _ffi.lib
f = lib.pbdx_debug_host_copy
rng = np.random.default_rng(1)
for n in (8 << 20, 12 << 20, 64 << 20):
    a = rng.integers(0, 255, n, dtype=np.uint8); b = np.zeros(n, dtype=np.uint8)
    f(C.c_void_p(b.ctypes.data), C.c_void_p(a.ctypes.data), n)
    t0 = time.perf_counter()
    for _ in range(20):
        f(C.c_void_p(b.ctypes.data), C.c_void_p(a.ctypes.data), n)
    dt = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        C.memmove(b.ctypes.data, a.ctypes.data, n)
    dm = (time.perf_counter() - t0) / 20
    print("host copy of %3d MiB: team %.3f ms (%.1f GB/s), one thread %.3f ms (%.1f GB/s)" % (n >> 20, dt * 1e3, n / dt / 1e9, dm * 1e3, n / dm / 1e9))

n = 1000 * 1000
x = rng.standard_normal((n, 3)).astype(np.float32); v = x * 0.5; mass = np.ones(n, dtype=np.float32)
for mirror in (0, 1):
    sol = pbd.Solver()
    sol.set_option(pbd.Solver.OPT_PIN_HOST, mirror)
    sol.set_particles(x, mass, v=v, old_x=x, last_x=x)
    out = [np.empty((n, 3), dtype=np.float32) for _ in range(4)]
    ups, downs = [], []
    for _ in range(8):
        t0 = time.perf_counter()
        sol.set_particles(x, mass, v=v, old_x=x, last_x=x)
        t1 = time.perf_counter()
        _ffi.check(lib.pbdx_solver_get_particles(sol._h, n, *[o.ctypes.data_as(_ffi.pf) for o in out]), "get")
        t2 = time.perf_counter()
        ups.append(t1 - t0); downs.append(t2 - t1)
    assert np.array_equal(out[0], x) and np.array_equal(out[1], v)
    print("1 M particles, PIN_HOST %d: up (56 MB) %.3f ms, down (48 MB) %.3f ms  (medians of 8; python wrapper included)" % (mirror, 1e3 * np.median(ups), 1e3 * np.median(downs)))
