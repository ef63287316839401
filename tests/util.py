"""Shared test helpers: scene specifications applied identically to the oracle
(the reference itself, oracle/_ref, or the plain-C port) and to the product
(positionbaseddynamics_amd).  A scene is a list of operations so that the same
list drives both sides; all scenes are deterministic."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from positionbaseddynamics_amd.scenes import (GRAVITY, rot_x_half_pi, cloth_spec, bar_spec, delaunay_cloth_spec,  # noqa: E402,F401
                                              delaunay_solid_spec, config5_like_spec, kitchen_sink_spec, build_model)
from oracle.scene_ref import apply_ref  # noqa: E402,F401

build_mine = build_model

def get_oracle(variant="f32"):
    """The strongest oracle available: the reference itself (oracle/_ref) if its library was
    built (it travels to the GPU box), otherwise the plain-C port (f32 / f64 only)."""
    from oracle import refdrv
    if refdrv.available(variant):
        return refdrv.Ref(variant)
    from oracle import port
    return port.Port(variant)


def oracle_run(ops, steps, sub_steps, iters, variant="f32", vel_method=0, h=0.005, gravity=GRAVITY, threads=1):
    o = get_oracle(variant)
    apply_ref(o, ops)
    o.set_num_threads(threads)
    o.set_time_step_size(h)
    o.set_gravity(gravity)
    o.set_params(sub_steps, iters, vel_method)
    o.step(steps)
    return o


def oracle_positions(ops, steps, sub_steps, iters, variant="f32", **kw):
    return oracle_run(ops, steps, sub_steps, iters, variant, **kw).positions()


def mine_run(ops, steps, sub_steps, iters, vel_method=0, h=0.005, gravity=GRAVITY, resident=False, options=None):
    """Step the product (GPU).  Returns (model, timestep)."""
    import positionbaseddynamics_amd as pbd
    m = build_mine(ops)
    sim = pbd.Simulation()
    pbd.Simulation.setCurrent(sim)
    sim.setVecValueFloat(pbd.Simulation.GRAVITATION, gravity)
    pbd.TimeManager.setCurrent(pbd.TimeManager())
    pbd.TimeManager.getCurrent().setTimeStepSize(h)
    ts = pbd.TimeStepController()
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, sub_steps)
    ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, iters)
    ts.setValueInt(pbd.TimeStepController.VELOCITY_UPDATE_METHOD, vel_method)
    if options:
        sol = ts.solver()
        for k, v in options.items():
            sol.set_option(k, v)
    if resident:
        ts.stepResident(m, steps)
        ts.syncToHost(m)
    else:
        for _ in range(steps):
            ts.step(m)
    return m, ts


def max_err(a, b):
    """max per-particle Euclidean distance."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.linalg.norm(a - b, axis=1))) if len(a) else 0.0


def bitwise_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def ulp_diff(a, b):
    """max ULP distance between two float32 arrays (same sign assumed for nonzero)."""
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return int(np.max(np.abs(a - b))) if a.size else 0


# scenes with golden outputs of the reference under tests/golden/ (tests/golden/make_golden.py):
# name -> (ops, sub_steps, iterations, horizons in steps)
GOLDEN_SCENES = {
    "c1_cloth50_pbd_dist_isobend_5it": (cloth_spec(50, 50, 1, 2), 1, 5, [1]),
    "c2_cloth50_xpbd_dist_isobend_10it": (cloth_spec(50, 50, 4, 3), 1, 10, [1, 10, 100]),
    "cloth40_femtri_dihedral_5it": (cloth_spec(40, 40, 2, 1), 1, 5, [1, 10]),
    "cloth40_straintri_5it": (cloth_spec(40, 40, 3, 0), 1, 5, [1, 10]),
    "c3_bar30x5x5_femtet_10it": (bar_spec(30, 5, 5, 2), 1, 10, [1, 10]),
    "c3_bar30x5x5_xpbd_distvol_10it": (bar_spec(30, 5, 5, 6), 1, 10, [1, 10]),
    "c3_bar30x5x5_distvol_10it": (bar_spec(30, 5, 5, 1), 1, 10, [10]),
    "c3_bar30x5x5_straintet_10it": (bar_spec(30, 5, 5, 4), 1, 10, [10]),
    "c3_bar30x5x5_shapematching_10it": (bar_spec(30, 5, 5, 5), 1, 10, [10]),
    "c3_bar30x5x5_xpbd_femtet_1it_5sub": (bar_spec(30, 5, 5, 3), 5, 1, [10]),
    "c4_cloth24_x3_instances_xpbd_10it": (cloth_spec(24, 24, 4, 3, instances=3), 1, 10, [5]),
}


