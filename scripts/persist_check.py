#!/usr/bin/env python3
"""Developer aid: persistent schedule (PBDX_OPT_PERSISTENT) against the multi-launch fused schedule --
bit-identical state and time per substep, on a few scenes."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import positionbaseddynamics_amd as pbd
from tests import util

def run(spec, persistent, steps, iters=10, extra=()):
    model = util.build_mine(spec)
    pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
    ts = pbd.TimeStepController()
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
    ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, iters)
    sol = ts.solver()
    sol.set_option(sol.OPT_FUSE, 1)
    sol.set_option(sol.OPT_PERSISTENT, persistent)
    for o, v in extra:
        sol.set_option(o, v)
    ts.stepResident(model, 5)
    t0 = time.perf_counter()
    ts.stepResident(model, steps)
    dt = (time.perf_counter() - t0) / steps
    ts.syncToHost(model)
    p = model.getParticles()
    return p.positions().copy(), p.array(2).copy(), dt, sol.plan_info(), sol.describe()

scenes = {
    "cloth 1000x1000": (util.cloth_spec(1000, 1000, 4, 3), 40),
    "cloth 300x300": (util.cloth_spec(300, 300, 4, 3), 40),
    "cloth 50x50 (PBD distance + bending)": (util.cloth_spec(50, 50, 1, 2), 40),
    "bar 101x21x11 FEM": (util.bar_spec(101, 21, 11, 2), 20),
    "64 cloths 200x200": (util.cloth_spec(200, 200, 4, 3, instances=16, instance_offset=(0.0, 0.0, 12.0)), 20),
}
only = sys.argv[1:] 
for name, (spec, steps) in scenes.items():
    if only and not any(o in name for o in only):
        continue
    x0, v0, t0, plan, d0 = run(spec, 0, steps)
    x1, v1, t1, _, d1 = run(spec, 2, steps)
    x2, v2, t2, _, d2 = run(spec, 3, steps)      # self-test: the first persistent launch is refused, the engine recovers
    rec = np.array_equal(x0.view(np.uint32), x2.view(np.uint32)) and np.array_equal(v0.view(np.uint32), v2.view(np.uint32))
    print("   refused-launch recovery bit-identical: %s  [%s]" % (rec, d2[d2.find("schedule="):][:120]))
    same = np.array_equal(x0.view(np.uint32), x1.view(np.uint32)) and np.array_equal(v0.view(np.uint32), v1.view(np.uint32))
    print("%-40s tiles %4d segs %d  multi-launch %.4f ms  persistent %.4f ms  (%.3fx)  bit-identical: %s" % (
        name, plan["num_tiles"], plan["num_segments"], 1e3 * t0, 1e3 * t1, t0 / t1, same), flush=True)
    if not same:
        d = np.abs(x0 - x1)
        print("   max |dx| = %g, differing particles %d of %d" % (d.max(), int((d.max(axis=1) > 0).sum()), len(x0)))
    print("   ", d1.replace("\n", " | ")[:300])
