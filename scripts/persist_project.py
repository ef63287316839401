import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import positionbaseddynamics_amd as pbd
from tests import util
spec = util.cloth_spec(1000, 1000, 4, 3)
for mode in (0, 2, 0, 2):
    model = util.build_mine(spec)
    pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
    ts = pbd.TimeStepController()
    ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 10)
    sol = ts.solver()
    sol.set_option(sol.OPT_FUSE, 1)
    sol.set_option(sol.OPT_PERSISTENT, mode)
    ts.stepResident(model, 3)
    for iters in (10, 24, 48):
        sol.project(0.005, iters)
        t0 = time.perf_counter()
        for _ in range(5):
            sol.project(0.005, iters)
        dt = (time.perf_counter() - t0) / 5
        print("mode %d  project(%d sweeps): %.1f us per sweep" % (mode, iters, 1e6 * dt / iters), flush=True)
