import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import positionbaseddynamics_amd as pbd
from tests import util
def run(spec, persistent, steps, iters):
    model = util.build_mine(spec)
    pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
    ts = pbd.TimeStepController()
    ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
    ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, iters)
    sol = ts.solver()
    sol.set_option(sol.OPT_FUSE, 1)
    sol.set_option(sol.OPT_PERSISTENT, persistent)
    ts.stepResident(model, steps)
    ts.syncToHost(model)
    return model.getParticles().positions().copy(), sol.plan_info()
inst = int(sys.argv[1]) if len(sys.argv) > 1 else 16
spec = util.cloth_spec(200, 200, 4, 3, instances=inst, instance_offset=(0.0, 0.0, 12.0))
for iters in (1, 2):
    x0, plan = run(spec, 0, 1, iters)
    x1, _ = run(spec, 2, 1, iters)
    d = np.nonzero((x0.view(np.uint32) != x1.view(np.uint32)).any(axis=1))[0]
    print("iters", iters, "tiles", plan["num_tiles"], "differing", len(d))
    if len(d):
        per = 200 * 200
        print("  first ids", d[:24])
        print("  (instance,row,col)", [(int(i // per), int((i % per) // 200), int(i % 200)) for i in d[:24]])
        print("  instances hit", np.unique(d // per)[:20], "rows range", (d % per // 200).min(), (d % per // 200).max())
