#!/usr/bin/env python3
"""Fixed host time of one stepResident(n) call on the 1000x1000 cloth: host clock against the sum of the per-substep device events."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import positionbaseddynamics_amd as pbd
from positionbaseddynamics_amd import scenes
import bench

class E:
    rank = 0; world = 1; hip_device = 0
    def shard(self, total): return 0, total
w = dict(workload="c2", size=1000, iters=10, instances=1, bars=False, solid_method=2, scaling="weak", total_instances=512)
if len(sys.argv) > 1: w["workload"] = sys.argv[1]
if w["workload"] == "c4": w["size"], w["instances"] = 200, 64
try:
    ops, desc, pins = bench.workload_spec(w, E())
except Exception as e:
    print("workload_spec:", e); raise
model = scenes.build_model(ops); model.initConstraintGroups()
pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
ts = pbd.TimeStepController(device=0)
ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1); ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 10)
sol = ts.solver(); S = pbd.Solver
ts.stepResident(model, 5)
def timed(n, label):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ts.stepResident(model, n); torch.cuda.synchronize(); t = time.perf_counter() - t0
    ev = sol.substep_times()
    print("%-44s n=%3d host %.3f ms = %.4f per step; device events sum %s; fixed %s ms" % (label, n, 1e3 * t, 1e3 * t / n,
          ("%.3f" % sum(ev)) if ev else "-", ("%.3f" % (1e3 * t - sum(ev))) if ev else "-"))
for n in (20, 20, 50):
    timed(n, "events off")
sol.set_option(S.OPT_SUBSTEP_EVENTS, 1)
timed(20, "events on, created inside the call"); timed(20, "events on, second call"); timed(50, "events on, 30 more created inside")
sol.set_option(S.OPT_SUBSTEP_EVENTS, 200)
timed(200, "events on, created in advance"); timed(20, "events on"); timed(1, "events on"); timed(1, "events on")
sol.set_option(S.OPT_SUBSTEP_EVENTS, 0)
timed(1, "events off"); timed(1, "events off")
