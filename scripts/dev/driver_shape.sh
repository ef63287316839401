for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('20/5 host', d['ms_per_substep'])"
python - <<'PY'
import json
d=json.load(open("bench_detail.json")); print("   ", d["config"].get("substep_device_times"))
PY
done
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import positionbaseddynamics_amd as pbd
from positionbaseddynamics_amd import scenes
import bench
class E: rank = 0; world = 1; hip_device = 0
ops, desc, pins = bench.workload_spec(dict(workload="c2", size=1000, iters=10), E())
model = scenes.build_model(ops); model.initConstraintGroups()
pbd.TimeManager.getCurrent().setTimeStepSize(0.005)
ts = pbd.TimeStepController(device=0)
ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1); ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 10)
sol = ts.solver(); S = pbd.Solver
sol.set_option(S.OPT_SUBSTEP_EVENTS, 64)
ts.stepResident(model, 5)
print("warm-up call, per substep:", ["%.3f" % t for t in sol.substep_times()])
torch.cuda.synchronize(); t0 = time.perf_counter(); ts.stepResident(model, 20); torch.cuda.synchronize(); t = time.perf_counter() - t0
ev = sol.substep_times()
print("timed call host %.3f ms (%.4f per step), events sum %.3f:" % (1e3 * t, 1e3 * t / 20, sum(ev)), ["%.3f" % x for x in ev])
PY
python - <<'PY'
import json
d=json.load(open("bench_detail.json")); print("   engine:", d["config"].get("engine"))
PY
for a in "--workload c3" "--workload c4"; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic --no-roofline $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$a 20/5 host', d['ms_per_substep'], d['config'].get('device_median_ms_per_substep'))"; done
