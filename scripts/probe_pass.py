#!/usr/bin/env python3
"""Developer aid (library built with -DPBDX_PASS_PROBE=1, PBDX_LIB pointing at it): the phases of a pass boundary of the persistent kernel on the
1000x1000 cloth -- wall-clock stamps (100 MHz) of every tile's first thread: tile descriptor arrived, wait for the neighbours started / over, LDS filled,
table staged, first colour step done, sweep done, write-back stores issued, stores acknowledged (vmcnt 0), published; and the gap to the tile's next pass."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import positionbaseddynamics_amd as pbd
from tests import util

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1000)
ap.add_argument("--instances", type=int, default=1, help="an ensemble block of that many sheets (configs[3]: --size 200 --instances 64)")
args = ap.parse_args()
model = util.build_mine(util.cloth_spec(args.size, args.size, 4, 3) if args.instances == 1 else
                        util.cloth_spec(args.size, args.size, 4, 3, instances=args.instances, instance_offset=(0.0, 0.0, 12.0), instanced=True))
ts = pbd.TimeStepController()
ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 10)
sol = ts.solver()
sol.set_option(sol.OPT_PERSISTENT, 2)
ts.stepResident(model, 10)
sol.set_option(sol.OPT_TRACE, 1)
sol.set_option(sol.OPT_USE_GRAPH, 0)
ts.stepResident(model, 1)
plan = sol.plan_info()
print(plan)
trs = [sol.trace(seg).astype(np.int64) for seg in range(plan["num_segments"])]
names = [("entry -> tile descriptor arrived", 0, 40), ("-> wait for the neighbours starts (ids in flight)", 40, 41), ("-> dependencies published (poll over)", 41, 42),
         ("-> LDS filled (copies landed, barrier)", 42, 1), ("-> table staged", 1, 43), ("-> first colour step done", 43, 2), ("-> sweep done", 2, 44),
         ("-> write-back stores issued", 44, 45), ("-> stores acknowledged (vmcnt 0)", 45, 46), ("-> published", 46, 47)]
for seg, tr in enumerate(trs):
    print("segment %d (its last pass of the substep):" % seg)
    for label, a, b in names:
        ok = (tr[:, a] > 0) & (tr[:, b] > 0)
        d = (tr[ok, b] - tr[ok, a]) * 0.01
        if d.size:
            print("   %-52s median %6.2f  p90 %6.2f  max %6.2f us" % (label, np.median(d), np.percentile(d, 90), d.max()))
    ok = (tr[:, 0] > 0) & (tr[:, 47] > 0)
    d = (tr[ok, 47] - tr[ok, 0]) * 0.01
    print("   %-52s median %6.2f us" % ("whole pass (entry -> published)", np.median(d)))
    if seg + 1 < len(trs):
        nx = trs[seg + 1]
        ok = (tr[:, 47] > 0) & (nx[:, 0] > 0)
        d = (nx[ok, 0] - tr[ok, 47]) * 0.01
        print("   %-52s median %6.2f  max %6.2f us" % ("published -> entry of the tile's next pass", np.median(d), d.max()))
