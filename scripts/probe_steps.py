#!/usr/bin/env python3
"""Developer aid (library built with -DPBDX_STEP_PROBE=1, PBDX_LIB pointing at it): where the time of ONE colour step of the configs[2] bar
(default) or of the 1000x1000 cloth of configs[1] (--cloth 1000) goes.
Cycle stamps (s_memtime) of the traced tile's thread 0: A sub-iteration entry,
B projection done and scatter issued, C next chunk descriptor read (LDS round trip; the scatter has landed), D colour barrier passed, E next record fetch issued."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import positionbaseddynamics_amd as pbd
from tests import util

ap = argparse.ArgumentParser()
ap.add_argument("--bar", type=int, default=2)
ap.add_argument("--persistent", type=int, default=2)
ap.add_argument("--cloth", type=int, default=0, help="N: the N x N cloth (XPBD distance + XPBD isometric bending) instead of the bar")
args = ap.parse_args()
model = util.build_mine(util.cloth_spec(args.cloth, args.cloth, 4, 3) if args.cloth else util.bar_spec(101, 21, 11, args.bar))
ts = pbd.TimeStepController()
ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 10)
sol = ts.solver()
sol.set_option(sol.OPT_PERSISTENT, args.persistent)
ts.stepResident(model, 10)
sol.set_option(sol.OPT_TRACE, 1)
sol.set_option(sol.OPT_USE_GRAPH, 0)
ts.stepResident(model, 1)
plan = sol.plan_info()
print(plan)
names = ["A->B record wait, LDS gather, projection, scatter issue", "", "", "B->C descriptor read (+scatter landed)", "C->D colour barrier",
         "D->E next record fetch issue", "E->A' loop"]
for seg in range(plan["num_segments"]):
    tr = sol.trace(seg).astype(np.int64)
    wall = tr[:, 2:12]
    pr = tr[:, 20:76].reshape(tr.shape[0], 8, 7)          # tile, step, stamp (A P0 P1 B C D E)
    ok = (pr[:, :, 0] > 0) & (pr[:, :, 6] > 0) & (pr[:, :, 5] > 0)
    # cycle counter frequency against the 100 MHz wall clock: step 1 .. step 4 ends
    dw = (wall[:, 4] - wall[:, 1]) * 10.0     # ns
    dc = pr[:, 4, 5] - pr[:, 1, 5]
    good = (dw > 0) & (dc > 0)
    ghz = np.median(dc[good] / dw[good]) if good.any() else float("nan")
    print("segment %d: s_memtime runs at %.3f GHz (against s_memrealtime)" % (seg, ghz))
    rows = []
    d = (pr[:, 1:7, 3] - pr[:, 1:7, 0])[ok[:, 1:7]]
    rows.append((names[0], np.median(d), np.percentile(d, 90)))
    for k in range(3, 6):
        d = (pr[:, 1:7, k + 1] - pr[:, 1:7, k])[ok[:, 1:7]]
        rows.append((names[k], np.median(d), np.percentile(d, 90)))
    nxt = (pr[:, 2:8, 0] - pr[:, 1:7, 6])[ok[:, 1:7] & ok[:, 2:8]]
    rows.append((names[6], np.median(nxt), np.percentile(nxt, 90)))
    tot = (pr[:, 2:8, 0] - pr[:, 1:7, 0])[ok[:, 1:7] & ok[:, 2:8]]
    for n, m, p90 in rows:
        print("   %-42s median %7.0f  p90 %7.0f counts  (%.3f us)" % (n, m, p90, m / ghz * 1e-3))
    print("   %-42s median %7.0f counts  (%.3f us)" % ("whole step (A -> A')", np.median(tot), np.median(tot) / ghz * 1e-3))
