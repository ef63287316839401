#!/usr/bin/env python3
"""Developer aid: per-colour-step timeline of the fused tile kernel on the headline cloth
(PBDX_OPT_TRACE).  Prints, per segment, the median/max over tiles of: LDS fill, every colour
step, write-back, and the spread of tile start/end times."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import positionbaseddynamics_amd as pbd
from tests import util

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1000)
ap.add_argument("--bar", type=int, default=0, help="trace the configs[2] bar (101x21x11 tets) with this solid method instead of the cloth")
ap.add_argument("--max-seg", type=int, default=None)
ap.add_argument("--tile", type=int, default=None)
ap.add_argument("--fuse-block", type=int, default=None)
ap.add_argument("--persistent", type=int, default=0, help="0 one launch per segment, 2 one launch per substep")
ap.add_argument("--graph", type=int, default=0, help="1: keep the hipGraph (kernel-to-kernel dead time as it is in production)")
args = ap.parse_args()
model = util.build_mine(util.bar_spec(101, 21, 11, args.bar) if args.bar else util.cloth_spec(args.size, args.size, 4, 3))
ts = pbd.TimeStepController()
ts.setValueUInt(pbd.TimeStepController.NUM_SUB_STEPS, 1)
ts.setValueUInt(pbd.TimeStepController.MAX_ITERATIONS, 10)
sol = ts.solver()
for v, o in ((args.max_seg, sol.OPT_MAX_SEGMENT_COLOURS), (args.tile, sol.OPT_TILE_PARTICLES), (args.fuse_block, sol.OPT_FUSE_BLOCK)):
    if v is not None:
        sol.set_option(o, v)
sol.set_option(sol.OPT_PERSISTENT, args.persistent)
ts.stepResident(model, 10)
sol.set_option(sol.OPT_TRACE, 1)
sol.set_option(sol.OPT_USE_GRAPH, args.graph)
ts.stepResident(model, 1)
plan = sol.plan_info()
print(plan)
# dead time between the last tile of a launch and the first tile of the next one (the trace holds the last
# launch of every segment: segment s -> s+1 inside the last iteration)
trs = [sol.trace(seg).astype(np.int64) for seg in range(plan["num_segments"])]
for seg in range(plan["num_segments"] - 1):
    print("dead time segment %d -> %d: %.2f us (last tile end -> first tile start); last tile end -> median tile start %.2f us" % (
        seg, seg + 1, (trs[seg + 1][:, 0].min() - trs[seg][:, -1].max()) * 0.01, (np.median(trs[seg + 1][:, 0]) - trs[seg][:, -1].max()) * 0.01))
for seg in range(plan["num_segments"]):
    si = sol.segment_info(seg)
    tr = trs[seg]
    t0 = tr[:, 0].min()
    start = (tr[:, 0] - t0) * 0.01
    end = (tr[:, -1] - t0) * 0.01
    fill = (tr[:, 1] - tr[:, 0]) * 0.01
    print("segment %d colours [%d,%d): tiles %d block %d; tile start spread %.2f us, end median %.2f max %.2f us; fill median %.2f max %.2f us" % (
        seg, si["colour_begin"], si["colour_end"], si["num_tiles"], si["block"], start.max(), np.median(end), end.max(), np.median(fill), fill.max()))
    nsteps = int((tr[:, 2:-1] > 0).sum(axis=1).max())
    prev = tr[:, 1]
    rows = []
    for i in range(nsteps):
        cur = tr[:, 2 + i]
        ok = cur > 0
        d = (cur[ok] - prev[ok]) * 0.01
        rows.append("%2d: med %.2f max %.2f" % (i, np.median(d), d.max()))
        prev = np.where(ok, cur, prev)
    print("   step durations (us): " + " | ".join(rows))
    wb = (tr[:, -1] - prev) * 0.01
    print("   write-back median %.2f max %.2f us" % (np.median(wb), wb.max()))
