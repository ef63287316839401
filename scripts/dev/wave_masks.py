"""no GPU: wave masks of the N x N cloth's plan (pbdx_debug_plan_wave_sync) and the LDS bank model with the bank-aware order per window / per step.
usage: python scripts/dev/wave_masks.py [N] [block] [tile_stride]"""
import sys, os, time, ctypes as C
from tests import util
import positionbaseddynamics_amd as pbd
from positionbaseddynamics_amd import _ffi
lib = _ffi.lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
block = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kind = sys.argv[4] if len(sys.argv) > 4 else "cloth"
spec = util.cloth_spec(n, n, 4, 3) if kind == "cloth" else util.bar_spec(n, 16, 16, int(kind))
m = util.build_mine(spec)
m.getConstraintGroups()
out = (C.c_uint64 * 8)()
t = time.time()
r = lib.pbdx_debug_plan_wave_sync(m._h, block, stride, 0, out)
o = list(out)
print("rc", r, lib.pbdx_last_error() if r else b"", "words %d bits %d waves %d colour ends %d widest %d empty %d  -> %.2f bits per mask   %.1f s" % (o[0], o[1], o[2], o[3], o[4], o[5], o[1] / max(o[0], 1), time.time() - t))
out = (C.c_uint64 * 8)()
r = lib.pbdx_debug_plan_lds_model(m._h, 1, 0, out)
o = list(out)
print("lds model (window %s): reads %d groups -> %d cycles (x%.3f)  writes %d -> %d (x%.3f)  table %d -> %d (x%.3f)" % (os.environ.get("PBDX_PLAN_BANK_WINDOW", "default"), o[0], o[1], o[1] / max(o[0], 1), o[2], o[3], o[3] / max(o[2], 1), o[4], o[5], o[5] / max(o[4], 1)))
