import sys, time
sys.path.insert(0, '.')
from tests import util
ops = util.cloth_spec(1000, 1000, 4, 3)
for rep in range(2):
    m = util.build_mine(ops)
    t0 = time.perf_counter(); m.initConstraintGroups(device=0); print("total %.3f s" % (time.perf_counter() - t0))
m = util.build_mine(util.bar_spec(101, 21, 11, 2))
t0 = time.perf_counter(); m.initConstraintGroups(device=0); print("bar total %.3f s" % (time.perf_counter() - t0))
