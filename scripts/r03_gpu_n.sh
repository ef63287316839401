#!/bin/bash
# round 3, GPU pass m/n: quad-lane STRAIN tet projection (lane = particle; n: per step, only where a step fits one quad chunk) -- suite + A/B
set -u
O=$PWD/gpurun_out/r03n; mkdir -p $O
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {
  local label="$1"; local lib="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  PBDX_LIB=$lib timeout 200 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
NS=$PWD/gpurun_variants/nostrainquad/libpbdx.so
for rep in 1 2; do
  run "c3 strain tets, quad lanes (rep $rep)" "" --workload c3 --solid-method 4
  run "c3 strain tets, one lane per constraint (rep $rep)" "$NS" --workload c3 --solid-method 4
done
run "c3 strain, quad, tile 64" "" --workload c3 --solid-method 4 --tile 64 --persistent 2 --fuse 1
run "c3 strain, quad, tile 200" "" --workload c3 --solid-method 4 --tile 200 --persistent 2 --fuse 1
run "c3 strain, quad, tile 507" "" --workload c3 --solid-method 4 --tile 507 --persistent 2 --fuse 1
run "c3 strain 32 bars, quad" "" --workload c3 --solid-method 4 --bars --instances 32
run "c3 strain 32 bars, scalar" "$NS" --workload c3 --solid-method 4 --bars --instances 32
timeout 200 python scripts/trace_tiles.py --bar 4 --persistent 2 > $O/trace_bar_strain.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head; cat $O/rc.txt; cat $O/ab.log; sed -n 1,8p $O/trace_bar_strain.log | cut -c1-300
