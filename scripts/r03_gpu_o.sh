#!/bin/bash
# round 3, GPU pass o: next pass's first ids + chunk descriptor requested before the write-back (PBDX_FILL_AHEAD) -- suite + A/B
set -u
O=$PWD/gpurun_out/r03o; mkdir -p $O
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {
  local label="$1"; local lib="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  PBDX_LIB=$lib timeout 200 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic --steps 100 --warmup 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
NA=$PWD/gpurun_variants/noahead/libpbdx.so
for rep in 1 2 3; do
  run "c2 fill ahead (rep $rep)" "" --workload c2
  run "c2 no fill ahead (rep $rep)" "$NA" --workload c2
done
run "c4 block ahead" "" --workload c4
run "c4 block no ahead" "$NA" --workload c4
run "c3 m2 ahead" "" --workload c3 --solid-method 2 --steps 30 --warmup 5
run "c3 m2 no ahead" "$NA" --workload c3 --solid-method 2 --steps 30 --warmup 5
run "c3 m6 ahead" "" --workload c3 --solid-method 6 --steps 30 --warmup 5
run "c3 m6 no ahead" "$NA" --workload c3 --solid-method 6 --steps 30 --warmup 5
run "cloth 200 ahead" "" --workload c2 --size 200
run "cloth 200 no ahead" "$NA" --workload c2 --size 200
timeout 200 python scripts/trace_tiles.py --persistent 2 > $O/trace_cloth_persistent.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head; cat $O/rc.txt; cat $O/ab.log; sed -n 2,12p $O/trace_cloth_persistent.log | cut -c1-330
