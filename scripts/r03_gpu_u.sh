#!/bin/bash
# round 3, GPU pass u: the first ring of a pass requested during the LDS fill (PBDX_PRIME_IN_FILL) -- suite + A/B + traces
set -u
O=$PWD/gpurun_out/r03u; mkdir -p $O
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {
  local label="$1"; local lib="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  PBDX_LIB=$lib timeout 200 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
NP=$PWD/gpurun_variants/noprime/libpbdx.so
for rep in 1 2 3; do
  run "c2 primed in the fill (rep $rep)" "" --workload c2 --steps 100 --warmup 30
  run "c2 primed by the first run (rep $rep)" "$NP" --workload c2 --steps 100 --warmup 30
done
for m in 2 4 6; do
  for rep in 1 2; do
    run "c3 m$m primed in the fill (rep $rep)" "" --workload c3 --solid-method $m --steps 30 --warmup 5
    run "c3 m$m primed by the first run (rep $rep)" "$NP" --workload c3 --solid-method $m --steps 30 --warmup 5
  done
done
run "c4 primed in the fill" "" --workload c4 --steps 50 --warmup 10
run "c4 primed by the first run" "$NP" --workload c4 --steps 50 --warmup 10
for sz in 100 300; do
  run "cloth $sz primed in the fill" "" --workload c2 --size $sz --steps 100 --warmup 20
  run "cloth $sz primed by the first run" "$NP" --workload c2 --size $sz --steps 100 --warmup 20
done
timeout 200 python scripts/trace_tiles.py --persistent 2 > $O/trace_cloth_persistent.log 2>&1
timeout 200 python scripts/trace_tiles.py --persistent 2 --bar 2 > $O/trace_bar_fem.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head; cat $O/rc.txt; cat $O/ab.log; sed -n 2,16p $O/trace_cloth_persistent.log | cut -c1-330;  sed -n 5,8p $O/trace_bar_fem.log | cut -c1-330
