#!/bin/bash
# round 3, GPU pass y: dictionary form of the bending records (per-tile tables of distinct records in LDS) -- suite + A/B (PBDX_NO_DICT=1 plans without it)
set -u
O=$PWD/gpurun_out/r03y; mkdir -p $O
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {
  local label="$1"; local nodict="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  if [ -n "$nodict" ]; then export PBDX_NO_DICT=1; else unset PBDX_NO_DICT; fi
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
for rep in 1 2 3; do
  run "c2 dictionary form (rep $rep)" "" --workload c2 --steps 100 --warmup 30
  run "c2 streamed (rep $rep)" "1" --workload c2 --steps 100 --warmup 30
done
run "c4 dictionary form" "" --workload c4 --steps 50 --warmup 10
run "c4 streamed" "1" --workload c4 --steps 50 --warmup 10
run "cloth 500 dictionary form" "" --workload c2 --size 500 --steps 100 --warmup 20
run "cloth 500 streamed" "1" --workload c2 --size 500 --steps 100 --warmup 20
unset PBDX_NO_DICT
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head; cat $O/rc.txt; cat $O/ab.log
