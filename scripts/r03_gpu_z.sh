#!/bin/bash
# round 3, GPU pass z: the dictionary form also in small scenes (PBDX_DICT_ALWAYS=1, developer switch) against the rule (only where 1 024 threads run)
set -u
O=$PWD/gpurun_out/r03z; mkdir -p $O
export TMPDIR=/tmp
run() {
  local label="$1"; local always="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  if [ -n "$always" ]; then export PBDX_DICT_ALWAYS=1; else unset PBDX_DICT_ALWAYS; fi
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
for sz in 100 200 300 360; do
  run "cloth $sz rule" "" --workload c2 --size $sz --steps 100 --warmup 20
  run "cloth $sz dictionary always" "1" --workload c2 --size $sz --steps 100 --warmup 20
done
cat $O/ab.log
