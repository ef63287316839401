#!/bin/bash
# round 3, GPU pass l: fewer, longer passes per sweep on the headline cloth (planner's hand-off cost scaled up)
set -u
O=$PWD/gpurun_out/r03l; mkdir -p $O
export TMPDIR=/tmp
run() {
  local label="$1"; shift
  echo "== $label" >> $O/sweep.log
  timeout 200 python bench.py --workload c2 --no-cpu-baseline --no-extras --no-roofline --no-traffic --steps 100 --warmup 30 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $O/sweep.log 2>&1
  python - >> $O/sweep.log 2>&1 <<PY
import json
d = json.load(open('bench_detail.json'))
p = d['config']['plan']; print({k: p[k] for k in ('num_segments', 'num_tiles', 'max_local', 'redundancy')})
PY
}
run "default"
for ns in 8000 15000 30000 60000; do PBDX_PLAN_LAUNCH_NS=$ns run "PBDX_PLAN_LAUNCH_NS=$ns"; done
for sc in 0.7 0.5; do PBDX_PLAN_SLOT_SCALE=$sc run "PBDX_PLAN_SLOT_SCALE=$sc"; done
PBDX_PLAN_SLOT_SCALE=0.5 PBDX_PLAN_LAUNCH_NS=30000 run "slot scale 0.5 + launch 30000"
run "default again"
cat $O/sweep.log
