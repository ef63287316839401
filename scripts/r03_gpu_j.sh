#!/bin/bash
# round 3, GPU pass j: several small tiles co-resident per CU for latency-bound scenes (each colour step is one lone wavefront's chain)
set -u
O=$PWD/gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
run() {
  local label="$1"; shift
  echo "== $label" >> $O/sweep.log
  timeout 120 python bench.py "$@" --persistent 2 --fuse 1 --no-cpu-baseline --no-extras --no-roofline --no-traffic --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/sweep.log 2>&1
  python - >> $O/sweep.log 2>&1 <<PY
import json
d = json.load(open('bench_detail.json'))
p = d['config']['plan']; q = d['config']['persistent']; print({k: p[k] for k in ('num_segments', 'num_tiles', 'max_local', 'redundancy')}, {k: q[k] for k in ('grid', 'block', 'lds_bytes')})
PY
}
for m in 2 6 4; do
  run "c3 m$m default" --workload c3 --solid-method $m
  for cfg in "64 2 256" "64 2 512" "46 3 256" "91 2 256" "32 4 256" "46 2 256"; do
    set -- $cfg
    run "c3 m$m tile $1 wgs/CU $2 block $3" --workload c3 --solid-method $m --tile $1 --wgs-per-cu $2 --fuse-block $3
  done
done
run "cloth 200 default" --workload c2 --size 200
run "cloth 200 tile 80 wgs 2 block 256" --workload c2 --size 200 --tile 80 --wgs-per-cu 2 --fuse-block 256
run "cloth 200 tile 80 wgs 2 block 512" --workload c2 --size 200 --tile 80 --wgs-per-cu 2 --fuse-block 512
cat $O/sweep.log
