#!/bin/bash
# round 3, GPU pass i: passes per sweep under the cheaper write-back (segment length sweep on the headline cloth and the c4 block)
set -u
O=$PWD/gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
run() {
  local label="$1"; shift
  echo "== $label" >> $O/sweep.log
  timeout 200 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic --steps 100 --warmup 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $O/sweep.log 2>&1
  python - >> $O/sweep.log 2>&1 <<PY
import json
d = json.load(open('bench_detail.json'))
p = d['config']['plan']; print({k: p[k] for k in ('num_segments', 'num_tiles', 'max_local', 'redundancy')})
PY
}
for seg in 0 6 7 8 10 12 14; do run "c2 max-seg $seg" --workload c2 --max-seg $seg; done
for seg in 0 9 10; do run "c4 max-seg $seg" --workload c4 --max-seg $seg; done
cat $O/sweep.log
