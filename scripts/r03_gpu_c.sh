#!/bin/bash
# round 3, third GPU pass: finer tile-size sweep for small scenes (single bar, small cloths)
set -u
O=$PWD/gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
run() {  # label, args...
  local label="$1"; shift
  echo "== $label" >> $O/sweep.log
  timeout 120 python bench.py "$@" --persistent 2 --fuse 1 --no-cpu-baseline --no-extras --no-traffic --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $O/sweep.log 2>&1
  python - >> $O/sweep.log 2>&1 <<PY
import json
d = json.load(open('bench_detail.json'))
p = d['config']['plan']; print({k: p[k] for k in ('num_segments', 'num_tiles', 'max_local', 'slots_per_sweep', 'redundancy')})
PY
}
for m in 2 4 6; do
  for tile in 80 91 100 110 128 145 160 200; do
    run "c3 method $m tile $tile block 512" --workload c3 --solid-method $m --tile $tile --fuse-block 512
  done
done
for m in 2 6; do for seg in 8 12 16; do run "c3 method $m tile 128 max-seg $seg" --workload c3 --solid-method $m --tile 128 --fuse-block 512 --max-seg $seg; done; done
for size in 100 200 300; do
  for tile in 0 100 160 256; do
    for blk in 512 1024; do run "cloth ${size}x${size} tile $tile block $blk" --workload c2 --size $size --tile $tile --fuse-block $blk; done
  done
done
cat $O/sweep.log
