#!/bin/bash
# round 3, second GPU pass: the whole GPU suite again + tile-size / block-size sweep of the single 100 k-tet bar (configs[2])
set -u
O=$PWD/gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
for m in 2 6; do
  for tile in 0 64 128 256 1024; do
    for blk in 256 512; do
      echo "== method $m tile $tile fuse-block $blk" >> $O/c3_sweep.log
      timeout 120 python bench.py --workload c3 --solid-method $m --tile $tile --fuse-block $blk --persistent 2 --fuse 1 --no-cpu-baseline --no-extras --no-traffic --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/c3_sweep.log 2>&1
      python - >> $O/c3_sweep.log 2>&1 <<PY
import json
d = json.load(open('bench_detail.json'))
p = d['config']['plan']; print({k: p[k] for k in ('num_segments', 'num_tiles', 'max_local', 'slots_per_sweep', 'redundancy')})
PY
    done
  done
done
tail -4 $O/pytest.log; cat $O/rc.txt; cat $O/c3_sweep.log
