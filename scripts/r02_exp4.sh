#!/bin/bash
set -u
O=$PWD/gpurun_out/r02h; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-traffic --no-extras --no-roofline --steps 40 --warmup 10"
run() { tag=$1; shift; timeout 300 $B "$@" > $O/$tag.json 2> $O/$tag.err; python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']; p=c['plan']; pe=c['persistent']
    print("%-26s %.4f ms  segs=%d tiles=%d red=%.2f max_local=%d persist=%d block=%d ok=%s" % (sys.argv[2], d['ms_per_substep'], p['num_segments'], p['num_tiles'], p['redundancy'], p['max_local'], pe['active'], pe['block'], c['state_ok']))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for m in 2 6; do
for sc in 1.0 0.5 0.25 0.1 0.0; do
  for ms in 16 24; do
    PBDX_PLAN_SLOT_SCALE=$sc run c3m${m}_scale${sc}_seg${ms} --workload c3 --solid-method $m --max-seg $ms --persistent 2
  done
done
done
