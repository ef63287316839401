#!/bin/bash
# A/B of library variants (gpurun_variants/<name>/libpbdx.so, plus the in-tree build as "tree") on given bench args, interleaved, N rounds
set -u
ROUNDS=${ROUNDS:-2}
O=$PWD/gpurun_out/${TAG:-ab}; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-traffic --no-extras --no-roofline --steps 40 --warmup 10"
for r in $(seq $ROUNDS); do
for v in tree $(ls gpurun_variants); do
  if [ "$v" = tree ]; then lib=$PWD/positionbaseddynamics_amd/_lib/libpbdx.so; else lib=$PWD/gpurun_variants/$v/libpbdx.so; fi
  PBDX_LIB=$lib timeout 300 $B "$@" > $O/$v.json 2> $O/$v.err
  python - $O/$v.json $v <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']; p=c['plan']; pe=c['persistent']
    print("%-14s %.4f ms  segs=%d tiles=%d persist=%d folded=%d block=%d ok=%s" % (sys.argv[2], d['ms_per_substep'], p['num_segments'], p['num_tiles'], pe['active'], pe['last_folded'], pe['block'], c['state_ok']))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done; done
