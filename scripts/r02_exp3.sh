#!/bin/bash
# A/B of library variants under gpurun_variants/<name>/libpbdx.so on the bar (configs[2]) and the cloth
set -u
O=$PWD/gpurun_out/r02e; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-traffic --no-extras --no-roofline --steps 40 --warmup 10"
run() { lib=$1; tag=$2; shift 2; PBDX_LIB=$lib timeout 300 $B "$@" > $O/$tag.json 2> $O/$tag.err; python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']; p=c['plan']; pe=c['persistent']
    print("%-34s %.4f ms  segs=%d tiles=%d persist=%d folded=%d block=%d ok=%s" % (sys.argv[2], d['ms_per_substep'], p['num_segments'], p['num_tiles'], pe['active'], pe['last_folded'], pe['block'], c['state_ok']))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for v in "" $(ls gpurun_variants); do
  if [ -z "$v" ]; then lib=$PWD/positionbaseddynamics_amd/_lib/libpbdx.so; name=base; else lib=$PWD/gpurun_variants/$v/libpbdx.so; name=$v; fi
  run $lib ${name}_c3m2 --workload c3 --solid-method 2
  run $lib ${name}_c3m4 --workload c3 --solid-method 4
  run $lib ${name}_c3m6 --workload c3 --solid-method 6
  run $lib ${name}_c2 --workload c2
done
