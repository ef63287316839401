#!/bin/bash
# quick GPU pass: the GPU test suite, then ms/substep of the headline cloth and the bar variants (no profiling passes)
set -u
O=$PWD/gpurun_out/${1:-quick}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-traffic --no-extras --no-roofline --steps 40 --warmup 10"
run() { tag=$1; shift; timeout 300 $B "$@" > $O/$tag.json 2> $O/$tag.err; python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']; p=c['plan']; pe=c['persistent']
    print("%-20s %.4f ms  segs=%d tiles=%d red=%.2f persist=%d folded=%d block=%d ok=%s" % (sys.argv[2], d['ms_per_substep'], p['num_segments'], p['num_tiles'], p['redundancy'], pe['active'], pe['last_folded'], pe['block'], c['state_ok']))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run c2 --workload c2
run c2_again --workload c2
run c3m2 --workload c3 --solid-method 2
run c3m4 --workload c3 --solid-method 4
run c3m6 --workload c3 --solid-method 6
run c4 --workload c4
