#!/bin/bash
set -u
O=$PWD/gpurun_out/r02c; mkdir -p $O
timeout 900 python -m pytest tests/test_plugin.py tests/test_gpu_parity.py -m gpu -q -s -x -k "plugin or contacts_per_particle" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log; grep -h "plug-in at" $O/pytest.log
B="python bench.py --no-cpu-baseline --no-traffic --no-extras --steps 50 --warmup 10"
run() { tag=$1; shift; timeout 300 $B "$@" > $O/$tag.json 2> $O/$tag.err; python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']; p=c['plan']; pe=c['persistent']
    print("%-28s %.4f ms  segs=%d tiles=%d red=%.3f max_local=%d persist=%d folded=%d grid=%d block=%d refusals=%d" % (sys.argv[2], d['ms_per_substep'], p['num_segments'], p['num_tiles'], p['redundancy'], p['max_local'], pe['active'], pe['last_folded'], pe['grid'], pe['block'], pe['refusals']))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run base
run base_p2 --persistent 2
run wg2_b512 --persistent 2 --wgs-per-cu 2 --fuse-block 512
run wg2_b512_seg2 --persistent 2 --wgs-per-cu 2 --fuse-block 512 --max-seg 14
run wg1_b512 --persistent 2 --fuse-block 512
run wg2_b256 --persistent 2 --wgs-per-cu 2 --fuse-block 256
run wg3_b256 --persistent 2 --wgs-per-cu 3 --fuse-block 256
run wg4_b256 --persistent 2 --wgs-per-cu 4 --fuse-block 256
