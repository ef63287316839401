#!/bin/bash
# One parameterised A/B runner (replaces the 40 one-shot r02_* / r03_gpu_* / r04_ab* scripts of earlier rounds; their command lines are
# listed in scripts/README.md, their results in profiles/HISTORY.md).  Run on the GPU box through gpurun:
#
#   bash scripts/ab.sh [--suite] [--reps N] [--steps K] [--warmup W] --arm "label[:VAR=value ...]" [--arm ...] -- "bench opts 1" ["bench opts 2" ...]
#
#   --arm    one configuration to compare: a label and environment assignments, e.g.
#              --arm "in-tree"                                   the library in positionbaseddynamics_amd/_lib
#              --arm "fma:PBDX_LIB=$PWD/positionbaseddynamics_amd/_lib/libpbdx_fma.so"
#              --arm "depth4:PBDX_LIB=$PWD/gpurun_variants/depth4/libpbdx.so"    (scripts/build_variant.sh depth4 -DPBDX_DEPTH_BIG=4)
#              --arm "no dictionary:PBDX_NO_DICT=1"
#   after -- one quoted string of bench.py options per workload ("" = the headline configs[1] sheet)
#   --suite  run the GPU parity tests first (gpurun_out/ab_pytest.log)
# Every (workload, arm, repetition) prints: host-clock ms per substep, median device ms per substep, passes per sweep, state check.
set -u
REPS=1; STEPS=50; WARMUP=20; SUITE=0; ARMS=()
while [ $# -gt 0 ]; do
  case "$1" in
    --suite) SUITE=1; shift;;
    --reps) REPS=$2; shift 2;;
    --steps) STEPS=$2; shift 2;;
    --warmup) WARMUP=$2; shift 2;;
    --arm) ARMS+=("$2"); shift 2;;
    --) shift; break;;
    *) echo "unknown option $1"; exit 2;;
  esac
done
[ ${#ARMS[@]} -eq 0 ] && ARMS=("in-tree")
[ $# -eq 0 ] && set -- ""
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ $SUITE = 1 ]; then
  timeout 2400 python -m pytest tests -m gpu -q -s -x > gpurun_out/ab_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/ab_pytest.log | tail -2
fi
for rep in $(seq 1 $REPS); do
  for w in "$@"; do
    for arm in "${ARMS[@]}"; do
      label=${arm%%:*}; envs=""; [ "$arm" != "$label" ] && envs=${arm#*:}
      env $envs PBDX_BENCH_DETAIL=/tmp/ab_detail.json timeout 600 python bench.py $w --no-cpu-baseline --no-traffic --no-extras --no-roofline --steps $STEPS --warmup $WARMUP 2>/dev/null | tail -1 | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read()); c = d['config']
    segs = json.load(open('/tmp/ab_detail.json'))['config']['plan']['num_segments']
    sch = c.get('schedule') or {}
    print('%-28s [%s] rep $rep: ms/substep %.4f  device median %.4f  passes/sweep %s  %s  %s' % ('$label', '$w', d['ms_per_substep'], c.get('device_median_ms_per_substep') or 0, segs,
        'one launch' if sch.get('persistent') else 'launch per segment' if sch.get('fused') else 'launch per colour', 'ok' if c['state_ok'] else 'STATE BAD'))
except Exception as e:
    print('%-28s [%s] rep $rep: FAILED %r' % ('$label', '$w', e))"
    done
  done
done
