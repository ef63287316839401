# round 4: fetch-before-barrier A/B (C2, c3 x3, c4, small cloths), step probes with it, planner cost-model sweep for small scenes
mkdir -p gpurun_out
one() { # lib label opts...
  lib=$1; label=$2; shift 2
  PBDX_LIB=$lib timeout 300 python bench.py "$@" --no-cpu-baseline --no-traffic --no-extras --steps 50 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label [$*]: ms/substep %.4f median %.4f' % (d['ms_per_substep'], d['config'].get('device_median_ms_per_substep') or 0), 'segs', json.load(open('bench_detail.json'))['config']['plan']['num_segments'], 'ok' if d['config']['state_ok'] else 'STATE BAD')"
}
IN=$PWD/positionbaseddynamics_amd/_lib/libpbdx.so
EA=$PWD/gpurun_variants/early/libpbdx.so
echo "=== fetch before the barrier A/B"
for rep in 1 2; do
for w in "" "--workload c3" "--workload c3 --solid-method 4" "--workload c3 --solid-method 6" "--workload c4" "--size 300" "--size 100"; do
  one $IN intree $w; one $EA early $w
done
done
echo "=== step probes with the fetch before the barrier, 1000x1000 cloth"
PBDX_LIB=$PWD/gpurun_variants/probe_early/libpbdx.so timeout 300 python scripts/probe_steps.py --cloth 1000 2>&1 | tail -21
echo "=== planner cost model, small scenes (in-tree library)"
for ss in 0.3 0.15; do for ln in 3000 5000 8000 12000; do
  for w in "--workload c3" "--workload c3 --solid-method 4" "--workload c3 --solid-method 6" "--size 100" "--size 200"; do
    PBDX_PLAN_SLOT_SCALE=$ss PBDX_PLAN_LAUNCH_NS=$ln one $IN "slot $ss launch $ln" $w
  done
done; done
