# round 4, one gpurun call: (1) the 12-byte LDS layout questions, (2) configs[2] passes-per-sweep sweep at the current tiling,
# (3) -ffp-contract=fast A/B on the current kernels with step probes
mkdir -p gpurun_out
echo "=== lds_pack12"; timeout 120 gpurun_variants/mb/lds_pack12
echo "=== c3 passes per sweep (max colours per segment)"
for ms in 7 8 10 14 20; do
  echo "-- max-seg $ms"
  for m in 2 6; do
  timeout 300 python bench.py --workload c3 --solid-method $m --max-seg $ms --no-cpu-baseline --no-traffic --no-extras --steps 30 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('method $m: ms/substep %.4f median %.4f' % (d['ms_per_substep'], d['config'].get('device_median_ms_per_substep') or 0), 'ok' if d['config']['state_ok'] else 'STATE BAD')"
  done
done
echo "=== fp-contract A/B (in-tree = off)"
for rep in 1 2; do
for v in intree fast; do
  lib=$PWD/gpurun_variants/$v/libpbdx.so; [ $v = intree ] && lib=$PWD/positionbaseddynamics_amd/_lib/libpbdx.so
  for w in "" "--workload c3" "--workload c4"; do
  PBDX_LIB=$lib timeout 300 python bench.py $w --no-cpu-baseline --no-traffic --no-extras --steps 50 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$w]: ms/substep %.4f median %.4f' % (d['ms_per_substep'], d['config'].get('device_median_ms_per_substep') or 0), 'ok' if d['config']['state_ok'] else 'STATE BAD')"
  done
done
done
echo "=== step probes, 1000x1000 cloth: contraction off / fast"
PBDX_LIB=$PWD/gpurun_variants/probe/libpbdx.so timeout 300 python scripts/probe_steps.py --cloth 1000 2>&1 | tail -30
PBDX_LIB=$PWD/gpurun_variants/probe_fast/libpbdx.so timeout 300 python scripts/probe_steps.py --cloth 1000 2>&1 | tail -30
