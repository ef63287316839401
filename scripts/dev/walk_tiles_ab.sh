timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "c4_ensemble or odd_pass" 2>&1 | tail -3
F="--no-cpu-baseline --no-traffic --no-extras --no-roofline --steps 30 --warmup 20"
for rep in 1 2; do for arm in walk nowalk; do for wl in "--workload c4" "--size 1500" "--size 2000"; do
  if [ $arm = nowalk ]; then export PBDX_LIB=$PWD/gpurun_variants/nowalk/libpbdx.so; else unset PBDX_LIB; fi
  python bench.py $F $wl 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('%-7s %-14s ms/substep %.4f  device median %.4f  ok=%s' % ('$arm', '$wl', d['ms_per_substep'], c.get('device_median_ms_per_substep') or 0, c.get('state_ok')))"
done; done; done
