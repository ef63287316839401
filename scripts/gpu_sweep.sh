# usage: bash scripts/gpu_sweep.sh [notest] -- GPU parity tests, then a bench sweep over launch options
mkdir -p gpurun_out
if [ "$1" != "notest" ]; then
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_sweep.log 2>&1; echo pytest exit $?; tail -5 gpurun_out/pytest_sweep.log
fi
while IFS= read -r opt; do
  echo "== $opt"
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 $opt 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
    print('ms/substep %.3f  proj/s %.3e  frac %.3f  plan %s' % (d['ms_per_substep'], d['value'], r.get('frac',0), {k:d['config']['plan'][k] for k in ('active','num_segments','num_tiles','max_local','redundancy','build_seconds','stream_bytes_per_sweep')}))
    for s in r.get('segments',[]): print('   seg', s['segment'], s['colours'], 'block', s['block'], 'avg_us %.1f'%s['avg_us'], 'alg GB/s %.0f'%s['algorithmic_GBs'], 'stream GB/s %.0f'%s['streamed_GBs'])
except Exception as e: print('ERR', e)
"
done < "${SWEEP_FILE:-scripts/sweep_opts.txt}"
