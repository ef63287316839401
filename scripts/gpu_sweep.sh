mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest3.log 2>&1; echo pytest exit $?; tail -5 gpurun_out/pytest3.log
for opt in "" "--fuse 0" "--tile 2048" "--tile 1024" "--tile 2048 --fuse-block 512" "--tile 1024 --fuse-block 256" "--fuse-block 512"; do
  echo "== $opt"
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 $opt 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
    print('ms/substep %.3f  proj/s %.3e  frac %.3f  plan %s' % (d['ms_per_substep'], d['value'], r.get('frac',0), {k:d['config']['plan'][k] for k in ('active','num_segments','num_tiles','max_local','redundancy','build_seconds')}))
    for s in r.get('segments',[]): print('   seg', s['segment'], s['colours'], 'block', s['block'], 'avg_us %.1f'%s['avg_us'], 'alg GB/s %.0f'%s['algorithmic_GBs'], 'stream GB/s %.0f'%s['streamed_GBs'])
except Exception as e: print('ERR', e)
"
done
