# A/B differently tuned builds of libpbdx.so (gpurun_variants/<name>/libpbdx.so) on the headline bench
# usage: bash scripts/ab_variants.sh "<bench opts 1>" "<bench opts 2>" ...
for d in gpurun_variants/*/; do
  for opts in "$@"; do
  echo "== $d $opts"
  PBDX_LIB=$PWD/$d/libpbdx.so timeout 300 python bench.py --no-cpu-baseline --no-traffic --steps 30 --warmup 10 $opts 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print('ms/substep %.3f  proj/s %.3e' % (d['ms_per_substep'], d['value']), [(s['block'], round(s['avg_us'],1)) for s in r.get('segments',[])], 'ok' if d['config']['state_ok'] else 'STATE BAD')
"
  done
done
