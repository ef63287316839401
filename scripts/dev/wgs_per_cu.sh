F="--no-cpu-baseline --no-traffic --no-extras --no-roofline"
run() { echo "== $*"; timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('   ms/substep %.4f  device median %s  ok=%s' % (d['ms_per_substep'], c.get('device_median_ms_per_substep'), c.get('state_ok')))"; python - <<'PY'
import json
try:
    d=json.load(open("bench_detail.json")); c=d["config"]
    print("   plan:", {k:c.get(k) for k in ("plan","persistent") if k in c})
except Exception as e: print("   (no detail)", e)
PY
}
run
run --wgs-per-cu 2
run --wgs-per-cu 2 --fuse-block 512
run --wgs-per-cu 2 --fuse-block 512 --max-seg 14
run --workload c4
run --workload c4 --wgs-per-cu 2 --fuse-block 512
run --workload c3
run --workload c3 --wgs-per-cu 2
