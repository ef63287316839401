# configs[3] block: passes per sweep (max colours per segment) under the current engine
F="--workload c4 --no-cpu-baseline --no-traffic --no-extras --no-roofline --steps 30 --warmup 20"
for a in "" "--max-seg 27" "--max-seg 10" "--max-seg 9" "--max-seg 7" "--lds-particles 9000" ; do
	python bench.py $F $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('%-22s ms/substep %.4f  device median %.4f  ok=%s' % ('$a', d['ms_per_substep'], c.get('device_median_ms_per_substep') or 0, c.get('state_ok')))"
	python - <<'PY'
import json
d=json.load(open("bench_detail.json")); p=d["config"]["plan"]
print("      segments %d tiles %d max_local %d redundancy %.3f" % (p["num_segments"], p["num_tiles"], p["max_local"], p["redundancy"]))
PY
done
