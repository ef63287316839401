# quick GPU check of a kernel change: step trace, headline bench (no PMC / CPU legs), GPU parity tests
python scripts/trace_tiles.py > gpurun_out/trace_now.log 2>&1; grep -E "segment|write-back|step dur" gpurun_out/trace_now.log
for w in "" "--workload c3"; do
python bench.py --no-cpu-baseline --no-traffic --steps 50 --warmup 20 $w > gpurun_out/b.log 2>&1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b.log").read().strip().splitlines()[-1])
print("ms/substep %.4f  proj/s %.4e" % (d["ms_per_substep"], d["value"]), [round(s["avg_us"], 1) for s in d.get("roofline", {}).get("segments", [])])
PY
done
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
