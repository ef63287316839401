# refresh the judged artifacts (run through gpurun; copy results from gpurun_out/final/ into profiles/)
OUT=gpurun_out/final; mkdir -p $OUT
python bench.py --gpus 1 --steps 50 --warmup 20 > $OUT/bench_final.json 2> $OUT/bench_final.err
python bench.py --workload c4 --steps 30 --warmup 10 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --workload c3 --steps 50 --warmup 20 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
bash scripts/profile.sh final > $OUT/profile.log 2>&1
cp gpurun_out/prof_final/kernel_stats.csv $OUT/kernel_stats.csv
python scripts/trace_tiles.py --graph 1 --persistent 2 > $OUT/tile_trace_persistent.log 2>&1
python scripts/trace_tiles.py --graph 1 --persistent 0 > $OUT/tile_trace_multilaunch.log 2>&1
timeout 300 python scripts/persist_check.py > $OUT/persist_check.log 2>&1
for f in bench_final bench_c4 bench_c3; do python - $OUT/$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {})
print(sys.argv[1], "ms/substep %.4f value %.4e frac %.3f kernel %s avg_us %.1f rocprof_us %s traffic %s persistent_active %s" % (
    d["ms_per_substep"], d["value"], r.get("frac", 0), r.get("kernel", "")[:30], r.get("avg_launch_us", 0), r.get("rocprofv3_mean_kernel_us"), r.get("traffic"), d["config"]["persistent"]["active"]))
PY
done
head -5 $OUT/kernel_stats.csv | cut -c1-200
