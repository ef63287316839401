#!/bin/bash
# Round-3 evidence, collected on the GPU box in one gpurun call; summaries are copied into profiles/ afterwards.
set -u
ulimit -c 0
O=$PWD/gpurun_out/r03final; mkdir -p $O; export TMPDIR=/tmp; REPO=$PWD
( rocminfo | grep -E "Marketing|Compute Unit|gfx" | head -8; echo "nproc $(nproc)"; grep -m1 "model name" /proc/cpuinfo ) > $O/box.txt 2>&1
# 1. the driver's command, as is (stdout = compact lines, headline last; full record = bench_detail.json)
( time timeout -k 5 900 python bench.py > $O/bench_stdout.txt 2> $O/bench.err ) 2>> $O/box.txt; echo "bench rc=$?" >> $O/box.txt
cp bench_detail.json $O/bench_detail.json 2>/dev/null
# 2. rocprofv3 kernel stats of the SAME workload with the schedule forced (every persistent_kernel dispatch is a 10-sweep substep)
( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r03 -- python $REPO/bench.py --no-traffic --no-cpu-baseline --no-extras --persistent 2 > $O/bench_under_rocprof.txt 2> $O/stats.log )
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python - $O <<'PY'
import csv,glob,sys
O=sys.argv[1]
d=[]
for f in glob.glob(O+"/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "persistent_kernel" in r["Kernel_Name"]: d.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
if d:
    d.sort()
    open(O+"/persistent_kernel_dispatches.txt","w").write("persistent_kernel dispatches (all 10 sweeps x 3 segments, schedule forced): n=%d min %.1f us median %.1f us mean %.1f us max %.1f us\n" % (len(d), d[0], d[len(d)//2], sum(d)/len(d), d[-1]))
PY
find $O/stats -name "*kernel_trace.csv" -delete; find $O/stats -name "*.db" -delete 2>/dev/null
# 3. SQ / TCC counters of the timed kernel at this commit
KERNEL=persistent_kernel OUT=$O/pmc timeout 900 bash scripts/pmc_sq.sh --persistent 2 > $O/sq_counters_persistent_c2.log 2>&1
rm -rf $O/pmc
# 4. the N>1 launcher shapes the driver uses on an 8-GPU node, on this one-GPU box (ranks share the device: a smoke test of the path, not a measurement)
( time timeout 600 python bench.py --gpus 8 --oversubscribe --steps 10 --warmup 3 > $O/bench_gpus8_c2_oversubscribed.txt 2> $O/bench_gpus8.err ) 2>> $O/box.txt; echo "gpus8 c2 rc=$?" >> $O/box.txt
( time timeout 600 python bench.py --gpus 8 --oversubscribe --workload c4 --scaling strong --total-instances 512 --steps 10 --warmup 3 > $O/bench_gpus8_c4_strong_oversubscribed.txt 2>> $O/bench_gpus8.err ) 2>> $O/box.txt; echo "gpus8 c4 rc=$?" >> $O/box.txt
timeout 300 python bench.py --gpus 2 --oversubscribe --workload c4 --scaling strong --total-instances 6 --size 40 --steps 5 --warmup 2 --check-shards > $O/bench_gpus2_check_shards.txt 2>> $O/bench_gpus8.err; echo "gpus2 check-shards rc=$?" >> $O/box.txt
# 5. tests
timeout -k 5 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/box.txt
# 5b. the plug-in's round trip on its own (in the suite the number carries whatever earlier tests left behind)
timeout 300 python -m pytest tests/test_plugin.py -m gpu -q -s -k full_size_c2 2>&1 | grep -E "plug-in|passed|failed" > $O/plugin_round_trip.log
# 6. per-step timelines
timeout 200 python scripts/trace_tiles.py --persistent 2 > $O/trace_cloth_persistent.log 2>&1
timeout 200 python scripts/trace_tiles.py --bar 2 --persistent 2 > $O/trace_bar_fem_persistent.log 2>&1
cat $O/box.txt; tail -3 $O/pytest_gpu.log; cat $O/persistent_kernel_dispatches.txt; tail -1 $O/bench_stdout.txt | cut -c1-600; cat $O/plugin_round_trip.log
