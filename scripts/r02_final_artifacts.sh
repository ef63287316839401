#!/bin/bash
# Round-2 evidence, collected on the GPU box in one gpurun call; summaries are copied into profiles/ afterwards.
set -u
ulimit -c 0
O=$PWD/gpurun_out/r02final; mkdir -p $O; export TMPDIR=/tmp; REPO=$PWD
( rocminfo | grep -E "Marketing|Compute Unit|gfx" | head -8; echo "nproc $(nproc)"; grep -m1 "model name" /proc/cpuinfo ) > $O/box.txt 2>&1
# 1. the driver's command, as is
timeout -k 5 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/box.txt
# 2. rocprofv3 kernel stats of the SAME workload with the schedule forced (no autotune launches: every persistent_kernel dispatch is a
#    10-sweep substep), no extras
( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r02 -- python $REPO/bench.py --no-traffic --no-cpu-baseline --no-extras --persistent 2 > $O/bench_under_rocprof.json 2> $O/stats.log )
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
if [ "${FULL_PMC:-0}" = 1 ]; then
KERNEL=persistent_kernel OUT=$O/pmc timeout 600 bash scripts/pmc_sq.sh --workload c4 --persistent 2 > $O/sq_counters_persistent_c4.log 2>&1
KERNEL=persistent_kernel OUT=$O/pmc timeout 600 bash scripts/pmc_sq.sh --workload c3 --solid-method 2 --persistent 2 --fuse 1 > $O/sq_counters_persistent_c3_fem.log 2>&1
fi
rm -rf $O/pmc
# 4. the N>1 launcher on this one-GPU box (ranks share the device: smoke test of the path, not a measurement)
timeout 300 python bench.py --gpus 2 --oversubscribe --steps 10 --warmup 3 --no-traffic --no-cpu-baseline --no-extras > $O/bench_gpus2_oversubscribed.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?" >> $O/box.txt
timeout 300 python bench.py --gpus 2 --oversubscribe --workload c4 --scaling strong --total-instances 16 --steps 10 --warmup 3 --no-traffic --no-cpu-baseline --no-extras > $O/bench_gpus2_c4_strong_oversubscribed.json 2>> $O/bench_gpus2.err; echo "gpus2 c4 strong rc=$?" >> $O/box.txt
# 5. tests
timeout -k 5 900 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/box.txt
# 5b. contacts between deformable solids: wall time per kernel (hipGraph off, a sync around every launch) and the chain microbenchmark
timeout -k 5 120 python scripts/dev/tet_profile.py 64,16,16 40 > $O/tetcontact.log 2>&1
( cd scripts/microbench && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o /tmp/chain chain.hip 2>/dev/null && timeout -k 5 60 /tmp/chain ) > $O/chain_microbench.log 2>&1
# 6. latency analysis of configs[2]
timeout 200 python scripts/trace_tiles.py --bar 2 --persistent 2 > $O/trace_bar_fem_persistent.log 2>&1
timeout 200 python scripts/trace_tiles.py --bar 6 --persistent 2 > $O/trace_bar_distvol_persistent.log 2>&1
timeout 200 python scripts/trace_tiles.py --persistent 2 > $O/trace_cloth_persistent.log 2>&1
( cd scripts/microbench && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/valu_single valu_single.hip 2>/dev/null && /tmp/valu_single ) > $O/valu_single.log 2>&1
cat $O/box.txt; tail -3 $O/pytest_gpu.log; cat $O/persistent_kernel_dispatches.txt; head -c 400 $O/bench.json
