#!/bin/bash
# The evidence a round is judged on, collected on the GPU box in ONE gpurun call (usage: bash scripts/final_artifacts.sh r04); summaries are
# copied from gpurun_out/<tag>final/ into profiles/<tag>_* afterwards.  Counters and traces are separate rocprofv3 passes.
set -u
ulimit -c 0
TAG=${1:-r05}
O=$PWD/gpurun_out/${TAG}final; mkdir -p $O; export TMPDIR=/tmp; REPO=$PWD
( rocminfo | grep -E "Marketing|Compute Unit|gfx" | head -8; echo "nproc $(nproc)"; grep -m1 "model name" /proc/cpuinfo ) > $O/box.txt 2>&1
# 1. the driver's command, as is (stdout = compact lines, headline last; full record = bench_detail.json)
( time timeout -k 5 1200 python bench.py > $O/bench_stdout.txt 2> $O/bench.err ) 2>> $O/box.txt; echo "bench rc=$?" >> $O/box.txt
cp bench_detail.json $O/bench_detail.json 2>/dev/null
# 2. rocprofv3 kernel stats of the SAME workload with the schedule forced (every persistent_kernel dispatch is a 10-sweep substep)
( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- python $REPO/bench.py --no-traffic --no-cpu-baseline --no-extras --persistent 2 > $O/bench_under_rocprof.txt 2> $O/stats.log )
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
rm -rf $O/stats
# 3. SQ / TCC counters of the timed kernel at this commit
KERNEL=persistent_kernel OUT=$O/pmc timeout 900 bash scripts/pmc_sq.sh --persistent 2 > $O/sq_counters_persistent_c2.log 2>&1
rm -rf $O/pmc
# 3b. the same counters for the other two single-GPU lines: configs[2] (100 k-tet bar; in the ab.sh runs of round 6 the engine picked one launch per segment
#     for it, hence both kernel names) and the configs[3] block (no counter pass existed for either)
KERNEL=_kernel OUT=$O/pmc3 timeout 900 bash scripts/pmc_sq.sh --workload c3 > $O/sq_counters_c3_bar.log 2>&1
rm -rf $O/pmc3
KERNEL=persistent_kernel OUT=$O/pmc4 timeout 900 bash scripts/pmc_sq.sh --workload c4 --persistent 2 > $O/sq_counters_c4_block.log 2>&1
rm -rf $O/pmc4
# 4. the N>1 launcher shapes the driver uses on an 8-GPU node, on this one-GPU box (ranks share the device: a smoke test of the path, not a
#    measurement), and ONE rank over RCCL (the backend of the real run)
( time timeout 600 python bench.py --gpus 8 --oversubscribe --steps 10 --warmup 3 > $O/bench_gpus8_c2_oversubscribed.txt 2> $O/bench_gpus8.err ) 2>> $O/box.txt; echo "gpus8 c2 rc=$?" >> $O/box.txt
( time timeout 600 python bench.py --gpus 8 --oversubscribe --workload c4 --scaling strong --total-instances 512 --steps 10 --warmup 3 > $O/bench_gpus8_c4_strong_oversubscribed.txt 2>> $O/bench_gpus8.err ) 2>> $O/box.txt; echo "gpus8 c4 rc=$?" >> $O/box.txt
timeout 300 python bench.py --gpus 2 --oversubscribe --workload c4 --scaling strong --total-instances 6 --size 40 --steps 5 --warmup 2 --check-shards > $O/bench_gpus2_check_shards.txt 2>> $O/bench_gpus8.err; echo "gpus2 check-shards rc=$?" >> $O/box.txt
( MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --dist-backend nccl --no-cpu-baseline --no-traffic --no-extras --steps 20 --warmup 5 > $O/bench_rccl_world1.txt 2> $O/bench_rccl_world1.err ); echo "rccl world 1 rc=$?" >> $O/box.txt
grep "bench rank" $O/bench_rccl_world1.err >> $O/bench_rccl_world1.txt
# 4b. the same ensemble in ONE process through the C ABI (pbdx_ensemble_*): two engines sharing this GPU (smoke test of the path)
timeout 300 python bench.py --gpus 2 --single-process --devices 0,0 --workload c4 --instances 16 --steps 10 --warmup 3 > $O/bench_single_process_two_engines_one_gpu.txt 2>> $O/bench_gpus8.err; echo "single-process ensemble rc=$?" >> $O/box.txt
# 4c. two 512-thread workgroups per CU on today's kernel (round 4: 0.684 against 0.624 ms)
bash scripts/ab.sh --reps 1 --arm "one workgroup of 1024 per CU" -- "" > $O/two_wgs_per_cu.log 2>&1
bash scripts/ab.sh --reps 1 --arm "two workgroups of 512 per CU" -- "--wgs-per-cu 2 --fuse-block 512" >> $O/two_wgs_per_cu.log 2>&1
# 5. tests (with durations) and smoke
# (SKIP_TESTS=1 when the suite has just been run on this commit in its own call)
[ -z "${SKIP_TESTS:-}" ] && { timeout -k 5 2400 python -m pytest tests -m gpu -q -s --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/box.txt; }
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/box.txt
# 6. per-step timelines and step probes (probe library: scripts/build_variant.sh probe -DPBDX_STEP_PROBE=1)
timeout 200 python scripts/trace_tiles.py --persistent 2 > $O/trace_cloth_persistent.log 2>&1
timeout 200 python scripts/trace_tiles.py --bar 2 --persistent 2 > $O/trace_bar_fem_persistent.log 2>&1
[ -f gpurun_variants/probe/libpbdx.so ] && PBDX_LIB=$PWD/gpurun_variants/probe/libpbdx.so timeout 300 python scripts/probe_steps.py --cloth 1000 > $O/step_probes_cloth.log 2>&1
# 7. the plug-in's step() round trip at 1000x1000 through the unmodified reference model, and the headline command on other sheet sizes
timeout 600 python -m pytest tests/test_plugin.py -m gpu -q -s -k full_size_c2 2>&1 | grep -E "round trip|passed|failed" > $O/plugin_round_trip.log
: > $O/size_sweep.log
for n in 100 200 300 500 750 1000 1500 2000; do
	timeout 300 python bench.py --size $n --no-cpu-baseline --no-traffic --no-extras --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('cloth %4d x %-4d particles %8d constraints %9d: ms/substep %.4f (device median %.4f)  projections/s %.3e  %s' % ($n, $n, c['particles'], c['constraints'], d['ms_per_substep'], c.get('device_median_ms_per_substep') or d.get('device_median_ms_per_substep') or float('nan'), d['value'], 'ok' if c.get('state_ok') else 'STATE NOT OK'))" >> $O/size_sweep.log 2>&1
done
cat $O/box.txt; tail -16 $O/pytest_gpu.log 2>/dev/null; cat $O/plugin_round_trip.log $O/size_sweep.log; cat $O/persistent_kernel_dispatches.txt; tail -1 $O/bench_stdout.txt | cut -c1-700; tail -3 $O/smoke.log
