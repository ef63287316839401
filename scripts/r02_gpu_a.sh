#!/bin/bash
# round 2, first GPU pass: tests, the driver's bench command, the N>1 launcher on one GPU, counters of the timed kernel
set -u
O=$PWD/gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing|Compute Unit|gfx" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo; grep -m1 flags /proc/cpuinfo | tr ' ' '\n' | grep -c avx512 ) > $O/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/box.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/box.txt
timeout 300 python bench.py --gpus 2 --oversubscribe --steps 10 --warmup 3 --no-traffic --no-cpu-baseline --no-extras > $O/bench_gpus2_oversub.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?" >> $O/box.txt
OUT=$O/pmc_sq timeout 900 bash scripts/pmc_sq.sh --persistent 2 > $O/pmc_sq_persistent.log 2>&1
rocprofv3 -L > $O/rocprof_counters_full.txt 2>&1
tail -3 $O/pytest.log; cat $O/box.txt; head -c 600 $O/bench.json
