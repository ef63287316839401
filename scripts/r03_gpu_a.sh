#!/bin/bash
# round 3, first GPU pass: tests, the driver's bench command (compact line), the N>1 launcher shapes on one GPU, the fp-contract=fast A/B
set -u
O=$PWD/gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
( rocminfo | grep -E "Marketing|Compute Unit|gfx" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo ) > $O/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/box.txt
( time timeout 600 python bench.py > $O/bench.out 2> $O/bench.err ) 2>> $O/box.txt; echo "bench rc=$?" >> $O/box.txt
cp bench_detail.json $O/bench_detail.json 2>/dev/null
( time timeout 600 python bench.py --gpus 8 --oversubscribe --steps 10 --warmup 3 > $O/bench_gpus8_c2_oversub.out 2> $O/bench_gpus8_c2.err ) 2>> $O/box.txt; echo "gpus8 c2 rc=$?" >> $O/box.txt
( time timeout 600 python bench.py --gpus 8 --oversubscribe --workload c4 --scaling strong --total-instances 512 --steps 10 --warmup 3 > $O/bench_gpus8_c4_strong_oversub.out 2> $O/bench_gpus8_c4.err ) 2>> $O/box.txt; echo "gpus8 c4 rc=$?" >> $O/box.txt
timeout 300 python bench.py --gpus 2 --oversubscribe --workload c4 --scaling strong --total-instances 6 --size 40 --steps 5 --warmup 2 --check-shards > $O/bench_gpus2_check_shards.out 2> $O/bench_gpus2_check.err; echo "gpus2 check-shards rc=$?" >> $O/box.txt
# what does bit-identity cost?  the same library built with -ffp-contract=fast (opt-in build, not the product default)
PBDX_LIB=$PWD/gpurun_variants/fast/libpbdx.so timeout 300 python bench.py --no-cpu-baseline --no-extras --no-traffic > $O/bench_fp_contract_fast.out 2> $O/bench_fast.err; echo "fast rc=$?" >> $O/box.txt
for m in 2 4 6; do PBDX_LIB=$PWD/gpurun_variants/fast/libpbdx.so timeout 200 python bench.py --workload c3 --solid-method $m --no-cpu-baseline --no-extras --no-traffic --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_fp_contract_fast_c3_m$m.out; done
tail -3 $O/pytest.log; cat $O/box.txt; tail -c 2500 $O/bench.out; echo; tail -c 600 $O/bench_fp_contract_fast.out
