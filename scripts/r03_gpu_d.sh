#!/bin/bash
# round 3, GPU pass d: suite with the velocity solve of tet contacts + armadillo scene, the bench with its c5 lines
set -u
O=$PWD/gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
( time timeout 900 python bench.py > $O/bench.out 2> $O/bench.err ) 2>> $O/rc.txt; echo "bench rc=$?" >> $O/rc.txt
cp bench_detail.json $O/bench_detail.json 2>/dev/null
grep -E "passed|failed|Error|error" $O/pytest.log | tail -8; cat $O/rc.txt; cat $O/bench.out | cut -c1-1500
