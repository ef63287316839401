#!/bin/bash
# round 3, GPU pass f: per-iteration persistent schedule with contacts, quad-lane variant test, plug-in timing breakdown
set -u
O=$PWD/gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python bench.py --workload c5 > $O/bench_c5.out 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/rc.txt
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head; cat $O/rc.txt
grep -E "plug-in|tet contacts timing|schedule with contacts|quad-lane build|armadillo scene" $O/pytest.log; cat $O/bench_c5.out | cut -c1-700
