#!/bin/bash
# round 5, GPU call Q: the record of the final commit -- the driver's bench command, the full GPU suite, smoke
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05q; mkdir -p $O; export TMPDIR=/tmp
( time timeout -k 5 1200 python bench.py > $O/bench_stdout.txt 2> $O/bench.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; cp bench_detail.json $O/bench_detail.json 2>/dev/null
tail -n 1 $O/bench_stdout.txt | cut -c1-600; cat $O/bench_time.txt
( timeout -k 5 2400 python -m pytest tests -m gpu -q -s --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 16 $O/pytest_gpu.log | cut -c1-200 )
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 3 $O/smoke.log
