#!/bin/bash
# round 3, GPU pass ae: dictionary form also for the FEM tet records (restVolume + Dm^-1) -- suite subset + A/B
set -u
O=$PWD/gpurun_out/r03ae; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_tetcontact.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {
  local label="$1"; local lib="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  PBDX_LIB=$lib timeout 300 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d.get('ms_per_step'), d['config'].get('schedule'))" >> $O/ab.log 2>&1
}
NF=$PWD/gpurun_variants/nodictfem/libpbdx.so
for rep in 1 2; do
  run "c3 FEM records from tables (rep $rep)" "" --workload c3 --solid-method 2 --steps 30 --warmup 5
  run "c3 FEM records streamed (rep $rep)" "$NF" --workload c3 --solid-method 2 --steps 30 --warmup 5
done
run "c3 FEM 16 bars, tables" "" --workload c3 --solid-method 2 --bars --instances 16 --steps 20 --warmup 5
run "c3 FEM 16 bars, streamed" "$NF" --workload c3 --solid-method 2 --bars --instances 16 --steps 20 --warmup 5
PBDX_PLAN_VERBOSE=1 timeout 100 python bench.py --workload c3 --solid-method 2 --no-cpu-baseline --no-extras --no-roofline --no-traffic --steps 3 --warmup 1 2>&1 | grep "dictionary form" | head -2 >> $O/ab.log
grep -E "passed|failed" $O/pytest.log | tail -2; cat $O/rc.txt; cat $O/ab.log
