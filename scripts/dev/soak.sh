#!/bin/bash
# Soak of the final library: N consecutive runs of the GPU suite (without its two long tests: the 100-step 1 M cloth and the
# range-checked build's selection, 170 s of 335) and M repetitions of the test a GPU memory fault at a host address once ended
# (test_plugin_tet_contacts_medium_scene_timing_and_parity, profiles/HISTORY.md [9]), each block of 50 in a fresh process.  A failing run is kept with
# its output; a "Memory access fault" is looked for in every log.   usage: soak.sh [suite runs] [tet repetitions]
export TMPDIR=/tmp
N=${1:-12}; M=${2:-200}
out=gpurun_out/soak; mkdir -p $out
fail=0; faults=0
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_full_size_c2_hundred_steps_and_f64_envelope \
     --deselect tests/test_gpu_parity.py::test_bounds_checked_build_finds_no_out_of_range_access > $out/suite_$i.log 2>&1
  rc=$?
  grep -q "Memory access fault" $out/suite_$i.log && faults=$((faults+1))
  if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "suite run $i: rc=$rc $(tail -1 $out/suite_$i.log)"; else echo "suite run $i: $(tail -1 $out/suite_$i.log)"; rm -f $out/suite_$i.log; fi
done
T=tests/test_tetcontact.py::test_plugin_tet_contacts_medium_scene_timing_and_parity
tfail=0; done_reps=0
while [ $done_reps -lt $M ]; do
  reps=$(( M - done_reps )); [ $reps -gt 50 ] && reps=50
  # (pytest runs a node id once however often it is named: the repetitions are separate pytest.main calls inside ONE process)
  timeout 1800 python -c "
import sys, pytest
for k in range($reps):
    rc = pytest.main(['$T', '-m', 'gpu', '-q', '-p', 'no:cacheprovider'])
    if rc != 0:
        print('repetition', k, 'failed with', rc); sys.exit(1)
print('$reps repetitions passed')
" > $out/tet_$done_reps.log 2>&1
  rc=$?
  grep -q "Memory access fault" $out/tet_$done_reps.log && faults=$((faults+1))
  if [ $rc -ne 0 ]; then tfail=$((tfail+1)); echo "tet block at $done_reps: rc=$rc $(tail -1 $out/tet_$done_reps.log)"; else echo "tet block at $done_reps ($reps repetitions): $(tail -1 $out/tet_$done_reps.log)"; rm -f $out/tet_$done_reps.log; fi
  done_reps=$(( done_reps + reps ))
done
echo "SOAK: $fail of $N suite runs failed, $tfail tet blocks failed ($M repetitions), $faults logs with a GPU memory fault"
