#!/bin/bash
# round 5, GPU call G: does `s_nop 4` in front of the hand-written memory instructions (VALU-written SGPR -> VMEM hazard) end the faults of the
# range-checked build?  + the full selection of the bounds test, + configs[3] tile count x passes with the walk
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05g; mkdir -p $O; export TMPDIR=/tmp
B=$PWD/positionbaseddynamics_amd/_lib/libpbdx_bounds.so
for cfg in "--sim 1 --bend 2" "--sim 2 --bend 2" "--sim 4 --bend 3"; do
  pass=0; fault=0
  for i in 1 2 3 4 5 6; do
    PBDX_LIB=$B timeout 120 python scripts/dev/fault_repro.py --reps 4 --persistent 2 $cfg > $O/hunt_$i.log 2>&1
    if grep -q "Memory access fault" $O/hunt_$i.log; then fault=$((fault+1)); elif grep -q "completed" $O/hunt_$i.log; then pass=$((pass+1)); fi
  done
  echo "hunt[bounds + s_nop 4, $cfg, forced persistent]: $pass completed, $fault memory faults of 6 processes x 4 solvers" | tee -a $O/hunt.log
  grep -h "bounds {" $O/hunt_*.log | grep -v "'violations': 0" | head -2 | tee -a $O/hunt.log
done
SEL="known_answer_projection or fem_tet_inversion_branch or scene_parity_vs_float_reference or fused_tiles_equal_per_colour_schedule or persistent_schedule_is_bit_identical or full_size_c2_million_particle_cloth_vs_reference or full_size_c2_odd_pass_count or c4_ensemble_block or example_runs_and_matches_reference or dictionary_form or full_size_c3_100k or walk_of_a_1500"
( PBDX_LIB=$B timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_examples.py -m gpu -q -x -k "$SEL" > $O/pytest_bounds_full_selection.log 2>&1; echo "bounds build, full selection rc=$?"; tail -n 2 $O/pytest_bounds_full_selection.log | cut -c1-200 ) | tee -a $O/hunt.log
bash scripts/ab.sh --reps 1 --arm "in-tree" -- "--workload c4" "--workload c4 --tile 3334" "--workload c4 --tile 2500" "--workload c4 --max-seg 9" "--workload c4 --tile 3334 --max-seg 9" "--workload c4 --tile 3334 --max-seg 7" 2>&1 | tee $O/c4_search.log
bash scripts/ab.sh --reps 2 --arm "in-tree" -- "" "--workload c3" 2>&1 | tee $O/ab.log
