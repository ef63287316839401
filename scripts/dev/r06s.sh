#!/bin/bash
# cheap levers on the final kernel: the compiler's scheduling strategy (same arithmetic, same order of floating-point operations: bit-identical by construction,
# checked by the bench's state check), and the planner's segment length (passes per sweep)
export TMPDIR=/tmp
V=$PWD/gpurun_variants
bash scripts/ab.sh --reps 2 --arm "in-tree" --arm "max-ilp:PBDX_LIB=$V/sched_ilp/libpbdx.so" --arm "max-memory-clause:PBDX_LIB=$V/sched_mem/libpbdx.so" --arm "iterative-ilp:PBDX_LIB=$V/sched_iter/libpbdx.so" -- "" "--workload c3" "--workload c4" 2>&1 | tee gpurun_out/r06s_sched.log
bash scripts/ab.sh --reps 1 --arm "in-tree" -- "--max-seg 5" "--max-seg 6" "--max-seg 7" "--max-seg 9" "--max-seg 14" 2>&1 | tee gpurun_out/r06s_maxseg.log
