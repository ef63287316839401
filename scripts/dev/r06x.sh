#!/bin/bash
# boundary ids alone in LDS for tiles that leave no room for the halo ids (the walk: configs[3] block, sheets above 1 M particles): parity first, then A/B
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "c4_ensemble_block or walk_of_a_1500 or full_size_c2_million or odd_pass or persistent_schedule_is_bit" 2>&1 | tail -3
bash scripts/ab.sh --reps 2 --arm "ids in LDS (boundary ids alone where the halo ids do not fit)" --arm "no ids in LDS:PBDX_NO_LDS_IDS=1" -- "--workload c4" "--size 1500" "--size 2000" 2>&1 | tee gpurun_out/r06x_bnd_ids.log
bash scripts/ab.sh --reps 1 --arm "in-tree" -- "" "--workload c3" 2>&1 | tee -a gpurun_out/r06x_bnd_ids.log
