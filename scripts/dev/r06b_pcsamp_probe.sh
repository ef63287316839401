#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06b_pcs; mkdir -p $out
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 rocprofv3-avail list --pc-sampling > $out/avail_list.txt 2>&1
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 rocprofv3-avail info --pc-sampling > $out/avail_info.txt 2>&1
cat $out/avail_list.txt $out/avail_info.txt | head -60
