#!/bin/bash
# round 3, GPU pass g: where does a single-step call of the plug-in spend its time?  (hash threads 32 / 8 / 1)
set -u
O=$PWD/gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
for th in 32 8 1; do
  echo "== hash threads $th" >> $O/plugin.log
  PBDX_PLUGIN_HASH_THREADS=$th timeout 600 python -m pytest tests/test_plugin.py -m gpu -q -s -k full_size_c2 2>&1 | grep -E "plug-in|passed|failed" >> $O/plugin.log
done
echo "== OMP_WAIT_POLICY=passive, hash threads 32" >> $O/plugin.log
OMP_WAIT_POLICY=passive PBDX_PLUGIN_HASH_THREADS=32 timeout 600 python -m pytest tests/test_plugin.py -m gpu -q -s -k full_size_c2 2>&1 | grep -E "plug-in|passed|failed" >> $O/plugin.log
cat $O/plugin.log
