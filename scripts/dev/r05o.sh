#!/bin/bash
# batch O: the plug-in's resident figure of batch N (1.20 ms per step against 0.65-0.68 everywhere else): box, or the bounce buffer?
O=gpurun_out/r05o; mkdir -p $O
for r in 1 2; do
  timeout 300 python -m pytest tests/test_plugin.py -m gpu -q -s -k "full_size_c2" -p no:cacheprovider > $O/plugin_$r.log 2>&1; echo "rc=$?"
  grep "plug-in at 1000\|round trip per step" $O/plugin_$r.log | cut -c1-300
done
