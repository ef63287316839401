#!/bin/bash
# batch R (the round's last seconds of GPU time): the plug-in's 1 M test and the transfer test on the final library
O=gpurun_out/r05r; mkdir -p $O
timeout 100 python -m pytest tests/test_plugin.py -m gpu -q -s -x -k "full_size_c2 or block_hashes" -p no:cacheprovider > $O/plugin.log 2>&1; echo "rc=$?"
grep "passed\|failed\|plug-in at 1000\|round trip per step\|engine of" $O/plugin.log | cut -c1-260
