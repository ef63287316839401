#!/bin/bash
# round 5, GPU call I: the plug-in tests after the device-ahead ordering fix (they had not run since the speculative step went in: earlier suite runs stopped at
# the bounds test), round trip with 128 pool threads
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05i; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_plugin.py tests/test_tetcontact.py -m gpu -q -x -s > $O/pytest_plugin.log 2>&1; echo "pytest plugin rc=$?"; tail -n 3 $O/pytest_plugin.log | cut -c1-200 )
grep -E "round trip|speculative|hook|mixed model" $O/pytest_plugin.log | cut -c1-400
PBDX_PLUGIN_HASH_THREADS=64 timeout 300 python -m pytest tests/test_plugin.py -m gpu -q -s -k full_size_c2 2>&1 | grep -E "round trip" | cut -c1-400 | sed 's/^/[64 threads] /'
