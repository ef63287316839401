#!/bin/bash
# batch M: particle transfers through the engine-owned page-locked mirror (no hipHostRegister of caller memory any more): the plug-in and
# tet-contact modules first; if green, the record of the commit (bench, full GPU suite, smoke) as in batch J
O=gpurun_out/r05m; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_plugin.py tests/test_tetcontact.py "tests/test_gpu_parity.py::test_resident_state_dirty_tracking_and_explicit_sync" -m gpu -q -s -x -p no:cacheprovider > $O/pytest_subset.log 2>&1
rc=$?; echo "subset rc=$rc"; grep -n "round trip\|passed\|failed\|Error\|fault" $O/pytest_subset.log | cut -c1-400 | tail -12
[ $rc -ne 0 ] && exit 1
bash scripts/dev/r05n.sh
