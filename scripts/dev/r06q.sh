#!/bin/bash
# dynamic rigid body as impulse sink (plug-in), then the ensemble tests of a shared device repeated (the soak's failure), then the plug-in + contact tests
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_plugin.py -m gpu -q -x -s -p no:cacheprovider -k "dynamic_rigid_body" > gpurun_out/r06q_dynamic.log 2>&1; echo "dynamic rc=$?"; tail -12 gpurun_out/r06q_dynamic.log
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_distributed.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "single_process_ensemble or known_answer_projection" > gpurun_out/r06q_ens_$i.log 2>&1; echo "ensemble run $i rc=$? $(tail -1 gpurun_out/r06q_ens_$i.log)"; done
timeout 1500 python -m pytest tests/test_plugin.py tests/test_contacts.py tests/test_tetcontact.py -m gpu -q -p no:cacheprovider > gpurun_out/r06q_plugin.log 2>&1; echo "plugin+contacts rc=$? $(tail -1 gpurun_out/r06q_plugin.log)"
