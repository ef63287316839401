#!/bin/bash
# batch P: host copies by a resident team of threads against threads started per call (scripts/dev/hostio_bench.py)
O=gpurun_out/r05p; mkdir -p $O
for lib in positionbaseddynamics_amd/_lib/libpbdx.so gpurun_variants/spawn/libpbdx.so; do
  echo "== $lib" | tee -a $O/hostio.log
  PBDX_LIB=$PWD/$lib timeout 200 python scripts/dev/hostio_bench.py 2>&1 | tee -a $O/hostio.log
done
