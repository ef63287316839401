#!/bin/bash
# Build a differently configured libpbdx.so into gpurun_variants/<name>/ (travels to the GPU box, stays out of git) for scripts/ab.sh:
#   bash scripts/build_variant.sh depth4 -DPBDX_DEPTH_BIG=4
#   bash scripts/build_variant.sh probe -DPBDX_STEP_PROBE=1          (scripts/probe_steps.py)
#   bash scripts/build_variant.sh passprobe -DPBDX_PASS_PROBE=1     (scripts/probe_pass.py)
set -eu
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
make -j8 -C "$root/positionbaseddynamics_amd/csrc" OUT="$root/gpurun_variants/$name" EXTRA="$*" "$root/gpurun_variants/$name/libpbdx.so"
