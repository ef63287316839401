# builds scan_bench variants (prefetch distance) into gpurun_variants/scan/ ; run them on the GPU box's host with different thread counts
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_variants/scan
for pf in 12; do
g++ -std=c++11 -O2 -fopenmp -w -ffp-contract=off -DPBDX_SCAN_PREFETCH=$pf -I/root/reference -I/root/reference/extern/eigen -Ioracle/shim -o gpurun_variants/scan/scan_bench_pf$pf scripts/dev/scan_bench.cpp \
  $(ls oracle/_ref/obj_f32/*.o | grep -v ref_driver) -Lpositionbaseddynamics_amd/_lib -lpbdx -Wl,-rpath,'$ORIGIN/../../positionbaseddynamics_amd/_lib'
done
