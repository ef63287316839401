# SQ-level counters of the dominant kernel (developer aid): where do the wave cycles go?
#   KERNEL=persistent_kernel (default) | fused_kernel | project_kernel ; arguments = bench.py workload flags
#   e.g.  bash scripts/pmc_sq.sh --persistent 2            (headline config, one launch per substep)
# Each counter set is its own rocprofv3 pass (--pmc with --kernel-trace only, as gpurun requires).
set -u
KERNEL=${KERNEL:-persistent_kernel}
OUT=${OUT:-$PWD/gpurun_out/pmc_sq}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|TCC_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+)\b" | sort -u | tr '\n' ' ' > $OUT/counters.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $REPO/bench.py --pmc-child "$@" > /dev/null 2> $OUT/$tag.log
  python - "$OUT/$tag" "$KERNEL" <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(d.items()): print("%-26s mean %.6g  (n=%d)"%(k,sum(v)/len(v),len(v)))
PY
  rm -rf $OUT/$tag
done
