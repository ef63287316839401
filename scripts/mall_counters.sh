#!/bin/bash
# Which rocprofv3 counters see Infinity-Cache (MALL) hits?  Two streaming kernels with the SAME number of fabric requests:
#   calib_read_b128    1 GiB read once            (cannot hit the 256 MiB Infinity Cache)
#   calib_reread_b128  128 MiB read eight times   (beyond the 4 MiB-per-XCD L2s, inside the Infinity Cache after pass 1)
# A counter that reports ~1/8 for the second kernel excludes Infinity-Cache hits (= true HBM traffic); one that reports the same sees them.
set -u
OUT=${OUT:-$PWD/gpurun_out/mall}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
cat > /tmp/mall_drv.py <<PY
import sys; sys.path.insert(0, "$REPO")
from positionbaseddynamics_amd import _ffi
_ffi.check(_ffi.lib.pbdx_debug_stream(0, 1 << 30, 1), "read once")
_ffi.check(_ffi.lib.pbdx_debug_stream(0, 128 << 20, 4), "reread")
PY
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUBBLE_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python /tmp/mall_drv.py > /dev/null 2> $OUT/$tag.log
  python - "$OUT/$tag" <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(dict)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("calib_read_b128","calib_reread_b128"):
            if k in r["Kernel_Name"]: d[r["Counter_Name"]][k]=float(r["Counter_Value"])
for c,v in sorted(d.items()):
    a,b=v.get("calib_read_b128"),v.get("calib_reread_b128")
    print("%-36s read-once(1 GiB) %.6g   reread(8 x 128 MiB) %.6g   ratio %s"%(c,a or 0,b or 0,("%.3f"%(b/a)) if a and b else "n/a"))
PY
  rm -rf $OUT/$tag
done
