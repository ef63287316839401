#!/bin/bash
# PC sampling of the headline kernel (rocprofv3 beta): where do the waves of persistent_kernel<18,1024> spend their cycles?
export TMPDIR=/tmp
out=gpurun_out/r06b_pcs; mkdir -p $out
METHOD=${METHOD:-stochastic}; UNIT=${UNIT:-cycles}; INTERVAL=${INTERVAL:-65536}
timeout 900 rocprofv3 --pc-sampling-beta-enabled 1 --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INTERVAL --kernel-trace --output-format csv -d /tmp/pcs -- \
  python bench.py --no-cpu-baseline --no-traffic --no-extras --no-roofline --persistent 2 --steps 30 --warmup 5 > $out/log.txt 2>&1
echo "rc=$?" >> $out/log.txt
find /tmp/pcs -type f | xargs ls -la >> $out/log.txt 2>&1
for f in $(find /tmp/pcs -name "*pc_sampling*csv"); do
  head -5 $f > $out/$(basename $f).head
  python - "$f" "$out/$(basename $f).agg" <<'PY'
import csv, sys, collections
f, o = sys.argv[1], sys.argv[2]
rd = csv.DictReader(open(f))
cols = rd.fieldnames
agg = collections.Counter()
n = 0
for r in rd:
    n += 1
    key = tuple(r.get(c, "") for c in cols if c not in ("Sample_Timestamp", "Exec_Mask", "Dispatch_Id", "Correlation_Id", "Wave_In_Group", "Chiplet", "Wave_Id", "Hw_Id", "Workgroup_Id_X", "Workgroup_Id_Y", "Workgroup_Id_Z", "Timestamp"))
    agg[key] += 1
with open(o, "w") as w:
    w.write("# %d samples; columns kept: %s\n" % (n, [c for c in cols]))
    for k, v in agg.most_common(4000):
        w.write("%8d  %s\n" % (v, " | ".join(k)))
PY
done
tail -5 $out/log.txt
