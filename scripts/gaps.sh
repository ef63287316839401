# kernel-to-kernel gaps of the headline bench under rocprofv3 --kernel-trace
export TMPDIR=/tmp; REPO=$PWD; OUT=$PWD/gpurun_out/gaps; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o g -- python $REPO/bench.py --no-traffic --no-cpu-baseline --no-roofline --steps 20 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/log.txt
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/kernel_gaps.py $f | cut -c1-220
rm -rf $OUT/t
