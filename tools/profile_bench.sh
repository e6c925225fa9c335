#!/bin/bash
# rocprofv3 kernel trace of one default bench pass -> gpurun_out/prof_* (condensed copies go to profiles/).  Run from the repo root on the GPU box.
# --sync-every 16: the host synchronises every 16 engine calls (round 1's tool crashes came with the host tens of thousands of launches ahead).
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 ${PROF_TIMEOUT:-500} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch-rows --no-big-rows --no-roofline --no-detail --sync-every 16 "$@" > $OUT/run.txt 2>&1)
echo "rocprofv3 rc=$?" >> $OUT/run.txt
T=$(find $OUT -name '*kernel_trace.csv' | head -1); S=$(find $OUT -name '*kernel_stats.csv' | head -1)
python3 tools/summarize_trace.py $T > gpurun_out/prof_by_grid.txt
python3 tools/summarize_stats.py $S > gpurun_out/prof_kernel_stats.txt 2>/dev/null
tail -3 $OUT/run.txt | cut -c1-400
cat gpurun_out/prof_by_grid.txt
find $OUT -name '*.csv' -size +20M -delete    # the raw trace is large; the condensed files are what travels back
