#!/bin/bash
# rocprofv3 kernel trace of tools/bench_variants.py for one family -> gpurun_out/prof_<family>_by_grid.txt (condensed copy goes to profiles/).
# Usage (GPU box, repo root): bash tools/profile_variants.sh gemma2|mixtral [extra bench_variants args]
FAM=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$FAM
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/tools/bench_variants.py $FAM --steps 16 "$@" > $OUT/run.txt 2>&1)
echo "rocprofv3 rc=$?" >> $OUT/run.txt
T=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py $T > gpurun_out/prof_${FAM}_by_grid.txt
tail -2 $OUT/run.txt | cut -c1-300
head -25 gpurun_out/prof_${FAM}_by_grid.txt | cut -c1-150
find $OUT -name '*.csv' -size +20M -delete
