#!/bin/bash
# Counter passes for profiles/r03_pmc_b256.json (run on the GPU box from the repo root).  One rocprofv3 --pmc pass per counter group (FETCH_SIZE
# and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace domains off, the program directly after `--`.
# The bench runs eager (--no-graph), with a stream synchronise after every engine call (--sync-every 1: the profiler serialises every
# dispatch, and without it the host runs tens of thousands of launches ahead of the device), at the key range of the middle of a
# default run (--prefill-len 768) for a few decode steps.
set -o pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-batch-rows --no-big-rows --no-detail --no-fetch --no-graph --sync-every 1 --prefill-len 768 --decode-steps ${PMC_STEPS:-6} --prefill-pack 8"
REGEX='k_attn_decode|k_attn_tile|k_gemm_ring|k_gemm<'
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf $OUT/$tag
  (cd /tmp && timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $grp --kernel-include-regex "$REGEX" --output-format csv -d $OUT/$tag -- python3 $REPO/bench.py $ARGS > $OUT/$tag.txt 2>&1)
  echo "pmc $tag rc=$?" | tee -a $OUT/$tag.txt
  tail -2 $OUT/$tag.txt | cut -c1-300
done
python3 tools/pmc_summary.py $OUT > gpurun_out/r03_pmc_b256.json && head -c 1500 gpurun_out/r03_pmc_b256.json
