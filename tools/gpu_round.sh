#!/bin/bash
# One GPU-box session: parity suite, dot micro-benchmark, short bench.  Usage: tools/gpu_round.sh [pytest-k-expr]
mkdir -p gpurun_out
python __graft_entry__.py build || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -q -x ${1:+-k "$1"} > gpurun_out/gpu_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/gpu_tests.txt
tail -15 gpurun_out/gpu_tests.txt
