#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_gpu.txt 2>&1; rc=$?; echo "pytest gpu rc=$rc"
tail -8 gpurun_out/r02_pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/bench_rlc.py --batch-log2 20 --steps 2 --invalid-every 256,16 > gpurun_out/r02_rlc_b.txt 2> gpurun_out/r02_rlc_b.err; echo "rlc rc=$?"
cat gpurun_out/r02_rlc_b.txt
