#!/bin/bash
# GPU suite + RLC sweep after the adaptive bypass
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_s.txt 2>&1; rc=$?
tail -5 gpurun_out/r02_pytest_s.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/bench_rlc.py --batch-log2 20 --steps 9 --invalid-every 0,256,16 > gpurun_out/r02_rlc_s.txt 2>&1 || { tail -5 gpurun_out/r02_rlc_s.txt; exit 1; }
cat gpurun_out/r02_rlc_s.txt
