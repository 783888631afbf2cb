#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_g.txt 2>&1; rc=$?; echo "pytest gpu rc=$rc"; tail -8 gpurun_out/r02_pytest_g.txt
grep -q -i "access fault" gpurun_out/r02_pytest_g.txt && exit 1
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_small.py > gpurun_out/r02_small_coop.txt 2>&1; echo "small coop rc=$?"; cat gpurun_out/r02_small_coop.txt
BN254_COOP=0 timeout -k 10 300 python tools/bench_small.py > gpurun_out/r02_small_lane.txt 2>&1; echo "small lane rc=$?"; cat gpurun_out/r02_small_lane.txt
