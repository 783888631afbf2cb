#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/r02_pytest_l.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r02_pytest_l.txt
grep -q -i "access fault" gpurun_out/r02_pytest_l.txt && exit 1
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/bench_rlc.py --batch-log2 20 --steps 3 --invalid-every 0,256,16 > gpurun_out/r02_rlc_glv.txt 2> gpurun_out/r02_rlc_glv.err; echo "rlc rc=$?"; cat gpurun_out/r02_rlc_glv.txt
