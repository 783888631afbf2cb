#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/r02_pytest_d.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02_pytest_d.txt
[ $rc -eq 0 ] || exit $rc
for sh in 2 1 0; do
  BN254_RLC_SHARE_LOG2=$sh timeout -k 10 300 python tools/bench_rlc.py --batch-log2 20 --steps 2 --invalid-every 0 > gpurun_out/r02_rlc_share$sh.txt 2> gpurun_out/r02_rlc_share$sh.err; echo "share $sh rc=$?"; cat gpurun_out/r02_rlc_share$sh.txt
done
