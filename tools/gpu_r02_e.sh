#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/r02_pytest_e.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02_pytest_e.txt
[ $rc -eq 0 ] || exit $rc
run() { tag=$1; shift; env "$@" timeout -k 10 300 python tools/bench_rlc.py --batch-log2 20 --steps 2 --invalid-every 0 > gpurun_out/r02_rlc_$tag.txt 2> gpurun_out/r02_rlc_$tag.err; echo "$tag rc=$?"; cat gpurun_out/r02_rlc_$tag.txt; }
run s2g5 BN254_RLC_SHARE_LOG2=2
run s3g5 BN254_RLC_SHARE_LOG2=3 BN254_RLC_SHARE_MIN_LANES=65536
run s2g6 BN254_RLC_SHARE_LOG2=2 BN254_RLC_GROUP_LOG2=6
run s2g7 BN254_RLC_SHARE_LOG2=2 BN254_RLC_GROUP_LOG2=7
