#!/bin/bash
# twelve-lane cooperative kernels: small-batch latency next to the six-lane generation, the GPU suite, PlonK and RLC (fallback) timings
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/bench_small.py > gpurun_out/r02_small_c12.txt 2>&1 || { tail -5 gpurun_out/r02_small_c12.txt; exit 1; }
cat gpurun_out/r02_small_c12.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_c12.txt 2>&1; rc=$?
tail -3 gpurun_out/r02_pytest_c12.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/bench_plonk.py > gpurun_out/r02_plonk_c12.txt 2>&1 || { tail -5 gpurun_out/r02_plonk_c12.txt; exit 1; }
tail -5 gpurun_out/r02_plonk_c12.txt
timeout -k 10 600 python tools/bench_rlc.py --batch-log2 20 --steps 2 --invalid-every 256 > gpurun_out/r02_rlc_c12.txt 2>&1 || { tail -5 gpurun_out/r02_rlc_c12.txt; exit 1; }
tail -5 gpurun_out/r02_rlc_c12.txt
