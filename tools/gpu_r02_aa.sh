#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_aa.txt 2>&1; rc=$?
tail -3 gpurun_out/r02_pytest_aa.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/bench_single.py > gpurun_out/r02_single.json 2> gpurun_out/r02_single.err || { tail -5 gpurun_out/r02_single.err; exit 1; }
cat gpurun_out/r02_single.json
BN254_KEY_CACHE=0 timeout -k 10 400 python tools/bench_single.py 2>/dev/null | tail -1
