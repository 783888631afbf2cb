#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "many_public or more_public or 4096 or precedence" > gpurun_out/r02_pytest_i.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r02_pytest_i.txt
grep -q -i "access fault" gpurun_out/r02_pytest_i.txt && exit 1
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --n-public 1024 --batch-log2 12 --steps 5 --warmup 1 --cpu-sample 64 > gpurun_out/r02_cfg5_coop.json 2> gpurun_out/r02_cfg5_coop.err; echo "cfg5 rc=$?"; cut -c1-330 gpurun_out/r02_cfg5_coop.json
