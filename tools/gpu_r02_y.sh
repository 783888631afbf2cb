#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_y.txt 2>&1; rc=$?
tail -3 gpurun_out/r02_pytest_y.txt
[ $rc -eq 0 ] || exit $rc
BN254_MSM_SPLIT=0 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k plonk > gpurun_out/r02_pytest_y2.txt 2>&1; rc=$?
tail -2 gpurun_out/r02_pytest_y2.txt
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
timeout -k 10 300 python tools/bench_plonk.py > gpurun_out/r02_plonk_y.json 2> gpurun_out/r02_plonk_y.err || { tail -5 gpurun_out/r02_plonk_y.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r02_plonk_y.json'));print(d['value'],d['ms_per_step'],d['stages_ms'],d['roofline']['frac'])"
done
