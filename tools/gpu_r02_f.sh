#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "plonk" > gpurun_out/r02_pytest_f.txt 2>&1; rc=$?; echo "pytest plonk rc=$rc"; tail -5 gpurun_out/r02_pytest_f.txt
grep -q -i "fault" gpurun_out/r02_pytest_f.txt && exit 1
[ $rc -eq 0 ] || exit $rc
BN254_PLONK_TIMING=1 timeout -k 10 300 python tools/bench_plonk.py > gpurun_out/r02_plonk_coop.json 2> gpurun_out/r02_plonk_coop.err; echo "plonk bench rc=$?"; tail -c 600 gpurun_out/r02_plonk_coop.json; tail -4 gpurun_out/r02_plonk_coop.err
BN254_COOP=0 BN254_PLONK_TIMING=1 timeout -k 10 300 python tools/bench_plonk.py > gpurun_out/r02_plonk_nocoop.json 2> gpurun_out/r02_plonk_nocoop.err; echo "plonk bench (no coop) rc=$?"; tail -c 600 gpurun_out/r02_plonk_nocoop.json; tail -4 gpurun_out/r02_plonk_nocoop.err
