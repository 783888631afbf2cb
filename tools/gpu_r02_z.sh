#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k plonk > gpurun_out/r02_pytest_z.txt 2>&1; rc=$?
tail -2 gpurun_out/r02_pytest_z.txt
[ $rc -eq 0 ] || exit $rc
for b in 1024 2048 4096; do
for sp in 0 default; do
if [ $sp = default ]; then unset BN254_MSM_SPLIT; else export BN254_MSM_SPLIT=$sp; fi
timeout -k 10 300 python tools/bench_plonk.py --batch $b --cpu-sample 8 > gpurun_out/r02_plonk_z.json 2> gpurun_out/r02_plonk_z.err || { tail -5 gpurun_out/r02_plonk_z.err; exit 1; }
python -c "import json,sys;d=json.load(open('gpurun_out/r02_plonk_z.json'));print(sys.argv[1],sys.argv[2],round(d['value']),round(d['ms_per_step'],3),d['stages_ms']['k_g1_scalar_mul_stage2'],d['stages_ms']['digest_msm_kernels'])" $b $sp
done; done
