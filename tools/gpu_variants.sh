#!/bin/bash
# the GPU suite under the non-default execution settings (one-proof-per-lane kernels for small batches, six-lane cooperative kernels, one lane per PlonK scalar multiplication, 1 and 4 sub-batch streams, RLC group/share sizes)
set -o pipefail
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/variants_pytest_$tag.txt 2>&1; rc=$?; echo "$tag rc=$rc $(tail -1 gpurun_out/variants_pytest_$tag.txt)"; grep -q -i "access fault" gpurun_out/variants_pytest_$tag.txt && exit 1; [ $rc -eq 0 ] || exit $rc; }
if [ -z "$ONLY_RLC" ]; then
run nocoop BN254_COOP=0
run streams1 BN254_STREAMS=1
run streams4 BN254_STREAMS=4
run msm_plain BN254_MSM_SPLIT=0 BN254_MSM_W2=0
run msm_split BN254_MSM_SPLIT=1
run steps0 BN254_MILLER_RUN_STEPS=0
run plonk_piece BN254_PLONK_PIECE=700 BN254_PLONK_WORKERS=3
run plonk_host BN254_PLONK_HOST=1
fi
run rlc_g3s1 BN254_RLC_GROUP_LOG2=3 BN254_RLC_SHARE_LOG2=1 BN254_RLC_SHARE_MIN_LANES=1
run rlc_g8s3 BN254_RLC_GROUP_LOG2=8 BN254_RLC_SHARE_LOG2=3 BN254_RLC_SHARE_MIN_LANES=1
timeout -k 10 900 python tools/gpu_fuzz.py --cases 200 > gpurun_out/variants_fuzz.txt 2>&1 || { tail -5 gpurun_out/variants_fuzz.txt; exit 1; }
tail -2 gpurun_out/variants_fuzz.txt
