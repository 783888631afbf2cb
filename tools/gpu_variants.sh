#!/bin/bash
# The GPU suite under the non-default execution settings: lane kernels for small batches, 1 and 4 sub-batch streams, the step kernels, the PlonK plans (chains of small
# passes, one pass, a lane budget that forces unsplit rows / rows of fixed windows only), the host stages, two RLC group shapes -- and on LIBRARY VARIANTS built beside the
# product by tools/build_variants.sh (the Fr product of the device stages in its 64-bit form, inlined by force, without the register barrier of FrCtx::from_be32: DESIGN.md section 9).
# Run on the GPU box: gpurun -- bash tools/gpu_variants.sh      (ONLY_RLC=1: the two RLC shapes only; ONLY_LIBS=1: the library variants only)
set -o pipefail
mkdir -p gpurun_out
FAILED=0
# a failing variant is recorded and the next one runs (a GPU fault or a timeout ends the script: no further GPU step after one)
run() { tag=$1; shift; env "$@" timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/variants_pytest_$tag.txt 2>&1; rc=$?; echo "$tag rc=$rc $(tail -1 gpurun_out/variants_pytest_$tag.txt)"; grep -q -i "access fault" gpurun_out/variants_pytest_$tag.txt && exit 1; [ $rc -ge 124 ] && exit $rc; [ $rc -eq 0 ] || FAILED=1; }
if [ -z "$ONLY_RLC" ] && [ -z "$ONLY_LIBS" ]; then
run nocoop BN254_COOP=0
run streams1 BN254_STREAMS=1
run streams4 BN254_STREAMS=4
run steps0 BN254_MILLER_RUN_STEPS=0
run plonk_piece BN254_PLONK_PIECE=700 BN254_PLONK_WORKERS=3 BN254_PLONK_BIG_FROM=1000000000
run plonk_onepass BN254_PLONK_BIG_FROM=1 BN254_PLONK_BIG_PIECE=65536
run plonk_onepass_max BN254_PLONK_BIG_FROM=1 BN254_PLONK_BIG_PIECE=262144
run msm_budget_small BN254_MSM_LANE_BUDGET=4096
run msm_budget_large BN254_MSM_LANE_BUDGET=1048576
run coop_fixed_off BN254_COOP_FIXED_MAX=0
run plonk_host BN254_PLONK_HOST=1
run tables_host BN254_TABLES_HOST=1
fi
if [ -z "$ONLY_LIBS" ]; then
run rlc_g3s1 BN254_RLC_GROUP_LOG2=3 BN254_RLC_SHARE_LOG2=1 BN254_RLC_SHARE_MIN_LANES=1
run rlc_g8s3 BN254_RLC_GROUP_LOG2=8 BN254_RLC_SHARE_LOG2=3 BN254_RLC_SHARE_MIN_LANES=1
fi
if [ -z "$ONLY_RLC" ]; then
for v in frmul64 frmul_inline_barrier frmul_outofline_nobarrier; do
  lib=$PWD/tools/exp/libbn254_$v.so
  [ -f $lib ] || { echo "$v: not built (tools/build_variants.sh)"; continue; }
  tag=lib_$v
  BN254_LIB_PATH=$lib timeout -k 10 900 python -m pytest tests -x -q -m gpu -k plonk > gpurun_out/variants_pytest_$tag.txt 2>&1; rc=$?
  echo "$tag rc=$rc $(tail -1 gpurun_out/variants_pytest_$tag.txt)"; [ $rc -ge 124 ] && exit $rc; [ $rc -eq 0 ] || FAILED=1
done
fi
if [ -z "$ONLY_LIBS" ]; then
timeout -k 10 900 python tools/gpu_fuzz.py --cases 200 > gpurun_out/variants_fuzz.txt 2>&1 || { tail -5 gpurun_out/variants_fuzz.txt; exit 1; }
tail -2 gpurun_out/variants_fuzz.txt
fi
exit $FAILED
