#!/bin/bash
# Library variants for tools/gpu_variants.sh (built here, they travel to the GPU box with the snapshot; tools/exp/ is not tracked): the Fr / Fp product of the PlonK
# device stages (csrc/bn254_plonk.hpp::FrCtx::mul) in its 64-bit form, inlined by force WITH the register barrier of from_be32, and out of line WITHOUT the barrier --
# each of the two workarounds of DESIGN.md section 9 alone.  (Both off -- -DBN254_FR_MUL_INLINE=1 -DBN254_FR_NO_BARRIER -- is the failing combination.)
set -e
cd "$(dirname "$0")/../snark-bn254-verifier_amd/csrc"
mkdir -p ../../tools/exp
make -j4 BUILD=build_frmul64 OUT=../../tools/exp/libbn254_frmul64.so EXTRA=-DBN254_FR_MUL_FORM=64 > ../../tools/exp/build_frmul64.log 2>&1 &
make -j4 BUILD=build_frinl OUT=../../tools/exp/libbn254_frmul_inline_barrier.so EXTRA=-DBN254_FR_MUL_INLINE=1 > ../../tools/exp/build_frinl.log 2>&1 &
wait
make -j8 BUILD=build_frnob OUT=../../tools/exp/libbn254_frmul_outofline_nobarrier.so EXTRA=-DBN254_FR_NO_BARRIER > ../../tools/exp/build_frnob.log 2>&1
ls -la ../../tools/exp/libbn254_fr*.so
