#!/bin/bash
# round 5, robustness calls (after the G1 formula and Fp12 product changes):  PART=fuzz | plonk | variants
set -o pipefail
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
PART=${PART:-fuzz}
if [ "$PART" = fuzz ]; then
  timeout -k 10 1100 python tools/gpu_fuzz.py --cases 500 --seed 5 > $O/fuzz.txt 2>&1 || { tail -5 $O/fuzz.txt; exit 1; }
  tail -1 $O/fuzz.txt
fi
if [ "$PART" = plonk ]; then
  timeout -k 10 500 python tools/gpu_fuzz_plonk.py --cases 800 > $O/fuzz_plonk.txt 2>&1 || { tail -5 $O/fuzz_plonk.txt; exit 1; }
  tail -1 $O/fuzz_plonk.txt
  timeout -k 10 560 python tools/gpu_soak_mixed.py > $O/soak_mixed.txt 2>&1 || { tail -5 $O/soak_mixed.txt; exit 1; }
  tail -1 $O/soak_mixed.txt
fi
if [ "$PART" = variants ]; then
  ONLY_LIBS= bash tools/gpu_variants.sh > $O/variants.txt 2>&1; rc=$?
  cat $O/variants.txt
  exit $rc
fi
