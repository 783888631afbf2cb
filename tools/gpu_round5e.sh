#!/bin/bash
# round 5, fifth GPU call: the PlonK tests with self-test v2, then the wrong-challenge builds: does the self-test refuse them, which optimisation settings keep the defect.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r05e
rm -rf $O; mkdir -p $O
fail() { echo "FAILED: $1"; tail -30 $2; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "plonk or Plonk or smoke or cpp" > $O/pytest_plonk.log 2>&1 || fail pytest $O/pytest_plonk.log
tail -2 $O/pytest_plonk.log
timeout -k 10 600 bash tools/gpu_repro.sh || true
timeout -k 10 600 bash tools/gpu_repro2.sh || true
echo "round 5e done"
