#!/bin/bash
# one GPU call: parity tests, bench, rocprofv3 kernel trace of the same bench command (outputs under gpurun_out/)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cut -c1-1500 gpurun_out/bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err || { tail -20 $R/gpurun_out/prof.err; exit 1; }
cd $R
ls gpurun_out/prof
cut -c1-600 gpurun_out/prof_bench.json
