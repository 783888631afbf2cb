#!/bin/bash
# kernel traces of the exact and the RLC pipelines
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
rm -rf gpurun_out/prof_exact gpurun_out/prof_rlc
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_exact -o exact --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_prof_exact_bench.json 2> gpurun_out/r02_prof_exact.err; echo "exact rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rlc -o rlc --output-format csv -- python3 tools/bench_rlc.py --batch-log2 20 --steps 2 --invalid-every 0 > gpurun_out/r02_prof_rlc.txt 2> gpurun_out/r02_prof_rlc.err; echo "rlc rc=$?"
find gpurun_out/prof_exact gpurun_out/prof_rlc -name "*kernel_stats.csv" | head
# keep only the small stats files (the traces are large)
find gpurun_out/prof_exact gpurun_out/prof_rlc -name "*kernel_trace.csv" -size +20M -delete
cat gpurun_out/r02_prof_rlc.txt
