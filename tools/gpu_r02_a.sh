#!/bin/bash
# round 2, first GPU visit: new tests, then bench (exact), then the RLC comparison
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/r02_pytest_new.txt 2>&1; echo "pytest new rc=$?" | tee -a gpurun_out/r02_pytest_new.txt
tail -5 gpurun_out/r02_pytest_new.txt
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 512 > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r02_bench_a.json
timeout -k 10 400 python tools/bench_rlc.py --batch-log2 20 --steps 2 > gpurun_out/r02_rlc_a.txt 2> gpurun_out/r02_rlc_a.err; echo "rlc rc=$?"
cat gpurun_out/r02_rlc_a.txt
