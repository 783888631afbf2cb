#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_v.txt 2>&1; rc=$?
tail -3 gpurun_out/r02_pytest_v.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/bench_small.py > gpurun_out/r02_small_v.txt 2>&1 || { tail -5 gpurun_out/r02_small_v.txt; exit 1; }
cat gpurun_out/r02_small_v.txt
timeout -k 10 300 python tools/bench_plonk.py > gpurun_out/r02_plonk_v.json 2> gpurun_out/r02_plonk_v.err || { tail -5 gpurun_out/r02_plonk_v.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r02_plonk_v.json'));print(d['value'],d['ms_per_step'],d['stages_ms'])"
timeout -k 10 600 python bench.py --no-cpu-baseline --no-rlc > gpurun_out/r02_bench_v.json 2> gpurun_out/r02_bench_v.err || { tail -5 gpurun_out/r02_bench_v.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r02_bench_v.json'));print(d['value'],d['ms_per_step'],d['kernels_ms'])"
timeout -k 10 600 python bench.py --n-public 1024 --batch-log2 12 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r02_cfg5_v.json 2> gpurun_out/r02_cfg5_v.err || { tail -5 gpurun_out/r02_cfg5_v.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r02_cfg5_v.json'));print(d['value'],d['ms_per_step'])"
