#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_u.txt 2>&1; rc=$?
tail -3 gpurun_out/r02_pytest_u.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --batch-log2 12 --steps 20 --warmup 2 --no-cpu-baseline --no-rlc > gpurun_out/r02_b12_u.json 2> gpurun_out/r02_b12_u.err || { tail -5 gpurun_out/r02_b12_u.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r02_b12_u.json'));print(d['value'],d['roofline'],d['valu_whole_path'])"
timeout -k 10 300 python tools/bench_plonk.py > gpurun_out/r02_plonk_u.json 2> gpurun_out/r02_plonk_u.err || { tail -5 gpurun_out/r02_plonk_u.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r02_plonk_u.json'));print(d['value'],d['roofline'],d['pairing_check'],d['stages_ms'])"
