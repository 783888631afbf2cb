#!/bin/bash
# quick GPU check: parity tests + bench without the CPU baseline
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
python bench.py --no-cpu-baseline --steps 2 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err || { tail -20 gpurun_out/bench_quick.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_quick.json'))
print("value %.0f proofs/s  %.1f ms/step" % (d['value'], d['ms_per_step']))
print("dominant", d['roofline']['kernel'], d['roofline']['avg_launch_ms'], "frac", round(d['roofline']['frac'],3))
for k,v in d['kernels_ms'].items(): print("  %-22s %4d  %8.3f ms" % (k, v['launches'], v['total_ms']))
print(d['phases_ms'])
PY
