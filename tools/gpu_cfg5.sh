#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python bench.py --n-public 1024 --batch-log2 12 --steps 5 --warmup 1 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err || { tail -20 gpurun_out/bench_cfg5.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_cfg5.json'))
print("cfg5 value %.0f proofs/s  %.2f ms/step" % (d['value'], d['ms_per_step'])); print(d['roofline']); print(d['cpu_baseline'])
for k,v in list(d['kernels_ms'].items())[:6]: print("  %-22s %4d  %8.3f ms" % (k, v['launches'], v['total_ms']))
PY
bash tools/gpu_quick.sh
