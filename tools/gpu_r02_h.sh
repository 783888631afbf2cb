#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_h.txt 2>&1; rc=$?; echo "pytest gpu rc=$rc"; tail -4 gpurun_out/r02_pytest_h.txt
grep -q -i "access fault" gpurun_out/r02_pytest_h.txt && exit 1
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 1024 --host-buffers > gpurun_out/r02_bench_h.json 2> gpurun_out/r02_bench_h.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_bench_h.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(d["phases_ms"]); print({k:v for k,v in d["kernels_ms"].items()})
print(d["roofline"]); print(d.get("valu_whole_path")); print(d.get("host_buffers")); print(d.get("cpu_baseline"))
PY
