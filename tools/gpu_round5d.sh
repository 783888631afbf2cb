#!/bin/bash
# round 5, fourth GPU call: parity with the re-ordered Fp12 product, the bench line, then the wrong-challenge reproduction (tools/gpu_repro.sh).
set -o pipefail
R=$PWD; O=$R/gpurun_out/r05d
rm -rf $O; mkdir -p $O
fail() { echo "FAILED: $1"; tail -30 $2; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || fail pytest $O/pytest_gpu.log
tail -2 $O/pytest_gpu.log
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || fail bench_default $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05d/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("valu_whole_path",{}).get("frac"))
print(d.get("kernels_ms"))
for k,v in d["configs"].items():
    print(k, v.get("value"), v.get("ms_per_step"), (v.get("valu_whole_path") or {}).get("frac"), (v.get("host_buffers") or {}).get("value"))
PY
timeout -k 10 600 bash tools/gpu_repro.sh || fail repro gpurun_out/repro/repro_kernels.txt
echo "round 5d done"
