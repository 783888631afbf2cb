#!/bin/bash
# round 5, third GPU call: parity, kbench (alias-tolerant third form of the Fp12 product; occupancy of the comb loop), the bench line.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r05c
rm -rf $O; mkdir -p $O
fail() { echo "FAILED: $1"; tail -30 $2; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || fail pytest $O/pytest_gpu.log
tail -2 $O/pytest_gpu.log
for r in 1 2; do for v in MUL_L MUL_T3E MUL_T3F; do timeout -k 10 120 tools/kbench/obj_$v/kb >> $O/kbench_f12_mul.txt 2>&1 || fail "kbench $v" $O/kbench_f12_mul.txt; done; done
cat $O/kbench_f12_mul.txt
for w in 3 4; do echo "launch bounds (256, $w)" >> $O/comb_occupancy.txt; timeout -k 10 300 tools/kbench/obj_COMB$w/kb 2>&1 | grep today >> $O/comb_occupancy.txt || fail comb $O/comb_occupancy.txt; done
cat $O/comb_occupancy.txt
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || fail bench_default $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05c/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("valu_whole_path",{}).get("frac"))
for k,v in d["configs"].items():
    print(k, v.get("value"), v.get("ms_per_step"), (v.get("valu_whole_path") or {}).get("frac"), (v.get("host_buffers") or {}).get("value"))
PY
echo "round 5c done"
