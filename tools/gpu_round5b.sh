#!/bin/bash
# round 5, second GPU call: parity after the G1 formula change, the comb-affine probe, the bench line with every config, kernel traces (headline, PlonK 262144).
set -o pipefail
R=$PWD; O=$R/gpurun_out/r05b
rm -rf $O; mkdir -p $O
fail() { echo "FAILED: $1"; tail -30 $2; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || fail pytest $O/pytest_gpu.log
tail -2 $O/pytest_gpu.log
timeout -k 10 300 tools/kbench/obj_COMB/kb > $O/comb_affine_probe.txt 2>&1 || fail comb $O/comb_affine_probe.txt
cat $O/comb_affine_probe.txt
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || fail bench_default $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05b/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("valu_whole_path",{}).get("frac"))
for k,v in d["configs"].items():
    print(k, v.get("value"), v.get("ms_per_step"), (v.get("valu_whole_path") or {}).get("frac"), (v.get("host_buffers") or {}).get("value"))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rlc --no-configs > $O/prof_bench.json 2> $O/prof.err || fail rocprof $O/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_plonk256k -o run -- python3 $R/tools/bench_plonk.py --batch 262144 --steps 2 --warmup 1 --cpu-sample 0 --no-in-flight > $O/prof_plonk256k.json 2> $O/prof_plonk256k.err || fail "rocprof plonk 256k" $O/prof_plonk256k.err
cd $R
find $O -name "*kernel_trace.csv" -size +30M -delete
ls $O/prof $O/prof_plonk256k
echo "round 5b done"
