#!/bin/bash
# 1024 public inputs, batch 4096 (BASELINE configs[4]) under the profiler: kernel trace, then SQ counters
set -o pipefail
R=$PWD; O=$R/gpurun_out/cfg5; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --n-public 1024 --batch-log2 12 --steps 5 --warmup 1 --no-cpu-baseline --no-rlc > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_INT64 --output-format csv -d $O/pmc -o run -- python3 $R/bench.py --n-public 1024 --batch-log2 12 --steps 1 --warmup 0 --no-cpu-baseline --no-rlc > $O/pmc.json 2> $O/pmc.err || { tail $O/pmc.err; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, collections
p = glob.glob("gpurun_out/cfg5/prof/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(p)))[:6]: print(r["Name"][:50], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"])
p = glob.glob("gpurun_out/cfg5/pmc/*counter_collection.csv")[0]
a = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); grid = {}
for r in csv.DictReader(open(p)):
    k = r["Kernel_Name"].split("(")[0].replace("bn254::", ""); a[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1; grid[k] = int(r["Grid_Size"]) // 64
for k, v in a.items():
    if k.startswith("k_g16_msm") or "coop" in k:
        wc = v["SQ_WAVE_CYCLES"] or 1
        print(k, "waves", grid[k], "valu_active", round(v["SQ_ACTIVE_INST_VALU"] / wc, 3), "wait_any", round(v["SQ_WAIT_ANY"] / wc, 3), "valu/wave", round(v["SQ_INSTS_VALU"] / cnt[(k, "SQ_INSTS_VALU")] / grid[k]), "int64/wave", round(v["SQ_INSTS_VALU_INT64"] / cnt[(k, "SQ_INSTS_VALU_INT64")] / grid[k]))
PY
