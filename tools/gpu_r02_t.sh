#!/bin/bash
# 64-bit integer VALU instruction counters (SQ_INSTS_VALU_INT64 / INT32): lane kernels (batch 2^18, one stream) for calibration against the
# static counts of tools/count_mads.py, then the cooperative kernels (Groth16 batch 4096, PlonK batch 4096)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02t
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BN254_STREAMS=1 rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_lane -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rlc --batch-log2 18 > $O/pmc_lane.json 2> $O/pmc_lane.err || { tail $O/pmc_lane.err; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_coop -o run -- python3 $R/bench.py --batch-log2 12 --steps 3 --warmup 1 --no-cpu-baseline --no-rlc > $O/pmc_coop.json 2> $O/pmc_coop.err || { tail $O/pmc_coop.err; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_plonk -o run -- python3 $R/tools/bench_plonk.py --steps 2 > $O/pmc_plonk.json 2> $O/pmc_plonk.err || { tail $O/pmc_plonk.err; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_lane", "pmc_coop", "pmc_plonk"):
    for p in glob.glob("gpurun_out/r02t/%s/**/*counter_collection.csv" % d, recursive=True):
        a = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); grid = {}
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].replace("bn254::", "").replace("void ", "")
            a[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1; grid[k] = int(r["Grid_Size"]) // 64
        for k, v in a.items():
            if k.startswith("k_"):
                print(d, k, "waves", grid[k], {c: round(x / cnt[(k, c)] / grid[k], 1) for c, x in v.items()})
PY
