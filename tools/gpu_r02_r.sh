#!/bin/bash
# PMC + kernel trace of the twelve-lane cooperative kernel at batch 4096
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02r
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small -o run -- python3 $R/bench.py --batch-log2 12 --steps 20 --warmup 2 --no-cpu-baseline --no-rlc > $O/small.json 2> $O/small.err || { tail $O/small.err; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_small -o run -- python3 $R/bench.py --batch-log2 12 --steps 3 --warmup 1 --no-cpu-baseline --no-rlc > $O/pmc_small.json 2> $O/pmc_small.err || { tail $O/pmc_small.err; exit 1; }
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_small2 -o run -- python3 $R/bench.py --batch-log2 12 --steps 3 --warmup 1 --no-cpu-baseline --no-rlc > $O/pmc_small2.json 2> $O/pmc_small2.err || { tail $O/pmc_small2.err; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_small", "pmc_small2"):
    for p in glob.glob("gpurun_out/r02r/%s/**/*counter_collection.csv" % d, recursive=True):
        a = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            a[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in a.items():
            if "coop" in k:
                print(k, {c: round(x / cnt[(k, c)]) for c, x in v.items()})
PY
grep -h coop $O/prof_small/*/*kernel_stats.csv | head
cut -c1-300 $O/small.json
