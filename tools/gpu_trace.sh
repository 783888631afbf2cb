#!/bin/bash
# kernel trace of the bench command (2 timed steps) -> gpurun_out/trace/
set -o pipefail
R=$PWD; O=$R/gpurun_out/trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rlc --no-configs > $O/prof_bench.json 2> $O/prof.err || { tail -5 $O/prof.err; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, collections
tr = glob.glob("gpurun_out/trace/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(tr)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# the last batch: take kernels after the last k_g16_prepare pair
prep = [i for i, r in enumerate(rows) if "k_g16_prepare" in r["Kernel_Name"]]
start = prep[-2] if len(prep) >= 2 else 0
out = open("gpurun_out/trace/timeline.txt", "w")
for r in rows[start:]:
    name = r["Kernel_Name"].split("(")[0].replace("bn254::", "").replace("void ", "")
    out.write("%-28s q=%s start=%9.3f end=%9.3f dur=%8.3f\n" % (name[:28], r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
out.close()
PY
find $O -name "*kernel_trace.csv" -size +20M -delete
head -40 $O/timeline.txt
