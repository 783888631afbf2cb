#!/bin/bash
# kernel + memcpy timeline of one PlonK batch of 4096 (last of 3 steps) -> gpurun_out/trace_plonk/timeline.txt
set -o pipefail
R=$PWD; O=$R/gpurun_out/trace_plonk; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof -o run -- python3 $R/tools/bench_plonk.py --steps 3 --warmup 1 --cpu-sample 8 > $O/bench.json 2> $O/prof.err || { tail -5 $O/prof.err; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/trace_plonk/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bn254::", "").replace("void ", "")[:40], "q" + r["Queue_Id"]))
for f in glob.glob("gpurun_out/trace_plonk/prof/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), ""))
rows.sort()
# last batch: from the last k_plonk_stage1
idx = [i for i, r in enumerate(rows) if "k_plonk_stage1" in r[2]]
start = idx[-1] - 3 if idx else 0
t0 = rows[start][0]
with open("gpurun_out/trace_plonk/timeline.txt", "w") as out:
    for s, e, n, q in rows[start:]:
        out.write("%-42s %-4s start=%9.3f end=%9.3f dur=%8.3f ms\n" % (n, q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
cat $O/bench.json | tail -1 | cut -c1-400
cat $O/timeline.txt | head -60
