#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03exp1; rm -rf $O; mkdir -p $O
for s in 1 2 3; do
  BN254_STREAMS=$s python bench.py --steps 5 --warmup 2 --no-configs --no-cpu-baseline --no-rlc > $O/streams_$s.json 2> $O/streams_$s.err || { tail -3 $O/streams_$s.err; exit 1; }
  python -c "
import json; d=json.load(open('$O/streams_$s.json')); print('streams=$s', round(d['value']), round(d['ms_per_step'],2), 'frac', d['roofline'].get('frac'))"
done
cd /tmp && export TMPDIR=/tmp
BN254_STREAMS=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_run -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rlc --no-configs --batch-log2 18 > $O/pmc_run.json 2> $O/pmc_run.err || { tail -3 $O/pmc_run.err; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r03exp1/pmc_run/**/*counter_collection.csv", recursive=True)[0]
a = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
rows = list(csv.DictReader(open(f))); first = rows[0]["Counter_Name"]
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("bn254::", "").replace("void ", "").split("<")[0].strip()
    a[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == first: cnt[k] += 1
for k in ("k_miller_run", "k_f12_mul", "k_f12_cyclo_sqr_n"):
    print(k, cnt[k], {c: round(v / cnt[k]) for c, v in a[k].items()})
PY
