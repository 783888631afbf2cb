#!/bin/bash
# round 3, first GPU call: batch-size sweep around the strong-scaling shard sizes (2^17 .. 2^20), stream counts, instruction-fetch counters
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03diag
rm -rf $O; mkdir -p $O
for lg in 16 17 18 19 20; do
  python bench.py --batch-log2 $lg --steps 5 --warmup 1 --no-cpu-baseline --no-rlc > $O/sweep_s2_$lg.json 2> $O/sweep_s2_$lg.err || { echo "sweep $lg failed"; tail -5 $O/sweep_s2_$lg.err; exit 1; }
  python - <<PY
import json; d=json.load(open("$O/sweep_s2_$lg.json")); print("streams=2 2^$lg", round(d["value"]), round(d["ms_per_step"],3), d["roofline"]["frac"])
PY
done
for s in 1 4; do for lg in 17 19; do
  BN254_STREAMS=$s python bench.py --batch-log2 $lg --steps 5 --warmup 1 --no-cpu-baseline --no-rlc > $O/sweep_s${s}_$lg.json 2> $O/sweep_s${s}_$lg.err || { echo "sweep s$s $lg failed"; exit 1; }
  python - <<PY
import json; d=json.load(open("$O/sweep_s${s}_$lg.json")); print("streams=$s 2^$lg", round(d["value"]), round(d["ms_per_step"],3))
PY
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/list_avail.txt 2>&1 || echo "list-avail failed"
grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_]*\|TCP_[A-Z_0-9]*ICACHE[A-Z_]*" $O/list_avail.txt | sort -u | tr '\n' ' '
echo
BN254_STREAMS=1 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_ifetch -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rlc --batch-log2 18 > $O/pmc_ifetch.json 2> $O/pmc_ifetch.err || { echo "pmc ifetch failed"; tail -5 $O/pmc_ifetch.err; }
echo "pmc ifetch done"
BN254_STREAMS=1 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/pmc_icache -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rlc --batch-log2 18 > $O/pmc_icache.json 2> $O/pmc_icache.err || { echo "pmc icache failed"; tail -5 $O/pmc_icache.err; }
echo "pmc icache done"
find $O -name "*kernel_trace.csv" -size +30M -delete
find $O -name "*.csv" -size +40M -delete
ls -la $O/pmc_ifetch/* 2>/dev/null | head
