#!/bin/bash
# PMC + kernel-trace evidence for the round-2 kernels: cooperative small-batch path (batch 4096) and the RLC mode (batch 2^20, all valid)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02m
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small -o run -- python3 $R/bench.py --batch-log2 12 --steps 20 --warmup 2 --no-cpu-baseline > $O/small.json 2> $O/small.err || { tail $O/small.err; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_small -o run -- python3 $R/bench.py --batch-log2 12 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_small.json 2> $O/pmc_small.err || { tail $O/pmc_small.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rlc -o run -- python3 $R/tools/bench_rlc.py --batch-log2 20 --steps 2 --invalid-every 0 > $O/rlc.txt 2> $O/rlc.err || { tail $O/rlc.err; exit 1; }
find $O -name "*kernel_trace.csv" -size +30M -delete
cd $R
find $O -name "*stats.csv" -o -name "*counter_collection.csv" | head
cat $O/rlc.txt
