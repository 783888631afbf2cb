#!/bin/bash
# round 5: does instruction fetch limit the long straight-line kernels?  (k_miller_run's loop body is ~0.5 MB of code against a 64 KB instruction cache.)
# Two counter passes over one batch of 2^18 proofs on one stream: instruction-cache requests / hits / misses, fetch latency.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r05i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BN254_STREAMS=1 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL --output-format csv -d $O/pmc_icache -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rlc --no-configs --batch-log2 18 > $O/pmc_icache.json 2> $O/pmc_icache.err || { tail -5 $O/pmc_icache.err; exit 1; }
BN254_STREAMS=1 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_ifetch -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rlc --no-configs --batch-log2 18 > $O/pmc_ifetch.json 2> $O/pmc_ifetch.err || { tail -5 $O/pmc_ifetch.err; exit 1; }
echo done
