#!/bin/bash
# round 3, one GPU call: parity tests, smoke, the default bench line (headline + `configs`), the driver's command shape under torch.distributed.run,
# rocprofv3 kernel trace of the bench command, PMC passes (SQ activity, instruction fetch / I-cache, HBM traffic), batch-size sweep of the
# strong-scaling shard sizes.  Outputs under gpurun_out/r03/; tools/summarize_r03.py turns them into profiles/r03_*
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03
rm -rf $O; mkdir -p $O
fail() { echo "FAILED: $1"; tail -20 $2; exit 1; }
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || fail pytest $O/pytest_gpu.log
tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || fail smoke $O/smoke.log
tail -1 $O/smoke.log
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || fail bench $O/bench.err
cut -c1-300 $O/bench.json
# the multi-rank code path with one rank (RCCL group, all_gather, max-reduce): the shape the driver launches for N > 1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-rlc > $O/bench_torchrun.json 2> $O/bench_torchrun.err || fail torchrun $O/bench_torchrun.err
cut -c1-200 $O/bench_torchrun.json
for lg in 16 17 18 19 20; do
  python bench.py --batch-log2 $lg --steps 5 --warmup 1 --no-cpu-baseline --no-rlc --no-configs > $O/sweep_$lg.json 2> $O/sweep_$lg.err || fail "sweep $lg" $O/sweep_$lg.err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rlc --no-configs > $O/prof_bench.json 2> $O/prof.err || fail rocprof $O/prof.err
echo "kernel trace done"
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_INT64" "FETCH_SIZE" "WRITE_SIZE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
  tag=$(echo $c | cut -d' ' -f1)
  BN254_STREAMS=1 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rlc --no-configs --batch-log2 18 > $O/pmc_$tag.json 2> $O/pmc_$tag.err || fail "pmc $tag" $O/pmc_$tag.err
  echo "pmc $tag done"
done
# the same counters for the one-launch-per-step kernels (BN254_MILLER_RUN_STEPS=0): the comparison behind the run kernel
BN254_MILLER_RUN_STEPS=0 BN254_STREAMS=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_steps_FETCH_SIZE -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rlc --no-configs --batch-log2 18 > $O/pmc_steps_F.json 2> $O/pmc_steps_F.err || fail "pmc steps fetch" $O/pmc_steps_F.err
BN254_MILLER_RUN_STEPS=0 BN254_STREAMS=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_steps_WRITE_SIZE -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rlc --no-configs --batch-log2 18 > $O/pmc_steps_W.json 2> $O/pmc_steps_W.err || fail "pmc steps write" $O/pmc_steps_W.err
cd $R
BN254_MILLER_RUN_STEPS=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rlc --no-configs > $O/bench_steps.json 2> $O/bench_steps.err || fail bench_steps $O/bench_steps.err
find $O -name "*kernel_trace.csv" -size +30M -delete
echo "round 3 GPU script done"
