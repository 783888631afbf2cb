#!/bin/bash
# one GPU call: parity tests, smoke, bench, rocprofv3 kernel trace of the bench command, PMC passes (outputs under gpurun_out/)
set -o pipefail
mkdir -p gpurun_out
R=$PWD
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cut -c1-900 gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof $R/gpurun_out/pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err || { tail -20 $R/gpurun_out/prof.err; exit 1; }
echo "kernel trace done"
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | cut -d' ' -f1)
  BN254_STREAMS=1 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --batch-log2 18 > $R/gpurun_out/pmc_$tag.json 2> $R/gpurun_out/pmc_$tag.err || { tail -20 $R/gpurun_out/pmc_$tag.err; exit 1; }
  echo "pmc $tag done"
done
cd $R
find gpurun_out/prof gpurun_out/pmc_* -name "*.csv" | head -20
# secondary configurations (BASELINE configs[3] and [4])
python tools/bench_plonk.py > gpurun_out/bench_plonk.json 2> gpurun_out/bench_plonk.err || { tail -20 gpurun_out/bench_plonk.err; exit 1; }
python bench.py --n-public 1024 --batch-log2 12 --steps 5 --warmup 1 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err || { tail -20 gpurun_out/bench_cfg5.err; exit 1; }
echo "secondary benches done"
