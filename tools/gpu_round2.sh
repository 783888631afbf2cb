#!/bin/bash
# round 2, one GPU call: parity tests, smoke, bench (+ host buffers), rocprofv3 kernel trace of the bench command, PMC passes, the secondary
# configurations (RLC mode, small batches, PlonK, 1024 public inputs).  Outputs under gpurun_out/r02/; tools/summarize_profiles.py r02_final gpurun_out/r02
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
fail() { echo "FAILED: $1"; tail -20 $2; exit 1; }
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || fail pytest $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || fail smoke $O/smoke.log
tail -1 $O/smoke.log
python bench.py --host-buffers > $O/bench.json 2> $O/bench.err || fail bench $O/bench.err
cut -c1-400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rlc > $O/prof_bench.json 2> $O/prof.err || fail rocprof $O/prof.err
echo "kernel trace done"
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | cut -d' ' -f1)
  BN254_STREAMS=1 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rlc --batch-log2 18 > $O/pmc_$tag.json 2> $O/pmc_$tag.err || fail "pmc $tag" $O/pmc_$tag.err
  echo "pmc $tag done"
done
cd $R
# the kernel traces are large: keep the per-kernel statistics, drop traces above 30 MB
find $O -name "*kernel_trace.csv" -size +30M -delete
python tools/bench_rlc.py --batch-log2 20 --steps 3 --invalid-every 0,256,16 > $O/rlc.txt 2> $O/rlc.err || fail rlc $O/rlc.err
cat $O/rlc.txt
python tools/bench_small.py > $O/small_coop12.txt 2> $O/small_coop12.err || fail small $O/small_coop12.err
BN254_COOP=0 python tools/bench_small.py > $O/small_lane.txt 2> $O/small_lane.err || fail small_lane $O/small_lane.err
grep 4096 $O/small_coop12.txt $O/small_coop.txt $O/small_lane.txt
# batch 4096 (BASELINE configs[1]) under the profiler: kernel trace, then SQ counters in their own passes
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small -o run -- python3 $R/bench.py --batch-log2 12 --steps 20 --warmup 2 --no-cpu-baseline --no-rlc > $O/small.json 2> $O/small.err || fail "rocprof small" $O/small.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_small -o run -- python3 $R/bench.py --batch-log2 12 --steps 3 --warmup 1 --no-cpu-baseline --no-rlc > $O/pmc_small.json 2> $O/pmc_small.err || fail "pmc small" $O/pmc_small.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_small2 -o run -- python3 $R/bench.py --batch-log2 12 --steps 3 --warmup 1 --no-cpu-baseline --no-rlc > $O/pmc_small2.json 2> $O/pmc_small2.err || fail "pmc small2" $O/pmc_small2.err
rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_int64_small -o run -- python3 $R/bench.py --batch-log2 12 --steps 3 --warmup 1 --no-cpu-baseline --no-rlc > $O/pmc_int64_small.json 2> $O/pmc_int64_small.err || fail "pmc int64 small" $O/pmc_int64_small.err
rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_int64_plonk -o run -- python3 $R/tools/bench_plonk.py --steps 2 > $O/pmc_int64_plonk.json 2> $O/pmc_int64_plonk.err || fail "pmc int64 plonk" $O/pmc_int64_plonk.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rlc -o run -- python3 $R/tools/bench_rlc.py --batch-log2 20 --steps 2 --invalid-every 0 > $O/rlc_prof.txt 2> $O/rlc_prof.err || fail "rocprof rlc" $O/rlc_prof.err
find $O -name "*kernel_trace.csv" -size +30M -delete
cd $R
python tools/bench_plonk.py > $O/plonk.json 2> $O/plonk.err || fail plonk $O/plonk.err
cut -c1-200 $O/plonk.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_plonk -o run -- python3 $R/tools/bench_plonk.py --steps 3 > $O/prof_plonk.json 2> $O/prof_plonk.err || fail "rocprof plonk" $O/prof_plonk.err
cd $R
python bench.py --n-public 1024 --batch-log2 12 --steps 5 --warmup 1 --cpu-sample 128 > $O/cfg5.json 2> $O/cfg5.err || fail cfg5 $O/cfg5.err
cut -c1-200 $O/cfg5.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -o run -- python3 $R/bench.py --n-public 1024 --batch-log2 12 --steps 5 --warmup 1 --no-cpu-baseline --no-rlc > $O/prof_cfg5.json 2> $O/prof_cfg5.err || fail "rocprof cfg5" $O/prof_cfg5.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_INT64 --output-format csv -d $O/pmc_cfg5 -o run -- python3 $R/bench.py --n-public 1024 --batch-log2 12 --steps 1 --warmup 0 --no-cpu-baseline --no-rlc > $O/pmc_cfg5.json 2> $O/pmc_cfg5.err || fail "pmc cfg5" $O/pmc_cfg5.err
cd $R
echo "round 2 GPU script done"
