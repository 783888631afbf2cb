#!/bin/bash
# round 5, one GPU call: parity tests, smoke, the driver's bench command and the 10-step line, the N > 1 command shape with one rank, the two-rank rehearsal on
# one GPU, rocprofv3 kernel traces (Groth16 headline, PlonK 4096 / 65536) and PMC passes (SQ activity, HBM traffic) for both paths.
# Outputs under gpurun_out/r05/; tools/summarize_r05.py turns them into profiles/r05_*
set -o pipefail
# Two calls (a call is limited to 20 minutes): `PART=a bash tools/gpu_round5.sh` (tests, smoke, bench lines, kernel traces), `PART=b bash tools/gpu_round5.sh` (the counter passes).
R=$PWD; O=$R/gpurun_out/r05
mkdir -p $O
PART=${PART:-a}
fail() { echo "FAILED: $1"; tail -20 $2; exit 1; }
if [ "$PART" = a ]; then
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || fail pytest $O/pytest_gpu.log
tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || fail smoke $O/smoke.log
tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err || fail bench_default $O/bench_default.err
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || fail bench $O/bench.err
cut -c1-300 $O/bench.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-rlc > $O/bench_torchrun.json 2> $O/bench_torchrun.err || fail torchrun $O/bench_torchrun.err
python bench.py --gpus 2 --rehearse-one-gpu --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-rlc > $O/bench_rehearse2.json 2> $O/bench_rehearse2.err || fail rehearse $O/bench_rehearse2.err
grep -h '^{' $O/bench_rehearse2.json | cut -c1-200
python tools/bench_plonk.py --batch 4096 --steps 20 --warmup 3 --cpu-sample 512 > $O/plonk4096.json 2> $O/plonk4096.err || fail plonk4096 $O/plonk4096.err
python tools/bench_plonk.py --batch 65536 --steps 5 --warmup 1 --cpu-sample 0 > $O/plonk65536.json 2> $O/plonk65536.err || fail plonk65536 $O/plonk65536.err
python tools/bench_plonk.py --batch 262144 --steps 3 --warmup 1 --cpu-sample 0 > $O/plonk262144.json 2> $O/plonk262144.err || fail plonk262144 $O/plonk262144.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rlc --no-configs > $O/prof_bench.json 2> $O/prof.err || fail rocprof $O/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_plonk -o run -- python3 $R/tools/bench_plonk.py --batch 4096 --steps 10 --warmup 2 --cpu-sample 0 --no-in-flight > $O/prof_plonk.json 2> $O/prof_plonk.err || fail "rocprof plonk" $O/prof_plonk.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_plonk64k -o run -- python3 $R/tools/bench_plonk.py --batch 65536 --steps 3 --warmup 1 --cpu-sample 0 > $O/prof_plonk64k.json 2> $O/prof_plonk64k.err || fail "rocprof plonk 64k" $O/prof_plonk64k.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_plonk256k -o run -- python3 $R/tools/bench_plonk.py --batch 262144 --steps 2 --warmup 1 --cpu-sample 0 --no-in-flight > $O/prof_plonk256k.json 2> $O/prof_plonk256k.err || fail "rocprof plonk 256k" $O/prof_plonk256k.err
echo "kernel traces done"
cd $R
find $O -name "*kernel_trace.csv" -size +30M -delete
fi
if [ "$PART" = b ]; then
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_INT64" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | cut -d' ' -f1)
  BN254_STREAMS=1 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rlc --no-configs --batch-log2 18 > $O/pmc_$tag.json 2> $O/pmc_$tag.err || fail "pmc $tag" $O/pmc_$tag.err
  rocprofv3 --pmc $c --output-format csv -d $O/pmcp_$tag -o run -- python3 $R/tools/bench_plonk.py --batch 4096 --steps 2 --warmup 1 --cpu-sample 0 --no-in-flight > $O/pmcp_$tag.json 2> $O/pmcp_$tag.err || fail "pmc plonk $tag" $O/pmcp_$tag.err
  echo "pmc $tag done"
done
cd $R
fi
echo "round 5 GPU script part $PART done"
