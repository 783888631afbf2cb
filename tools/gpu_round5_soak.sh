#!/bin/bash
# round 5, robustness calls (after the G1 formula and Fp12 product changes):  PART=fuzz | plonk | variants
set -o pipefail
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
PART=${PART:-fuzz}
if [ "$PART" = fuzz ]; then
  timeout -k 10 1100 python tools/gpu_fuzz.py --cases 500 --seed 5 > $O/fuzz.txt 2>&1 || { tail -5 $O/fuzz.txt; exit 1; }
  tail -1 $O/fuzz.txt
fi
if [ "$PART" = plonk ]; then
  timeout -k 10 500 python tools/gpu_fuzz_plonk.py --cases 800 > $O/fuzz_plonk.txt 2>&1 || { tail -5 $O/fuzz_plonk.txt; exit 1; }
  tail -1 $O/fuzz_plonk.txt
  timeout -k 10 560 python tools/gpu_soak_mixed.py > $O/soak_mixed.txt 2>&1 || { tail -5 $O/soak_mixed.txt; exit 1; }
  tail -1 $O/soak_mixed.txt
fi
if [ "$PART" = new ]; then
  python -m pytest tests/test_edge_keys.py -m gpu -x -q > $O/pytest_edge_keys.log 2>&1 || { tail -30 $O/pytest_edge_keys.log; exit 1; }
  tail -2 $O/pytest_edge_keys.log
  python bench.py --plonk --steps 3 --warmup 1 > $O/bench_plonk_sharded_1.json 2> $O/bench_plonk_sharded_1.err || { tail -20 $O/bench_plonk_sharded_1.err; exit 1; }
  cut -c1-400 $O/bench_plonk_sharded_1.json
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --plonk --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_plonk_torchrun.json 2> $O/bench_plonk_torchrun.err || { tail -20 $O/bench_plonk_torchrun.err; exit 1; }
  python bench.py --gpus 2 --rehearse-one-gpu --plonk --steps 3 --warmup 1 > $O/bench_plonk_rehearse2.json 2> $O/bench_plonk_rehearse2.err || { tail -20 $O/bench_plonk_rehearse2.err; exit 1; }
  grep -h '^{' $O/bench_plonk_rehearse2.json | cut -c1-300
fi
if [ "$PART" = variants ]; then
  ONLY_LIBS= bash tools/gpu_variants.sh > $O/variants.txt 2>&1; rc=$?
  cat $O/variants.txt
  exit $rc
fi
