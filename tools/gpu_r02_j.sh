#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
# bench.py as ONE rank of a torch.distributed.run job: the RCCL process group, all_gather, all_reduce and barriers run (single rank)
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_torchrun.json 2> gpurun_out/r02_bench_torchrun.err; rc=$?; echo "torchrun bench rc=$rc"; tail -3 gpurun_out/r02_bench_torchrun.err; cut -c1-700 gpurun_out/r02_bench_torchrun.json
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --batch-log2 12 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_4096.json 2> gpurun_out/r02_bench_4096.err; echo "bench 4096 rc=$?"; cut -c1-1500 gpurun_out/r02_bench_4096.json
