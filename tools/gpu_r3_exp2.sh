#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03exp2; rm -rf $O; mkdir -p $O
p() { python -c "
import json,sys
d=json.loads([l for l in open('$1').read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('$2', round(d['value']), round(d['ms_per_step'],2), 'overlap', r.get('overlap'), 'frac', r.get('frac'), 'rlc', (d.get('rlc_mode') or {}).get('rlc'))"; }
for q in 4 8; do
  for lg in 17 18; do for s in 1 2; do
    GPU_MAX_HW_QUEUES=$q BN254_STREAMS=$s python bench.py --batch-log2 $lg --steps 6 --warmup 2 --no-cpu-baseline --no-rlc --no-configs > $O/a.json 2> $O/a.err || { tail -3 $O/a.err; exit 1; }
    p $O/a.json "queues=$q 2^$lg streams=$s"
  done; done
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; exit 1; }
  p $O/b.json "queues=$q 2^20 with rlc line"
  GPU_MAX_HW_QUEUES=$q python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 4 --warmup 1 --no-configs --no-cpu-baseline --no-rlc > $O/c.json 2> $O/c.err || { tail -3 $O/c.err; exit 1; }
  p $O/c.json "queues=$q torchrun"
done
