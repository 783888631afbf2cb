#!/bin/bash
# throughput against batch size (inputs resident, 2 sub-batch streams above 32768 proofs)
for b in 10 12 14 16 18 20; do
  python bench.py --batch-log2 $b --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('2^%d  %9.0f proofs/s  %8.3f ms/batch' % ($b, d['value'], d['ms_per_step']))"
done
