#!/bin/bash
# experiment: workspace chunk size for a 2^20 batch (BN254_CHUNK_LOG2)
for c in 20 19 18 17; do
  BN254_CHUNK_LOG2=$c python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('chunk 2^$c  %9.0f proofs/s  %8.3f ms/batch' % (d['value'], d['ms_per_step']))"
done
