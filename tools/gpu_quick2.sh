#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "many_public or more_public or infinity" 2>&1 | tail -3 || exit 1
for v in 1 0; do
BN254_WIDE_COMB=$v timeout -k 10 300 python bench.py --n-public 1024 --batch-log2 12 --steps 5 --warmup 1 --no-cpu-baseline --no-rlc 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('comb=$v',round(d['value']),round(d['ms_per_step'],3),{k:v['total_ms'] for k,v in d['kernels_ms'].items() if 'msm' in k})"
done
