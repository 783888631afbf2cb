#!/bin/bash
set -o pipefail
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 300 python tools/bench_plonk.py > $O/plonk.json 2> $O/plonk.err || { tail -5 $O/plonk.err; exit 1; }
cut -c1-300 $O/plonk.json
