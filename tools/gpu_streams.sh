#!/bin/bash
# experiment: concurrent sub-batches on 1..4 streams (BN254_STREAMS)
for s in 1 2 3 4; do
  echo "STREAMS=$s"
  BN254_STREAMS=$s python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/streams_$s.json 2> gpurun_out/streams_$s.err || exit 1
  python -c "import json;d=json.load(open('gpurun_out/streams_$s.json'));print(d['value'], d['ms_per_step'])"
done
