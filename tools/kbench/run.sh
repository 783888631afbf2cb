#!/bin/bash
# run every built variant (on the GPU box: gpurun -- bash tools/kbench/run.sh)
cd "$(dirname "$0")"
for d in obj_*; do [ -x $d/kb ] && timeout -k 10 120 $d/kb; done
