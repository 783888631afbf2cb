#!/bin/bash
# Build kernel micro-benchmark variants in parallel: build.sh NAME "flags" [NAME "flags" ...]
#   e.g. build.sh LF_CUR "-DV_LF_CUR" LF_K "-DV_LF_K"      (variants: see the #if chain in kbench.hip)
# Each variant is compiled in its own directory obj_NAME (binary obj_NAME/kb, ISA and resource usage beside it).
cd "$(dirname "$0")"
while [ $# -gt 0 ]; do
  v=$1; fl=$2; shift 2
  ( mkdir -p obj_$v && cd obj_$v && timeout 1200 /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $fl -DVNAME=\"$v\" ../kbench.hip -o kb -save-temps=obj -Rpass-analysis=kernel-resource-usage > log 2>&1; echo "$v exit $?" >> log ) &
done
wait
tail -qn1 obj_*/log
