#!/bin/bash
# builds the cooperative-layout microbenchmark (tools/coop/coop_bench.hip) for gfx950
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../snark-bn254-verifier_amd/csrc coop_bench.hip -o coop_bench
