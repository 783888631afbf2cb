#!/bin/bash
# quick GPU check: parity suite, smoke, RLC against the exact path at three sizes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/quick_pytest.txt 2>&1; rc=$?
tail -3 gpurun_out/quick_pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
for b in 12 17 18; do timeout -k 10 300 python tools/bench_rlc.py --batch-log2 $b --steps 5 --invalid-every 0 2>/dev/null | tail -1 | cut -c1-330; done
