#!/bin/bash
# round 5, first GPU call: the new parity tests (edge keys, PlonK entries / large passes), then the whole GPU suite, then the PlonK pass-size sweep.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r05a
rm -rf $O; mkdir -p $O
fail() { echo "FAILED: $1"; tail -30 $2; exit 1; }
timeout -k 10 900 python -m pytest tests/test_edge_keys.py tests/test_gpu_round5.py -m gpu -x -q > $O/pytest_new.log 2>&1 || fail "new tests" $O/pytest_new.log
tail -2 $O/pytest_new.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || fail pytest $O/pytest_gpu.log
tail -2 $O/pytest_gpu.log
timeout -k 10 600 python tools/bench_plonk_piece.py --sizes 65536,131072,262144,524288 > $O/plonk_piece_sweep.txt 2> $O/plonk_piece_sweep.err || fail sweep $O/plonk_piece_sweep.err
cat $O/plonk_piece_sweep.txt | cut -c1-420
timeout -k 10 300 python tools/bench_plonk_piece.py --sizes 131072,262144 --flags 2 --steps 2 > $O/plonk_piece_sweep_rlc.txt 2> $O/plonk_piece_sweep_rlc.err || fail sweep_rlc $O/plonk_piece_sweep_rlc.err
cat $O/plonk_piece_sweep_rlc.txt | cut -c1-200
for v in MUL_L MUL_T3 MUL_T3B MUL_T3D MUL_T3E; do timeout -k 10 120 tools/kbench/obj_$v/kb >> $O/kbench_f12_mul.txt 2>&1 || fail "kbench $v" $O/kbench_f12_mul.txt; done
for v in MUL_L MUL_T3 MUL_T3B MUL_T3D MUL_T3E; do timeout -k 10 120 tools/kbench/obj_$v/kb >> $O/kbench_f12_mul.txt 2>&1 || fail "kbench $v" $O/kbench_f12_mul.txt; done
cat $O/kbench_f12_mul.txt
echo "round 5a done"
