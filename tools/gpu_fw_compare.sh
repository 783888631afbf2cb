#!/bin/bash
# round 5: the window width of the PlonK key-point tables (bn254_fw.h MSM_FW_BITS): 8 (rounds 3-4), 13 (the default), 16 -- parity of the default, then every width's
# throughput, table construction and first call with a new key.  tools/exp/libbn254_fw{8,16}.so: `make BUILD=build_fwN OUT=../../tools/exp/libbn254_fwN.so EXTRA=-DMSM_FW_BITS=N`
set -o pipefail
O=$PWD/gpurun_out/r05w; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
: > $O/fw_compare.txt
for bits in 8 13 16; do
  lib=$PWD/tools/exp/libbn254_fw$bits.so; [ $bits = 13 ] && lib=$PWD/snark-bn254-verifier_amd/libbn254_verify_amd.so
  echo "# MSM_FW_BITS=$bits" >> $O/fw_compare.txt
  BN254_LIB_PATH=$lib python tools/bench_plonk_cold.py >> $O/fw_compare.txt 2>$O/fw.err || { tail -5 $O/fw.err; exit 1; }
  for n in 4096 65536 262144; do
    BN254_LIB_PATH=$lib python tools/bench_plonk.py --batch $n --steps 5 --warmup 1 --cpu-sample 0 --no-in-flight > $O/fw${bits}_$n.json 2> $O/fw.err || { tail -5 $O/fw.err; exit 1; }
    python - $O/fw${bits}_$n.json >> $O/fw_compare.txt <<'PY'
import json, sys
d = None
for l in open(sys.argv[1]):
    if l.startswith("{"): d = json.loads(l)
print(json.dumps({"batch": d["batch"], "proofs_per_s": round(d["value"]), "ms": round(d["ms_per_step"], 3), "rows_digest_ms": round(d["stages_ms"]["k_g1_msm_rows_digest"], 3), "rows_kzg_ms": round(d["stages_ms"]["k_g1_msm_rows_kzg"], 3), "mads_per_proof": round(d["valu_whole_path"]["mads_per_proof"]), "rlc_proofs_per_s": round((d.get("rlc_mode") or {}).get("rlc") or 0), "footprint": d.get("context_footprint")}))
PY
  done
done
cat $O/fw_compare.txt
