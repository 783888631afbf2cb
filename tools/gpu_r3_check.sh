#!/bin/bash
# round 3: parity tests, smoke (with the torch-free C++ host), the default bench line (headline + `configs`), the library loaded BEFORE torch
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03check
rm -rf $O; mkdir -p $O
fail() { echo "FAILED: $1"; tail -30 $2; exit 1; }
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || fail pytest $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || fail smoke $O/smoke.log
tail -2 $O/smoke.log
# load order: the library first, torch afterwards -- one HIP runtime, devices visible
python - > $O/load_order.log 2>&1 <<'PY' || fail load_order $O/load_order.log
import importlib, sys
sys.path.insert(0, ".")
pkg = importlib.import_module("snark-bn254-verifier_amd")
vk, proofs, inputs, exp = pkg.synth_groth16(7, 2, 64, invalid_every=4)
pvk = pkg.PreparedVk(vk)
assert pvk.verify_batch(proofs, inputs) == exp          # before torch is imported
import torch
assert torch.cuda.is_available() and torch.zeros(4, device="cuda").sum().item() == 0
assert pvk.verify_batch(proofs, inputs) == exp
print("runtimes mapped:", sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)))
print("load order ok")
PY
tail -2 $O/load_order.log
python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err || fail bench $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("headline", round(d["value"]), round(d["ms_per_step"], 2), d["scaling"], "frac", round(d["roofline"]["frac"], 3), d["phases_ms"])
for k, v in d["configs"].items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_step", "ratio_to_resident", "ms", "groth16_verify_ms", "plonk_verify_ms", "status_check")}, "frac", (v.get("roofline") or {}).get("frac"))
print("rlc", d.get("rlc_mode"))
PY
