#!/bin/bash
# round 5: comb tables built on the device -- parity, and what a 1024-input key costs now (prepare on the host, first use on a device, a batch)
set -o pipefail
O=$PWD/gpurun_out/r05c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
python - > $O/wide_key_costs.txt 2>$O/wide_key_costs.err <<'PY' || { tail -5 $O/wide_key_costs.err; exit 1; }
import importlib, json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("snark-bn254-verifier_amd")
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
n, npub = 4096, 1024
t = time.perf_counter(); vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540004, npub, n, invalid_every=16, agree=True, threads=16); gen = time.perf_counter() - t
for mode in ("device", "host"):
    if mode == "host": os.environ["BN254_TABLES_HOST"] = "1"
    t = time.perf_counter(); pvk = pkg.PreparedVk(vk); prep = time.perf_counter() - t
    t = time.perf_counter(); pvk.reserve(n, 0); torch.cuda.synchronize(); first = time.perf_counter() - t
    t = time.perf_counter(); st = pvk.verify_batch(proofs, inputs, n_public=npub) if hasattr(pvk, "verify_batch") else None; dt = time.perf_counter() - t
    ok = (bytes(st) == exp) if st is not None else None
    print(json.dumps({"tables_built_on": mode, "vk_prepare_s": round(prep, 4), "reserve_first_use_on_device_s": round(first, 4), "first_batch_4096_s": round(dt, 4), "statuses_ok": ok}), flush=True)
    pvk.close() if hasattr(pvk, "close") else None
PY
cat $O/wide_key_costs.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rlc > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=None
for l in open("gpurun_out/r05c/bench.json"):
    if l.startswith("{"): d=json.loads(l)
x=d["configs"]["groth16_1024x4096"]; print("1024x4096:", round(x["value"]), round(x["ms_per_step"],3), x["status_check"]); print("headline", round(d["value"]), round(d["ms_per_step"],2))
PY
