#!/bin/bash
# variants of the wrong-challenge build (both workarounds off) under other optimisation settings: which of them still verify the reference's valid fixtures wrongly?
O=$PWD/gpurun_out/repro; mkdir -p $O
for v in bad_O3 bad_nsa bad_O2 bad_O1; do
BN254_PLONK_SELFTEST=0 BN254_VARIANT=$v python - >> $O/variants.txt 2>&1 <<'PY'
import os, sys, json, importlib
v = os.environ["BN254_VARIANT"]
os.environ["BN254_LIB_PATH"] = os.path.join(os.getcwd(), "tools/repro/build/libbn254_%s.so" % v)
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("snark-bn254-verifier_amd")
fx = json.load(open("tests/golden/fixtures.json")); vk = open("tests/golden/plonk_vk.bin", "rb").read()
cases = [(bytes.fromhex(f["raw_proof"]), [int(x) for x in f["public_inputs"]]) for f in fx.values() if f["variant"] == "plonk"]
pb = b"".join(c[0] for c in cases); ib = b"".join(b"".join(int(x).to_bytes(32, "big") for x in c[1]) for c in cases)
pvk = pkg.PreparedPlonkVk(vk)
print(v, "statuses of the four valid fixtures:", list(pvk.verify_batch(pb, ib)), " batch of 300:", sorted(set(pvk.verify_batch(pb * 75, ib * 75))))
PY
done
grep -v amdgpu.ids $O/variants.txt
