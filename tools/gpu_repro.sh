#!/bin/bash
# The wrong-challenge build of round 4 on the GPU box: (1) the library built with -DBN254_FR_MUL_INLINE=1 -DBN254_FR_NO_BARRIER (tools/repro/build/libbn254_bad.so) --
# the device self-test must refuse it on a key's first use, and with the self-test off the statuses differ from the oracle; (2) tools/repro/plonk_challenge_repro built
# with and without the two flags.  Output: gpurun_out/repro/
set -o pipefail
O=$PWD/gpurun_out/repro; rm -rf $O; mkdir -p $O
python - > $O/bad_library.txt 2>&1 <<'PY'
import os, sys, json, importlib
os.environ["BN254_LIB_PATH"] = os.path.join(os.getcwd(), "tools/repro/build/libbn254_bad.so")
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("snark-bn254-verifier_amd")
from oracle import oracle as O
fx = json.load(open("tests/golden/fixtures.json")); vk = open("tests/golden/plonk_vk.bin", "rb").read()
cases = [(bytes.fromhex(f["raw_proof"]), [int(x) for x in f["public_inputs"]]) for f in fx.values() if f["variant"] == "plonk"]
pb = b"".join(c[0] for c in cases); ib = b"".join(b"".join(int(x).to_bytes(32, "big") for x in c[1]) for c in cases)
pvk = pkg.PreparedPlonkVk(vk)
try:
    st = pvk.verify_batch(pb, ib)
    print("self-test ON : the bad library verified the fixtures:", list(st), "(the defect did not show in this build)")
except pkg.Bn254Error as e:
    print("self-test ON : refused ->", str(e)[:400])
PY
cat $O/bad_library.txt
BN254_PLONK_SELFTEST=0 python - > $O/bad_library_noselftest.txt 2>&1 <<'PY'
import os, sys, json, importlib
os.environ["BN254_LIB_PATH"] = os.path.join(os.getcwd(), "tools/repro/build/libbn254_bad.so")
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("snark-bn254-verifier_amd")
fx = json.load(open("tests/golden/fixtures.json")); vk = open("tests/golden/plonk_vk.bin", "rb").read()
cases = [(bytes.fromhex(f["raw_proof"]), [int(x) for x in f["public_inputs"]]) for f in fx.values() if f["variant"] == "plonk"]
pb = b"".join(c[0] for c in cases); ib = b"".join(b"".join(int(x).to_bytes(32, "big") for x in c[1]) for c in cases)
pvk = pkg.PreparedPlonkVk(vk)
print("self-test OFF: statuses of the reference's four valid fixtures:", list(pvk.verify_batch(pb, ib)), "(1 = ACCEPT is the right answer)")
PY
cat $O/bad_library_noselftest.txt
for v in good bad; do echo "== plonk_challenge_repro, $v flags" >> $O/repro_kernels.txt; timeout -k 10 120 tools/repro/build/repro_$v tests/golden/plonk_vk.bin >> $O/repro_kernels.txt 2>&1; echo "exit $?" >> $O/repro_kernels.txt; done
cat $O/repro_kernels.txt
echo "repro done"
