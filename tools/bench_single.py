#!/usr/bin/env python3
"""Latency of the reference-shaped entry points: Groth16Verifier.verify (one proof, vk bytes parsed per call as lib.rs:44-49 does), the host-buffer
batch entry at small sizes, PlonkVerifier.verify.  One JSON line."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # loads the HIP runtime torch ships before the library does
pkg = importlib.import_module("snark-bn254-verifier_amd")
vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540042, 2, 4096, invalid_every=0, agree=True, threads=8)
ins = [int.from_bytes(inputs[32 * i:32 * i + 32], "big") for i in range(2)]
out = {}
for name, reps in (("groth16_verify_single_ms", 20),):
    pkg.Groth16Verifier.verify(proofs[:256], vk, ins)
    t = time.perf_counter()
    for _ in range(reps):
        st = pkg.Groth16Verifier.verify(proofs[:256], vk, ins)
    out[name] = (time.perf_counter() - t) / reps * 1e3
    assert st == pkg.ACCEPT
pvk = pkg.PreparedVk(vk)
for n in (1, 64, 1024, 4096):
    pvk.verify_batch(proofs[:256 * n], inputs[:64 * n], n)
    t = time.perf_counter()
    for _ in range(10):
        st = pvk.verify_batch(proofs[:256 * n], inputs[:64 * n], n)
    out["groth16_host_buffers_batch_%d_ms" % n] = (time.perf_counter() - t) / 10 * 1e3
    assert st == exp[:n]
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
pvkb = open(os.path.join(ROOT, "tests", "golden", "plonk_vk.bin"), "rb").read()
f = [f for f in fx.values() if f["variant"] == "plonk"][0]
pp, pi = bytes.fromhex(f["raw_proof"]), [int(x) for x in f["public_inputs"]]
pkg.PlonkVerifier.verify(pp, pvkb, pi)
t = time.perf_counter()
for _ in range(20):
    st = pkg.PlonkVerifier.verify(pp, pvkb, pi)
out["plonk_verify_single_ms"] = (time.perf_counter() - t) / 20 * 1e3
assert st == pkg.ACCEPT
print(json.dumps(out))
