#!/usr/bin/env python3
"""Cost of a key the single-proof entries have not seen: `Groth16Verifier::verify(proof, vk bytes, inputs)` with a DIFFERENT key on every call (twelve keys through the
four-slot cache: every call is a miss), against the same call with a cached key.  One JSON line.  BN254_TABLES_HOST=1: the host construction of the key's tables (rounds 1-4)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # loads the HIP runtime torch ships before the library does
pkg = importlib.import_module("snark-bn254-verifier_amd")
out = {"tables_built_on": "host" if os.environ.get("BN254_TABLES_HOST", "0") != "0" else "device"}
for n_public in (2, 16):
    keys = [pkg.synth_groth16(0xB2548000 + 100 * n_public + k, n_public, 1, invalid_every=0, agree=True, threads=2) for k in range(12)]
    vk0, p0, i0, _ = keys[0]
    ins0 = [int.from_bytes(i0[32 * j:32 * j + 32], "big") for j in range(n_public)]
    for _ in range(3):
        assert pkg.Groth16Verifier.verify(p0[:256], vk0, ins0) == pkg.ACCEPT           # warm: device state, streams, the first key
    t = time.perf_counter()
    for _ in range(20):
        pkg.Groth16Verifier.verify(p0[:256], vk0, ins0)
    hit = (time.perf_counter() - t) / 20 * 1e3
    ms = []
    for rnd in range(2):                                                                     # every call a key that is not (any more) in the cache
        for vk, p, i, _ in keys[1:]:
            ins = [int.from_bytes(i[32 * j:32 * j + 32], "big") for j in range(n_public)]
            t = time.perf_counter()
            st = pkg.Groth16Verifier.verify(p[:256], vk, ins)
            ms.append((time.perf_counter() - t) * 1e3)
            assert st == pkg.ACCEPT
    ms.sort()
    out["inputs_%d" % n_public] = {"cached_key_ms": round(hit, 3), "new_key_median_ms": round(ms[len(ms) // 2], 3), "new_key_min_ms": round(ms[0], 3), "new_key_max_ms": round(ms[-1], 3), "calls": len(ms)}
print(json.dumps(out))
