#!/usr/bin/env python3
"""Known-answer fixtures for Groth16 `verify()` at VERDICT level, from an implementation that is independent of both the oracle and the product.

The reference's own Groth16 test needs a verifying key that lives outside its repository (examples/script/src/main.rs:178-180), so no
reference-held file pins an accept / reject.  This script is the compensation SURVEY.md section 8(c) names: synthetic gnark-format keys and proofs
(the product's HOST-side generator, bn254_synth_groth16 -- no GPU involved) are judged by tests/pyref_groth16.py (pure Python: integers, affine
arithmetic, polynomial-basis Fp12, plain final exponentiation; it restates lib.rs:44-49, groth16/converter.rs:14-89, converter.rs:23-153 and
groth16/verify.rs:53-78) in BOTH readings of a compressed G2 point, and the verdicts are written to groth16_verdicts.json.

  python tests/golden/make_groth16_verdicts.py        # about six minutes (3 s per pairing, 4 pairings per verdict)

tests/test_groth16_verdict_pin.py then holds the oracle (CPU) and the product (GPU) to these verdicts and re-derives a few of them live."""
import importlib, json, os, sys, time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref_groth16 as PY  # noqa: E402


def main():
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    keys, cases = {}, []
    # key "agree": sampled where both readings of the key coincide on every proof; 12 proofs, every 2nd one invalid,
    # cycling through the five classes (input + 1, C + G, A.y + 1, B outside G2, A.x >= p)
    # key "disagree": sampled without that constraint: valid proofs that gnark accepts and -- for most such keys -- the reference's literal
    # reading rejects
    for name, seed, agree, n, inv in (("agree", 0xB25400A1, True, 12, 2), ("disagree", 0xB25400A2, False, 4, 0), ("one_input", 0xB25400A3, True, 2, 0)):
        n_public = 1 if name == "one_input" else 2
        vk, proofs, inputs, exp = pkg.synth_groth16(seed, n_public, n, invalid_every=inv, agree=agree, threads=2)
        keys[name] = {"vk": vk.hex(), "n_public": n_public}
        for i in range(n):
            cases.append({"key": name, "proof": proofs[256 * i:256 * i + 256].hex(), "inputs": inputs[32 * n_public * i:32 * n_public * (i + 1)].hex(), "generator_expects": exp[i]})
    # wrong number of public inputs (PrepareInputsFailed), a short proof buffer, public inputs x and x + r
    c0 = cases[0]
    cases.append({"key": "agree", "proof": c0["proof"], "inputs": c0["inputs"][:64], "generator_expects": None, "note": "one public input for a two-input key"})
    cases.append({"key": "agree", "proof": c0["proof"][:200], "inputs": c0["inputs"], "generator_expects": None, "note": "short proof buffer"})
    x0 = int(c0["inputs"][:64], 16)
    if x0 + PY.R < 1 << 256:
        cases.append({"key": "agree", "proof": c0["proof"], "inputs": (x0 + PY.R).to_bytes(32, "big").hex() + c0["inputs"][64:], "generator_expects": None, "note": "first input + r: Fr is not range-checked"})
    t0 = time.time()
    for k, c in enumerate(cases):
        key = keys[c["key"]]
        vk, proof = bytes.fromhex(key["vk"]), bytes.fromhex(c["proof"])
        ib = bytes.fromhex(c["inputs"])
        ins = [int.from_bytes(ib[32 * j:32 * j + 32], "big") for j in range(len(ib) // 32)]
        c["verdict_reference"] = PY.verify(proof, vk, ins, PY.MODE_REFERENCE)
        c["verdict_gnark"] = PY.verify(proof, vk, ins, PY.MODE_GNARK)
        print("case %2d key %-9s reference %d gnark %d generator %s  (%.0f s)" % (k, c["key"], c["verdict_reference"], c["verdict_gnark"], c["generator_expects"], time.time() - t0), flush=True)
    json.dump({"_note": "written by tests/golden/make_groth16_verdicts.py; verdicts are status bytes of include/bn254_verify.h from tests/pyref_groth16.py",
               "keys": keys, "cases": cases}, open(os.path.join(HERE, "groth16_verdicts.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
