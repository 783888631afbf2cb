#!/usr/bin/env python3
"""Randomised differential run on the GPU box: many (seed, batch size, public-input count, invalid fraction, key mode, flags) combinations through the
C ABI; every status byte is compared with the generator's prediction, a random sample of every batch with the CPU oracle, and the three
execution paths (cooperative small-batch kernels, one-proof-per-lane kernels, RLC mode) with each other where more than one applies.
  python tools/gpu_fuzz.py [--cases 40] [--seed 1]"""
import os
os.environ.setdefault("BN254_RLC_MIN_BATCH", "64")      # run the RLC kernels on the small random batches too (default: from 200 000 proofs)
import argparse, importlib, json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch  # noqa: F401  (HIP runtime load order)
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    from oracle import oracle as O
    O.build(); O.lib(); O.set_threads(16)
    rng = random.Random(args.seed)
    t0 = time.time()
    checked = oracle_checked = 0
    for case in range(args.cases):
        n_public = rng.choice([0, 1, 2, 2, 2, 3, 5, 8, 12, 20])
        n = rng.choice([1, 7, 63, 64, 65, 100, 333, 1000, 2049, 4096, 5000, 10240, 10241, 20000, 30720, 30721, 40961, 65536, 65537, 70001, 131073, 150000, 300000])   # the last sizes: 11 / 22 / 44 steps per k_miller_run launch
        inv = rng.choice([0, 2, 3, 5, 16, 50])
        mode = rng.choice([pkg.VK_REFERENCE, pkg.VK_GNARK])
        seed = 0xF0220000 + rng.randrange(1 << 16)
        vk, proofs, inputs, exp = pkg.synth_groth16(seed, n_public, n, invalid_every=inv, agree=True, threads=16)
        pvk = pkg.PreparedVk(vk, mode)
        res = {}
        for name, flags in (("exact", 0), ("rlc", pkg.FLAG_RLC), ("strict", pkg.FLAG_STRICT_SCALARS)):
            res[name] = pvk.verify_batch(proofs, inputs, n, 256, n_public, 0, flags)
            assert res[name] == exp, (case, name, n, n_public, inv, mode, seed)
        # the oracle on a random sample (it is slow: ~3 k proofs/s on 16 threads)
        m = min(n, 24)
        idx = sorted(rng.sample(range(n), m))
        sp = b"".join(proofs[256 * i:256 * i + 256] for i in idx); si = b"".join(inputs[32 * n_public * i:32 * n_public * (i + 1)] for i in idx)
        ref = O.groth16_verify_many(sp, 256, vk, si, n_public, m, O.MODE_REFERENCE if mode == pkg.VK_REFERENCE else O.MODE_GNARK)
        assert ref == bytes(exp[i] for i in idx), (case, "oracle", n, n_public, inv, mode, seed)
        checked += 3 * n; oracle_checked += m
        pvk.close()
        print(json.dumps({"case": case, "n": n, "n_public": n_public, "invalid_every": inv, "mode": mode, "seed": seed, "ok": True,
                          "classes": {str(k): exp.count(bytes([k])) for k in sorted(set(exp))}}), flush=True)
    print(json.dumps({"cases": args.cases, "status_bytes_checked": checked, "oracle_checked": oracle_checked, "seconds": round(time.time() - t0, 1), "all_ok": True}))


if __name__ == "__main__":
    main()
