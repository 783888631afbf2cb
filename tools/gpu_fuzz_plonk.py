#!/usr/bin/env python3
"""Randomised differential run of the PlonK batch entry on the GPU box: a pool of distinct cases (the reference's four fixtures and mutations of their proofs
and public inputs, every status the path can return) gets its verdicts from the CPU oracle once; then many batches of random sizes -- 1 .. 21 000 proofs, with
the sizes around the limits of the row plans, the row sums and the batch plan (and round 3's launch-form limits) over-represented -- are drawn from the pool
in random order, on ONE prepared key (its contexts keep the capacities earlier batches gave them), and every status byte is compared.
  python tools/gpu_fuzz_plonk.py [--cases 60] [--seed 1]"""
import argparse, importlib, json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mutate(rng, proof, pis):
    p, q = bytearray(proof), bytearray(pis)
    kind = rng.randrange(9)
    if kind == 0: q[rng.randrange(64)] ^= 1 << rng.randrange(8)                 # public input
    elif kind == 1: p[516 + rng.randrange(32 * 7)] ^= 1 << rng.randrange(8)     # a claimed value
    elif kind == 2: p[rng.randrange(512)] ^= 1 << rng.randrange(8)              # a commitment coordinate
    elif kind == 3: p[0:32] = bytes([0xff]) * 32                                # coordinate >= p
    elif kind == 4: p[740 + rng.randrange(128)] ^= 1 << rng.randrange(8)        # an opening proof
    elif kind == 5: p[512:516] = (0).to_bytes(4, "big")                         # claimed-value count
    elif kind == 6: q[0:32] = bytes([0xff]) * 32                                # public input >= r
    elif kind == 7: p[rng.randrange(len(p))] ^= 0x80
    return bytes(p), bytes(q)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--pool", type=int, default=160)
    args = ap.parse_args()
    import torch  # noqa: F401  (HIP runtime load order)
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    from oracle import oracle as O
    O.build(); O.lib()
    rng = random.Random(args.seed)
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
    vk = open(os.path.join(ROOT, "tests", "golden", "plonk_vk.bin"), "rb").read()
    base = [(bytes.fromhex(f["raw_proof"]), b"".join(int(x).to_bytes(32, "big") for x in f["public_inputs"])) for f in fx.values() if f["variant"] == "plonk"]
    pool = list(base)
    while len(pool) < args.pool:
        pool.append(mutate(rng, *base[rng.randrange(len(base))]))
    t0 = time.time()
    verdict = [O.plonk_verify(p, vk, [int.from_bytes(q[:32], "big"), int.from_bytes(q[32:], "big")]) for p, q in pool]
    oracle_s = time.time() - t0
    pvk = pkg.PreparedPlonkVk(vk)
    # the limits of round 4's forms: split rows while 2 x variable terms x lanes <= 65536 (KZG launch: 4096 proofs, digest launch: 6528), four lanes per row sum up to
    # 8192 (two sums) / 16384 proofs (one), passes of <= 5040 proofs up to 9000 per call, one pass up to 20 000, two up to 40 000, passes of 65 536 above, joint MSM rows from 49 089 proofs per pass; and round 3's
    sizes = [1, 2, 63, 64, 65, 255, 256, 257, 1000, 2166, 2167, 2519, 2520, 2521, 2560, 4095, 4096, 4097, 4332, 4333, 4864, 4865, 5039, 5040, 5041, 5042, 5120, 5121, 6000,
             6528, 6529, 8192, 8193, 9000, 9001, 10079, 10080, 10081, 12345, 15120, 15121, 16384, 16385, 20000, 20001, 20160, 20161, 21000, 40000, 40001, 49088, 49089, 49152, 65536, 65537]
    checked = 0
    classes = {}
    for case in range(args.cases):
        n = rng.choice(sizes) if rng.random() < 0.8 else rng.randrange(1, 21000)
        valid_share = rng.choice([1.0, 0.9, 0.5, 0.0])
        idx = [rng.randrange(len(base)) if rng.random() < valid_share else rng.randrange(len(pool)) for _ in range(n)]
        flags = pkg.FLAG_RLC if rng.random() < 0.5 else 0        # half of the cases with the pairing checks batched across proofs (honoured from 8192 proofs per pass)
        st = pvk.verify_batch(b"".join(pool[i][0] for i in idx), b"".join(pool[i][1] for i in idx), flags=flags)
        exp = bytes(verdict[i] for i in idx)
        assert st == exp, (case, n, [(k, st[k], exp[k]) for k in range(n) if st[k] != exp[k]][:5])
        checked += n
        for b in set(exp): classes[str(b)] = classes.get(str(b), 0) + exp.count(bytes([b]))
        print(json.dumps({"case": case, "n": n, "valid_share": valid_share, "rlc": bool(flags), "ok": True}), flush=True)
    pvk.close()
    print(json.dumps({"cases": args.cases, "pool": len(pool), "pool_verdicts": {str(v): verdict.count(v) for v in sorted(set(verdict))}, "oracle_seconds": round(oracle_s, 1),
                      "status_bytes_checked": checked, "status_classes_seen": classes, "seconds": round(time.time() - t0, 1), "all_ok": True}))


if __name__ == "__main__":
    main()
