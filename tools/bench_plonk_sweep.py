#!/usr/bin/env python3
"""PlonK batch plans against each other on one MI355X: for every batch size the chain form (passes of <= 5040 proofs, up to 8 side by side) and the large-pass
form (passes of up to 65536 proofs) with 1 / 2 / 8 sub-batches in flight -- bn254_set_plonk_params switches between them in one process.  Prints one JSON line
per (size, plan); status bytes of every plan must equal the chain form's."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4096,5040,6144,8192,10240,12288,16384,20480,24576,32768,40960,49152,65536,131072,262144")
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    import bench
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    sizes = [int(x) for x in args.sizes.split(",")]
    vk, pb, ib, _, _ = bench.plonk_workload(max(sizes))
    pvk = pkg.PreparedPlonkVk(vk)
    for n in sizes:
        p, q = pb[:904 * n], ib[:64 * n]
        ref = None
        plans = [("chains5040_w8", dict(piece=5040, workers=8, big_from=1 << 30))]
        # chains of smaller passes: up to 4096 proofs the KZG launch keeps its split rows (2 x 8 variable terms x 4096 lanes = one wavefront per SIMD)
        plans += [("chains%d_w8" % pc, dict(piece=pc, workers=8, big_from=1 << 30)) for pc in (4096, 2560) if n > pc]
        for w in (1, 2, 8):
            for bp in (16384, 32768, 65536):
                if bp >= n and w > 1:
                    continue
                if bp > 2 * n:
                    continue
                plans.append(("big%d_w%d" % (bp, w), dict(piece=5040, workers=w, big_from=1, big_piece=bp)))
        for name, kw in plans:
            pkg.set_plonk_params(**kw)
            st = pvk.verify_batch(p, q, n)
            t = time.perf_counter()
            for _ in range(args.steps):
                st = pvk.verify_batch(p, q, n)
            dt = (time.perf_counter() - t) / args.steps
            if ref is None:
                ref = st
                assert st.count(bytes([pkg.ACCEPT])) == n - n // 8
            assert st == ref, "statuses differ between plans"
            print(json.dumps({"n": n, "plan": name, "ms": round(dt * 1e3, 3), "proofs_per_s": round(n / dt)}), flush=True)


if __name__ == "__main__":
    main()
