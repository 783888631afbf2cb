#!/usr/bin/env python3
"""PlonK batches above 65 536 proofs: proofs per pass x passes in flight, on the device-resident entry (bn254_plonk_verify_batch_device) and, for the chosen plan, on
the host-buffer entry.  One JSON line per (batch, plan) with the stage durations of the first pass (HIP events on its stream) -- what PLONK_BIG_PIECE_DEFAULT in
csrc/bn254_capi.hip is read off (profiles/r05_plonk_piece_sweep.txt).  Status bytes of every plan must be equal."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="65536,98304,131072,196608,262144,524288")
    ap.add_argument("--pieces", default="32768,65536,131072,262144")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--flags", type=int, default=0)
    args = ap.parse_args()
    import torch
    import bench
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    sizes = [int(x) for x in args.sizes.split(",")]
    vk, pb, ib, _, _ = bench.plonk_workload(max(sizes))
    pvk = pkg.PreparedPlonkVk(vk)
    dev = torch.device("cuda:0")
    d_p = torch.frombuffer(bytearray(pb), dtype=torch.uint8).to(dev); d_q = torch.frombuffer(bytearray(ib), dtype=torch.uint8).to(dev)
    stream = torch.cuda.current_stream(dev)
    for n in sizes:
        d_st = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
        ref = None
        for piece in sorted({min(int(x), n) for x in args.pieces.split(",")}):
            pkg.set_plonk_params(piece=5040, workers=8, big_from=1, big_piece=piece)

            def run():
                pvk.verify_batch_device(d_p.data_ptr(), d_q.data_ptr(), d_st.data_ptr(), n, proof_stride=904, stream=stream.cuda_stream, flags=args.flags)
            run()
            t = time.perf_counter()
            for _ in range(args.steps):
                run()
            dt = (time.perf_counter() - t) / args.steps
            st = bytes(d_st.cpu().numpy().tobytes())
            if ref is None:
                ref = st
                assert st.count(bytes([pkg.ACCEPT])) == n - n // 8
            assert st == ref, "statuses differ between plans"
            ms, lanes = pvk.last_timing()
            held = pvk.footprint()
            print(json.dumps({"n": n, "piece": piece, "passes": -(-n // piece), "ms": round(dt * 1e3, 3), "proofs_per_s": round(n / dt), "first_pass_ms": {k: round(v, 3) for k, v in ms.items()},
                              "footprint_gb": round(held[0] / 1e9, 2), "contexts": held[1]}), flush=True)
        del d_st
    pkg.set_plonk_params(piece=5040, workers=8, big_from=0, big_piece=131072)
    for n in sizes:
        p, q = pb[:904 * n], ib[:64 * n]
        pvk.verify_batch(p, q, n)
        t = time.perf_counter()
        for _ in range(args.steps):
            pvk.verify_batch(p, q, n)
        dt = (time.perf_counter() - t) / args.steps
        print(json.dumps({"n": n, "plan": "default, host buffers", "ms": round(dt * 1e3, 3), "proofs_per_s": round(n / dt)}), flush=True)


if __name__ == "__main__":
    main()
