#!/usr/bin/env python3
"""Random-linear-combination batch mode against the exact path: throughput for several invalid fractions.
  python tools/bench_rlc.py [--batch-log2 20] [--steps 3]
Inputs resident in HBM; every measurement checks the status bytes against the generator's expected statuses."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--invalid-every", type=str, default="0,256,16")
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    dev = torch.device("cuda:0")
    n = 1 << args.batch_log2
    rows = []
    for inv in [int(x) for x in args.invalid_every.split(",")]:
        vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540020 + inv, 2, n, invalid_every=inv, agree=True, threads=16)
        if inv:
            # the generator places its invalid proofs periodically; the groups are index classes, so shuffle the records to get the
            # random placement a real batch has (otherwise all invalid proofs fall into 1/16 of the groups)
            import numpy as np
            perm = np.random.default_rng(inv).permutation(n)
            proofs = np.frombuffer(proofs, dtype=np.uint8).reshape(n, 256)[perm].tobytes()
            inputs = np.frombuffer(inputs, dtype=np.uint8).reshape(n, 64)[perm].tobytes()
            exp = np.frombuffer(exp, dtype=np.uint8)[perm].tobytes()
        pvk = pkg.PreparedVk(vk)
        pvk.reserve(n, 0)
        dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev)
        di = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(dev)
        ds = torch.zeros(n, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev)
        row = {"invalid_every": inv, "rejects": exp.count(b"\x00"), "loader_errors": n - exp.count(b"\x00") - exp.count(b"\x01"), "batch": n}
        for name, flags in (("exact", 0), ("rlc", pkg.FLAG_RLC)):
            for it in range(args.steps + 1):
                if it == 1:
                    torch.cuda.synchronize(dev); t = time.perf_counter()
                ds.zero_()
                pvk.verify_batch_device(dp.data_ptr(), di.data_ptr(), ds.data_ptr(), n, 256, 2, 0, st.cuda_stream, flags=flags)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t) / args.steps
            assert bytes(ds.cpu().numpy().tobytes()) == exp, (name, inv)
            row[name + "_ms"] = dt * 1e3
            row[name + "_proofs_per_s"] = n / dt
        row["speedup"] = row["exact_ms"] / row["rlc_ms"]
        row["group_log2"] = int(os.environ.get("BN254_RLC_GROUP_LOG2", "5"))
        rows.append(row)
        print(json.dumps(row), flush=True)
        pvk.close()
        del dp, di, ds


if __name__ == "__main__":
    main()
