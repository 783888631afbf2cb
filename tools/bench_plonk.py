#!/usr/bin/env python3
"""PlonK batch verification rate (BASELINE configs[3]: batch 4096, SP1 circuit, 2 public inputs) on one MI355X: bench.plonk_config as a
stand-alone command (the default bench.py run carries the same measurement in its `configs` block).  Prints one JSON line."""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=64)
    ap.add_argument("--no-in-flight", action="store_true", help="skip the two / four calls in flight measurement (kernel traces: one call at a time only)")
    args = ap.parse_args()
    import bench
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    bench.plonk_config(pkg, args.batch, 2, 1, 0, in_flight=False)      # warms the GPU: the peak probe reads low on idle clocks
    bench.measure_valu_peak(pkg)
    r = bench.plonk_config(pkg, args.batch, args.steps, args.warmup, args.cpu_sample, in_flight=args.batch <= 8192 and not args.no_in_flight)
    r.update({"metric": "PlonK verifies/sec at batch=%d (proofs resident in HBM; `host_buffers` beside it)" % args.batch, "n_gpus": 1, "warmup": args.warmup,
              "higher_is_better": True, "dtype": "int64", "data": "reference fixtures + mutations", "config": {"workload": r["workload"]}})
    print(json.dumps(r))


if __name__ == "__main__":
    main()
