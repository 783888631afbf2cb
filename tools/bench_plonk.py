#!/usr/bin/env python3
"""PlonK batch verification rate (BASELINE configs[3]: batch 4096, SP1 circuit, 2 public inputs) on one MI355X.
The workload is the reference's 4 PlonK fixtures plus mutated copies (every 8th proof invalid), host buffers in, status bytes out:
the host stages (transcripts, scalar-field arithmetic) are part of the path, so the rate is PCIe- and host-inclusive.
Prints one JSON line; cpu_baseline = the oracle's reference-faithful PlonK verifier on one core."""
import argparse, importlib, json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=64)
    args = ap.parse_args()
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    from oracle import oracle as O
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
    vk = open(os.path.join(ROOT, "tests", "golden", "plonk_vk.bin"), "rb").read()
    base = [(bytes.fromhex(f["raw_proof"]), b"".join(int(x).to_bytes(32, "big") for x in f["public_inputs"])) for f in fx.values() if f["variant"] == "plonk"]
    rng = random.Random(4)
    proofs, inputs = [], []
    for i in range(args.batch):
        p, q = base[i % len(base)]
        if i % 8 == 7:
            q = bytearray(q); q[rng.randrange(64)] ^= 1 << rng.randrange(8); q = bytes(q)
        proofs.append(p); inputs.append(q)
    pb, ib = b"".join(proofs), b"".join(inputs)
    pvk = pkg.PreparedPlonkVk(vk)
    for _ in range(args.warmup):
        st = pvk.verify_batch(pb, ib)
    t = time.perf_counter()
    for _ in range(args.steps):
        st = pvk.verify_batch(pb, ib)
    dt = time.perf_counter() - t
    # the dominant GPU kernel against the VALU peak: the merged k_g1_scalar_mul launch of stage 2 (one lane per scalar multiplication,
    # multiply-adds per lane from the code object: tools/count_mads.py), and the pairing check on the cooperative kernel
    stage_ms, lanes = pvk.last_timing()
    km = json.load(open(os.path.join(ROOT, "profiles", "kernel_mads.json")))["kernels"]
    peak = 35.1e12
    env = os.environ.get
    split = (lanes[1] * 2 <= 65536 or env("BN254_MSM_SPLIT") == "1") and env("BN254_MSM_SPLIT") != "0"      # bn254_g1_msm_split
    w2 = lanes[1] * (2 if split else 1) <= 65536 and env("BN254_MSM_W2", "1") != "0"                          # bn254_g1_msm_tab_lanes
    e = km["k_g1_scalar_mul" + ("_split" if split else "") + ("_w2" if w2 else "")]
    sm = e["mads_per_proof_launch"]                                                            # the longest lane's chain
    useful = km["k_g1_scalar_mul"]["mads_per_proof_launch"] * lanes[1]                          # the work of the unsplit algorithm
    ach = useful / (stage_ms["k_g1_scalar_mul_stage2"] * 1e-3)
    n_lanes = lanes[1] * (2 if split else 1)
    roofline = {"bound": "valu", "kernel": "k_g1_scalar_mul" + ("<split>" if split else "") + ("<two-bit windows>" if w2 else ""), "unit": "T mad/s", "peak": peak / 1e12, "achieved": ach / 1e12, "frac": ach / peak,
                "avg_launch_ms": stage_ms["k_g1_scalar_mul_stage2"], "terms_per_launch": lanes[1], "lanes_per_launch": n_lanes, "mads_per_term_unsplit": km["k_g1_scalar_mul"]["mads_per_proof_launch"],
                "longest_lane_chain_mads": sm, "traffic": None,
                "note": "%d lanes = %.2f wavefronts per SIMD: the launch lasts as long as its longest lane's chain (%d multiply-adds); achieved = multiply-adds of the "
                        "one-lane-per-term algorithm / launch time; peak = measured issue rate with full occupancy (profiles/r01_ubench_valu.txt)" % (n_lanes, n_lanes / 64 / 1024.0, int(sm))}
    pc = km.get("k_coop12_miller_fixed")
    pairing = None
    if pc and args.batch <= 40960:
        a2 = pc["mads_per_proof_launch"] * min(args.batch, 65536) / (stage_ms["pairing_check"] * 1e-3)
        pairing = {"kernel": "k_coop12_miller_fixed", "ms": stage_ms["pairing_check"], "mads_per_proof": pc["mads_per_proof_launch"], "achieved": a2 / 1e12, "frac": a2 / peak}
    m = min(args.cpu_sample, args.batch)
    t = time.perf_counter()
    ref = bytes(O.plonk_verify(proofs[i], vk, [int.from_bytes(inputs[i][:32], "big"), int.from_bytes(inputs[i][32:], "big")]) for i in range(m))
    cdt = time.perf_counter() - t
    assert st[:m] == ref, "GPU statuses differ from the oracle"
    assert st.count(bytes([pkg.ACCEPT])) == args.batch - args.batch // 8
    print(json.dumps({"metric": "PlonK verifies/sec at batch=%d (host buffers in, status bytes out)" % args.batch, "value": args.batch * args.steps / dt,
                      "unit": "proofs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
                      "higher_is_better": True, "dtype": "int64", "data": "reference fixtures + mutations",
                      "config": {"workload": "BASELINE configs[3]: PlonK batch %d, 904-byte proofs, 2 public inputs, 1/8 invalid" % args.batch},
                      "roofline": roofline, "pairing_check": pairing, "stages_ms": {k: round(v, 3) for k, v in stage_ms.items()},
                      "cpu_baseline": {"value": m / cdt, "unit": "proofs/s", "cores": 1, "kind": "port", "sample": "first %d proofs, %.1f s" % (m, cdt)}}))


if __name__ == "__main__":
    main()
