#!/usr/bin/env python3
"""PlonK at large passes with the MSM rows' JOINT group size forced (BN254_MSM_JOINT=g in the environment of THIS process: read once by the library): one JSON line
per batch size.  usage: BN254_MSM_JOINT=4 python tools/bench_plonk_joint.py 65536,262144"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "65536").split(",")]
    vk, pb, ib, _, _ = bench.plonk_workload(max(sizes))
    pvk = pkg.PreparedPlonkVk(vk)
    flags = pkg.FLAG_RLC if os.environ.get("BENCH_PLONK_RLC") == "1" else 0      # BENCH_PLONK_RLC=1: the pairing checks batched across proofs
    for n in sizes:
        p, q = pb[:904 * n], ib[:64 * n]
        st = pvk.verify_batch(p, q, n, flags=flags)
        assert st.count(bytes([pkg.ACCEPT])) == n - n // 8
        steps = 4
        t = time.perf_counter()
        for _ in range(steps):
            st = pvk.verify_batch(p, q, n, flags=flags)
        dt = (time.perf_counter() - t) / steps
        ms, _ = pvk.last_timing()
        print(json.dumps({"joint": os.environ.get("BN254_MSM_JOINT", "auto"), "rlc": bool(flags), "n": n, "ms": round(dt * 1e3, 3), "proofs_per_s": round(n / dt),
                          "rows_digest_ms": round(ms["k_g1_msm_rows_digest"], 3), "rows_kzg_ms": round(ms["k_g1_msm_rows_kzg"], 3), "pairing_ms": round(ms["pairing_check"], 3)}), flush=True)


if __name__ == "__main__":
    main()
