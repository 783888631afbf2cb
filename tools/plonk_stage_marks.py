#!/usr/bin/env python3
"""Where does one lane's chain of the PlonK device stages spend its time?  Needs the diagnostics build of the library
(make -C snark-bn254-verifier_amd/csrc BUILD=build_marks OUT=../../tools/exp/libbn254_marks.so EXTRA=-DBN254_PLONK_MARKS), in which the first lane
of each wavefront of k_plonk_stage1 (chain lane, helper lane) and of k_plonk_stage2 stamps the 100 MHz wall clock at marked points (bn254_plonk.hpp::PL_MARK).  Prints the intervals in microseconds."""
import ctypes as C, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BN254_LIB_PATH", os.path.join(ROOT, "tools", "exp", "libbn254_marks.so"))

NAMES = {0: "start", 2: "(chain lane) layout", 3: "gamma transcript", 4: "beta, alpha, zeta transcripts",
         5: "zeta^n, denominators", 13: "inversion (constant-time binary GCD)", 12: "wait at the barrier for the helper lane", 6: "status hand-over", 7: "inverses + public-input sum",
         8: "BSB22 term (hash_to_field from the helper lane)", 9: "opening check + scalars",
         10: "put_term x T1 (GLV split, point digits)", 11: "end of stage 1",
         20: "helper lane: start", 21: "helper: lambda (ChaCha20 + reduce) + clearing the terms", 22: "helper: parse + curve checks of 9 points", 23: "helper: BSB22 hash_to_field",
         16: "stage 2 start", 17: "folding transcript", 18: "powers + folded evaluation", 19: "put_term x (T2 + 2)"}
CHAIN = [(0, 3), (3, 4), (4, 5), (5, 13), (13, 12), (12, 6), (6, 7), (7, 8), (8, 9), (9, 10), (10, 11), (0, 11)]
HELPER = [(20, 21), (21, 22), (22, 23), (20, 23)]
STAGE2 = [(16, 17), (17, 18), (18, 19), (16, 19)]


def main():
    import bench
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    vk, proofs, inputs, _, _ = bench.plonk_workload(n)
    pvk = pkg.PreparedPlonkVk(vk)
    for _ in range(3):
        st = pvk.verify_batch(proofs, inputs, n)
    assert st.count(bytes([pkg.ACCEPT])) == n - n // 8, "statuses differ"
    marks = (C.c_ulonglong * 32)()
    assert pkg.lib().bn254_dbg_plonk_marks(marks) == 0
    m = list(marks)
    out = {}
    for a, b in CHAIN + HELPER + STAGE2:
        out["%d->%d %s" % (a, b, NAMES[b] if (a, b) not in ((0, 11), (20, 23), (16, 19)) else "TOTAL")] = round((m[b] - m[a]) / 100.0, 1)
    print(json.dumps({"n": n, "timing": pvk.last_timing()[0], "intervals_us": out}, indent=1))


if __name__ == "__main__":
    main()
