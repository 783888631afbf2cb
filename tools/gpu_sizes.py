import sys, importlib, time
sys.path.insert(0, '.')
pkg = importlib.import_module("snark-bn254-verifier_amd")
pkg.lib().bn254_set_profiling(1)
for logn in (9, 12, 16, 18, 20):
    n = 1 << logn
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540002, 2, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    t = time.time(); st = pvk.verify_batch(proofs, inputs); dt = time.time() - t
    bad = [i for i in range(n) if st[i] != exp[i]]
    print("n=2^%d ok=%s mismatches=%d first=%s  %.3fs" % (logn, st == exp, len(bad), [(i, st[i], exp[i]) for i in bad[:6]], dt), pvk.last_kernel_ms(), flush=True)
    pvk.close()
