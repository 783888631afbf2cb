"""One-off check of the chunking path: a batch larger than 2^20 (two workspace chunks, launches capped at 786 432 proofs)."""
import sys, importlib, time
sys.path.insert(0, '.')
pkg = importlib.import_module("snark-bn254-verifier_amd")
n = (1 << 20) + 300_001
t = time.time()
vk, proofs, inputs, exp = pkg.synth_groth16(0xB254BEEF, 2, n, invalid_every=16, agree=True, threads=16)
print("generated %d proofs in %.1f s" % (n, time.time() - t), flush=True)
pvk = pkg.PreparedVk(vk)
for streams in ("default",):
    t = time.time(); st = pvk.verify_batch(proofs, inputs); dt = time.time() - t
    print("verify_batch (host buffers): %.1f ms, %.2f M proofs/s, statuses ok = %s" % (dt * 1e3, n / dt / 1e6, st == exp), flush=True)
    assert st == exp
