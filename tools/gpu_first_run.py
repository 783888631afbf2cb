"""First GPU bring-up script: device probes vs oracle, a small verify_batch, and a timing."""
import sys, time, random, importlib, ctypes as C
sys.path.insert(0, '.')
pkg = importlib.import_module("snark-bn254-verifier_amd")
from oracle import oracle as O
L = pkg.lib()
P, R = O.P, O.R
random.seed(5)
def chk(rc):
    assert rc == 0, L.bn254_last_error()
n = 1000
a = b"".join(O.be32(random.randrange(P)) for _ in range(n)); b = b"".join(O.be32(random.randrange(P)) for _ in range(n))
out = (C.c_uint8 * (32 * n))()
chk(L.bn254_dbg_fp_mul(a, b, out, C.c_size_t(n), 0))
out = bytes(out)
for i in range(n):
    x = int.from_bytes(a[32*i:32*i+32], 'big'); y = int.from_bytes(b[32*i:32*i+32], 'big')
    assert int.from_bytes(out[32*i:32*i+32], 'big') == x * y % P, i
print("gpu fp_mul ok")
n = 64
A12 = b"".join(O.be32(random.randrange(P)) for _ in range(12 * n)); B12 = b"".join(O.be32(random.randrange(P)) for _ in range(12 * n))
for op, oop in ((0, 0), (1, 1), (2, 2), (4, 3)):
    out = (C.c_uint8 * (384 * n))()
    chk(L.bn254_dbg_fp12_op(op, A12, B12 if op == 0 else None, out, C.c_size_t(n), 0))
    out = bytes(out)
    for i in range(n):
        exp = O.fp12_op(oop, A12[384*i:384*i+384], B12[384*i:384*i+384] if op == 0 else None)
        assert out[384*i:384*i+384] == exp, (op, i)
print("gpu fp12 mul/sqr/inv/frob ok")
g1 = O.g1_gen(); g2 = O.g2_gen()
n = 8
g1s = b"".join(O.g1_mul(g1, random.randrange(1, R)) for _ in range(n)); g2s = b"".join(O.g2_mul(g2, random.randrange(1, R)) for _ in range(n))
out = (C.c_uint8 * (384 * n))()
t = time.time(); chk(L.bn254_dbg_pairing(g1s, g2s, out, C.c_size_t(n), 0)); print("gpu pairing probe %.2fs" % (time.time() - t))
out = bytes(out)
for i in range(n):
    assert out[384*i:384*i+384] == O.pairing(g1s[64*i:64*i+64], g2s[128*i:128*i+128]), i
print("gpu pairing == oracle")
fl = (C.c_uint8 * n)()
chk(L.bn254_dbg_g2_subgroup(g2s, fl, C.c_size_t(n), 0)); assert bytes(fl) == b"\x01" * n
print("gpu subgroup ok")
vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540001, 2, 512, invalid_every=4, agree=True, threads=16)
pvk = pkg.PreparedVk(vk, pkg.VK_REFERENCE)
t = time.time(); st = pvk.verify_batch(proofs, inputs); print("gpu verify_batch(512) %.3fs" % (time.time() - t))
print("status histogram", {k: st.count(bytes([k])) for k in set(st)})
assert st == exp, [(i, st[i], exp[i]) for i in range(len(st)) if st[i] != exp[i]][:10]
ost = O.groth16_verify_many(proofs[:256*64], 256, vk, inputs[:64*64], 2, 64, O.MODE_REFERENCE)
assert ost == st[:64]
print("gpu verify_batch == expected == oracle")
L.bn254_set_profiling(1)
for n in (4096, 65536):
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540002, 2, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk, pkg.VK_REFERENCE)
    pvk.verify_batch(proofs[:256*256], inputs[:64*256], 256)
    t = time.time(); st = pvk.verify_batch(proofs, inputs); dt = time.time() - t
    print("n=%d: %.3fs => %.0f proofs/s (incl. H2D/D2H), ok=%s" % (n, dt, n / dt, st == exp), pvk.last_kernel_ms())
