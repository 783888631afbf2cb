"""GPU parity tests (run on the MI355X box): everything goes through the C ABI of libbn254_verify_amd.so and is compared
bit-for-bit with the CPU oracle (integer work: no tolerance anywhere)."""
import ctypes as C
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def be(v):
    return int(v).to_bytes(32, "big")


def _chk(L, rc):
    assert rc == 0, L.bn254_last_error()


@pytest.fixture(scope="module")
def L(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; the product has no CPU fallback"
    return pkg.lib()


@pytest.fixture(scope="module")
def workload(pkg):
    """512 synthetic proofs, every 4th invalid, cycling through the 5 failure classes."""
    return pkg.synth_groth16(0xB2540001, 2, 512, invalid_every=4, agree=True, threads=16)


def test_device_fp_mul(L):
    rng = random.Random(5)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 1 << 253, (1 << 254) - 1, 3, 9]
    xs = edge + [rng.randrange(P) for _ in range(4096 - len(edge))]
    ys = list(reversed(edge)) + [rng.randrange(P) for _ in range(4096 - len(edge))]
    n = len(xs)
    out = (C.c_uint8 * (32 * n))()
    _chk(L, L.bn254_dbg_fp_mul(b"".join(map(be, xs)), b"".join(map(be, ys)), out, C.c_size_t(n), 0))
    out = bytes(out)
    for i in range(n):
        assert int.from_bytes(out[32 * i:32 * i + 32], "big") == xs[i] * ys[i] % P, i


def test_device_fp12_ops(L, O):
    rng = random.Random(6)
    n = 64
    a = b"".join(be(rng.randrange(P)) for _ in range(12 * n)); b = b"".join(be(rng.randrange(P)) for _ in range(12 * n))
    for op, oop in ((0, 0), (1, 1), (2, 2), (4, 3)):  # mul, sqr, inv, frobenius
        out = (C.c_uint8 * (384 * n))()
        _chk(L, L.bn254_dbg_fp12_op(op, a, b if op == 0 else None, out, C.c_size_t(n), 0))
        out = bytes(out)
        for i in range(n):
            assert out[384 * i:384 * i + 384] == O.fp12_op(oop, a[384 * i:384 * i + 384], b[384 * i:384 * i + 384] if op == 0 else None), (op, i)
    # cyclotomic squaring after the easy part of the final exponentiation
    out = (C.c_uint8 * (384 * 8))()
    _chk(L, L.bn254_dbg_fp12_op(3, a, None, out, C.c_size_t(8), 0))
    out = bytes(out)
    for i in range(8):
        x = a[384 * i:384 * i + 384]
        c = O.fp12_op(0, O.fp12_op(7, x), O.fp12_op(2, x)); c = O.fp12_op(0, O.fp12_op(4, c), c)
        assert out[384 * i:384 * i + 384] == O.fp12_op(1, c), i


def test_device_pairing_and_bilinearity(L, O):
    rng = random.Random(7)
    g1, g2 = O.g1_gen(), O.g2_gen()
    n = 16
    sa = [rng.randrange(1, R) for _ in range(n)]; sb = [rng.randrange(1, R) for _ in range(n)]
    g1s = b"".join(O.g1_mul(g1, a) for a in sa); g2s = b"".join(O.g2_mul(g2, b) for b in sb)
    out = (C.c_uint8 * (384 * n))()
    _chk(L, L.bn254_dbg_pairing(g1s, g2s, out, C.c_size_t(n), 0))
    out = bytes(out)
    for i in range(n):
        assert out[384 * i:384 * i + 384] == O.pairing(g1s[64 * i:64 * i + 64], g2s[128 * i:128 * i + 128]), i
    # e(aP, bQ) == e(abP, Q), on the device
    g1ab = b"".join(O.g1_mul(g1, a * b % R) for a, b in zip(sa, sb))
    out2 = (C.c_uint8 * (384 * n))()
    _chk(L, L.bn254_dbg_pairing(g1ab, g2 * n, out2, C.c_size_t(n), 0))
    assert bytes(out2) == out


def _twist_point(O, rng):
    bt = O.fp2_op(2, O.fp2_op(3, (9, 1)), (3, 0))
    while True:
        x = (rng.randrange(P), rng.randrange(P))
        rhs = O.fp2_op(0, O.fp2_op(2, O.fp2_op(5, x), x), bt)
        y = O.fp2_op(4, rhs)
        if y != (0, 0) and O.fp2_op(5, y) == rhs:
            return be(x[1]) + be(x[0]) + be(y[1]) + be(y[0])


def test_device_pairing_on_reference_plonk_fixtures(L, O, fixtures):
    """The reference's own end-to-end fixtures through the device pairing: for each of the 4 PlonK proofs of
    examples/binaries/ the oracle derives the two (G1, G2) operands of the final KZG check (plonk/kzg.rs:175-187); the device
    computes e(P0, Q0) and e(P1, Q1) (Miller program + final exponentiation of the product path) and their product must be 1.
    A tampered public input must not give 1."""
    fx, vk = fixtures
    one = be(1) + bytes(352)
    good = 0
    for name, f in fx.items():
        if f["variant"] != "plonk":
            continue
        proof = bytes.fromhex(f["raw_proof"])
        pis = [int(x) for x in f["public_inputs"]]
        for tamper in (0, 1):
            st, ps, qs = O.plonk_pairing_inputs(proof, vk, [pis[0], pis[1] + tamper])
            if tamper and st not in (O.ACCEPT, O.ERR_PAIRING_FAILED):
                continue  # rejected before the pairing (opening mismatch): nothing to feed the device
            out = (C.c_uint8 * (384 * 2))()
            _chk(L, L.bn254_dbg_pairing(ps[0] + ps[1], qs[0] + qs[1], out, C.c_size_t(2), 0))
            out = bytes(out)
            assert out[:384] == O.pairing(ps[0], qs[0]) and out[384:] == O.pairing(ps[1], qs[1]), name
            prod = O.fp12_op(0, out[:384], out[384:])
            if tamper:
                assert prod != one, name
            else:
                assert st == O.ACCEPT and prod == one, name
                good += 1
    assert good == 4


def test_device_g2_subgroup(L, O):
    rng = random.Random(8)
    g2 = O.g2_gen()
    pts = [O.g2_mul(g2, rng.randrange(1, R)) for _ in range(6)] + [_twist_point(O, rng) for _ in range(6)]
    tq = _twist_point(O, rng)
    pts.append(O.g2_mul(tq, 2 * P - R))          # cofactor cleared: in G2
    low = O.g2_mul(tq, R)                        # order divides the cofactor: not in G2
    if low != bytes(128):
        pts.append(low)
    off = bytearray(pts[0]); off[127] ^= 1; pts.append(bytes(off))  # not on the twist at all
    n = len(pts)
    fl = (C.c_uint8 * n)()
    _chk(L, L.bn254_dbg_g2_subgroup(b"".join(pts), fl, C.c_size_t(n), 0))
    for i, q in enumerate(pts):
        assert fl[i] == (1 if O.g2_subgroup_check(q) == 1 else 0), i


def test_device_g2_subgroup_from_miller_loop(L, O):
    """The product's r-torsion test (ate relation on the Miller loop's final point) against the oracle's naive check: G2 points,
    random twist points, cofactor-cleared points, points of the cofactor group (also of small prime order) and G2 + cofactor sums."""
    rng = random.Random(18)
    g1, g2 = O.g1_gen(), O.g2_gen()
    pts = [O.g2_mul(g2, rng.randrange(1, R)) for _ in range(40)] + [_twist_point(O, rng) for _ in range(16)]
    tq = _twist_point(O, rng)
    h2 = 2 * P - R
    pts.append(O.g2_mul(tq, h2))
    cof = O.g2_mul(tq, R)
    assert cof != bytes(128)
    pts.append(cof)
    for ell in (10069, 5864401):
        small = O.g2_mul(cof, h2 // ell)
        if small != bytes(128):
            pts.append(small)
            pts.append(O.g2_add(O.g2_mul(g2, rng.randrange(1, R)), small))
    n = len(pts)
    g1s = b"".join(O.g1_mul(g1, rng.randrange(1, R)) for _ in range(n))
    fl = (C.c_uint8 * n)()
    _chk(L, L.bn254_dbg_g2_subgroup_ate(g1s, b"".join(pts), fl, C.c_size_t(n), 0))
    exp = [1 if O.g2_subgroup_check(q) == 1 else 0 for q in pts]
    assert list(fl) == exp
    assert sum(exp) == 41 and len(exp) >= 60


def test_verify_batch_vs_oracle_all_classes(pkg, O, workload, L):
    vk, proofs, inputs, exp = workload
    n = 512
    assert set(exp) == {0, 1, 2, 3, 4}
    for mode, omode in ((pkg.VK_REFERENCE, O.MODE_REFERENCE), (pkg.VK_GNARK, O.MODE_GNARK)):
        pvk = pkg.PreparedVk(vk, mode)
        st = pvk.verify_batch(proofs, inputs)
        assert st == exp
        m = 96  # oracle (reference-faithful CPU path) on a prefix that contains every class
        assert O.groth16_verify_many(proofs[:256 * m], 256, vk, inputs[:64 * m], 2, m, omode) == st[:m]
        pvk.close()


def _plonk_cases(O, fixtures, rng, per_fixture):
    """(proof, inputs) pairs built from the reference's 4 PlonK fixtures: the originals and single-byte / single-bit mutations of
    the proof and the public inputs that exercise every status the PlonK path can return."""
    fx, vk = fixtures
    cases = []
    for name, f in fx.items():
        if f["variant"] != "plonk":
            continue
        proof = bytes.fromhex(f["raw_proof"])
        pis = b"".join(be(int(x)) for x in f["public_inputs"])
        cases.append((proof, pis))
        for _ in range(per_fixture):
            p, q = bytearray(proof), bytearray(pis)
            kind = rng.randrange(8)
            if kind == 0: q[rng.randrange(64)] ^= 1 << rng.randrange(8)                 # public input
            elif kind == 1: p[516 + rng.randrange(32 * 7)] ^= 1 << rng.randrange(8)     # a claimed value
            elif kind == 2: p[rng.randrange(512)] ^= 1 << rng.randrange(8)              # a commitment coordinate
            elif kind == 3: p[0:32] = bytes([0xff]) * 32                                # coordinate >= p
            elif kind == 4: p[515] ^= 1                                                 # number of claimed values
            elif kind == 5: p[516 + 32 * 7 + 64 + rng.randrange(32)] ^= 1               # z(zeta omega)
            elif kind == 6: p[448 + rng.randrange(64)] ^= 1 << rng.randrange(8)         # batched opening proof
            else: p[516 + 32 * 7 + 64 + 32 + 3] ^= 1                                    # number of BSB22 commitments
            cases.append((bytes(p), bytes(q)))
    return cases, vk


def test_plonk_reference_fixtures_and_mutations_vs_oracle(pkg, O, fixtures):
    """BASELINE configs[3]: the reference's PlonK fixtures (examples/binaries/*_plonk_proof.bin, vk from the guest ELF) must be
    accepted by the GPU path, and every mutated proof must get the oracle's status byte."""
    rng = random.Random(31)
    cases, vk = _plonk_cases(O, fixtures, rng, 30)
    pvk = pkg.PreparedPlonkVk(vk)
    assert pvk.n_public == 2
    proofs = b"".join(c[0] for c in cases); inputs = b"".join(c[1] for c in cases)
    st = pvk.verify_batch(proofs, inputs, proof_stride=904)
    exp = bytes(O.plonk_verify(c[0], vk, [int.from_bytes(c[1][:32], "big"), int.from_bytes(c[1][32:], "big")]) for c in cases)
    assert st == exp
    assert st[0] == pkg.ACCEPT and st.count(bytes([pkg.ACCEPT])) >= 4 and len(set(st)) >= 4, sorted(set(st))
    # single-proof entry point (PlonkVerifier::verify), wrong number of public inputs
    assert pkg.PlonkVerifier.verify(cases[0][0], vk, [cases[0][1][:32], cases[0][1][32:]]) == pkg.ACCEPT
    assert pkg.PlonkVerifier.verify(cases[0][0], vk, [cases[0][1][:32]]) == O.plonk_verify(cases[0][0], vk, [int.from_bytes(cases[0][1][:32], "big")])
    pvk.close()


def test_plonk_batch_4096(pkg, O, fixtures):
    """Batch of 4096 (the size BASELINE quotes): fixtures and mutations repeated; statuses must be those of the small run."""
    rng = random.Random(32)
    cases, vk = _plonk_cases(O, fixtures, rng, 7)
    pvk = pkg.PreparedPlonkVk(vk)
    small = pvk.verify_batch(b"".join(c[0] for c in cases), b"".join(c[1] for c in cases))
    reps = 4096 // len(cases)
    big = pvk.verify_batch(b"".join(c[0] for c in cases) * reps, b"".join(c[1] for c in cases) * reps)
    assert big == small * reps
    # ... and the small run's statuses are the ORACLE's, case by case (not only self-consistent)
    exp = bytes(O.plonk_verify(c[0], vk, [int.from_bytes(c[1][:32], "big"), int.from_bytes(c[1][32:], "big")]) for c in cases)
    assert small == exp and len(set(exp)) >= 3
    pvk.close()


def test_plonk_calls_in_flight_on_one_key(pkg, O, fixtures):
    """Several host threads on ONE prepared PlonK key (the library hands each call a context of the key's pool): batches of different
    sizes -- one of them cut into two sub-batches, so that it needs two contexts at once -- must give the statuses of the calls made one after the
    other, and those are the oracle's on the distinct cases."""
    import threading
    rng = random.Random(33)
    cases, vk = _plonk_cases(O, fixtures, rng, 5)
    exp = bytes(O.plonk_verify(c[0], vk, [int.from_bytes(c[1][:32], "big"), int.from_bytes(c[1][32:], "big")]) for c in cases)
    pb, ib = b"".join(c[0] for c in cases), b"".join(c[1] for c in cases)
    pvk = pkg.PreparedPlonkVk(vk)
    sizes = [4096 // len(cases), 1, 17000 // len(cases), 37, 200, 3]          # repetitions of the case list per call
    serial = [pvk.verify_batch(pb * r, ib * r) for r in sizes]
    assert all(s == exp * r for s, r in zip(serial, sizes))
    out = [[None] * 3 for _ in sizes]
    def work(j):
        for it in range(3):
            out[j][it] = pvk.verify_batch(pb * sizes[j], ib * sizes[j])
    th = [threading.Thread(target=work, args=(j,)) for j in range(len(sizes))]
    for t in th: t.start()
    for t in th: t.join()
    for j in range(len(sizes)):
        assert all(o == serial[j] for o in out[j]), j
    pvk.close()


def test_plonk_batch_sizes_around_the_window_table_limit(pkg, O, fixtures):
    """Regression (round 3): the scratch of the two-bit-window scalar multiplications was sized from a context's CAPACITY (a multiple of 256 proofs), while
    the launch that uses it is chosen by the batch's own lane count -- 5000 proofs on a context of 5120 (13 terms: 65 000 lanes, 56 320 allocated), 2500 on
    2560 with two lanes per term, 4500 on a context that an earlier call had grown to 5120 -- and wrote past its end.  Sizes on both sides of each limit
    (65 536 lanes: 5041 proofs x 13 terms, 2520 x 13 x 2), one prepared key throughout so that the contexts keep their capacity; statuses against the oracle's."""
    rng = random.Random(34)
    cases, vk = _plonk_cases(O, fixtures, rng, 5)
    exp = bytes(O.plonk_verify(c[0], vk, [int.from_bytes(c[1][:32], "big"), int.from_bytes(c[1][32:], "big")]) for c in cases)
    pb, ib = b"".join(c[0] for c in cases), b"".join(c[1] for c in cases)
    pvk = pkg.PreparedPlonkVk(vk)
    k = len(cases)
    for n in (2496, 2520, 2544, 4992, 5040, 5064, 4488, 4344, 2184, 9984, 10080, 20064):
        reps, tail = divmod(n, k)
        st = pvk.verify_batch(pb * reps + pb[:904 * tail], ib * reps + ib[:64 * tail])
        assert st == exp * reps + exp[:tail], n
    pvk.close()


@pytest.mark.parametrize("n_public", [17, 40, 300, 1024])
def test_many_public_inputs_vs_oracle(pkg, O, n_public):
    """Keys with many public inputs (BASELINE configs[4]: 1024) take the wide MSM path: the inputs of one proof are summed by
    n_public/16 lanes and reduced.  Every status class, both key readings, the oracle on a prefix, and a wrong input count."""
    n = 80 if n_public == 1024 else 200
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540005 + n_public, n_public, n, invalid_every=4, agree=True, threads=8)
    assert len(vk) == 292 + 32 * (n_public + 1) + 4 + 128 and set(exp) == {0, 1, 2, 3, 4}
    for mode, omode in ((pkg.VK_REFERENCE, O.MODE_REFERENCE), (pkg.VK_GNARK, O.MODE_GNARK)):
        pvk = pkg.PreparedVk(vk, mode)
        assert pvk.n_public == n_public
        st = pvk.verify_batch(proofs, inputs)
        assert st == exp
        m = 12 if n_public == 1024 else 24
        assert O.groth16_verify_many(proofs[:256 * m], 256, vk, inputs[:32 * n_public * m], n_public, m, omode) == st[:m]
        if mode == pkg.VK_REFERENCE:
            # one input too few: PrepareInputsFailed after every loader error (groth16/verify.rs:54-56)
            short = b"".join(inputs[32 * n_public * i:32 * n_public * i + 32 * (n_public - 1)] for i in range(n))
            st2 = pvk.verify_batch(proofs, short, n_public=n_public - 1)
            assert all(b == (a if a in (2, 3, 4) else pkg.ERR_INPUT_LEN) for a, b in zip(exp, st2))
            ref2 = O.groth16_verify_many(proofs[:256 * 8], 256, vk, short[:32 * (n_public - 1) * 8], n_public - 1, 8, omode)
            assert st2[:8] == ref2
        pvk.close()


def test_ragged_empty_and_strides(pkg, O, workload, L):
    vk, proofs, inputs, exp = workload
    pvk = pkg.PreparedVk(vk)
    assert pvk.verify_batch(b"", b"", n=0) == b""
    for n in (1, 63, 65, 257, 300):  # partial waves and partial blocks
        assert pvk.verify_batch(proofs[:256 * n], inputs[:64 * n], n) == exp[:n], n
    # gnark raw proofs are 324 bytes (A|B|C|u32 nCommitments|64-byte PoK); bytes past 256 are ignored (groth16/converter.rs:15-25)
    n = 70
    raw = b"".join(proofs[256 * i:256 * i + 256] + b"\0\0\0\0" + bytes(64) for i in range(n))
    assert pvk.verify_batch(raw, inputs[:64 * n], n, proof_stride=324) == exp[:n]
    # a stride that is not a multiple of 4 takes the unaligned load path
    raw = b"".join(proofs[256 * i:256 * i + 256] + b"\xa5" for i in range(n))
    assert pvk.verify_batch(raw, inputs[:64 * n], n, proof_stride=257) == exp[:n]
    pvk.close()


def test_real_sp1_groth16_proof_bytes(pkg, O, fixtures, workload, L):
    """The 4 Groth16 fixtures of the reference (vk unavailable): their A, B, C must parse as valid points; against a synthetic key
    the verdict is REJECT on both sides."""
    fx, _ = fixtures
    vk = workload[0]
    pvk = pkg.PreparedVk(vk)
    for name, f in sorted(fx.items()):
        if f["variant"] != "groth16":
            continue
        raw = bytes.fromhex(f["raw_proof"])
        pis = b"".join(be(int(x)) for x in f["public_inputs"])
        got = pvk.verify_batch(raw, pis, 1, proof_stride=324)
        assert got == bytes([O.groth16_verify(raw, vk, [int(x) for x in f["public_inputs"]])]) == bytes([pkg.REJECT]), name
    pvk.close()


def test_error_precedence_and_edge_inputs(pkg, O, workload, L):
    vk, proofs, inputs, exp = workload
    good = next(i for i in range(512) if exp[i] == 1)
    base = bytearray(proofs[256 * good:256 * good + 256]); inp = inputs[64 * good:64 * good + 64]
    cases = []

    def add(p, i=inp):
        cases.append((bytes(p), bytes(i)))

    add(base)                                                          # valid
    p = bytearray(base); p[0:64] = bytes(64); add(p)                   # A = (0,0): not on curve (infinity is not representable)
    p = bytearray(base); p[0:32] = be(P); add(p)                       # A.x == p exactly: not a member
    p = bytearray(base); p[32:64] = be(P - 1); add(p)                  # A.y = p-1: member, off curve
    p = bytearray(base); p[64:96] = b"\xff" * 32; add(p)               # B.x.c1 >= p
    p = bytearray(base); p[191] ^= 1; add(p)                           # B off the twist
    p = bytearray(base); p[192:224] = b"\xff" * 32; add(p)             # C.x >= p
    p = bytearray(base); p[255] ^= 1; add(p)                           # C off curve
    p = bytearray(base); p[0:32] = b"\xff" * 32; p[255] ^= 1; add(p)   # A and C both bad: A's error wins
    p = bytearray(base); p[191] ^= 1; p[255] ^= 1; add(p)              # B (curve) before C
    bad_b = next(i for i in range(512) if exp[i] == 4)
    p = bytearray(base); p[64:192] = proofs[256 * bad_b + 64:256 * bad_b + 192]; p[255] ^= 1; add(p)   # B subgroup error before C's
    x0 = int.from_bytes(inp[:32], "big")
    add(base, be(x0 + R) + inp[32:])                                   # public input >= r: same verdict as x (no range check, used mod r)
    add(base, b"\xff" * 32 + inp[32:])                                 # 2^256 - 1
    add(base, bytes(64))                                               # zero inputs: REJECT
    pr = b"".join(c[0] for c in cases); ii = b"".join(c[1] for c in cases)
    n = len(cases)
    pvk = pkg.PreparedVk(vk)
    got = pvk.verify_batch(pr, ii, n)
    want = O.groth16_verify_many(pr, 256, vk, ii, 2, n, O.MODE_REFERENCE)
    assert got == want
    assert got[0] == 1 and got[1] == 3 and got[2] == 2 and got[8] == 2 and got[9] == 3 and got[10] == 4 and got[11] == 1 and got[13] == 0
    # wrong number of public inputs: Err(PrepareInputsFailed) unless the proof itself is invalid (loader errors come first)
    got = pvk.verify_batch(pr, ii + bytes(32 * n), n, n_public=3)
    want = bytes(O.groth16_verify(cases[k][0], vk, [0, 0, 0]) for k in range(n))
    assert got == want and got[0] == pkg.ERR_INPUT_LEN and got[1] == 3
    pvk.close()


def test_mode_disagreement_on_device(pkg, O, L):
    for seed in range(1, 12):
        vk, proofs, inputs, exp = pkg.synth_groth16(seed, 1, 4, invalid_every=0, agree=False, threads=2)
        r = O.groth16_verify_many(proofs, 256, vk, inputs, 1, 4, O.MODE_REFERENCE)
        if r != exp:
            for mode, want in ((pkg.VK_REFERENCE, r), (pkg.VK_GNARK, exp)):
                pvk = pkg.PreparedVk(vk, mode)
                assert pvk.verify_batch(proofs, inputs) == want
                pvk.close()
            return
    pytest.fail("no disagreeing key found")


def test_single_verify_mirrors_reference_api(pkg, O, workload, L):
    vk, proofs, inputs, exp = workload
    V = pkg.Groth16Verifier
    for i in (0, 3, 11, 15, 19):
        xs = [int.from_bytes(inputs[64 * i + 32 * k:64 * i + 32 * k + 32], "big") for k in range(2)]
        assert V.verify(proofs[256 * i:256 * i + 256], vk, xs) == exp[i] == O.groth16_verify(proofs[256 * i:256 * i + 256], vk, xs)
    assert V.verify(proofs[:200], vk, [1, 2]) == pkg.ERR_MALFORMED == O.groth16_verify(proofs[:200], vk, [1, 2])
    assert V.verify(proofs[:256], vk[:300], [1, 2]) == pkg.ERR_MALFORMED == O.groth16_verify(proofs[:256], vk[:300], [1, 2])
    st = V.verify_batch([proofs[256 * i:256 * i + 256] for i in range(8)], vk,
                        [[int.from_bytes(inputs[64 * i + 32 * k:64 * i + 32 * k + 32], "big") for k in range(2)] for i in range(8)])
    assert bytes(st) == exp[:8]


def test_device_resident_entry_and_streams(pkg, workload, L):
    import torch
    vk, proofs, inputs, exp = workload
    dev = torch.device("cuda:0")
    pvk = pkg.PreparedVk(vk)
    n = 512
    pvk.reserve(n)
    dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev)
    di = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(dev)
    side = torch.cuda.Stream(dev)
    for stream in (torch.cuda.current_stream(dev), side):
        ds = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        pvk.verify_batch_device(dp.data_ptr(), di.data_ptr(), ds.data_ptr(), n, 256, 2, 0, stream.cuda_stream)
        stream.synchronize()
        assert bytes(ds.cpu().numpy().tobytes()) == exp
    pvk.close()


def test_batch_4096_properties(pkg, O, L):
    """BASELINE config 2 size: statuses must equal the generator's prediction for all 4096 proofs; idempotence; a permuted
    batch gives the permuted statuses (proofs are independent); an oracle spot-check on a strided sample."""
    n = 4096
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540002, 2, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    st = pvk.verify_batch(proofs, inputs)
    assert st == exp
    assert pvk.verify_batch(proofs, inputs) == st
    rng = random.Random(9)
    perm = list(range(n)); rng.shuffle(perm)
    pp = b"".join(proofs[256 * j:256 * j + 256] for j in perm); ip = b"".join(inputs[64 * j:64 * j + 64] for j in perm)
    assert pvk.verify_batch(pp, ip) == bytes(exp[j] for j in perm)
    sample = list(range(0, n, 97))
    sp = b"".join(proofs[256 * j:256 * j + 256] for j in sample); si = b"".join(inputs[64 * j:64 * j + 64] for j in sample)
    assert O.groth16_verify_many(sp, 256, vk, si, 2, len(sample), O.MODE_REFERENCE) == bytes(st[j] for j in sample)
    pvk.close()


def _oracle_sample(O, vk, proofs, inputs, n_public, idx, mode=None):
    """The oracle's verdicts on the proofs idx of a batch (strided samples of the big runs)."""
    sz = 32 * n_public
    sp = b"".join(proofs[256 * j:256 * j + 256] for j in idx); si = b"".join(inputs[sz * j:sz * j + sz] for j in idx)
    return O.groth16_verify_many(sp, 256, vk, si, n_public, len(idx), O.MODE_REFERENCE if mode is None else mode)


def _mixed_sample(n, valid, per_class, lo=0, invalid_every=16):
    """Indices in [lo, n): `valid` strided ones plus `per_class` proofs of each of the generator's five failure classes (proof i is invalid
    when i % invalid_every == invalid_every - 1, its class (i // invalid_every) % 5), spread over the range."""
    idx = set(range(lo, n, max(1, (n - lo) // valid)))
    q_lo, q_hi = (lo + invalid_every - 1) // invalid_every, n // invalid_every
    for cls in range(5):
        qs = [q for q in range(q_lo, q_hi) if q % 5 == cls]
        step = max(1, len(qs) // per_class)
        idx.update(invalid_every * q + invalid_every - 1 for q in qs[::step][:per_class])
    return sorted(i for i in idx if lo <= i < n)


def test_batch_65536_expected_statuses(pkg, O, L):
    n = 1 << 16
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540003, 2, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    st = pvk.verify_batch(proofs, inputs)
    assert st == exp
    assert st.count(bytes([1])) == n - n // 16
    # not only the generator's word for it: the oracle on ~256 proofs spread over the batch, every failure class among them
    idx = _mixed_sample(n, 200, 11) + [n - 1]
    assert _oracle_sample(O, vk, proofs, inputs, 2, idx) == bytes(st[j] for j in idx)
    assert len(set(st[j] for j in idx)) == 5
    pvk.close()


def test_full_size_batch_two_chunks(pkg, O, L):
    """BASELINE's full batch size and beyond: 2^20 + 777 proofs run as two workspace chunks (and sub-batch streams); the status
    vector must be the generator's (every 16th proof invalid, cycling through the five failure classes)."""
    n = (1 << 20) + 777
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540077, 2, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    st = pvk.verify_batch(proofs, inputs)
    assert len(st) == n and st == exp
    assert st.count(bytes([pkg.ACCEPT])) == n - n // 16
    # the oracle on a strided 256-proof sample of the full-size batch, the second chunk's 777 proofs included
    idx = _mixed_sample(1 << 20, 190, 10) + _mixed_sample(n, 6, 2, lo=1 << 20)
    assert 240 <= len(idx) <= 280 and _oracle_sample(O, vk, proofs, inputs, 2, idx) == bytes(st[j] for j in idx)
    assert len(set(st[j] for j in idx)) == 5 and len(set(st[j] for j in idx if j >= 1 << 20)) == 5
    pvk.close()


def test_config5_full_size_4096_x_1024_inputs(pkg, O, L):
    """BASELINE configs[4] at its full size: 4096 proofs x 1024 public inputs (262 144 MSM lanes, 168 MB of comb digits) against the generator's
    statuses, the oracle on 24 strided proofs."""
    n, n_public = 4096, 1024
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540051, n_public, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    st = pvk.verify_batch(proofs, inputs)
    assert st == exp and set(exp) == {0, 1, 2, 3, 4}
    idx = _mixed_sample(n, 14, 2)
    assert 20 <= len(idx) <= 28 and _oracle_sample(O, vk, proofs, inputs, n_public, idx) == bytes(st[j] for j in idx)
    assert len(set(st[j] for j in idx)) == 5
    pvk.close()


def test_wide_msm_slicing_70000_x_40_inputs(pkg, O, L):
    """More proofs than one wide-MSM launch holds (G16_WIDE_MSM_MAX_PROOFS = 65536): the batch runs in slices that reuse the partial-sum buffer."""
    n, n_public = 70000, 40
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540052, n_public, n, invalid_every=16, agree=True, threads=16)
    pvk = pkg.PreparedVk(vk)
    st = pvk.verify_batch(proofs, inputs)
    assert st == exp
    idx = sorted(set(_mixed_sample(n, 14, 2) + [65535, 65536, 65551, n - 1]))      # both sides of the slice boundary
    assert _oracle_sample(O, vk, proofs, inputs, n_public, idx) == bytes(st[j] for j in idx)
    pvk.close()


def test_more_public_inputs(pkg, O, L):
    """nPublic = 5: a different key shape through the same kernels (window tables per input)."""
    n = 40
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540004, 5, n, invalid_every=4, agree=True, threads=8)
    pvk = pkg.PreparedVk(vk)
    st = pvk.verify_batch(proofs, inputs)
    assert st == exp == O.groth16_verify_many(proofs, 256, vk, inputs, 5, n, O.MODE_REFERENCE)
    pvk.close()


def test_plonk_device_transcripts_vs_oracle_stage_goldens(pkg, O, fixtures, L):
    """The Fiat-Shamir chain as the DEVICE computes it (bn254_k_plonk.hip: SHA-256, the key-side prefix, gamma -> beta -> alpha -> zeta) against the oracle's
    stage digests -- for the reference's four fixtures these are the goldens of SURVEY.md Appendix B.3 -- and the stage-1 verdicts (opening check,
    loader errors) of mutated proofs."""
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    rng = random.Random(33)
    cases, vk = _plonk_cases(O, fixtures, rng, 12)
    n = len(cases)
    pvk = pkg.PreparedPlonkVk(vk)
    L.bn254_dbg_plonk_stage1.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    zeta = (C.c_uint8 * (32 * n))(); st = (C.c_uint8 * n)()
    assert L.bn254_dbg_plonk_stage1(pvk._h, b"".join(c[0] for c in cases), 904, b"".join(c[1] for c in cases), 2, n, zeta, st, 0) == 0
    zeta = bytes(zeta); st = bytes(st)
    alive = 0
    for i, (p, q) in enumerate(cases):
        ins = [int.from_bytes(q[:32], "big"), int.from_bytes(q[32:], "big")]
        ost, dig = O.plonk_stage_digests(p, vk, ins)
        final = O.plonk_verify(p, vk, ins)
        if st[i] == pkg.ACCEPT:                      # alive after stage 1: the oracle must not have failed it before the KZG stage
            assert final in (pkg.ACCEPT, 8), (i, final)
            alive += 1
        else:
            assert st[i] == final, (i, st[i], final)
        if st[i] in (pkg.ACCEPT, 7) and ost == O.ACCEPT:      # the challenges exist: zeta = digest mod r
            assert int.from_bytes(zeta[32 * i:32 * i + 32], "big") == int.from_bytes(dig["zeta"], "big") % R, i
    assert alive >= 4
    pvk.close()
