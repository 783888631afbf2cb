"""Degenerate and mutated verifying keys: the status byte of every (key, proof) pair must be the oracle's, in the reference's order -- proof loader (lib.rs:45),
key loader (lib.rs:46), PrepareInputsFailed (groth16/verify.rs:54-56), the equation.

The CPU half runs what needs no GPU: a single-proof call against key bytes that do not load answers from the host (nothing can be launched without a key).  The GPU
half (-m gpu) runs the same comparison for keys that do load, plus the key shapes the other GPU tests never use: no K points at all (nK = 0), no public inputs
(nK = 1), key points that are on the twist but outside the r-torsion (the key loader is "unchecked", converter.rs:113-133)."""
import ctypes as C
import os
import random
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
U = 4965661367192848881
E_NO_DEVICE, E_HIP = -2, -3


def be(v):
    return int(v).to_bytes(32, "big")


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def _single(pkg, proof, vk, inputs, mode=0):
    """(return code, status byte) of bn254_groth16_verify"""
    st = C.c_uint8(0xEE)
    ib = b"".join(be(x) for x in inputs)
    rc = pkg.lib().bn254_groth16_verify(bytes(proof), len(proof), bytes(vk), len(vk), ib, len(inputs), mode, C.byref(st))
    return rc, st.value


def _key_without_k(vk1):
    """a key for one K point (no public inputs) rewritten to hold none: u32 nK = 0, the K point removed"""
    return vk1[:288] + (0).to_bytes(4, "big") + vk1[292 + 32:]


def _mutations(vk, rng, count):
    """(label, key bytes): byte flips over the whole key, the compressed-point flags, the K count, truncations"""
    out = []
    for k in range(count):
        b = bytearray(vk)
        kind = k % 8
        if kind in (0, 1, 2):                               # one random byte of a random field
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif kind == 3:                                     # the two flag bits of one compressed point
            off = rng.choice([0, 32, 64, 128, 192, 224, 292, 292 + 32 * rng.randrange(3)])
            b[off] = (b[off] & 0x3F) | (rng.randrange(4) << 6)
        elif kind == 4:                                     # the K count: 0, one less, one more, far too many
            b[288:292] = rng.choice([0, 2, 4, 5, 1000, 2 ** 32 - 1]).to_bytes(4, "big")
        elif kind == 5:                                     # truncation
            b = b[:rng.randrange(0, len(b))]
        elif kind == 6:                                     # an x coordinate replaced by a random one (half of them have no square root)
            off = rng.choice([0, 32, 192, 292, 324, 356])
            b[off:off + 32] = be(rng.randrange(P)); b[off] = (b[off] & 0x3F) | rng.choice([0x80, 0xC0])
        else:                                               # a G2 x coordinate replaced: on the twist but (almost surely) outside the r-torsion, or no root
            off = rng.choice([64, 128, 224])
            b[off:off + 64] = be(rng.randrange(P)) + be(rng.randrange(P)); b[off] = (b[off] & 0x3F) | rng.choice([0x80, 0xC0])
        out.append(("m%d/%d" % (k, kind), bytes(b)))
    return out


def _proofs_three(pkg):
    """a valid proof, the same with A off the curve, the same with C.x >= p -- and the key + inputs they belong to"""
    vk, proofs, inputs, exp = pkg.synth_groth16(0xED6E, 2, 1, invalid_every=0, agree=True, threads=1)
    assert exp == bytes([pkg.ACCEPT])
    good = proofs[:256]
    off = bytearray(good); off[63] ^= 1
    big = bytearray(good); big[192:224] = b"\xff" * 32
    xs = [int.from_bytes(inputs[32 * k:32 * k + 32], "big") for k in range(2)]
    return vk, [good, bytes(off), bytes(big)], xs


def _compare_mutated_keys(pkg, O, count, seed):
    gpu = _have_gpu()
    vk, proofs, xs = _proofs_three(pkg)
    rng = random.Random(seed)
    seen = {"host": 0, "device": 0}
    statuses = set()
    for label, key in _mutations(vk, rng, count) + [("no-K", _key_without_k(pkg.synth_groth16(3, 0, 0, invalid_every=0, agree=True, threads=1)[0]))]:
        key_loads = O.groth16_verify(proofs[0], key, xs) != O.ERR_MALFORMED      # a valid proof: MALFORMED can only be the key's
        for pi, pr in enumerate(proofs):
            want = O.groth16_verify(pr, key, xs)
            rc, got = _single(pkg, pr, key, xs)
            if key_loads and not gpu:
                assert rc in (E_NO_DEVICE, E_HIP), (label, pi, rc, got)               # the key prepared: the call needs the GPU, and says so
                seen["device"] += 1
                continue
            assert rc == 0 and got == want, (label, pi, rc, got, want)
            seen["host" if not key_loads else "device"] += 1
            statuses.add(got)
    return seen, statuses



def _plonk_key_mutations(vk, rng, count):
    """(label, key bytes) for the PlonK key (plonk/converter.rs:24-119): truncations, bit flips in the eight compressed points, in the scalars in front of them, in the
    Qcp count / KZG points behind them and in the trailing index list, flag bits, coordinates >= p"""
    out = []
    n = len(vk)
    for k in range(count):
        b = bytearray(vk)
        kind = k % 8
        if kind == 0:
            b = b[:rng.randrange(0, n)]
        elif kind == 1:
            b[rng.randrange(112, 372)] ^= 1 << rng.randrange(8)
        elif kind == 2:
            j = rng.randrange(8); b[112 + 32 * j:144 + 32 * j] = b"\xff" * 32
        elif kind == 3:
            j = rng.randrange(8); b[112 + 32 * j] = (b[112 + 32 * j] & 0x3F) | (rng.randrange(4) << 6)
        elif kind == 4:
            b[rng.randrange(368, 372 + 64 + 160)] ^= 1 << rng.randrange(8)
        elif kind == 5:
            b[rng.randrange(0, 112)] ^= 1 << rng.randrange(8)
        elif kind == 6:
            b[rng.randrange(n - 64, n)] ^= 1 << rng.randrange(8)
        else:
            b[rng.randrange(372, 372 + 64 + 160)] = rng.randrange(256)
        out.append(("p%d/%d" % (k, kind), bytes(b)))
    return out


def _compare_mutated_plonk_keys(pkg, O, fixtures, count, seed):
    """Every (mutated key, fixture proof): the product's status byte == the oracle's.  Keys that do not load answer on the host; keys that load need the GPU (-m gpu)."""
    gpu = _have_gpu()
    fx, vk = fixtures
    base = [(bytes.fromhex(f["raw_proof"]), [int(x) for x in f["public_inputs"]]) for f in fx.values() if f["variant"] == "plonk"]
    off_curve = bytearray(base[0][0]); off_curve[63] ^= 1
    proofs = [base[0], base[1], (bytes(off_curve), base[0][1])]
    rng = random.Random(seed)
    seen = {"host": 0, "device": 0}
    statuses = set()
    L = pkg.lib()
    for label, key in _plonk_key_mutations(vk, rng, count):
        h = C.c_void_p()
        loads = L.bn254_plonk_vk_prepare(key, len(key), C.byref(h)) == 0
        if loads:
            L.bn254_plonk_vk_free(h)
        for pi, (pr, xs) in enumerate(proofs):
            want = O.plonk_verify(pr, key, xs)
            st = C.c_uint8(0xEE)
            ib = b"".join(be(x) for x in xs)
            rc = L.bn254_plonk_verify(pr, len(pr), key, len(key), ib, len(xs), C.byref(st))
            if loads and not gpu:
                assert rc in (E_NO_DEVICE, E_HIP), (label, pi, rc, st.value)        # the key prepared: the call needs the GPU, and says so
                seen["device"] += 1
                continue
            assert rc == 0 and st.value == want, (label, pi, rc, st.value, want)
            seen["device" if loads else "host"] += 1
            statuses.add(st.value)
    return seen, statuses


def test_mutated_keys_single_proof_host_side(pkg, O):
    """200 mutated keys x {valid, off-curve, >= p} proofs through bn254_groth16_verify and the oracle.  Keys that no longer load are answered on the host: the
    proof's loader error (3, 2) if it has one -- lib.rs:45 runs before :46 -- else MALFORMED.  Keys that still load need the GPU (checked by the -m gpu twin)."""
    seen, statuses = _compare_mutated_keys(pkg, O, 200, 11)
    assert seen["host"] >= 150 and seen["device"] >= 60, seen
    if not _have_gpu():
        assert statuses == {pkg.ERR_MALFORMED, pkg.ERR_NOT_ON_CURVE, pkg.ERR_NOT_MEMBER}


def test_unparsable_key_keeps_the_proof_loaders_order(pkg, O):
    """key that does not load x every loader error a proof can have, B's r-torsion included (the product answered 6 for all of them)"""
    vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540001, 2, 64, invalid_every=2, agree=True, threads=4)
    bad_key = bytearray(vk); bad_key[0] &= 0x3F                                     # flag 0b00 on alpha (constants.rs:24)
    seen = set()
    for i in range(64):
        pr = proofs[256 * i:256 * i + 256]
        xs = [int.from_bytes(inputs[64 * i + 32 * k:64 * i + 32 * k + 32], "big") for k in range(2)]
        want = O.groth16_verify(pr, bytes(bad_key), xs)
        assert want == (exp[i] if exp[i] in (2, 3, 4) else pkg.ERR_MALFORMED)
        assert _single(pkg, pr, bad_key, xs) == (0, want), i
        seen.add(want)
    assert seen == {2, 3, 4, 6}
    assert _single(pkg, proofs[:255], bad_key, [1, 2]) == (0, pkg.ERR_MALFORMED)   # short proof: the slice panic comes first of all


def test_unparsable_plonk_key_keeps_the_proof_loaders_order(pkg, O, fixtures):
    fx, vk = fixtures
    f = next(v for v in fx.values() if v["variant"] == "plonk")
    proof = bytes.fromhex(f["raw_proof"]); xs = [int(x) for x in f["public_inputs"]]
    bad_key = vk[:300]
    cases = [proof, proof[:500]]
    p = bytearray(proof); p[0:32] = b"\xff" * 32; cases.append(bytes(p))            # L.x >= p
    p = bytearray(proof); p[63] ^= 1; cases.append(bytes(p))                        # L off the curve
    p = bytearray(proof); p[64 * 7 + 63] ^= 1; cases.append(bytes(p))               # the last of the eight leading points off the curve
    p = bytearray(proof); p[512:516] = (99).to_bytes(4, "big"); cases.append(bytes(p))   # claimed-value count beyond the layout
    got = []
    for c in cases:
        st = C.c_uint8(0xEE)
        ib = b"".join(be(x) for x in xs)
        assert pkg.lib().bn254_plonk_verify(c, len(c), bad_key, len(bad_key), ib, len(xs), C.byref(st)) == 0
        assert st.value == O.plonk_verify(c, bad_key, xs), len(got)
        got.append(st.value)
    assert got == [6, 6, 2, 3, 3, 6]


def test_mutated_plonk_keys_host_side(pkg, O, fixtures):
    """400 mutated PlonK keys x {two valid proofs, one with a point off the curve}: a key that does not load answers on the host with the oracle's byte -- lib.rs:70
    before :71: the proof loader's error if it has one, else the key's."""
    seen, statuses = _compare_mutated_plonk_keys(pkg, O, fixtures, 400, 0x91A)
    assert seen["host"] >= 300
    assert {3, 6} <= statuses


def test_key_without_k_points_prepares(pkg):
    """nK = 0: the reference's loader succeeds and prepare_inputs answers Err(PrepareInputsFailed) for every input count (groth16/verify.rs:54-56: len + 1 != 0).
    Round 4 refused such a key (BN254_E_VK -> status 6, a panic in the Rust wrapper)."""
    vk1 = pkg.synth_groth16(3, 0, 0, invalid_every=0, agree=True, threads=1)[0]
    p1 = pkg.PreparedVk(vk1)
    assert p1.n_public == 0
    p1.close()
    p0 = pkg.PreparedVk(_key_without_k(vk1))
    assert p0.n_public == 2 ** 64 - 1                                               # no input count matches such a key
    p0.close()


def test_line_tables_exist_for_every_twist_point():
    """Why bn254_groth16_vk_prepare cannot fail on a key that parsed.  A key's G2 elements are on the twist by construction (y is computed from x) but unchecked
    otherwise, and the product walks them with AFFINE line tables, which have no value where the walk meets T = +-S or T = O.  That needs (a) in the main loop: the
    order of Q divides k - d, k + d or 2k for a prefix k of NAF(6u+2) and its next digit d -- the order divides #E'(Fp2) = r (2p - r), all of whose divisors are
    enumerated here -- or (b) in the two Frobenius steps: psi(Q) = +-[6u+2]Q resp. [6u+2]Q + psi(Q) = +-psi^2(Q), which on a prime-order component means an
    eigenvalue of psi (a root of X^2 - tX + p) equal to +-(6u+2) resp. a root of X^2 -+ X -+ (6u+2).  Neither happens, so the tables exist for every point
    except the identity, and the product computes on such points by the same group law as the reference's projective formulas."""
    from itertools import combinations
    hdr = open(os.path.join(ROOT, "snark-bn254-verifier_amd", "csrc", "bn254_constants.h")).read()
    naf = [int(x) for x in re.search(r"BN_ATE_NAF\[66\] = \{([^}]*)\}", hdr).group(1).split(",")]
    p = 36 * U ** 4 + 36 * U ** 3 + 24 * U ** 2 + 6 * U + 1
    r = 36 * U ** 4 + 36 * U ** 3 + 18 * U ** 2 + 6 * U + 1
    t = 6 * U * U + 1
    assert p == P and r == R
    primes = [10069, 5864401, 1875725156269, 197620364512881247228717050342013327560683201906968909, r]
    m = 1
    for q in primes:
        m *= q
    assert m == r * (2 * p - r)                                                     # #E'(Fp2), squarefree: the group is cyclic, every prime component 1-dimensional
    vals, k = [], 1
    for d in naf[1:]:
        k *= 2
        vals.append(k)                                                              # T = O after the doubling
        if d:
            vals += [k - d, k + d]                                                  # T = S, T = -S at the addition
            k += d
    assert k == 6 * U + 2
    divisors = []
    for n in range(1, len(primes) + 1):
        for c in combinations(primes, n):
            v = 1
            for x in c:
                v *= x
            divisors.append(v)
    assert not [(dv, v) for dv in divisors for v in vals if v % dv == 0]

    def roots(a, b, c, q):                                                         # roots of a X^2 + b X + c mod the prime q
        disc = (b * b - 4 * a * c) % q
        if disc and pow(disc, (q - 1) // 2, q) != 1:
            return []
        s = _sqrt_mod(disc, q)
        i2a = pow(2 * a, -1, q)
        return [(-b + s) * i2a % q, (-b - s) * i2a % q]

    w = 6 * U + 2
    for q in primes:
        for lam in roots(1, -t, p, q):                                              # either eigenvalue of psi (only one of them acts on E'(Fp2)[q])
            assert (lam - w) % q and (lam + w) % q                                  # psi(Q) != +-[6u+2] Q
            assert (w + lam - lam * lam) % q and (w + lam + lam * lam) % q          # [6u+2]Q + psi(Q) != +-psi^2(Q)
            assert (w + lam) % q                                                    # ... and is not the identity


def _sqrt_mod(a, q):
    a %= q
    if a == 0:
        return 0
    if q % 4 == 3:
        return pow(a, (q + 1) // 4, q)
    s, qq = 0, q - 1
    while qq % 2 == 0:
        qq //= 2; s += 1
    z = 2
    while pow(z, (q - 1) // 2, q) != q - 1:
        z += 1
    mm, c, tt, rr = s, pow(z, qq, q), pow(a, qq, q), pow(a, (qq + 1) // 2, q)
    while tt != 1:
        i, x = 0, tt
        while x != 1:
            x = x * x % q; i += 1
        b = pow(c, 1 << (mm - i - 1), q)
        mm, c = i, b * b % q
        tt, rr = tt * c % q, rr * b % q
    return rr


# ------------------------------------------------------------------------------------------------------------------ GPU half
def _twist_point(O, rng):
    bt = O.fp2_op(2, O.fp2_op(3, (9, 1)), (3, 0))
    while True:
        x = (rng.randrange(P), rng.randrange(P))
        rhs = O.fp2_op(0, O.fp2_op(2, O.fp2_op(5, x), x), bt)
        y = O.fp2_op(4, rhs)
        if y != (0, 0) and O.fp2_op(5, y) == rhs:
            return be(x[1]) + be(x[0]) + be(y[1]) + be(y[0])


@pytest.mark.gpu
def test_mutated_keys_single_proof_on_device(pkg, O):
    """the twin of test_mutated_keys_single_proof_host_side with a GPU: every (key, proof) pair compared, keys that still load included (their G2 elements may have
    left the r-torsion, the K count may no longer match: REJECT, INPUT_LEN, the proof's loader errors)"""
    seen, statuses = _compare_mutated_keys(pkg, O, 120, 12)
    assert seen["device"] >= 40 and {pkg.REJECT, pkg.ERR_MALFORMED, pkg.ERR_NOT_ON_CURVE, pkg.ERR_NOT_MEMBER, pkg.ERR_INPUT_LEN} <= statuses, (seen, statuses)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [300, 40000])     # cooperative kernels / lane kernels
def test_key_without_public_inputs(pkg, O, n):
    """nK = 1: L = K[0], no scalar multiplication at all.  Exact path (both launch forms) and the RLC mode, against the generator's statuses and the oracle on a sample."""
    vk, proofs, inputs, exp = pkg.synth_groth16(0x2E80, 0, n, invalid_every=4, agree=True, threads=16)
    assert inputs == b"" and set(exp) == {pkg.ACCEPT, pkg.REJECT, pkg.ERR_NOT_ON_CURVE, pkg.ERR_NOT_IN_SUBGROUP, pkg.ERR_NOT_MEMBER}
    pvk = pkg.PreparedVk(vk)
    assert pvk.n_public == 0
    got = pvk.verify_batch(proofs, b"", n, n_public=0)
    assert got == exp
    m = 48
    assert got[:m] == O.groth16_verify_many(proofs[:256 * m], 256, vk, b"", 0, m, O.MODE_REFERENCE)
    pkg.set_rlc_params(min_batch=64)
    try:
        assert pvk.verify_batch(proofs, b"", n, n_public=0, flags=pkg.FLAG_RLC) == exp
    finally:
        pkg.set_rlc_params(min_batch=200000)
    # a caller that passes one input to this key: loader errors first, then PrepareInputsFailed
    k = 64
    got1 = pvk.verify_batch(proofs[:256 * k], bytes(32 * k), k, n_public=1)
    assert got1 == bytes(e if e in (2, 3, 4) else pkg.ERR_INPUT_LEN for e in exp[:k])
    assert got1[:16] == bytes(O.groth16_verify(proofs[256 * i:256 * i + 256], vk, [0]) for i in range(16))
    pvk.close()
    if n == 300:
        for i in range(6):
            assert pkg.Groth16Verifier.verify(proofs[256 * i:256 * i + 256], vk, []) == exp[i]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [200, 33000])
def test_key_without_k_points_on_device(pkg, O, n):
    """nK = 0: every proof is answered with its loader error, else BN254_ERR_INPUT_LEN -- for every input count, zero included"""
    vk1, proofs, _, exp = pkg.synth_groth16(0x2E81, 0, n, invalid_every=3, agree=True, threads=16)
    vk0 = _key_without_k(vk1)
    want = bytes(e if e in (2, 3, 4) else pkg.ERR_INPUT_LEN for e in exp)
    pvk = pkg.PreparedVk(vk0)
    for n_public in (0, 2):
        got = pvk.verify_batch(proofs, bytes(32 * n_public * n), n, n_public=n_public)
        assert got == want, n_public
        assert got[:24] == bytes(O.groth16_verify(proofs[256 * i:256 * i + 256], vk0, [0] * n_public) for i in range(24))
    pkg.set_rlc_params(min_batch=64)
    try:
        assert pvk.verify_batch(proofs, b"", n, n_public=0, flags=pkg.FLAG_RLC) == want      # the flag is not honoured for a key nothing matches: same bytes
    finally:
        pkg.set_rlc_params(min_batch=200000)
    pvk.close()
    for i in range(4):
        assert pkg.Groth16Verifier.verify(proofs[256 * i:256 * i + 256], vk0, [5]) == want[i]


@pytest.mark.gpu
def test_key_elements_outside_the_r_torsion(pkg, O):
    """gamma2 / delta2 / beta2 replaced by twist points outside G2 -- of full order, and of the SMALL orders the cofactor offers (10069, 5864401, their product):
    the key loader does not check them (converter.rs:113-133), the reference computes on, and so must the product -- same verdicts as the oracle, which follows
    `bn`'s projective formulas where the product reads affine line tables (test_line_tables_exist_for_every_twist_point: they always exist)."""
    rng = random.Random(77)
    vk, proofs, inputs, exp = pkg.synth_groth16(0x2E82, 2, 20, invalid_every=2, agree=True, threads=4)
    h = 2 * P - R
    g2 = O.g2_gen()
    pts = [_twist_point(O, rng)]
    for order in (10069, 5864401, 10069 * 5864401):
        while True:
            q = O.g2_mul(_twist_point(O, rng), R)                                      # kill the r-component (scalars are 256-bit: two steps)
            q = O.g2_mul(q, h // order) if q != bytes(128) else q
            if q != bytes(128):
                break
        assert O.g2_mul(q, order) == bytes(128) and O.g2_subgroup_check(q) == 0
        pts.append(q)
    pts.append(O.g2_add(O.g2_mul(g2, 12345), pts[1]))                                 # a G2 point plus a small-order component
    seen = set()
    for off in (64, 128, 224):
        for q in pts:
            key = bytearray(vk); key[off:off + 64] = O.compress_g2(q)
            for mode, omode in ((pkg.VK_REFERENCE, O.MODE_REFERENCE), (pkg.VK_GNARK, O.MODE_GNARK)):
                pvk = pkg.PreparedVk(bytes(key), mode)
                got = pvk.verify_batch(proofs, inputs, 20)
                pvk.close()
                assert got == O.groth16_verify_many(proofs, 256, bytes(key), inputs, 2, 20, omode), (off, mode)
                seen |= set(got)
    assert seen == {pkg.REJECT, pkg.ERR_NOT_ON_CURVE, pkg.ERR_NOT_IN_SUBGROUP, pkg.ERR_NOT_MEMBER}     # no proof verifies against a key with a foreign element; loader errors first


@pytest.mark.gpu
def test_mutated_plonk_keys_on_device(pkg, O, fixtures):
    """The same comparison on the GPU: the keys that still load (flipped bits in the selector commitments, the scalars, the KZG points, the index list) run the whole
    pipeline -- key tables built from the mutated points -- and must give the oracle's byte (OpeningPolyMismatch, PairingCheckFailed, InvalidWitness, ACCEPT ...)."""
    seen, statuses = _compare_mutated_plonk_keys(pkg, O, fixtures, 160, 0x91B)
    assert seen["device"] >= 100
    assert {1, 7, 8} <= statuses, statuses
