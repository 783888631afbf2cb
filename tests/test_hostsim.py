"""The product's device arithmetic (snark-bn254-verifier_amd/csrc/*.h), compiled for the HOST with the bound tracker on
(tests/hostsim), against the oracle.  This is how the exact algorithms of the HIP kernels are checked in a GPU-less
container; every call also asserts the value-/limb-bound assumptions of bn254_fp.h.  `inflate` shifts inputs by
multiples of p so that the lazy (unreduced) ranges are exercised, not only canonical values."""
import ctypes as C
import random

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def be(v):
    return int(v).to_bytes(32, "big")


def b2(x):
    return be(x[0]) + be(x[1])


def test_fp(hostsim):
    hs = hostsim

    def fp(op, a, b=0, inf=0):
        o = (C.c_uint8 * 32)(); hs.hs_fp_op(op, o, be(a), be(b), inf); return int.from_bytes(bytes(o), "big")

    random.seed(7)
    vals = [0, 1, 2, P - 1, P - 2, 1 << 253, (1 << 254) - 1, P // 2, 3, 9] + [random.randrange(P) for _ in range(150)]
    for inf in (0, 1, 3):
        for i, a in enumerate(vals):
            b = vals[(i * 7 + 3) % len(vals)]
            assert fp(0, a, b, inf) == (a + b) % P and fp(1, a, b, inf) == (a - b) % P
            assert fp(2, a, b, inf) == a * b % P and fp(6, a, 0, inf) == a * a % P
            assert fp(5, a, 0, inf) == (-a) % P and fp(7, a, 0, inf) == a % P
            assert fp(8, a, b, inf) == (9 * a - b) % P
            assert fp(9, a, 0, inf) == (1 if a % P == 0 else 0)
    for a in vals[:25]:
        assert fp(3, a) == fp(4, a) == (pow(a, -1, P) if a else 0)
    # the binary-GCD inversion (bn254_fp.h::fp_inv): every operand size its digit scan and its exact / approximated stand-ins distinguish
    # (below 2^58, three digits with a small top digit, long), lazily reduced inputs, values next to 0 and p
    inv_vals = [(1 << k) + d for k in (28, 29, 30, 57, 58, 59, 60, 61, 62, 63, 86, 87, 88, 116, 145, 232, 253) for d in (-1, 0, 1, 12345)]
    inv_vals += [random.getrandbits(k) for k in range(1, 255) for _ in range(4)] + [P - 1 - random.getrandbits(k) for k in range(1, 200, 3)]
    inv_vals += [random.randrange(P) for _ in range(1500)]
    for i, a in enumerate(inv_vals):
        a %= P
        for inf in ((0, 1, 3) if i < 300 else (0,)):
            assert fp(3, a, 0, inf) == (pow(a, -1, P) if a else 0), hex(a)


def test_fp2_fp12(hostsim, O):
    hs = hostsim

    def fp2(op, a, b=(0, 0), inf=0):
        o = (C.c_uint8 * 64)(); hs.hs_fp2_op(op, o, b2(a), b2(b), inf); r = bytes(o)
        return int.from_bytes(r[:32], "big"), int.from_bytes(r[32:], "big")

    def fp12(op, a, b=None, inf=0):
        o = (C.c_uint8 * 384)(); hs.hs_fp12_op(op, o, a, b if b is not None else bytes(384), inf); return bytes(o)

    random.seed(8)
    rnd2 = lambda: (random.randrange(P), random.randrange(P))
    for inf in (0, 2):
        for _ in range(40):
            a, b = rnd2(), rnd2()
            for op in (0, 1, 2):
                assert fp2(op, a, b, inf) == O.fp2_op(op, a, b)
            assert fp2(5, a, b, inf) == O.fp2_op(5, a)
            assert fp2(6, a, b, inf) == O.fp2_op(2, a, (9, 1))
            if inf == 0:  # the Karatsuba form of the Fp2 dot product (operand sums need normalised digits)
                ab, aa = O.fp2_op(2, a, b), O.fp2_op(5, a)
                assert fp2(10, a, b) == ab and fp2(11, a, b) == aa
                assert fp2(12, a, b) == O.fp2_op(0, ab, aa)
                m2 = O.fp2_op(1, (0, 0), O.fp2_op(0, ab, ab))
                assert fp2(13, a, b) == O.fp2_op(0, m2, O.fp2_op(2, a, (b[0], 0)))
    for _ in range(5):
        a = rnd2(); assert fp2(3, a) == O.fp2_op(3, a)
    r12 = lambda: b"".join(be(random.randrange(P)) for _ in range(12))
    for inf in (0, 1, 2):
        for _ in range(8):
            a, b = r12(), r12()
            assert fp12(0, a, b, inf) == O.fp12_op(0, a, b)
            assert fp12(1, a, None, inf) == O.fp12_op(1, a)
            for op in (3, 4, 5, 7):
                assert fp12(op, a, None, inf) == O.fp12_op(op, a), op
            d0, d3, d4 = rnd2(), rnd2(), rnd2()
            dense = b2(d0) + bytes(128) + b2(d3) + b2(d4) + bytes(64)  # d0 + (d3 + d4 v) w
            assert fp12(8, a, b2(d0) + b2(d3) + b2(d4), inf) == O.fp12_op(0, a, dense)
            dense = b2((d0[0], 0)) + bytes(128) + b2(d3) + b2(d4) + bytes(64)
            assert fp12(9, a, b2((d0[0], 0)) + b2(d3) + b2(d4), inf) == O.fp12_op(0, a, dense)
    a = r12()
    assert fp12(2, a) == O.fp12_op(2, a)
    c = O.fp12_op(0, O.fp12_op(7, a), O.fp12_op(2, a)); c = O.fp12_op(0, O.fp12_op(4, c), c)  # easy part: cyclotomic subgroup
    for inf in (0, 1):
        assert fp12(6, c, None, inf) == O.fp12_op(1, c)


def _twist_point(O, rng):
    bt = O.fp2_op(2, O.fp2_op(3, (9, 1)), (3, 0))
    while True:
        x = (rng.randrange(P), rng.randrange(P))
        rhs = O.fp2_op(0, O.fp2_op(2, O.fp2_op(5, x), x), bt)
        y = O.fp2_op(4, rhs)
        if y != (0, 0) and O.fp2_op(5, y) == rhs:
            return be(x[1]) + be(x[0]) + be(y[1]) + be(y[0])


def test_g1_complete_addition(hostsim, O):
    hs = hostsim

    def g1op(op, p, q=bytes(64)):
        o = (C.c_uint8 * 64)(); hs.hs_g1_op(op, o, p, q); return bytes(o)

    rng = random.Random(11)
    g1 = O.g1_gen(); Z = bytes(64)
    pts = [O.g1_mul(g1, rng.randrange(1, R)) for _ in range(8)]
    for i, p in enumerate(pts):
        q = pts[(i + 1) % len(pts)]
        neg = p[:32] + be(P - int.from_bytes(p[32:], "big"))
        assert g1op(0, p, q) == O.g1_add(p, q) and g1op(2, p, q) == O.g1_add(p, q)
        assert g1op(1, p) == O.g1_add(p, p) and g1op(0, p, p) == O.g1_add(p, p)    # P = Q through the addition law
        assert g1op(0, p, neg) == Z and g1op(2, p, neg) == Z                      # P = -Q -> identity
        assert g1op(0, Z, q) == q and g1op(2, p, Z) == p and g1op(2, Z, Z) == Z   # identity operands
        assert g1op(3, p)[0] == 1
        assert g1op(3, p[:32] + be((int.from_bytes(p[32:], "big") + 1) % P))[0] == 0
    o = (C.c_uint8 * 64)(); hs.hs_g1_sum(o, b"".join(pts), len(pts))
    acc = pts[0]
    for p in pts[1:]:
        acc = O.g1_add(acc, p)
    assert bytes(o) == acc


def test_g2_subgroup_check(hostsim, O):
    """psi-based check == the reference's naive [r-1]Q + Q == O (oracle), on G2 points, random twist points, cofactor-cleared
    points and points of order dividing the cofactor."""
    hs = hostsim
    rng = random.Random(12)
    g2 = O.g2_gen()
    for _ in range(3):
        q = O.g2_mul(g2, rng.randrange(1, R))
        assert hs.hs_g2_on_curve(q) == 1 and hs.hs_g2_in_subgroup(q) == 1
    rejected = 0
    for _ in range(5):
        q = _twist_point(O, rng)
        assert hs.hs_g2_on_curve(q) == 1
        exp = O.g2_subgroup_check(q)
        assert hs.hs_g2_in_subgroup(q) == exp
        rejected += exp == 0
    assert rejected >= 4
    h2 = 2 * P - R
    tq = _twist_point(O, rng)
    q = O.g2_mul(tq, h2)
    assert O.g2_subgroup_check(q) == 1 and hs.hs_g2_in_subgroup(q) == 1
    q = O.g2_mul(tq, R)
    if q != bytes(128):
        assert O.g2_subgroup_check(q) == 0 and hs.hs_g2_in_subgroup(q) == 0


def test_g2_ate_relation_subgroup_check(hostsim, O):
    """The product's r-torsion test (final point of the Miller program against -psi^3(B), bn254_vm.h::vm_g2_ate_check) has the
    accept set of the reference's naive check (oracle): G2 points, random twist points, cofactor-cleared points, points whose
    order divides the cofactor (including small prime orders) and sums of a G2 point with a cofactor point."""
    hs = hostsim
    rng = random.Random(21)
    g1, g2 = O.g1_gen(), O.g2_gen()
    pa = O.g1_mul(g1, rng.randrange(1, R))
    for _ in range(3):
        q = O.g2_mul(g2, rng.randrange(1, R))
        assert O.g2_subgroup_check(q) == 1 and hs.hs_vm_g2_ate_check(pa, q) == 1
    for _ in range(4):
        q = _twist_point(O, rng)
        assert hs.hs_vm_g2_ate_check(pa, q) == O.g2_subgroup_check(q)
    h2 = 2 * P - R
    tq = _twist_point(O, rng)
    q = O.g2_mul(tq, h2)
    assert O.g2_subgroup_check(q) == 1 and hs.hs_vm_g2_ate_check(pa, q) == 1
    cof = O.g2_mul(tq, R)                      # order divides h2
    assert cof != bytes(128)
    assert O.g2_subgroup_check(cof) == 0 and hs.hs_vm_g2_ate_check(pa, cof) == 0
    for ell in (10069, 5864401):
        assert h2 % ell == 0
        small = O.g2_mul(cof, h2 // ell)       # order ell (or the identity)
        if small != bytes(128):
            assert O.g2_subgroup_check(small) == 0 and hs.hs_vm_g2_ate_check(pa, small) == 0
            mixed = O.g2_add(O.g2_mul(g2, 12345), small)
            assert O.g2_subgroup_check(mixed) == 0 and hs.hs_vm_g2_ate_check(pa, mixed) == 0


def test_pairing_matches_oracle_bytes(hostsim, O):
    hs = hostsim
    rng = random.Random(13)
    g1, g2 = O.g1_gen(), O.g2_gen()
    a, b, c, d = [rng.randrange(1, R) for _ in range(4)]
    pa, qb = O.g1_mul(g1, a), O.g2_mul(g2, b)
    o = (C.c_uint8 * 384)()
    assert hs.hs_miller(o, pa, qb, 0, None, None) == 1
    hs.hs_final_exp(o, bytes(o))
    assert bytes(o) == O.pairing(pa, qb)
    pf = O.g1_mul(g1, c) + O.g1_mul(g1, d); qf = O.g2_mul(g2, c + 5) + O.g2_mul(g2, d + 7)
    assert hs.hs_miller(o, pa, qb, 2, pf, qf) == 1
    hs.hs_final_exp(o, bytes(o))
    assert bytes(o) == O.pairing(pa + pf, qb + qf)


def test_reference_plonk_fixtures_through_product_arithmetic(hostsim, O, fixtures):
    """The reference's 4 PlonK fixtures pin the PRODUCT's arithmetic headers (compiled for the host): the operands of each
    proof's final KZG pairing check (derived by the oracle, plonk/kzg.rs:175-187) go through the VM Miller program and final
    exponentiation (hs_vm_pairing3 with the two key-side G2 points as the fixed pairs), and the result must be exactly 1."""
    hs = hostsim
    fx, vk = fixtures
    g1, g2 = O.g1_gen(), O.g2_gen()
    one = (1).to_bytes(32, "big") + bytes(352)
    n = 0
    for name, f in fx.items():
        if f["variant"] != "plonk":
            continue
        proof = bytes.fromhex(f["raw_proof"])
        pis = [int(x) for x in f["public_inputs"]]
        st, ps, qs = O.plonk_pairing_inputs(proof, vk, pis)
        assert st == O.ACCEPT
        o = (C.c_uint8 * 384)()
        # e(g1, g2) * e(P0, Q0) * e(P1, Q1) with (g1, g2) as the variable pair: the fixed-line tables are built from Q0, Q1
        assert hs.hs_vm_pairing3(o, g1, g2, ps[0], qs[0], ps[1], qs[1], 0) == 1
        assert bytes(o) == O.pairing(g1, g2), name
        # and directly: Miller loop of the two pairs as one variable + one fixed pair, then the final exponentiation
        assert hs.hs_miller(o, ps[0], qs[0], 1, ps[1], qs[1]) == 1
        hs.hs_final_exp(o, bytes(o))
        assert bytes(o) == one, name
        n += 1
    assert n == 4


def test_latency_mode_program_matches(hostsim, O):
    """The small-batch launch plan (three separate Miller chains multiplied at the end) under the bound tracker: same GT element
    as the oracle and as the fused per-step program."""
    hs = hostsim
    rng = random.Random(41)
    g1, g2 = O.g1_gen(), O.g2_gen()
    pa, pl, pc = (O.g1_mul(g1, rng.randrange(1, R)) for _ in range(3))
    qb, qg, qd = (O.g2_mul(g2, rng.randrange(1, R)) for _ in range(3))
    for l_inf in (0, 1):
        o1 = (C.c_uint8 * 384)(); o2 = (C.c_uint8 * 384)()
        assert hs.hs_vm_pairing3(o1, pa, qb, pl, qg, pc, qd, l_inf) == 1
        assert hs.hs_vm_pairing3_split(o2, pa, qb, pl, qg, pc, qd, l_inf) == 1
        assert bytes(o1) == bytes(o2)
        exp = O.pairing(pa + pc, qb + qd) if l_inf else O.pairing(pa + pl + pc, qb + qg + qd)
        assert bytes(o2) == exp
        # the run form of the Miller loop (k_miller_run: runs of doubling steps + the following addition step as one operation)
        for per_run in (0, 1, 7, 22):            # the whole loop as one operation; one, seven, twenty-two steps per operation
            o3 = (C.c_uint8 * 384)()
            assert hs.hs_vm_pairing3(o3, pa, qb, pl, qg, pc, qd, l_inf | 2 | (per_run << 2)) == 1
            assert bytes(o3) == exp, per_run


def test_two_fixed_pair_run_on_reference_plonk_fixtures(hostsim, O, fixtures):
    """The lane path of PlonK's KZG check (bn254_vm.h::vm_miller_run_fixed2: both pairs table-driven, the accumulator in flight through the whole loop, then the final
    exponentiation program) under the bound tracker: exactly 1 on the operands of the reference's four fixtures, the oracle's pairing on random points, identity
    flags, and the same value whatever the number of steps per operation."""
    hs = hostsim
    fx, vk = fixtures
    one = (1).to_bytes(32, "big") + bytes(352)
    o = (C.c_uint8 * 384)()
    n = 0
    for name, f in fx.items():
        if f["variant"] != "plonk":
            continue
        st, ps, qs = O.plonk_pairing_inputs(bytes.fromhex(f["raw_proof"]), vk, [int(x) for x in f["public_inputs"]])
        assert st == O.ACCEPT
        assert hs.hs_vm_pairing2_fixed(o, ps[0], qs[0], ps[1], qs[1], 0, 0) == 1 and bytes(o) == one, name
        n += 1
    assert n == 4
    rng = random.Random(43)
    g1, g2 = O.g1_gen(), O.g2_gen()
    p0, p1 = (O.g1_mul(g1, rng.randrange(1, R)) for _ in range(2))
    q0, q1 = (O.g2_mul(g2, rng.randrange(1, R)) for _ in range(2))
    for per_run in (0, 1, 11, 44):
        assert hs.hs_vm_pairing2_fixed(o, p0, q0, p1, q1, 0, per_run) == 1 and bytes(o) == O.pairing(p0 + p1, q0 + q1), per_run
    assert hs.hs_vm_pairing2_fixed(o, p0, q0, p1, q1, 1, 0) == 1 and bytes(o) == O.pairing(p1, q1)
    assert hs.hs_vm_pairing2_fixed(o, p0, q0, p1, q1, 2, 0) == 1 and bytes(o) == O.pairing(p0, q0)
    assert hs.hs_vm_pairing2_fixed(o, p0, q0, p1, q1, 3, 0) == 1 and bytes(o) == one
