"""Arithmetic checks of the oracle that do not depend on the reference's fixtures: Python-integer cross-checks, algebraic
identities (EIP-197 style), and a structurally independent pairing (polynomial-basis Fp12, textbook Miller loop over the
bits of 6u+2, plain exponentiation) compared value-for-value with the oracle's tower/NAF/line-table implementation."""
import random

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
U = 4965661367192848881


def be(v):
    return int(v).to_bytes(32, "big")


def test_field_ops(O):
    random.seed(1)
    vals = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2] + [random.randrange(P) for _ in range(200)]
    for i, a in enumerate(vals):
        b = vals[(7 * i + 3) % len(vals)]
        assert O.fp_op(0, a, b) == (a + b) % P and O.fp_op(1, a, b) == (a - b) % P and O.fp_op(2, a, b) == a * b % P
        assert O.fp_op(5, a) == (-a) % P
        assert O.fp_op(3, a) == (pow(a, -1, P) if a else 0)
        s = O.fp_op(4, a * a % P)
        assert s in (a, (-a) % P)
    for _ in range(100):
        a, b = random.randrange(R), random.randrange(R)
        assert O.fp_op(2, a, b, 1) == a * b % R and O.fp_op(3, a, 0, 1) == pow(a, -1, R)
    # values >= p reduce (probe API reduces; the codecs reject)
    assert O.fp_op(0, P + 5, 0) == 5


def test_fp2(O):
    random.seed(2)
    for _ in range(50):
        a = (random.randrange(P), random.randrange(P)); b = (random.randrange(P), random.randrange(P))
        assert O.fp2_op(2, a, b) == ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
        sq = O.fp2_op(5, a)
        assert sq == O.fp2_op(2, a, a)
        assert O.fp2_op(4, sq) in (a, ((-a[0]) % P, (-a[1]) % P))
        assert O.fp2_op(2, a, O.fp2_op(3, a)) == (1, 0)
    # non-residue: sqrt reports failure as (0,0).  xi = 9+i is a non-square in Fp2
    assert O.fp2_op(4, (9, 1)) == (0, 0)


def test_group_laws_and_bilinearity(O):
    random.seed(3)
    g1, g2 = O.g1_gen(), O.g2_gen()
    a, b = random.randrange(1, R), random.randrange(1, R)
    assert O.g1_add(O.g1_mul(g1, a), O.g1_mul(g1, b)) == O.g1_mul(g1, (a + b) % R)
    assert O.g2_add(O.g2_mul(g2, a), O.g2_mul(g2, b)) == O.g2_mul(g2, (a + b) % R)
    assert O.g1_mul(g1, R) == bytes(64) and O.g2_mul(g2, R) == bytes(128)
    # scalars are used as raw 256-bit integers (bn::Fr::from_slice does not reduce): x and x + r act alike
    assert O.g1_mul(g1, a + R) == O.g1_mul(g1, a)
    e = O.pairing(g1, g2)
    assert O.pairing(O.g1_mul(g1, a), O.g2_mul(g2, b)) == O.pairing(O.g1_mul(g1, a * b % R), g2)
    assert O.pairing(O.g1_mul(g1, a), O.g2_mul(g2, b)) != e
    neg = g1[:32] + be(P - int.from_bytes(g1[32:], "big"))
    one = O.pairing(g1 + neg, g2 + g2)
    assert one == be(1) + bytes(352)
    # pairs with an identity operand are skipped (bn::pairing_batch)
    assert O.pairing(g1 + bytes(64), g2 + g2) == e


def test_final_exp_chain_is_power_of_plain(O):
    """The exp_by_neg_z chain computes the plain (p^12-1)/r power raised to a constant c coprime to r."""
    z = U
    A = -z; B = 2 * A; C = 2 * B; D = C + B; E = -z * D; F = 2 * E; G = -z * F; H = -D; I = -G; J = I + E; K = J + H
    L = K + B; M = K + E; N = M + 1; Oe = P * L; Pe = Oe + N; Q = P * P * K; Rr = Q + Pe; S = -1; T = S + L; Ue = P ** 3 * T
    V = Ue + Rr
    phi = P ** 4 - P ** 2 + 1
    hard = phi // R
    assert phi % R == 0 and (V % phi) % hard == 0
    c = (V % phi) // hard
    from math import gcd
    assert gcd(c, R) == 1
    f = O.miller_loop(O.g1_gen(), O.g2_gen())
    chain, plain = O.final_exp(f), O.final_exp(f, plain=True)
    acc = be(1) + bytes(352)
    for bit in bin(c)[2:]:
        acc = O.fp12_op(1, acc)
        if bit == "1":
            acc = O.fp12_op(0, acc, plain)
    assert acc == chain


# ---------------------------------------------------------------- independent pairing (py_ecc style): tests/pyref_groth16.py
from pyref_groth16 import slow_pairing_plain  # noqa: E402  polynomial-basis Fp12, Miller loop over the bits of 6u+2, plain pow((p^12-1)/r)


def tower_bytes_to_poly(b):
    """Oracle Fp12 bytes (tower order c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2; each (re, im)) -> coefficients of w^0..w^11."""
    c = [(int.from_bytes(b[64 * i:64 * i + 32], "big"), int.from_bytes(b[64 * i + 32:64 * i + 64], "big")) for i in range(6)]
    wpow = [0, 2, 4, 1, 3, 5]  # c0.c0 -> w^0, c0.c1 -> w^2 (v), c0.c2 -> w^4, c1.c0 -> w, c1.c1 -> w^3, c1.c2 -> w^5
    poly = [0] * 12
    for (re, im), k in zip(c, wpow):
        poly[k] = (poly[k] + re - 9 * im) % P
        poly[k + 6] = (poly[k + 6] + im) % P
    return poly


def test_independent_pairing_value(O):
    random.seed(4)
    g1, g2 = O.g1_gen(), O.g2_gen()
    a, b = random.randrange(1, R), random.randrange(1, R)
    pb, qb = O.g1_mul(g1, a), O.g2_mul(g2, b)
    p = (int.from_bytes(pb[:32], "big"), int.from_bytes(pb[32:], "big"))
    q = ((int.from_bytes(qb[32:64], "big"), int.from_bytes(qb[0:32], "big")), (int.from_bytes(qb[96:128], "big"), int.from_bytes(qb[64:96], "big")))
    slow = slow_pairing_plain(p, q)
    fast = tower_bytes_to_poly(O.final_exp(O.miller_loop(pb, qb), plain=True))
    assert slow == fast


def test_ate_relation_is_a_subgroup_test():
    """Number theory behind bn254_vm.h::vm_g2_ate_check.  On the twist E'(Fp2), of order r * h2, the endomorphism psi satisfies
    m(X) = X^2 - t X + p.  A relation chi(psi) Q = O that holds on G2 (chi(p) = 0 mod r) characterises G2 iff chi(psi) is
    injective on the h2-torsion, which holds when Res(chi, m) is coprime to h2.  Checked for the optimal-ate relation
    chi(X) = X^3 - X^2 + X + (6u+2) used by the product and, as a control, for gnark's (u+1) + u X + u X^2 - 2u X^3."""
    from math import gcd
    u = 4965661367192848881
    p = 36 * u**4 + 36 * u**3 + 24 * u**2 + 6 * u + 1
    r = 36 * u**4 + 36 * u**3 + 18 * u**2 + 6 * u + 1
    t = 6 * u * u + 1
    assert p + 1 - t == r
    h2 = p - 1 + t
    assert h2 == 21888242871839275222246405745257275088844257914179612981679871602714643921549  # the G2 cofactor of alt_bn128
    assert gcd(r, h2) == 1

    def res_with_m(c3, c2, c1, c0):
        # chi mod m = a X + b using X^2 = tX - p, X^3 = (t^2 - p) X - t p; Res(chi, m) = prod over the roots x of m of (a x + b)
        a = c3 * (t * t - p) + c2 * t + c1
        b = -c3 * t * p - c2 * p + c0
        return a * a * p + a * b * t + b * b

    assert (6 * u + 2 + p - p * p + p**3) % r == 0
    res = res_with_m(1, -1, 1, 6 * u + 2)
    assert res % r == 0 and gcd(res, h2) == 1
    assert ((u + 1) + u * p + u * p * p - 2 * u * p**3) % r == 0
    res_gnark = res_with_m(-2 * u, u, u, u + 1)
    assert res_gnark % r == 0 and gcd(res_gnark, h2) == 1
