"""TEST INFRASTRUCTURE: an independent, pure-Python restatement of the reference's Groth16 `verify()` at VERDICT level.

It shares no code and no representation with oracle/ (C, 4x64 Montgomery limbs, Fp2/Fp6/Fp12 tower, NAF Miller loop with line tables,
exp_by_neg_z chain) or with the product (29-bit balanced digits, sums of products): Python integers, affine curve arithmetic with modular
inverses, Fp12 as polynomials Fp[w]/(w^12 - 18 w^6 + 82), a textbook Miller loop over the BITS of 6u+2 with the points untwisted into
E(Fp12), and the final exponentiation as a plain pow(f, (p^12-1)/r).  About 3 s per pairing -- for known-answer fixtures
(tests/golden/make_groth16_verdicts.py) and a few live verdicts in the CPU suite, never on a hot path.

What it follows in the reference (paths relative to /root/reference/verifier/src):
  * Groth16Verifier::verify                 lib.rs:44-49          proof loader, then key loader, then verify_groth16
  * load_groth16_proof_from_bytes           groth16/converter.rs:14-26, converter.rs:78-88 (G1), converter.rs:135-153 (G2: x.c1 | x.c0 | y.c1 | y.c0)
  * load_groth16_verifying_key_from_bytes   groth16/converter.rs:28-89 (alpha1@0 beta1@32 beta2@64 gamma2@128 delta1@192 delta2@224, u32 nK@288, K..;
                                            beta1 and beta2 NEGATED on store, :74,79), converter.rs:23-43 (flags), :62-76 (G1), :113-133 (G2)
  * prepare_inputs / verify_groth16         groth16/verify.rs:53-78: e(A,B) e(L,gamma') e(C,-delta') == e(alpha, beta'')  with beta'' = -beta'
  * the `bn` facts the glue relies on: Fq::from_slice rejects >= p; Fr is not range-checked (x acts as x mod r);
    AffineG2::new = curve equation then [r]Q = O; get_ys_from_x_unchecked returns the two roots ordered -- G1: numerically smaller first;
    G2: by the REAL part c0 only (mode "reference"); gnark orders by (c1, c0) (mode "gnark", with gnark's equation
    e(A,B) = e(alpha,beta) e(L,gamma) e(C,delta)); pairing_batch skips pairs with an identity operand.
Status bytes: those of include/bn254_verify.h."""
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
U = 4965661367192848881
REJECT, ACCEPT, ERR_NOT_MEMBER, ERR_NOT_ON_CURVE, ERR_NOT_IN_SUBGROUP, ERR_INPUT_LEN, ERR_MALFORMED = range(7)
MODE_REFERENCE, MODE_GNARK = 0, 1
HALF = (P - 1) // 2


class Malformed(Exception):
    pass


# ------------------------------------------------------------------------------------------------ Fp, Fp2 (pairs of ints, i^2 = -1)
def fp_sqrt(a):
    """p = 3 mod 4: a^((p+1)/4), checked."""
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a % P else None


def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, (-a[1]) * n % P)


def f2_sqrt(a):
    """A square root of a in Fp2 (either one), or None: the norm method."""
    if a == (0, 0):
        return (0, 0)
    alpha = fp_sqrt((a[0] * a[0] + a[1] * a[1]) % P)
    if alpha is None:
        return None
    inv2 = pow(2, -1, P)
    for sgn in (1, -1):
        delta = (a[0] + sgn * alpha) * inv2 % P
        x0 = fp_sqrt(delta)
        if x0 is None or x0 == 0:
            continue
        x1 = a[1] * pow(2 * x0, -1, P) % P
        if f2_mul((x0, x1), (x0, x1)) == (a[0] % P, a[1] % P):
            return (x0, x1)
    if a[1] % P == 0:                                   # a in Fp and a non-residue there: sqrt(a) = sqrt(-a) * i
        s = fp_sqrt((-a[0]) % P)
        if s is not None:
            return (0, s)
    return None


B1 = 3
B2 = f2_mul((3, 0), f2_inv((9, 1)))                    # twist: y^2 = x^3 + 3 / (9 + i)


# ------------------------------------------------------------------------------------------------ affine curve arithmetic, None = identity
class Curve:
    def __init__(self, add, sub, mul, inv, neg, zero):
        self.add, self.sub, self.mul, self.inv, self.neg, self.zero = add, sub, mul, inv, neg, zero

    def dbl(self, p):
        if p is None or p[1] == self.zero:
            return None
        x, y = p
        xx = self.mul(x, x)
        lam = self.mul(self.add(self.add(xx, xx), xx), self.inv(self.add(y, y)))
        nx = self.sub(self.sub(self.mul(lam, lam), x), x)
        return (nx, self.sub(self.mul(lam, self.sub(x, nx)), y))

    def plus(self, p, q):
        if p is None: return q
        if q is None: return p
        if p[0] == q[0]:
            return self.dbl(p) if p[1] == q[1] else None
        lam = self.mul(self.sub(q[1], p[1]), self.inv(self.sub(q[0], p[0])))
        nx = self.sub(self.sub(self.mul(lam, lam), p[0]), q[0])
        return (nx, self.sub(self.mul(lam, self.sub(p[0], nx)), p[1]))

    def times(self, p, k):
        acc = None
        for bit in bin(k)[2:] if k else "":
            acc = self.dbl(acc)
            if bit == "1":
                acc = self.plus(acc, p)
        return acc

    def negate(self, p):
        return None if p is None else (p[0], self.neg(p[1]))


G1C = Curve(lambda a, b: (a + b) % P, lambda a, b: (a - b) % P, lambda a, b: a * b % P, lambda a: pow(a, -1, P), lambda a: (-a) % P, 0)
G2C = Curve(f2_add, f2_sub, f2_mul, f2_inv, f2_neg, (0, 0))


def g1_on_curve(p): return (p[1] * p[1] - p[0] * p[0] * p[0] - B1) % P == 0
def g2_on_curve(p): return f2_sub(f2_mul(p[1], p[1]), f2_add(f2_mul(f2_mul(p[0], p[0]), p[0]), B2)) == (0, 0)


# ------------------------------------------------------------------------------------------------ gnark point codecs (converter.rs)
def _flagged_x(buf):
    """deserialize_with_flags, converter.rs:23-43: (x mod p, flag)."""
    if len(buf) != 32:
        raise Malformed("x length")
    flag = buf[0] >> 6
    if flag == 0b00:
        raise Malformed("flag 0b00 (constants.rs:24 panics)")
    if flag == 0b01:
        if (buf[0] & 0x3f) or any(buf[1:]):
            raise Malformed("infinity flag with non-zero bits")
        return 0, flag
    return int.from_bytes(bytes([buf[0] & 0x3f]) + bytes(buf[1:]), "big") % P, flag


def decompress_g1(buf):
    """unchecked_compressed_x_to_g1_point, converter.rs:62-76 (no special case for the infinity flag: x = 0 falls through)."""
    x, flag = _flagged_x(buf)
    y = fp_sqrt((x * x * x + B1) % P)
    if y is None:
        raise Malformed("no square root")
    lo, hi = min(y, P - y), max(y, P - y)              # get_ys_from_x_unchecked: (smaller, larger)
    return (x, hi if flag == 0b11 else lo)


def decompress_g2(buf, mode):
    """unchecked_compressed_x_to_g2_point, converter.rs:113-133; the order of the two roots is the mode."""
    if len(buf) != 64:
        raise Malformed("x length")
    x1, flag = _flagged_x(buf[:32])
    x0 = int.from_bytes(buf[32:64], "big") % P
    if flag == 0b01:
        raise Malformed("identity in a key")          # the reference returns the generator here (converter.rs:122-124); never in a real key
    x = (x0, x1)
    y = f2_sqrt(f2_add(f2_mul(f2_mul(x, x), x), B2))
    if y is None:
        raise Malformed("no square root")
    ny = f2_neg(y)
    if mode == MODE_REFERENCE:
        first, second = (y, ny) if y[0] <= ny[0] else (ny, y)          # bn: ordered by c0 only
    else:
        large = lambda v: v[1] > HALF or (v[1] == 0 and v[0] > HALF)   # gnark: lexicographically largest by (c1, c0)
        first, second = (ny, y) if large(y) else (y, ny)
    return (x, second if flag == 0b11 else first)


def _fq(buf):
    v = int.from_bytes(buf, "big")
    return v if v < P else None


def load_proof(proof):
    """load_groth16_proof_from_bytes: returns (status, None) on the first loader error in the reference's order (A; B member, curve,
    r-torsion; C), else (None, (A, B, C))."""
    if len(proof) < 256:
        return ERR_MALFORMED, None
    ax, ay = _fq(proof[0:32]), _fq(proof[32:64])
    if ax is None or ay is None: return ERR_NOT_MEMBER, None
    if not g1_on_curve((ax, ay)): return ERR_NOT_ON_CURVE, None
    bx1, bx0, by1, by0 = (_fq(proof[64 + 32 * i:96 + 32 * i]) for i in range(4))
    if None in (bx1, bx0, by1, by0): return ERR_NOT_MEMBER, None
    B = ((bx0, bx1), (by0, by1))
    if not g2_on_curve(B): return ERR_NOT_ON_CURVE, None
    if G2C.times(B, R) is not None: return ERR_NOT_IN_SUBGROUP, None
    cx, cy = _fq(proof[192:224]), _fq(proof[224:256])
    if cx is None or cy is None: return ERR_NOT_MEMBER, None
    if not g1_on_curve((cx, cy)): return ERR_NOT_ON_CURVE, None
    return None, ((ax, ay), B, (cx, cy))


def load_vk(vk, mode):
    """load_groth16_verifying_key_from_bytes: dict of parsed points (beta as stored: negated)."""
    if len(vk) < 292:
        raise Malformed("short key")
    nk = int.from_bytes(vk[288:292], "big")
    if len(vk) < 292 + 32 * nk + 4:
        raise Malformed("short key")
    k = [decompress_g1(vk[292 + 32 * i:324 + 32 * i]) for i in range(nk)]
    return {"alpha": decompress_g1(vk[0:32]), "beta2": G2C.negate(decompress_g2(vk[64:128], mode)), "gamma2": decompress_g2(vk[128:192], mode),
            "delta2": decompress_g2(vk[224:288], mode), "k": k}


# ------------------------------------------------------------------------------------------------ pairing in the polynomial basis
def p12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(22, 11, -1):
        v = t[k]
        if v:
            t[k - 6] += 18 * v
            t[k - 12] -= 82 * v
    return [x % P for x in t[:12]]


def p12_pow(a, e):
    r = [1] + [0] * 11
    while e:
        if e & 1:
            r = p12_mul(r, a)
        a = p12_mul(a, a)
        e >>= 1
    return r


def p12_inv(a): return p12_pow(a, P ** 12 - 2)
def p12_sub(a, b): return [(x - y) % P for x, y in zip(a, b)]
def p12_scalar(v): return [v % P] + [0] * 11


def untwist(q):
    """q on the twist -> E(Fp12): i = w^6 - 9, point (x w^2, y w^3)."""
    (x0, x1), (y0, y1) = q
    x = [0] * 12; y = [0] * 12
    x[2] = (x0 - 9 * x1) % P; x[8] = x1
    y[3] = (y0 - 9 * y1) % P; y[9] = y1
    return x, y


def _ec12_double(p):
    x, y = p
    lam = p12_mul(p12_mul(p12_scalar(3), p12_mul(x, x)), p12_inv(p12_mul(p12_scalar(2), y)))
    nx = p12_sub(p12_mul(lam, lam), p12_mul(p12_scalar(2), x))
    return nx, p12_sub(p12_mul(lam, p12_sub(x, nx)), y), lam


def _ec12_add(p, q):
    lam = p12_mul(p12_sub(q[1], p[1]), p12_inv(p12_sub(q[0], p[0])))
    nx = p12_sub(p12_sub(p12_mul(lam, lam), p[0]), q[0])
    return nx, p12_sub(p12_mul(lam, p12_sub(p[0], nx)), p[1]), lam


def miller(g1, q):
    """f_{6u+2,Q}(P) times the two Frobenius lines, over the bits of 6u+2; identity operands give 1 (bn::pairing_batch skips such pairs)."""
    if g1 is None or q is None:
        return p12_scalar(1)
    px, py = p12_scalar(g1[0]), p12_scalar(g1[1])
    Q = untwist(q)

    def line(t, lam):
        return p12_sub(p12_sub(py, t[1]), p12_mul(lam, p12_sub(px, t[0])))

    f = p12_scalar(1)
    T = Q
    for bit in bin(6 * U + 2)[3:]:
        nx, ny, lam = _ec12_double(T)
        f = p12_mul(p12_mul(f, f), line(T, lam))
        T = (nx, ny)
        if bit == "1":
            nx, ny, lam = _ec12_add(T, Q)
            f = p12_mul(f, line(T, lam))
            T = (nx, ny)
    frob = lambda pt: (p12_pow(pt[0], P), p12_pow(pt[1], P))
    Q1 = frob(Q); Q2 = frob(Q1); nQ2 = (Q2[0], [(-v) % P for v in Q2[1]])
    nx, ny, lam = _ec12_add(T, Q1); f = p12_mul(f, line(T, lam)); T = (nx, ny)
    nx, ny, lam = _ec12_add(T, nQ2); f = p12_mul(f, line(T, lam))
    return f


def final_exp(f):
    return p12_pow(f, (P ** 12 - 1) // R)


def slow_pairing_plain(g1, q):
    return final_exp(miller(g1, q))


# ------------------------------------------------------------------------------------------------ verify()
def prepare_inputs(k, inputs):
    """groth16/verify.rs:53-63: K0 + sum x_i K_(i+1); the scalars are raw 256-bit integers consumed bit by bit (x acts as x mod r)."""
    acc = k[0]
    for x, b in zip(inputs, k[1:]):
        acc = G1C.plus(acc, G1C.times(b, x))
    return acc


def verify(proof, vk, inputs, mode=MODE_REFERENCE):
    """Status byte of Groth16Verifier::verify on these bytes; inputs: list of ints (32-byte big-endian values)."""
    st, pts = load_proof(proof)                        # lib.rs:45: the proof first
    if st is not None:
        return st
    try:
        key = load_vk(vk, mode)                        # lib.rs:46
    except Malformed:
        return ERR_MALFORMED
    if len(inputs) + 1 != len(key["k"]):               # verify.rs:54-56
        return ERR_INPUT_LEN
    A, B, C = pts
    L = prepare_inputs(key["k"], inputs)
    if mode == MODE_REFERENCE:
        # verify.rs:70-77, literally: pairing_batch([(A,B), (L,gamma), (C,-delta)]) == pairing(alpha, beta)   (beta stored negated)
        lhs = miller(A, B)
        lhs = p12_mul(lhs, miller(L, key["gamma2"]))
        lhs = p12_mul(lhs, miller(C, G2C.negate(key["delta2"])))
        rhs = miller(key["alpha"], key["beta2"])
    else:
        # gnark: e(A,B) == e(alpha,beta) e(L,gamma) e(C,delta), with gnark-exact decompression (beta2 here is the negated stored value)
        lhs = miller(A, B)
        rhs = miller(key["alpha"], G2C.negate(key["beta2"]))
        rhs = p12_mul(rhs, miller(L, key["gamma2"]))
        rhs = p12_mul(rhs, miller(C, key["delta2"]))
    return ACCEPT if final_exp(lhs) == final_exp(rhs) else REJECT
