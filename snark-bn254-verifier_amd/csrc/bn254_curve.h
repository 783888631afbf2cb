// bn254_curve.h -- G1 / G2 group arithmetic for the batch verifier (device + host).
//
// G1: y^2 = x^3 + 3 over Fp, prime order r.  Homogeneous projective coordinates with the COMPLETE addition law of
//     Renes-Costello-Batina (2016, Alg. 7-9 for a = 0): no exceptional cases, identity = (0 : 1 : 0), so the
//     public-input linear combination K0 + sum x_i K_i (reference groth16/verify.rs:53-63) needs no branches even
//     for adversarial scalars (P = Q, P = -Q, identity all handled by the same straight-line code).
// G2: y^2 = x^3 + 3/xi over Fp2 (D-type sextic twist), homogeneous projective, Costello-Lange-Naehrig doubling /
//     mixed addition.  These are NOT complete; every exceptional input (T = +-Q, 2-torsion, identity) drives Z to 0
//     and Z = 0 is sticky under both formulas, so callers only test the final Z (see g2_in_subgroup).
// Replaces bn::{G1, G2, AffineG1, AffineG2} as used at groth16/verify.rs:58-62 and converter.rs:87,152.
#pragma once
#include "bn254_tower.h"

namespace bn254 {

struct G1Aff { Fp x, y; };
struct G1Proj { Fp x, y, z; };
struct G2Aff { Fp2 x, y; };
struct G2Proj { Fp2 x, y, z; };

// ------------------------------------------------------------------ G1
BN_HD G1Proj g1_identity() { G1Proj r; r.x = fp_zero(); r.y = fp_one(); r.z = fp_zero(); return r; }
BN_HD G1Proj g1_from_affine(const G1Aff& a) { G1Proj r; r.x = a.x; r.y = a.y; r.z = fp_one(); return r; }
BN_HD Fp fp_mul9(const Fp& a) { return fp_lincomb_reduce(9, a, 0, a); }  // b3 = 3 b = 9
BN_HD bool g1_on_curve(const G1Aff& p) {
  Fp lhs = fp_sqr(p.y);
  Fp rhs = fp_add(fp_mul(fp_sqr(p.x), p.x), fp_from_limbs(BN_THREE));
  return fp_eq(lhs, rhs);
}
// The three formulas below are RCB16 Algorithms 7-9 in SUM-OF-PRODUCTS form (round 5): every output coordinate is ONE fp_dot over two products (one Montgomery
// reduction instead of two and no addition after it), and the sums that only feed products stay lazy (digit-wise, no carry pass: fp_dot takes digits up to 1.5 x 2^29
// here, checked by the bound tracker of tests/hostsim).  Same values as the textbook sequence; per mixed addition 1815 -> 1580 multiply-adds and a quarter of the
// carry passes (10 900 -> 8 800 issue cycles at the measured rates of the two instruction classes).
// RCB16 Algorithm 8: complete mixed addition, a = 0.  q must be a finite affine point.
BN_HD G1Proj g1_add_mixed(const G1Proj& p, const G1Aff& q) {
  const Fp t0 = fp_mul_nl(p.x, q.x);
  const Fp t1 = fp_mul_nl(p.y, q.y);
  // X1 Y2 + X2 Y1 = (X2 + Y2)(X1 + Y1) - t0 - t1, left lazy (digits up to 1.5 x 2^29): it only enters products
  const Fp t3 = fp_sub_lazy(fp_sub_lazy(fp_mul_nl(fp_add_lazy(q.x, q.y), fp_add_lazy(p.x, p.y)), t0), t1);
  const Fp t4 = fp_add_lazy(fp_mul_nl(q.y, p.z), p.y);                 // Y2 Z1 + Y1
  const Fp y3 = fp_mul9(fp_add_lazy(fp_mul_nl(q.x, p.z), p.x));        // b3 (X2 Z1 + X1)
  const Fp t0x3 = fp_add_lazy(fp_add_lazy(t0, t0), t0);                 // 3 X1 X2
  const Fp t2 = fp_mul9(p.z);                                           // b3 Z1
  const Fp zs = fp_add_lazy(t1, t2), td = fp_sub_lazy(t1, t2);          // Y1 Y2 +- b3 Z1
  G1Proj r;
  r.x = fp_dot(dplus(t3, td), dplus(t4, fp_neg(y3)));                   // (a negated operand instead of a minus term: no second accumulator)
  r.y = fp_dot(dplus(td, zs), dplus(y3, t0x3));
  r.z = fp_dot(dplus(zs, t4), dplus(t0x3, t3));
  return r;
}
// RCB16 Algorithm 7: complete projective addition, a = 0
BN_HD G1Proj g1_add(const G1Proj& p, const G1Proj& q) {
  const Fp t0 = fp_mul_nl(p.x, q.x), t1 = fp_mul_nl(p.y, q.y), t2 = fp_mul_nl(p.z, q.z);
  const Fp t3 = fp_sub_lazy(fp_sub_lazy(fp_mul_nl(fp_add_lazy(p.x, p.y), fp_add_lazy(q.x, q.y)), t0), t1);             // X1 Y2 + X2 Y1 (lazy: enters products only)
  const Fp t4 = fp_sub_lazy(fp_sub_lazy(fp_mul_nl(fp_add_lazy(p.y, p.z), fp_add_lazy(q.y, q.z)), t1), t2);             // Y1 Z2 + Y2 Z1
  const Fp xz = fp_sub_lazy(fp_sub_lazy(fp_mul_nl(fp_add_lazy(p.x, p.z), fp_add_lazy(q.x, q.z)), t0), t2);             // X1 Z2 + X2 Z1
  const Fp y3 = fp_mul9(fp_norm(xz));                                   // b3 (X1 Z2 + X2 Z1)
  const Fp t0x3 = fp_add_lazy(fp_add_lazy(t0, t0), t0);                 // 3 X1 X2
  const Fp t2b = fp_mul9(t2);                                           // b3 Z1 Z2
  const Fp zs = fp_add_lazy(t1, t2b), td = fp_sub_lazy(t1, t2b);
  const Fp t4n = fp_norm(t4);
  G1Proj r;
  r.x = fp_dot(dplus(t3, td), dplus(t4n, fp_neg(y3)));
  r.y = fp_dot(dplus(td, zs), dplus(y3, t0x3));
  r.z = fp_dot(dplus(zs, t4n), dplus(t0x3, t3));
  return r;
}
// RCB16 Algorithm 9: complete doubling, a = 0
BN_HD G1Proj g1_dbl(const G1Proj& p) {
  const Fp t0 = fp_sqr_nl(p.y);
  const Fp z8 = fp_lincomb(8, t0, 0, t0);                               // 8 Y^2 (one carry pass)
  const Fp t1 = fp_mul_nl(p.y, p.z);
  const Fp t2 = fp_mul9(fp_sqr_nl(p.z));                                // b3 Z^2
  const Fp ys = fp_add_lazy(t0, t2);                                    // Y^2 + b3 Z^2
  const Fp td = fp_lincomb(1, t0, -3, t2);                              // Y^2 - 3 b3 Z^2
  const Fp xy = fp_mul_nl(p.x, p.y);
  G1Proj r;
  r.x = fp_dot(dterm<2>(td, xy));
  r.y = fp_dot(dplus(t2, z8), dplus(td, ys));
  r.z = fp_mul_nl(t1, z8);
  return r;
}
BN_HD bool g1_is_identity(const G1Proj& p) { return fp_is_zero(p.z); }
BN_HD G1Aff g1_to_affine(const G1Proj& p) {  // identity -> (0, 0)
  Fp zi = fp_inv(p.z);
  G1Aff r; r.x = fp_mul(p.x, zi); r.y = fp_mul(p.y, zi);
  return r;
}
BN_HD G1Aff g1_neg(const G1Aff& p) { G1Aff r; r.x = p.x; r.y = fp_neg(p.y); return r; }

// ------------------------------------------------------------------ G2
BN_HD Fp2 g2_twist_b() { return fp2_from_limbs(BN_TWIST_B0, BN_TWIST_B1); }
BN_HD bool g2_on_curve(const G2Aff& p) {
  Fp2 lhs = fp2_sqr(p.y);
  Fp2 rhs = fp2_add(fp2_mul(fp2_sqr(p.x), p.x), g2_twist_b());
  return fp2_eq(lhs, rhs);
}
BN_HD G2Proj g2_from_affine(const G2Aff& a) { G2Proj r; r.x = a.x; r.y = a.y; r.z = fp2_one(); return r; }
BN_HD G2Aff g2_neg(const G2Aff& p) { G2Aff r; r.x = p.x; r.y = fp2_neg(p.y); return r; }
BN_HD G2Aff g2_select(bool c, const G2Aff& a, const G2Aff& b) { G2Aff r; r.x = fp2_select(c, a.x, b.x); r.y = fp2_select(c, a.y, b.y); return r; }

// Line through the running point, evaluated later at P = (xP, yP):  l = r0 * yP + (r1 * xP) w + r2 w^3
struct G2Line { Fp2 r0, r1, r2; };

// doubling step (Costello-Lange-Naehrig, scaled by 4 to avoid halvings): T <- 2T, returns the tangent line at T
BN_HD G2Line g2_double_step(G2Proj& t) {
  Fp2 A = fp2_mul(t.x, t.y);          // X Y          (= 2 A_cln)
  Fp2 B = fp2_sqr(t.y);               // Y^2
  Fp2 C = fp2_sqr(t.z);               // Z^2
  Fp2 E = fp2_mul(fp2_from_limbs(BN_TWIST_3B0, BN_TWIST_3B1), C);  // 3 b' Z^2
  Fp2 F = fp2_mul_small(E, 3);           // 9 b' Z^2
  Fp2 H = fp2_sub2(fp2_sqr(fp2_add(t.y, t.z)), B, C);              // 2 Y Z
  Fp2 J = fp2_sqr(t.x);               // X^2
  Fp2 BF = fp2_add(B, F);
  G2Line l;
  l.r0 = fp2_neg(H);
  l.r1 = fp2_mul_small(J, 3);
  l.r2 = fp2_sub(E, B);
  // X3 = 2 X Y (B - F), Y3 = (B + F)^2 - 12 E^2, Z3 = 4 B H   (all x4 relative to CLN: same projective point)
  Fp2 E2 = fp2_sqr(E);
  t.x = fp2_dbl(fp2_mul(A, fp2_sub(B, F)));
  t.y = fp2_sub(fp2_sqr(BF), fp2_mul_small(E2, 12));
  t.z = fp2_mul_small(fp2_mul(B, H), 4);
  return l;
}
// mixed addition step: T <- T + Q (Q affine), returns the line through T and Q
BN_HD G2Line g2_add_step(G2Proj& t, const G2Aff& q) {
  Fp2 O = fp2_sub(t.y, fp2_mul(q.y, t.z));
  Fp2 L = fp2_sub(t.x, fp2_mul(q.x, t.z));
  Fp2 C = fp2_sqr(O), D = fp2_sqr(L);
  Fp2 E = fp2_mul(L, D);
  Fp2 F = fp2_mul(t.z, C);
  Fp2 G = fp2_mul(t.x, D);
  Fp2 H = fp2_sub(fp2_add(E, F), fp2_dbl(G));
  G2Line l;
  l.r0 = L;
  l.r1 = fp2_neg(O);
  l.r2 = fp2_dotp(pp(q.x, O), pm(L, q.y));
  Fp2 y3 = fp2_dotp(pp(fp2_sub(G, H), O), pm(t.y, E));
  t.x = fp2_mul(L, H);
  t.y = y3;
  t.z = fp2_mul(E, t.z);
  return l;
}
// full projective addition (no line), same incomplete law; used once or twice per subgroup check
BN_HD G2Proj g2_add_proj(const G2Proj& p, const G2Proj& q) {
  // bring both to the common denominator Z1 Z2 and reuse the mixed formulas' structure
  Fp2 y2z1 = fp2_mul(q.y, p.z), x2z1 = fp2_mul(q.x, p.z);
  Fp2 y1z2 = fp2_mul(p.y, q.z), x1z2 = fp2_mul(p.x, q.z);
  Fp2 zz = fp2_mul(p.z, q.z);
  Fp2 O = fp2_sub(y1z2, y2z1), L = fp2_sub(x1z2, x2z1);
  Fp2 C = fp2_sqr(O), D = fp2_sqr(L);
  Fp2 E = fp2_mul(L, D);
  Fp2 F = fp2_mul(zz, C);
  Fp2 G = fp2_mul(x1z2, D);
  Fp2 H = fp2_sub(fp2_add(E, F), fp2_dbl(G));
  G2Proj r;
  r.x = fp2_mul(L, H);
  r.y = fp2_dotp(pp(fp2_sub(G, H), O), pm(y1z2, E));
  r.z = fp2_mul(E, zz);
  return r;
}
// the untwist-Frobenius-twist endomorphism psi on projective points: (conj X * g2, conj Y * g3, conj Z)
BN_HD G2Proj g2_psi(const G2Proj& p) {
  G2Proj r;
  r.x = fp2_mul(fp2_conj(p.x), frob_coeff(1, 2));
  r.y = fp2_mul(fp2_conj(p.y), frob_coeff(1, 3));
  r.z = fp2_conj(p.z);
  return r;
}
BN_HD G2Aff g2_psi_affine(const G2Aff& p) {
  G2Aff r;
  r.x = fp2_mul(fp2_conj(p.x), frob_coeff(1, 2));
  r.y = fp2_mul(fp2_conj(p.y), frob_coeff(1, 3));
  return r;
}
BN_HD G2Aff g2_psi2_affine(const G2Aff& p) {  // psi^2: constants lie in Fp
  G2Aff r;
  r.x = fp2_mul_fp(p.x, frob_coeff(2, 2).c0);
  r.y = fp2_mul_fp(p.y, frob_coeff(2, 3).c0);
  return r;
}
BN_HD bool g2_proj_eq(const G2Proj& a, const G2Proj& b) {  // both finite
  return fp2_eq(fp2_mul(a.x, b.z), fp2_mul(b.x, a.z)) & fp2_eq(fp2_mul(a.y, b.z), fp2_mul(b.y, a.z));
}
// [u]Q over the NAF of u, Q a finite point ON THE TWIST (any order): exceptional steps zero Z, which then sticks.
// LQ is a callable returning Q: the kernels re-load it from the workspace at every addition instead of keeping Q (and -Q)
// in 72 registers across the loop.
template <class LQ>
BN_HD G2Proj g2_mul_u_ld(const LQ& load_q) {
  G2Proj t = g2_from_affine(load_q());
  for (int i = 1; i < BN_U_NAF_LEN; i++) {
    (void)g2_double_step(t);
    int d = BN_U_NAF[i];
    if (d != 0) {  // public constant: wave-uniform branch
      G2Aff q = load_q();
      if (d < 0) q.y = fp2_neg(q.y);
      (void)g2_add_step(t, q);
    }
  }
  return t;
}
// r-torsion test for a point already known to be on the twist (El Housni-Guillevic-Piellard 2022, as gnark-crypto):
//   [u+1]Q + psi([u]Q) + psi^2([u]Q) == psi^3([2u]Q)
// Same accept set as the reference's [r-1]Q + Q == O (bn's AffineG2::new, reference converter.rs:152).
// Any exceptional case of the incomplete formulas means ord(Q) is small or a relation that no point of prime order r
// satisfies, hence Q is not in G2; it leaves Z = 0 on one side, which is rejected.
// Evaluation order: one accumulator and one running psi-image, so that at most two projective points are live.
template <class LQ>
BN_HD bool g2_in_subgroup_ld(const LQ& load_q) {
  G2Proj b = g2_mul_u_ld(load_q);     // [u]Q
  G2Proj acc = b;
  { G2Aff q = load_q(); (void)g2_add_step(acc, q); }   // [u+1]Q
  b = g2_psi(b);                      // psi([u]Q)
  acc = g2_add_proj(b, acc);
  b = g2_psi(b);                      // psi^2([u]Q)
  acc = g2_add_proj(b, acc);
  b = g2_psi(b);                      // psi^3([u]Q)
  (void)g2_double_step(b);            // psi^3([2u]Q)
  bool finite = !fp2_is_zero(acc.z) & !fp2_is_zero(b.z);
  return finite & g2_proj_eq(acc, b);
}
struct G2AffRef { const G2Aff& q; BN_HD G2Aff operator()() const { return q; } };
BN_HD G2Proj g2_mul_u(const G2Aff& q) { return g2_mul_u_ld(G2AffRef{q}); }
BN_HD bool g2_in_subgroup(const G2Aff& q) { return g2_in_subgroup_ld(G2AffRef{q}); }

}  // namespace bn254
