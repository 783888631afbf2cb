// bn254_tower.h -- Fp2 / Fp6 / Fp12 tower over bn254_fp.h.
//   Fp2 = Fp[i]/(i^2+1), Fp6 = Fp2[v]/(v^3 - xi), xi = 9 + i, Fp12 = Fp6[w]/(w^2 - v)   
// Device replacement for bn::{Fq2, Fq6, Fq12} (Fq2::{mul,inverse}, Fq6::{mul,squared,inverse,
// frobenius_map}, Fq12::{mul, squared, mul_by_024, cyclotomic_squared, frobenius_map, inverse}).
//
// Bound discipline (see bn254_fp.h): products come back with |x| <~ 1.4 p, sums add bounds, and the only
// operations that multiply a bound are the xi-multiplications (x10); those reduce in the same pass
// (fp_lincomb_reduce).  Every public function here accepts component bounds <= 6 and returns component
// bounds <= 6 unless stated, which tests/hostsim verifies with the bound tracker on every call.
#pragma once
#include "bn254_fp.h"

namespace bn254 {

struct Fp2 { Fp c0, c1; };
struct Fp6 { Fp2 c0, c1, c2; };
struct Fp12 { Fp6 c0, c1; };

// ------------------------------------------------------------------ Fp2
BN_HD Fp2 fp2_zero() { Fp2 r; r.c0 = fp_zero(); r.c1 = fp_zero(); return r; }
BN_HD Fp2 fp2_one() { Fp2 r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }
BN_HD Fp2 fp2_add(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_add(a.c0, b.c0); r.c1 = fp_add(a.c1, b.c1); return r; }
BN_HD Fp2 fp2_sub(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_sub(a.c0, b.c0); r.c1 = fp_sub(a.c1, b.c1); return r; }
BN_HD Fp2 fp2_neg(const Fp2& a) { Fp2 r; r.c0 = fp_neg(a.c0); r.c1 = fp_neg(a.c1); return r; }
BN_HD Fp2 fp2_dbl(const Fp2& a) { Fp2 r; r.c0 = fp_dbl(a.c0); r.c1 = fp_dbl(a.c1); return r; }
BN_HD Fp2 fp2_conj(const Fp2& a) { Fp2 r; r.c0 = a.c0; r.c1 = fp_neg(a.c1); return r; }
BN_HD Fp2 fp2_reduce(const Fp2& a) { Fp2 r; r.c0 = fp_reduce(a.c0); r.c1 = fp_reduce(a.c1); return r; }
BN_HD Fp2 fp2_select(bool c, const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_select(c, a.c0, b.c0); r.c1 = fp_select(c, a.c1, b.c1); return r; }
// a + b + c, a - b - c with a single carry pass each
BN_HD Fp fp_sub2(const Fp& a, const Fp& b, const Fp& c) { return fp_norm(fp_sub_lazy(fp_sub_lazy(a, b), c)); }
BN_HD Fp2 fp2_sub2(const Fp2& a, const Fp2& b, const Fp2& c) { Fp2 r; r.c0 = fp_sub2(a.c0, b.c0, c.c0); r.c1 = fp_sub2(a.c1, b.c1, c.c1); return r; }

// Karatsuba: 3 Fp products.  One of the two operand sums is normalised, the other enters the product lazily.
BN_HD Fp2 fp2_mul(const Fp2& a, const Fp2& b) {
  Fp v0 = fp_mul(a.c0, b.c0);
  Fp v1 = fp_mul(a.c1, b.c1);
  Fp s = fp_mul(fp_add(a.c0, a.c1), fp_add_lazy(b.c0, b.c1));
  Fp2 r;
  r.c0 = fp_sub(v0, v1);
  r.c1 = fp_sub2(s, v0, v1);
  return r;
}
// complex squaring: 2 Fp products
BN_HD Fp2 fp2_sqr(const Fp2& a) {
  Fp2 r;
  r.c0 = fp_mul(fp_add(a.c0, a.c1), fp_sub_lazy(a.c0, a.c1));
  r.c1 = fp_mul(fp_add_lazy(a.c0, a.c0), a.c1);
  return r;
}
BN_HD Fp2 fp2_mul_fp(const Fp2& a, const Fp& f) { Fp2 r; r.c0 = fp_mul(a.c0, f); r.c1 = fp_mul(a.c1, f); return r; }
// (a0 + a1 i)(9 + i) = (9 a0 - a1) + (a0 + 9 a1) i, reduced in the same pass
BN_HD Fp2 fp2_mul_xi(const Fp2& a) {
  Fp2 r;
  r.c0 = fp_lincomb_reduce(9, a.c0, -1, a.c1);
  r.c1 = fp_lincomb_reduce(1, a.c0, 9, a.c1);
  return r;
}
BN_HD Fp2 fp2_mul_small(const Fp2& a, int32_t k) {  // k * a reduced
  Fp2 r;
  r.c0 = fp_lincomb_reduce(k, a.c0, 0, a.c0);
  r.c1 = fp_lincomb_reduce(k, a.c1, 0, a.c1);
  return r;
}
BN_HD Fp2 fp2_inv(const Fp2& a) {  // 0 -> 0
  Fp n = fp_add(fp_sqr(a.c0), fp_sqr(a.c1));
  Fp ni = fp_inv(n);
  Fp2 r;
  r.c0 = fp_mul(a.c0, ni);
  r.c1 = fp_neg(fp_mul(a.c1, ni));
  return r;
}
BN_HD bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) & fp_is_zero(a.c1); }
BN_HD bool fp2_eq(const Fp2& a, const Fp2& b) { return fp_eq(a.c0, b.c0) & fp_eq(a.c1, b.c1); }
BN_HD Fp2 fp2_from_limbs(const uint32_t* c0, const uint32_t* c1) { Fp2 r; r.c0 = fp_from_limbs(c0); r.c1 = fp_from_limbs(c1); return r; }

// out-of-line copies for the places where code size matters more than the call
BN_HD_NOINLINE Fp2 fp2_mul_nl(const Fp2& a, const Fp2& b) { return fp2_mul(a, b); }
BN_HD_NOINLINE Fp2 fp2_sqr_nl(const Fp2& a) { return fp2_sqr(a); }

// ------------------------------------------------------------------ Fp6
BN_HD Fp6 fp6_zero() { Fp6 r; r.c0 = fp2_zero(); r.c1 = fp2_zero(); r.c2 = fp2_zero(); return r; }
BN_HD Fp6 fp6_one() { Fp6 r; r.c0 = fp2_one(); r.c1 = fp2_zero(); r.c2 = fp2_zero(); return r; }
BN_HD Fp6 fp6_add(const Fp6& a, const Fp6& b) { Fp6 r; r.c0 = fp2_add(a.c0, b.c0); r.c1 = fp2_add(a.c1, b.c1); r.c2 = fp2_add(a.c2, b.c2); return r; }
BN_HD Fp6 fp6_sub(const Fp6& a, const Fp6& b) { Fp6 r; r.c0 = fp2_sub(a.c0, b.c0); r.c1 = fp2_sub(a.c1, b.c1); r.c2 = fp2_sub(a.c2, b.c2); return r; }
BN_HD Fp6 fp6_neg(const Fp6& a) { Fp6 r; r.c0 = fp2_neg(a.c0); r.c1 = fp2_neg(a.c1); r.c2 = fp2_neg(a.c2); return r; }
BN_HD Fp6 fp6_reduce(const Fp6& a) { Fp6 r; r.c0 = fp2_reduce(a.c0); r.c1 = fp2_reduce(a.c1); r.c2 = fp2_reduce(a.c2); return r; }
BN_HD Fp6 fp6_select(bool c, const Fp6& a, const Fp6& b) { Fp6 r; r.c0 = fp2_select(c, a.c0, b.c0); r.c1 = fp2_select(c, a.c1, b.c1); r.c2 = fp2_select(c, a.c2, b.c2); return r; }
BN_HD Fp6 fp6_mul_v(const Fp6& a) { Fp6 r; r.c0 = fp2_mul_xi(a.c2); r.c1 = a.c0; r.c2 = a.c1; return r; }
// Karatsuba / Toom-style: 6 Fp2 products
BN_HD Fp6 fp6_mul(const Fp6& a, const Fp6& b) {
  Fp2 v0 = fp2_mul_nl(a.c0, b.c0), v1 = fp2_mul_nl(a.c1, b.c1), v2 = fp2_mul_nl(a.c2, b.c2);
  Fp2 t12 = fp2_sub2(fp2_mul_nl(fp2_add(a.c1, a.c2), fp2_add(b.c1, b.c2)), v1, v2);
  Fp2 t01 = fp2_sub2(fp2_mul_nl(fp2_add(a.c0, a.c1), fp2_add(b.c0, b.c1)), v0, v1);
  Fp2 t02 = fp2_sub2(fp2_mul_nl(fp2_add(a.c0, a.c2), fp2_add(b.c0, b.c2)), v0, v2);
  Fp6 r;
  r.c0 = fp2_add(v0, fp2_mul_xi(t12));
  r.c1 = fp2_reduce(fp2_add(t01, fp2_mul_xi(v2)));
  r.c2 = fp2_reduce(fp2_add(t02, v1));
  return r;
}
// Chung-Hasan SQR2: 2 products + 3 squarings in Fp2
BN_HD Fp6 fp6_sqr(const Fp6& a) {
  Fp2 s0 = fp2_sqr_nl(a.c0);
  Fp2 s1 = fp2_dbl(fp2_mul_nl(a.c0, a.c1));
  Fp2 s2 = fp2_sqr_nl(fp2_add(fp2_sub(a.c0, a.c1), a.c2));
  Fp2 s3 = fp2_dbl(fp2_mul_nl(a.c1, a.c2));
  Fp2 s4 = fp2_sqr_nl(a.c2);
  Fp6 r;
  r.c0 = fp2_add(s0, fp2_mul_xi(s3));
  r.c1 = fp2_add(s1, fp2_mul_xi(s4));
  r.c2 = fp2_reduce(fp2_sub2(fp2_add(fp2_add(s1, s2), s3), s0, s4));
  return r;
}
BN_HD Fp6 fp6_mul_fp2(const Fp6& a, const Fp2& b) { Fp6 r; r.c0 = fp2_mul_nl(a.c0, b); r.c1 = fp2_mul_nl(a.c1, b); r.c2 = fp2_mul_nl(a.c2, b); return r; }
BN_HD Fp6 fp6_mul_fp(const Fp6& a, const Fp& b) { Fp6 r; r.c0 = fp2_mul_fp(a.c0, b); r.c1 = fp2_mul_fp(a.c1, b); r.c2 = fp2_mul_fp(a.c2, b); return r; }
// a * (b0 + b1 v): 5 Fp2 products
BN_HD Fp6 fp6_mul_by_01(const Fp6& a, const Fp2& b0, const Fp2& b1) {
  Fp2 v0 = fp2_mul_nl(a.c0, b0), v1 = fp2_mul_nl(a.c1, b1);
  Fp2 t12 = fp2_sub(fp2_mul_nl(fp2_add(a.c1, a.c2), b1), v1);                       // a2 b1
  Fp2 t01 = fp2_sub2(fp2_mul_nl(fp2_add(a.c0, a.c1), fp2_add(b0, b1)), v0, v1);     // a0 b1 + a1 b0
  Fp2 t02 = fp2_sub(fp2_mul_nl(fp2_add(a.c0, a.c2), b0), v0);                       // a2 b0
  Fp6 r;
  r.c0 = fp2_add(v0, fp2_mul_xi(t12));
  r.c1 = fp2_reduce(t01);
  r.c2 = fp2_reduce(fp2_add(t02, v1));
  return r;
}
BN_HD Fp6 fp6_inv(const Fp6& a) {
  // A = a0^2 - xi a1 a2, B = xi a2^2 - a0 a1, C = a1^2 - a0 a2, F = a0 A + xi (a2 B + a1 C); 1/a = (A, B, C)/F
  Fp2 A = fp2_sub(fp2_sqr_nl(a.c0), fp2_mul_xi(fp2_mul_nl(a.c1, a.c2)));
  Fp2 B = fp2_sub(fp2_mul_xi(fp2_sqr_nl(a.c2)), fp2_mul_nl(a.c0, a.c1));
  Fp2 C = fp2_sub(fp2_sqr_nl(a.c1), fp2_mul_nl(a.c0, a.c2));
  Fp2 F = fp2_add(fp2_mul_nl(a.c0, A), fp2_mul_xi(fp2_add(fp2_mul_nl(a.c2, B), fp2_mul_nl(a.c1, C))));
  Fp2 Fi = fp2_inv(F);
  Fp6 r;
  r.c0 = fp2_mul_nl(A, Fi); r.c1 = fp2_mul_nl(B, Fi); r.c2 = fp2_mul_nl(C, Fi);
  return r;
}

// ------------------------------------------------------------------ Fp12
BN_HD Fp12 fp12_one() { Fp12 r; r.c0 = fp6_one(); r.c1 = fp6_zero(); return r; }
BN_HD Fp12 fp12_conj(const Fp12& a) { Fp12 r; r.c0 = a.c0; r.c1 = fp6_neg(a.c1); return r; }
BN_HD Fp12 fp12_reduce(const Fp12& a) { Fp12 r; r.c0 = fp6_reduce(a.c0); r.c1 = fp6_reduce(a.c1); return r; }
BN_HD Fp12 fp12_select(bool c, const Fp12& a, const Fp12& b) { Fp12 r; r.c0 = fp6_select(c, a.c0, b.c0); r.c1 = fp6_select(c, a.c1, b.c1); return r; }
BN_HD_NOINLINE Fp6 fp6_mul_nl(const Fp6& a, const Fp6& b) { return fp6_mul(a, b); }
// Karatsuba over Fp6: 3 Fp6 products = 18 Fp2 = 54 Fp
BN_HD Fp12 fp12_mul(const Fp12& a, const Fp12& b) {
  Fp6 v0 = fp6_mul_nl(a.c0, b.c0);
  Fp6 v1 = fp6_mul_nl(a.c1, b.c1);
  Fp6 s = fp6_mul_nl(fp6_add(a.c0, a.c1), fp6_add(b.c0, b.c1));
  Fp12 r;
  r.c0 = fp6_add(v0, fp6_mul_v(v1));
  r.c1 = fp6_sub(fp6_sub(s, v0), v1);
  return r;
}
// complex squaring over Fp6: 2 Fp6 products = 36 Fp
BN_HD Fp12 fp12_sqr(const Fp12& a) {
  Fp6 v0 = fp6_mul_nl(a.c0, a.c1);
  Fp6 t = fp6_mul_nl(fp6_add(a.c0, a.c1), fp6_add(a.c0, fp6_mul_v(a.c1)));
  Fp12 r;
  r.c0 = fp6_sub(fp6_sub(t, v0), fp6_mul_v(v0));
  r.c1 = fp6_add(v0, v0);
  return r;
}
// f * (d0 + (d3 + d4 v) w) with d0 in Fp2: the line value of a projective (variable-Q) step.  13 Fp2 products.
BN_HD Fp12 fp12_mul_by_034(const Fp12& f, const Fp2& d0, const Fp2& d3, const Fp2& d4) {
  Fp6 a = fp6_mul_fp2(f.c0, d0);
  Fp6 b = fp6_mul_by_01(f.c1, d3, d4);
  Fp6 e = fp6_mul_by_01(fp6_add(f.c0, f.c1), fp2_add(d0, d3), d4);
  Fp12 r;
  r.c0 = fp6_add(a, fp6_mul_v(b));
  r.c1 = fp6_sub(fp6_sub(e, a), b);
  return r;
}
// same with d0 in Fp (affine precomputed line scaled so that the constant term is y_P): 6 Fp + 10 Fp2 products
BN_HD Fp12 fp12_mul_by_034_fp(const Fp12& f, const Fp& d0, const Fp2& d3, const Fp2& d4) {
  Fp6 a = fp6_mul_fp(f.c0, d0);
  Fp6 b = fp6_mul_by_01(f.c1, d3, d4);
  Fp2 d03; d03.c0 = fp_add(d3.c0, d0); d03.c1 = d3.c1;
  Fp6 e = fp6_mul_by_01(fp6_add(f.c0, f.c1), d03, d4);
  Fp12 r;
  r.c0 = fp6_add(a, fp6_mul_v(b));
  r.c1 = fp6_sub(fp6_sub(e, a), b);
  return r;
}
BN_HD Fp12 fp12_inv(const Fp12& a) {
  // 1/(a0 + a1 w) = (a0 - a1 w) / (a0^2 - v a1^2)
  Fp6 d = fp6_sub(fp6_sqr(a.c0), fp6_mul_v(fp6_sqr(a.c1)));
  Fp6 di = fp6_inv(d);
  Fp12 r;
  r.c0 = fp6_mul_nl(a.c0, di);
  r.c1 = fp6_neg(fp6_mul_nl(a.c1, di));
  return r;
}
// Frobenius x -> x^(p^j), j = 1, 2, 3: conjugate (j odd) every Fp2 coefficient and scale the w^k coefficient by
// xi^(k (p^j - 1)/6)  (tables BN_FROB_G1/G2/G3; the p^2 constants lie in Fp)
BN_HD Fp2 frob_coeff(int j, int k) {
  const uint32_t(*t)[2][BN_NL] = j == 1 ? BN_FROB_G1 : j == 2 ? BN_FROB_G2 : BN_FROB_G3;
  return fp2_from_limbs(t[k][0], t[k][1]);
}
BN_HD Fp12 fp12_frob(const Fp12& a, int j) {
  // coefficient of w^k: k = 0 c0.c0, 1 c1.c0, 2 c0.c1, 3 c1.c1, 4 c0.c2, 5 c1.c2
  Fp12 r;
  const bool odd = (j & 1) != 0;
  Fp2 x0 = odd ? fp2_conj(a.c0.c0) : a.c0.c0, x1 = odd ? fp2_conj(a.c1.c0) : a.c1.c0, x2 = odd ? fp2_conj(a.c0.c1) : a.c0.c1;
  Fp2 x3 = odd ? fp2_conj(a.c1.c1) : a.c1.c1, x4 = odd ? fp2_conj(a.c0.c2) : a.c0.c2, x5 = odd ? fp2_conj(a.c1.c2) : a.c1.c2;
  r.c0.c0 = x0;
  if (j == 2) {
    r.c1.c0 = fp2_mul_fp(x1, frob_coeff(2, 1).c0); r.c0.c1 = fp2_mul_fp(x2, frob_coeff(2, 2).c0);
    r.c1.c1 = fp2_mul_fp(x3, frob_coeff(2, 3).c0); r.c0.c2 = fp2_mul_fp(x4, frob_coeff(2, 4).c0);
    r.c1.c2 = fp2_mul_fp(x5, frob_coeff(2, 5).c0);
  } else {
    r.c1.c0 = fp2_mul_nl(x1, frob_coeff(j, 1)); r.c0.c1 = fp2_mul_nl(x2, frob_coeff(j, 2));
    r.c1.c1 = fp2_mul_nl(x3, frob_coeff(j, 3)); r.c0.c2 = fp2_mul_nl(x4, frob_coeff(j, 4));
    r.c1.c2 = fp2_mul_nl(x5, frob_coeff(j, 5));
  }
  return r;
}
// Granger-Scott squaring, valid for elements of the cyclotomic subgroup (after the easy part): 9 Fp2 squarings
BN_HD Fp12 fp12_cyclo_sqr(const Fp12& x) {
  Fp2 t0 = fp2_sqr_nl(x.c1.c1), t1 = fp2_sqr_nl(x.c0.c0);
  Fp2 t6 = fp2_sub2(fp2_sqr_nl(fp2_add(x.c1.c1, x.c0.c0)), t0, t1);
  Fp2 t2 = fp2_sqr_nl(x.c0.c2), t3 = fp2_sqr_nl(x.c1.c0);
  Fp2 t7 = fp2_sub2(fp2_sqr_nl(fp2_add(x.c0.c2, x.c1.c0)), t2, t3);
  Fp2 t4 = fp2_sqr_nl(x.c1.c2), t5 = fp2_sqr_nl(x.c0.c1);
  Fp2 t8 = fp2_mul_xi(fp2_sub2(fp2_sqr_nl(fp2_add(x.c1.c2, x.c0.c1)), t4, t5));
  t0 = fp2_add(fp2_mul_xi(t0), t1);
  t2 = fp2_add(fp2_mul_xi(t2), t3);
  t4 = fp2_add(fp2_mul_xi(t4), t5);
  Fp12 z;
  // 3 t - 2 x  and  3 t + 2 x, reduced in one pass each
  z.c0.c0.c0 = fp_lincomb_reduce(3, t0.c0, -2, x.c0.c0.c0); z.c0.c0.c1 = fp_lincomb_reduce(3, t0.c1, -2, x.c0.c0.c1);
  z.c0.c1.c0 = fp_lincomb_reduce(3, t2.c0, -2, x.c0.c1.c0); z.c0.c1.c1 = fp_lincomb_reduce(3, t2.c1, -2, x.c0.c1.c1);
  z.c0.c2.c0 = fp_lincomb_reduce(3, t4.c0, -2, x.c0.c2.c0); z.c0.c2.c1 = fp_lincomb_reduce(3, t4.c1, -2, x.c0.c2.c1);
  z.c1.c0.c0 = fp_lincomb_reduce(3, t8.c0, 2, x.c1.c0.c0); z.c1.c0.c1 = fp_lincomb_reduce(3, t8.c1, 2, x.c1.c0.c1);
  z.c1.c1.c0 = fp_lincomb_reduce(3, t6.c0, 2, x.c1.c1.c0); z.c1.c1.c1 = fp_lincomb_reduce(3, t6.c1, 2, x.c1.c1.c1);
  z.c1.c2.c0 = fp_lincomb_reduce(3, t7.c0, 2, x.c1.c2.c0); z.c1.c2.c1 = fp_lincomb_reduce(3, t7.c1, 2, x.c1.c2.c1);
  return z;
}
BN_HD bool fp6_eq(const Fp6& a, const Fp6& b) { return fp2_eq(a.c0, b.c0) & fp2_eq(a.c1, b.c1) & fp2_eq(a.c2, b.c2); }
BN_HD bool fp12_eq(const Fp12& a, const Fp12& b) { return fp6_eq(a.c0, b.c0) & fp6_eq(a.c1, b.c1); }

}  // namespace bn254
