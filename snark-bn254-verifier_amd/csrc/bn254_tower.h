// bn254_tower.h -- Fp2 / Fp6 / Fp12 tower over bn254_fp.h.
//   Fp2 = Fp[i]/(i^2+1), Fp6 = Fp2[v]/(v^3 - xi), xi = 9 + i, Fp12 = Fp6[w]/(w^2 - v)   
// Device replacement for bn::{Fq2, Fq6, Fq12} (Fq2::{mul,inverse}, Fq6::{mul,squared,inverse,
// frobenius_map}, Fq12::{mul, squared, mul_by_024, cyclotomic_squared, frobenius_map, inverse}).
//
// Everything multiplicative is written as SUMS OF PRODUCTS reduced once (fp_dot): an Fp2 product is two dot products
// (no Karatsuba: on this machine three multiplications plus five additions cost more than four multiplications with
// two reductions, and need more registers), and the Fp12 operations of the Miller loop work on the six Fp2 coefficients of
//     f = k0 + k1 w + k2 w^2 + k3 w^3 + k4 w^4 + k5 w^5,   w^6 = xi,
// each output coefficient being ONE sum of up to four Fp2 products (= two fp_dot calls of up to 12 weighted terms).  No
// Fp6-sized temporaries exist, which is what keeps these kernels near the 256-VGPR budget of two waves per SIMD.
#pragma once
#include "bn254_fp.h"

namespace bn254 {

struct Fp2 { Fp c0, c1; };
struct Fp6 { Fp2 c0, c1, c2; };
struct Fp12 { Fp6 c0, c1; };  // storage order (tower): c0 = (k0, k2, k4), c1 = (k1, k3, k5) in the w-power numbering

// ------------------------------------------------------------------ Fp2
BN_HD Fp2 fp2_zero() { Fp2 r; r.c0 = fp_zero(); r.c1 = fp_zero(); return r; }
BN_HD Fp2 fp2_one() { Fp2 r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }
BN_HD Fp2 fp2_add(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_add(a.c0, b.c0); r.c1 = fp_add(a.c1, b.c1); return r; }
BN_HD Fp2 fp2_sub(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_sub(a.c0, b.c0); r.c1 = fp_sub(a.c1, b.c1); return r; }
BN_HD Fp2 fp2_add_lazy(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_add_lazy(a.c0, b.c0); r.c1 = fp_add_lazy(a.c1, b.c1); return r; }
BN_HD Fp2 fp2_sub_lazy(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_sub_lazy(a.c0, b.c0); r.c1 = fp_sub_lazy(a.c1, b.c1); return r; }
BN_HD Fp2 fp2_neg(const Fp2& a) { Fp2 r; r.c0 = fp_neg(a.c0); r.c1 = fp_neg(a.c1); return r; }
BN_HD Fp2 fp2_dbl(const Fp2& a) { Fp2 r; r.c0 = fp_dbl(a.c0); r.c1 = fp_dbl(a.c1); return r; }
BN_HD Fp2 fp2_conj(const Fp2& a) { Fp2 r; r.c0 = a.c0; r.c1 = fp_neg(a.c1); return r; }
BN_HD Fp2 fp2_reduce(const Fp2& a) { Fp2 r; r.c0 = fp_reduce(a.c0); r.c1 = fp_reduce(a.c1); return r; }
BN_HD Fp2 fp2_norm(const Fp2& a) { Fp2 r; r.c0 = fp_norm(a.c0); r.c1 = fp_norm(a.c1); return r; }
BN_HD Fp2 fp2_select(bool c, const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = fp_select(c, a.c0, b.c0); r.c1 = fp_select(c, a.c1, b.c1); return r; }
BN_HD Fp2 fp2_sub2(const Fp2& a, const Fp2& b, const Fp2& c) {
  Fp2 r; r.c0 = fp_norm(fp_sub_lazy(fp_sub_lazy(a.c0, b.c0), c.c0)); r.c1 = fp_norm(fp_sub_lazy(fp_sub_lazy(a.c1, b.c1), c.c1)); return r;
}

// weighted sum of Fp2 products: sum_t W_t x_t y_t, W in {+1,-1,+2,-2}; two dot products, no temporaries
template <int W>
struct P2 {
  const Fp2& x;
  const Fp2& y;
};
BN_HD P2<1> pp(const Fp2& x, const Fp2& y) { return P2<1>{x, y}; }
BN_HD P2<-1> pm(const Fp2& x, const Fp2& y) { return P2<-1>{x, y}; }
BN_HD P2<2> pp2(const Fp2& x, const Fp2& y) { return P2<2>{x, y}; }
BN_HD P2<-2> pm2(const Fp2& x, const Fp2& y) { return P2<-2>{x, y}; }
template <int... W>
BN_HD Fp2 fp2_dotp(const P2<W>&... t) {
  Fp2 r;
  r.c0 = fp_dot(dterm<W>(t.x.c0, t.y.c0)..., dterm<-W>(t.x.c1, t.y.c1)...);
  r.c1 = fp_dot(dterm<W>(t.x.c0, t.y.c1)..., dterm<W>(t.x.c1, t.y.c0)...);
  return r;
}

// ---- Karatsuba form of the same sum: re and im of sum_t W_t x_t y_t retired TOGETHER, column by column ------------------------
// Three 9x9 digit products per Fp2 product instead of four.  Per column and weight class |W| in {1, 2} two partial sums
//   A = sum x0 y0,   B = sum (-x1) y1        (x1 negated once, digit-wise: balanced digits need no carry for that)
// and, straight into the im accumulator, |W| (x0 + x1)(y0 + y1).  Then re += |W| (A + B), im += |W| (B - A).  The partial sums
// live in 64-bit accumulators that may wrap (two's complement): only the combined columns have to fit, and they are the same
// columns fp2_dotp forms.  A negative weight negates x.  Squares and Fp2 x Fp products go straight into re / im:
//   ksq(x):  re += (x0+x1)(x0-x1), im += (2 x0) x1      kfp(x, f):  re += x0 f, im += x1 f
struct KAcc { uint64_t re, im, a[2][2]; };
#define BN_KCOL_RANGE constexpr int LO = K < BN_NL ? 0 : K - (BN_NL - 1); constexpr int HI = K < BN_NL ? K : BN_NL - 1
BN_HD uint64_t bn_dmul(int32_t a, int32_t b) { return (uint64_t)((int64_t)a * (int64_t)b); }
template <int AW>  // |weight| 1 or 2
struct KProd {
  Fp x0, nx1, y0, y1, sxw, sy;
  static constexpr int cls = AW;
  template <int K> BN_HD void col(KAcc& c) const {
    BN_KCOL_RANGE;
#pragma unroll
    for (int i = LO; i <= HI; i++) {
      c.a[AW - 1][0] += bn_dmul(x0.v[i], y0.v[K - i]);
      c.a[AW - 1][1] += bn_dmul(nx1.v[i], y1.v[K - i]);
      c.im += bn_dmul(sxw.v[i], sy.v[K - i]);
    }
  }
#if BN_TRACKING
  double cap_re() const { return 9.0 * AW * (x0.lb * y0.lb + nx1.lb * y1.lb); }
  double cap_im() const { return 9.0 * AW * (x0.lb * y1.lb + nx1.lb * y0.lb); }
  double val_re() const { return AW * (x0.vb * y0.vb + nx1.vb * y1.vb); }
  double val_im() const { return AW * (x0.vb * y1.vb + nx1.vb * y0.vb); }
  void check() const { if (sxw.lb > 3.99 || sy.lb > 3.99) fp_dbg_fail("fp2_dotk: operand sum overflows a digit", sxw.lb + sy.lb); }
#endif
};
struct KDirect {  // re += u_re * v_re, im += u_im * v_im (plain products)
  Fp ure, vre, uim, vim;
  static constexpr int cls = 0;
  template <int K> BN_HD void col(KAcc& c) const {
    BN_KCOL_RANGE;
#pragma unroll
    for (int i = LO; i <= HI; i++) {
      c.re += bn_dmul(ure.v[i], vre.v[K - i]);
      c.im += bn_dmul(uim.v[i], vim.v[K - i]);
    }
  }
#if BN_TRACKING
  double cap_re() const { return 9.0 * ure.lb * vre.lb; }
  double cap_im() const { return 9.0 * uim.lb * vim.lb; }
  double val_re() const { return ure.vb * vre.vb; }
  double val_im() const { return uim.vb * vim.vb; }
  void check() const { if (ure.lb > 3.99 || vre.lb > 3.99 || uim.lb > 3.99 || vim.lb > 3.99) fp_dbg_fail("fp2_dotk: direct operand overflows a digit", ure.lb); }
#endif
};
template <int AW, bool NEG> BN_HD KProd<AW> kprod(const Fp2& x, const Fp2& y) {
  KProd<AW> t;
  t.x0 = NEG ? fp_neg(x.c0) : x.c0; t.nx1 = NEG ? x.c1 : fp_neg(x.c1); t.y0 = y.c0; t.y1 = y.c1;
  t.sy = fp_add_lazy(y.c0, y.c1);
#pragma unroll
  for (int i = 0; i < BN_NL; i++) t.sxw.v[i] = (NEG ? -AW : AW) * (x.c0.v[i] + x.c1.v[i]);
  BN_SETB(t.sxw, AW * (BN_VB(x.c0) + BN_VB(x.c1)), AW * (BN_LBD(x.c0) + BN_LBD(x.c1)));
  return t;
}
BN_HD KProd<1> kp(const Fp2& x, const Fp2& y) { return kprod<1, false>(x, y); }
BN_HD KProd<1> km(const Fp2& x, const Fp2& y) { return kprod<1, true>(x, y); }
BN_HD KProd<2> kp2(const Fp2& x, const Fp2& y) { return kprod<2, false>(x, y); }
BN_HD KProd<2> km2(const Fp2& x, const Fp2& y) { return kprod<2, true>(x, y); }
BN_HD KDirect ksq(const Fp2& x) {  // x^2
  KDirect t; t.ure = fp_add_lazy(x.c0, x.c1); t.vre = fp_sub_lazy(x.c0, x.c1); t.uim = fp_dbl_lazy(x.c0); t.vim = x.c1; return t;
}
BN_HD KDirect kfp(const Fp2& x, const Fp& f) { KDirect t; t.ure = x.c0; t.vre = f; t.uim = x.c1; t.vim = f; return t; }

template <int K, class... T>
BN_HD void dotk_step(int64_t& cre, int64_t& cim, int32_t (&mre)[BN_NL], int32_t (&mim)[BN_NL], Fp2& r, const T&... t) {
  constexpr bool has1 = ((T::cls == 1) || ...), has2 = ((T::cls == 2) || ...);
  KAcc c; c.re = (uint64_t)cre; c.im = (uint64_t)cim;
  c.a[0][0] = c.a[0][1] = c.a[1][0] = c.a[1][1] = 0;
  (t.template col<K>(c), ...);
  if constexpr (K < BN_NL) {
#pragma unroll
    for (int i = 0; i < K; i++) { c.re += bn_dmul(mre[i], bn_p_limb(K - i)); c.im += bn_dmul(mim[i], bn_p_limb(K - i)); }
  } else {
#pragma unroll
    for (int i = K - (BN_NL - 1); i < BN_NL; i++) { c.re += bn_dmul(mre[i], bn_p_limb(K - i)); c.im += bn_dmul(mim[i], bn_p_limb(K - i)); }
  }
  if (has1) { c.re += c.a[0][0] + c.a[0][1]; c.im += c.a[0][1] - c.a[0][0]; }
  if (has2) { c.re += (c.a[1][0] + c.a[1][1]) << 1; c.im += (c.a[1][1] - c.a[1][0]) << 1; }
  if constexpr (K < BN_NL) {
    mre[K] = bn_sext29((uint32_t)c.re * BN_PINV);
    mim[K] = bn_sext29((uint32_t)c.im * BN_PINV);
    c.re += bn_dmul(mre[K], bn_p_limb(0));
    c.im += bn_dmul(mim[K], bn_p_limb(0));
    cre = (int64_t)c.re >> BN_LB;  // exact: the low 29 bits are zero
    cim = (int64_t)c.im >> BN_LB;
  } else {
    c.re += BN_HALF; c.im += BN_HALF;
    r.c0.v[K - BN_NL] = (int32_t)((uint32_t)c.re & BN_MASK) - BN_HALF;
    r.c1.v[K - BN_NL] = (int32_t)((uint32_t)c.im & BN_MASK) - BN_HALF;
    cre = (int64_t)c.re >> BN_LB;
    cim = (int64_t)c.im >> BN_LB;
  }
}
template <int... KS, class... T>
BN_HD void dotk_all(std::integer_sequence<int, KS...>, int64_t& cre, int64_t& cim, int32_t (&mre)[BN_NL], int32_t (&mim)[BN_NL], Fp2& r, const T&... t) {
  (dotk_step<KS, T...>(cre, cim, mre, mim, r, t...), ...);
}
template <class... T>
BN_HD Fp2 fp2_dotk(const T&... t) {
  BN_SCHED_FENCE();
  int64_t cre = 0, cim = 0;
  int32_t mre[BN_NL], mim[BN_NL];
  Fp2 r;
  dotk_all(std::make_integer_sequence<int, 2 * BN_NL - 1>{}, cre, cim, mre, mim, r, t...);
  r.c0.v[BN_NL - 1] = (int32_t)cre;
  r.c1.v[BN_NL - 1] = (int32_t)cim;
  BN_SCHED_FENCE();
#if BN_TRACKING
  (t.check(), ...);
  double cr = (t.cap_re() + ...), ci = (t.cap_im() + ...);
  if (cr + 9.0 * 0.25 + 0.01 > 32.0 || ci + 9.0 * 0.25 + 0.01 > 32.0) fp_dbg_fail("fp2_dotk: accumulator may overflow", cr > ci ? cr : ci);
  BN_SETB(r.c0, 1.0 + (t.val_re() + ...) / 169.0 + 1e-6, 0.5);
  BN_SETB(r.c1, 1.0 + (t.val_im() + ...) / 169.0 + 1e-6, 0.5);
#endif
  return r;
}
BN_HD Fp2 fp2_mul(const Fp2& a, const Fp2& b) { return fp2_dotp(pp(a, b)); }
// complex squaring: (a0+a1)(a0-a1), 2 a0 a1: two single products with lazy operand sums
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
BN_HD Fp2 fp2_lincomb_reduce(int32_t k1, const Fp2& a, int32_t k2, const Fp2& b) {  // k1 a + k2 b, reduced
  Fp2 r;
  r.c0 = fp_lincomb_reduce(k1, a.c0, k2, b.c0);
  r.c1 = fp_lincomb_reduce(k1, a.c1, k2, b.c1);
  return r;
}
BN_HD Fp2 fp2_inv(const Fp2& a) {  // 0 -> 0
  Fp n = fp_dot(dplus(a.c0, a.c0), dplus(a.c1, a.c1));
  Fp ni = fp_inv(n);
  Fp2 r;
  r.c0 = fp_mul(a.c0, ni);
  r.c1 = fp_neg(fp_mul(a.c1, ni));
  return r;
}
BN_HD bool fp2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) & fp_is_zero(a.c1); }
BN_HD bool fp2_eq(const Fp2& a, const Fp2& b) { return fp_eq(a.c0, b.c0) & fp_eq(a.c1, b.c1); }
BN_HD Fp2 fp2_from_limbs(const int32_t* c0, const int32_t* c1) { Fp2 r; r.c0 = fp_from_limbs(c0); r.c1 = fp_from_limbs(c1); return r; }

// out-of-line copies for code that is not on the hot path
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
// (x0 + x1 v + x2 v^2)(y0 + y1 v + y2 v^2), v^3 = xi: three sums of three Fp2 products, xi folded into y1, y2 up front
BN_HD Fp6 fp6_mul(const Fp6& x, const Fp6& y) {
  // Karatsuba form of the Fp2 products (fp2_dotk): 27 digit products instead of 36; with 144 resident operand registers this is
  // the one place where the operand sums still fit (237 VGPRs in k_f12_mul, no scratch, -6.6 % time:
  // profiles/r01_kbench_memory_exposure.txt).  Operands must have normalised digits.
  Fp2 Y1 = fp2_mul_xi(y.c1), Y2 = fp2_mul_xi(y.c2);
  Fp6 r;
  r.c0 = fp2_dotk(kp(x.c0, y.c0), kp(x.c1, Y2), kp(x.c2, Y1));
  r.c1 = fp2_dotk(kp(x.c0, y.c1), kp(x.c1, y.c0), kp(x.c2, Y2));
  r.c2 = fp2_dotk(kp(x.c0, y.c2), kp(x.c1, y.c1), kp(x.c2, y.c0));
  return r;
}
// the same product with plain Fp2 dot products (36 digit products): for callers whose other live values leave no room for the
// operand sums (k_f12_inv)
BN_HD Fp6 fp6_mul_plain(const Fp6& x, const Fp6& y) {
  Fp2 Y1 = fp2_mul_xi(y.c1), Y2 = fp2_mul_xi(y.c2);
  Fp6 r;
  r.c0 = fp2_dotp(pp(x.c0, y.c0), pp(x.c1, Y2), pp(x.c2, Y1));
  r.c1 = fp2_dotp(pp(x.c0, y.c1), pp(x.c1, y.c0), pp(x.c2, Y2));
  r.c2 = fp2_dotp(pp(x.c0, y.c2), pp(x.c1, y.c1), pp(x.c2, y.c0));
  return r;
}
BN_HD Fp6 fp6_sqr(const Fp6& x) {
  Fp2 X2 = fp2_mul_xi(x.c2);
  Fp6 r;
  r.c0 = fp2_dotp(pp(x.c0, x.c0), pp2(x.c1, X2));
  r.c1 = fp2_dotp(pp2(x.c0, x.c1), pp(x.c2, X2));
  r.c2 = fp2_dotp(pp2(x.c0, x.c2), pp(x.c1, x.c1));
  return r;
}
BN_HD Fp6 fp6_mul_fp2(const Fp6& a, const Fp2& b) { Fp6 r; r.c0 = fp2_mul(a.c0, b); r.c1 = fp2_mul(a.c1, b); r.c2 = fp2_mul(a.c2, b); return r; }
BN_HD_NOINLINE Fp6 fp6_mul_nl(const Fp6& a, const Fp6& b) { return fp6_mul(a, b); }
BN_HD Fp6 fp6_inv(const Fp6& a) {
  // A = a0^2 - xi a1 a2, B = xi a2^2 - a0 a1, C = a1^2 - a0 a2, F = a0 A + xi (a2 B + a1 C); 1/a = (A, B, C)/F
  Fp2 X1 = fp2_mul_xi(a.c1), X2 = fp2_mul_xi(a.c2);
  Fp2 A = fp2_dotp(pp(a.c0, a.c0), pm(X1, a.c2));
  Fp2 B = fp2_dotp(pp(X2, a.c2), pm(a.c0, a.c1));
  Fp2 C = fp2_dotp(pp(a.c1, a.c1), pm(a.c0, a.c2));
  Fp2 F = fp2_dotp(pp(a.c0, A), pp(X2, B), pp(X1, C));
  Fp2 Fi = fp2_inv(F);
  Fp6 r;
  r.c0 = fp2_mul(A, Fi); r.c1 = fp2_mul(B, Fi); r.c2 = fp2_mul(C, Fi);
  return r;
}

// ------------------------------------------------------------------ Fp12
BN_HD Fp12 fp12_one() { Fp12 r; r.c0 = fp6_one(); r.c1 = fp6_zero(); return r; }
BN_HD Fp12 fp12_conj(const Fp12& a) { Fp12 r; r.c0 = a.c0; r.c1 = fp6_neg(a.c1); return r; }
BN_HD Fp12 fp12_reduce(const Fp12& a) { Fp12 r; r.c0 = fp6_reduce(a.c0); r.c1 = fp6_reduce(a.c1); return r; }
BN_HD Fp12 fp12_select(bool c, const Fp12& a, const Fp12& b) { Fp12 r; r.c0 = fp6_select(c, a.c0, b.c0); r.c1 = fp6_select(c, a.c1, b.c1); return r; }
// w-power view: f = k0 + k1 w + ... + k5 w^5
#define K0(f) ((f).c0.c0)
#define K1(f) ((f).c1.c0)
#define K2(f) ((f).c0.c1)
#define K3(f) ((f).c1.c1)
#define K4(f) ((f).c0.c2)
#define K5(f) ((f).c1.c2)

// general product, Karatsuba over Fp6 (3 Fp6 products of 6 dot products each); used by the final exponentiation only
BN_HD Fp12 fp12_mul(const Fp12& a, const Fp12& b) {
  Fp6 v0 = fp6_mul_nl(a.c0, b.c0);
  Fp6 v1 = fp6_mul_nl(a.c1, b.c1);
  Fp6 s = fp6_mul_nl(fp6_add(a.c0, a.c1), fp6_add(b.c0, b.c1));
  Fp12 r;
  // c0 = v0 + v * v1,  c1 = s - v0 - v1
  Fp2 x = fp2_mul_xi(v1.c2);
  r.c0.c0 = fp2_add(v0.c0, x); r.c0.c1 = fp2_add(v0.c1, v1.c0); r.c0.c2 = fp2_add(v0.c2, v1.c1);
  r.c1.c0 = fp2_sub2(s.c0, v0.c0, v1.c0); r.c1.c1 = fp2_sub2(s.c1, v0.c1, v1.c1); r.c1.c2 = fp2_sub2(s.c2, v0.c2, v1.c2);
  return r;
}
// squaring, coefficient-wise: 12 dot products over 21 distinct Fp2 products (doubled cross terms carry weight 2)
//   r0 = k0^2 + xi (2 k1 k5 + 2 k2 k4 + k3^2)      r1 = 2 k0 k1 + xi (2 k2 k5 + 2 k3 k4)
//   r2 = 2 k0 k2 + k1^2 + xi (2 k3 k5 + k4^2)      r3 = 2 k0 k3 + 2 k1 k2 + xi (2 k4 k5)
//   r4 = 2 k0 k4 + 2 k1 k3 + k2^2 + xi k5^2        r5 = 2 k0 k5 + 2 k1 k4 + 2 k2 k3
BN_HD Fp12 fp12_sqr(const Fp12& f) {
  Fp2 x3 = fp2_mul_xi(K3(f)), x4 = fp2_mul_xi(K4(f)), x5 = fp2_mul_xi(K5(f));
  Fp12 r;
  K0(r) = fp2_dotp(pp(K0(f), K0(f)), pp2(K1(f), x5), pp2(K2(f), x4), pp(K3(f), x3));
  K1(r) = fp2_dotp(pp2(K0(f), K1(f)), pp2(K2(f), x5), pp2(K3(f), x4));
  K2(r) = fp2_dotp(pp2(K0(f), K2(f)), pp(K1(f), K1(f)), pp2(K3(f), x5), pp(K4(f), x4));
  K3(r) = fp2_dotp(pp2(K0(f), K3(f)), pp2(K1(f), K2(f)), pp2(K4(f), x5));
  K4(r) = fp2_dotp(pp2(K0(f), K4(f)), pp2(K1(f), K3(f)), pp(K2(f), K2(f)), pp(K5(f), x5));
  K5(r) = fp2_dotp(pp2(K0(f), K5(f)), pp2(K1(f), K4(f)), pp2(K2(f), K3(f)));
  return r;
}
// f * (d0 + d3 w + d4 w^3): the line value of a projective (variable-Q) Miller step, all three coefficients in Fp2
//   r0 = d0 k0 + xi d3 k5 + xi d4 k3    r1 = d0 k1 + d3 k0 + xi d4 k4    r2 = d0 k2 + d3 k1 + xi d4 k5
//   r3 = d0 k3 + d3 k2 + d4 k0          r4 = d0 k4 + d3 k3 + d4 k1       r5 = d0 k5 + d3 k4 + d4 k2
BN_HD Fp12 fp12_mul_by_034(const Fp12& f, const Fp2& d0, const Fp2& d3, const Fp2& d4) {
  Fp2 x3 = fp2_mul_xi(d3), x4 = fp2_mul_xi(d4);
  Fp12 r;
  K0(r) = fp2_dotp(pp(d0, K0(f)), pp(x3, K5(f)), pp(x4, K3(f)));
  K1(r) = fp2_dotp(pp(d0, K1(f)), pp(d3, K0(f)), pp(x4, K4(f)));
  K2(r) = fp2_dotp(pp(d0, K2(f)), pp(d3, K1(f)), pp(x4, K5(f)));
  K3(r) = fp2_dotp(pp(d0, K3(f)), pp(d3, K2(f)), pp(d4, K0(f)));
  K4(r) = fp2_dotp(pp(d0, K4(f)), pp(d3, K3(f)), pp(d4, K1(f)));
  K5(r) = fp2_dotp(pp(d0, K5(f)), pp(d3, K4(f)), pp(d4, K2(f)));
  return r;
}
// same with d0 in Fp (affine precomputed line, constant term y_P); x4 = xi * d4 comes precomputed with the line table
BN_HD Fp2 fp2_dot_line(const Fp& d0, const Fp2& k, const Fp2& da, const Fp2& ka, const Fp2& db, const Fp2& kb) {
  Fp2 r;  // d0 k + da ka + db kb
  r.c0 = fp_dot(dplus(d0, k.c0), dplus(da.c0, ka.c0), dminus(da.c1, ka.c1), dplus(db.c0, kb.c0), dminus(db.c1, kb.c1));
  r.c1 = fp_dot(dplus(d0, k.c1), dplus(da.c0, ka.c1), dplus(da.c1, ka.c0), dplus(db.c0, kb.c1), dplus(db.c1, kb.c0));
  return r;
}
BN_HD Fp12 fp12_mul_by_034_fp(const Fp12& f, const Fp& d0, const Fp2& d3, const Fp2& d4, const Fp2& x4) {
  Fp2 x3 = fp2_mul_xi(d3);
  Fp12 r;
  K0(r) = fp2_dot_line(d0, K0(f), x3, K5(f), x4, K3(f));
  K1(r) = fp2_dot_line(d0, K1(f), d3, K0(f), x4, K4(f));
  K2(r) = fp2_dot_line(d0, K2(f), d3, K1(f), x4, K5(f));
  K3(r) = fp2_dot_line(d0, K3(f), d3, K2(f), d4, K0(f));
  K4(r) = fp2_dot_line(d0, K4(f), d3, K3(f), d4, K1(f));
  K5(r) = fp2_dot_line(d0, K5(f), d3, K4(f), d4, K2(f));
  return r;
}
BN_HD Fp12 fp12_inv(const Fp12& a) {
  // 1/(a0 + a1 w) = (a0 - a1 w) / (a0^2 - v a1^2)
  Fp6 s0 = fp6_sqr(a.c0), s1 = fp6_mul_v(fp6_sqr(a.c1));
  Fp6 di = fp6_inv(fp6_sub(s0, s1));
  Fp12 r;
  r.c0 = fp6_mul_nl(a.c0, di);
  r.c1 = fp6_neg(fp6_mul_nl(a.c1, di));
  return r;
}
// Frobenius x -> x^(p^j), j = 1, 2, 3: conjugate (j odd) every Fp2 coefficient and scale the w^k coefficient by
// xi^(k (p^j - 1)/6)  (tables BN_FROB_G1/G2/G3; the p^2 constants lie in Fp)
BN_HD Fp2 frob_coeff(int j, int k) {
  const int32_t(*t)[2][BN_NL] = j == 1 ? BN_FROB_G1 : j == 2 ? BN_FROB_G2 : BN_FROB_G3;
  return fp2_from_limbs(t[k][0], t[k][1]);
}
BN_HD Fp12 fp12_frob(const Fp12& a, int j) {
  Fp12 r;
  const bool odd = (j & 1) != 0;
  Fp2 x1 = odd ? fp2_conj(K1(a)) : K1(a), x2 = odd ? fp2_conj(K2(a)) : K2(a), x3 = odd ? fp2_conj(K3(a)) : K3(a);
  Fp2 x4 = odd ? fp2_conj(K4(a)) : K4(a), x5 = odd ? fp2_conj(K5(a)) : K5(a);
  K0(r) = odd ? fp2_conj(K0(a)) : K0(a);
  if (j == 2) {
    K1(r) = fp2_mul_fp(x1, frob_coeff(2, 1).c0); K2(r) = fp2_mul_fp(x2, frob_coeff(2, 2).c0);
    K3(r) = fp2_mul_fp(x3, frob_coeff(2, 3).c0); K4(r) = fp2_mul_fp(x4, frob_coeff(2, 4).c0);
    K5(r) = fp2_mul_fp(x5, frob_coeff(2, 5).c0);
  } else {
    K1(r) = fp2_mul_nl(x1, frob_coeff(j, 1)); K2(r) = fp2_mul_nl(x2, frob_coeff(j, 2));
    K3(r) = fp2_mul_nl(x3, frob_coeff(j, 3)); K4(r) = fp2_mul_nl(x4, frob_coeff(j, 4));
    K5(r) = fp2_mul_nl(x5, frob_coeff(j, 5));
  }
  return r;
}
// Granger-Scott squaring on the cyclotomic subgroup, three independent pairs of coefficients (a, b):
//   S = xi b^2 + a^2,  T = 2 a b   ->   za = 3 S - 2 sub,   zb = 3 T + 2 add   (T multiplied by xi for the third pair)
//   (a, b; sub, add) = (c0.c0, c1.c1; c0.c0, c1.c1), (c1.c0, c0.c2; c0.c1, c1.c2), (c0.c1, c1.c2; c0.c2, c1.c0)
// each bracket is one sum of products; the linear part is folded in by a reducing linear combination.
BN_HD void gs_pair(Fp2& za, Fp2& zb, const Fp2& a, const Fp2& b, const Fp2& sub, const Fp2& add, bool xi_on_cross) {
  Fp2 xb = fp2_mul_xi(b);
  Fp2 S = fp2_dotp(pp(xb, b), pp(a, a));
  Fp2 T = xi_on_cross ? fp2_dotp(pp2(a, xb)) : fp2_dotp(pp2(a, b));
  za = fp2_lincomb_reduce(3, S, -2, sub);
  zb = fp2_lincomb_reduce(3, T, 2, add);
}
BN_HD Fp12 fp12_cyclo_sqr(const Fp12& x) {
  Fp12 z;
  gs_pair(z.c0.c0, z.c1.c1, x.c0.c0, x.c1.c1, x.c0.c0, x.c1.c1, false);
  gs_pair(z.c0.c1, z.c1.c2, x.c1.c0, x.c0.c2, x.c0.c1, x.c1.c2, false);
  gs_pair(z.c0.c2, z.c1.c0, x.c0.c1, x.c1.c2, x.c0.c2, x.c1.c0, true);
  return z;
}
BN_HD bool fp6_eq(const Fp6& a, const Fp6& b) { return fp2_eq(a.c0, b.c0) & fp2_eq(a.c1, b.c1) & fp2_eq(a.c2, b.c2); }
BN_HD bool fp12_eq(const Fp12& a, const Fp12& b) { return fp6_eq(a.c0, b.c0) & fp6_eq(a.c1, b.c1); }

}  // namespace bn254
