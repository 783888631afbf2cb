// bn254_fp.h -- BN254 base-field arithmetic for gfx950 (and, compiled unchanged, for the host).
//
// Representation (DESIGN.md "Fp on the VALU"): an element is 9 signed 32-bit limbs holding BALANCED 29-bit digits,
//     value = sum_i v[i] * 2^(29 i),   v[i] in [-2^28, 2^28) once normalised,
// in Montgomery form with R = 2^261.  gfx950 has no 64x64 multiplier; its widest integer multiply is
// v_mad_i64_i32 (32x32 + 64 -> 64), measured at ~1.9 ns per wave-instruction per SIMD, less than two plain
// v_add_u32 (profiles/r01_ubench_valu.txt).  So the design rule is: spend mads, avoid carry/add instructions.
//   * 29-bit digits: a whole column of a schoolbook product accumulates in ONE 64-bit register pair, no carries.
//   * balanced digits: |a_i b_j| <= 2^56, so a column has room for 2^63 / 2^56 = 128 partial products.  That is
//     enough to accumulate a SUM OF UP TO 13 PRODUCTS  sum_t a_t * b_t  and reduce it ONCE (fp_dot): every Fp2 /
//     Fp6 / Fp12 product becomes a handful of dot products with no intermediate additions, subtractions,
//     normalisations or temporaries (lazy reduction in the sense of Aranha et al., here at the column level).
//
// Values are signed and loosely bounded: |value| <= vb * p, tracked by the host-side bound tracker (BN_TRACK_BOUNDS,
// tests/hostsim) which asserts on every operation that (1) the value stays representable (|v| < 2^260 ~ 84 p),
// (2) the 64-bit column accumulators cannot overflow, (3) the tracked bound dominates the actual value.  Because
// R / p ~ 169, a reduced sum of products comes back in (-S/169, 1 + S/169) * p with S = sum vb_a vb_b: multiplication
// contracts bounds, so no conditional subtraction exists anywhere; fp_reduce()/fp_lincomb_reduce() bring linear
// combinations back to (-eps, 1 + eps) * p in one carry pass (quotient estimated from the top digit).
//
// Replaces (behaviourally) bn::Fq of substrate-bn, used by the reference at verifier/src/converter.rs:85-86,
// 144-147 (Fq::from_slice) and everywhere below bn::pairing_batch (groth16/verify.rs:70-77).
#pragma once
#include <stdint.h>
#include <utility>
#include "bn254_constants.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#if defined(__HIP_DEVICE_COMPILE__)
#define BN_HD __host__ __device__ __forceinline__
#else
#define BN_HD __host__ __device__ inline   // host pass: let the compiler decide (forced inlining makes the host build 4x slower)
#endif
#if defined(BN_INLINE_ALL)
#define BN_HD_NOINLINE __host__ __device__ __forceinline__
#else
#define BN_HD_NOINLINE __host__ __device__ inline __attribute__((noinline))
#endif
// out of line on the host (compile time), inline on the device: a device call passes its Fp arguments through scratch memory
#if defined(__HIP_DEVICE_COMPILE__)
#define BN_HD_DEVINLINE __host__ __device__ __forceinline__
#else
#define BN_HD_DEVINLINE __host__ __device__ inline __attribute__((noinline))
#endif
#else
#if defined(BN_TRACK_BOUNDS) || defined(BN_HOST_PLAIN_INLINE)   // (sanitizer builds of the host half, tests/hostsan: forced inlining makes them take an hour)
#define BN_HD inline
#else
#define BN_HD inline __attribute__((always_inline))
#endif
#define BN_HD_NOINLINE inline __attribute__((noinline))
#define BN_HD_DEVINLINE inline __attribute__((noinline))
#endif

// scheduling fence: stops the machine scheduler from interleaving independent field multiplications (which
// multiplies live ranges and forces spills); device only, no instruction emitted
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BN_NO_SCHED_FENCE)
#define BN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define BN_SCHED_FENCE() do { } while (0)
#endif

#if defined(BN_TRACK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <execinfo.h>
#define BN_TRACKING 1
#else
#define BN_TRACKING 0
#endif

namespace bn254 {

#define BN_HALF (1 << (BN_LB - 1))  // 2^28

struct Fp {
  int32_t v[BN_NL];
#if BN_TRACKING
  double vb;  // |value| <= vb * p
  double lb;  // max |digit| <= lb * 2^29   (0.5 when normalised)
#endif
};

#if BN_TRACKING
static const long double BN_P_LD = 21888242871839275222246405745257275088696311157297823662689037894645226208583.0L;
inline long double fp_dbg_value(const Fp& a) {
  long double s = 0;
  for (int i = BN_NL - 1; i >= 0; i--) s = s * 536870912.0L + (long double)a.v[i];
  return s;
}
inline void fp_dbg_fail(const char* what, double x) {
  fprintf(stderr, "bn254 bound violation: %s (%g)\n", what, x);
  void* bt[32]; int n = backtrace(bt, 32); backtrace_symbols_fd(bt, n, 2);
  abort();
}
inline void fp_dbg_check(const Fp& a, const char* where) {
  long double val = fabsl(fp_dbg_value(a)) / BN_P_LD;
  if (val > (long double)a.vb * (1 + 1e-12L) + 1e-9L) fp_dbg_fail(where, (double)val);
  if (a.vb > 80.0) fp_dbg_fail("value bound exceeds representable range (|v| < 2^260)", a.vb);
  for (int i = 0; i < BN_NL; i++)
    if (fabs((double)a.v[i]) > a.lb * 536870912.0 + 0.5) fp_dbg_fail("digit bound wrong", (double)a.v[i]);
}
#define BN_SETB(x, VB, LB) do { (x).vb = (VB); (x).lb = (LB); fp_dbg_check((x), __func__); } while (0)
#define BN_VB(x) ((x).vb)
#define BN_LBD(x) ((x).lb)
#else
#define BN_SETB(x, VB, LB) do { } while (0)
#define BN_VB(x) 0.0
#define BN_LBD(x) 0.0
#endif

BN_HD int32_t bn_p_limb(int i) {
  // compile-time constants after unrolling: SGPR / literal operands, never memory
  switch (i) {
    case 0: return BN_P0; case 1: return BN_P1; case 2: return BN_P2; case 3: return BN_P3; case 4: return BN_P4;
    case 5: return BN_P5; case 6: return BN_P6; case 7: return BN_P7; default: return BN_P8;
  }
}
BN_HD int32_t bn_sext29(uint32_t x) { return (int32_t)(x << 3) >> 3; }  // low 29 bits as a balanced digit

BN_HD Fp fp_from_limbs(const int32_t* c) {  // a precomputed constant (balanced digits, value in [0, p))
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = c[i];
  BN_SETB(r, 1.0, 0.5);
  return r;
}
BN_HD Fp fp_zero() {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = 0;
  BN_SETB(r, 0.0, 0.5);
  return r;
}
BN_HD Fp fp_one() { return fp_from_limbs(BN_ONE); }

// ---- carry propagation back to balanced digits ----------------------------------------------------------------------
BN_HD Fp fp_norm(const Fp& a) {
  Fp r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < BN_NL - 1; i++) {
    int32_t t = a.v[i] + c + BN_HALF;
    r.v[i] = (t & (int32_t)BN_MASK) - BN_HALF;
    c = t >> BN_LB;  // arithmetic shift = floor
  }
  r.v[BN_NL - 1] = a.v[BN_NL - 1] + c;
#if BN_TRACKING
  if (a.lb > 3.4) fp_dbg_fail("fp_norm: digit overflow", a.lb);
  BN_SETB(r, a.vb, 0.5);
#endif
  return r;
}

// ---- lazy (digit-wise, no carry) add / sub / neg -------------------------------------------------------------------------
BN_HD Fp fp_add_lazy(const Fp& a, const Fp& b) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = a.v[i] + b.v[i];
  BN_SETB(r, BN_VB(a) + BN_VB(b), BN_LBD(a) + BN_LBD(b));
  return r;
}
BN_HD Fp fp_sub_lazy(const Fp& a, const Fp& b) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = a.v[i] - b.v[i];
  BN_SETB(r, BN_VB(a) + BN_VB(b), BN_LBD(a) + BN_LBD(b));
  return r;
}
BN_HD Fp fp_neg(const Fp& a) {  // balanced digits negate digit-wise: no carry pass needed
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = -a.v[i];
  BN_SETB(r, BN_VB(a), BN_LBD(a) + 1e-9);
  return r;
}
BN_HD Fp fp_add(const Fp& a, const Fp& b) { return fp_norm(fp_add_lazy(a, b)); }
BN_HD Fp fp_sub(const Fp& a, const Fp& b) { return fp_norm(fp_sub_lazy(a, b)); }
BN_HD Fp fp_dbl(const Fp& a) { return fp_add(a, a); }
BN_HD Fp fp_dbl_lazy(const Fp& a) { return fp_add_lazy(a, a); }

// ---- small linear combinations in one carry pass -----------------------------------------------------------------------------
BN_HD Fp fp_lincomb(int32_t k1, const Fp& a, int32_t k2, const Fp& b) {
  Fp r;
  int64_t acc = 0;
#pragma unroll
  for (int i = 0; i < BN_NL - 1; i++) {
    acc += (int64_t)k1 * (int64_t)a.v[i] + (int64_t)k2 * (int64_t)b.v[i] + BN_HALF;
    r.v[i] = (int32_t)((uint32_t)acc & BN_MASK) - BN_HALF;
    acc >>= BN_LB;
  }
  acc += (int64_t)k1 * (int64_t)a.v[BN_NL - 1] + (int64_t)k2 * (int64_t)b.v[BN_NL - 1];
  r.v[BN_NL - 1] = (int32_t)acc;
#if BN_TRACKING
  BN_SETB(r, a.vb * (k1 < 0 ? -k1 : k1) + b.vb * (k2 < 0 ? -k2 : k2), 0.5);
#endif
  return r;
}
// reduce(k1*a + k2*b): q = floor(top * C / 2^52), C = floor(2^284 / p), estimates (k1 a + k2 b) / p from the top digits alone
// (the lower digits move the quotient by < (|k1| + |k2|) * 2^232 / p < 1e-5); result in (-1e-5, 1 + 1e-5) * p.  The
// combination never has to be representable (64-bit pass), so any input bounds are fine.
BN_HD Fp fp_lincomb_reduce(int32_t k1, const Fp& a, int32_t k2, const Fp& b) {
  const int64_t C = 1420063842;
  int64_t top = (int64_t)k1 * (int64_t)a.v[BN_NL - 1] + (int64_t)k2 * (int64_t)b.v[BN_NL - 1];
  int32_t q = (int32_t)((top * C) >> 52);
  Fp r;
  int64_t acc = 0;
#pragma unroll
  for (int i = 0; i < BN_NL - 1; i++) {
    acc += (int64_t)k1 * (int64_t)a.v[i] + (int64_t)k2 * (int64_t)b.v[i] - (int64_t)q * (int64_t)bn_p_limb(i) + BN_HALF;
    r.v[i] = (int32_t)((uint32_t)acc & BN_MASK) - BN_HALF;
    acc >>= BN_LB;
  }
  acc += top - (int64_t)q * (int64_t)bn_p_limb(BN_NL - 1);
  r.v[BN_NL - 1] = (int32_t)acc;
#if BN_TRACKING
  if (a.vb * (k1 < 0 ? -k1 : k1) + b.vb * (k2 < 0 ? -k2 : k2) > 1e5) fp_dbg_fail("fp_lincomb_reduce: out of range", a.vb + b.vb);
  if (a.lb > 1.01 || b.lb > 1.01) fp_dbg_fail("fp_lincomb_reduce: digits too large", a.lb + b.lb);
  BN_SETB(r, 1.0 + 3e-5, 0.5);
#endif
  return r;
}
BN_HD Fp fp_reduce(const Fp& a) { return fp_lincomb_reduce(1, a, 0, a); }

// ---- sum-of-products Montgomery reduction ----------------------------------------------------------------------------------
// fp_dot(T...) = (sum_t sign_t * a_t * b_t) / R  mod p, reduced once.  Finely integrated product scanning: for each of the
// 17 columns the partial products of ALL terms go into one 64-bit accumulator (a second one collects the terms with a minus
// sign), then one Montgomery digit is retired.  Head-room: sum_t 9 lb_a lb_b 2^58 + 9 * 2^56 < 2^63.
template <int SIGN>  // +1, -1, +2, -2: weight of the product in the sum
struct DotTerm {
  const Fp& a;
  const Fp& b;
};
BN_HD DotTerm<1> dplus(const Fp& a, const Fp& b) { return DotTerm<1>{a, b}; }
BN_HD DotTerm<-1> dminus(const Fp& a, const Fp& b) { return DotTerm<-1>{a, b}; }
template <int S> BN_HD DotTerm<S> dterm(const Fp& a, const Fp& b) { return DotTerm<S>{a, b}; }

struct DotAcc { int64_t p1, m1, p2, m2; };  // +1, -1, +2, -2 weighted partial sums of one column
// column k of one term: sum_{i=LO..HI} a_i b_{k-i}; LO, HI are compile-time so that everything unrolls into straight mads
template <int SIGN, int K>
BN_HD void dot_col(DotAcc& c, const DotTerm<SIGN>& t) {
  constexpr int LO = K < BN_NL ? 0 : K - (BN_NL - 1);
  constexpr int HI = K < BN_NL ? K : BN_NL - 1;
#pragma unroll
  for (int i = LO; i <= HI; i++) {
    int64_t pr = (int64_t)t.a.v[i] * (int64_t)t.b.v[K - i];
    if (SIGN == 1) c.p1 += pr; else if (SIGN == -1) c.m1 += pr; else if (SIGN == 2) c.p2 += pr; else c.m2 += pr;
  }
}
constexpr int bn_iabs(int x) { return x < 0 ? -x : x; }
template <int K, int... SIGNS>
BN_HD void dot_step(int64_t& acc, int32_t (&m)[BN_NL], Fp& r, const DotTerm<SIGNS>&... terms) {
  constexpr bool has_m1 = ((SIGNS == -1) || ...), has_p2 = ((SIGNS == 2) || ...), has_m2 = ((SIGNS == -2) || ...);
  DotAcc c; c.p1 = acc; c.m1 = 0; c.p2 = 0; c.m2 = 0;
  (dot_col<SIGNS, K>(c, terms), ...);
  acc = c.p1;
  if (has_m1) acc -= c.m1;
  if (has_p2 && has_m2) acc += 2 * (c.p2 - c.m2);
  else if (has_p2) acc += 2 * c.p2;
  else if (has_m2) acc -= 2 * c.m2;
  if constexpr (K < BN_NL) {
#pragma unroll
    for (int i = 0; i < K; i++) acc += (int64_t)m[i] * (int64_t)bn_p_limb(K - i);
    m[K] = bn_sext29((uint32_t)acc * BN_PINV);
    acc += (int64_t)m[K] * (int64_t)bn_p_limb(0);
    acc >>= BN_LB;  // exact: the low 29 bits are zero
  } else {
#pragma unroll
    for (int i = K - (BN_NL - 1); i < BN_NL; i++) acc += (int64_t)m[i] * (int64_t)bn_p_limb(K - i);
    acc += BN_HALF;
    r.v[K - BN_NL] = (int32_t)((uint32_t)acc & BN_MASK) - BN_HALF;
    acc >>= BN_LB;
  }
}
template <int... KS, int... SIGNS>
BN_HD void dot_all(std::integer_sequence<int, KS...>, int64_t& acc, int32_t (&m)[BN_NL], Fp& r, const DotTerm<SIGNS>&... terms) {
  (dot_step<KS, SIGNS...>(acc, m, r, terms...), ...);
}
template <int... SIGNS>
BN_HD Fp fp_dot(const DotTerm<SIGNS>&... terms) {
  BN_SCHED_FENCE();
  int64_t acc = 0;
  int32_t m[BN_NL];
  Fp r;
  dot_all(std::make_integer_sequence<int, 2 * BN_NL - 1>{}, acc, m, r, terms...);
  r.v[BN_NL - 1] = (int32_t)acc;
  BN_SCHED_FENCE();
#if BN_TRACKING
  double cap = ((9.0 * bn_iabs(SIGNS) * terms.a.lb * terms.b.lb) + ...);  // in units of 2^58
  if (cap + 9.0 * 0.25 + 0.01 > 32.0) fp_dbg_fail("fp_dot: accumulator may overflow", cap);
  double s = ((bn_iabs(SIGNS) * terms.a.vb * terms.b.vb) + ...);
  BN_SETB(r, 1.0 + s / 169.0 + 1e-6, 0.5);
#endif
  return r;
}
BN_HD Fp fp_mul(const Fp& a, const Fp& b) { return fp_dot(dplus(a, b)); }
// squaring: cross products once against a doubled operand (45 + 81 mads instead of 162)
BN_HD Fp fp_sqr(const Fp& a) {
  BN_SCHED_FENCE();
  int64_t acc = 0;
  int32_t m[BN_NL];
  int32_t a2[BN_NL];
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) a2[i] = a.v[i] * 2;
#pragma unroll
  for (int k = 0; k < 2 * BN_NL - 1; k++) {
#pragma unroll
    for (int i = 0; i < BN_NL; i++) {
      int j = k - i;
      if (j >= 0 && j < BN_NL && i < j) acc += (int64_t)a2[i] * (int64_t)a.v[j];
    }
    if ((k & 1) == 0) acc += (int64_t)a.v[k / 2] * (int64_t)a.v[k / 2];
    if (k < BN_NL) {
#pragma unroll
      for (int i = 0; i < k; i++) acc += (int64_t)m[i] * (int64_t)bn_p_limb(k - i);
      m[k] = bn_sext29((uint32_t)acc * BN_PINV);
      acc += (int64_t)m[k] * (int64_t)bn_p_limb(0);
      acc >>= BN_LB;
    } else {
#pragma unroll
      for (int i = k - (BN_NL - 1); i < BN_NL; i++) acc += (int64_t)m[i] * (int64_t)bn_p_limb(k - i);
      acc += BN_HALF;
      r.v[k - BN_NL] = (int32_t)((uint32_t)acc & BN_MASK) - BN_HALF;
      acc >>= BN_LB;
    }
  }
  r.v[BN_NL - 1] = (int32_t)acc;
  BN_SCHED_FENCE();
#if BN_TRACKING
  if (9.0 * a.lb * a.lb + 2.26 > 32.0) fp_dbg_fail("fp_sqr: accumulator may overflow", a.lb);
  BN_SETB(r, 1.0 + a.vb * a.vb / 169.0 + 1e-6, 0.5);
#endif
  return r;
}

// ---- canonical form [0, p) with unique digits; equality / zero tests; byte output --------------------------------------------
BN_HD bool fp_is_negative(const Fp& a) {  // sign of the value = sign of the most significant non-zero digit
  int32_t s = a.v[BN_NL - 1];
#pragma unroll
  for (int i = BN_NL - 2; i >= 0; i--) s = (s != 0) ? s : a.v[i];
  return s < 0;
}
BN_HD Fp fp_select(bool c, const Fp& a, const Fp& b) {  // c ? a : b, branch-free
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = c ? a.v[i] : b.v[i];
#if BN_TRACKING
  BN_SETB(r, a.vb > b.vb ? a.vb : b.vb, a.lb > b.lb ? a.lb : b.lb);
#endif
  return r;
}
BN_HD Fp fp_canon(const Fp& a) {
  Fp r = fp_reduce(fp_norm(a));  // in (-eps p, (1 + eps) p)
  Fp pl = fp_from_limbs(BN_P);
  r = fp_select(fp_is_negative(r), fp_norm(fp_add_lazy(r, pl)), r);
  Fp dn = fp_norm(fp_sub_lazy(r, pl));
  r = fp_select(!fp_is_negative(dn), dn, r);
  BN_SETB(r, 1.0, 0.5);
  return r;
}
BN_HD bool fp_is_zero(const Fp& a) {
  Fp c = fp_canon(a);
  int32_t d = 0;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) d |= c.v[i];
  return d == 0;
}
BN_HD bool fp_eq(const Fp& a, const Fp& b) { return fp_is_zero(fp_sub(a, b)); }

// ---- conversions: 8 x 32-bit little-endian words (plain integer) <-> Montgomery digits ---------------------------------------
BN_HD Fp fp_from_words_raw(const uint32_t w[8]) {  // plain integer < 2^256 to balanced digits, NOT Montgomery
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) {
    int bit = BN_LB * i, wi = bit >> 5, sh = bit & 31;
    uint64_t lo = w[wi];
    uint64_t hi = (wi + 1 < 8) ? w[wi + 1] : 0;
    r.v[i] = (int32_t)((uint32_t)(((hi << 32) | lo) >> sh) & BN_MASK);
  }
  BN_SETB(r, 5.3, 1.0);  // < 2^256, digits still unsigned
  return fp_norm(r);
}
BN_HD Fp fp_from_words(const uint32_t w[8]) {  // integer < 2^256 (any) -> Montgomery form of it mod p
  return fp_mul(fp_from_words_raw(w), fp_from_limbs(BN_R2));
}
BN_HD void fp_to_words(uint32_t w[8], const Fp& a) {  // Montgomery -> canonical integer in [0, p)
  Fp one_plain = fp_zero();
  one_plain.v[0] = 1;
  Fp c = fp_canon(fp_mul(a, one_plain));
  // balanced -> unsigned digits (the value is non-negative)
  uint32_t u[BN_NL];
  int32_t cy = 0;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) {
    int32_t t = c.v[i] + cy;
    u[i] = (uint32_t)t & BN_MASK;
    cy = t >> BN_LB;
  }
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = 0;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) {
    int bit = BN_LB * i, wi = bit >> 5, sh = bit & 31;
    uint64_t v = (uint64_t)u[i] << sh;
    if (wi < 8) w[wi] |= (uint32_t)v;
    if (wi + 1 < 8) w[wi + 1] |= (uint32_t)(v >> 32);
  }
}
// little-endian word compare: a >= b
BN_HD bool words_ge(const uint32_t a[8], const uint32_t b[8]) {
  bool ge = true;
#pragma unroll
  for (int i = 0; i < 8; i++) ge = (a[i] > b[i]) || (a[i] == b[i] && ge);
  return ge;
}
BN_HD void words_from_be(uint32_t w[8], const uint8_t* be32) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint8_t* p = be32 + 4 * (7 - i);
    w[i] = (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | (uint32_t)p[3];
  }
}
BN_HD void words_to_be(uint8_t* be32, const uint32_t w[8]) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint8_t* p = be32 + 4 * (7 - i);
    p[0] = (uint8_t)(w[i] >> 24); p[1] = (uint8_t)(w[i] >> 16); p[2] = (uint8_t)(w[i] >> 8); p[3] = (uint8_t)w[i];
  }
}

// ---- exponentiation by a fixed public exponent (bit table from bn254_constants.h), inversion -------------------------------------
BN_HD_DEVINLINE Fp fp_mul_nl(const Fp& a, const Fp& b) { return fp_mul(a, b); }
BN_HD_DEVINLINE Fp fp_sqr_nl(const Fp& a) { return fp_sqr(a); }
BN_HD Fp fp_pow_bits(const Fp& a, const uint8_t* bits, int nbits) {  // bits[0] = leading 1
  Fp acc = a;
  for (int i = 1; i < nbits; i++) {
    acc = fp_sqr_nl(acc);
    if (bits[i]) acc = fp_mul_nl(acc, a);  // public exponent: the branch is wave-uniform
  }
  return acc;
}
BN_HD Fp fp_inv_fermat(const Fp& a) { return fp_pow_bits(fp_reduce(fp_norm(a)), BN_EXP_PM2_BITS, BN_EXP_PM2_NBITS); }  // 0 -> 0

// ---- inversion by the binary extended GCD with approximated operands (T. Pornin, "Optimized Binary GCD for Modular Inversion", 2020, algorithm 2,
// with k = 30 so that its divisions by 2^(k-1) are shifts by one 29-bit digit).  Invariants a = u y / C, b = v y / C (mod p), a, b >= 0, start
// (a, u, b, v) = (y, C, p, 0).  Each of the 18 outer rounds runs 29 steps of the binary GCD on 60-bit stand-ins for a and b -- their 29 low bits
// and the 31 bits below the top bit of the larger (exact values once both fit 62 bits) -- which yields the update factors f, g with
// |f| + |g| <= 2^29; the factors are then applied to the full numbers (one multiply-accumulate pass each, the quotient by 2^29 exact) and to
// u, v modulo p (one Montgomery digit retired per round, the same -1/p mod 2^29 as fp_dot).  18 * 29 = 522 >= 2 * 254 - 1 steps, so b ends as
// gcd(y, p) = 1 and v = C / y; with y = x R and C = R^2 that is the Montgomery form of 1 / x.  y = 0 leaves v = 0: the inverse of 0 is 0, as
// with the Fermat form.  About 17 k instructions instead of the 70 k of x^(p-2) (253 squarings + 109 products); fixed iteration counts, no
// data-dependent branch.
BN_HD int bn_clz64(uint64_t x) {  // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
  return __clzll((long long)x);
#else
  return __builtin_clzll(x);
#endif
}
BN_HD void fp_unsigned_digits(int32_t d[BN_NL], const int32_t* balanced) {  // value >= 0 in balanced digits -> digits in [0, 2^29)
  int32_t cy = 0;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) { int32_t t = balanced[i] + cy; d[i] = t & (int32_t)BN_MASK; cy = t >> BN_LB; }
}
BN_HD Fp fp_inv(const Fp& x) {
  const Fp y = fp_canon(x);
  int32_t a[BN_NL], b[BN_NL], u[BN_NL], v[BN_NL];
  fp_unsigned_digits(a, y.v);
  {
    int32_t pl[BN_NL];
#pragma unroll
    for (int i = 0; i < BN_NL; i++) pl[i] = bn_p_limb(i);
    fp_unsigned_digits(b, pl);
  }
  {
    const Fp c = fp_from_limbs(BN_R2);
#pragma unroll
    for (int i = 0; i < BN_NL; i++) { u[i] = c.v[i]; v[i] = 0; }
  }
#pragma unroll 1
  for (int round = 0; round < 18; round++) {
    // the three digits from the highest position where a or b is non-zero (positions 2..8); none: both numbers are below 2^58
    int32_t ah = 0, am = 0, al = 0, bh = 0, bm = 0, bl = 0, low = 0;   // low: the position of *l is 0 (three digits are the whole number)
    bool found = false;
#pragma unroll
    for (int i = BN_NL - 1; i >= 2; i--) {
      const bool take = !found && ((a[i] | b[i]) != 0);
      ah = take ? a[i] : ah; am = take ? a[i - 1] : am; al = take ? a[i - 2] : al;
      bh = take ? b[i] : bh; bm = take ? b[i - 1] : bm; bl = take ? b[i - 2] : bl;
      low = take ? (i == 2 ? 1 : 0) : low;
      found = found || take;
    }
    const uint64_t hiA = ((uint64_t)(uint32_t)ah << BN_LB) | (uint32_t)am, hiB = ((uint64_t)(uint32_t)bh << BN_LB) | (uint32_t)bm;
    const int len = 64 - bn_clz64(hiA | hiB | 1);                    // 30..58 when found
    const int sh = len - 31;                                         // -1..27
    const uint64_t topA = sh >= 0 ? (hiA >> (sh & 63)) : ((hiA << 1) | ((uint32_t)al >> 28));
    const uint64_t topB = sh >= 0 ? (hiB >> (sh & 63)) : ((hiB << 1) | ((uint32_t)bl >> 28));
    const uint64_t lowA = ((uint64_t)(uint32_t)a[1] << BN_LB) | (uint32_t)a[0], lowB = ((uint64_t)(uint32_t)b[1] << BN_LB) | (uint32_t)b[0];
    // exact stand-ins when the numbers fit 62 bits: fewer than three digits, or three digits with a top digit below 2^4
    const bool exact3 = found && low != 0 && len <= 33;
    uint64_t A = !found ? lowA : exact3 ? ((hiA << BN_LB) | (uint32_t)al) : ((topA << BN_LB) | (uint32_t)a[0]);
    uint64_t B = !found ? lowB : exact3 ? ((hiB << BN_LB) | (uint32_t)bl) : ((topB << BN_LB) | (uint32_t)b[0]);
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 1
    for (int j = 0; j < BN_LB; j++) {
      const bool odd = (A & 1) != 0;
      const bool swap = odd && A < B;
      const uint64_t tA = swap ? B : A, tB = swap ? A : B;
      const int32_t tf0 = swap ? f1 : f0, tg0 = swap ? g1 : g0, tf1 = swap ? f0 : f1, tg1 = swap ? g0 : g1;
      A = (tA - (odd ? tB : 0)) >> 1; B = tB;
      f0 = tf0 - (odd ? tf1 : 0); g0 = tg0 - (odd ? tg1 : 0);
      f1 = (int32_t)((uint32_t)tf1 << 1); g1 = (int32_t)((uint32_t)tg1 << 1);   // (as unsigned: shifting a negative value left is undefined before C++20)
    }
    // (a, b) <- (a f0 + b g0, a f1 + b g1) / 2^29, made non-negative by negating the number together with its factors
    int32_t na[BN_NL], nb[BN_NL];
    {
      int64_t ca = (int64_t)a[0] * f0 + (int64_t)b[0] * g0, cb = (int64_t)a[0] * f1 + (int64_t)b[0] * g1;
      ca >>= BN_LB; cb >>= BN_LB;                                    // exact: the low 29 bits are zero
#pragma unroll
      for (int i = 1; i < BN_NL; i++) {
        ca += (int64_t)a[i] * f0 + (int64_t)b[i] * g0; cb += (int64_t)a[i] * f1 + (int64_t)b[i] * g1;
        na[i - 1] = (int32_t)((uint32_t)ca & BN_MASK); nb[i - 1] = (int32_t)((uint32_t)cb & BN_MASK);
        ca >>= BN_LB; cb >>= BN_LB;
      }
      na[BN_NL - 1] = (int32_t)ca; nb[BN_NL - 1] = (int32_t)cb;     // 0 or -1 (the numbers stay below 2^254)
    }
    const bool nega = na[BN_NL - 1] < 0, negb = nb[BN_NL - 1] < 0;
    {
      int32_t ba = 0, bb = 0;
#pragma unroll
      for (int i = 0; i < BN_NL; i++) {
        const int32_t ta = -na[i] - ba, tb = -nb[i] - bb;
        const int32_t da = i < BN_NL - 1 ? (ta & (int32_t)BN_MASK) : ta, db = i < BN_NL - 1 ? (tb & (int32_t)BN_MASK) : tb;
        ba = i < BN_NL - 1 ? ((ta >> BN_LB) & 1) : 0; bb = i < BN_NL - 1 ? ((tb >> BN_LB) & 1) : 0;
        a[i] = nega ? da : na[i]; b[i] = negb ? db : nb[i];
      }
    }
    f0 = nega ? -f0 : f0; g0 = nega ? -g0 : g0; f1 = negb ? -f1 : f1; g1 = negb ? -g1 : g1;
    // (u, v) <- (u f0 + v g0, u f1 + v g1) / 2^29 mod p
    {
      int64_t cu = (int64_t)u[0] * f0 + (int64_t)v[0] * g0, cv = (int64_t)u[0] * f1 + (int64_t)v[0] * g1;
      const int32_t qu = bn_sext29((uint32_t)cu * BN_PINV), qv = bn_sext29((uint32_t)cv * BN_PINV);
      cu += (int64_t)qu * bn_p_limb(0); cv += (int64_t)qv * bn_p_limb(0);
      cu >>= BN_LB; cv >>= BN_LB;                                    // exact
      int32_t nu[BN_NL], nv[BN_NL];
#pragma unroll
      for (int i = 1; i < BN_NL; i++) {
        cu += (int64_t)u[i] * f0 + (int64_t)v[i] * g0 + (int64_t)qu * bn_p_limb(i);
        cv += (int64_t)u[i] * f1 + (int64_t)v[i] * g1 + (int64_t)qv * bn_p_limb(i);
        nu[i - 1] = (int32_t)((uint32_t)cu & BN_MASK); nv[i - 1] = (int32_t)((uint32_t)cv & BN_MASK);
        cu >>= BN_LB; cv >>= BN_LB;
      }
      nu[BN_NL - 1] = (int32_t)cu; nv[BN_NL - 1] = (int32_t)cv;
#pragma unroll
      for (int i = 0; i < BN_NL; i++) { u[i] = nu[i]; v[i] = nv[i]; }
    }
  }
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = v[i];
  BN_SETB(r, 32.0, 1.0);     // |u|, |v| grow by at most p / 2 + ... per round: below 12 p after 18 rounds
  return fp_reduce(fp_norm(r));
}

}  // namespace bn254
