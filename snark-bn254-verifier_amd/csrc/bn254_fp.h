// bn254_fp.h -- BN254 base-field arithmetic for gfx950 (and, compiled unchanged, for the host).
//
// Representation (DESIGN.md "Fp on the VALU"): an element is 9 signed 32-bit limbs of 29 bits,
//     value = sum_i v[i] * 2^(29 i),   v[0..7] in [0, 2^29) once normalised, v[8] signed,
// held in Montgomery form with R = 2^261.  gfx950 has no 64x64 multiplier; its widest integer multiply is
// v_mad_i64_i32 / v_mad_u64_u32 (32x32 + 64 -> 64, measured at ~1.9 ns per wave-instruction per SIMD, about
// twice a plain v_add_u32: profiles/r01_ubench_valu.txt).  With 29-bit limbs a whole column of the
// schoolbook product (<= 9 + 9 partial products < 2^58 each) accumulates in ONE 64-bit register pair with no
// carry handling at all, so a Montgomery product is 162 mads + 9 mul_lo + ~50 shifts/ands, against 136 multiplies
// plus >250 carry instructions for saturated 8 x 32-bit limbs.
//
// Values are signed and only loosely bounded: |value| <= vb * p with vb tracked statically by the author and
// checked dynamically by the host-side bound tracker (BN_TRACK_BOUNDS, used by tests/hostsim).  Because
// R / p ~ 2^7.4, a product of inputs with vb_a * vb_b <= 169 comes back in (-vb_a vb_b/169 - eps, 1 + vb_a vb_b/169) * p,
// i.e. multiplication contracts bounds; additions and subtractions just add them (no conditional
// subtraction, no "+ k p").  fp_reduce() brings any representable value back to (-eps, 1 + eps) * p for ~60
// simple instructions where a formula would otherwise exceed the representable range (|value| < 2^261 ~ 169 p,
// so that the top limb stays below 2^29 like the others).
//
// Replaces (behaviourally) bn::Fq of substrate-bn, used by the reference at verifier/src/converter.rs:85-86,
// 144-147 (Fq::from_slice) and everywhere below bn::pairing_batch (groth16/verify.rs:70-77).
#pragma once
#include <stdint.h>
#include "bn254_constants.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BN_HD __host__ __device__ __forceinline__
#if defined(BN_INLINE_ALL)
#define BN_HD_NOINLINE __host__ __device__ __forceinline__
#else
#define BN_HD_NOINLINE __host__ __device__ inline __attribute__((noinline))
#endif
#else
#if defined(BN_TRACK_BOUNDS)
#define BN_HD inline
#else
#define BN_HD inline __attribute__((always_inline))
#endif
#define BN_HD_NOINLINE inline __attribute__((noinline))
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

struct Fp {
  int32_t v[BN_NL];
#if BN_TRACKING
  double vb;  // |value| <= vb * p
  double lb;  // max |limb| <= lb * 2^29 (limbs 0..7)
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
  if (a.vb > 160.0) fp_dbg_fail("value bound exceeds representable range (|v| < 2^261)", a.vb);
  for (int i = 0; i < 8; i++)
    if (fabs((double)a.v[i]) > a.lb * 536870912.0 + 0.5) fp_dbg_fail("limb bound wrong", (double)a.v[i]);
}
#define BN_SETB(x, VB, LB) do { (x).vb = (VB); (x).lb = (LB); fp_dbg_check((x), __func__); } while (0)
#else
#define BN_SETB(x, VB, LB) do { } while (0)
#endif

BN_HD int32_t bn_p_limb(int i) {
  // compile-time constants after unrolling: live in SGPRs / literals, never in memory
  switch (i) {
    case 0: return (int32_t)BN_P0; case 1: return (int32_t)BN_P1; case 2: return (int32_t)BN_P2;
    case 3: return (int32_t)BN_P3; case 4: return (int32_t)BN_P4; case 5: return (int32_t)BN_P5;
    case 6: return (int32_t)BN_P6; case 7: return (int32_t)BN_P7; default: return (int32_t)BN_P8;
  }
}

BN_HD Fp fp_from_limbs(const uint32_t* c) {  // a precomputed constant: normalised, value in [0, 16p]
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = (int32_t)c[i];
  BN_SETB(r, 1.0, 1.0);
  return r;
}
BN_HD Fp fp_zero() {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = 0;
  BN_SETB(r, 0.0, 1.0);
  return r;
}
BN_HD Fp fp_one() { return fp_from_limbs(BN_ONE); }

// ---- carry propagation: limbs 0..7 back into [0, 2^29), top limb keeps the sign --------------------------------
BN_HD Fp fp_norm(const Fp& a) {
  Fp r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < BN_NL - 1; i++) {
    int32_t t = a.v[i] + c;
    r.v[i] = t & (int32_t)BN_MASK;
    c = t >> BN_LB;  // arithmetic shift: floor division, exact for negative limbs too
  }
  r.v[BN_NL - 1] = a.v[BN_NL - 1] + c;
#if BN_TRACKING
  if (a.lb > 3.9) fp_dbg_fail("fp_norm: limb overflow", a.lb);
  BN_SETB(r, a.vb, 1.0);
#endif
  return r;
}

// ---- lazy (limb-wise, no carry) add / sub; callers normalise before the limbs can leave (-2^31, 2^31) ----------
BN_HD Fp fp_add_lazy(const Fp& a, const Fp& b) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = a.v[i] + b.v[i];
  BN_SETB(r, a.vb + b.vb, a.lb + b.lb);
  return r;
}
BN_HD Fp fp_sub_lazy(const Fp& a, const Fp& b) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = a.v[i] - b.v[i];
  BN_SETB(r, a.vb + b.vb, a.lb + b.lb);
  return r;
}
BN_HD Fp fp_add(const Fp& a, const Fp& b) { return fp_norm(fp_add_lazy(a, b)); }
BN_HD Fp fp_sub(const Fp& a, const Fp& b) { return fp_norm(fp_sub_lazy(a, b)); }
BN_HD Fp fp_neg(const Fp& a) {
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = -a.v[i];
  BN_SETB(r, a.vb, a.lb);
  return fp_norm(r);
}
BN_HD Fp fp_dbl(const Fp& a) { return fp_add(a, a); }
// ---- small linear combinations in one carry pass: normalised k1*a + k2*b (|k| small), optionally reduced -------
// inputs normalised (lb <= 1); |k1| + |k2| <= 32 keeps the 64-bit chain far from overflow
BN_HD Fp fp_lincomb(int32_t k1, const Fp& a, int32_t k2, const Fp& b) {
  Fp r;
  int64_t acc = 0;
#pragma unroll
  for (int i = 0; i < BN_NL - 1; i++) {
    acc += (int64_t)k1 * (int64_t)a.v[i] + (int64_t)k2 * (int64_t)b.v[i];
    r.v[i] = (int32_t)((uint32_t)acc & BN_MASK);
    acc >>= BN_LB;
  }
  acc += (int64_t)k1 * (int64_t)a.v[BN_NL - 1] + (int64_t)k2 * (int64_t)b.v[BN_NL - 1];
  r.v[BN_NL - 1] = (int32_t)acc;
#if BN_TRACKING
  if (a.lb > 1.0 || b.lb > 1.0) fp_dbg_fail("fp_lincomb: input not normalised", a.lb + b.lb);
  BN_SETB(r, a.vb * (k1 < 0 ? -k1 : k1) + b.vb * (k2 < 0 ? -k2 : k2), 1.0);
#endif
  return r;
}
// reduce(k1*a + k2*b) in the same pass: q is estimated from the top limbs alone (the lower limbs move value/p by
// less than (|k1|+|k2|) * 2^232 / p < 1e-5), result in (-1e-5, 1 + 1e-5) * p.  The combination itself never has to be
// representable in limbs (the pass runs in 64-bit), so any input bounds are fine.
BN_HD Fp fp_lincomb_reduce(int32_t k1, const Fp& a, int32_t k2, const Fp& b) {
  const int64_t C = 1420063842;  // floor(2^284 / p)
  int64_t top = (int64_t)k1 * (int64_t)a.v[BN_NL - 1] + (int64_t)k2 * (int64_t)b.v[BN_NL - 1];
  int32_t q = (int32_t)((top * C) >> 52);
  Fp r;
  int64_t acc = 0;
#pragma unroll
  for (int i = 0; i < BN_NL - 1; i++) {
    acc += (int64_t)k1 * (int64_t)a.v[i] + (int64_t)k2 * (int64_t)b.v[i] - (int64_t)q * (int64_t)bn_p_limb(i);
    r.v[i] = (int32_t)((uint32_t)acc & BN_MASK);
    acc >>= BN_LB;
  }
  acc += top - (int64_t)q * (int64_t)bn_p_limb(BN_NL - 1);
  r.v[BN_NL - 1] = (int32_t)acc;
#if BN_TRACKING
  if (a.lb > 1.0 || b.lb > 1.0) fp_dbg_fail("fp_lincomb_reduce: input not normalised", a.lb + b.lb);
  if (a.vb * (k1 < 0 ? -k1 : k1) + b.vb * (k2 < 0 ? -k2 : k2) > 1e5) fp_dbg_fail("fp_lincomb_reduce: out of range", a.vb + b.vb);
  BN_SETB(r, 1.0 + 2e-5, 1.0);
#endif
  return r;
}

// ---- Montgomery product (finely-integrated product scanning): one 64-bit accumulator, no carries ---------------
// requires lb_a * lb_b <= 2.5 (9 * 2^58 * lb_a lb_b + 9 * 2^58 + carry < 2^63)
BN_HD Fp fp_mul(const Fp& a, const Fp& b) {
  BN_SCHED_FENCE();
  int64_t acc = 0;
  int32_t m[BN_NL];
  Fp r;
#pragma unroll
  for (int k = 0; k < BN_NL; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (int64_t)a.v[i] * (int64_t)b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (int64_t)m[i] * (int64_t)bn_p_limb(k - i);
    m[k] = (int32_t)(((uint32_t)acc * BN_PINV) & BN_MASK);
    acc += (int64_t)m[k] * (int64_t)bn_p_limb(0);
    acc >>= BN_LB;
  }
#pragma unroll
  for (int k = BN_NL; k < 2 * BN_NL - 1; k++) {
#pragma unroll
    for (int i = k - (BN_NL - 1); i < BN_NL; i++) acc += (int64_t)a.v[i] * (int64_t)b.v[k - i];
#pragma unroll
    for (int i = k - (BN_NL - 1); i < BN_NL; i++) acc += (int64_t)m[i] * (int64_t)bn_p_limb(k - i);
    r.v[k - BN_NL] = (int32_t)((uint32_t)acc & BN_MASK);
    acc >>= BN_LB;
  }
  r.v[BN_NL - 1] = (int32_t)acc;
  BN_SCHED_FENCE();
#if BN_TRACKING
  if (a.lb * b.lb > 2.5) fp_dbg_fail("fp_mul: accumulator may overflow", a.lb * b.lb);
  BN_SETB(r, 1.0 + a.vb * b.vb / 169.0 + 1e-6, 1.0);
#endif
  return r;
}
// squaring: the 36 cross products are taken once against a doubled operand (45 + 81 mads instead of 162)
BN_HD Fp fp_sqr(const Fp& a) {
  BN_SCHED_FENCE();
  int64_t acc = 0;
  int32_t m[BN_NL];
  int32_t a2[BN_NL];
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) a2[i] = a.v[i] * 2;
#pragma unroll
  for (int k = 0; k < BN_NL; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) acc += (int64_t)a2[i] * (int64_t)a.v[k - i];
    if ((k & 1) == 0) acc += (int64_t)a.v[k / 2] * (int64_t)a.v[k / 2];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (int64_t)m[i] * (int64_t)bn_p_limb(k - i);
    m[k] = (int32_t)(((uint32_t)acc * BN_PINV) & BN_MASK);
    acc += (int64_t)m[k] * (int64_t)bn_p_limb(0);
    acc >>= BN_LB;
  }
#pragma unroll
  for (int k = BN_NL; k < 2 * BN_NL - 1; k++) {
#pragma unroll
    for (int i = k - (BN_NL - 1); 2 * i < k; i++) acc += (int64_t)a2[i] * (int64_t)a.v[k - i];
    if ((k & 1) == 0) acc += (int64_t)a.v[k / 2] * (int64_t)a.v[k / 2];
#pragma unroll
    for (int i = k - (BN_NL - 1); i < BN_NL; i++) acc += (int64_t)m[i] * (int64_t)bn_p_limb(k - i);
    r.v[k - BN_NL] = (int32_t)((uint32_t)acc & BN_MASK);
    acc >>= BN_LB;
  }
  r.v[BN_NL - 1] = (int32_t)acc;
  BN_SCHED_FENCE();
#if BN_TRACKING
  if (a.lb * a.lb > 1.25) fp_dbg_fail("fp_sqr: accumulator may overflow", a.lb);  // doubled operand: 2 lb^2 <= 2.5
  BN_SETB(r, 1.0 + a.vb * a.vb / 169.0 + 1e-6, 1.0);
#endif
  return r;
}

// ---- quotient-estimate reduction: any normalised value with |value| < 2^260 -> (-2^-20, 1 + 2^-20) * p -----------
// q = floor(v[8] * C / 2^52) with C = floor(2^284 / p) is floor(value / p) or one less (DESIGN.md "fp_reduce").
BN_HD Fp fp_reduce(const Fp& a) {
  const int64_t C = 1420063842;  // floor(2^284 / p)
  int32_t q = (int32_t)(((int64_t)a.v[BN_NL - 1] * C) >> 52);
  Fp r;
  int64_t acc = 0;
#pragma unroll
  for (int i = 0; i < BN_NL - 1; i++) {
    acc += (int64_t)a.v[i] - (int64_t)q * (int64_t)bn_p_limb(i);
    r.v[i] = (int32_t)((uint32_t)acc & BN_MASK);
    acc >>= BN_LB;
  }
  acc += (int64_t)a.v[BN_NL - 1] - (int64_t)q * (int64_t)bn_p_limb(BN_NL - 1);
  r.v[BN_NL - 1] = (int32_t)acc;
#if BN_TRACKING
  if (a.lb > 1.0) fp_dbg_fail("fp_reduce: input not normalised", a.lb);
  BN_SETB(r, 1.0 + 1e-5, 1.0);
#endif
  return r;
}

// ---- canonical form [0, p), unique limbs; used for equality / zero tests and for byte output ---------------------
BN_HD bool fp_limbs_eq(const Fp& a, const Fp& b) {
  int32_t d = 0;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) d |= a.v[i] ^ b.v[i];
  return d == 0;
}
BN_HD Fp fp_canon(const Fp& a) {
  Fp r = fp_reduce(fp_norm(a));  // in (-eps p, (1 + eps) p)
  Fp pl = fp_from_limbs(BN_P);
  // negative -> + p
  Fp up = fp_norm(fp_add_lazy(r, pl));
  bool neg = r.v[BN_NL - 1] < 0;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = neg ? up.v[i] : r.v[i];
  // >= p -> - p
  Fp dn = fp_norm(fp_sub_lazy(r, pl));
  bool ge = dn.v[BN_NL - 1] >= 0;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = ge ? dn.v[i] : r.v[i];
  BN_SETB(r, 1.0, 1.0);
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
BN_HD Fp fp_select(bool c, const Fp& a, const Fp& b) {  // c ? a : b, branch-free
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) r.v[i] = c ? a.v[i] : b.v[i];
#if BN_TRACKING
  BN_SETB(r, a.vb > b.vb ? a.vb : b.vb, a.lb > b.lb ? a.lb : b.lb);
#endif
  return r;
}

// ---- conversions: 8 x 32-bit little-endian words (plain integer) <-> Montgomery limbs ---------------------------
BN_HD Fp fp_from_words_raw(const uint32_t w[8]) {  // plain integer < 2^256 to limbs, NOT Montgomery
  Fp r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) {
    int bit = BN_LB * i, wi = bit >> 5, sh = bit & 31;
    uint64_t lo = w[wi];
    uint64_t hi = (wi + 1 < 8) ? w[wi + 1] : 0;
    r.v[i] = (int32_t)((uint32_t)(((hi << 32) | lo) >> sh) & BN_MASK);
  }
  BN_SETB(r, 5.3, 1.0);  // < 2^256
  return r;
}
BN_HD Fp fp_from_words(const uint32_t w[8]) {  // integer < 2^256 (any) -> Montgomery form of it mod p
  return fp_mul(fp_from_words_raw(w), fp_from_limbs(BN_R2));
}
BN_HD void fp_to_words(uint32_t w[8], const Fp& a) {  // Montgomery -> canonical integer in [0, p)
  Fp one_plain = fp_zero();
  one_plain.v[0] = 1;
  Fp c = fp_canon(fp_mul(a, one_plain));
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = 0;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) {
    int bit = BN_LB * i, wi = bit >> 5, sh = bit & 31;
    uint64_t v = (uint64_t)(uint32_t)c.v[i] << sh;
    w[wi] |= (uint32_t)v;
    if (wi + 1 < 8) w[wi + 1] |= (uint32_t)(v >> 32);
  }
}
// little-endian word compare: a >= b
BN_HD bool words_ge(const uint32_t a[8], const uint32_t b[8]) {
  bool ge = true;  // equal so far, scanning from the least significant word
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

// ---- exponentiation by a fixed public exponent (bit table from bn254_constants.h), inversion, square root ------
BN_HD_NOINLINE Fp fp_mul_nl(const Fp& a, const Fp& b) { return fp_mul(a, b); }
BN_HD_NOINLINE Fp fp_sqr_nl(const Fp& a) { return fp_sqr(a); }
BN_HD Fp fp_pow_bits(const Fp& a, const uint8_t* bits, int nbits) {  // bits[0] = leading 1
  Fp acc = a;
  for (int i = 1; i < nbits; i++) {
    acc = fp_sqr_nl(acc);
    if (bits[i]) acc = fp_mul_nl(acc, a);  // public exponent: the branch is wave-uniform
  }
  return acc;
}
BN_HD Fp fp_inv(const Fp& a) { return fp_pow_bits(fp_reduce(fp_norm(a)), BN_EXP_PM2_BITS, BN_EXP_PM2_NBITS); }  // 0 -> 0

}  // namespace bn254
