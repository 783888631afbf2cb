/* oracle/field.c -- u256, Montgomery prime fields Fp / Fr, and the Fp2 / Fp6 / Fp12 tower of BN254.
 * TEST INFRASTRUCTURE (see oracle.h).  Restates the arithmetic of the un-vendored `bn` crate
 * (substrate-bn 0.7.0 @ sp1-patches 3c53d25; SURVEY.md Appendix B.1 for the constants, C.2 for the
 * function inventory: U256::{add,sub,mul,invert}, Fq::sqrt, Fq2::{mul,inverse,sqrt}, Fq6::{mul,squared,
 * inverse,frobenius_map}, Fq12::{mul,squared,inverse,mul_by_024,cyclotomic_squared,frobenius_map}).
 * Tower: Fp2 = Fp[i]/(i^2+1), Fp6 = Fp2[v]/(v^3 - xi), xi = 9+i, Fp12 = Fp6[w]/(w^2 - v). */
#include "oracle.h"
#include <string.h>

typedef unsigned __int128 u128;

fctx FP, FR;
__thread uint64_t orc_fp_mul_count = 0; /* per thread: a shared counter would serialise the OpenMP baseline on one cache line */
static int g_init = 0;

/* p and r, big-endian hex split into 64-bit little-endian limbs */
static const u256 P_MOD = {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
static const u256 R_MOD = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};

/* Frobenius coefficients gamma[k] = xi^(k(p-1)/6), k = 0..5 (computed in orc_init) */
static fp2 FROB_G[6];
static fp2 XI;           /* 9 + i */
fp2 TWIST_B;             /* 3 / xi : b' of the D-type twist y^2 = x^3 + 3/xi */
fp FP_TWO_INV;
static u256 P_MINUS_1_OVER_6, P_PLUS_1_OVER_4, P_MINUS_3_OVER_4, P_MINUS_1_OVER_2, P_MINUS_2, R_MINUS_2;

/* ---------------- u256 ---------------- */
int u256_cmp(const u256* a, const u256* b) {
  for (int i = 3; i >= 0; i--) {
    if (a->l[i] < b->l[i]) return -1;
    if (a->l[i] > b->l[i]) return 1;
  }
  return 0;
}
int u256_is_zero(const u256* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
void u256_from_be(u256* o, const uint8_t* b) {
  for (int i = 0; i < 4; i++) {
    uint64_t v = 0;
    for (int j = 0; j < 8; j++) v = (v << 8) | b[(3 - i) * 8 + j];
    o->l[i] = v;
  }
}
void u256_to_be(uint8_t* b, const u256* a) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 8; j++) b[(3 - i) * 8 + j] = (uint8_t)(a->l[i] >> (56 - 8 * j));
}
int u256_bit(const u256* a, int i) { return (int)((a->l[i >> 6] >> (i & 63)) & 1); }
static uint64_t u256_add_raw(u256* o, const u256* a, const u256* b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; o->l[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t u256_sub_raw(u256* o, const u256* a, const u256* b) {
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a->l[i] - b->l[i] - br;
    o->l[i] = (uint64_t)d;
    br = (uint64_t)(d >> 64) & 1;
  }
  return br;
}
static void u256_shr1(u256* a) {
  for (int i = 0; i < 4; i++) a->l[i] = (a->l[i] >> 1) | (i < 3 ? a->l[i + 1] << 63 : 0);
}
/* o = a / d for small d, returns remainder */
static uint64_t u256_div_small(u256* o, const u256* a, uint64_t d) {
  u128 rem = 0;
  for (int i = 3; i >= 0; i--) { u128 cur = (rem << 64) | a->l[i]; o->l[i] = (uint64_t)(cur / d); rem = cur % d; }
  return (uint64_t)rem;
}

/* ---------------- generic Montgomery field ---------------- */
void f_add(const fctx* F, u256* o, const u256* a, const u256* b) {
  u256 t; uint64_t c = u256_add_raw(&t, a, b);
  u256 s; uint64_t br = u256_sub_raw(&s, &t, &F->m);
  *o = (c || !br) ? s : t;
}
void f_sub(const fctx* F, u256* o, const u256* a, const u256* b) {
  u256 t; uint64_t br = u256_sub_raw(&t, a, b);
  if (br) u256_add_raw(&t, &t, &F->m);
  *o = t;
}
void f_neg(const fctx* F, u256* o, const u256* a) {
  if (u256_is_zero(a)) { *o = *a; return; }
  u256_sub_raw(o, &F->m, a);
}
/* CIOS Montgomery multiplication, 4 x 64-bit limbs */
void f_mul(const fctx* F, u256* o, const u256* a, const u256* b) {
  if (F == &FP) orc_fp_mul_count++;
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * F->inv;
    c = (u128)m * F->m.l[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * F->m.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  u256 r = {{t[0], t[1], t[2], t[3]}};
  u256 s; uint64_t br = u256_sub_raw(&s, &r, &F->m);
  *o = (t[4] || !br) ? s : r;
}
void f_sqr(const fctx* F, u256* o, const u256* a) { f_mul(F, o, a, a); }
void f_to_mont(const fctx* F, u256* o, const u256* a) { f_mul(F, o, a, &F->r2); }
void f_from_mont(const fctx* F, u256* o, const u256* a) { u256 one = {{1, 0, 0, 0}}; f_mul(F, o, a, &one); }
void f_pow(const fctx* F, u256* o, const u256* a, const u256* e) {
  u256 acc = F->r1, base = *a;
  for (int i = 255; i >= 0; i--) {
    f_sqr(F, &acc, &acc);
    if (u256_bit(e, i)) f_mul(F, &acc, &acc, &base);
  }
  *o = acc;
}
int f_inv(const fctx* F, u256* o, const u256* a) {
  if (u256_is_zero(a)) { memset(o, 0, sizeof *o); return 0; }
  f_pow(F, o, a, F == &FP ? &P_MINUS_2 : &R_MINUS_2);
  return 1;
}
/* big-endian byte string of any length, reduced mod m, to Montgomery form (Fq::from_be_bytes_mod_order) */
void f_reduce_be(const fctx* F, u256* o, const uint8_t* be, size_t n) {
  u256 acc; memset(&acc, 0, sizeof acc);
  u256 c256; { u256 v = {{256, 0, 0, 0}}; f_to_mont(F, &c256, &v); }
  for (size_t i = 0; i < n; i++) {
    u256 d = {{be[i], 0, 0, 0}}, dm;
    f_to_mont(F, &dm, &d);
    f_mul(F, &acc, &acc, &c256);
    f_add(F, &acc, &acc, &dm);
  }
  *o = acc;
}
/* p = 3 mod 4: candidate a^((p+1)/4) */
int fp_sqrt(fp* o, const fp* a) {
  fp c, c2;
  f_pow(&FP, &c, a, &P_PLUS_1_OVER_4);
  f_sqr(&FP, &c2, &c);
  if (u256_cmp(&c2, a) != 0) return 0;
  *o = c;
  return 1;
}

/* ---------------- Fp2 ---------------- */
void fp2_add(fp2* o, const fp2* a, const fp2* b) { f_add(&FP, &o->c0, &a->c0, &b->c0); f_add(&FP, &o->c1, &a->c1, &b->c1); }
void fp2_sub(fp2* o, const fp2* a, const fp2* b) { f_sub(&FP, &o->c0, &a->c0, &b->c0); f_sub(&FP, &o->c1, &a->c1, &b->c1); }
void fp2_neg(fp2* o, const fp2* a) { f_neg(&FP, &o->c0, &a->c0); f_neg(&FP, &o->c1, &a->c1); }
void fp2_conj(fp2* o, const fp2* a) { o->c0 = a->c0; f_neg(&FP, &o->c1, &a->c1); }
void fp2_mul(fp2* o, const fp2* a, const fp2* b) {
  fp aa, bb, s, t, u;
  f_mul(&FP, &aa, &a->c0, &b->c0);
  f_mul(&FP, &bb, &a->c1, &b->c1);
  f_add(&FP, &s, &a->c0, &a->c1);
  f_add(&FP, &t, &b->c0, &b->c1);
  f_mul(&FP, &u, &s, &t);
  f_sub(&FP, &u, &u, &aa);
  f_sub(&FP, &u, &u, &bb);
  f_sub(&FP, &o->c0, &aa, &bb);
  o->c1 = u;
}
void fp2_sqr(fp2* o, const fp2* a) {
  fp s, d, m;
  f_add(&FP, &s, &a->c0, &a->c1);
  f_sub(&FP, &d, &a->c0, &a->c1);
  f_mul(&FP, &m, &a->c0, &a->c1);
  f_mul(&FP, &o->c0, &s, &d);
  f_add(&FP, &o->c1, &m, &m);
}
void fp2_mul_fp(fp2* o, const fp2* a, const fp* b) { f_mul(&FP, &o->c0, &a->c0, b); f_mul(&FP, &o->c1, &a->c1, b); }
/* (a0 + a1 i)(9 + i) = (9 a0 - a1) + (9 a1 + a0) i */
void fp2_mul_xi(fp2* o, const fp2* a) {
  fp t0, t1, n0, n1;
  f_add(&FP, &t0, &a->c0, &a->c0); f_add(&FP, &t0, &t0, &t0); f_add(&FP, &t0, &t0, &t0); f_add(&FP, &t0, &t0, &a->c0); /* 9 a0 */
  f_add(&FP, &t1, &a->c1, &a->c1); f_add(&FP, &t1, &t1, &t1); f_add(&FP, &t1, &t1, &t1); f_add(&FP, &t1, &t1, &a->c1); /* 9 a1 */
  f_sub(&FP, &n0, &t0, &a->c1);
  f_add(&FP, &n1, &t1, &a->c0);
  o->c0 = n0; o->c1 = n1;
}
int fp2_inv(fp2* o, const fp2* a) {
  fp n, t, ni;
  f_sqr(&FP, &n, &a->c0); f_sqr(&FP, &t, &a->c1); f_add(&FP, &n, &n, &t);
  if (!f_inv(&FP, &ni, &n)) { memset(o, 0, sizeof *o); return 0; }
  f_mul(&FP, &o->c0, &a->c0, &ni);
  f_mul(&FP, &t, &a->c1, &ni);
  f_neg(&FP, &o->c1, &t);
  return 1;
}
int fp2_eq(const fp2* a, const fp2* b) { return u256_cmp(&a->c0, &b->c0) == 0 && u256_cmp(&a->c1, &b->c1) == 0; }
int fp2_is_zero(const fp2* a) { return u256_is_zero(&a->c0) && u256_is_zero(&a->c1); }
static void fp2_pow(fp2* o, const fp2* a, const u256* e) {
  fp2 acc; acc.c0 = FP.r1; memset(&acc.c1, 0, sizeof acc.c1);
  for (int i = 255; i >= 0; i--) {
    fp2_sqr(&acc, &acc);
    if (u256_bit(e, i)) fp2_mul(&acc, &acc, a);
  }
  *o = acc;
}
/* square root in Fp2 for p = 3 mod 4 (complex method, Adj--Rodriguez-Henriquez Alg. 9: constants (p-3)/4, (p-1)/2;
 * SURVEY.md C.2b notes Fq2::sqrt in `bn` has this shape).  Returns 0 if a is not a square.  Which of the two
 * roots is returned is unspecified: callers order the pair themselves. */
int fp2_sqrt(fp2* o, const fp2* a) {
  if (fp2_is_zero(a)) { *o = *a; return 1; }
  fp2 a1, alpha, a0, x0, t;
  fp2_pow(&a1, a, &P_MINUS_3_OVER_4);
  fp2_sqr(&t, &a1); fp2_mul(&alpha, &t, a);            /* alpha = a1^2 a */
  fp2_conj(&t, &alpha); fp2_mul(&a0, &t, &alpha);       /* a0 = alpha^p alpha = norm */
  fp2 minus_one; f_neg(&FP, &minus_one.c0, &FP.r1); memset(&minus_one.c1, 0, sizeof(fp));
  if (fp2_eq(&a0, &minus_one)) return 0;
  fp2_mul(&x0, &a1, a);
  if (fp2_eq(&alpha, &minus_one)) {
    /* x = i * x0 */
    fp2 r; f_neg(&FP, &r.c0, &x0.c1); r.c1 = x0.c0; *o = r;
  } else {
    fp2 b; fp2 one; one.c0 = FP.r1; memset(&one.c1, 0, sizeof(fp));
    fp2_add(&t, &alpha, &one);
    fp2_pow(&b, &t, &P_MINUS_1_OVER_2);
    fp2_mul(o, &b, &x0);
  }
  fp2_sqr(&t, o);
  return fp2_eq(&t, a);
}

/* ---------------- Fp6 = Fp2[v]/(v^3 - xi) ---------------- */
static void fp6_add(fp6* o, const fp6* a, const fp6* b) { fp2_add(&o->c0, &a->c0, &b->c0); fp2_add(&o->c1, &a->c1, &b->c1); fp2_add(&o->c2, &a->c2, &b->c2); }
static void fp6_sub(fp6* o, const fp6* a, const fp6* b) { fp2_sub(&o->c0, &a->c0, &b->c0); fp2_sub(&o->c1, &a->c1, &b->c1); fp2_sub(&o->c2, &a->c2, &b->c2); }
static void fp6_neg(fp6* o, const fp6* a) { fp2_neg(&o->c0, &a->c0); fp2_neg(&o->c1, &a->c1); fp2_neg(&o->c2, &a->c2); }
/* multiply by v: (c0, c1, c2) -> (xi c2, c0, c1) */
static void fp6_mul_v(fp6* o, const fp6* a) { fp2 t; fp2_mul_xi(&t, &a->c2); fp2 c0 = a->c0, c1 = a->c1; o->c0 = t; o->c1 = c0; o->c2 = c1; }
/* schoolbook (9 Fp2 products): deliberately the plain definition, so that it cannot share a mistake with the
 * Karatsuba/Toom forms used on the device */
void fp6_mul(fp6* o, const fp6* a, const fp6* b) {
  fp2 p00, p01, p02, p10, p11, p12, p20, p21, p22, t, r0, r1, r2;
  fp2_mul(&p00, &a->c0, &b->c0); fp2_mul(&p01, &a->c0, &b->c1); fp2_mul(&p02, &a->c0, &b->c2);
  fp2_mul(&p10, &a->c1, &b->c0); fp2_mul(&p11, &a->c1, &b->c1); fp2_mul(&p12, &a->c1, &b->c2);
  fp2_mul(&p20, &a->c2, &b->c0); fp2_mul(&p21, &a->c2, &b->c1); fp2_mul(&p22, &a->c2, &b->c2);
  fp2_add(&t, &p12, &p21); fp2_mul_xi(&t, &t); fp2_add(&r0, &p00, &t);            /* c0 = a0b0 + xi(a1b2 + a2b1) */
  fp2_mul_xi(&t, &p22); fp2_add(&r1, &p01, &p10); fp2_add(&r1, &r1, &t);          /* c1 = a0b1 + a1b0 + xi a2b2  */
  fp2_add(&r2, &p02, &p11); fp2_add(&r2, &r2, &p20);                              /* c2 = a0b2 + a1b1 + a2b0     */
  o->c0 = r0; o->c1 = r1; o->c2 = r2;
}
static int fp6_inv(fp6* o, const fp6* a) {
  /* standard: A = a0^2 - xi a1 a2, B = xi a2^2 - a0 a1, C = a1^2 - a0 a2, F = a0 A + xi (a2 B + a1 C) */
  fp2 A, B, C, t, F, Fi;
  fp2_sqr(&A, &a->c0); fp2_mul(&t, &a->c1, &a->c2); fp2_mul_xi(&t, &t); fp2_sub(&A, &A, &t);
  fp2_sqr(&B, &a->c2); fp2_mul_xi(&B, &B); fp2_mul(&t, &a->c0, &a->c1); fp2_sub(&B, &B, &t);
  fp2_sqr(&C, &a->c1); fp2_mul(&t, &a->c0, &a->c2); fp2_sub(&C, &C, &t);
  fp2 u; fp2_mul(&t, &a->c2, &B); fp2_mul(&u, &a->c1, &C); fp2_add(&t, &t, &u); fp2_mul_xi(&t, &t);
  fp2_mul(&F, &a->c0, &A); fp2_add(&F, &F, &t);
  if (!fp2_inv(&Fi, &F)) { memset(o, 0, sizeof *o); return 0; }
  fp2_mul(&o->c0, &A, &Fi); fp2_mul(&o->c1, &B, &Fi); fp2_mul(&o->c2, &C, &Fi);
  return 1;
}

/* ---------------- Fp12 = Fp6[w]/(w^2 - v) ---------------- */
void fp12_one(fp12* o) { memset(o, 0, sizeof *o); o->c0.c0.c0 = FP.r1; }
void fp12_mul(fp12* o, const fp12* a, const fp12* b) {
  fp6 aa, bb, ab, ba, t;
  fp6_mul(&aa, &a->c0, &b->c0);
  fp6_mul(&bb, &a->c1, &b->c1);
  fp6_mul(&ab, &a->c0, &b->c1);
  fp6_mul(&ba, &a->c1, &b->c0);
  fp6_mul_v(&t, &bb);
  fp6_add(&o->c0, &aa, &t);
  fp6_add(&o->c1, &ab, &ba);
}
void fp12_sqr(fp12* o, const fp12* a) { fp12 t = *a; fp12_mul(o, &t, &t); }
void fp12_conj(fp12* o, const fp12* a) { o->c0 = a->c0; fp6_neg(&o->c1, &a->c1); }
int fp12_inv(fp12* o, const fp12* a) {
  /* 1/(a0 + a1 w) = (a0 - a1 w)/(a0^2 - v a1^2) */
  fp6 t0, t1, d, di;
  fp6_mul(&t0, &a->c0, &a->c0);
  fp6_mul(&t1, &a->c1, &a->c1);
  fp6_mul_v(&t1, &t1);
  fp6_sub(&d, &t0, &t1);
  if (!fp6_inv(&di, &d)) { memset(o, 0, sizeof *o); return 0; }
  fp6 n1; fp6_neg(&n1, &a->c1);
  fp6 r0, r1;
  fp6_mul(&r0, &a->c0, &di);
  fp6_mul(&r1, &n1, &di);
  o->c0 = r0; o->c1 = r1;
  return 1;
}
int fp12_eq(const fp12* a, const fp12* b) { return memcmp(a, b, sizeof *a) == 0; }
int fp12_is_one(const fp12* a) { fp12 one; fp12_one(&one); return fp12_eq(a, &one); }

/* coefficient of w^k: k=0 c0.c0, 1 c1.c0, 2 c0.c1, 3 c1.c1, 4 c0.c2, 5 c1.c2 */
static fp2* fp12_coeff(fp12* a, int k) {
  fp6* h = (k & 1) ? &a->c1 : &a->c0;
  return (k >> 1) == 0 ? &h->c0 : (k >> 1) == 1 ? &h->c1 : &h->c2;
}
/* x -> x^p : conjugate every Fp2 coefficient and multiply the w^k coefficient by xi^(k(p-1)/6) */
static void fp12_frob1(fp12* o, const fp12* a) {
  fp12 r = *a;
  for (int k = 0; k < 6; k++) {
    fp2* c = fp12_coeff(&r, k);
    fp2 t; fp2_conj(&t, c);
    fp2_mul(c, &t, &FROB_G[k]);
  }
  *o = r;
}
void fp12_frob(fp12* o, const fp12* a, int power) {
  fp12 r = *a;
  for (int i = 0; i < power; i++) fp12_frob1(&r, &r);
  *o = r;
}
/* sparse multiplication by ell_0 + ell_vw (v w) + ell_vv v^2 -- by definition (dense product with the sparse element) */
void fp12_mul_by_024(fp12* o, const fp12* a, const fp2* ell_0, const fp2* ell_vw, const fp2* ell_vv) {
  fp12 l; memset(&l, 0, sizeof l);
  l.c0.c0 = *ell_0; l.c0.c2 = *ell_vv; l.c1.c1 = *ell_vw;
  fp12_mul(o, a, &l);
}
/* valid on the cyclotomic subgroup; the oracle simply squares (exact same value there) */
void fp12_cyclo_sqr(fp12* o, const fp12* a) { fp12_sqr(o, a); }

/* ---------------- init ---------------- */
static uint64_t mont_inv64(uint64_t m0) {
  uint64_t x = 1;
  for (int i = 0; i < 6; i++) x *= 2 - m0 * x; /* Newton: x = m0^-1 mod 2^64 */
  return (uint64_t)(0 - x);
}
static void ctx_init(fctx* F, const u256* m) {
  F->m = *m;
  F->inv = mont_inv64(m->l[0]);
  u256 x = {{1, 0, 0, 0}};
  /* 2^256 mod m, then 2^512 mod m, by repeated modular doubling (r1/r2 not needed by f_add) */
  for (int i = 0; i < 512; i++) {
    f_add(F, &x, &x, &x);
    if (i == 255) F->r1 = x;
  }
  F->r2 = x;
}
void orc_init(void) {
  if (g_init) return;
  ctx_init(&FP, &P_MOD);
  ctx_init(&FR, &R_MOD);
  u256 one = {{1, 0, 0, 0}}, two = {{2, 0, 0, 0}}, three = {{3, 0, 0, 0}}, t;
  u256_sub_raw(&t, &P_MOD, &one); u256_div_small(&P_MINUS_1_OVER_6, &t, 6);
  P_MINUS_1_OVER_2 = t; u256_shr1(&P_MINUS_1_OVER_2);
  u256_add_raw(&t, &P_MOD, &one); P_PLUS_1_OVER_4 = t; u256_shr1(&P_PLUS_1_OVER_4); u256_shr1(&P_PLUS_1_OVER_4);
  u256_sub_raw(&t, &P_MOD, &three); P_MINUS_3_OVER_4 = t; u256_shr1(&P_MINUS_3_OVER_4); u256_shr1(&P_MINUS_3_OVER_4);
  u256_sub_raw(&P_MINUS_2, &P_MOD, &two);
  u256_sub_raw(&R_MINUS_2, &R_MOD, &two);
  g_init = 1; /* field ops usable from here on */
  u256 nine = {{9, 0, 0, 0}};
  f_to_mont(&FP, &XI.c0, &nine); XI.c1 = FP.r1;
  fp2 xi_inv; fp2_inv(&xi_inv, &XI);
  fp three_m; f_to_mont(&FP, &three_m, &three);
  fp2_mul_fp(&TWIST_B, &xi_inv, &three_m);
  fp two_m; f_to_mont(&FP, &two_m, &two); f_inv(&FP, &FP_TWO_INV, &two_m);
  fp2 g; fp2_pow(&g, &XI, &P_MINUS_1_OVER_6);
  FROB_G[0].c0 = FP.r1; memset(&FROB_G[0].c1, 0, sizeof(fp));
  for (int k = 1; k < 6; k++) fp2_mul(&FROB_G[k], &FROB_G[k - 1], &g);
}
