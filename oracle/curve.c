/* oracle/curve.c -- G1 / G2 group law and the optimal-ate pairing of BN254, as the `bn` crate computes it.
 * TEST INFRASTRUCTURE (see oracle.h).
 *
 * Call sites restated: bn::pairing (reference verifier/src/groth16/verify.rs:70), bn::pairing_batch
 * (groth16/verify.rs:73, plonk/kzg.rs:180), AffineG1 * Fr / + (groth16/verify.rs:58-62), AffineG2::new
 * (converter.rs:152: on-curve then r-torsion by full scalar multiplication -- SURVEY.md C.2b).
 * Algorithm (published zcash/libff alt_bn128 design that substrate-bn inherits):
 *   precompute(Q): homogeneous-projective doubling / mixed-addition steps over the NAF of 6u+2, each emitting
 *                  (ell_0, ell_VW, ell_VV); then the two Frobenius steps with pi(Q), -pi^2(Q);
 *   miller_loop_batch: f <- f^2 once per NAF digit for all pairs, f <- f * (ell_0 + ell_VW*yP (v w) + ell_VV*xP v^2);
 *   final_exponentiation: f^(p^6-1)(p^2+1) then the exp_by_neg_z chain.
 */
#include "oracle.h"
#include <string.h>
#include <stdlib.h>
#include <stdio.h>

extern fp2 TWIST_B;
extern fp FP_TWO_INV;

static const u256 R_MOD_MINUS_1 = {{0x43e1f593f0000000ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
#define BN_U 4965661367192848881ull /* SURVEY.md Appendix B.1 */

static fp FP_B;  /* 3 */
static fp2 XI_C; /* 9+i */
static fp2 PSI_X, PSI_Y; /* xi^((p-1)/3), xi^((p-1)/2): pi(x,y) = (conj(x) PSI_X, conj(y) PSI_Y) on the twist */
static int8_t NAF[70]; static int NAF_LEN;
static int c_init = 0;

static void fp_set_u64(fp* o, uint64_t v) { u256 t = {{v, 0, 0, 0}}; f_to_mont(&FP, o, &t); }
static void fp2_zero(fp2* o) { memset(o, 0, sizeof *o); }
static void fp2_one(fp2* o) { o->c0 = FP.r1; memset(&o->c1, 0, sizeof(fp)); }

static void curve_init(void) {
  if (c_init) return;
  orc_init();
  fp_set_u64(&FP_B, 3);
  fp_set_u64(&XI_C.c0, 9); XI_C.c1 = FP.r1;
  /* xi^((p-1)/3) and xi^((p-1)/2) via the Frobenius of w^2 and w^3: (w^k)^p = w^k * xi^(k(p-1)/6) */
  fp12 w2, w3, t; memset(&w2, 0, sizeof w2); memset(&w3, 0, sizeof w3);
  w2.c0.c1.c0 = FP.r1; /* v = w^2 */
  w3.c1.c1.c0 = FP.r1; /* v w = w^3 */
  fp12_frob(&t, &w2, 1); PSI_X = t.c0.c1;
  fp12_frob(&t, &w3, 1); PSI_Y = t.c1.c1;
  /* NAF of 6u+2 (65 bits) */
  unsigned __int128 n = (unsigned __int128)6 * BN_U + 2;
  int len = 0;
  while (n) {
    int d = 0;
    if (n & 1) { d = 2 - (int)(n & 3); if (d < 0) n += 1; else n -= 1; }
    NAF[len++] = (int8_t)d;
    n >>= 1;
  }
  NAF_LEN = len;
  c_init = 1;
}

/* ======================= G1 : y^2 = x^3 + 3, Jacobian ======================= */
int g1_on_curve(const fp* x, const fp* y) {
  curve_init();
  fp l, r;
  f_sqr(&FP, &l, y);
  f_sqr(&FP, &r, x); f_mul(&FP, &r, &r, x); f_add(&FP, &r, &r, &FP_B);
  return u256_cmp(&l, &r) == 0;
}
void g1_generator(g1a* o) { curve_init(); fp_set_u64(&o->x, 1); fp_set_u64(&o->y, 2); o->inf = 0; }
void g1_from_affine(g1j* o, const g1a* a) {
  curve_init();
  if (a->inf) { memset(o, 0, sizeof *o); o->y = FP.r1; return; }
  o->x = a->x; o->y = a->y; o->z = FP.r1;
}
void g1_to_affine(g1a* o, const g1j* a) {
  if (u256_is_zero(&a->z)) { memset(o, 0, sizeof *o); o->inf = 1; return; }
  fp zi, zi2, zi3;
  f_inv(&FP, &zi, &a->z); f_sqr(&FP, &zi2, &zi); f_mul(&FP, &zi3, &zi2, &zi);
  f_mul(&FP, &o->x, &a->x, &zi2); f_mul(&FP, &o->y, &a->y, &zi3); o->inf = 0;
}
void g1_neg_affine(g1a* o, const g1a* a) { *o = *a; f_neg(&FP, &o->y, &a->y); }
void g1_double(g1j* o, const g1j* p) {
  if (u256_is_zero(&p->z)) { *o = *p; return; }
  fp a, b, c, d, e, f, t, x3, y3, z3;
  f_sqr(&FP, &a, &p->x); f_sqr(&FP, &b, &p->y); f_sqr(&FP, &c, &b);
  f_add(&FP, &t, &p->x, &b); f_sqr(&FP, &t, &t); f_sub(&FP, &t, &t, &a); f_sub(&FP, &t, &t, &c); f_add(&FP, &d, &t, &t);
  f_add(&FP, &e, &a, &a); f_add(&FP, &e, &e, &a);
  f_sqr(&FP, &f, &e);
  f_sub(&FP, &x3, &f, &d); f_sub(&FP, &x3, &x3, &d);
  f_sub(&FP, &t, &d, &x3); f_mul(&FP, &y3, &e, &t);
  fp c8; f_add(&FP, &c8, &c, &c); f_add(&FP, &c8, &c8, &c8); f_add(&FP, &c8, &c8, &c8);
  f_sub(&FP, &y3, &y3, &c8);
  f_mul(&FP, &z3, &p->y, &p->z); f_add(&FP, &z3, &z3, &z3);
  o->x = x3; o->y = y3; o->z = z3;
}
void g1_add(g1j* o, const g1j* p, const g1j* q) {
  if (u256_is_zero(&p->z)) { *o = *q; return; }
  if (u256_is_zero(&q->z)) { *o = *p; return; }
  fp z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
  f_sqr(&FP, &z1z1, &p->z); f_sqr(&FP, &z2z2, &q->z);
  f_mul(&FP, &u1, &p->x, &z2z2); f_mul(&FP, &u2, &q->x, &z1z1);
  f_mul(&FP, &s1, &p->y, &q->z); f_mul(&FP, &s1, &s1, &z2z2);
  f_mul(&FP, &s2, &q->y, &p->z); f_mul(&FP, &s2, &s2, &z1z1);
  if (u256_cmp(&u1, &u2) == 0) {
    if (u256_cmp(&s1, &s2) == 0) { g1_double(o, p); return; }
    memset(o, 0, sizeof *o); o->y = FP.r1; return;
  }
  f_sub(&FP, &h, &u2, &u1);
  f_add(&FP, &i, &h, &h); f_sqr(&FP, &i, &i);
  f_mul(&FP, &j, &h, &i);
  f_sub(&FP, &r, &s2, &s1); f_add(&FP, &r, &r, &r);
  f_mul(&FP, &v, &u1, &i);
  f_sqr(&FP, &x3, &r); f_sub(&FP, &x3, &x3, &j); f_sub(&FP, &x3, &x3, &v); f_sub(&FP, &x3, &x3, &v);
  f_sub(&FP, &t, &v, &x3); f_mul(&FP, &y3, &r, &t);
  f_mul(&FP, &t, &s1, &j); f_add(&FP, &t, &t, &t); f_sub(&FP, &y3, &y3, &t);
  f_add(&FP, &z3, &p->z, &q->z); f_sqr(&FP, &z3, &z3); f_sub(&FP, &z3, &z3, &z1z1); f_sub(&FP, &z3, &z3, &z2z2);
  f_mul(&FP, &z3, &z3, &h);
  o->x = x3; o->y = y3; o->z = z3;
}
/* left-to-right double-and-add over all 256 bits of k, k used as an integer without reduction (bn's G * Fr walks
 * the stored 256-bit value bit by bit) */
void g1_mul(g1j* o, const g1j* a, const u256* k) {
  g1j acc; memset(&acc, 0, sizeof acc); acc.y = FP.r1;
  for (int i = 255; i >= 0; i--) {
    g1_double(&acc, &acc);
    if (u256_bit(k, i)) g1_add(&acc, &acc, a);
  }
  *o = acc;
}

/* ======================= G2 : y^2 = x^3 + 3/xi over Fp2, Jacobian ======================= */
int g2_on_curve(const fp2* x, const fp2* y) {
  curve_init();
  fp2 l, r;
  fp2_sqr(&l, y);
  fp2_sqr(&r, x); fp2_mul(&r, &r, x); fp2_add(&r, &r, &TWIST_B);
  return fp2_eq(&l, &r);
}
void g2_generator(g2a* o) {
  curve_init();
  static const char* hex[4] = { /* x.c0, x.c1, y.c0, y.c1 (decimal there) */
    "1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed",
    "198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2",
    "12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa",
    "090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b"};
  fp* dst[4] = {&o->x.c0, &o->x.c1, &o->y.c0, &o->y.c1};
  for (int k = 0; k < 4; k++) {
    uint8_t b[32];
    for (int i = 0; i < 32; i++) { unsigned v; sscanf(hex[k] + 2 * i, "%2x", &v); b[i] = (uint8_t)v; }
    u256 t; u256_from_be(&t, b); f_to_mont(&FP, dst[k], &t);
  }
  o->inf = 0;
}
void g2_from_affine(g2j* o, const g2a* a) {
  curve_init();
  if (a->inf) { memset(o, 0, sizeof *o); fp2_one(&o->y); return; }
  o->x = a->x; o->y = a->y; fp2_one(&o->z);
}
void g2_to_affine(g2a* o, const g2j* a) {
  if (fp2_is_zero(&a->z)) { memset(o, 0, sizeof *o); o->inf = 1; return; }
  fp2 zi, zi2, zi3;
  fp2_inv(&zi, &a->z); fp2_sqr(&zi2, &zi); fp2_mul(&zi3, &zi2, &zi);
  fp2_mul(&o->x, &a->x, &zi2); fp2_mul(&o->y, &a->y, &zi3); o->inf = 0;
}
void g2_neg_affine(g2a* o, const g2a* a) { *o = *a; fp2_neg(&o->y, &a->y); }
void g2_double(g2j* o, const g2j* p) {
  if (fp2_is_zero(&p->z)) { *o = *p; return; }
  fp2 a, b, c, d, e, f, t, x3, y3, z3, c8;
  fp2_sqr(&a, &p->x); fp2_sqr(&b, &p->y); fp2_sqr(&c, &b);
  fp2_add(&t, &p->x, &b); fp2_sqr(&t, &t); fp2_sub(&t, &t, &a); fp2_sub(&t, &t, &c); fp2_add(&d, &t, &t);
  fp2_add(&e, &a, &a); fp2_add(&e, &e, &a);
  fp2_sqr(&f, &e);
  fp2_sub(&x3, &f, &d); fp2_sub(&x3, &x3, &d);
  fp2_sub(&t, &d, &x3); fp2_mul(&y3, &e, &t);
  fp2_add(&c8, &c, &c); fp2_add(&c8, &c8, &c8); fp2_add(&c8, &c8, &c8);
  fp2_sub(&y3, &y3, &c8);
  fp2_mul(&z3, &p->y, &p->z); fp2_add(&z3, &z3, &z3);
  o->x = x3; o->y = y3; o->z = z3;
}
void g2_add(g2j* o, const g2j* p, const g2j* q) {
  if (fp2_is_zero(&p->z)) { *o = *q; return; }
  if (fp2_is_zero(&q->z)) { *o = *p; return; }
  fp2 z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
  fp2_sqr(&z1z1, &p->z); fp2_sqr(&z2z2, &q->z);
  fp2_mul(&u1, &p->x, &z2z2); fp2_mul(&u2, &q->x, &z1z1);
  fp2_mul(&s1, &p->y, &q->z); fp2_mul(&s1, &s1, &z2z2);
  fp2_mul(&s2, &q->y, &p->z); fp2_mul(&s2, &s2, &z1z1);
  if (fp2_eq(&u1, &u2)) {
    if (fp2_eq(&s1, &s2)) { g2_double(o, p); return; }
    memset(o, 0, sizeof *o); fp2_one(&o->y); return;
  }
  fp2_sub(&h, &u2, &u1);
  fp2_add(&i, &h, &h); fp2_sqr(&i, &i);
  fp2_mul(&j, &h, &i);
  fp2_sub(&r, &s2, &s1); fp2_add(&r, &r, &r);
  fp2_mul(&v, &u1, &i);
  fp2_sqr(&x3, &r); fp2_sub(&x3, &x3, &j); fp2_sub(&x3, &x3, &v); fp2_sub(&x3, &x3, &v);
  fp2_sub(&t, &v, &x3); fp2_mul(&y3, &r, &t);
  fp2_mul(&t, &s1, &j); fp2_add(&t, &t, &t); fp2_sub(&y3, &y3, &t);
  fp2_add(&z3, &p->z, &q->z); fp2_sqr(&z3, &z3); fp2_sub(&z3, &z3, &z1z1); fp2_sub(&z3, &z3, &z2z2);
  fp2_mul(&z3, &z3, &h);
  o->x = x3; o->y = y3; o->z = z3;
}
void g2_mul(g2j* o, const g2j* a, const u256* k) {
  g2j acc; memset(&acc, 0, sizeof acc); fp2_one(&acc.y);
  for (int i = 255; i >= 0; i--) {
    g2_double(&acc, &acc);
    if (u256_bit(k, i)) g2_add(&acc, &acc, a);
  }
  *o = acc;
}
/* bn's AffineG2::new order check: [r-1]Q + Q must be the identity */
int g2_in_subgroup_naive(const g2a* q) {
  g2j j, t;
  g2_from_affine(&j, q);
  g2_mul(&t, &j, &R_MOD_MINUS_1);
  g2_add(&t, &t, &j);
  return fp2_is_zero(&t.z);
}

/* ======================= pairing ======================= */
typedef struct { fp2 ell_0, ell_vw, ell_vv; } ell_coeffs;
typedef struct { fp2 x, y, z; } g2h; /* homogeneous projective: (X/Z, Y/Z) */
#define MAX_COEFFS 104

static void doubling_step(g2h* r, ell_coeffs* c) {
  fp2 a, b, cc, d, e, f, g, h, i, j, e2, t;
  fp2_mul(&a, &r->x, &r->y); fp2_mul_fp(&a, &a, &FP_TWO_INV);
  fp2_sqr(&b, &r->y);
  fp2_sqr(&cc, &r->z);
  fp2_add(&d, &cc, &cc); fp2_add(&d, &d, &cc);
  fp2_mul(&e, &TWIST_B, &d);
  fp2_add(&f, &e, &e); fp2_add(&f, &f, &e);
  fp2_add(&g, &b, &f); fp2_mul_fp(&g, &g, &FP_TWO_INV);
  fp2_add(&h, &r->y, &r->z); fp2_sqr(&h, &h); fp2_add(&t, &b, &cc); fp2_sub(&h, &h, &t);
  fp2_sub(&i, &e, &b);
  fp2_sqr(&j, &r->x);
  fp2_sqr(&e2, &e);
  fp2_sub(&t, &b, &f); fp2_mul(&r->x, &a, &t);
  fp2_sqr(&t, &g); fp2 e23; fp2_add(&e23, &e2, &e2); fp2_add(&e23, &e23, &e2); fp2_sub(&r->y, &t, &e23);
  fp2_mul(&r->z, &b, &h);
  fp2_mul_xi(&c->ell_0, &i);
  fp2_neg(&c->ell_vw, &h);
  fp2_add(&c->ell_vv, &j, &j); fp2_add(&c->ell_vv, &c->ell_vv, &j);
}
static void mixed_addition_step(g2h* r, const fp2* x2, const fp2* y2, ell_coeffs* c) {
  fp2 d, e, f, g, h, i, j, t, u;
  fp2_mul(&t, x2, &r->z); fp2_sub(&d, &r->x, &t);
  fp2_mul(&t, y2, &r->z); fp2_sub(&e, &r->y, &t);
  fp2_sqr(&f, &d);
  fp2_sqr(&g, &e);
  fp2_mul(&h, &d, &f);
  fp2_mul(&i, &r->x, &f);
  fp2_mul(&t, &r->z, &g); fp2_add(&j, &h, &t); fp2_sub(&j, &j, &i); fp2_sub(&j, &j, &i);
  fp2_mul(&t, &e, x2); fp2_mul(&u, &d, y2); fp2_sub(&t, &t, &u); fp2_mul_xi(&c->ell_0, &t);
  fp2_neg(&c->ell_vv, &e);
  c->ell_vw = d;
  fp2 x3, y3, z3;
  fp2_mul(&x3, &d, &j);
  fp2_sub(&t, &i, &j); fp2_mul(&y3, &e, &t); fp2_mul(&u, &h, &r->y); fp2_sub(&y3, &y3, &u);
  fp2_mul(&z3, &r->z, &h);
  r->x = x3; r->y = y3; r->z = z3;
}
/* pi(Q) = (conj(x) xi^((p-1)/3), conj(y) xi^((p-1)/2)) */
static void g2_mul_by_q(fp2* ox, fp2* oy, const fp2* x, const fp2* y) {
  fp2 t;
  fp2_conj(&t, x); fp2_mul(ox, &t, &PSI_X);
  fp2_conj(&t, y); fp2_mul(oy, &t, &PSI_Y);
}
static int g2_precompute(ell_coeffs* cs, const g2a* q) {
  g2h r; r.x = q->x; r.y = q->y; fp2_one(&r.z);
  fp2 ny; fp2_neg(&ny, &q->y);
  int n = 0;
  for (int i = NAF_LEN - 2; i >= 0; i--) {
    doubling_step(&r, &cs[n++]);
    if (NAF[i] == 1) mixed_addition_step(&r, &q->x, &q->y, &cs[n++]);
    else if (NAF[i] == -1) mixed_addition_step(&r, &q->x, &ny, &cs[n++]);
  }
  fp2 q1x, q1y, q2x, q2y;
  g2_mul_by_q(&q1x, &q1y, &q->x, &q->y);
  g2_mul_by_q(&q2x, &q2y, &q1x, &q1y);
  fp2_neg(&q2y, &q2y);
  mixed_addition_step(&r, &q1x, &q1y, &cs[n++]);
  mixed_addition_step(&r, &q2x, &q2y, &cs[n++]);
  return n;
}
static void ell_eval_mul(fp12* f, const ell_coeffs* c, const g1a* p) {
  fp2 vw, vv;
  fp2_mul_fp(&vw, &c->ell_vw, &p->y);
  fp2_mul_fp(&vv, &c->ell_vv, &p->x);
  fp12_mul_by_024(f, f, &c->ell_0, &vw, &vv);
}
void miller_loop_batch(fp12* f, const g1a* ps, const g2a* qs, int n) {
  curve_init();
  ell_coeffs* tab = (ell_coeffs*)malloc(sizeof(ell_coeffs) * MAX_COEFFS * (n > 0 ? n : 1));
  int* live = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
  for (int k = 0; k < n; k++) {
    live[k] = !(ps[k].inf || qs[k].inf); /* bn::pairing_batch skips pairs with an identity operand */
    if (live[k]) g2_precompute(tab + (size_t)k * MAX_COEFFS, &qs[k]);
  }
  fp12_one(f);
  int idx = 0;
  for (int i = NAF_LEN - 2; i >= 0; i--) {
    fp12_sqr(f, f);
    for (int k = 0; k < n; k++) if (live[k]) ell_eval_mul(f, &tab[(size_t)k * MAX_COEFFS + idx], &ps[k]);
    idx++;
    if (NAF[i] != 0) {
      for (int k = 0; k < n; k++) if (live[k]) ell_eval_mul(f, &tab[(size_t)k * MAX_COEFFS + idx], &ps[k]);
      idx++;
    }
  }
  for (int s = 0; s < 2; s++) {
    for (int k = 0; k < n; k++) if (live[k]) ell_eval_mul(f, &tab[(size_t)k * MAX_COEFFS + idx], &ps[k]);
    idx++;
  }
  free(tab); free(live);
}
static void exp_by_neg_z(fp12* o, const fp12* a) {
  /* a^u by square-and-multiply on the cyclotomic subgroup, then conjugate (= inverse there) */
  fp12 acc = *a;
  for (int i = 61; i >= 0; i--) { /* u has 63 bits; top bit consumed by acc = a */
    fp12_cyclo_sqr(&acc, &acc);
    if ((BN_U >> i) & 1) fp12_mul(&acc, &acc, a);
  }
  fp12_conj(o, &acc);
}
static void final_exp_first_chunk(fp12* o, const fp12* f) {
  fp12 b, a, c, d;
  fp12_inv(&b, f);          /* the Miller loop cannot produce zero */
  fp12_conj(&a, f);
  fp12_mul(&c, &a, &b);     /* f^(p^6-1) */
  fp12_frob(&d, &c, 2);
  fp12_mul(o, &d, &c);      /* ^(p^2+1) */
}
void final_exponentiation(fp12* o, const fp12* fin) {
  curve_init();
  fp12 elt, A, B, C, D, E, F, G, H, I, J, K, L, M, N, O, P, Q, R, S, T, U;
  final_exp_first_chunk(&elt, fin);
  exp_by_neg_z(&A, &elt);
  fp12_cyclo_sqr(&B, &A);
  fp12_cyclo_sqr(&C, &B);
  fp12_mul(&D, &C, &B);
  exp_by_neg_z(&E, &D);
  fp12_cyclo_sqr(&F, &E);
  exp_by_neg_z(&G, &F);
  fp12_conj(&H, &D);
  fp12_conj(&I, &G);
  fp12_mul(&J, &I, &E);
  fp12_mul(&K, &J, &H);
  fp12_mul(&L, &K, &B);
  fp12_mul(&M, &K, &E);
  fp12_mul(&N, &M, &elt);
  fp12_frob(&O, &L, 1);
  fp12_mul(&P, &O, &N);
  fp12_frob(&Q, &K, 2);
  fp12_mul(&R, &Q, &P);
  fp12_conj(&S, &elt);
  fp12_mul(&T, &S, &L);
  fp12_frob(&U, &T, 3);
  fp12_mul(o, &U, &R);
}
/* cross-check only: hard part by plain exponentiation with (p^4 - p^2 + 1)/r, given as a 762-bit big-endian constant */
void final_exponentiation_plain(fp12* o, const fp12* fin) {
  curve_init();
  /* (p^4 - p^2 + 1)/r, computed offline (tests/test_oracle_arith.py recomputes it with Python integers) */
  static const char* HARD_HEX =
      "1baaa710b0759ad331ec15183177faf6c0eb522d5b122784e529a5861876f6b3b1b1355d189227d79581e16f3fd90c66b887d56d5095f23aaa441e3954bcf8adcc7b44c87cdbacff1154e7e1da014fd5abf5cc4f49c36d4e81bb482ccdf42b1";
  fp12 elt, acc;
  final_exp_first_chunk(&elt, fin);
  fp12_one(&acc);
  size_t nh = strlen(HARD_HEX);
  for (size_t i = 0; i < nh; i++) {
    char ch = HARD_HEX[i];
    int v = (ch >= '0' && ch <= '9') ? ch - '0' : ch - 'a' + 10;
    for (int b = 3; b >= 0; b--) {
      fp12_sqr(&acc, &acc);
      if ((v >> b) & 1) fp12_mul(&acc, &acc, &elt);
    }
  }
  *o = acc;
}
void pairing_batch(fp12* o, const g1a* ps, const g2a* qs, int n) {
  fp12 f;
  miller_loop_batch(&f, ps, qs, n);
  final_exponentiation(o, &f);
}
