/* oracle/protocol.c -- gnark byte codecs, Groth16 and PlonK verification glue, SHA-256 helpers.
 * TEST INFRASTRUCTURE (see oracle.h).  Each function cites the reference lines it restates
 * (paths relative to /root/reference/verifier/src). */
#include "oracle.h"
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const u256 P_HALF = {{0x9e10460b6c3e7ea3ull, 0xcbc0b548b438e546ull, 0xdc2822db40c0ac2eull, 0x183227397098d014ull}}; /* (p-1)/2 */

/* ======================= SHA-256 (sha2 0.10.8 in the reference; FIPS 180-4) ======================= */
static const uint32_t K256[64] = {
  0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,
  0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
  0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,
  0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
  0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,
  0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t h[8], const uint8_t* b) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = (uint32_t)b[4 * i] << 24 | (uint32_t)b[4 * i + 1] << 16 | (uint32_t)b[4 * i + 2] << 8 | b[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25), ch = (e & f) ^ (~e & g);
    uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
    uint32_t S0 = ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22), mj = (a & bb) ^ (a & c) ^ (bb & c);
    uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
  }
  h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
void sha256_init(sha256_ctx* c) {
  static const uint32_t iv[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
  memcpy(c->h, iv, sizeof iv); c->len = 0;
}
void sha256_update(sha256_ctx* c, const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; i++) {
    c->buf[c->len & 63] = d[i];
    c->len++;
    if ((c->len & 63) == 0) sha256_block(c->h, c->buf);
  }
}
void sha256_final(sha256_ctx* c, uint8_t out[32]) {
  uint64_t bits = c->len * 8;
  uint8_t pad = 0x80; sha256_update(c, &pad, 1);
  pad = 0; while ((c->len & 63) != 56) sha256_update(c, &pad, 1);
  uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
  sha256_update(c, lb, 8);
  for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(c->h[i] >> 24); out[4 * i + 1] = (uint8_t)(c->h[i] >> 16); out[4 * i + 2] = (uint8_t)(c->h[i] >> 8); out[4 * i + 3] = (uint8_t)c->h[i]; }
}

/* ======================= gnark point codecs (converter.rs) ======================= */
/* Fq::from_slice: big-endian, reject >= p */
static int fp_from_slice(fp* o, const uint8_t* b32) {
  u256 t; u256_from_be(&t, b32);
  if (u256_cmp(&t, &FP.m) >= 0) return 0;
  f_to_mont(&FP, o, &t);
  return 1;
}
static void fp_to_be(uint8_t* b32, const fp* a) { u256 t; f_from_mont(&FP, &t, a); u256_to_be(b32, &t); }
static int fp_is_large(const fp* a) { u256 t; f_from_mont(&FP, &t, a); return u256_cmp(&t, &P_HALF) > 0; }
static int fp_canon_cmp(const fp* a, const fp* b) { u256 x, y; f_from_mont(&FP, &x, a); f_from_mont(&FP, &y, b); return u256_cmp(&x, &y); }

/* converter.rs:78-88 */
int dec_g1_uncompressed(g1a* o, const uint8_t* b) {
  orc_init();
  if (!fp_from_slice(&o->x, b) || !fp_from_slice(&o->y, b + 32)) return ORC_ERR_NOT_MEMBER;
  o->inf = 0;
  if (!g1_on_curve(&o->x, &o->y)) return ORC_ERR_NOT_ON_CURVE;
  return ORC_ACCEPT;
}
/* converter.rs:135-153: x.c1 | x.c0 | y.c1 | y.c0, AffineG2::new = on-curve then subgroup */
int dec_g2_uncompressed(g2a* o, const uint8_t* b) {
  orc_init();
  if (!fp_from_slice(&o->x.c1, b) || !fp_from_slice(&o->x.c0, b + 32) || !fp_from_slice(&o->y.c1, b + 64) || !fp_from_slice(&o->y.c0, b + 96))
    return ORC_ERR_NOT_MEMBER;
  o->inf = 0;
  if (!g2_on_curve(&o->x, &o->y)) return ORC_ERR_NOT_ON_CURVE;
  if (!g2_in_subgroup_naive(o)) return ORC_ERR_NOT_IN_SUBGROUP;
  return ORC_ACCEPT;
}
/* converter.rs:23-43 deserialize_with_flags: returns flag (0b10,0b11,0b01) in *flag, x reduced mod p; ORC_ERR_MALFORMED on 0b00
 * (constants.rs:24 panics) or a non-zero infinity encoding (InvalidPoint -> unwrap panic) */
static int deserialize_with_flags(fp* x, int* flag, const uint8_t* b32) {
  int m = b32[0] >> 6;
  if (m == 1) {
    if (b32[0] & 0x3f) return ORC_ERR_MALFORMED;
    for (int i = 1; i < 32; i++) if (b32[i]) return ORC_ERR_MALFORMED;
    memset(x, 0, sizeof *x); *flag = 1; return ORC_ACCEPT;
  }
  if (m == 0) return ORC_ERR_MALFORMED;
  uint8_t t[32]; memcpy(t, b32, 32); t[0] &= 0x3f;
  f_reduce_be(&FP, x, t, 32); /* from_be_bytes_mod_order: silently reduced (converter.rs:39) */
  *flag = m;
  return ORC_ACCEPT;
}
/* converter.rs:62-76.  get_ys_from_x_unchecked returns (smaller y, larger y), so flag 10 -> smaller,
 * 11 -> larger, and the infinity flag falls through with x = 0 (no root of 3 exists mod p -> InvalidPoint -> panic). */
int dec_g1_compressed_unchecked(g1a* o, const uint8_t* b32) {
  orc_init();
  fp x, y, ny, rhs, three; int flag;
  int st = deserialize_with_flags(&x, &flag, b32);
  if (st != ORC_ACCEPT) return st;
  u256 t3 = {{3, 0, 0, 0}}; f_to_mont(&FP, &three, &t3);
  f_sqr(&FP, &rhs, &x); f_mul(&FP, &rhs, &rhs, &x); f_add(&FP, &rhs, &rhs, &three);
  if (!fp_sqrt(&y, &rhs)) return ORC_ERR_MALFORMED;
  f_neg(&FP, &ny, &y);
  if (fp_canon_cmp(&y, &ny) > 0) { fp s = y; y = ny; ny = s; } /* (y, neg_y) ordered smaller first */
  o->x = x; o->inf = 0;
  o->y = (flag == 3) ? ny : y;
  return ORC_ACCEPT;
}
/* converter.rs:113-133.  mode ORC_MODE_REFERENCE: roots ordered by c0 only, flag 10 -> first.
 * mode ORC_MODE_GNARK: flag 10 -> lexicographically smallest (c1 first, c0 on a tie), 11 -> largest. */
int dec_g2_compressed_unchecked(g2a* o, const uint8_t* b64, int mode) {
  orc_init();
  extern fp2 TWIST_B;
  fp x1, x0; int flag;
  int st = deserialize_with_flags(&x1, &flag, b64);
  if (st != ORC_ACCEPT) return st;
  f_reduce_be(&FP, &x0, b64 + 32, 32);
  if (flag == 1) { g2_generator(o); return ORC_ACCEPT; } /* AffineG2::one() quirk, converter.rs:122-124 */
  fp2 x, rhs, y, ny;
  x.c0 = x0; x.c1 = x1;
  fp2_sqr(&rhs, &x); fp2_mul(&rhs, &rhs, &x); fp2_add(&rhs, &rhs, &TWIST_B);
  if (!fp2_sqrt(&y, &rhs)) return ORC_ERR_MALFORMED;
  fp2_neg(&ny, &y);
  int y_first;
  if (mode == ORC_MODE_REFERENCE) {
    y_first = fp_canon_cmp(&y.c0, &ny.c0) < 0;
  } else {
    int y_large = u256_is_zero(&y.c1) ? fp_is_large(&y.c0) : fp_is_large(&y.c1);
    y_first = !y_large;
  }
  const fp2* first = y_first ? &y : &ny;
  const fp2* second = y_first ? &ny : &y;
  o->x = x; o->inf = 0;
  o->y = (flag == 2) ? *first : *second;
  return ORC_ACCEPT;
}
void enc_g1_uncompressed(uint8_t* b, const g1a* p) { fp_to_be(b, &p->x); fp_to_be(b + 32, &p->y); }
void enc_g2_uncompressed(uint8_t* b, const g2a* p) { fp_to_be(b, &p->x.c1); fp_to_be(b + 32, &p->x.c0); fp_to_be(b + 64, &p->y.c1); fp_to_be(b + 96, &p->y.c0); }
void enc_g1_compressed(uint8_t* b, const g1a* p) {
  if (p->inf) { memset(b, 0, 32); b[0] = 0x40; return; }
  fp_to_be(b, &p->x);
  b[0] |= fp_is_large(&p->y) ? 0xc0 : 0x80;
}
void enc_g2_compressed(uint8_t* b, const g2a* p) {
  if (p->inf) { memset(b, 0, 64); b[0] = 0x40; return; }
  fp_to_be(b, &p->x.c1); fp_to_be(b + 32, &p->x.c0);
  int large = u256_is_zero(&p->y.c1) ? fp_is_large(&p->y.c0) : fp_is_large(&p->y.c1);
  b[0] |= large ? 0xc0 : 0x80;
}

/* ======================= Groth16 ======================= */
typedef struct {
  g1a alpha, beta_neg, delta; g1a* k; uint32_t nk;
  g2a beta_neg2, gamma, delta2;
} g16_vk;

static uint32_t be32(const uint8_t* b) { return (uint32_t)b[0] << 24 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 8 | b[3]; }
static uint64_t be64(const uint8_t* b) { return (uint64_t)be32(b) << 32 | be32(b + 4); }

/* groth16/converter.rs:28-89.  Short buffers are slice-index panics in the reference -> ORC_ERR_MALFORMED. */
static int load_g16_vk(g16_vk* vk, const uint8_t* b, size_t n, int mode) {
  int st;
  vk->k = NULL; vk->nk = 0;
  if (n < 292) return ORC_ERR_MALFORMED;
  g1a beta1; g2a beta2;
  if ((st = dec_g1_compressed_unchecked(&vk->alpha, b)) != ORC_ACCEPT) return st;
  if ((st = dec_g1_compressed_unchecked(&beta1, b + 32)) != ORC_ACCEPT) return st;
  if ((st = dec_g2_compressed_unchecked(&beta2, b + 64, mode)) != ORC_ACCEPT) return st;
  if ((st = dec_g2_compressed_unchecked(&vk->gamma, b + 128, mode)) != ORC_ACCEPT) return st;
  if ((st = dec_g1_compressed_unchecked(&vk->delta, b + 192)) != ORC_ACCEPT) return st;
  if ((st = dec_g2_compressed_unchecked(&vk->delta2, b + 224, mode)) != ORC_ACCEPT) return st;
  uint32_t nk = be32(b + 288);
  size_t off = 292;
  if ((n - off) / 32 < nk) return ORC_ERR_MALFORMED;
  vk->k = (g1a*)malloc(sizeof(g1a) * (nk ? nk : 1)); vk->nk = nk;
  for (uint32_t i = 0; i < nk; i++, off += 32)
    if ((st = dec_g1_compressed_unchecked(&vk->k[i], b + off)) != ORC_ACCEPT) return st;
  if (n < off + 4) return ORC_ERR_MALFORMED;
  uint32_t outer = be32(b + off); off += 4;
  for (uint32_t i = 0; i < outer; i++) {
    if (n < off + 4) return ORC_ERR_MALFORMED;
    uint32_t cnt = be32(b + off); off += 4;
    if ((n - off) / 4 < cnt) return ORC_ERR_MALFORMED;
    off += 4 * (size_t)cnt;
  }
  if (n < off + 128) return ORC_ERR_MALFORMED;
  g2a ck;
  if ((st = dec_g2_compressed_unchecked(&ck, b + off, mode)) != ORC_ACCEPT) return st;      /* parsed, never used */
  if ((st = dec_g2_compressed_unchecked(&ck, b + off + 64, mode)) != ORC_ACCEPT) return st;
  g1_neg_affine(&vk->beta_neg, &beta1);   /* groth16/converter.rs:74 */
  g2_neg_affine(&vk->beta_neg2, &beta2);  /* groth16/converter.rs:79 */
  return ORC_ACCEPT;
}

/* groth16/verify.rs:65-78 given the loaded key and (batch mode only) a precomputed right-hand side */
static void g16_rhs(fp12* rhs, const g16_vk* vk, int mode) {
  if (mode == ORC_MODE_REFERENCE) {
    pairing_batch(rhs, &vk->alpha, &vk->beta_neg2, 1);          /* verify.rs:70: pairing(alpha, vk.g2.beta) with g2.beta = -beta */
  } else {
    g2a beta; g2_neg_affine(&beta, &vk->beta_neg2);
    pairing_batch(rhs, &vk->alpha, &beta, 1);
  }
}
static int g16_core(const g1a* A, const g2a* B, const g1a* C, const g16_vk* vk, const fp12* rhs_pre,
                    const uint8_t* inputs, size_t n_inputs, int mode) {
  fp12 lhs, rhs;
  g1a ps[3]; g2a qs[3];
  if (rhs_pre) rhs = *rhs_pre; else g16_rhs(&rhs, vk, mode);
  /* verify.rs:53-63 prepare_inputs */
  if (n_inputs + 1 != vk->nk) return ORC_ERR_INPUT_LEN;
  g1j acc; g1_from_affine(&acc, &vk->k[0]);
  for (size_t i = 0; i < n_inputs; i++) {
    u256 s; u256_from_be(&s, inputs + 32 * i); /* Fr::from_slice: stored as-is, no range check */
    g1j b, t; g1_from_affine(&b, &vk->k[i + 1]);
    g1_mul(&t, &b, &s);
    g1a ta; g1_to_affine(&ta, &t);             /* AffineG1 * Fr yields an affine point, then affine + */
    g1j tj; g1_from_affine(&tj, &ta);
    g1_add(&acc, &acc, &tj);
  }
  g1a L; g1_to_affine(&L, &acc);
  ps[0] = *A; qs[0] = *B;
  ps[1] = L;
  ps[2] = *C;
  if (mode == ORC_MODE_REFERENCE) {
    qs[1] = vk->gamma;                           /* verify.rs:75 */
    g2_neg_affine(&qs[2], &vk->delta2);          /* verify.rs:76 */
    pairing_batch(&lhs, ps, qs, 3);
  } else {
    /* gnark: e(A,B) e(L,-gamma) e(C,-delta) == e(alpha,beta) */
    g2_neg_affine(&qs[1], &vk->gamma);
    g2_neg_affine(&qs[2], &vk->delta2);
    pairing_batch(&lhs, ps, qs, 3);
  }
  return fp12_eq(&lhs, &rhs) ? ORC_ACCEPT : ORC_REJECT;
}
/* groth16/converter.rs:14-26 (lib.rs:45: unwrap) */
static int g16_load_proof(g1a* A, g2a* B, g1a* C, const uint8_t* proof, size_t proof_len) {
  int st;
  if (proof_len < 256) return ORC_ERR_MALFORMED;
  if ((st = dec_g1_uncompressed(A, proof)) != ORC_ACCEPT) return st;
  if ((st = dec_g2_uncompressed(B, proof + 64)) != ORC_ACCEPT) return st;
  if ((st = dec_g1_uncompressed(C, proof + 192)) != ORC_ACCEPT) return st;
  return ORC_ACCEPT;
}

int orc_groth16_verify(const uint8_t* proof, size_t proof_len, const uint8_t* vkb, size_t vk_len,
                       const uint8_t* inputs, size_t n_inputs, int mode) {
  orc_init();
  int st;
  g1a A, C; g2a B;
  if ((st = g16_load_proof(&A, &B, &C, proof, proof_len)) != ORC_ACCEPT) return st;
  /* lib.rs:46 */
  g16_vk vk;
  if ((st = load_g16_vk(&vk, vkb, vk_len, mode)) != ORC_ACCEPT) { free(vk.k); return st; }
  st = g16_core(&A, &B, &C, &vk, NULL, inputs, n_inputs, mode);
  free(vk.k);
  return st;
}

int orc_groth16_verify_many(const uint8_t* proofs, size_t proof_stride, const uint8_t* vk, size_t vk_len,
                            const uint8_t* inputs, size_t n_inputs, size_t n, int mode, uint8_t* status) {
  orc_init();
  { g1a g; g1_generator(&g); } /* force curve init before threads */
#pragma omp parallel for schedule(dynamic, 1)
  for (long i = 0; i < (long)n; i++)
    status[i] = (uint8_t)orc_groth16_verify(proofs + (size_t)i * proof_stride, proof_stride, vk, vk_len,
                                            inputs + (size_t)i * n_inputs * 32, n_inputs, mode);
  return 0;
}

/* Batch mode of the same algorithm (second CPU row): the key is loaded once (lib.rs:46 hoisted) and
 * pairing(alpha, beta) computed once (groth16/verify.rs:70 hoisted); everything per proof is unchanged (naive subgroup check,
 * bit-by-bit prepare_inputs, 3-pair pairing_batch with its final exponentiation).  Same status bytes as orc_groth16_verify_many. */
int orc_groth16_verify_many_prepared(const uint8_t* proofs, size_t proof_stride, const uint8_t* vkb, size_t vk_len,
                                     const uint8_t* inputs, size_t n_inputs, size_t n, int mode, uint8_t* status) {
  orc_init();
  { g1a g; g1_generator(&g); }
  g16_vk vk;
  int vst = load_g16_vk(&vk, vkb, vk_len, mode);
  fp12 rhs;
  if (vst == ORC_ACCEPT) g16_rhs(&rhs, &vk, mode);
#pragma omp parallel for schedule(dynamic, 1)
  for (long i = 0; i < (long)n; i++) {
    g1a A, C; g2a B;
    int st = g16_load_proof(&A, &B, &C, proofs + (size_t)i * proof_stride, proof_stride);   /* proof errors come first (lib.rs:45) */
    if (st == ORC_ACCEPT) st = vst;
    if (st == ORC_ACCEPT) st = g16_core(&A, &B, &C, &vk, &rhs, inputs + (size_t)i * n_inputs * 32, n_inputs, mode);
    status[i] = (uint8_t)st;
  }
  free(vk.k);
  return 0;
}

/* ======================= PlonK ======================= */
#define MAX_QCP 8
#define MAX_CLAIMED 16
typedef struct {
  uint64_t size; fr size_inv, generator, coset_shift; uint64_t nb_public;
  g1a s[3], ql, qr, qm, qo, qk, qcp[MAX_QCP]; uint32_t n_qcp;
  g1a kzg_g1; g2a kzg_g2[2];
  uint64_t cci[MAX_QCP]; uint64_t n_cci;
} plonk_vk;
typedef struct {
  g1a lro[3], z, h[3], batch_h; fr claimed[MAX_CLAIMED]; uint32_t n_claimed;
  g1a zs_h; fr zs_value; g1a bsb[MAX_QCP]; uint32_t n_bsb;
  uint8_t claimed_raw[MAX_CLAIMED][32]; uint8_t zs_raw[32];
} plonk_proof;

/* Fr::from_slice: no range check, value stored as-is; arithmetic on it is mod r, so reduce here */
static void fr_from_slice(fr* o, const uint8_t* b32) { f_reduce_be(&FR, o, b32, 32); }
static void fr_to_be(uint8_t* b, const fr* a) { u256 t; f_from_mont(&FR, &t, a); u256_to_be(b, &t); }
static void fr_set_u64(fr* o, uint64_t v) { u256 t = {{v, 0, 0, 0}}; f_to_mont(&FR, o, &t); }

/* plonk/converter.rs:18-119 */
static int load_plonk_vk(plonk_vk* vk, const uint8_t* b, size_t n) {
  int st;
  if (n < 372) return ORC_ERR_MALFORMED;
  vk->size = be64(b);
  fr_from_slice(&vk->size_inv, b + 8);
  fr_from_slice(&vk->generator, b + 40);
  vk->nb_public = be64(b + 72);
  fr_from_slice(&vk->coset_shift, b + 80);
  g1a* pts[8] = {&vk->s[0], &vk->s[1], &vk->s[2], &vk->ql, &vk->qr, &vk->qm, &vk->qo, &vk->qk};
  for (int i = 0; i < 8; i++) if ((st = dec_g1_compressed_unchecked(pts[i], b + 112 + 32 * i)) != ORC_ACCEPT) return st;
  vk->n_qcp = be32(b + 368);
  if (vk->n_qcp > MAX_QCP) return ORC_ERR_MALFORMED;
  size_t off = 372;
  if (n < off + 32 * (size_t)vk->n_qcp + 160 + 33788 + 8) return ORC_ERR_MALFORMED;
  for (uint32_t i = 0; i < vk->n_qcp; i++, off += 32) if ((st = dec_g1_compressed_unchecked(&vk->qcp[i], b + off)) != ORC_ACCEPT) return st;
  if ((st = dec_g1_compressed_unchecked(&vk->kzg_g1, b + off)) != ORC_ACCEPT) return st;
  if ((st = dec_g2_compressed_unchecked(&vk->kzg_g2[0], b + off + 32, ORC_MODE_REFERENCE)) != ORC_ACCEPT) return st;
  if ((st = dec_g2_compressed_unchecked(&vk->kzg_g2[1], b + off + 96, ORC_MODE_REFERENCE)) != ORC_ACCEPT) return st;
  off += 160 + 33788; /* plonk/converter.rs:58: precomputed lines skipped */
  vk->n_cci = be64(b + off); off += 8;
  if (vk->n_cci > MAX_QCP || n < off + 8 * vk->n_cci) return ORC_ERR_MALFORMED;
  for (uint64_t i = 0; i < vk->n_cci; i++, off += 8) vk->cci[i] = be64(b + off);
  return ORC_ACCEPT;
}
/* plonk/converter.rs:121-178 */
static int load_plonk_proof(plonk_proof* p, const uint8_t* b, size_t n) {
  int st;
  if (n < 516) return ORC_ERR_MALFORMED;
  g1a* pts[8] = {&p->lro[0], &p->lro[1], &p->lro[2], &p->z, &p->h[0], &p->h[1], &p->h[2], &p->batch_h};
  for (int i = 0; i < 8; i++) if ((st = dec_g1_uncompressed(pts[i], b + 64 * i)) != ORC_ACCEPT) return st;
  p->n_claimed = be32(b + 512);
  if (p->n_claimed > MAX_CLAIMED) return ORC_ERR_MALFORMED;
  size_t off = 516;
  if (n < off + 32 * (size_t)p->n_claimed + 100) return ORC_ERR_MALFORMED;
  for (uint32_t i = 0; i < p->n_claimed; i++, off += 32) { fr_from_slice(&p->claimed[i], b + off); memcpy(p->claimed_raw[i], b + off, 32); }
  if ((st = dec_g1_uncompressed(&p->zs_h, b + off)) != ORC_ACCEPT) return st;
  fr_from_slice(&p->zs_value, b + off + 64); memcpy(p->zs_raw, b + off + 64, 32);
  p->n_bsb = be32(b + off + 96);
  if (p->n_bsb > MAX_QCP) return ORC_ERR_MALFORMED;
  off += 100;
  if (n < off + 64 * (size_t)p->n_bsb) return ORC_ERR_MALFORMED;
  for (uint32_t i = 0; i < p->n_bsb; i++, off += 64) if ((st = dec_g1_uncompressed(&p->bsb[i], b + off)) != ORC_ACCEPT) return st;
  return ORC_ACCEPT;
}

/* transcript.rs: challenge = SHA256(name | previous challenge digest (if position > 0) | bindings in order) */
typedef struct { sha256_ctx h; } tr_chal;
static void tr_begin(tr_chal* t, const char* name, const uint8_t* prev /* NULL for position 0 */) {
  sha256_init(&t->h);
  sha256_update(&t->h, (const uint8_t*)name, strlen(name));
  if (prev) sha256_update(&t->h, prev, 32);
}
static void tr_bind(tr_chal* t, const uint8_t* d, size_t n) { sha256_update(&t->h, d, n); }
static void tr_bind_g1(tr_chal* t, const g1a* p) { uint8_t b[64]; enc_g1_uncompressed(b, p); tr_bind(t, b, 64); } /* plonk/converter.rs:180-185 */
static void tr_finish(tr_chal* t, uint8_t digest[32], fr* x) { sha256_final(&t->h, digest); if (x) f_reduce_be(&FR, x, digest, 32); }

/* hash_to_field.rs:45-97 (RFC 9380 expand_message_xmd with SHA-256) */
void orc_expand_msg_xmd(uint8_t* out, size_t len, const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len) {
  size_t ell = (len + 31) / 32;
  uint8_t b0[32], b1[32], z[64] = {0}, sx[32];
  sha256_ctx h;
  uint8_t lib[3] = {(uint8_t)(len >> 8), (uint8_t)len, 0}, dl = (uint8_t)dst_len, idx;
  sha256_init(&h); sha256_update(&h, z, 64); sha256_update(&h, msg, msg_len); sha256_update(&h, lib, 3);
  sha256_update(&h, dst, dst_len); sha256_update(&h, &dl, 1); sha256_final(&h, b0);
  idx = 1;
  sha256_init(&h); sha256_update(&h, b0, 32); sha256_update(&h, &idx, 1); sha256_update(&h, dst, dst_len); sha256_update(&h, &dl, 1); sha256_final(&h, b1);
  memcpy(out, b1, len < 32 ? len : 32);
  for (size_t i = 2; i <= ell; i++) {
    for (int j = 0; j < 32; j++) sx[j] = b0[j] ^ b1[j];
    idx = (uint8_t)i;
    sha256_init(&h); sha256_update(&h, sx, 32); sha256_update(&h, &idx, 1); sha256_update(&h, dst, dst_len); sha256_update(&h, &dl, 1); sha256_final(&h, b1);
    size_t start = 32 * (i - 1), end = start + 32 < len ? start + 32 : len;
    memcpy(out + start, b1, end - start);
  }
}

/* AffineG1::msm: naive sum of independent scalar multiplications */
static void g1_msm(g1a* o, const g1a* pts, const fr* sc, int n) {
  g1j acc; memset(&acc, 0, sizeof acc); acc.y = FP.r1;
  for (int i = 0; i < n; i++) {
    u256 k; f_from_mont(&FR, &k, &sc[i]);
    g1j b, t; g1_from_affine(&b, &pts[i]); g1_mul(&t, &b, &k); g1_add(&acc, &acc, &t);
  }
  g1_to_affine(o, &acc);
}

static __thread uint8_t* plonk_pair_out = NULL;  /* when set: the operands of the final pairing check (2 x 64 B G1, 2 x 128 B G2) */
static int plonk_core(const uint8_t* proof_b, size_t proof_len, const uint8_t* vk_b, size_t vk_len, const uint8_t* inputs,
                      size_t n_inputs, const uint8_t* lambda32, uint8_t* stage_out) {
  orc_init();
  { g1a g; g1_generator(&g); }
  int st;
  plonk_proof pr; plonk_vk vk;
  if ((st = load_plonk_proof(&pr, proof_b, proof_len)) != ORC_ACCEPT) return st;   /* lib.rs:70 */
  if ((st = load_plonk_vk(&vk, vk_b, vk_len)) != ORC_ACCEPT) return st;           /* lib.rs:71 */
  /* plonk/verify.rs:52-59 */
  if (pr.n_bsb != vk.n_qcp) return ORC_ERR_BSB22_MISMATCH;
  if (n_inputs != vk.nb_public) return ORC_ERR_INPUT_LEN;
  if (pr.n_claimed != 6 + vk.n_qcp || vk.n_cci != vk.n_qcp) return ORC_ERR_MALFORMED; /* index panics in the reference */
  fr one = FR.r1;
  /* Fiat-Shamir (verify.rs:62-95, 319-362) */
  uint8_t dg[32], db[32], da[32], dz[32];
  fr gamma, beta, alpha, zeta;
  tr_chal t;
  tr_begin(&t, "gamma", NULL);
  tr_bind_g1(&t, &vk.s[0]); tr_bind_g1(&t, &vk.s[1]); tr_bind_g1(&t, &vk.s[2]);
  tr_bind_g1(&t, &vk.ql); tr_bind_g1(&t, &vk.qr); tr_bind_g1(&t, &vk.qm); tr_bind_g1(&t, &vk.qo); tr_bind_g1(&t, &vk.qk);
  for (uint32_t i = 0; i < vk.n_qcp; i++) tr_bind_g1(&t, &vk.qcp[i]);
  for (size_t i = 0; i < n_inputs; i++) tr_bind(&t, inputs + 32 * i, 32); /* into_u256().to_bytes_be(): the stored value */
  tr_bind_g1(&t, &pr.lro[0]); tr_bind_g1(&t, &pr.lro[1]); tr_bind_g1(&t, &pr.lro[2]);
  tr_finish(&t, dg, &gamma);
  tr_begin(&t, "beta", dg); tr_finish(&t, db, &beta);
  tr_begin(&t, "alpha", db);
  for (uint32_t i = 0; i < pr.n_bsb; i++) tr_bind_g1(&t, &pr.bsb[i]);
  tr_bind_g1(&t, &pr.z);
  tr_finish(&t, da, &alpha);
  tr_begin(&t, "zeta", da);
  tr_bind_g1(&t, &pr.h[0]); tr_bind_g1(&t, &pr.h[1]); tr_bind_g1(&t, &pr.h[2]);
  tr_finish(&t, dz, &zeta);
  if (stage_out) { memcpy(stage_out, dg, 32); memcpy(stage_out + 32, db, 32); memcpy(stage_out + 64, da, 32); memcpy(stage_out + 96, dz, 32); }

  /* verify.rs:97-107 */
  u256 nint = {{vk.size, 0, 0, 0}};
  fr zeta_n, zh_zeta, lagrange_one, tmp;
  f_pow(&FR, &zeta_n, &zeta, &nint);
  f_sub(&FR, &zh_zeta, &zeta_n, &one);
  f_sub(&FR, &tmp, &zeta, &one);
  if (!f_inv(&FR, &lagrange_one, &tmp)) return ORC_ERR_INVERSE;
  f_mul(&FR, &lagrange_one, &lagrange_one, &zh_zeta);
  f_mul(&FR, &lagrange_one, &lagrange_one, &vk.size_inv);
  /* verify.rs:109-137: PI = sum L_i(zeta) w_i.  batch_invert skips zeros (verify.rs:377,389) */
  fr pi; memset(&pi, 0, sizeof pi);
  fr accw = one;
  for (size_t i = 0; i < n_inputs; i++) {
    fr den, inv, x, w;
    f_sub(&FR, &den, &zeta, &accw);
    if (!f_inv(&FR, &inv, &den)) inv = den; /* zero stays zero */
    f_mul(&FR, &x, &zh_zeta, &inv);
    f_mul(&FR, &x, &x, &vk.size_inv);
    f_mul(&FR, &x, &x, &accw);
    fr_from_slice(&w, inputs + 32 * i);
    f_mul(&FR, &x, &x, &w);
    f_mul(&FR, &accw, &accw, &vk.generator);
    f_add(&FR, &pi, &pi, &x);
  }
  /* verify.rs:139-163 BSB22 */
  for (uint64_t i = 0; i < vk.n_cci; i++) {
    uint8_t cb[64], hb[48];
    enc_g1_uncompressed(cb, &pr.bsb[i]);
    orc_expand_msg_xmd(hb, 48, cb, 64, (const uint8_t*)"BSB22-Plonk", 11);
    if (stage_out && i == 0) memcpy(stage_out + 128, hb, 48);
    fr hashed, wpow, den, lag, di;
    f_reduce_be(&FR, &hashed, hb, 48);
    u256 e = {{vk.nb_public + vk.cci[i], 0, 0, 0}};
    f_pow(&FR, &wpow, &vk.generator, &e);
    f_sub(&FR, &den, &zeta, &wpow);
    f_mul(&FR, &lag, &zh_zeta, &wpow);
    f_inv(&FR, &di, &den); /* `/=`: no zero check in the reference (verify.rs:157) */
    f_mul(&FR, &lag, &lag, &di);
    f_mul(&FR, &lag, &lag, &vk.size_inv);
    f_mul(&FR, &lag, &lag, &hashed);
    f_add(&FR, &pi, &pi, &lag);
  }
  /* verify.rs:165-214 */
  const fr *l = &pr.claimed[1], *r = &pr.claimed[2], *o = &pr.claimed[3], *s1 = &pr.claimed[4], *s2 = &pr.claimed[5], *zu = &pr.zs_value;
  fr a2l1, const_lin, t1;
  f_mul(&FR, &a2l1, &lagrange_one, &alpha); f_mul(&FR, &a2l1, &a2l1, &alpha);
  f_mul(&FR, &t1, &beta, s1); f_add(&FR, &t1, &t1, &gamma); f_add(&FR, &t1, &t1, l); const_lin = t1;
  f_mul(&FR, &t1, &beta, s2); f_add(&FR, &t1, &t1, &gamma); f_add(&FR, &t1, &t1, r); f_mul(&FR, &const_lin, &const_lin, &t1);
  f_add(&FR, &t1, o, &gamma); f_mul(&FR, &const_lin, &const_lin, &t1);
  f_mul(&FR, &const_lin, &const_lin, &alpha); f_mul(&FR, &const_lin, &const_lin, zu);
  f_sub(&FR, &const_lin, &const_lin, &a2l1); f_add(&FR, &const_lin, &const_lin, &pi);
  f_neg(&FR, &const_lin, &const_lin);
  { /* Fr == compares the stored words: an unreduced claimed value (>= r) can never equal the reduced const_lin */
    u256 raw0; u256_from_be(&raw0, pr.claimed_raw[0]);
    if (u256_cmp(&raw0, &FR.m) >= 0 || u256_cmp(&const_lin, &pr.claimed[0]) != 0) return ORC_ERR_OPENING_MISMATCH;
  }
  /* verify.rs:216-250 */
  fr _s1, _s2, coeff_z, rl, u, t2;
  f_mul(&FR, &_s1, &beta, s1); f_add(&FR, &_s1, &_s1, l); f_add(&FR, &_s1, &_s1, &gamma);
  f_mul(&FR, &t1, &beta, s2); f_add(&FR, &t1, &t1, r); f_add(&FR, &t1, &t1, &gamma);
  f_mul(&FR, &_s1, &_s1, &t1); f_mul(&FR, &_s1, &_s1, &beta); f_mul(&FR, &_s1, &_s1, &alpha); f_mul(&FR, &_s1, &_s1, zu);
  f_mul(&FR, &_s2, &beta, &zeta); f_add(&FR, &_s2, &_s2, &gamma); f_add(&FR, &_s2, &_s2, l);
  f_mul(&FR, &u, &beta, &vk.coset_shift); f_mul(&FR, &t1, &u, &zeta); f_add(&FR, &t1, &t1, &gamma); f_add(&FR, &t1, &t1, r);
  f_mul(&FR, &_s2, &_s2, &t1);
  f_mul(&FR, &t2, &u, &vk.coset_shift); f_mul(&FR, &t1, &t2, &zeta); f_add(&FR, &t1, &t1, &gamma); f_add(&FR, &t1, &t1, o);
  f_mul(&FR, &_s2, &_s2, &t1); f_mul(&FR, &_s2, &_s2, &alpha); f_neg(&FR, &_s2, &_s2);
  f_add(&FR, &coeff_z, &a2l1, &_s2);
  f_mul(&FR, &rl, l, r);
  u256 n2 = {{vk.size + 2, 0, 0, 0}};
  fr zn2, zn2sq, zh;
  f_pow(&FR, &zn2, &zeta, &n2);
  f_mul(&FR, &zn2sq, &zn2, &zn2);
  f_mul(&FR, &zn2, &zn2, &zh_zeta); f_neg(&FR, &zn2, &zn2);
  f_mul(&FR, &zn2sq, &zn2sq, &zh_zeta); f_neg(&FR, &zn2sq, &zn2sq);
  f_neg(&FR, &zh, &zh_zeta);
  /* verify.rs:252-284 */
  g1a pts[MAX_QCP + 10]; fr sc[MAX_QCP + 10]; int np = 0;
  for (uint32_t i = 0; i < pr.n_bsb; i++) { pts[np] = pr.bsb[i]; sc[np++] = pr.claimed[6 + i]; }
  pts[np] = vk.ql; sc[np++] = *l;   pts[np] = vk.qr; sc[np++] = *r;   pts[np] = vk.qm; sc[np++] = rl;
  pts[np] = vk.qo; sc[np++] = *o;   pts[np] = vk.qk; sc[np++] = one;  pts[np] = vk.s[2]; sc[np++] = _s1;
  pts[np] = pr.z; sc[np++] = coeff_z; pts[np] = pr.h[0]; sc[np++] = zh; pts[np] = pr.h[1]; sc[np++] = zn2; pts[np] = pr.h[2]; sc[np++] = zn2sq;
  g1a lin_digest; g1_msm(&lin_digest, pts, sc, np);
  /* verify.rs:286-303 + kzg.rs:87-126 fold_proof */
  int nd = 6 + (int)vk.n_qcp;
  g1a dig[MAX_QCP + 6];
  dig[0] = lin_digest; dig[1] = pr.lro[0]; dig[2] = pr.lro[1]; dig[3] = pr.lro[2]; dig[4] = vk.s[0]; dig[5] = vk.s[1];
  for (uint32_t i = 0; i < vk.n_qcp; i++) dig[6 + i] = vk.qcp[i];
  uint8_t b32[32], dgam[32];
  fr kgamma;
  tr_begin(&t, "gamma", NULL);                              /* kzg.rs:46-72: a fresh transcript */
  fr_to_be(b32, &zeta); tr_bind(&t, b32, 32);
  for (int i = 0; i < nd; i++) tr_bind_g1(&t, &dig[i]);
  for (int i = 0; i < nd; i++) tr_bind(&t, pr.claimed_raw[i], 32); /* into_u256() of the value as stored by Fr::from_slice: the raw bytes */
  tr_bind(&t, pr.zs_raw, 32);
  tr_finish(&t, dgam, &kgamma);
  fr gi[MAX_QCP + 6]; gi[0] = one; if (nd > 1) gi[1] = kgamma;
  for (int i = 2; i < nd; i++) f_mul(&FR, &gi[i], &gi[i - 1], &kgamma);
  fr folded_eval; memset(&folded_eval, 0, sizeof folded_eval);
  for (int i = 0; i < nd; i++) { f_mul(&FR, &t1, &pr.claimed[i], &gi[i]); f_add(&FR, &folded_eval, &folded_eval, &t1); }
  g1a folded_digest; g1_msm(&folded_digest, dig, gi, nd);
  /* kzg.rs:128-190 batch_verify_multi_points with digests [folded_digest, z], proofs [(batch_h, folded_eval), (zs_h, zu)], points [zeta, zeta*g] */
  fr lam;
  if (lambda32) fr_from_slice(&lam, lambda32); else fr_set_u64(&lam, 0x9e3779b97f4a7c15ull);
  fr rn[2] = {one, lam};
  g1a quot[2] = {pr.batch_h, pr.zs_h};
  g1a folded_quot; g1_msm(&folded_quot, quot, rn, 2);
  g1a d2[2] = {folded_digest, pr.z};
  fr ev[2] = {folded_eval, *zu};
  fr fe; memset(&fe, 0, sizeof fe);
  for (int i = 0; i < 2; i++) { f_mul(&FR, &t1, &ev[i], &rn[i]); f_add(&FR, &fe, &fe, &t1); }
  g1a fd; g1_msm(&fd, d2, rn, 2);
  g1a fec; g1_msm(&fec, &vk.kzg_g1, &fe, 1);
  g1a nfec; g1_neg_affine(&nfec, &fec);
  g1j j1, j2, j3; g1_from_affine(&j1, &fd); g1_from_affine(&j2, &nfec); g1_add(&j3, &j1, &j2);
  fr shifted; f_mul(&FR, &shifted, &zeta, &vk.generator);
  fr rp[2]; f_mul(&FR, &rp[0], &rn[0], &zeta); f_mul(&FR, &rp[1], &rn[1], &shifted);
  g1a fpq; g1_msm(&fpq, quot, rp, 2);
  g1_from_affine(&j1, &fpq); g1_add(&j3, &j3, &j1);
  g1a ps[2]; g2a qs[2];
  g1_to_affine(&ps[0], &j3);
  g1_neg_affine(&ps[1], &folded_quot); if (folded_quot.inf) ps[1] = folded_quot;
  qs[0] = vk.kzg_g2[0]; qs[1] = vk.kzg_g2[1];
  if (plonk_pair_out && !ps[0].inf && !ps[1].inf) {
    enc_g1_uncompressed(plonk_pair_out, &ps[0]); enc_g1_uncompressed(plonk_pair_out + 64, &ps[1]);
    enc_g2_uncompressed(plonk_pair_out + 128, &qs[0]); enc_g2_uncompressed(plonk_pair_out + 256, &qs[1]);
  }
  fp12 e; pairing_batch(&e, ps, qs, 2);
  if (!fp12_is_one(&e)) return ORC_ERR_PAIRING_FAILED;
  return ORC_ACCEPT;
}
int orc_plonk_verify(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len, const uint8_t* inputs, size_t n_inputs, const uint8_t* lambda32) {
  return plonk_core(proof, proof_len, vk, vk_len, inputs, n_inputs, lambda32, NULL);
}
/* the (G1, G2) operands of the KZG pairing check (plonk/kzg.rs:175-187) for a proof: tests push them through the device pairing */
int orc_plonk_pairing_inputs(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len, const uint8_t* inputs, size_t n_inputs, uint8_t* out384) {
  memset(out384, 0, 384);
  plonk_pair_out = out384;
  int st = plonk_core(proof, proof_len, vk, vk_len, inputs, n_inputs, NULL, NULL);
  plonk_pair_out = NULL;
  return st;
}
/* same with an explicit KZG batching scalar (tests/test_oracle_golden.py::test_kzg_batching_scalar_must_be_unpredictable) */
int orc_plonk_pairing_inputs_lam(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len, const uint8_t* inputs, size_t n_inputs,
                                 const uint8_t* lambda32, uint8_t* out384) {
  memset(out384, 0, 384);
  plonk_pair_out = out384;
  int st = plonk_core(proof, proof_len, vk, vk_len, inputs, n_inputs, lambda32, NULL);
  plonk_pair_out = NULL;
  return st;
}
int orc_plonk_stage_digests(const uint8_t* proof, size_t proof_len, const uint8_t* vk, size_t vk_len, const uint8_t* inputs, size_t n_inputs, uint8_t* out176) {
  memset(out176, 0, 176);
  return plonk_core(proof, proof_len, vk, vk_len, inputs, n_inputs, NULL, out176);
}

/* ======================= byte-level probes ======================= */
static void fp2_from_be(fp2* o, const uint8_t* b) { u256 t; u256_from_be(&t, b); f_to_mont(&FP, &o->c0, &t); u256_from_be(&t, b + 32); f_to_mont(&FP, &o->c1, &t); }
static void fp2_to_be(uint8_t* b, const fp2* a) { fp_to_be(b, &a->c0); fp_to_be(b + 32, &a->c1); }
static void fp12_from_be(fp12* o, const uint8_t* b) {
  fp2* c[6] = {&o->c0.c0, &o->c0.c1, &o->c0.c2, &o->c1.c0, &o->c1.c1, &o->c1.c2};
  for (int i = 0; i < 6; i++) fp2_from_be(c[i], b + 64 * i);
}
static void fp12_to_be(uint8_t* b, const fp12* a) {
  const fp2* c[6] = {&a->c0.c0, &a->c0.c1, &a->c0.c2, &a->c1.c0, &a->c1.c1, &a->c1.c2};
  for (int i = 0; i < 6; i++) fp2_to_be(b + 64 * i, c[i]);
}
void orc_fp_op(int op, uint8_t* o32, const uint8_t* a32, const uint8_t* b32, int field) {
  orc_init();
  const fctx* F = field ? &FR : &FP;
  u256 a, b, r; memset(&r, 0, sizeof r); memset(&b, 0, sizeof b);
  f_reduce_be(F, &a, a32, 32); if (b32) f_reduce_be(F, &b, b32, 32);
  switch (op) {
    case 0: f_add(F, &r, &a, &b); break;
    case 1: f_sub(F, &r, &a, &b); break;
    case 2: f_mul(F, &r, &a, &b); break;
    case 3: f_inv(F, &r, &a); break;
    case 4: if (!fp_sqrt(&r, &a)) memset(&r, 0, sizeof r); break;
    case 5: f_neg(F, &r, &a); break;
  }
  u256 c; f_from_mont(F, &c, &r); u256_to_be(o32, &c);
}
void orc_fp2_op(int op, uint8_t* o64, const uint8_t* a64, const uint8_t* b64) {
  orc_init();
  fp2 a, b, r; memset(&r, 0, sizeof r); memset(&b, 0, sizeof b);
  fp2_from_be(&a, a64); if (b64) fp2_from_be(&b, b64);
  switch (op) {
    case 0: fp2_add(&r, &a, &b); break;
    case 1: fp2_sub(&r, &a, &b); break;
    case 2: fp2_mul(&r, &a, &b); break;
    case 3: fp2_inv(&r, &a); break;
    case 4: if (!fp2_sqrt(&r, &a)) memset(&r, 0, sizeof r); break;
    case 5: fp2_sqr(&r, &a); break;
  }
  fp2_to_be(o64, &r);
}
void orc_fp12_op(int op, uint8_t* o384, const uint8_t* a384, const uint8_t* b384) {
  orc_init();
  { g1a g; g1_generator(&g); }
  fp12 a, b, r; memset(&r, 0, sizeof r); memset(&b, 0, sizeof b);
  fp12_from_be(&a, a384); if (b384) fp12_from_be(&b, b384);
  switch (op) {
    case 0: fp12_mul(&r, &a, &b); break;
    case 1: fp12_sqr(&r, &a); break;
    case 2: fp12_inv(&r, &a); break;
    case 3: fp12_frob(&r, &a, 1); break;
    case 4: fp12_frob(&r, &a, 2); break;
    case 5: fp12_frob(&r, &a, 3); break;
    case 6: fp12_cyclo_sqr(&r, &a); break;
    case 7: fp12_conj(&r, &a); break;
  }
  fp12_to_be(o384, &r);
}
static void g1_from_bytes_raw(g1a* o, const uint8_t* b) { u256 t; u256_from_be(&t, b); f_to_mont(&FP, &o->x, &t); u256_from_be(&t, b + 32); f_to_mont(&FP, &o->y, &t); o->inf = 0;
  /* all-zero encodes infinity in the probes */ int z = 1; for (int i = 0; i < 64; i++) if (b[i]) z = 0; o->inf = z; }
static void g2_from_bytes_raw(g2a* o, const uint8_t* b) {
  u256 t;
  u256_from_be(&t, b); f_to_mont(&FP, &o->x.c1, &t); u256_from_be(&t, b + 32); f_to_mont(&FP, &o->x.c0, &t);
  u256_from_be(&t, b + 64); f_to_mont(&FP, &o->y.c1, &t); u256_from_be(&t, b + 96); f_to_mont(&FP, &o->y.c0, &t);
  int z = 1; for (int i = 0; i < 128; i++) if (b[i]) z = 0; o->inf = z;
}
int orc_g1_scalar_mul(uint8_t* o64, const uint8_t* p64, const uint8_t* k32) {
  orc_init(); { g1a g; g1_generator(&g); }
  g1a p, r; g1j j, t; u256 k;
  g1_from_bytes_raw(&p, p64); u256_from_be(&k, k32);
  g1_from_affine(&j, &p); g1_mul(&t, &j, &k); g1_to_affine(&r, &t);
  if (r.inf) { memset(o64, 0, 64); return 0; }
  enc_g1_uncompressed(o64, &r); return 1;
}
int orc_g1_add_bytes(uint8_t* o64, const uint8_t* p64, const uint8_t* q64) {
  orc_init(); { g1a g; g1_generator(&g); }
  g1a p, q, r; g1j a, b, t;
  g1_from_bytes_raw(&p, p64); g1_from_bytes_raw(&q, q64);
  g1_from_affine(&a, &p); g1_from_affine(&b, &q); g1_add(&t, &a, &b); g1_to_affine(&r, &t);
  if (r.inf) { memset(o64, 0, 64); return 0; }
  enc_g1_uncompressed(o64, &r); return 1;
}
int orc_g2_scalar_mul(uint8_t* o128, const uint8_t* p128, const uint8_t* k32) {
  orc_init(); { g1a g; g1_generator(&g); }
  g2a p, r; g2j j, t; u256 k;
  g2_from_bytes_raw(&p, p128); u256_from_be(&k, k32);
  g2_from_affine(&j, &p); g2_mul(&t, &j, &k); g2_to_affine(&r, &t);
  if (r.inf) { memset(o128, 0, 128); return 0; }
  enc_g2_uncompressed(o128, &r); return 1;
}
int orc_g2_add_bytes(uint8_t* o128, const uint8_t* p128, const uint8_t* q128) {
  orc_init(); { g1a g; g1_generator(&g); }
  g2a p, q, r; g2j a, b, t;
  g2_from_bytes_raw(&p, p128); g2_from_bytes_raw(&q, q128);
  g2_from_affine(&a, &p); g2_from_affine(&b, &q); g2_add(&t, &a, &b); g2_to_affine(&r, &t);
  if (r.inf) { memset(o128, 0, 128); return 0; }
  enc_g2_uncompressed(o128, &r); return 1;
}
void orc_g1_gen(uint8_t* o64) { g1a g; g1_generator(&g); enc_g1_uncompressed(o64, &g); }
void orc_g2_gen(uint8_t* o128) { g2a g; g2_generator(&g); enc_g2_uncompressed(o128, &g); }
int orc_g2_subgroup_check(const uint8_t* p128) {
  orc_init(); { g1a g; g1_generator(&g); }
  g2a p; g2_from_bytes_raw(&p, p128);
  if (!g2_on_curve(&p.x, &p.y)) return -1;
  return g2_in_subgroup_naive(&p);
}
void orc_miller_loop(uint8_t* o384, const uint8_t* g1s, const uint8_t* g2s, int n) {
  orc_init(); { g1a g; g1_generator(&g); }
  g1a* ps = (g1a*)malloc(sizeof(g1a) * (n ? n : 1)); g2a* qs = (g2a*)malloc(sizeof(g2a) * (n ? n : 1));
  for (int i = 0; i < n; i++) { g1_from_bytes_raw(&ps[i], g1s + 64 * i); g2_from_bytes_raw(&qs[i], g2s + 128 * i); }
  fp12 f; miller_loop_batch(&f, ps, qs, n); fp12_to_be(o384, &f);
  free(ps); free(qs);
}
void orc_final_exp(uint8_t* o384, const uint8_t* f384, int plain) {
  orc_init(); { g1a g; g1_generator(&g); }
  fp12 f, r; fp12_from_be(&f, f384);
  if (plain) final_exponentiation_plain(&r, &f); else final_exponentiation(&r, &f);
  fp12_to_be(o384, &r);
}
void orc_pairing_bytes(uint8_t* o384, const uint8_t* g1s, const uint8_t* g2s, int n) {
  uint8_t m[384]; orc_miller_loop(m, g1s, g2s, n); orc_final_exp(o384, m, 0);
}
int orc_decompress_g1(uint8_t* o64, const uint8_t* b32) { g1a p; int st = dec_g1_compressed_unchecked(&p, b32); if (st == ORC_ACCEPT) enc_g1_uncompressed(o64, &p); return st; }
int orc_decompress_g2(uint8_t* o128, const uint8_t* b64, int mode) { g2a p; int st = dec_g2_compressed_unchecked(&p, b64, mode); if (st == ORC_ACCEPT) enc_g2_uncompressed(o128, &p); return st; }
void orc_compress_g1(uint8_t* o32, const uint8_t* b64) { orc_init(); g1a p; g1_from_bytes_raw(&p, b64); enc_g1_compressed(o32, &p); }
void orc_compress_g2(uint8_t* o64, const uint8_t* b128) { orc_init(); g2a p; g2_from_bytes_raw(&p, b128); enc_g2_compressed(o64, &p); }
void orc_sha256(uint8_t* o32, const uint8_t* d, size_t n) { sha256_ctx c; sha256_init(&c); sha256_update(&c, d, n); sha256_final(&c, o32); }
uint64_t orc_get_fp_mul_count(void) { return orc_fp_mul_count; }
void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
void orc_reset_fp_mul_count(void) { orc_fp_mul_count = 0; }
#ifndef ORC_CFLAGS
#define ORC_CFLAGS "unknown flags"
#endif
const char* orc_build_flags(void) { return "gcc " __VERSION__ " " ORC_CFLAGS; }
