// bn254_host.hpp -- host-side (once-per-key) work of the library: gnark verifying-key parsing and decompression,
// e(alpha, beta), Miller-loop line tables and fixed-base window tables, plus the synthetic workload generator.
// Uses the SAME arithmetic headers as the kernels, compiled for the host; nothing here is on the per-proof path
// and nothing comes from oracle/.
//
// Restates (paths relative to /root/reference/verifier/src): converter.rs:23-43 (flags), :62-76 (compressed G1),
// :113-133 (compressed G2), groth16/converter.rs:28-89 (vk layout), groth16/verify.rs:70 (e(alpha,beta)).
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "bn254_vm.h"
#include "bn254_kernels.h"

namespace bn254host {
using namespace bn254;

// ---------------------------------------------------------------- small helpers
inline void fp_to_limbs(int32_t* out, const Fp& a) {
  Fp c = fp_reduce(fp_norm(a));
  for (int i = 0; i < BN_NL; i++) out[i] = c.v[i];
}
inline Fp fp_from_be(const uint8_t* be32) { uint32_t w[8]; words_from_be(w, be32); return fp_from_words(w); }
inline void fp_to_be(uint8_t* be32, const Fp& a) { uint32_t w[8]; fp_to_words(w, a); words_to_be(be32, w); }
inline bool be_lt_p(const uint8_t* be32) { uint32_t w[8]; words_from_be(w, be32); return !words_ge(w, BN_P_WORDS); }
inline bool fp_is_large(const Fp& a) {  // canonical value > (p-1)/2
  uint32_t w[8]; fp_to_words(w, a);
  bool ge = words_ge(w, BN_P_HALF_WORDS);
  bool eq = true; for (int i = 0; i < 8; i++) eq &= (w[i] == BN_P_HALF_WORDS[i]);
  return ge && !eq;
}
inline int fp_cmp_canon(const Fp& a, const Fp& b) {
  uint32_t x[8], y[8]; fp_to_words(x, a); fp_to_words(y, b);
  for (int i = 7; i >= 0; i--) { if (x[i] < y[i]) return -1; if (x[i] > y[i]) return 1; }
  return 0;
}
inline bool fp_sqrt(Fp& out, const Fp& a) {  // p = 3 mod 4
  Fp r = fp_pow_bits(fp_reduce(fp_norm(a)), BN_EXP_SQRT_BITS, BN_EXP_SQRT_NBITS);
  if (!fp_eq(fp_sqr(r), a)) return false;
  out = r;
  return true;
}
inline Fp2 fp2_pow_bits(const Fp2& a, const uint8_t* bits, int nbits) {
  Fp2 acc = a;
  for (int i = 1; i < nbits; i++) { acc = fp2_sqr(acc); if (bits[i]) acc = fp2_mul(acc, a); }
  return acc;
}
// square root in Fp2 = Fp[i]/(i^2+1), p = 3 mod 4 (complex method); which root comes back is unspecified
inline bool fp2_sqrt(Fp2& out, const Fp2& a) {
  if (fp2_is_zero(a)) { out = fp2_zero(); return true; }
  Fp2 a1 = fp2_pow_bits(a, BN_EXP_PM3O4_BITS, BN_EXP_PM3O4_NBITS);
  Fp2 alpha = fp2_mul(fp2_sqr(a1), a);
  Fp2 a0 = fp2_mul(fp2_conj(alpha), alpha);
  Fp2 minus_one; minus_one.c0 = fp_neg(fp_one()); minus_one.c1 = fp_zero();
  if (fp2_eq(a0, minus_one)) return false;
  Fp2 x0 = fp2_mul(a1, a), r;
  if (fp2_eq(alpha, minus_one)) { r.c0 = fp_neg(x0.c1); r.c1 = x0.c0; }
  else { Fp2 b = fp2_pow_bits(fp2_add(alpha, fp2_one()), BN_EXP_PM1O2_BITS, BN_EXP_PM1O2_NBITS); r = fp2_mul(b, x0); }
  if (!fp2_eq(fp2_sqr(r), a)) return false;
  out = r;
  return true;
}
inline bool fp2_lex_large(const Fp2& y) { return fp_is_zero(y.c1) ? fp_is_large(y.c0) : fp_is_large(y.c1); }  // gnark's LexicographicallyLargest

// ---------------------------------------------------------------- gnark codecs
enum { DEC_OK = 0, DEC_MALFORMED = 1 };
// converter.rs:23-43: flag in the top two bits, x silently reduced mod p, infinity must be all zero, flag 0b00 panics
inline int deserialize_with_flags(Fp& x, int& flag, const uint8_t* b32) {
  int m = b32[0] >> 6;
  if (m == 0) return DEC_MALFORMED;
  if (m == 1) {
    if (b32[0] & 0x3f) return DEC_MALFORMED;
    for (int i = 1; i < 32; i++) if (b32[i]) return DEC_MALFORMED;
    x = fp_zero(); flag = 1; return DEC_OK;
  }
  uint8_t t[32]; memcpy(t, b32, 32); t[0] &= 0x3f;
  x = fp_from_be(t);  // fp_from_words reduces any 256-bit value mod p
  flag = m;
  return DEC_OK;
}
// converter.rs:62-76 (unchecked): y = sqrt(x^3+3), flag 10 -> smaller root, 11 -> larger; the infinity flag falls through
// with x = 0, and 3 is a non-residue mod p, so it ends in InvalidPoint (a panic in the reference)
inline int dec_g1_compressed(G1Aff& o, const uint8_t* b32) {
  Fp x; int flag;
  if (deserialize_with_flags(x, flag, b32) != DEC_OK) return DEC_MALFORMED;
  Fp y;
  if (!fp_sqrt(y, fp_add(fp_mul(fp_sqr(x), x), fp_from_limbs(BN_THREE)))) return DEC_MALFORMED;
  Fp ny = fp_neg(y);
  if (fp_cmp_canon(y, ny) > 0) { Fp s = y; y = ny; ny = s; }
  o.x = x; o.y = (flag == 3) ? ny : y;
  return DEC_OK;
}
// converter.rs:113-133 (unchecked).  mode 0 (reference): the two roots are ordered by the real part c0 alone, as the pinned
// `bn` does, flag 10 -> first; mode 1 (gnark): flag 10 -> lexicographically smallest.  The infinity flag
// yields the G2 GENERATOR (AffineG2::one(), converter.rs:122-124).
inline int dec_g2_compressed(G2Aff& o, const uint8_t* b64, int mode) {
  Fp x1; int flag;
  if (deserialize_with_flags(x1, flag, b64) != DEC_OK) return DEC_MALFORMED;
  Fp x0 = fp_from_be(b64 + 32);
  if (flag == 1) {
    o.x.c0 = fp_from_limbs(BN_G2_GEN[0]); o.x.c1 = fp_from_limbs(BN_G2_GEN[1]);
    o.y.c0 = fp_from_limbs(BN_G2_GEN[2]); o.y.c1 = fp_from_limbs(BN_G2_GEN[3]);
    return DEC_OK;
  }
  Fp2 x; x.c0 = x0; x.c1 = x1;
  Fp2 y;
  if (!fp2_sqrt(y, fp2_add(fp2_mul(fp2_sqr(x), x), g2_twist_b()))) return DEC_MALFORMED;
  Fp2 ny = fp2_neg(y);
  bool y_first = (mode == 0) ? (fp_cmp_canon(y.c0, ny.c0) < 0) : !fp2_lex_large(y);
  o.x = x;
  o.y = (flag == 2) ? (y_first ? y : ny) : (y_first ? ny : y);
  return DEC_OK;
}
inline void enc_g1_uncompressed(uint8_t* b64, const G1Aff& p) { fp_to_be(b64, p.x); fp_to_be(b64 + 32, p.y); }
inline void enc_g2_uncompressed(uint8_t* b, const G2Aff& p) { fp_to_be(b, p.x.c1); fp_to_be(b + 32, p.x.c0); fp_to_be(b + 64, p.y.c1); fp_to_be(b + 96, p.y.c0); }
inline void enc_g1_compressed(uint8_t* b32, const G1Aff& p) { fp_to_be(b32, p.x); b32[0] |= fp_is_large(p.y) ? 0xc0 : 0x80; }
inline void enc_g2_compressed(uint8_t* b64, const G2Aff& p) { fp_to_be(b64, p.x.c1); fp_to_be(b64 + 32, p.x.c0); b64[0] |= fp2_lex_large(p.y) ? 0xc0 : 0x80; }

// ---------------------------------------------------------------- parsed verifying key
struct G16Key {
  G1Aff alpha;
  G2Aff beta, gamma, delta;  // as decoded (before the reference's negations)
  std::vector<G1Aff> k;
};
inline uint32_t be32(const uint8_t* b) { return (uint32_t)b[0] << 24 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 8 | b[3]; }
// groth16/converter.rs:28-89; every slice-index panic of the reference becomes DEC_MALFORMED
inline int parse_g16_vk(G16Key& vk, const uint8_t* b, size_t n, int mode) {
  if (n < 292) return DEC_MALFORMED;
  G1Aff beta1, delta1;
  if (dec_g1_compressed(vk.alpha, b) || dec_g1_compressed(beta1, b + 32) || dec_g2_compressed(vk.beta, b + 64, mode) ||
      dec_g2_compressed(vk.gamma, b + 128, mode) || dec_g1_compressed(delta1, b + 192) || dec_g2_compressed(vk.delta, b + 224, mode))
    return DEC_MALFORMED;
  uint32_t nk = be32(b + 288);
  size_t off = 292;
  if ((n - off) / 32 < nk) return DEC_MALFORMED;
  vk.k.resize(nk);
  for (uint32_t i = 0; i < nk; i++, off += 32) if (dec_g1_compressed(vk.k[i], b + off)) return DEC_MALFORMED;
  if (n < off + 4) return DEC_MALFORMED;
  uint32_t outer = be32(b + off); off += 4;
  for (uint32_t i = 0; i < outer; i++) {
    if (n < off + 4) return DEC_MALFORMED;
    uint32_t cnt = be32(b + off); off += 4;
    if ((n - off) / 4 < cnt) return DEC_MALFORMED;
    off += 4 * (size_t)cnt;
  }
  if (n < off + 128) return DEC_MALFORMED;
  G2Aff ck;  // Pedersen commitment key: parsed (so that its errors surface like in the reference), never used
  if (dec_g2_compressed(ck, b + off, mode) || dec_g2_compressed(ck, b + off + 64, mode)) return DEC_MALFORMED;
  return DEC_OK;
}

// ---------------------------------------------------------------- prepared key (host image of what the kernels read)
struct G16Prepared {
  size_t n_k = 0;                        // len(vk.K); 0 is a key the reference loads and then answers PrepareInputsFailed for every input count (groth16/verify.rs:54-56)
  size_t key_inputs() const { return n_k ? n_k - 1 : 0; }                       // public inputs the tables are built for
  bool inputs_match(size_t n_public) const { return n_public + 1 == n_k; }      // groth16/verify.rs:54 (never true for n_k = 0)
  std::vector<int32_t> k0;               // 18
  std::vector<int32_t> gtab, dtab;       // BN_ATE_STEPS * FIXED_LINE_DWORDS
  std::vector<int32_t> target;           // 108
  std::vector<int32_t> msm;              // (n_k - 1) * 32 * 255 * MSM_ENTRY_DWORDS; keys with many inputs (msm_comb): (n_k - 1) * (1 << G16_COMB_TEETH) = 8192 entries of MSM_ENTRY_DWORDS each
  bool msm_comb = false;                 // the table is in comb form (build_comb_table): read by k_g16_msm_partial_comb only
  std::vector<int32_t> kpts;             // tables built on the device (bn254_k_comb.hip; the default): K[1 ..] as affine digits, 18 dwords each -- `msm` then stays empty on the host
  G1Aff alpha, k0_pt; G2Aff b_arg;       // kept for the RLC tables (prepare_g16_rlc): alpha, K[0] and the G2 argument of the target pairing
};
inline void put_fp2(int32_t* o, const Fp2& a) { fp_to_limbs(o, a.c0); fp_to_limbs(o + BN_NL, a.c1); }
inline void put_fp12(int32_t* o, const Fp12& a) {  // w-power (k) order, as the workspace stores Fp12 values (bn254_vm.h)
  put_fp2(o, K0(a)); put_fp2(o + 2 * BN_NL, K1(a)); put_fp2(o + 4 * BN_NL, K2(a));
  put_fp2(o + 6 * BN_NL, K3(a)); put_fp2(o + 8 * BN_NL, K4(a)); put_fp2(o + 10 * BN_NL, K5(a));
}
// the fixed-base tables are built on the host (and uploaded) instead of on the device: the construction of rounds 1-4, kept for comparison
inline bool bn254_tables_on_host() {
  static const bool v = [] { const char* a = getenv("BN254_TABLES_HOST"); const char* b = getenv("BN254_COMB_HOST"); return (a && atoi(a) != 0) || (b && atoi(b) != 0); }();
  return v;
}
// batch conversion of projective points to affine with one inversion (Montgomery's trick); none may be the identity
inline void g1_batch_to_affine(G1Aff* out, const G1Proj* in, size_t n) {
  std::vector<Fp> pre(n);
  Fp acc = fp_one();
  for (size_t i = 0; i < n; i++) { pre[i] = acc; acc = fp_mul(acc, fp_reduce(fp_norm(in[i].z))); }
  Fp inv = fp_inv(acc);
  for (size_t i = n; i-- > 0;) {
    Fp zi = fp_mul(inv, pre[i]);
    inv = fp_mul(inv, fp_reduce(fp_norm(in[i].z)));
    out[i].x = fp_mul(in[i].x, zi);
    out[i].y = fp_mul(in[i].y, zi);
  }
}
// window table of one base: entry [w][d-1] = d * 2^(8w) * base, d = 1..255, w = 0..31
inline void build_window_table(int32_t* out /* 32*255*MSM_ENTRY_DWORDS */, const G1Aff& base) {
  std::vector<G1Proj> pts(32 * 255);
  G1Proj bw = g1_from_affine(base);
  for (int w = 0; w < 32; w++) {
    G1Aff bwa = g1_to_affine(bw);
    G1Proj acc = bw;
    pts[(size_t)w * 255] = acc;
    for (int d = 2; d <= 255; d++) { acc = g1_add_mixed(acc, bwa); pts[(size_t)w * 255 + d - 1] = acc; }
    bw = g1_add_mixed(acc, bwa);  // 256 * 2^(8w) * base
  }
  std::vector<G1Aff> aff(pts.size());
  g1_batch_to_affine(aff.data(), pts.data(), pts.size());
  for (size_t e = 0; e < aff.size(); e++) {
    int32_t* o = out + e * MSM_ENTRY_DWORDS;
    fp_to_limbs(o, aff[e].x); fp_to_limbs(o + BN_NL, aff[e].y); o[18] = 0; o[19] = 0;
  }
}
// comb table of one base (keys with many public inputs): G16_COMB_TEETH = 13 teeth G16_COMB_COLS = 20 bit positions apart (13 x 20 = 260 >= 256
// bits); entry [idx] = sum over the set bits i of idx of 2^(20 i) * base, idx = 1..8191 (entry 0 unused).  A 256-bit scalar x is then
//   x * base = sum_{c = 0..19} 2^c * entry[ bits c, c + 20, ..., c + 240 of x ]
// i.e. 20 additions per input and 20 doublings that ALL inputs of a lane share, against 32 additions with the byte windows -- from a table
// of the same size (8192 entries per input against 8160).  No entry is the identity: none of the 8191 sums of powers 2^(20 i), i < 13, is a
// multiple of r (they are below 2^241 < r, and tests/test_capi_cpu.py::test_comb_table_constants_never_vanish checks it from the header's parameters).
inline void build_comb_table(int32_t* out /* (1 << G16_COMB_TEETH) * MSM_ENTRY_DWORDS */, const G1Aff& base) {
  G1Aff tooth[G16_COMB_TEETH];
  {
    G1Proj t = g1_from_affine(base);
    G1Proj tp[G16_COMB_TEETH];
    for (int i = 0; i < G16_COMB_TEETH; i++) { tp[i] = t; for (int d = 0; d < G16_COMB_COLS; d++) t = g1_dbl(t); }
    g1_batch_to_affine(tooth, tp, G16_COMB_TEETH);
  }
  const size_t n = (size_t)1 << G16_COMB_TEETH;
  std::vector<G1Proj> pts(n);
  pts[0] = g1_from_affine(base);   // placeholder (entry 0 is never read); keeps the batch conversion free of the identity
  for (size_t idx = 1; idx < n; idx++) {
    const size_t lb = idx & (~idx + 1);
    int i = 0; while (((size_t)1 << i) != lb) i++;
    pts[idx] = idx == lb ? g1_from_affine(tooth[i]) : g1_add_mixed(pts[idx ^ lb], tooth[i]);
  }
  std::vector<G1Aff> aff(n);
  g1_batch_to_affine(aff.data(), pts.data(), n);
  for (size_t e = 0; e < n; e++) {
    int32_t* o = out + e * MSM_ENTRY_DWORDS;
    fp_to_limbs(o, aff[e].x); fp_to_limbs(o + BN_NL, aff[e].y); o[18] = 0; o[19] = 0;
  }
}
// mode 0: reference-literal equation  e(A,B) e(L, gamma') e(C, -delta') == e(alpha, -beta')   (groth16/verify.rs:70-77, converter.rs:79)
// mode 1: gnark                       e(A,B) e(L, -gamma) e(C, -delta)  == e(alpha, beta)
// Returns false only if a line table cannot be built.  That does not happen for a key that parsed: its G2 elements are ON THE TWIST by construction (y is computed
// from x, converter.rs:113-133), and for no point of E'(Fp2) other than the identity does a step of the optimal-ate walk meet T = +-S or T = O -- the order of such a point
// would have to divide one of the walk's partial multipliers (all below 2^66; none is a multiple of a prime factor of #E'(Fp2) = r * 10069 * 5864401 * 1875725156269 * c177),
// or the Frobenius eigenvalue on one of its prime components would have to equal +-(6u+2) (tests/test_capi_cpu.py::test_line_tables_exist_for_every_twist_point
// enumerates both).  So "unchecked" key points (off the r-torsion) are computed on exactly as the reference does, by the same group law.
inline bool prepare_g16(G16Prepared& out, const G16Key& vk, int mode) {
  out.n_k = vk.k.size();
  out.k0.resize(2 * BN_NL);
  // a key without K points never reaches prepare_inputs' sum: the generator stands in for K[0] so that the loader checks of the proofs (which come first, lib.rs:45)
  // still run through the same kernels; every proof that passes them is answered BN254_ERR_INPUT_LEN (inputs_match is never true)
  G1Aff k0; if (out.n_k) k0 = vk.k[0]; else { k0.x = fp_one(); k0.y = fp_add(fp_one(), fp_one()); }
  fp_to_limbs(out.k0.data(), k0.x); fp_to_limbs(out.k0.data() + BN_NL, k0.y);
  G2Aff g = (mode == 0) ? vk.gamma : g2_neg(vk.gamma);
  G2Aff d = g2_neg(vk.delta);
  G2Aff b = (mode == 0) ? g2_neg(vk.beta) : vk.beta;
  std::vector<FixedLine> tg(BN_ATE_STEPS), td(BN_ATE_STEPS);
  if (!fixed_line_table(tg.data(), g) || !fixed_line_table(td.data(), d)) return false;
  out.gtab.resize((size_t)BN_ATE_STEPS * FIXED_LINE_DWORDS); out.dtab.resize((size_t)BN_ATE_STEPS * FIXED_LINE_DWORDS);
  for (int s = 0; s < BN_ATE_STEPS; s++) {
    int32_t* g_ = out.gtab.data() + (size_t)s * FIXED_LINE_DWORDS;
    int32_t* d_ = out.dtab.data() + (size_t)s * FIXED_LINE_DWORDS;
    put_fp2(g_, tg[s].m); put_fp2(g_ + 2 * BN_NL, tg[s].c); put_fp2(g_ + 4 * BN_NL, tg[s].xc);
    put_fp2(d_, td[s].m); put_fp2(d_ + 2 * BN_NL, td[s].c); put_fp2(d_ + 4 * BN_NL, td[s].xc);
  }
  out.alpha = vk.alpha; out.k0_pt = k0; out.b_arg = b;
  Fp12 t = final_exponentiation(miller_loop<0>(vk.alpha, b, nullptr, nullptr));
  out.target.resize(12 * BN_NL);
  put_fp12(out.target.data(), t);
  size_t nb = out.key_inputs();
  // keys with more than G16_WIDE_MSM_MIN_INPUTS inputs: comb tables (BN254_WIDE_COMB=0 keeps the byte-window form for comparison)
  const char* ce = getenv("BN254_WIDE_COMB");
  out.msm_comb = nb > (size_t)G16_WIDE_MSM_MIN_INPUTS && !(ce && atoi(ce) == 0);
  const size_t per_base = (out.msm_comb ? ((size_t)1 << G16_COMB_TEETH) : (size_t)32 * 255) * MSM_ENTRY_DWORDS;
  // the tables are built on the device that uses them (bn254_k_comb.hip: milliseconds instead of 2.2 s of host threads and a 671 MB upload for 1024 inputs, 0.18 s for 16);
  // BN254_TABLES_HOST=1 (or its round-5 name BN254_COMB_HOST=1) keeps the host construction
  if (nb > 0 && (!bn254_tables_on_host() || nb <= (size_t)G16_WIDE_MSM_MIN_INPUTS)) {      // keys with up to 16 inputs: 13-bit windows (bn254_fw.h), a device construction only
    out.kpts.resize(nb * 2 * BN_NL);
    for (size_t i = 0; i < nb; i++) { fp_to_limbs(out.kpts.data() + i * 2 * BN_NL, vk.k[i + 1].x); fp_to_limbs(out.kpts.data() + i * 2 * BN_NL + BN_NL, vk.k[i + 1].y); }
    return true;
  }
  out.msm.assign(nb * per_base, 0);
  unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 1; if (hw > 16) hw = 16;
  if (nb < hw) hw = (unsigned)(nb ? nb : 1);
  std::vector<std::thread> th;
  for (unsigned t_ = 0; t_ < hw; t_++)
    th.emplace_back([&, t_]() {
      for (size_t i = t_; i < nb; i += hw) {
        if (out.msm_comb) build_comb_table(out.msm.data() + i * per_base, vk.k[i + 1]);
        else build_window_table(out.msm.data() + i * per_base, vk.k[i + 1]);
      }
    });
  for (auto& x : th) x.join();
  return true;
}

// tables of the RLC batch mode (bn254_rlc.h): lines of the G2 argument paired with alpha, window tables of -alpha and K[0], 1 in GT
struct G16PreparedRlc {
  bool ready = false;
  std::vector<int32_t> btab, tab, one;
  std::vector<int32_t> pts;              // -alpha and K[0] as affine digits: their window tables are built on the device (`tab` stays empty)
};
inline bool prepare_g16_rlc(G16PreparedRlc& out, const G16Prepared& base) {
  std::vector<FixedLine> tb(BN_ATE_STEPS);
  if (!fixed_line_table(tb.data(), base.b_arg)) return false;
  out.btab.resize((size_t)BN_ATE_STEPS * FIXED_LINE_DWORDS);
  for (int s = 0; s < BN_ATE_STEPS; s++) {
    int32_t* b_ = out.btab.data() + (size_t)s * FIXED_LINE_DWORDS;
    put_fp2(b_, tb[s].m); put_fp2(b_ + 2 * BN_NL, tb[s].c); put_fp2(b_ + 4 * BN_NL, tb[s].xc);
  }
  {
    // the two window tables (-alpha, K[0]; 13-bit windows, bn254_fw.h) are built on the device like the key's own (bn254_k_comb.hip)
    const G1Aff na = g1_neg(base.alpha);
    out.pts.resize(2 * 2 * BN_NL);
    fp_to_limbs(out.pts.data(), na.x); fp_to_limbs(out.pts.data() + BN_NL, na.y);
    fp_to_limbs(out.pts.data() + 2 * BN_NL, base.k0_pt.x); fp_to_limbs(out.pts.data() + 3 * BN_NL, base.k0_pt.y);
  }
  out.one.resize(12 * BN_NL);
  put_fp12(out.one.data(), fp12_one());
  out.ready = true;
  return true;
}

// ---------------------------------------------------------------- scalar field (mod r) for the generator: 256-bit, 4 x u64
struct U256 { uint64_t l[4]; };
inline U256 u256_r() { U256 r; for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)BN_R_WORDS[2 * i] | ((uint64_t)BN_R_WORDS[2 * i + 1] << 32); return r; }
inline int u256_cmp(const U256& a, const U256& b) { for (int i = 3; i >= 0; i--) { if (a.l[i] < b.l[i]) return -1; if (a.l[i] > b.l[i]) return 1; } return 0; }
inline uint64_t u256_add(U256& o, const U256& a, const U256& b) { unsigned __int128 c = 0; for (int i = 0; i < 4; i++) { c += (unsigned __int128)a.l[i] + b.l[i]; o.l[i] = (uint64_t)c; c >>= 64; } return (uint64_t)c; }
inline uint64_t u256_sub(U256& o, const U256& a, const U256& b) { uint64_t br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - br; o.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } return br; }
inline U256 fr_add(const U256& a, const U256& b) { U256 t, s, m = u256_r(); uint64_t c = u256_add(t, a, b); uint64_t br = u256_sub(s, t, m); return (c || !br) ? s : t; }
inline U256 fr_sub(const U256& a, const U256& b) { U256 t, m = u256_r(); if (u256_sub(t, a, b)) u256_add(t, t, m); return t; }
inline U256 fr_mul(const U256& a, const U256& b) {  // double-and-add; a, b < r.  Slow and simple: a handful per proof
  U256 acc = {{0, 0, 0, 0}};
  for (int i = 255; i >= 0; i--) { acc = fr_add(acc, acc); if ((b.l[i >> 6] >> (i & 63)) & 1) acc = fr_add(acc, a); }
  return acc;
}
inline U256 fr_inv(const U256& a) {  // a^(r-2)
  U256 e = u256_r(), two = {{2, 0, 0, 0}}; u256_sub(e, e, two);
  U256 acc = {{1, 0, 0, 0}};
  for (int i = 255; i >= 0; i--) { acc = fr_mul(acc, acc); if ((e.l[i >> 6] >> (i & 63)) & 1) acc = fr_mul(acc, a); }
  return acc;
}
inline void u256_to_be(uint8_t* b, const U256& a) { for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) b[(3 - i) * 8 + j] = (uint8_t)(a.l[i] >> (56 - 8 * j)); }

struct SplitMix64 { uint64_t s; uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); } };
inline U256 fr_random(SplitMix64& g, bool nonzero) {
  for (;;) {
    U256 v; for (int i = 0; i < 4; i++) v.l[i] = g.next();
    v.l[3] &= 0x3fffffffffffffffull;  // < 2^254, then rejection below r
    if (u256_cmp(v, u256_r()) >= 0) continue;
    if (nonzero && !(v.l[0] | v.l[1] | v.l[2] | v.l[3])) continue;
    return v;
  }
}

// fixed-base tables of the generators for the synthetic prover
struct GenTables {
  std::vector<G1Aff> g1;  // [32][255]
  std::vector<G2Aff> g2;  // [32][255]
};
inline G2Proj g2_dbl_nolines(G2Proj t) { (void)g2_double_step(t); return t; }
inline G2Aff g2_to_affine(const G2Proj& p) { Fp2 zi = fp2_inv(p.z); G2Aff r; r.x = fp2_mul(p.x, zi); r.y = fp2_mul(p.y, zi); return r; }
inline void build_gen_tables(GenTables& t) {
  t.g1.resize(32 * 255); t.g2.resize(32 * 255);
  {
    G1Aff gen; gen.x = fp_one(); gen.y = fp_add(fp_one(), fp_one());
    std::vector<G1Proj> pts(32 * 255);
    G1Proj bw = g1_from_affine(gen);
    for (int w = 0; w < 32; w++) {
      G1Aff bwa = g1_to_affine(bw); G1Proj acc = bw; pts[(size_t)w * 255] = acc;
      for (int d = 2; d <= 255; d++) { acc = g1_add_mixed(acc, bwa); pts[(size_t)w * 255 + d - 1] = acc; }
      bw = g1_add_mixed(acc, bwa);
    }
    g1_batch_to_affine(t.g1.data(), pts.data(), pts.size());
  }
  {
    G2Aff gen; gen.x.c0 = fp_from_limbs(BN_G2_GEN[0]); gen.x.c1 = fp_from_limbs(BN_G2_GEN[1]); gen.y.c0 = fp_from_limbs(BN_G2_GEN[2]); gen.y.c1 = fp_from_limbs(BN_G2_GEN[3]);
    G2Aff bwa = gen;
    for (int w = 0; w < 32; w++) {
      G2Proj acc = g2_from_affine(bwa);
      t.g2[(size_t)w * 255] = bwa;
      for (int d = 2; d <= 255; d++) {
        if (d == 2) acc = g2_dbl_nolines(acc); else (void)g2_add_step(acc, bwa);
        t.g2[(size_t)w * 255 + d - 1] = g2_to_affine(acc);
      }
      (void)g2_add_step(acc, bwa);
      bwa = g2_to_affine(acc);
    }
  }
}
inline G1Proj g1_mul_gen(const GenTables& t, const U256& k) {
  G1Proj acc = g1_identity();
  for (int w = 0; w < 32; w++) { unsigned d = (unsigned)(k.l[w >> 3] >> (8 * (w & 7))) & 0xff; if (d) acc = g1_add_mixed(acc, t.g1[(size_t)w * 255 + d - 1]); }
  return acc;
}
inline G2Aff g2_mul_gen(const GenTables& t, const U256& k) {  // k != 0 mod r; incomplete additions are safe for random k
  G2Proj acc; bool have = false;
  for (int w = 0; w < 32; w++) {
    unsigned d = (unsigned)(k.l[w >> 3] >> (8 * (w & 7))) & 0xff;
    if (!d) continue;
    const G2Aff& e = t.g2[(size_t)w * 255 + d - 1];
    if (!have) { acc = g2_from_affine(e); have = true; } else (void)g2_add_step(acc, e);
  }
  return g2_to_affine(acc);
}

}  // namespace bn254host
