// bn254_rlc.h -- random-linear-combination batch mode of the Groth16 check (BN254_FLAG_RLC).
//
// Exact check per proof i (groth16/verify.rs:73-77, key-side G2 arguments g', d', b' as prepared by bn254_host.hpp):
//     e(A_i, B_i) e(L_i, g') e(C_i, d') == e(alpha, b')
// Batched with independent random weights r_i of 128 bits of entropy (GLV form, below) (the reference batches its KZG openings the same way, plonk/kzg.rs:149-187):
//     prod_i e(r_i A_i, B_i)  *  e(sum_i r_i L_i, g')  *  e(sum_i r_i C_i, d')  *  e((sum_i r_i)(-alpha), b')  ==  1
// and, because L_i = K_0 + sum_j x_ij K_j,   sum_i r_i L_i = (sum_i r_i) K_0 + sum_j (sum_i r_i x_ij) K_j :  the public-input MSM is
// done ONCE PER GROUP with scalars accumulated in Fr.  Per proof that leaves: two 128-bit G1 scalar multiplications (A, C), one
// variable-argument Miller loop (which also yields the r-torsion test of B, bn254_vm.h::vm_g2_ate_check) and n_public + 1 products
// in Fr.  Per group: a fold of the per-proof values (Fp12 products, G1 additions, Fr additions), three table-driven pairs, ONE final
// exponentiation.  If a group's product is not 1 its proofs go through the exact path (bn254_capi.hip), so status bytes are exact
// except for a false ACCEPT with probability ~ 2^-128 per forged proof.
//
// Everything here is templated on the workspace accessor like bn254_vm.h, so tests/hostsim runs the same code on the CPU.
#pragma once
#include "bn254_vm.h"
#include "bn254_rlc_plan.h"
#include "bn254_fw.h"

namespace bn254 {

// ---- workspace elements of the RLC mode (all inside VE_S1 = 34..45, which the per-proof stage does not use otherwise) ---------------------
enum {
  RLC_C = VE_S1,            // r_i C_i, projective (3 Fp)
  RLC_T = VE_S1 + 3,        // t_0 = r_i, t_j = r_i x_ij mod r  (j = 1..n_public): Fr values, 8 x 32-bit words in the low 8 digits of a slot
  RLC_MAX_PUBLIC = 8,       // 3 + 1 + 8 = 12 slots
  RLC_ACC = VE_S2,          // group stage: accumulator of the three table-driven pairs
  RLC_HI = 512              // fold accessor: element ids >= RLC_HI address the partner lane
};

// ---- ChaCha20 block function (RFC 8439): the weights are r_i = first 128 bits of block(key, counter = i, nonce) -------------------------------
BN_HD uint32_t bn_rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define BN_CHACHA_QR(a, b, c, d) do { a += b; d ^= a; d = bn_rotl(d, 16); c += d; b ^= c; b = bn_rotl(b, 12); a += b; d ^= a; d = bn_rotl(d, 8); c += d; b ^= c; b = bn_rotl(b, 7); } while (0)
struct ChaChaKey { uint32_t k[8]; uint32_t nonce[3]; };
BN_HD void chacha20_block4(uint32_t out[4], const ChaChaKey& key, uint32_t counter) {
  uint32_t s0 = 0x61707865u, s1 = 0x3320646eu, s2 = 0x79622d32u, s3 = 0x6b206574u;
  uint32_t x0 = s0, x1 = s1, x2 = s2, x3 = s3, x4 = key.k[0], x5 = key.k[1], x6 = key.k[2], x7 = key.k[3], x8 = key.k[4], x9 = key.k[5],
           x10 = key.k[6], x11 = key.k[7], x12 = counter, x13 = key.nonce[0], x14 = key.nonce[1], x15 = key.nonce[2];
  for (int i = 0; i < 10; i++) {
    BN_CHACHA_QR(x0, x4, x8, x12); BN_CHACHA_QR(x1, x5, x9, x13); BN_CHACHA_QR(x2, x6, x10, x14); BN_CHACHA_QR(x3, x7, x11, x15);
    BN_CHACHA_QR(x0, x5, x10, x15); BN_CHACHA_QR(x1, x6, x11, x12); BN_CHACHA_QR(x2, x7, x8, x13); BN_CHACHA_QR(x3, x4, x9, x14);
  }
  out[0] = x0 + s0; out[1] = x1 + s1; out[2] = x2 + s2; out[3] = x3 + s3;
}

// ---- Fr = Z / r, 8 x 32-bit words, Montgomery products with R = 2^256 (CIOS on v_mad_u64_u32) ---------------------------------------------------
struct Fr8 { uint32_t w[8]; };
BN_HD uint32_t bn_r_word(int i) {
  switch (i) {
    case 0: return 0xf0000001u; case 1: return 0x43e1f593u; case 2: return 0x79b97091u; case 3: return 0x2833e848u;
    case 4: return 0x8181585du; case 5: return 0xb85045b6u; case 6: return 0xe131a029u; default: return 0x30644e72u;
  }
}
#define BN_R_NINV32 0xefffffffu   // -r^-1 mod 2^32
BN_HD uint32_t bn_r2_word(int i) {  // 2^512 mod r
  switch (i) {
    case 0: return 0xae216da7u; case 1: return 0x1bb8e645u; case 2: return 0xe35c59e3u; case 3: return 0x53fe3ab1u;
    case 4: return 0x53bb8085u; case 5: return 0x8c49833du; case 6: return 0x7f4e44a5u; default: return 0x0216d0b1u;
  }
}
BN_HD Fr8 fr8_cond_sub_r(const uint32_t t[9]) {  // t < 2r (9 words) -> t mod r
  uint32_t d[8]; uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { uint64_t x = (uint64_t)t[i] - bn_r_word(i) - br; d[i] = (uint32_t)x; br = (x >> 32) & 1; }
  const bool ge = t[8] != 0 || br == 0;
  Fr8 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = ge ? d[i] : t[i];
  return r;
}
// a * b / 2^256 mod r, for a * b < r * 2^256 (a < 2^256 and b < r, or the reverse)
BN_HD Fr8 fr8_mont_mul(const Fr8& a, const Fr8& b) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { uint64_t s = (uint64_t)a.w[j] * b.w[i] + t[j] + c; t[j] = (uint32_t)s; c = s >> 32; }
    uint64_t s8 = (uint64_t)t[8] + c; t[8] = (uint32_t)s8; t[9] = (uint32_t)(s8 >> 32);
    const uint32_t m = t[0] * BN_R_NINV32;
    c = ((uint64_t)m * bn_r_word(0) + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) { uint64_t s = (uint64_t)m * bn_r_word(j) + t[j] + c; t[j - 1] = (uint32_t)s; c = s >> 32; }
    uint64_t s = (uint64_t)t[8] + c; t[7] = (uint32_t)s; t[8] = t[9] + (uint32_t)(s >> 32);
  }
  return fr8_cond_sub_r(t);
}
BN_HD Fr8 fr8_add(const Fr8& a, const Fr8& b) {  // a, b < r
  uint32_t t[9]; uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)a.w[i] + b.w[i]; t[i] = (uint32_t)c; c >>= 32; }
  t[8] = (uint32_t)c;
  return fr8_cond_sub_r(t);
}
BN_HD Fr8 fr8_zero() { Fr8 r; for (int i = 0; i < 8; i++) r.w[i] = 0; return r; }
// x * k mod r for any x < 2^256 and k < r: mont(mont(x, 2^512), k)
BN_HD Fr8 fr8_mul_plain(const Fr8& x, const Fr8& k) {
  Fr8 r2;
#pragma unroll
  for (int i = 0; i < 8; i++) r2.w[i] = bn_r2_word(i);
  return fr8_mont_mul(fr8_mont_mul(x, r2), k);
}
// an Fr value travels through the workspace in the low 8 digits of an Fp slot (raw words, no field meaning)
BN_HD Fp fr8_to_slot(const Fr8& a) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = (int32_t)a.w[i];
  r.v[8] = 0;
#if BN_TRACKING
  r.vb = 0; r.lb = 0;
#endif
  return r;
}
BN_HD Fr8 fr8_from_slot(const Fp& a) {
  Fr8 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = (uint32_t)a.v[i];
  return r;
}

// ---- the weights and their scalar multiplications: GLV form ------------------------------------------------------------------------------------------------
// BN254's G1 has the endomorphism phi(x, y) = (beta x, y) = [lambda](x, y), beta^3 = 1 in Fp, lambda^2 + lambda + 1 = 0 in Fr.  A weight is
//     r_i = k1 + k2 * lambda  (mod r),   k1, k2 uniform 64-bit values  (the 128 bits of the ChaCha20 block),
// so r_i P = k1 P + k2 phi(P) costs 64 doublings and 64 additions instead of 128 + 64.  The map (k1, k2) -> r_i is injective (a collision would
// be a vector of the GLV lattice shorter than 2^65, and its shortest vectors are ~ 2^127 long), so the weight takes 2^128 distinct values mod r:
// a forged proof passes its group's check with probability <= 2^-128, as with a plain 128-bit weight.
BN_FP_CONST(BN_GLV_BETA, {171145621, -175301341, -123954279, -218476986, -127948067, 24572270, 250450874, -198635200, 789247});   /* beta, Montgomery form */
BN_HD uint32_t bn_glv_lambda_word(int i) {  // lambda = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
  switch (i) {
    case 0: return 0xb99c90ddu; case 1: return 0x8b17ea66u; case 2: return 0x8d8daaa7u; case 3: return 0x5bfc4108u;
    case 4: return 0x41a91758u; case 5: return 0xb3c4d79du; default: return 0u;
  }
}
// the weight as an element of Fr (canonical words): k1 + k2 lambda mod r
BN_HD Fr8 rlc_weight(const uint32_t k[4]) {
  Fr8 k2 = fr8_zero(), lam, k1 = fr8_zero();
  k1.w[0] = k[0]; k1.w[1] = k[1]; k2.w[0] = k[2]; k2.w[1] = k[3];
#pragma unroll
  for (int i = 0; i < 8; i++) lam.w[i] = bn_glv_lambda_word(i);
  return fr8_add(fr8_mul_plain(k2, lam), k1);
}
// (+-k1) P + (+-k2) phi(P) for W-word magnitudes k1, k2: joint double-and-add over the 32 W bit positions with the table {P1, P2, P1 + P2},
// P1 = +-P, P2 = +-phi(P); complete formulas, data-independent control flow.  W = 2: the RLC weights; W = 4: the GLV halves of a full scalar
// (PlonK's MSMs, decomposed on the host by bn254_plonk.hpp::glv_decompose).
template <int W>
BN_HD G1Proj g1_mul_glv_w(const G1Aff& P, const uint32_t* k1, bool neg1, const uint32_t* k2, bool neg2) {
  G1Aff P1 = P, P2;
  P1.y = fp_select(neg1, fp_neg(P.y), P.y);
  P2.x = fp_mul(P.x, fp_from_limbs(BN_GLV_BETA)); P2.y = fp_select(neg2, fp_neg(P.y), P.y);
  const G1Proj T1 = g1_from_affine(P1), T2 = g1_from_affine(P2), T3 = g1_add_mixed(T1, P2);
  G1Proj acc = g1_identity();
  uint32_t a[W], b[W];
#pragma unroll
  for (int i = 0; i < W; i++) { a[i] = k1[i]; b[i] = k2[i]; }
  for (int bit = 0; bit < 32 * W; bit++) {
    const uint32_t d1 = a[W - 1] >> 31, d2 = b[W - 1] >> 31;
#pragma unroll
    for (int i = W - 1; i > 0; i--) { a[i] = (a[i] << 1) | (a[i - 1] >> 31); b[i] = (b[i] << 1) | (b[i - 1] >> 31); }
    a[0] <<= 1; b[0] <<= 1;
    acc = g1_dbl(acc);
    G1Proj q;
    q.x = fp_select(d2 != 0, fp_select(d1 != 0, T3.x, T2.x), T1.x);
    q.y = fp_select(d2 != 0, fp_select(d1 != 0, T3.y, T2.y), T1.y);
    q.z = fp_select(d2 != 0, fp_select(d1 != 0, T3.z, T2.z), T1.z);
    G1Proj c = g1_add(acc, q);
    const bool take = (d1 | d2) != 0;
    acc.x = fp_select(take, c.x, acc.x); acc.y = fp_select(take, c.y, acc.y); acc.z = fp_select(take, c.z, acc.z);
  }
  return acc;
}
BN_HD G1Proj g1_mul_glv(const G1Aff& P, const uint32_t k[4]) { return g1_mul_glv_w<2>(P, k, false, k + 2, false); }
// ---- per proof: A <- r A (affine), C' <- r C (projective), t_0 = r, t_j = r x_j ---------------------------------------------------------------------
// LX(j, out_words[8]): the j-th public input of this proof as little-endian words (raw 256-bit value, used modulo r like bn::Fr)
template <class W, class LX>
BN_HD void vm_rlc_scale(W& w, const uint32_t r[4], int n_public, const LX& load_input) {
  {
    G1Aff A; A.x = w.ld(VE_AX); A.y = w.ld(VE_AY);
    G1Aff Ar = g1_to_affine(g1_mul_glv(A, r));    // A has order r (on the curve, cofactor 1): the identity only for the weight 0 (probability 2^-128)
    w.st(VE_AX, Ar.x); w.st(VE_AY, Ar.y);
  }
  {
    G1Aff Cc; Cc.x = w.ld(VE_CX); Cc.y = w.ld(VE_CY);
    G1Proj Cr = g1_mul_glv(Cc, r);
    w.st(RLC_C, fp_reduce(Cr.x)); w.st(RLC_C + 1, fp_reduce(Cr.y)); w.st(RLC_C + 2, fp_reduce(Cr.z));
  }
  const Fr8 rr = rlc_weight(r);
  w.st(RLC_T, fr8_to_slot(rr));
  for (int j = 0; j < n_public; j++) {
    Fr8 x; load_input(j, x.w);
    w.st(RLC_T + 1 + j, fr8_to_slot(fr8_mul_plain(x, rr)));
  }
}
// ---- one Miller step of G variable pairs SHARING the accumulator f (one lane walks G proofs) -------------------------------------------------
// f <- [f^2] * prod_q line_q(A_q): the squaring is paid once per lane instead of once per proof.  The accessor addresses the current
// proof's column (w.sel(q): T_q, B_q, A_q; proof q of lane j sits m lanes further per q) and the lane's own column for f (w.ld0 / w.st0).
// Between the stages f is in flight exactly as in bn254_vm.h::vm_miller_step (k0..k3 parked, k4, k5 in registers).  deadmask bit q: proof q of
// this lane takes no part (loader error, or it does not exist): its line is replaced by 1 by keeping f, so arbitrary bytes cannot reach f.
// kind: 0 doubling, 1..4 addition of +B, -B, psi(B), -psi^2(B).
template <bool DO_SQR, class W>
BN_HD void vm_miller_var_multi(W& w, int kind, int G, uint32_t deadmask) {
  Fp2 f4, f5;
  {
    Fp2 k0, k1, k2, k3, k4, k5;
    k0.c0 = w.ld0(VE_F); k0.c1 = w.ld0(VE_F + 1); k1.c0 = w.ld0(VE_F + 2); k1.c1 = w.ld0(VE_F + 3); k2.c0 = w.ld0(VE_F + 4); k2.c1 = w.ld0(VE_F + 5);
    k3.c0 = w.ld0(VE_F + 6); k3.c1 = w.ld0(VE_F + 7); k4.c0 = w.ld0(VE_F + 8); k4.c1 = w.ld0(VE_F + 9); k5.c0 = w.ld0(VE_F + 10); k5.c1 = w.ld0(VE_F + 11);
    if constexpr (DO_SQR) {
      Fp2 x3 = fp2_mul_xi(k3), x4 = fp2_mul_xi(k4), x5 = fp2_mul_xi(k5);
      w.park(0, fp2_dotp(pp(k0, k0), pp2(k1, x5), pp2(k2, x4), pp(k3, x3)));
      w.park(1, fp2_dotp(pp2(k0, k1), pp2(k2, x5), pp2(k3, x4)));
      w.park(2, fp2_dotp(pp2(k0, k2), pp(k1, k1), pp2(k3, x5), pp(k4, x4)));
      w.park(3, fp2_dotp(pp2(k0, k3), pp2(k1, k2), pp2(k4, x5)));
      f4 = fp2_dotp(pp2(k0, k4), pp2(k1, k3), pp(k2, k2), pp(k5, x5));
      f5 = fp2_dotp(pp2(k0, k5), pp2(k1, k4), pp2(k2, k3));
    } else {
      w.park(0, k0); w.park(1, k1); w.park(2, k2); w.park(3, k3); f4 = k4; f5 = k5;
    }
  }
  for (int q = 0; q < G; q++) {
    w.sel(q);
    const bool dead = ((deadmask >> q) & 1u) != 0;
    G2Line l;
    {
      G2Proj t; t.x = vld2(w, VE_T); t.y = vld2(w, VE_T + 2); t.z = vld2(w, VE_T + 4);
      if (kind == 0) {
        l = g2_double_step(t);
      } else {
        G2Aff b; b.x = vld2(w, VE_B); b.y = vld2(w, VE_B + 2);
        if (kind == 2) b = g2_neg(b);
        else if (kind == 3) b = g2_psi_affine(b);
        else if (kind == 4) b = g2_neg(g2_psi2_affine(b));
        l = g2_add_step(t, b);
      }
      vst2(w, VE_T, t.x); vst2(w, VE_T + 2, t.y); vst2(w, VE_T + 4, t.z);
    }
    BN_SCHED_FENCE();
    Fp px = w.ld(VE_AX), py = w.ld(VE_AY);
    Fp2 d0 = fp2_mul_fp(l.r0, py), d3 = fp2_mul_fp(l.r1, px), d4 = l.r2;
    Fp2 x3 = fp2_mul_xi(d3), x4 = fp2_mul_xi(d4);
    BN_SCHED_FENCE();
    Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3), k4 = f4, k5 = f5;
    w.park(0, fp2_select(dead, k0, fp2_dotp(pp(d0, k0), pp(x3, k5), pp(x4, k3))));
    w.park(1, fp2_select(dead, k1, fp2_dotp(pp(d0, k1), pp(d3, k0), pp(x4, k4))));
    w.park(2, fp2_select(dead, k2, fp2_dotp(pp(d0, k2), pp(d3, k1), pp(x4, k5))));
    w.park(3, fp2_select(dead, k3, fp2_dotp(pp(d0, k3), pp(d3, k2), pp(d4, k0))));
    Fp2 n4 = fp2_select(dead, k4, fp2_dotp(pp(d0, k4), pp(d3, k3), pp(d4, k1)));
    f5 = fp2_select(dead, k5, fp2_dotp(pp(d0, k5), pp(d3, k4), pp(d4, k2)));
    f4 = n4;
  }
  {
    Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3);
    w.st0(VE_F, k0.c0); w.st0(VE_F + 1, k0.c1); w.st0(VE_F + 2, k1.c0); w.st0(VE_F + 3, k1.c1); w.st0(VE_F + 4, k2.c0); w.st0(VE_F + 5, k2.c1);
    w.st0(VE_F + 6, k3.c0); w.st0(VE_F + 7, k3.c1); w.st0(VE_F + 8, f4.c0); w.st0(VE_F + 9, f4.c1); w.st0(VE_F + 10, f5.c0); w.st0(VE_F + 11, f5.c1);
  }
}

// ---- one Miller step of THREE table-driven pairs in one operation (group stage): [f <- f^2,] f <- f * l0(P0) * l1(P1) * l2(P2) -----------------
// Same in-flight scheme as bn254_vm.h::vm_miller_step: k0..k3 parked, k4 / k5 in registers, only the last product goes back to the workspace.
template <bool DO_SQR, class W>
BN_HD void vm_miller_step_fixed3(W& w, int e, const FixedLine& l0, int e_p0, bool inf0, const FixedLine& l1, int e_p1, bool inf1,
                                 const FixedLine& l2, int e_p2, bool inf2) {
  Fp2 f4, f5;
  {
    Fp2 k0 = vld2(w, e), k1 = vld2(w, e + 2), k2 = vld2(w, e + 4), k3 = vld2(w, e + 6), k4 = vld2(w, e + 8), k5 = vld2(w, e + 10);
    if constexpr (DO_SQR) {
      Fp2 x3 = fp2_mul_xi(k3), x4 = fp2_mul_xi(k4), x5 = fp2_mul_xi(k5);
      w.park(0, fp2_dotp(pp(k0, k0), pp2(k1, x5), pp2(k2, x4), pp(k3, x3)));
      w.park(1, fp2_dotp(pp2(k0, k1), pp2(k2, x5), pp2(k3, x4)));
      w.park(2, fp2_dotp(pp2(k0, k2), pp(k1, k1), pp2(k3, x5), pp(k4, x4)));
      w.park(3, fp2_dotp(pp2(k0, k3), pp2(k1, k2), pp2(k4, x5)));
      f4 = fp2_dotp(pp2(k0, k4), pp2(k1, k3), pp(k2, k2), pp(k5, x5));
      f5 = fp2_dotp(pp2(k0, k5), pp2(k1, k4), pp2(k2, k3));
    } else {
      w.park(0, k0); w.park(1, k1); w.park(2, k2); w.park(3, k3); f4 = k4; f5 = k5;
    }
  }
  for (int t = 0; t < 2; t++) {   // the first two lines: plain dot products, result stays in flight
    const FixedLine& l = t == 0 ? l0 : l1;
    const bool inf = t == 0 ? inf0 : inf1;
    const int e_p = t == 0 ? e_p0 : e_p1;
    BN_SCHED_FENCE();
    Fp px = w.ld(e_p), d0 = w.ld(e_p + 1);
    Fp2 d3 = fp2_mul_fp(l.m, px);
    Fp2 x3 = fp2_mul_xi(d3);
    const Fp2 &d4 = l.c, &x4 = l.xc;
    Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3), k4 = f4, k5 = f5;
    w.park(0, fp2_select(inf, k0, fp2_dot_line(d0, k0, x3, k5, x4, k3)));
    w.park(1, fp2_select(inf, k1, fp2_dot_line(d0, k1, d3, k0, x4, k4)));
    w.park(2, fp2_select(inf, k2, fp2_dot_line(d0, k2, d3, k1, x4, k5)));
    w.park(3, fp2_select(inf, k3, fp2_dot_line(d0, k3, d3, k2, d4, k0)));
    f4 = fp2_select(inf, k4, fp2_dot_line(d0, k4, d3, k3, d4, k1));
    f5 = fp2_select(inf, k5, fp2_dot_line(d0, k5, d3, k4, d4, k2));
  }
  {
    BN_SCHED_FENCE();
    Fp px = w.ld(e_p2), d0 = w.ld(e_p2 + 1);
    Fp2 d3 = fp2_mul_fp(l2.m, px);
    Fp2 x3 = fp2_mul_xi(d3);
    const Fp2 &d4 = l2.c, &x4 = l2.xc;
    Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3);
    const Fp2 &k4 = f4, &k5 = f5;
    vst2(w, e, fp2_select(inf2, k0, fp2_dotk(kfp(k0, d0), kp(x3, k5), kp(x4, k3))));
    vst2(w, e + 2, fp2_select(inf2, k1, fp2_dotk(kfp(k1, d0), kp(d3, k0), kp(x4, k4))));
    vst2(w, e + 4, fp2_select(inf2, k2, fp2_dotk(kfp(k2, d0), kp(d3, k1), kp(x4, k5))));
    vst2(w, e + 6, fp2_select(inf2, k3, fp2_dotk(kfp(k3, d0), kp(d3, k2), kp(d4, k0))));
    vst2(w, e + 8, fp2_select(inf2, k4, fp2_dotk(kfp(k4, d0), kp(d3, k3), kp(d4, k1))));
    vst2(w, e + 10, fp2_select(inf2, k5, fp2_dotk(kfp(k5, d0), kp(d3, k4), kp(d4, k2))));
  }
}

// a lane that contributes nothing to its group (loader error, r-torsion failure, wrong input count): f = 1, C' = O, t = 0
template <class W>
BN_HD void vm_rlc_neutral(W& w, int n_public, bool with_f = true) {
  if (with_f) {
    w.st(VE_F, fp_one());
    for (int e = 1; e < 12; e++) w.st(VE_F + e, fp_zero());
  }
  w.st(RLC_C, fp_zero()); w.st(RLC_C + 1, fp_one()); w.st(RLC_C + 2, fp_zero());
  for (int j = 0; j <= n_public; j++) w.st(RLC_T + j, fr8_to_slot(fr8_zero()));
}
// fold the partner lane (element ids + RLC_HI) into this one: f *= f', C' += C'', t_j += t_j'
template <class W>
BN_HD void vm_rlc_fold(W& w, int n_public, bool with_f = true) {
  if (with_f) vm_f12_mul(w, VE_F, VE_F, RLC_HI + VE_F, false);
  {
    G1Proj a, b;
    a.x = w.ld(RLC_C); a.y = w.ld(RLC_C + 1); a.z = w.ld(RLC_C + 2);
    b.x = w.ld(RLC_HI + RLC_C); b.y = w.ld(RLC_HI + RLC_C + 1); b.z = w.ld(RLC_HI + RLC_C + 2);
    BN_SETB(a.x, 1.01, 0.5); BN_SETB(a.y, 1.01, 0.5); BN_SETB(a.z, 1.01, 0.5); BN_SETB(b.x, 1.01, 0.5); BN_SETB(b.y, 1.01, 0.5); BN_SETB(b.z, 1.01, 0.5);
    G1Proj s = g1_add(a, b);
    w.st(RLC_C, fp_reduce(s.x)); w.st(RLC_C + 1, fp_reduce(s.y)); w.st(RLC_C + 2, fp_reduce(s.z));
  }
  for (int j = 0; j <= n_public; j++)
    w.st(RLC_T + j, fr8_to_slot(fr8_add(fr8_from_slot(w.ld(RLC_T + j)), fr8_from_slot(w.ld(RLC_HI + RLC_T + j)))));
}
// sum over the MSM_FW_WINDOWS windows (MSM_FW_BITS bits each, bn254_fw.h) of a 256-bit scalar: acc += sum_w T[w][digit_w(k) - 1].  TL(w, d) returns table entry d of window w.
template <class TL>
BN_HD G1Proj g1_window_sum(G1Proj acc, const Fr8& k, const TL& entry) {
  uint32_t sw[8];
#pragma unroll
  for (int i = 0; i < 8; i++) sw[i] = k.w[i];
  for (int wi = 0; wi < MSM_FW_WINDOWS; wi++) {
    const uint32_t dig = sw[0] & MSM_FW_ENTRIES;
#pragma unroll
    for (int i = 0; i < 7; i++) sw[i] = (sw[i] >> MSM_FW_BITS) | (sw[i + 1] << (32 - MSM_FW_BITS));
    sw[7] >>= MSM_FW_BITS;
    if (dig != 0) acc = g1_add_mixed(acc, entry(wi, (int)dig - 1));
  }
  return acc;
}
// group stage, one lane per group: the three G1 arguments of the table-driven pairs from the folded scalars and the folded C'
//   VE_LX/LY <- t_0 K_0 + sum_j t_j K_j      VE_CX/CY <- affine(C')      VE_AX/AY <- t_0 (-alpha)      RLC_ACC <- 1
// TAB(b, w, d): entry d of window w of base b (0: -alpha, 1: K_0, 1 + j: K_j).  Returns the identity flags (bit 0: L, 1: C, 2: alpha term).
template <class W, class TAB>
BN_HD int vm_rlc_group_points(W& w, int n_public, const TAB& tab) {
  const Fr8 t0 = fr8_from_slot(w.ld(RLC_T));
  G1Proj Pa = g1_window_sum(g1_identity(), t0, [&](int wi, int d) { return tab(0, wi, d); });
  G1Proj L = g1_window_sum(g1_identity(), t0, [&](int wi, int d) { return tab(1, wi, d); });
  for (int j = 1; j <= n_public; j++) {
    const Fr8 tj = fr8_from_slot(w.ld(RLC_T + j));
    L = g1_window_sum(L, tj, [&](int wi, int d) { return tab(1 + j, wi, d); });
  }
  G1Proj Cg; Cg.x = w.ld(RLC_C); Cg.y = w.ld(RLC_C + 1); Cg.z = w.ld(RLC_C + 2);
  BN_SETB(Cg.x, 1.01, 0.5); BN_SETB(Cg.y, 1.01, 0.5); BN_SETB(Cg.z, 1.01, 0.5);
  const bool ia = g1_is_identity(Pa), il = g1_is_identity(L), ic = g1_is_identity(Cg);
  // one inversion for the three denominators (identity: denominator replaced by 1, the point becomes (0, 1) + flag)
  Fp za = fp_select(ia, fp_one(), fp_reduce(fp_norm(Pa.z))), zl = fp_select(il, fp_one(), fp_reduce(fp_norm(L.z))), zc = fp_select(ic, fp_one(), fp_reduce(fp_norm(Cg.z)));
  Fp zal = fp_mul(za, zl);
  Fp inv = fp_inv(fp_mul(zal, zc));
  Fp izc = fp_mul(inv, zal);
  Fp izal = fp_mul(inv, zc);
  Fp iza = fp_mul(izal, zl), izl = fp_mul(izal, za);
  w.st(VE_AX, fp_mul(Pa.x, iza)); w.st(VE_AY, fp_select(ia, fp_one(), fp_mul(Pa.y, iza)));
  w.st(VE_LX, fp_mul(L.x, izl)); w.st(VE_LY, fp_select(il, fp_one(), fp_mul(L.y, izl)));
  w.st(VE_CX, fp_mul(Cg.x, izc)); w.st(VE_CY, fp_select(ic, fp_one(), fp_mul(Cg.y, izc)));
  w.st(RLC_ACC, fp_one());
  for (int e = 1; e < 12; e++) w.st(RLC_ACC + e, fp_zero());
  return (il ? 1 : 0) | (ic ? 2 : 0) | (ia ? 4 : 0);
}
}  // namespace bn254
