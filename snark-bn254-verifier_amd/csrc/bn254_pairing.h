// bn254_pairing.h -- optimal-ate Miller loop (one variable-Q pair + any number of fixed-Q pairs with precomputed
// affine line tables) and final exponentiation.  Device + host.
//
// Replaces bn::pairing / bn::pairing_batch (reference groth16/verify.rs:70,73; plonk/kzg.rs:180).  Same mathematical
// value: f_{6u+2,Q}(P) * l_{T,pi(Q)}(P) * l_{T,-pi^2(Q)}(P), raised to (p^12-1)/r * c with the same constant c as the
// exp_by_neg_z chain of the reference's `bn`, so GT bytes can be compared with the oracle directly.
//
// Line of the D-type twist (untwist (x', y') -> (x' w^2, y' w^3)), up to factors in proper subfields of Fp12, which the
// final exponentiation removes:   l(P) = r0 yP + (r1 xP) w + r2 w^3   (variable Q, projective T: G2Line)
//                                  l(P) = yP + (m xP) w + c w^3          (fixed Q, affine T: m = -lambda, c = lambda x_T - y_T)
#pragma once
#include "bn254_curve.h"

namespace bn254 {

// one precomputed step of a fixed G2 argument: m, c and xi*c (3 Fp2 = 54 limbs = 216 bytes)
struct FixedLine { Fp2 m, c, xc; };

// ---- multiply f by line values ---------------------------------------------------------------------------------
BN_HD Fp12 miller_mul_var(const Fp12& f, const G2Line& l, const G1Aff& p) {
  return fp12_mul_by_034(f, fp2_mul_fp(l.r0, p.y), fp2_mul_fp(l.r1, p.x), l.r2);
}
BN_HD Fp12 miller_mul_fixed_aff(const Fp12& f, const FixedLine& l, const G1Aff& p) {
  return fp12_mul_by_034_fp(f, p.y, fp2_mul_fp(l.m, p.x), l.c, l.xc);
}
// same for a G1 argument that may be the identity (encoded x = 0, y = 1, inf = true): the line value must then be 1, i.e. the
// w^3 coefficient has to vanish as well (bn::pairing_batch skips pairs with an identity operand)
BN_HD Fp12 miller_mul_fixed_aff_or_inf(const Fp12& f, const FixedLine& l, const G1Aff& p, bool inf) {
  Fp2 z = fp2_zero();
  return fp12_mul_by_034_fp(f, p.y, fp2_mul_fp(l.m, p.x), fp2_select(inf, z, l.c), fp2_select(inf, z, l.xc));
}

// ---- host-side table construction for a fixed Q (runs once per verifying key) -------------------------------------
// Affine coordinates: slopes need inversions, which are irrelevant here.  Returns false if an exceptional step occurs
// (Q of small order: impossible for a valid vk element).
inline bool fixed_line_table(FixedLine* out /* BN_ATE_STEPS */, const G2Aff& q) {
  G2Aff t = q;
  G2Aff nq = g2_neg(q);
  int n = 0;
  auto dbl = [&](void) -> bool {
    Fp2 den = fp2_dbl(t.y);
    if (fp2_is_zero(den)) return false;
    Fp2 lam = fp2_mul(fp2_mul_small(fp2_sqr(t.x), 3), fp2_inv(den));
    out[n].m = fp2_neg(lam);
    out[n].c = fp2_sub(fp2_mul(lam, t.x), t.y);
    out[n].xc = fp2_mul_xi(out[n].c);
    n++;
    Fp2 x3 = fp2_sub(fp2_sqr(lam), fp2_dbl(t.x));
    Fp2 y3 = fp2_sub(fp2_mul(lam, fp2_sub(t.x, x3)), t.y);
    t.x = fp2_reduce(x3); t.y = fp2_reduce(y3);
    return true;
  };
  auto add = [&](const G2Aff& s) -> bool {
    Fp2 den = fp2_sub(t.x, s.x);
    if (fp2_is_zero(den)) return false;
    Fp2 lam = fp2_mul(fp2_sub(t.y, s.y), fp2_inv(den));
    out[n].m = fp2_neg(lam);
    out[n].c = fp2_sub(fp2_mul(lam, t.x), t.y);
    out[n].xc = fp2_mul_xi(out[n].c);
    n++;
    Fp2 x3 = fp2_sub(fp2_sub(fp2_sqr(lam), t.x), s.x);
    Fp2 y3 = fp2_sub(fp2_mul(lam, fp2_sub(t.x, x3)), t.y);
    t.x = fp2_reduce(x3); t.y = fp2_reduce(y3);
    return true;
  };
  for (int i = 1; i < BN_ATE_NAF_LEN; i++) {
    if (!dbl()) return false;
    int d = BN_ATE_NAF[i];
    if (d != 0 && !add(d > 0 ? q : nq)) return false;
  }
  G2Aff q1 = g2_psi_affine(q);
  G2Aff q2 = g2_neg(g2_psi2_affine(q));
  if (!add(q1)) return false;
  if (!add(q2)) return false;
  return n == BN_ATE_STEPS;
}

// ---- generic Miller loop: one variable pair (pa, qb) and NF fixed pairs (affine G1 points) -----------------------
// Used on the host (vk preparation: e(alpha, beta)), by the PlonK path and by tests; the Groth16 kernel has its own
// loop with the same structure (bn254_kernels.hip) so that its G1 arguments can stay projective.
template <int NF>
BN_HD Fp12 miller_loop(const G1Aff& pa, const G2Aff& qb, const G1Aff* pf, const FixedLine* const* tabs) {
  Fp12 f = fp12_one();
  G2Proj t = g2_from_affine(qb);
  G2Aff nqb = g2_neg(qb);
  int idx = 0;
  for (int i = 1; i < BN_ATE_NAF_LEN; i++) {
    f = fp12_sqr(f);
    {
      G2Line l = g2_double_step(t);
      f = miller_mul_var(f, l, pa);
      for (int k = 0; k < NF; k++) f = miller_mul_fixed_aff(f, tabs[k][idx], pf[k]);
      idx++;
    }
    int d = BN_ATE_NAF[i];
    if (d != 0) {
      G2Line l = g2_add_step(t, d > 0 ? qb : nqb);
      f = miller_mul_var(f, l, pa);
      for (int k = 0; k < NF; k++) f = miller_mul_fixed_aff(f, tabs[k][idx], pf[k]);
      idx++;
    }
  }
  G2Aff q1 = g2_psi_affine(qb);
  G2Aff q2 = g2_neg(g2_psi2_affine(qb));
  {
    G2Line l = g2_add_step(t, q1);
    f = miller_mul_var(f, l, pa);
    for (int k = 0; k < NF; k++) f = miller_mul_fixed_aff(f, tabs[k][idx], pf[k]);
    idx++;
  }
  {
    G2Line l = g2_add_step(t, q2);
    f = miller_mul_var(f, l, pa);
    for (int k = 0; k < NF; k++) f = miller_mul_fixed_aff(f, tabs[k][idx], pf[k]);
    idx++;
  }
  return f;
}

// ---- final exponentiation --------------------------------------------------------------------------------------------
// x^u on the cyclotomic subgroup over NAF(u) (inverse = conjugate there)
BN_HD Fp12 fp12_exp_u(const Fp12& x) {
  Fp12 acc = x;
  Fp12 xc = fp12_conj(x);
  for (int i = 1; i < BN_U_NAF_LEN; i++) {
    acc = fp12_cyclo_sqr(acc);
    int d = BN_U_NAF[i];
    if (d != 0) acc = fp12_mul(acc, d > 0 ? x : xc);  // public constant: wave-uniform
  }
  return acc;
}
BN_HD Fp12 final_exp_easy(const Fp12& f) {
  Fp12 a = fp12_mul(fp12_conj(f), fp12_inv(f));  // f^(p^6 - 1)
  return fp12_mul(fp12_frob(a, 2), a);            // ^(p^2 + 1)
}
// hard part: exponent p^3 (12u^3+6u^2+4u-1) + p^2 (12u^3+6u^2+6u) + p (12u^3+6u^2+4u) + (12u^3+12u^2+6u+1), a multiple of
// (p^4-p^2+1)/r coprime to r (Fuentes-Castaneda et al. / Duquesne-Ghammam scheduling)
BN_HD Fp12 final_exp_hard(const Fp12& m) {
  Fp12 t0 = fp12_conj(fp12_exp_u(m));         // m^-u
  t0 = fp12_cyclo_sqr(t0);                     // -2u
  Fp12 t1 = fp12_cyclo_sqr(t0);                // -4u
  t1 = fp12_mul(t0, t1);                       // -6u
  Fp12 t2 = fp12_conj(fp12_exp_u(t1));         // 6u^2
  Fp12 t3 = fp12_conj(t1);                     // 6u
  t1 = fp12_mul(t2, t3);                       // 6u^2 + 6u
  t3 = fp12_cyclo_sqr(t2);                     // 12u^2
  Fp12 t4 = fp12_exp_u(t3);                    // 12u^3
  t4 = fp12_mul(t1, t4);                       // 12u^3 + 6u^2 + 6u
  t3 = fp12_mul(t0, t4);                       // 12u^3 + 6u^2 + 4u
  t0 = fp12_mul(t2, t4);                       // 12u^3 + 12u^2 + 6u
  t0 = fp12_mul(m, t0);                        // + 1
  t2 = fp12_frob(t3, 1);
  t0 = fp12_mul(t2, t0);
  t2 = fp12_frob(t4, 2);
  t0 = fp12_mul(t2, t0);
  t2 = fp12_mul(fp12_conj(m), t3);             // 12u^3 + 6u^2 + 4u - 1
  t2 = fp12_frob(t2, 3);
  return fp12_mul(t2, t0);
}
BN_HD Fp12 final_exponentiation(const Fp12& f) { return final_exp_hard(final_exp_easy(f)); }

}  // namespace bn254
