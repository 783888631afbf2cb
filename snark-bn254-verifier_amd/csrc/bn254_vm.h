// bn254_vm.h -- the Miller loop and the final exponentiation as a sequence of out-of-line Fp12-level operations whose operands
// live in the per-proof SoA workspace (HBM / Infinity Cache), not in registers.
//
// Why: an Fp12 value is 108 registers; squaring it needs the 6 input coefficients, 3 xi-multiples and an output coefficient
// live at once (~200 VGPRs).  Keeping f, the G2 point T, the current line and the caller's temporaries in registers ACROSS
// operations is what made the first version spill 4-5 KB per lane (profiles/r01_ubench_tower_karatsuba.txt).  Here every
// operation loads its operands, has the whole 256-VGPR budget of a 2-waves-per-SIMD kernel to itself, and stores its result:
// ~0.6 MB of workspace traffic per proof, all of it 4-byte-per-lane coalesced rows.
//
// Everything is templated on the workspace accessor W (ld/st of an Fp by element index): the kernels instantiate it with buffer
// loads (bn254_kernels.hip: SGPR row offset + one VGPR lane offset), tests/hostsim with plain arrays, so the exact operation
// sequence the GPU runs is checked against the oracle on the CPU.
#pragma once
#include "bn254_pairing.h"

namespace bn254 {

// ---- workspace element map (unit: one Fp = 9 limbs).  Fp12 values are stored in w-power order k0..k5, 2 Fp each. ----------------
enum {
  VE_AX = 0, VE_AY = 1, VE_B = 2 /* x.c0 x.c1 y.c0 y.c1 */, VE_CX = 6, VE_CY = 7, VE_LX = 8, VE_LY = 9,
  VE_F = 10,                 // Miller accumulator, then m = easy part of the final exponentiation
  VE_T = 22,                 // Miller loop: running G2 point (X, Y, Z)
  VE_S0 = 22, VE_S1 = 34, VE_S2 = 46, VE_S3 = 58, VE_S4 = 70,  // final exponentiation slots (S0 aliases T)
  VE_P3 = 82,                   // x^3 of the windowed exp-by-u
  VE_TMPA = 94, VE_TMPB = 100,  // two Fp6 temporaries of the general Fp12 product
  VE_P5 = 106, VE_P7 = 118,     // x^5, x^7
  VE_COUNT = 130
};

template <class W> BN_HD Fp2 vld2(W& w, int e) { Fp2 r; r.c0 = w.ld(e); r.c1 = w.ld(e + 1); return r; }
template <class W> BN_HD void vst2(W& w, int e, const Fp2& a) { w.st(e, a.c0); w.st(e + 1, a.c1); }
template <class W> BN_HD Fp6 vld6(W& w, int e) { Fp6 r; r.c0 = vld2(w, e); r.c1 = vld2(w, e + 2); r.c2 = vld2(w, e + 4); return r; }
template <class W> BN_HD void vst6(W& w, int e, const Fp6& a) { vst2(w, e, a.c0); vst2(w, e + 2, a.c1); vst2(w, e + 4, a.c2); }

// ---- f <- f^2 (in place) ------------------------------------------------------------------------------------------------------------
template <class W>
BN_HD void vm_f12_sqr(W& w, int e) {
  Fp2 k0 = vld2(w, e), k1 = vld2(w, e + 2), k2 = vld2(w, e + 4), k3 = vld2(w, e + 6), k4 = vld2(w, e + 8), k5 = vld2(w, e + 10);
  Fp2 x3 = fp2_mul_xi(k3), x4 = fp2_mul_xi(k4), x5 = fp2_mul_xi(k5);
  vst2(w, e, fp2_dotp(pp(k0, k0), pp2(k1, x5), pp2(k2, x4), pp(k3, x3)));
  vst2(w, e + 2, fp2_dotp(pp2(k0, k1), pp2(k2, x5), pp2(k3, x4)));
  vst2(w, e + 4, fp2_dotp(pp2(k0, k2), pp(k1, k1), pp2(k3, x5), pp(k4, x4)));
  vst2(w, e + 6, fp2_dotp(pp2(k0, k3), pp2(k1, k2), pp2(k4, x5)));
  vst2(w, e + 8, fp2_dotp(pp2(k0, k4), pp2(k1, k3), pp(k2, k2), pp(k5, x5)));
  vst2(w, e + 10, fp2_dotp(pp2(k0, k5), pp2(k1, k4), pp2(k2, k3)));
}
// ---- f <- f * (yP + (m xP) w + c w^3): precomputed affine line of a fixed G2 argument; inf: the G1 point is the identity -----------
template <class W>
BN_HD void vm_f12_mul_line_fixed(W& w, int e, const FixedLine& l, int e_px, bool inf) {
  // Karatsuba Fp2 dot products (fp2_dotk); c and xi*c stay wave-uniform (scalar registers on the GPU: their operand sums and
  // negations are scalar work), and a lane whose G1 point is the identity keeps f (line value 1) by selecting the OUTPUT, so
  // no per-lane copy of the line exists.  220 VGPRs, no scratch, -4.4 % against the plain form (tools/kbench: LF_K vs LF_CUR).
  Fp px = w.ld(e_px), d0 = w.ld(e_px + 1);
  Fp2 d3 = fp2_mul_fp(l.m, px);
  Fp2 x3 = fp2_mul_xi(d3);
  const Fp2 &d4 = l.c, &x4 = l.xc;
  Fp2 k0 = vld2(w, e), k1 = vld2(w, e + 2), k2 = vld2(w, e + 4), k3 = vld2(w, e + 6), k4 = vld2(w, e + 8), k5 = vld2(w, e + 10);
  vst2(w, e, fp2_select(inf, k0, fp2_dotk(kfp(k0, d0), kp(x3, k5), kp(x4, k3))));
  vst2(w, e + 2, fp2_select(inf, k1, fp2_dotk(kfp(k1, d0), kp(d3, k0), kp(x4, k4))));
  vst2(w, e + 4, fp2_select(inf, k2, fp2_dotk(kfp(k2, d0), kp(d3, k1), kp(x4, k5))));
  vst2(w, e + 6, fp2_select(inf, k3, fp2_dotk(kfp(k3, d0), kp(d3, k2), kp(d4, k0))));
  vst2(w, e + 8, fp2_select(inf, k4, fp2_dotk(kfp(k4, d0), kp(d3, k3), kp(d4, k1))));
  vst2(w, e + 10, fp2_select(inf, k5, fp2_dotk(kfp(k5, d0), kp(d3, k4), kp(d4, k2))));
}
// ---- f <- f * l1(P1) * l2(P2): both precomputed lines of a Miller step in one operation.  The intermediate product does not go
// back to the workspace: four of its six coefficients are parked with w.park() (LDS on the GPU: 72 dwords per lane), two stay
// in registers.  The first product uses plain dot products (176 VGPRs + the two resident coefficients), the second the Karatsuba
// form; together 222 VGPRs, no scratch, 12 % faster than two launches (tools/kbench: LF2 against 2 x LF_K).
template <class W>
BN_HD void vm_f12_mul_line_fixed2(W& w, int e, const FixedLine& l1, int e_px1, bool inf1, const FixedLine& l2, int e_px2, bool inf2) {
  Fp2 r4, r5;
  {
    Fp px = w.ld(e_px1), d0 = w.ld(e_px1 + 1);
    Fp2 d3 = fp2_mul_fp(l1.m, px);
    Fp2 x3 = fp2_mul_xi(d3);
    const Fp2 &d4 = l1.c, &x4 = l1.xc;
    Fp2 k0 = vld2(w, e), k1 = vld2(w, e + 2), k2 = vld2(w, e + 4), k3 = vld2(w, e + 6), k4 = vld2(w, e + 8), k5 = vld2(w, e + 10);
    w.park(0, fp2_select(inf1, k0, fp2_dot_line(d0, k0, x3, k5, x4, k3)));
    w.park(1, fp2_select(inf1, k1, fp2_dot_line(d0, k1, d3, k0, x4, k4)));
    w.park(2, fp2_select(inf1, k2, fp2_dot_line(d0, k2, d3, k1, x4, k5)));
    w.park(3, fp2_select(inf1, k3, fp2_dot_line(d0, k3, d3, k2, d4, k0)));
    r4 = fp2_select(inf1, k4, fp2_dot_line(d0, k4, d3, k3, d4, k1));
    r5 = fp2_select(inf1, k5, fp2_dot_line(d0, k5, d3, k4, d4, k2));
  }
  {
    Fp px = w.ld(e_px2), d0 = w.ld(e_px2 + 1);
    Fp2 d3 = fp2_mul_fp(l2.m, px);
    Fp2 x3 = fp2_mul_xi(d3);
    const Fp2 &d4 = l2.c, &x4 = l2.xc;
    Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3);
    const Fp2 &k4 = r4, &k5 = r5;
    vst2(w, e, fp2_select(inf2, k0, fp2_dotk(kfp(k0, d0), kp(x3, k5), kp(x4, k3))));
    vst2(w, e + 2, fp2_select(inf2, k1, fp2_dotk(kfp(k1, d0), kp(d3, k0), kp(x4, k4))));
    vst2(w, e + 4, fp2_select(inf2, k2, fp2_dotk(kfp(k2, d0), kp(d3, k1), kp(x4, k5))));
    vst2(w, e + 6, fp2_select(inf2, k3, fp2_dotk(kfp(k3, d0), kp(d3, k2), kp(d4, k0))));
    vst2(w, e + 8, fp2_select(inf2, k4, fp2_dotk(kfp(k4, d0), kp(d3, k3), kp(d4, k1))));
    vst2(w, e + 10, fp2_select(inf2, k5, fp2_dotk(kfp(k5, d0), kp(d3, k4), kp(d4, k2))));
  }
}
// ---- fused Miller step of the variable pair: T <- 2T (or T + Q), f <- f * line(P); the line never leaves the registers -----------
template <class W>
BN_HD void vm_f12_mul_line_regs(W& w, int e, const G2Line& l, int e_px) {
  Fp px = w.ld(e_px), py = w.ld(e_px + 1);
  Fp2 d0 = fp2_mul_fp(l.r0, py), d3 = fp2_mul_fp(l.r1, px), d4 = l.r2;
  Fp2 x3 = fp2_mul_xi(d3), x4 = fp2_mul_xi(d4);
  Fp2 k0 = vld2(w, e), k1 = vld2(w, e + 2), k2 = vld2(w, e + 4), k3 = vld2(w, e + 6), k4 = vld2(w, e + 8), k5 = vld2(w, e + 10);
  vst2(w, e, fp2_dotp(pp(d0, k0), pp(x3, k5), pp(x4, k3)));
  vst2(w, e + 2, fp2_dotp(pp(d0, k1), pp(d3, k0), pp(x4, k4)));
  vst2(w, e + 4, fp2_dotp(pp(d0, k2), pp(d3, k1), pp(x4, k5)));
  vst2(w, e + 6, fp2_dotp(pp(d0, k3), pp(d3, k2), pp(d4, k0)));
  vst2(w, e + 8, fp2_dotp(pp(d0, k4), pp(d3, k3), pp(d4, k1)));
  vst2(w, e + 10, fp2_dotp(pp(d0, k5), pp(d3, k4), pp(d4, k2)));
}
template <class W>
BN_HD void vm_miller_dbl_var(W& w, int e_t, int e, int e_px) {
  G2Proj t; t.x = vld2(w, e_t); t.y = vld2(w, e_t + 2); t.z = vld2(w, e_t + 4);
  G2Line l = g2_double_step(t);
  vst2(w, e_t, t.x); vst2(w, e_t + 2, t.y); vst2(w, e_t + 4, t.z);
  vm_f12_mul_line_regs(w, e, l, e_px);
}
template <class W>
BN_HD void vm_miller_add_var(W& w, int e_t, int e_b, int which, int e, int e_px) {
  G2Aff q; q.x = vld2(w, e_b); q.y = vld2(w, e_b + 2);
  if (which == 1) q = g2_neg(q);
  else if (which == 2) q = g2_psi_affine(q);
  else if (which == 3) q = g2_neg(g2_psi2_affine(q));
  G2Proj t; t.x = vld2(w, e_t); t.y = vld2(w, e_t + 2); t.z = vld2(w, e_t + 4);
  G2Line l = g2_add_step(t, q);
  vst2(w, e_t, t.x); vst2(w, e_t + 2, t.y); vst2(w, e_t + 4, t.z);
  vm_f12_mul_line_regs(w, e, l, e_px);
}

// ---- fused doubling step: f <- f^2, T <- 2T, f <- f * line(P).  f^2 is not written back: four coefficients are parked (w.park),
// two stay in registers across the G2 doubling, then the line product consumes them.
template <class W>
BN_HD void vm_miller_sqr_dbl_var(W& w, int e_t, int e, int e_px) {
  Fp2 r4, r5;
  {
    Fp2 k0 = vld2(w, e), k1 = vld2(w, e + 2), k2 = vld2(w, e + 4), k3 = vld2(w, e + 6), k4 = vld2(w, e + 8), k5 = vld2(w, e + 10);
    Fp2 x3 = fp2_mul_xi(k3), x4 = fp2_mul_xi(k4), x5 = fp2_mul_xi(k5);
    w.park(0, fp2_dotp(pp(k0, k0), pp2(k1, x5), pp2(k2, x4), pp(k3, x3)));
    w.park(1, fp2_dotp(pp2(k0, k1), pp2(k2, x5), pp2(k3, x4)));
    w.park(2, fp2_dotp(pp2(k0, k2), pp(k1, k1), pp2(k3, x5), pp(k4, x4)));
    w.park(3, fp2_dotp(pp2(k0, k3), pp2(k1, k2), pp2(k4, x5)));
    r4 = fp2_dotp(pp2(k0, k4), pp2(k1, k3), pp(k2, k2), pp(k5, x5));
    r5 = fp2_dotp(pp2(k0, k5), pp2(k1, k4), pp2(k2, k3));
  }
  G2Line l;
  {
    G2Proj t; t.x = vld2(w, e_t); t.y = vld2(w, e_t + 2); t.z = vld2(w, e_t + 4);
    l = g2_double_step(t);
    vst2(w, e_t, t.x); vst2(w, e_t + 2, t.y); vst2(w, e_t + 4, t.z);
  }
  BN_SCHED_FENCE();  // keep the parked coefficients out of the registers until the G2 step is done
  Fp px = w.ld(e_px), py = w.ld(e_px + 1);
  Fp2 d0 = fp2_mul_fp(l.r0, py), d3 = fp2_mul_fp(l.r1, px), d4 = l.r2;
  Fp2 x3 = fp2_mul_xi(d3), x4 = fp2_mul_xi(d4);
  BN_SCHED_FENCE();
  Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3);
  const Fp2 &k4 = r4, &k5 = r5;
  vst2(w, e, fp2_dotp(pp(d0, k0), pp(x3, k5), pp(x4, k3)));
  vst2(w, e + 2, fp2_dotp(pp(d0, k1), pp(d3, k0), pp(x4, k4)));
  vst2(w, e + 4, fp2_dotp(pp(d0, k2), pp(d3, k1), pp(x4, k5)));
  vst2(w, e + 6, fp2_dotp(pp(d0, k3), pp(d3, k2), pp(d4, k0)));
  vst2(w, e + 8, fp2_dotp(pp(d0, k4), pp(d3, k3), pp(d4, k1)));
  vst2(w, e + 10, fp2_dotp(pp(d0, k5), pp(d3, k4), pp(d4, k2)));
}

// ---- one whole Miller step in one operation: [f <- f^2,] T <- 2T or T + Q, f <- f * line_T(A) * line_0(P0) * line_1(P1) ------------
// Between the stages f travels "in flight": coefficients k0..k3 parked (w.park / w.unpark: LDS on the GPU), k4 and k5 in registers;
// only the last product writes f back to the workspace.  Every stage reads slot j before it overwrites it (each output
// coefficient r_j depends on the input coefficient k_j), so the four slots can be reused in place.
// kind: 0 doubling, 1..4 addition of +B, -B, psi(B), -psi^2(B).
template <bool DO_SQR, class W>
BN_HD void vm_miller_step(W& w, int kind, int e_t, int e_b, int e, int e_pa, const FixedLine& l0, int e_p0, bool inf0,
                          const FixedLine& l1, int e_p1, bool inf1) {
  Fp2 f4, f5;
  if constexpr (DO_SQR) {
    Fp2 k0 = vld2(w, e), k1 = vld2(w, e + 2), k2 = vld2(w, e + 4), k3 = vld2(w, e + 6), k4 = vld2(w, e + 8), k5 = vld2(w, e + 10);
    Fp2 x3 = fp2_mul_xi(k3), x4 = fp2_mul_xi(k4), x5 = fp2_mul_xi(k5);
    w.park(0, fp2_dotp(pp(k0, k0), pp2(k1, x5), pp2(k2, x4), pp(k3, x3)));
    w.park(1, fp2_dotp(pp2(k0, k1), pp2(k2, x5), pp2(k3, x4)));
    w.park(2, fp2_dotp(pp2(k0, k2), pp(k1, k1), pp2(k3, x5), pp(k4, x4)));
    w.park(3, fp2_dotp(pp2(k0, k3), pp2(k1, k2), pp2(k4, x5)));
    f4 = fp2_dotp(pp2(k0, k4), pp2(k1, k3), pp(k2, k2), pp(k5, x5));
    f5 = fp2_dotp(pp2(k0, k5), pp2(k1, k4), pp2(k2, k3));
  }
  G2Line l;
  {
    G2Proj t; t.x = vld2(w, e_t); t.y = vld2(w, e_t + 2); t.z = vld2(w, e_t + 4);
    if (kind == 0) {
      l = g2_double_step(t);
    } else {
      G2Aff q; q.x = vld2(w, e_b); q.y = vld2(w, e_b + 2);
      if (kind == 2) q = g2_neg(q);
      else if (kind == 3) q = g2_psi_affine(q);
      else if (kind == 4) q = g2_neg(g2_psi2_affine(q));
      l = g2_add_step(t, q);
    }
    vst2(w, e_t, t.x); vst2(w, e_t + 2, t.y); vst2(w, e_t + 4, t.z);
  }
  {  // the variable pair's line
    Fp px = w.ld(e_pa), py = w.ld(e_pa + 1);
    Fp2 d0 = fp2_mul_fp(l.r0, py), d3 = fp2_mul_fp(l.r1, px), d4 = l.r2;
    Fp2 x3 = fp2_mul_xi(d3), x4 = fp2_mul_xi(d4);
    Fp2 k0, k1, k2, k3, k4, k5;
    if constexpr (DO_SQR) { k0 = w.unpark(0); k1 = w.unpark(1); k2 = w.unpark(2); k3 = w.unpark(3); k4 = f4; k5 = f5; }
    else { k0 = vld2(w, e); k1 = vld2(w, e + 2); k2 = vld2(w, e + 4); k3 = vld2(w, e + 6); k4 = vld2(w, e + 8); k5 = vld2(w, e + 10); }
    w.park(0, fp2_dotp(pp(d0, k0), pp(x3, k5), pp(x4, k3)));
    w.park(1, fp2_dotp(pp(d0, k1), pp(d3, k0), pp(x4, k4)));
    w.park(2, fp2_dotp(pp(d0, k2), pp(d3, k1), pp(x4, k5)));
    w.park(3, fp2_dotp(pp(d0, k3), pp(d3, k2), pp(d4, k0)));
    Fp2 n4 = fp2_dotp(pp(d0, k4), pp(d3, k3), pp(d4, k1));
    f5 = fp2_dotp(pp(d0, k5), pp(d3, k4), pp(d4, k2));
    f4 = n4;
  }
  {  // first key-side line: plain dot products (the two in-flight coefficients leave no room for the Karatsuba operand sums)
    Fp px = w.ld(e_p0), d0 = w.ld(e_p0 + 1);
    Fp2 d3 = fp2_mul_fp(l0.m, px);
    Fp2 x3 = fp2_mul_xi(d3);
    const Fp2 &d4 = l0.c, &x4 = l0.xc;
    Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3), k4 = f4, k5 = f5;
    w.park(0, fp2_select(inf0, k0, fp2_dot_line(d0, k0, x3, k5, x4, k3)));
    w.park(1, fp2_select(inf0, k1, fp2_dot_line(d0, k1, d3, k0, x4, k4)));
    w.park(2, fp2_select(inf0, k2, fp2_dot_line(d0, k2, d3, k1, x4, k5)));
    w.park(3, fp2_select(inf0, k3, fp2_dot_line(d0, k3, d3, k2, d4, k0)));
    f4 = fp2_select(inf0, k4, fp2_dot_line(d0, k4, d3, k3, d4, k1));
    f5 = fp2_select(inf0, k5, fp2_dot_line(d0, k5, d3, k4, d4, k2));
  }
  {  // second key-side line: Karatsuba dot products, result to the workspace
    Fp px = w.ld(e_p1), d0 = w.ld(e_p1 + 1);
    Fp2 d3 = fp2_mul_fp(l1.m, px);
    Fp2 x3 = fp2_mul_xi(d3);
    const Fp2 &d4 = l1.c, &x4 = l1.xc;
    Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3);
    const Fp2 &k4 = f4, &k5 = f5;
    vst2(w, e, fp2_select(inf1, k0, fp2_dotk(kfp(k0, d0), kp(x3, k5), kp(x4, k3))));
    vst2(w, e + 2, fp2_select(inf1, k1, fp2_dotk(kfp(k1, d0), kp(d3, k0), kp(x4, k4))));
    vst2(w, e + 4, fp2_select(inf1, k2, fp2_dotk(kfp(k2, d0), kp(d3, k1), kp(x4, k5))));
    vst2(w, e + 6, fp2_select(inf1, k3, fp2_dotk(kfp(k3, d0), kp(d3, k2), kp(d4, k0))));
    vst2(w, e + 8, fp2_select(inf1, k4, fp2_dotk(kfp(k4, d0), kp(d3, k3), kp(d4, k1))));
    vst2(w, e + 10, fp2_select(inf1, k5, fp2_dotk(kfp(k5, d0), kp(d3, k4), kp(d4, k2))));
  }
}

// ---- a RUN of Miller steps in one operation: n_dbl consecutive doubling steps, then optionally one addition step -------------------------------
// f is loaded once, travels "in flight" through the whole run (k0..k3 parked -- LDS on the GPU -- k4, k5 in registers: the state the stages of
// vm_miller_step hand each other) and is stored once; the line tables of step s come from `lines.get(t, s)` (scalar loads indexed by the step on
// the GPU).  The operation sequence per step is that of vm_miller_step; only the round trips of f through the workspace between the steps of a run
// are gone (864 of a step's 1512 bytes per proof) and with them the launch boundaries.  The running G2 point still goes through the workspace.
template <class W> BN_HD void mf_load(W& w, int e, Fp2& f4, Fp2& f5) {
  w.park(0, vld2(w, e)); w.park(1, vld2(w, e + 2)); w.park(2, vld2(w, e + 4)); w.park(3, vld2(w, e + 6));
  f4 = vld2(w, e + 8); f5 = vld2(w, e + 10);
}
template <class W> BN_HD void mf_store(W& w, int e, const Fp2& f4, const Fp2& f5) {
  vst2(w, e, w.unpark(0)); vst2(w, e + 2, w.unpark(1)); vst2(w, e + 4, w.unpark(2)); vst2(w, e + 6, w.unpark(3));
  vst2(w, e + 8, f4); vst2(w, e + 10, f5);
}
template <class W> BN_HD void mf_sqr(W& w, Fp2& f4, Fp2& f5) {
  Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3), k4 = f4, k5 = f5;
  Fp2 x3 = fp2_mul_xi(k3), x4 = fp2_mul_xi(k4), x5 = fp2_mul_xi(k5);
  w.park(0, fp2_dotp(pp(k0, k0), pp2(k1, x5), pp2(k2, x4), pp(k3, x3)));
  w.park(1, fp2_dotp(pp2(k0, k1), pp2(k2, x5), pp2(k3, x4)));
  w.park(2, fp2_dotp(pp2(k0, k2), pp(k1, k1), pp2(k3, x5), pp(k4, x4)));
  w.park(3, fp2_dotp(pp2(k0, k3), pp2(k1, k2), pp2(k4, x5)));
  Fp2 n4 = fp2_dotp(pp2(k0, k4), pp2(k1, k3), pp(k2, k2), pp(k5, x5));
  f5 = fp2_dotp(pp2(k0, k5), pp2(k1, k4), pp2(k2, k3));
  f4 = n4;
}
template <class W> BN_HD G2Line mf_g2(W& w, int kind, int e_t, int e_b) {
  G2Line l;
  G2Proj t; t.x = vld2(w, e_t); t.y = vld2(w, e_t + 2); t.z = vld2(w, e_t + 4);
  if (kind == 0) {
    l = g2_double_step(t);
  } else {
    G2Aff q; q.x = vld2(w, e_b); q.y = vld2(w, e_b + 2);
    if (kind == 2) q = g2_neg(q);
    else if (kind == 3) q = g2_psi_affine(q);
    else if (kind == 4) q = g2_neg(g2_psi2_affine(q));
    l = g2_add_step(t, q);
  }
  vst2(w, e_t, t.x); vst2(w, e_t + 2, t.y); vst2(w, e_t + 4, t.z);
  return l;
}
template <class W> BN_HD void mf_line_var(W& w, const G2Line& l, int e_pa, Fp2& f4, Fp2& f5) {
  Fp px = w.ld(e_pa), py = w.ld(e_pa + 1);
  Fp2 d0 = fp2_mul_fp(l.r0, py), d3 = fp2_mul_fp(l.r1, px), d4 = l.r2;
  Fp2 x3 = fp2_mul_xi(d3), x4 = fp2_mul_xi(d4);
  Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3), k4 = f4, k5 = f5;
  w.park(0, fp2_dotp(pp(d0, k0), pp(x3, k5), pp(x4, k3)));
  w.park(1, fp2_dotp(pp(d0, k1), pp(d3, k0), pp(x4, k4)));
  w.park(2, fp2_dotp(pp(d0, k2), pp(d3, k1), pp(x4, k5)));
  w.park(3, fp2_dotp(pp(d0, k3), pp(d3, k2), pp(d4, k0)));
  Fp2 n4 = fp2_dotp(pp(d0, k4), pp(d3, k3), pp(d4, k1));
  f5 = fp2_dotp(pp(d0, k5), pp(d3, k4), pp(d4, k2));
  f4 = n4;
}
template <class W> BN_HD void mf_line_fixed(W& w, const FixedLine& l0, int e_p0, bool inf0, Fp2& f4, Fp2& f5) {   // plain dot products
  Fp px = w.ld(e_p0), d0 = w.ld(e_p0 + 1);
  Fp2 d3 = fp2_mul_fp(l0.m, px);
  Fp2 x3 = fp2_mul_xi(d3);
  const Fp2 &d4 = l0.c, &x4 = l0.xc;
  Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3), k4 = f4, k5 = f5;
  w.park(0, fp2_select(inf0, k0, fp2_dot_line(d0, k0, x3, k5, x4, k3)));
  w.park(1, fp2_select(inf0, k1, fp2_dot_line(d0, k1, d3, k0, x4, k4)));
  w.park(2, fp2_select(inf0, k2, fp2_dot_line(d0, k2, d3, k1, x4, k5)));
  w.park(3, fp2_select(inf0, k3, fp2_dot_line(d0, k3, d3, k2, d4, k0)));
  f4 = fp2_select(inf0, k4, fp2_dot_line(d0, k4, d3, k3, d4, k1));
  f5 = fp2_select(inf0, k5, fp2_dot_line(d0, k5, d3, k4, d4, k2));
}
template <class W> BN_HD void mf_line_fixed_k(W& w, const FixedLine& l1, int e_p1, bool inf1, Fp2& f4, Fp2& f5) {   // Karatsuba dot products
  Fp px = w.ld(e_p1), d0 = w.ld(e_p1 + 1);
  Fp2 d3 = fp2_mul_fp(l1.m, px);
  Fp2 x3 = fp2_mul_xi(d3);
  const Fp2 &d4 = l1.c, &x4 = l1.xc;
  Fp2 k0 = w.unpark(0), k1 = w.unpark(1), k2 = w.unpark(2), k3 = w.unpark(3);
  const Fp2 k4 = f4, k5 = f5;
  w.park(0, fp2_select(inf1, k0, fp2_dotk(kfp(k0, d0), kp(x3, k5), kp(x4, k3))));
  w.park(1, fp2_select(inf1, k1, fp2_dotk(kfp(k1, d0), kp(d3, k0), kp(x4, k4))));
  w.park(2, fp2_select(inf1, k2, fp2_dotk(kfp(k2, d0), kp(d3, k1), kp(x4, k5))));
  w.park(3, fp2_select(inf1, k3, fp2_dotk(kfp(k3, d0), kp(d3, k2), kp(d4, k0))));
  f4 = fp2_select(inf1, k4, fp2_dotk(kfp(k4, d0), kp(d3, k3), kp(d4, k1)));
  f5 = fp2_select(inf1, k5, fp2_dotk(kfp(k5, d0), kp(d3, k4), kp(d4, k2)));
}
template <class W, class LINES, class KINDS>
BN_HD void vm_miller_run(W& w, const LINES& lines, const KINDS& kinds, int s_begin, int s_end, int e_t, int e_b, int e, int e_pa, int e_p0, bool inf0, int e_p1, bool inf1) {
  // steps [s_begin, s_end) of the loop; kinds.get(s): 0 doubling (with the squaring of f except in step 0), 1..4 addition of +B, -B, psi(B), -psi^2(B).
  // The doubling and the addition side of the (wave-uniform) branch each carry their own G2 formulas AND their own variable-line product, so that
  // the only state that crosses the join is f in flight; the two key-side line products are shared.  (Joining right after the G2 step instead --
  // the line and the new point as merged values -- costs 165 spilled registers.)
  Fp2 f4, f5;
  mf_load(w, e, f4, f5);
  for (int s = s_begin; s < s_end; s++) {
    const int kind = kinds.get(s);
    if (kind == 0) {
      if (s != 0) mf_sqr(w, f4, f5);
      BN_SCHED_FENCE();
      G2Line l = mf_g2(w, 0, e_t, e_b);
      mf_line_var(w, l, e_pa, f4, f5);
    } else {
      G2Line l = mf_g2(w, kind, e_t, e_b);
      mf_line_var(w, l, e_pa, f4, f5);
    }
    BN_SCHED_FENCE();
    mf_line_fixed(w, lines.get(0, s), e_p0, inf0, f4, f5);
    mf_line_fixed_k(w, lines.get(1, s), e_p1, inf1, f4, f5);
  }
  mf_store(w, e, f4, f5);
}

// The same run for TWO table-driven pairs and no variable pair (PlonK's KZG check e(P0, g2[0]) e(P1, g2[1]), plonk/kzg.rs:175-187): f <- [f^2] l0(P0) l1(P1)
// per step, no G2 arithmetic at all -- the accumulator in flight from the first step to the last.
template <class W, class LINES, class KINDS>
BN_HD void vm_miller_run_fixed2(W& w, const LINES& lines, const KINDS& kinds, int s_begin, int s_end, int e, int e_p0, bool inf0, int e_p1, bool inf1) {
  Fp2 f4, f5;
  mf_load(w, e, f4, f5);
  for (int s = s_begin; s < s_end; s++) {
    if (kinds.get(s) == 0 && s != 0) mf_sqr(w, f4, f5);
    BN_SCHED_FENCE();
    mf_line_fixed(w, lines.get(0, s), e_p0, inf0, f4, f5);
    mf_line_fixed_k(w, lines.get(1, s), e_p1, inf1, f4, f5);
  }
  mf_store(w, e, f4, f5);
}

// ---- r-torsion test of B from the point the Miller loop has already computed ----------------------------------------------------------
// After vm_miller_program the running point is T = [6u+2]B + psi(B) - psi^2(B).  For B on the twist E'(Fp2):
//     B in G2  <=>  T == -psi^3(B)          (T finite)
// "=>" is the optimal-ate relation 6u+2 + p - p^2 + p^3 = 0 (mod r) with psi = [p] on G2.  "<=": #E'(Fp2) = r h2 with
// gcd(r, h2) = 1, psi satisfies X^2 - tX + p on E', and the resultant of chi(X) = X^3 - X^2 + X + (6u+2) with X^2 - tX + p is
// coprime to h2 (tests/test_oracle_arith.py::test_ate_relation_is_a_subgroup_test computes it), so chi(psi) is injective on the
// h2-torsion: a point with a component outside G2 cannot satisfy the relation.  Exceptional cases of the incomplete addition
// formulas (only possible outside G2) zero Z, which sticks and is rejected here.  Same accept set as the reference's
// [r-1]Q + Q == O (bn's AffineG2::new, reference converter.rs:152) and as gnark's relation (bn254_curve.h::g2_in_subgroup),
// at the cost of six Fp2 products instead of a 63-bit scalar multiplication.
template <class W>
BN_HD bool vm_g2_ate_check(W& w, int e_t, int e_b) {
  G2Aff b; b.x = vld2(w, e_b); b.y = vld2(w, e_b + 2);
  Fp2 sx = fp2_mul(fp2_conj(b.x), frob_coeff(3, 2));            // psi^3(B).x
  Fp2 sy = fp2_neg(fp2_mul(fp2_conj(b.y), frob_coeff(3, 3)));   // -psi^3(B).y
  Fp2 X = vld2(w, e_t), Y = vld2(w, e_t + 2), Z = vld2(w, e_t + 4);
  return !fp2_is_zero(Z) & fp2_eq(X, fp2_mul(sx, Z)) & fp2_eq(Y, fp2_mul(sy, Z));
}

// ---- general Fp12 product dst <- a * b (dst may alias a or b), Karatsuba over Fp6; temporaries: 3 parking slots ----------------------
// tower halves in k-order storage: c0 = (k0, k2, k4), c1 = (k1, k3, k5)
template <class W> BN_HD Fp6 vld_half(W& w, int e, int h) { Fp6 r; r.c0 = vld2(w, e + 2 * h); r.c1 = vld2(w, e + 4 + 2 * h); r.c2 = vld2(w, e + 8 + 2 * h); return r; }
template <class W> BN_HD void vst_half(W& w, int e, int h, const Fp6& a) { vst2(w, e + 2 * h, a.c0); vst2(w, e + 4 + 2 * h, a.c1); vst2(w, e + 8 + 2 * h, a.c2); }
// conj_b: multiply by conj(b) = b^(p^6) (the inverse of b on the cyclotomic subgroup): the odd half of b is negated on load
template <class W>
BN_HD void vm_f12_mul(W& w, int e_dst, int e_a, int e_b, bool conj_b = false) {
  // Round 5 (tools/kbench MUL_T3F: 632 against 695 us per 2^20 products): v0 = a0 b0 waits in the parking slots (LDS); v1 = a1 b1 stays in registers just long enough to
  // form c0 = v0 + (xi v1.c2, v1.c0, v1.c1) -- held back in registers -- and t = v0 + v1, which takes v0's place in the parking slots; a0 and b0 are then read a SECOND
  // time straight into the operand sums a0 + a1, b0 + b1 (a1 and b1 are still in registers), and only after that is the first half of dst written, so dst may alias a or
  // b; finally s = (a0 + a1)(b0 + b1) and c1 = s - t.  Until round 4 both halves of a and of b were read twice and two thirds of v1 went through a workspace temporary:
  // 2451 B of HBM traffic per product (counters) against 1728 B now (1296 B is the floor: operands once, result once).
  { Fp6 v0 = fp6_mul(vld_half(w, e_a, 0), vld_half(w, e_b, 0)); w.park(0, v0.c0); w.park(1, v0.c1); w.park(2, v0.c2); }
  BN_SCHED_FENCE();
  Fp6 a1 = vld_half(w, e_a, 1), b1 = vld_half(w, e_b, 1);
  if (conj_b) b1 = fp6_neg(b1);
  Fp6 c0;
  {
    const Fp6 v1 = fp6_mul(a1, b1);
    BN_SCHED_FENCE();
    { const Fp2 x = w.unpark(0); c0.c0 = fp2_add(x, fp2_mul_xi(v1.c2)); w.park(0, fp2_add(x, v1.c0)); }
    { const Fp2 x = w.unpark(1); c0.c1 = fp2_add(x, v1.c0); w.park(1, fp2_add(x, v1.c1)); }
    { const Fp2 x = w.unpark(2); c0.c2 = fp2_add(x, v1.c1); w.park(2, fp2_add(x, v1.c2)); }
  }
  BN_SCHED_FENCE();
  a1 = fp6_add(vld_half(w, e_a, 0), a1);
  b1 = fp6_add(vld_half(w, e_b, 0), b1);
  vst_half(w, e_dst, 0, c0);
  BN_SCHED_FENCE();
  const Fp6 s = fp6_mul(a1, b1);
  BN_SCHED_FENCE();
  Fp6 c1;
  c1.c0 = fp2_sub(s.c0, w.unpark(0)); c1.c1 = fp2_sub(s.c1, w.unpark(1)); c1.c2 = fp2_sub(s.c2, w.unpark(2));
  vst_half(w, e_dst, 1, c1);
}
// ---- Granger-Scott squaring dst <- src^2 on the cyclotomic subgroup (dst may alias src) ---------------------------------------------
template <class W>
BN_HD void vm_f12_cyclo_sqr(W& w, int e_dst, int e_src) {
  // tower names -> k index: c0.c0 = k0, c1.c1 = k3, c1.c0 = k1, c0.c2 = k4, c0.c1 = k2, c1.c2 = k5.
  // The first pair only touches (k0, k3); the other two pairs read and write (k1, k2, k4, k5), so those four are loaded before
  // either result is stored (dst may alias src).  Splitting the work this way keeps the live set near 150 registers.
  {
    Fp2 k0 = vld2(w, e_src), k3 = vld2(w, e_src + 6);
    Fp2 za, zb;
    gs_pair(za, zb, k0, k3, k0, k3, false);
    vst2(w, e_dst, za); vst2(w, e_dst + 6, zb);
  }
  {
    Fp2 k1 = vld2(w, e_src + 2), k2 = vld2(w, e_src + 4), k4 = vld2(w, e_src + 8), k5 = vld2(w, e_src + 10);
    Fp2 z2, z5, z4, z1;
    gs_pair(z2, z5, k1, k4, k2, k5, false);   // z.c0.c1 (k2), z.c1.c2 (k5)
    gs_pair(z4, z1, k2, k5, k4, k1, true);    // z.c0.c2 (k4), z.c1.c0 (k1)
    vst2(w, e_dst + 4, z2); vst2(w, e_dst + 10, z5); vst2(w, e_dst + 8, z4); vst2(w, e_dst + 2, z1);
  }
}
// ---- dst <- src^(2^count): a run of Granger-Scott squarings with all six coefficients resident in registers -------------------------
// (one load and one store of the element per run instead of per squaring; exp-by-u has runs of 4 to 7 squarings between products)
template <class W>
BN_HD void vm_f12_cyclo_sqr_n(W& w, int e_dst, int e_src, int count) {
  Fp2 k0 = vld2(w, e_src), k1 = vld2(w, e_src + 2), k2 = vld2(w, e_src + 4), k3 = vld2(w, e_src + 6), k4 = vld2(w, e_src + 8), k5 = vld2(w, e_src + 10);
  for (int it = 0; it < count; it++) {
    Fp2 n0, n3, n2, n5, n4, n1;
    gs_pair(n0, n3, k0, k3, k0, k3, false);
    k0 = n0; k3 = n3;
    gs_pair(n2, n5, k1, k4, k2, k5, false);
    gs_pair(n4, n1, k2, k5, k4, k1, true);
    k1 = n1; k2 = n2; k4 = n4; k5 = n5;
  }
  vst2(w, e_dst, k0); vst2(w, e_dst + 2, k1); vst2(w, e_dst + 4, k2); vst2(w, e_dst + 6, k3); vst2(w, e_dst + 8, k4); vst2(w, e_dst + 10, k5);
}
// ---- cheap unary operations ------------------------------------------------------------------------------------------------------------
template <class W>
BN_HD void vm_f12_copy(W& w, int e_dst, int e_src) { for (int k = 0; k < 12; k++) w.st(e_dst + k, w.ld(e_src + k)); }
template <class W>
BN_HD void vm_f12_conj(W& w, int e_dst, int e_src) {  // negate the odd coefficients
  for (int k = 0; k < 6; k++) { Fp2 x = vld2(w, e_src + 2 * k); vst2(w, e_dst + 2 * k, (k & 1) ? fp2_neg(x) : x); }
}
template <class W>
BN_HD void vm_f12_frob(W& w, int e_dst, int e_src, int j) {
  const bool odd = (j & 1) != 0;
  for (int k = 0; k < 6; k++) {
    Fp2 x = vld2(w, e_src + 2 * k);
    if (odd) x = fp2_conj(x);
    if (k > 0) { Fp2 c = frob_coeff(j, k); x = (j == 2) ? fp2_mul_fp(x, c.c0) : fp2_mul(x, c); }
    vst2(w, e_dst + 2 * k, x);
  }
}
// ---- dst <- 1 / src --------------------------------------------------------------------------------------------------------------------
template <class W>
BN_HD void vm_f12_inv(W& w, int e_dst, int e_src) {
  Fp6 d;
  {
    Fp6 s0 = fp6_sqr(vld_half(w, e_src, 0));
    Fp6 s1 = fp6_mul_v(fp6_sqr(vld_half(w, e_src, 1)));
    d = fp6_sub(s0, s1);
  }
  Fp6 di = fp6_inv(d);
  { Fp6 r0 = fp6_mul_plain(vld_half(w, e_src, 0), di); Fp6 r1 = fp6_neg(fp6_mul_plain(vld_half(w, e_src, 1), di)); vst_half(w, e_dst, 0, r0); vst_half(w, e_dst, 1, r1); }
}
template <class W>
BN_HD bool vm_f12_eq_const(W& w, int e, const int32_t* target /* 12 Fp in k-order, uniform memory */) {
  bool ok = true;
  for (int k = 0; k < 12; k++) {
    Fp t;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) t.v[l] = target[k * BN_NL + l];
    BN_SETB(t, 1.0, 0.5);
    ok &= fp_eq(w.ld(e + k), t);
  }
  return ok;
}

// =======================================================================================================================================
// Programs.  OPS is the dispatcher that decides how an operation is invoked: LaunchOps (bn254_kernels.hip) enqueues one kernel
// launch per operation, HostOps (tests/hostsim) calls the operation directly on plain arrays.
// =======================================================================================================================================
// step table of the optimal-ate loop: one entry per Miller step (BN_ATE_STEPS = 88): 0 = doubling (with f squaring),
// 1 = add +B, 2 = add -B, 3 = add pi(B), 4 = add -pi^2(B)
BN_HD int miller_step_kind(int s) {
  // expand NAF(6u+2) on the fly: steps are (dbl [, add]) per digit after the leading one, then the two Frobenius additions
  int idx = 0;
  for (int it = 1; it < BN_ATE_NAF_LEN; it++) {
    if (idx == s) return 0;
    idx++;
    int d = BN_ATE_NAF[it];
    if (d != 0) { if (idx == s) return d > 0 ? 1 : 2; idx++; }
  }
  return s == idx ? 3 : 4;
}

template <class OPS>
BN_HD void vm_miller_program(OPS& ops, const uint8_t* step_kinds /* BN_ATE_STEPS, uniform */, bool with_fixed_pairs = true) {
  // f = 1, T = B are set by the caller
  for (int s = 0; s < BN_ATE_STEPS; s++) {
    int kind = ops.uni(step_kinds[s]);
    if (with_fixed_pairs) {
      // the whole step is one operation: [squaring,] G2 step, variable line, both key-side lines (tables 0 / 1, step s)
      ops.miller_step(kind == 0 && s != 0, kind, s, VE_T, VE_B, VE_F, VE_AX, VE_LX, VE_CX);
      continue;
    }
    if (kind == 0) { if (s != 0) ops.miller_sqr_dbl_var(VE_T, VE_F, VE_AX); else ops.miller_dbl_var(VE_T, VE_F, VE_AX); }
    else ops.miller_add_var(VE_T, VE_B, kind - 1, VE_F, VE_AX);
  }
}
// the same loop as RUNS of consecutive steps, one operation each (ops.miller_run(first step, one past the last step)): `per_run` steps per
// operation (BN_ATE_STEPS: the whole loop is ONE operation)
template <class OPS>
BN_HD void vm_miller_program_runs(OPS& ops, int per_run) {
  if (per_run < 1) per_run = 1;
  for (int s = 0; s < BN_ATE_STEPS; s += per_run)
    ops.miller_run(s, s + per_run < BN_ATE_STEPS ? s + per_run : BN_ATE_STEPS, VE_T, VE_B, VE_F, VE_AX, VE_LX, VE_CX);
}
// x^u on the cyclotomic subgroup: dst <- src^u (dst != src), width-4 signed windows of u (BN_U_W4: digits +-1, +-3, +-5, +-7).
// x^3, x^5, x^7 go to VE_P3/P5/P7; a negative digit multiplies by the conjugate (= inverse) of the table entry.
// 63 cyclotomic squarings + 16 products (NAF(u): 62 + 23 and two conjugations).
template <class OPS>
BN_HD void vm_exp_u(OPS& ops, int e_dst, int e_src) {
  ops.f12_cyclo_sqr(e_dst, e_src);                 // x^2
  ops.f12_mul(VE_P3, e_dst, e_src, false);
  ops.f12_mul(VE_P5, VE_P3, e_dst, false);
  ops.f12_mul(VE_P7, VE_P5, e_dst, false);
  // leading digit of BN_U_W4 is +1; x^2 in e_dst is no longer needed once the table is built
  int run = 0, first = 1;
  for (int i = 1; i < BN_U_W4_LEN; i++) {
    run++;
    int d = BN_U_W4[i];
    if (d != 0) {
      // the first run squares src straight into dst (leading digit +1: the accumulator starts as x)
      ops.f12_cyclo_sqr_n(e_dst, first ? e_src : e_dst, run);
      run = 0; first = 0;
      int a = d < 0 ? -d : d;
      ops.f12_mul(e_dst, e_dst, a == 1 ? e_src : a == 3 ? VE_P3 : a == 5 ? VE_P5 : VE_P7, d < 0);
    }
  }
  // BN_U_W4 ends in a non-zero digit (u is odd): no trailing run
}
// final exponentiation of VE_F; the result ends in VE_S0 (same exponent as bn254_pairing.h::final_exponentiation)
template <class OPS>
BN_HD void vm_final_exp_program(OPS& ops) {
  // easy part: m = f^((p^6-1)(p^2+1)) -> VE_F
  ops.f12_inv(VE_S0, VE_F);
  ops.f12_conj(VE_S1, VE_F);
  ops.f12_mul(VE_S0, VE_S1, VE_S0, false);
  ops.f12_frob(VE_S1, VE_S0, 2);
  ops.f12_mul(VE_F, VE_S1, VE_S0);
  // hard part
  vm_exp_u(ops, VE_S0, VE_F); ops.f12_conj(VE_S0, VE_S0);      // t0 = m^-u
  ops.f12_cyclo_sqr(VE_S0, VE_S0);                                     // -2u
  ops.f12_cyclo_sqr(VE_S1, VE_S0);                                     // -4u
  ops.f12_mul(VE_S1, VE_S0, VE_S1);                                    // t1 = -6u
  vm_exp_u(ops, VE_S2, VE_S1); ops.f12_conj(VE_S2, VE_S2);     // t2 = 6u^2
  ops.f12_conj(VE_S3, VE_S1);                                          // t3 = 6u
  ops.f12_mul(VE_S1, VE_S2, VE_S3);                                    // t1 = 6u^2 + 6u
  ops.f12_cyclo_sqr(VE_S3, VE_S2);                                     // t3 = 12u^2
  vm_exp_u(ops, VE_S4, VE_S3);                                  // t4 = 12u^3
  ops.f12_mul(VE_S4, VE_S1, VE_S4);                                    // t4 = 12u^3 + 6u^2 + 6u
  ops.f12_mul(VE_S3, VE_S0, VE_S4);                                    // t3 = 12u^3 + 6u^2 + 4u
  ops.f12_mul(VE_S0, VE_S2, VE_S4);                                    // t0 = 12u^3 + 12u^2 + 6u
  ops.f12_mul(VE_S0, VE_F, VE_S0);                                     // + 1
  ops.f12_frob(VE_S2, VE_S3, 1); ops.f12_mul(VE_S0, VE_S2, VE_S0);
  ops.f12_frob(VE_S2, VE_S4, 2); ops.f12_mul(VE_S0, VE_S2, VE_S0);
  ops.f12_conj(VE_S2, VE_F); ops.f12_mul(VE_S2, VE_S2, VE_S3);        // 12u^3 + 6u^2 + 4u - 1
  ops.f12_frob(VE_S2, VE_S2, 3);
  ops.f12_mul(VE_S0, VE_S2, VE_S0);
}

}  // namespace bn254
