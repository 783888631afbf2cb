// tests/hostsim/hostsim.cpp -- TEST HARNESS: compiles the product's device arithmetic headers
// (snark-bn254-verifier_amd/csrc/*.h) for the host CPU with the bound tracker enabled, so that the exact algorithms the
// HIP kernels run can be checked against the oracle in this GPU-less container, and every value-bound assumption
// of bn254_fp.h is asserted on every operation.  Not part of the product; never loaded by it.
#define BN_TRACK_BOUNDS 1
#include "../../snark-bn254-verifier_amd/csrc/bn254_fp.h"
#include "../../snark-bn254-verifier_amd/csrc/bn254_tower.h"
#ifdef HS_WITH_CURVE
#include "../../snark-bn254-verifier_amd/csrc/bn254_curve.h"
#include "../../snark-bn254-verifier_amd/csrc/bn254_pairing.h"
#include "../../snark-bn254-verifier_amd/csrc/bn254_vm.h"
#include "../../snark-bn254-verifier_amd/csrc/bn254_rlc.h"
#include "../../snark-bn254-verifier_amd/csrc/bn254_msm.h"
#endif
#include <cstring>
using namespace bn254;

static Fp fp_in(const uint8_t* be, int inflate) {
  uint32_t w[8]; words_from_be(w, be);
  Fp r = fp_from_words(w);
  Fp pl = fp_from_limbs(BN_P);
  for (int i = 0; i < (inflate < 0 ? -inflate : inflate); i++) r = inflate < 0 ? fp_sub(r, pl) : fp_add(r, pl);
  return r;
}
static void fp_out(uint8_t* be, const Fp& a) { uint32_t w[8]; fp_to_words(w, a); words_to_be(be, w); }
static Fp2 fp2_in(const uint8_t* b, int inf) { Fp2 r; r.c0 = fp_in(b, inf); r.c1 = fp_in(b + 32, -inf); return r; }
static void fp2_out(uint8_t* b, const Fp2& a) { fp_out(b, a.c0); fp_out(b + 32, a.c1); }
static Fp12 fp12_in(const uint8_t* b, int inf) {
  Fp12 r;
  Fp2* c[6] = {&r.c0.c0, &r.c0.c1, &r.c0.c2, &r.c1.c0, &r.c1.c1, &r.c1.c2};
  for (int i = 0; i < 6; i++) *c[i] = fp2_in(b + 64 * i, (i & 1) ? inf : -inf);
  return r;
}
static void fp12_out(uint8_t* b, const Fp12& a) {
  const Fp2* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  for (int i = 0; i < 6; i++) fp2_out(b + 64 * i, *c[i]);
}

extern "C" {
// op: 0 add 1 sub 2 mul 3 inv (binary GCD) 4 inv (Fermat) 5 neg 6 sqr 7 reduce(canon) 8 lincomb_reduce(9a - b) 9 is_zero -> o[31]
void hs_fp_op(int op, uint8_t* o, const uint8_t* a, const uint8_t* b, int inflate) {
  Fp x = fp_in(a, inflate), y = fp_in(b, -inflate), r = fp_zero();
  switch (op) {
    case 0: r = fp_add(x, y); break;
    case 1: r = fp_sub(x, y); break;
    case 2: r = fp_mul(x, y); break;
    case 3: r = fp_inv(x); break;
    case 4: r = fp_inv_fermat(x); break;
    case 5: r = fp_neg(x); break;
    case 6: r = fp_sqr(x); break;
    case 7: r = fp_reduce(x); break;
    case 8: r = fp_lincomb_reduce(9, x, -1, y); break;
    case 9: memset(o, 0, 32); o[31] = fp_is_zero(x) ? 1 : 0; return;
  }
  fp_out(o, r);
}
// op: 0 add 1 sub 2 mul 3 inv 5 sqr 6 mul_xi; Karatsuba dot (fp2_dotk): 10 x*y, 11 x^2, 12 2xy - yx + x^2 (= xy + x^2), 13 -2 x*y + x*(y.c0)
void hs_fp2_op(int op, uint8_t* o, const uint8_t* a, const uint8_t* b, int inflate) {
  Fp2 x = fp2_in(a, inflate), y = fp2_in(b, inflate), r = fp2_zero();
  switch (op) {
    case 0: r = fp2_add(x, y); break;
    case 1: r = fp2_sub(x, y); break;
    case 2: r = fp2_mul(x, y); break;
    case 3: r = fp2_inv(x); break;
    case 5: r = fp2_sqr(x); break;
    case 6: r = fp2_mul_xi(x); break;
    case 10: r = fp2_dotk(kp(x, y)); break;
    case 11: r = fp2_dotk(ksq(x)); break;
    case 12: r = fp2_dotk(kp2(x, y), km(y, x), ksq(x)); break;
    case 13: r = fp2_dotk(km2(x, y), kfp(x, y.c0)); break;
  }
  fp2_out(o, r);
}
// op: 0 mul 1 sqr 2 inv 3 frob1 4 frob2 5 frob3 6 cyclo_sqr 7 conj 8 mul_by_034 (b = d0|d3|d4 Fp2s) 9 mul_by_034_fp (b = d0(Fp, first 32 B)|..|d3|d4)
void hs_fp12_op(int op, uint8_t* o, const uint8_t* a, const uint8_t* b, int inflate) {
  Fp12 x = fp12_in(a, inflate), r = fp12_one();
  switch (op) {
    case 0: r = fp12_mul(x, fp12_in(b, inflate)); break;
    case 1: r = fp12_sqr(x); break;
    case 2: r = fp12_inv(x); break;
    case 3: r = fp12_frob(x, 1); break;
    case 4: r = fp12_frob(x, 2); break;
    case 5: r = fp12_frob(x, 3); break;
    case 6: r = fp12_cyclo_sqr(x); break;
    case 7: r = fp12_conj(x); break;
    case 8: r = fp12_mul_by_034(x, fp2_in(b, inflate), fp2_in(b + 64, inflate), fp2_in(b + 128, inflate)); break;
    case 9: { Fp2 d4 = fp2_in(b + 128, inflate); r = fp12_mul_by_034_fp(x, fp_in(b, inflate), fp2_in(b + 64, inflate), d4, fp2_mul_xi(d4)); break; }
  }
  fp12_out(o, r);
}
#ifdef HS_WITH_CURVE
#include "hostsim_curve.inc"
#endif
}
