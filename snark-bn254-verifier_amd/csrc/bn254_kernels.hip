// bn254_kernels.hip -- the gfx950 kernels of the Groth16 batch verifier, one proof per lane.
//
// Pipeline per batch (DESIGN.md "Kernels"):
//   k_g16_prepare     parse 256 proof bytes (coalesced through LDS), range / on-curve checks of A, B, C, Montgomery
//                     conversion, L = K0 + sum x_i K_i by fixed-base 8-bit windows      (groth16/converter.rs:14-26, verify.rs:53-63)
//   k_vm_init, k_miller_step_dbl, k_miller_step_add
//                     the shared Miller loop f = Miller(A,B) * lines_G(L) * lines_D(C), one launch per STEP (squaring, G2 step,
//                     three line products; bn254_vm.h::vm_miller_step); G/D line tables shared by the batch  (verify.rs:73-77)
//   k_g16_subgroup    r-torsion test of B from the loop's final G2 point, status precedence   (converter.rs:152)
//   k_f12_inv, k_f12_conj, k_f12_frob, k_f12_mul, k_f12_cyclo_sqr(_n), k_f12_copy
//                     f^((p^12-1)/r), bn254_vm.h::vm_final_exp_program                          (verify.rs:77)
//   k_g16_compare     == e(alpha, beta) -> status byte                                          (verify.rs:77)
//
// Workspace (bn254_vm.h element map): element e, digit l, proof i at dword (e * 9 + l) * n + i, accessed through ONE buffer
// descriptor: the row offset (e, l) is wave-uniform and travels in an SGPR (soffset), the lane offset i * 4 is one VGPR shared
// by every access, so no per-access address arithmetic exists and a wave-level access is one contiguous 256-byte segment
// (buffer_load_dword / buffer_store_dword).  Lanes past the end of the batch get an out-of-range offset: the descriptor's bounds
// check returns 0 for their loads and drops their stores.
//
// Every operation is its own kernel: it gets the full 256-VGPR budget of a 2-waves-per-SIMD launch and keeps nothing in
// registers between operations.  The host walks the program (LaunchOps below) and enqueues ~210 launches per batch.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "bn254_devws.h"
#include "bn254_rlc.h"
#include "bn254_g16_plan.h"

namespace bn254 {


__global__ void __launch_bounds__(256, 2) k_vm_init(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status) {  // f = 1, T = B
  VM_KERNEL_PROLOGUE();
  w.st(VE_F, fp_one());
  for (int e = 1; e < 12; e++) w.st(VE_F + e, fp_zero());
  for (int e = 0; e < 4; e++) w.st(VE_T + e, w.ld(VE_B + e));
  w.st(VE_T + 4, fp_one()); w.st(VE_T + 5, fp_zero());
}
__global__ void __launch_bounds__(256, 2) k_f12_sqr(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int e) { VM_KERNEL_PROLOGUE(); vm_f12_sqr(w, e); }
__global__ void __launch_bounds__(256, 2) k_f12_mul_line_fixed(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int e, const int32_t* __restrict__ entry, int e_px, int inf_mask) {
  VM_KERNEL_PROLOGUE();
  FixedLine l; l.m = uni_ld2(entry); l.c = uni_ld2(entry + 2 * BN_NL); l.xc = uni_ld2(entry + 4 * BN_NL);
  vm_f12_mul_line_fixed(w, e, l, e_px, (st & inf_mask) != 0);  // inf_mask: the status bit that marks this pair's G1 point as the identity
}
__global__ void __launch_bounds__(256, 2) k_f12_mul_line_fixed2(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int e, const int32_t* __restrict__ entry0,
                                                                int e_px0, int inf_mask0, const int32_t* __restrict__ entry1, int e_px1, int inf_mask1) {
  __shared__ int32_t park_lds[72 * 256];
  VM_KERNEL_PROLOGUE();
  w.lds = park_lds;
  FixedLine l0, l1;
  l0.m = uni_ld2(entry0); l0.c = uni_ld2(entry0 + 2 * BN_NL); l0.xc = uni_ld2(entry0 + 4 * BN_NL);
  l1.m = uni_ld2(entry1); l1.c = uni_ld2(entry1 + 2 * BN_NL); l1.xc = uni_ld2(entry1 + 4 * BN_NL);
  vm_f12_mul_line_fixed2(w, e, l0, e_px0, (st & inf_mask0) != 0, l1, e_px1, (st & inf_mask1) != 0);
}
__global__ void __launch_bounds__(256, 2) k_miller_dbl_var(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int e_t, int e, int e_px) {
  VM_KERNEL_PROLOGUE(); vm_miller_dbl_var(w, e_t, e, e_px);
}
__global__ void __launch_bounds__(256, 2) k_miller_sqr_dbl_var(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int e_t, int e, int e_px) {
  __shared__ int32_t park_lds[72 * 256];
  VM_KERNEL_PROLOGUE();
  w.lds = park_lds;
  vm_miller_sqr_dbl_var(w, e_t, e, e_px);
}
__global__ void __launch_bounds__(256, 2) k_miller_add_var(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int e_t, int e_b, int which, int e, int e_px) {
  VM_KERNEL_PROLOGUE(); vm_miller_add_var(w, e_t, e_b, which, e, e_px);
}
__global__ void __launch_bounds__(256, 2) k_f12_mul(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int d, int a, int b, int conj_b) {
  __shared__ int32_t park_lds[72 * 256];
  VM_KERNEL_PROLOGUE();
  w.lds = park_lds;
  vm_f12_mul(w, d, a, b, conj_b != 0);
}
__global__ void __launch_bounds__(256, 2) k_f12_copy(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int d, int a) { VM_KERNEL_PROLOGUE(); vm_f12_copy(w, d, a); }
__global__ void __launch_bounds__(256, 2) k_f12_cyclo_sqr(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int d, int a) { VM_KERNEL_PROLOGUE(); vm_f12_cyclo_sqr(w, d, a); }
__global__ void __launch_bounds__(256, 2) k_f12_cyclo_sqr_n(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int d, int a, int count) {
  VM_KERNEL_PROLOGUE(); vm_f12_cyclo_sqr_n(w, d, a, __builtin_amdgcn_readfirstlane(count));
}
__global__ void __launch_bounds__(256, 2) k_f12_conj(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int d, int a) { VM_KERNEL_PROLOGUE(); vm_f12_conj(w, d, a); }
__global__ void __launch_bounds__(256, 2) k_f12_frob(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int d, int a, int j) { VM_KERNEL_PROLOGUE(); vm_f12_frob(w, d, a, j); }
__global__ void __launch_bounds__(256, 2) k_f12_inv(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int d, int a) { VM_KERNEL_PROLOGUE(); vm_f12_inv(w, d, a); }
__global__ void __launch_bounds__(256, 2) k_g16_compare(int32_t* ws, uint32_t n, uint8_t* __restrict__ status, const int32_t* __restrict__ target, int reject_code) {
  VM_KERNEL_PROLOGUE();
  bool acc = vm_f12_eq_const(w, VE_S0, target);
  if (i < n && (st & BN254_ST_PENDING)) status[i] = acc ? BN254_ST_ACCEPT : (uint8_t)reject_code;
}

// big-endian 32-byte field (8 dwords as loaded little-endian from memory) -> little-endian words
__device__ __forceinline__ void be_field_to_words(uint32_t w[8], const uint32_t* d) {
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = __builtin_bswap32(d[7 - i]);
}
__device__ __forceinline__ bool words_lt_p(const uint32_t w[8]) { return !words_ge(w, BN_P_WORDS); }

// =====================================================================================================================
// k_g16_prepare
// =====================================================================================================================
#define PREP_LDS_ROW 65  // 64 proof dwords + 1 pad: lane-per-proof reads hit 64 different banks
__global__ void __launch_bounds__(256, 2)
k_g16_prepare(const uint8_t* __restrict__ proofs, size_t stride, const uint8_t* __restrict__ inputs, int n_public, uint32_t n,
              int32_t* ws, uint8_t* __restrict__ status, const int32_t* __restrict__ msm_tab, const int32_t* __restrict__ k0,
              int inputs_match_key, int wide_msm) {
  __shared__ uint32_t lds[4 * 64 * PREP_LDS_ROW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t first = blockIdx.x * 256u + (uint32_t)wave * 64u;
  uint32_t* wl = lds + wave * 64 * PREP_LDS_ROW;
  const bool aligned = ((((uintptr_t)proofs) | stride) & 3) == 0;
  if (aligned) {
    // record j of this wave: one 256-byte contiguous segment per load instruction; SIXTEEN records per step, so that sixteen loads are in flight together (one
    // record at a time compiled to load / wait / store: 64 dependent round trips to HBM per wavefront)
    for (int j0 = 0; j0 < 64; j0 += 16) {
      uint32_t v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { const uint32_t rec = first + (uint32_t)(j0 + u); v[u] = rec < n ? *(const uint32_t*)(proofs + (size_t)rec * stride + (size_t)lane * 4) : 0u; }
#pragma unroll
      for (int u = 0; u < 16; u++) wl[(j0 + u) * PREP_LDS_ROW + lane] = v[u];
    }
  } else {
    for (int j = 0; j < 64; j++) {
      uint32_t rec = first + j;
      uint32_t v = 0;
      if (rec < n) {
        const uint8_t* p = proofs + (size_t)rec * stride + (size_t)lane * 4;
        v = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
      }
      wl[j * PREP_LDS_ROW + lane] = v;
    }
  }
  __syncthreads();
  const uint32_t i = first + lane;
  const bool live = i < n;
  const uint32_t ii = live ? i : n - 1;
  // dead lanes get an out-of-range lane offset: the descriptor's bounds check drops their stores
  DevWs w(ws, n, live ? i : DEAD_LANE);
  const uint32_t* my = wl + lane * PREP_LDS_ROW;
  uint32_t d[8], wx[8], wy[8];
  int err = 0;       // first error in the reference's order: A, then B (member, curve), B subgroup (next kernel), then C
  int err_c = 0;

  // ---- A
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = my[k];
  be_field_to_words(wx, d);
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = my[8 + k];
  be_field_to_words(wy, d);
  bool memb = words_lt_p(wx) & words_lt_p(wy);
  G1Aff A; A.x = fp_from_words(wx); A.y = fp_from_words(wy);
  if (!memb) err = BN254_ST_NOT_MEMBER; else if (!g1_on_curve(A)) err = BN254_ST_NOT_ON_CURVE;
  w.st(VE_AX, A.x); w.st(VE_AY, A.y);

  // ---- B : x.c1 | x.c0 | y.c1 | y.c0
  G2Aff B;
  bool membb = true;
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = my[16 + k];
  be_field_to_words(wx, d); membb &= words_lt_p(wx); B.x.c1 = fp_from_words(wx);
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = my[24 + k];
  be_field_to_words(wx, d); membb &= words_lt_p(wx); B.x.c0 = fp_from_words(wx);
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = my[32 + k];
  be_field_to_words(wx, d); membb &= words_lt_p(wx); B.y.c1 = fp_from_words(wx);
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = my[40 + k];
  be_field_to_words(wx, d); membb &= words_lt_p(wx); B.y.c0 = fp_from_words(wx);
  if (err == 0) { if (!membb) err = BN254_ST_NOT_MEMBER; else if (!g2_on_curve(B)) err = BN254_ST_NOT_ON_CURVE; }
  vst2(w, VE_B, B.x); vst2(w, VE_B + 2, B.y);

  // ---- C
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = my[48 + k];
  be_field_to_words(wx, d);
#pragma unroll
  for (int k = 0; k < 8; k++) d[k] = my[56 + k];
  be_field_to_words(wy, d);
  memb = words_lt_p(wx) & words_lt_p(wy);
  G1Aff C; C.x = fp_from_words(wx); C.y = fp_from_words(wy);
  if (!memb) err_c = BN254_ST_NOT_MEMBER; else if (!g1_on_curve(C)) err_c = BN254_ST_NOT_ON_CURVE;
  w.st(VE_CX, C.x); w.st(VE_CY, C.y);

  // ---- L = K0 + sum_i x_i K_i, x_i taken as raw 256-bit integers (no range check, as bn::Fr::from_slice)
  if (wide_msm) {
    // many public inputs: L comes from k_g16_msm_partial / k_g16_msm_reduce (they also set the identity flag)
    if (live) status[i] = err ? (uint8_t)err : (uint8_t)(BN254_ST_PENDING | err_c);
    return;
  }
  G1Aff K0; K0.x = uni_ld(k0); K0.y = uni_ld(k0 + BN_NL);
  G1Proj L = g1_from_affine(K0);
  if (inputs_match_key) {
    for (int s = 0; s < n_public; s++) {
      const uint8_t* sp = inputs + ((size_t)ii * (size_t)n_public + s) * 32;
      uint32_t sw[8];
      if ((((uintptr_t)inputs) & 3) == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) sw[k] = ((const uint32_t*)sp)[k];
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) sw[k] = (uint32_t)sp[4 * k] | (uint32_t)sp[4 * k + 1] << 8 | (uint32_t)sp[4 * k + 2] << 16 | (uint32_t)sp[4 * k + 3] << 24;
      }
      // sw[] holds the big-endian scalar as it lies in memory; kw[i] = bits 32 i .. 32 i + 31 of its value.  Window wi (MSM_FW_BITS = 13 bits, weight 2^(13 wi): bn254_fw.h --
      // 20 table additions per input where the byte windows of rounds 1-4 made 32) is consumed from the low end of kw[0] and the 256-bit array is shifted down by 13 each
      // time: no dynamic register indexing.  The 80-byte table entry of window wi + 1 (18 digits + 2 pad, 16-byte aligned: five 16-byte loads per lane) is in flight
      // while the addition of window wi runs.
      uint32_t kw[8];
#pragma unroll
      for (int k = 0; k < 8; k++) kw[k] = __builtin_bswap32(sw[7 - k]);
      auto next_digit = [&]() -> uint32_t {
        const uint32_t dg = kw[0] & MSM_FW_ENTRIES;
#pragma unroll
        for (int k = 0; k < 7; k++) kw[k] = (kw[k] >> MSM_FW_BITS) | (kw[k + 1] << (32 - MSM_FW_BITS));
        kw[7] >>= MSM_FW_BITS;
        return dg;
      };
      uint32_t dig = next_digit();
      G1Aff q = msm_entry(msm_tab, (size_t)(s * MSM_FW_WINDOWS) * MSM_FW_ENTRIES + (dig ? dig - 1 : 0));
      for (int j = 0; j < MSM_FW_WINDOWS; j++) {
        uint32_t dn = 0; G1Aff qn = q;
        if (j + 1 < MSM_FW_WINDOWS) { dn = next_digit(); qn = msm_entry(msm_tab, (size_t)(s * MSM_FW_WINDOWS + j + 1) * MSM_FW_ENTRIES + (dn ? dn - 1 : 0)); }
        if (dig != 0) L = g1_add_mixed(L, q);
        dig = dn; q = qn;
      }
    }
  }
  // to affine (one inversion per proof); the identity -- unreachable without a discrete-log relation between the K_i -- is kept
  // as (0, 1) plus a flag bit, and the Miller kernel makes its line value 1 (bn::pairing_batch skips such pairs)
  bool l_inf = g1_is_identity(L);
  G1Aff La = g1_to_affine(L);
  La.y = fp_select(l_inf, fp_one(), La.y);
  w.st(VE_LX, La.x); w.st(VE_LY, La.y);
  if (live) status[i] = err ? (uint8_t)err : (uint8_t)(BN254_ST_PENDING | (l_inf ? BN254_ST_LINF : 0) | err_c);
}

// BN254_FLAG_STRICT_SCALARS: a public input >= r makes the proof's status NOT_MEMBER, ahead of every other outcome (the
// reference's bn::Fr::from_slice does not range-check; this is the opt-in stricter policy).  Runs right after
// k_g16_prepare, so that such proofs are no longer pending for the rest of the pipeline.
__global__ void __launch_bounds__(256, 2)
k_g16_check_scalars(const uint8_t* __restrict__ inputs, int n_public, uint32_t n, uint8_t* __restrict__ status) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  bool bad = false;
  for (int s = 0; s < n_public; s++) {
    const uint8_t* sp = inputs + ((size_t)i * (size_t)n_public + s) * 32;
    uint32_t w[8];
    words_from_be(w, sp);
    bad |= words_ge(w, BN_R_WORDS);
  }
  if (bad) status[i] = BN254_ST_NOT_MEMBER;
}

// =====================================================================================================================
// public-input MSM for keys with many inputs (BASELINE config 5: 1024): the inputs of one proof are spread over `chunks` lanes
// =====================================================================================================================
// lane g = c * n + i: chunk c of proof i sums inputs [c * per, min((c+1) * per, n_public)); partial sums (projective, 27 dwords)
// go to part[(c * 27 + k) * n + i]: every access of a wave is contiguous over proofs.
__global__ void __launch_bounds__(256, 2)
k_g16_msm_partial(const uint8_t* __restrict__ inputs, int n_public, uint32_t n, int per, int chunks, const uint8_t* __restrict__ status,
                  const int32_t* __restrict__ msm_tab, int32_t* __restrict__ part) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= n * (uint32_t)chunks) return;
  const uint32_t c = g / n, i = g - c * n;
  G1Proj acc = g1_identity();
  if (status[i] & BN254_ST_PENDING) {
    const int s_end = (int)min((uint32_t)n_public, (c + 1) * (uint32_t)per);
    for (int s = (int)(c * per); s < s_end; s++) {
      const uint8_t* sp = inputs + ((size_t)i * (size_t)n_public + s) * 32;
      uint32_t sw[8];
      if ((((uintptr_t)inputs) & 3) == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) sw[k] = ((const uint32_t*)sp)[k];
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) sw[k] = (uint32_t)sp[4 * k] | (uint32_t)sp[4 * k + 1] << 8 | (uint32_t)sp[4 * k + 2] << 16 | (uint32_t)sp[4 * k + 3] << 24;
      }
      for (int j = 0; j < 32; j++) {  // byte j of the big-endian scalar = window 31 - j (as in k_g16_prepare)
        const int wi = 31 - j;
        uint32_t dig = sw[0] & 0xff;
#pragma unroll
        for (int k = 0; k < 7; k++) sw[k] = (sw[k] >> 8) | (sw[k + 1] << 24);
        sw[7] >>= 8;
        if (dig != 0) acc = g1_add_mixed(acc, msm_entry(msm_tab, (size_t)(s * 32 + wi) * 255 + (dig - 1)));
      }
    }
  }
  int32_t* o = part + (size_t)c * 27 * n + i;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { o[(size_t)l * n] = acc.x.v[l]; o[(size_t)(9 + l) * n] = acc.y.v[l]; o[(size_t)(18 + l) * n] = acc.z.v[l]; }
}
// The same sum from COMB tables (bn254_host.hpp::build_comb_table; keys prepared with msm_comb): the lane walks the G16_COMB_COLS = 20 columns
// from the top, doubling its accumulator once per column and adding, for each of its inputs, the entry selected by bits c, c + 20, ..., c + 240 of
// the scalar (G16_COMB_TEETH = 13 bits): 20 additions per input and 20 doublings per lane instead of 32 additions per input.
// k_g16_comb_digits first turns every scalar (big-endian bytes: bit b of the integer is bit b % 8 of byte 31 - b / 8) into its 20 column digits,
// stored transposed -- digits[(col * n_public + s) * n + i] -- so that the lanes of a wavefront (consecutive proofs) read consecutive u16.
__global__ void __launch_bounds__(256)
k_g16_comb_digits(const uint8_t* __restrict__ inputs, int n_public, uint32_t n, uint16_t* __restrict__ digits) {
  const size_t g = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (g >= (size_t)n * (size_t)n_public) return;
  const uint32_t s = (uint32_t)(g / n), i = (uint32_t)(g - (size_t)s * n);
  const uint8_t* sp = inputs + ((size_t)i * (size_t)n_public + s) * 32;
  uint32_t w[8];   // w[k]: bits 32 k .. 32 k + 31 of the integer
#pragma unroll
  for (int k = 0; k < 8; k++) { const uint8_t* q = sp + 28 - 4 * k; w[k] = (uint32_t)q[0] << 24 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 8 | (uint32_t)q[3]; }
  for (int col = 0; col < G16_COMB_COLS; col++) {
    const uint32_t idx = g16_comb_digit(w, col);
    digits[((size_t)col * (size_t)n_public + s) * n + i] = (uint16_t)idx;
  }
}
__global__ void __launch_bounds__(256, 2)
k_g16_msm_partial_comb(const uint16_t* __restrict__ digits, int n_public, uint32_t n, int per, int chunks, const uint8_t* __restrict__ status,
                       const int32_t* __restrict__ msm_tab, int32_t* __restrict__ part) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= n * (uint32_t)chunks) return;
  const uint32_t c = g / n, i = g - c * n;
  G1Proj acc = g1_identity();
  if (status[i] & BN254_ST_PENDING) {
    const int s_begin = (int)(c * per), s_end = (int)min((uint32_t)n_public, (c + 1) * (uint32_t)per);
    // Software pipeline (round 4): the table entry of addition k + 1 (a random 80-byte read of a 671 MB table) and the digit of addition k + 2 are in flight while
    // addition k computes.  Without it every addition waited for its own digit, then for its own entry, with two wavefronts per SIMD to hide that behind
    // (VALU-active 0.37, profiles/r02_cfg5_pmc_summary.csv).  (col, s) walks the columns from the top, the chunk's inputs inside a column.
    auto digit_at = [&](int col, int s) -> uint32_t { return digits[((size_t)col * (size_t)n_public + (size_t)s) * n + i]; };
    auto entry_at = [&](int s, uint32_t idx) -> G1Aff { return msm_entry(msm_tab, ((size_t)s << G16_COMB_TEETH) + idx); };   // idx 0: a valid (unused) slot of the table
    auto advance = [&](int& col, int& s) { if (++s == s_end) { s = s_begin; col--; } };
    if (s_begin < s_end) {
      int col1 = G16_COMB_COLS - 1, s1 = s_begin;            // position of addition k + 1 while addition k runs
      uint32_t d_cur = digit_at(col1, s1);
      G1Aff e_cur = entry_at(s1, d_cur);
      int col0 = col1, s0 = s1;
      advance(col1, s1);
      uint32_t d_nxt = col1 >= 0 ? digit_at(col1, s1) : 0u;
      int col2 = col1, s2 = s1;                               // position of addition k + 2
      if (col2 >= 0) advance(col2, s2);
      while (col0 >= 0) {
        G1Aff e_nxt = e_cur;
        uint32_t d_nn = 0u;
        if (col1 >= 0) e_nxt = entry_at(s1, d_nxt);
        if (col2 >= 0) d_nn = digit_at(col2, s2);
        if (s0 == s_begin && col0 != G16_COMB_COLS - 1) acc = g1_dbl(acc);
        if (d_cur != 0) acc = g1_add_mixed(acc, e_cur);
        col0 = col1; s0 = s1; d_cur = d_nxt; e_cur = e_nxt;
        col1 = col2; s1 = s2; d_nxt = d_nn;
        if (col2 >= 0) advance(col2, s2);
      }
    }
  }
  int32_t* o = part + (size_t)c * 27 * n + i;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { o[(size_t)l * n] = acc.x.v[l]; o[(size_t)(9 + l) * n] = acc.y.v[l]; o[(size_t)(18 + l) * n] = acc.z.v[l]; }
}
// L = K0 + sum of the chunk sums (complete projective additions), to affine, identity flag into the status byte
__global__ void __launch_bounds__(256, 2)
k_g16_msm_reduce(const int32_t* __restrict__ part, int chunks, uint32_t n, int32_t* ws, uint8_t* __restrict__ status, const int32_t* __restrict__ k0) {
  VM_KERNEL_PROLOGUE();
  const uint32_t ii = i < n ? i : n - 1;
  G1Aff K0; K0.x = uni_ld(k0); K0.y = uni_ld(k0 + BN_NL);
  G1Proj L = g1_from_affine(K0);
  for (int c = 0; c < chunks; c++) {
    const int32_t* o = part + (size_t)c * 27 * n + ii;
    G1Proj q;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { q.x.v[l] = o[(size_t)l * n]; q.y.v[l] = o[(size_t)(9 + l) * n]; q.z.v[l] = o[(size_t)(18 + l) * n]; }
    BN_SETB(q.x, 3.0, 0.5); BN_SETB(q.y, 3.0, 0.5); BN_SETB(q.z, 3.0, 0.5);
    L = g1_add(L, q);
  }
  bool l_inf = g1_is_identity(L);
  G1Aff La = g1_to_affine(L);
  La.y = fp_select(l_inf, fp_one(), La.y);
  w.st(VE_LX, La.x); w.st(VE_LY, La.y);
  if (i < n && (st & BN254_ST_PENDING) && l_inf) status[i] = st | BN254_ST_LINF;
}

// The same with EIGHT lanes per proof (small batches, where the launch lasts as long as one lane's chain of `chunks` additions): lane `sub` adds the
// chunk sums sub, sub + 8, ..., the eight partial sums meet in LDS, lane 0 adds them to K0 and finishes: chunks / 8 + 8 additions in a row
// instead of chunks.  256 threads = 32 proofs.
__global__ void __launch_bounds__(256, 2)
k_g16_msm_reduce8(const int32_t* __restrict__ part, int chunks, uint32_t n, int32_t* ws, uint8_t* __restrict__ status, const int32_t* __restrict__ k0) {
  __shared__ int32_t sums[32 * 8 * 27];
  const uint32_t p = blockIdx.x * 32u + (threadIdx.x >> 3), sub = threadIdx.x & 7u;
  const uint32_t pp = p < n ? p : n - 1;
  G1Proj L = g1_identity();
  for (int c = (int)sub; c < chunks; c += 8) {
    const int32_t* o = part + (size_t)c * 27 * n + pp;
    G1Proj q;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { q.x.v[l] = o[(size_t)l * n]; q.y.v[l] = o[(size_t)(9 + l) * n]; q.z.v[l] = o[(size_t)(18 + l) * n]; }
    BN_SETB(q.x, 3.0, 0.5); BN_SETB(q.y, 3.0, 0.5); BN_SETB(q.z, 3.0, 0.5);
    L = g1_add(L, q);
  }
  {
    int32_t* o = sums + threadIdx.x * 27;
    const Fp x = fp_reduce(L.x), y = fp_reduce(L.y), z = fp_reduce(L.z);
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { o[l] = x.v[l]; o[9 + l] = y.v[l]; o[18 + l] = z.v[l]; }
  }
  __syncthreads();
  const bool lead = sub == 0 && p < n;
  const uint8_t st = status[pp];
  G1Aff K0; K0.x = uni_ld(k0); K0.y = uni_ld(k0 + BN_NL);
  G1Proj T = g1_from_affine(K0);
  for (int j = 0; j < 8; j++) {       // every lane runs the tail (no divergence); only the leading lane's result is stored
    const int32_t* o = sums + ((threadIdx.x & ~7u) + j) * 27;
    G1Proj q;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { q.x.v[l] = o[l]; q.y.v[l] = o[9 + l]; q.z.v[l] = o[18 + l]; }
    BN_SETB(q.x, 1.01, 0.5); BN_SETB(q.y, 1.01, 0.5); BN_SETB(q.z, 1.01, 0.5);
    T = g1_add(T, q);
  }
  const bool l_inf = g1_is_identity(T);
  G1Aff La = g1_to_affine(T);
  La.y = fp_select(l_inf, fp_one(), La.y);
  DevWs w(ws, n, lead ? p : DEAD_LANE);
  w.st(VE_LX, La.x); w.st(VE_LY, La.y);
  if (lead && (st & BN254_ST_PENDING) && l_inf) status[p] = st | BN254_ST_LINF;
}

// =====================================================================================================================
// k_g16_subgroup
// =====================================================================================================================
__global__ void __launch_bounds__(256, 2)
k_g16_subgroup(uint32_t n, int32_t* ws, uint8_t* __restrict__ status, int inputs_match_key, int e_t) {
  // runs AFTER the Miller loop: the r-torsion test of B reads the loop's final G2 point (bn254_vm.h::vm_g2_ate_check)
  VM_KERNEL_PROLOGUE();
  bool ok = vm_g2_ate_check(w, e_t, VE_B);
  if (i < n && (st & BN254_ST_PENDING)) {
    uint8_t out;
    if (!ok) out = BN254_ST_NOT_IN_SUBGROUP;
    else if (st & 0x3f) out = st & 0x3f;                      // deferred error of C
    else if (!inputs_match_key) out = BN254_ST_INPUT_LEN;     // PrepareInputsFailed comes after every loader error
    else out = BN254_ST_PENDING | (st & BN254_ST_LINF);
    status[i] = out;
  }
}
__global__ void __launch_bounds__(256, 2) k_dbg_g2_ate(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, uint8_t* o) {
  VM_KERNEL_PROLOGUE();
  bool ok = vm_g2_ate_check(w, VE_T, VE_B);
  if (i < n) o[i] = ok ? 1 : 0;
}

// =====================================================================================================================
// random-linear-combination batch mode (bn254_rlc.h)
// =====================================================================================================================
// accessor of the fold: element ids >= RLC_HI address the partner lane's column of the same workspace
struct DevWs2 {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t row_bytes, voff_lo, voff_hi;
  int32_t* lds;
  __device__ __forceinline__ DevWs2(int32_t* base, uint32_t n, uint32_t lane_lo, uint32_t lane_hi, int32_t* lds_) : lds(lds_) {
    DevWs t(base, n, lane_lo);
    rsrc = t.rsrc; row_bytes = t.row_bytes; voff_lo = lane_lo * 4u; voff_hi = lane_hi * 4u;
  }
  __device__ __forceinline__ void park(int slot, const Fp2& a) const {
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { lds[(slot * 18 + l) * 256 + threadIdx.x] = a.c0.v[l]; lds[(slot * 18 + BN_NL + l) * 256 + threadIdx.x] = a.c1.v[l]; }
  }
  __device__ __forceinline__ Fp2 unpark(int slot) const {
    Fp2 r;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { r.c0.v[l] = lds[(slot * 18 + l) * 256 + threadIdx.x]; r.c1.v[l] = lds[(slot * 18 + BN_NL + l) * 256 + threadIdx.x]; }
    return r;
  }
  __device__ __forceinline__ Fp ld(int e) const {
    Fp r;
    uint32_t eu = __builtin_amdgcn_readfirstlane((uint32_t)e);
    const bool hi = eu >= (uint32_t)RLC_HI;   // wave-uniform
    if (hi) eu -= (uint32_t)RLC_HI;
    const uint32_t vo = hi ? voff_hi : voff_lo;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) r.v[l] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, (eu * (uint32_t)BN_NL + (uint32_t)l) * row_bytes, 0);
    return r;
  }
  __device__ __forceinline__ void st(int e, const Fp& a) const {
    uint32_t eu = __builtin_amdgcn_readfirstlane((uint32_t)e);
#pragma unroll
    for (int l = 0; l < BN_NL; l++) __builtin_amdgcn_raw_buffer_store_b32(a.v[l], rsrc, voff_lo, (eu * (uint32_t)BN_NL + (uint32_t)l) * row_bytes, 0);
  }
};
// accessor of the shared-accumulator Miller loop: the lane's own column (f) and the column of the current proof (sel(q): lane + q * m)
struct DevWsM {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t row_bytes, voff0, lane, m, n, voff;
  int32_t* lds;
  __device__ __forceinline__ DevWsM(int32_t* base, uint32_t n_, uint32_t lane_, uint32_t m_, int32_t* lds_) : lane(lane_), m(m_), n(n_), lds(lds_) {
    DevWs t(base, n_, lane_);
    rsrc = t.rsrc; row_bytes = t.row_bytes; voff0 = lane_ < n_ ? lane_ * 4u : 0xfffffffcu; voff = voff0;
  }
  __device__ __forceinline__ void sel(int q) { const uint32_t i = lane + (uint32_t)q * m; voff = (lane < m && i < n) ? i * 4u : 0xfffffffcu; }
  __device__ __forceinline__ void park(int slot, const Fp2& a) const {
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { lds[(slot * 18 + l) * 256 + threadIdx.x] = a.c0.v[l]; lds[(slot * 18 + BN_NL + l) * 256 + threadIdx.x] = a.c1.v[l]; }
  }
  __device__ __forceinline__ Fp2 unpark(int slot) const {
    Fp2 r;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { r.c0.v[l] = lds[(slot * 18 + l) * 256 + threadIdx.x]; r.c1.v[l] = lds[(slot * 18 + BN_NL + l) * 256 + threadIdx.x]; }
    return r;
  }
  __device__ __forceinline__ Fp ldv(int e, uint32_t vo) const {
    Fp r;
    uint32_t eu = __builtin_amdgcn_readfirstlane((uint32_t)e);
#pragma unroll
    for (int l = 0; l < BN_NL; l++) r.v[l] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, (eu * (uint32_t)BN_NL + (uint32_t)l) * row_bytes, 0);
    return r;
  }
  __device__ __forceinline__ void stv(int e, const Fp& a, uint32_t vo) const {
    uint32_t eu = __builtin_amdgcn_readfirstlane((uint32_t)e);
#pragma unroll
    for (int l = 0; l < BN_NL; l++) __builtin_amdgcn_raw_buffer_store_b32(a.v[l], rsrc, vo, (eu * (uint32_t)BN_NL + (uint32_t)l) * row_bytes, 0);
  }
  __device__ __forceinline__ Fp ld(int e) const { return ldv(e, voff); }
  __device__ __forceinline__ void st(int e, const Fp& a) const { stv(e, a, voff); }
  __device__ __forceinline__ Fp ld0(int e) const { return ldv(e, voff0); }
  __device__ __forceinline__ void st0(int e, const Fp& a) const { stv(e, a, voff0); }
};
// one Miller step of the G proofs of a lane with a shared accumulator (bn254_rlc.h::vm_miller_var_multi); lanes [0, m)
template <bool DO_SQR>
__global__ void __launch_bounds__(256, 2)
k_rlc_miller_multi(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int kind, int G, uint32_t m) {
  __shared__ int32_t park_lds[72 * 256];
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  uint32_t deadmask = 0;
  const int g = __builtin_amdgcn_readfirstlane(G);
  for (int q = 0; q < g; q++) {
    const uint32_t i = j + (uint32_t)q * m;
    const bool live = j < m && i < n && (status[i < n ? i : n - 1] & BN254_ST_PENDING) != 0;
    if (!live) deadmask |= 1u << q;
  }
  if (__builtin_amdgcn_ballot_w64(j < m) == 0) return;
  DevWsM w(ws, n, j < m ? j : 0xffffffffu, m, park_lds);
  vm_miller_var_multi<DO_SQR>(w, __builtin_amdgcn_readfirstlane(kind), g, deadmask);
}
// f = 1 on the lanes that carry an accumulator; T = B on every pending proof (k_vm_init does both for the one-proof-per-lane layout)
__global__ void __launch_bounds__(256, 2) k_rlc_init_f(int32_t* ws, uint32_t n, uint32_t m) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  DevWs w(ws, n, j < m ? j : DEAD_LANE);
  w.st(VE_F, fp_one());
  for (int e = 1; e < 12; e++) w.st(VE_F + e, fp_zero());
}
// per pending proof: weight r_i = ChaCha20(key, counter_base + i), A <- r A, C' <- r C, t_j = r x_j (bn254_rlc.h::vm_rlc_scale)
__global__ void __launch_bounds__(256, 2)
k_rlc_scale(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, const uint8_t* __restrict__ inputs, int n_public, ChaChaKey key, uint32_t counter_base) {
  VM_KERNEL_PROLOGUE();
  const uint32_t ii = i < n ? i : n - 1;
  uint32_t r[4];
  chacha20_block4(r, key, counter_base + ii);
  const uint8_t* in = inputs + (size_t)ii * (size_t)n_public * 32;
  vm_rlc_scale(w, r, n_public, [&](int j, uint32_t* o) { words_from_be(o, in + 32 * j); });
}
// lanes that are no longer pending contribute the neutral element to their group
__global__ void __launch_bounds__(256, 2)
k_rlc_neutral(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int n_public, int with_f) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const bool dead = i < n && (status[i] & BN254_ST_PENDING) == 0;
  if (__builtin_amdgcn_ballot_w64(dead) == 0) return;
  DevWs w(ws, n, dead ? i : DEAD_LANE);
  vm_rlc_neutral(w, n_public, __builtin_amdgcn_readfirstlane(with_f) != 0);
}
// one fold round: lane j < cur - half takes lane j + half (bn254_rlc.h::vm_rlc_fold)
__global__ void __launch_bounds__(256, 2)
k_rlc_fold(int32_t* ws, uint32_t n, uint32_t cur, uint32_t half, int n_public, int with_f) {
  __shared__ int32_t park_lds[72 * 256];
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  const bool act = j + half < cur;
  if (__builtin_amdgcn_ballot_w64(act) == 0) return;
  DevWs2 w(ws, n, act ? j : DEAD_LANE, act ? j + half : DEAD_LANE, park_lds);
  vm_rlc_fold(w, n_public, __builtin_amdgcn_readfirstlane(with_f) != 0);
}
// group stage, one lane per group: the G1 arguments of the three table-driven pairs (bn254_rlc.h::vm_rlc_group_points)
__global__ void __launch_bounds__(256, 2)
k_rlc_group_points(int32_t* ws, uint32_t n, uint8_t* __restrict__ grp_status, uint32_t groups, int n_public, const int32_t* __restrict__ rlc_tab,
                   const int32_t* __restrict__ msm_tab) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (__builtin_amdgcn_ballot_w64(g < groups) == 0) return;
  DevWs w(ws, n, g < groups ? g : DEAD_LANE);
  const int fl = vm_rlc_group_points(w, n_public, [&](int b, int wi, int d) {
    return b < 2 ? msm_entry(rlc_tab, (size_t)(b * MSM_FW_WINDOWS + wi) * MSM_FW_ENTRIES + d) : msm_entry(msm_tab, (size_t)((b - 2) * MSM_FW_WINDOWS + wi) * MSM_FW_ENTRIES + d);
  });
  if (g < groups) grp_status[g] = (uint8_t)(BN254_ST_PENDING | ((fl & 1) ? BN254_ST_LINF : 0) | ((fl & 2) ? BN254_ST_LINF2 : 0) | ((fl & 4) ? BN254_ST_LINF3 : 0));
}
// group stage: one Miller step of the three table-driven pairs (bn254_rlc.h::vm_miller_step_fixed3)
template <bool DO_SQR>
__global__ void __launch_bounds__(256, 2)
k_rlc_group_step(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int e, const int32_t* __restrict__ entry0, const int32_t* __restrict__ entry1,
                 const int32_t* __restrict__ entry2) {
  __shared__ int32_t park_lds[72 * 256];
  VM_KERNEL_PROLOGUE();
  w.lds = park_lds;
  FixedLine l0, l1, l2;
  l0.m = uni_ld2(entry0); l0.c = uni_ld2(entry0 + 2 * BN_NL); l0.xc = uni_ld2(entry0 + 4 * BN_NL);
  l1.m = uni_ld2(entry1); l1.c = uni_ld2(entry1 + 2 * BN_NL); l1.xc = uni_ld2(entry1 + 4 * BN_NL);
  l2.m = uni_ld2(entry2); l2.c = uni_ld2(entry2 + 2 * BN_NL); l2.xc = uni_ld2(entry2 + 4 * BN_NL);
  vm_miller_step_fixed3<DO_SQR>(w, e, l0, VE_LX, (st & BN254_ST_LINF) != 0, l1, VE_CX, (st & BN254_ST_LINF2) != 0, l2, VE_AX, (st & BN254_ST_LINF3) != 0);
}
// every pending proof takes its group's verdict: ACCEPT, or it stays pending (0x80) for the exact path
__global__ void __launch_bounds__(256, 2)
k_rlc_scatter(uint8_t* __restrict__ status, uint32_t n, const uint8_t* __restrict__ grp_status, RlcPlan plan) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint8_t st = status[i];
  if (!(st & BN254_ST_PENDING)) return;
  status[i] = grp_status[rlc_group_of(i, plan)] == BN254_ST_ACCEPT ? (uint8_t)BN254_ST_ACCEPT : (uint8_t)BN254_ST_PENDING;
}
// fallback plumbing: rows idx[k] of a strided byte array -> packed rows; statuses back
__global__ void k_gather_rows(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t src_stride, uint32_t row_bytes, const uint32_t* __restrict__ idx, uint32_t m) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t per = row_bytes / 4;
  if (t >= (size_t)m * per) return;
  const uint32_t k = (uint32_t)(t / per), c = (uint32_t)(t % per);
  const uint8_t* p = src + (size_t)idx[k] * src_stride + 4 * (size_t)c;
  uint32_t v;
  if ((((uintptr_t)src) | src_stride) & 3) v = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
  else v = *(const uint32_t*)p;
  ((uint32_t*)dst)[t] = v;
}
__global__ void k_scatter_status(uint8_t* __restrict__ status, const uint8_t* __restrict__ fb_status, const uint32_t* __restrict__ idx, uint32_t m) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k < m) status[idx[k]] = fb_status[k];
}

// =====================================================================================================================
// the issue rate of the instruction every field product is made of, measured on THIS device: sixteen independent v_mad_u64_u32 chains per lane, two
// wavefronts per SIMD (tools/ubench_valu.hip: 1.92 ns per wavefront-instruction and SIMD on the box of profiles/r01_ubench_valu.txt = 35.1 T lane-mads/s;
// boxes of one pool differ by a few percent, so bench.py prices its rooflines with the rate of the box it runs on)
// =====================================================================================================================
#define VALU_PEAK_ITERS 16384
__global__ void __launch_bounds__(256) k_valu_peak(uint32_t* out, uint32_t seed, int iters) {
  const uint32_t tid = threadIdx.x + blockIdx.x * blockDim.x;
  const uint32_t a = seed * 2654435761u + tid, b = (seed ^ 0x9e3779b9u) + tid * 7u;
  uint64_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = (uint64_t)(a + i) << 7 | (uint32_t)i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
  }
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) x ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32);
  out[tid] = x;
}

// =====================================================================================================================
// probes for the GPU parity tests
// =====================================================================================================================
__device__ __forceinline__ Fp probe_ld_fp(const uint8_t* p) {
  uint32_t w[8];
  words_from_be(w, p);
  return fp_from_words(w);
}
__device__ __forceinline__ void probe_st_fp(uint8_t* p, const Fp& a) {
  uint32_t w[8];
  fp_to_words(w, a);
  words_to_be(p, w);
}
__global__ void __launch_bounds__(256, 2) k_dbg_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  probe_st_fp(o + 32 * i, fp_mul(probe_ld_fp(a + 32 * i), probe_ld_fp(b + 32 * i)));
}
// Fp12 / pairing probes: bytes (tower order) <-> workspace (k order); the operations themselves are the product's VM kernels
__global__ void __launch_bounds__(256, 2) k_dbg_load(int32_t* ws, uint32_t n, uint8_t* status, int e, const uint8_t* src, int kind) {
  // kind 0: Fp12 (384 B, tower order) -> element e; kind 1: G1 (64 B) -> VE_AX; kind 2: G2 (128 B, gnark order) -> VE_B; also marks the lane pending
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  DevWs w(ws, n, i);
  if (kind == 0) {
    const int korder[6] = {0, 2, 4, 1, 3, 5};
    const uint8_t* p = src + 384 * (size_t)i;
    for (int t = 0; t < 6; t++) { w.st(e + 2 * korder[t], probe_ld_fp(p + 64 * t)); w.st(e + 2 * korder[t] + 1, probe_ld_fp(p + 64 * t + 32)); }
  } else if (kind == 1) {
    w.st(VE_AX, probe_ld_fp(src + 64 * (size_t)i)); w.st(VE_AY, probe_ld_fp(src + 64 * (size_t)i + 32));
  } else {
    const uint8_t* q = src + 128 * (size_t)i;
    w.st(VE_B + 1, probe_ld_fp(q)); w.st(VE_B, probe_ld_fp(q + 32)); w.st(VE_B + 3, probe_ld_fp(q + 64)); w.st(VE_B + 2, probe_ld_fp(q + 96));
  }
  status[i] = BN254_ST_PENDING;
}
__global__ void __launch_bounds__(256, 2) k_dbg_store(int32_t* ws, uint32_t n, int e, uint8_t* dst) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  DevWs w(ws, n, i);
  const int korder[6] = {0, 2, 4, 1, 3, 5};
  uint8_t* p = dst + 384 * (size_t)i;
  for (int t = 0; t < 6; t++) { probe_st_fp(p + 64 * t, w.ld(e + 2 * korder[t])); probe_st_fp(p + 64 * t + 32, w.ld(e + 2 * korder[t] + 1)); }
}
__global__ void __launch_bounds__(256, 2) k_dbg_g2_subgroup(const uint8_t* g2, uint8_t* o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  G2Aff q; q.x.c1 = probe_ld_fp(g2 + 128 * i); q.x.c0 = probe_ld_fp(g2 + 128 * i + 32);
  q.y.c1 = probe_ld_fp(g2 + 128 * i + 64); q.y.c0 = probe_ld_fp(g2 + 128 * i + 96);
  o[i] = (g2_on_curve(q) && g2_in_subgroup(q)) ? 1 : 0;
}

}  // namespace bn254

// ---- launch wrappers (C++ linkage, declared in bn254_kernels.h) -----------------------------------------------------------
using namespace bn254;
static inline unsigned grid_for(size_t n) { return (unsigned)((n + 255) / 256); }

const char* const bn254_kernel_kind_names[KID_COUNT] = {
  "k_g16_prepare", "k_g16_subgroup", "k_vm_init", "k_f12_sqr", "k_f12_mul_line_fixed",
  "k_f12_mul", "k_f12_cyclo_sqr", "k_f12_conj", "k_f12_frob", "k_f12_inv", "k_g16_compare", "k_f12_copy", "k_f12_cyclo_sqr_n", "k_miller_dbl_var", "k_miller_add_var", "k_g16_msm_partial", "k_g16_msm_reduce", "k_f12_mul_line_fixed2", "k_miller_sqr_dbl_var", "k_miller_step_dbl", "k_miller_step_add", "k_coop12_miller_g16", "k_miller_run"};
struct ProfScope {  // records the event pair around one launch (no-op without a profile or for unselected kinds)
  G16Prof* p; hipStream_t s; int slot;
  ProfScope(G16Prof* p_, int kid, hipStream_t s_) : p(p_), s(s_), slot(-1) {
    if (p && ((p->mask >> kid) & 1u) && p->used < p->cap) { slot = p->used++; p->kid[slot] = (uint8_t)kid; (void)hipEventRecord(p->ev[2 * slot], s); }
  }
  ~ProfScope() { if (slot >= 0) (void)hipEventRecord(p->ev[2 * slot + 1], s); }
};
#define BN_LAUNCH(KID, KERNEL, ...) do { ProfScope ps_(prof, KID, s); hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(256), 0, s, __VA_ARGS__); } while (0)
// host-side OPS for the VM programs: every operation is one kernel launch on the stream
struct LaunchOps {
  int32_t* ws; uint32_t n; const uint8_t* status; unsigned grid; hipStream_t s;
  const int32_t* tab[3];
  G16Prof* prof;
  int inf_mask[3] = {BN254_ST_LINF, 0, 0};   // status bits marking the G1 point of fixed pair 0 / 1 / 2 as the identity
  int uni(int x) { return x; }
  void f12_sqr(int e) { BN_LAUNCH(KID_F12_SQR, k_f12_sqr, ws, n, status, e); }
  void miller_dbl_var(int et, int e, int ep) { BN_LAUNCH(KID_MILLER_DBL_VAR, k_miller_dbl_var, ws, n, status, et, e, ep); }
  void miller_step(bool do_sqr, int kind, int st_, int et, int eb, int e, int epa, int ep0, int ep1) {
    const int32_t *t0 = tab[0] + (size_t)st_ * FIXED_LINE_DWORDS, *t1 = tab[1] + (size_t)st_ * FIXED_LINE_DWORDS;
    ProfScope ps_(prof, kind == 0 ? KID_MILLER_STEP_DBL : KID_MILLER_STEP_ADD, s);
    bn254_launch_miller_step(do_sqr, kind, ws, n, status, grid, s, et, eb, e, epa, t0, ep0, inf_mask[0], t1, ep1, inf_mask[1]);
  }
  void miller_run(int s_begin, int s_end, int et, int eb, int e, int epa, int ep0, int ep1) {
    static const MillerKinds kinds = [] { MillerKinds k; memset(&k, 0, sizeof k); for (int st_ = 0; st_ < BN_ATE_STEPS; st_++) k.nib[st_ >> 1] |= (uint8_t)(miller_step_kind(st_) << ((st_ & 1) * 4)); return k; }();
    ProfScope ps_(prof, KID_MILLER_RUN, s);
    bn254_launch_miller_run(kinds, s_begin, s_end, ws, n, status, grid, s, et, eb, e, epa, tab[0], ep0, inf_mask[0], tab[1], ep1, inf_mask[1]);
  }
  void miller_run_fixed2(int s_begin, int s_end, int e, int ep0, int ep1) {
    static const MillerKinds kinds = [] { MillerKinds k; memset(&k, 0, sizeof k); for (int st_ = 0; st_ < BN_ATE_STEPS; st_++) k.nib[st_ >> 1] |= (uint8_t)(miller_step_kind(st_) << ((st_ & 1) * 4)); return k; }();
    ProfScope ps_(prof, KID_MILLER_RUN, s);
    bn254_launch_miller_run_fixed2(kinds, s_begin, s_end, ws, n, status, grid, s, e, tab[0], ep0, inf_mask[0], tab[1], ep1, inf_mask[1]);
  }
  void miller_sqr_dbl_var(int et, int e, int ep) { BN_LAUNCH(KID_MILLER_SQR_DBL_VAR, k_miller_sqr_dbl_var, ws, n, status, et, e, ep); }
  void miller_add_var(int et, int eb, int which, int e, int ep) { BN_LAUNCH(KID_MILLER_ADD_VAR, k_miller_add_var, ws, n, status, et, eb, which, e, ep); }
  void f12_mul_line_fixed(int e, int t, int st_, int ep) {
    BN_LAUNCH(KID_MUL_LINE_FIXED, k_f12_mul_line_fixed, ws, n, status, e, tab[t] + (size_t)st_ * FIXED_LINE_DWORDS, ep, inf_mask[t]);
  }
  void f12_mul_line_fixed2(int e, int st_, int ep0, int ep1) {
    BN_LAUNCH(KID_MUL_LINE_FIXED2, k_f12_mul_line_fixed2, ws, n, status, e, tab[0] + (size_t)st_ * FIXED_LINE_DWORDS, ep0, inf_mask[0],
              tab[1] + (size_t)st_ * FIXED_LINE_DWORDS, ep1, inf_mask[1]);
  }
  void f12_mul(int d, int a, int b, bool conj_b = false) { BN_LAUNCH(KID_F12_MUL, k_f12_mul, ws, n, status, d, a, b, conj_b ? 1 : 0); }
  void f12_copy(int d, int a) { BN_LAUNCH(KID_F12_COPY, k_f12_copy, ws, n, status, d, a); }
  void f12_cyclo_sqr(int d, int a) { BN_LAUNCH(KID_CYCLO_SQR, k_f12_cyclo_sqr, ws, n, status, d, a); }
  void f12_cyclo_sqr_n(int d, int a, int count) { BN_LAUNCH(KID_CYCLO_SQR_N, k_f12_cyclo_sqr_n, ws, n, status, d, a, count); }
  void f12_conj(int d, int a) { BN_LAUNCH(KID_F12_CONJ, k_f12_conj, ws, n, status, d, a); }
  void f12_frob(int d, int a, int j) { BN_LAUNCH(KID_F12_FROB, k_f12_frob, ws, n, status, d, a, j); }
  void f12_inv(int d, int a) { BN_LAUNCH(KID_F12_INV, k_f12_inv, ws, n, status, d, a); }
};
static uint8_t g_step_kinds[BN_ATE_STEPS];
static const uint8_t* step_kinds_host() {
  static bool init = false;
  if (!init) { for (int s_ = 0; s_ < BN_ATE_STEPS; s_++) g_step_kinds[s_] = (uint8_t)miller_step_kind(s_); init = true; }
  return g_step_kinds;
}

hipError_t bn254_launch_g16(const G16LaunchArgs& a, hipStream_t s, hipEvent_t* ev /* 5 events or nullptr */, G16Prof* prof) {
  unsigned grid = grid_for(a.n);
  uint32_t n = (uint32_t)a.n;
  if (ev) (void)hipEventRecord(ev[0], s);
  static const bool coop_on = [] { const char* e = getenv("BN254_COOP"); return !e || atoi(e) != 0; }();
  // BN254_MILLER_RUN_STEPS overrides the steps per k_miller_run launch (0: the one-launch-per-step kernels k_miller_step_dbl / _add)
  static const int run_steps_env = [] { const char* e = getenv("BN254_MILLER_RUN_STEPS"); int v = e ? atoi(e) : -1; return v < -1 ? -1 : v; }();
  // the form of this launch -- cooperative kernels (small batches: the public-input MSM moves into the cooperative kernel, twelve lanes per proof, L kept
  // projective, so k_g16_prepare stops after C; keys with many inputs keep their wide MSM kernels and hand L over through the workspace), lane kernels, or
  // their latency mode -- is a pure function of the sizes (bn254_g16_plan.h), shared with the plan probe
  const G16Form form = g16_launch_form(a.n, (size_t)a.n_public, a.inputs_match_key != 0, a.msm_part != nullptr, a.part_of_larger != 0,
                                       a.split_streams[0] && a.split_streams[1], coop_on, run_steps_env);
  const bool wide = form.wide, coop = form.form == G16_FORM_COOP;
  BN_LAUNCH(KID_PREPARE, k_g16_prepare, a.proofs, a.stride, a.inputs, a.n_public, n, a.ws, a.status, a.msm_tab, a.k0, a.inputs_match_key, (wide || coop) ? 1 : 0);
  if (a.strict_scalars && a.n_public > 0) hipLaunchKernelGGL(k_g16_check_scalars, dim3(grid), dim3(256), 0, s, a.inputs, a.n_public, n, a.status);
  if (wide) {
    // inputs of one proof spread over `chunks` lanes, proofs in slices that fit the partial-sum buffer
    // BN254_WIDE_PER (experiments): inputs per lane, at least G16_WIDE_MSM_INPUTS_PER_LANE (the partial-sum buffer is sized for that many chunks)
    static const int per_env = [] { const char* e = getenv("BN254_WIDE_PER"); int v = e ? atoi(e) : 0; return v >= G16_WIDE_MSM_INPUTS_PER_LANE && v <= 256 ? v : 0; }();
    const int per = per_env ? per_env : G16_WIDE_MSM_INPUTS_PER_LANE, chunks = (a.n_public + per - 1) / per;
    unsigned pg = (unsigned)(((size_t)n * chunks + 255) / 256);
    {
      ProfScope ps_(prof, KID_MSM_PARTIAL, s);
      if (a.msm_comb) {
        const size_t scalars = (size_t)n * (size_t)a.n_public;
        hipLaunchKernelGGL(k_g16_comb_digits, dim3((unsigned)((scalars + 255) / 256)), dim3(256), 0, s, a.inputs, a.n_public, n, a.msm_digits);
        hipLaunchKernelGGL(k_g16_msm_partial_comb, dim3(pg), dim3(256), 0, s, (const uint16_t*)a.msm_digits, a.n_public, n, per, chunks, (const uint8_t*)a.status, a.msm_tab, a.msm_part);
      }
      else hipLaunchKernelGGL(k_g16_msm_partial, dim3(pg), dim3(256), 0, s, a.inputs, a.n_public, n, per, chunks, (const uint8_t*)a.status, a.msm_tab, a.msm_part);
    }
    if (chunks >= 16 && a.n <= 65536) {
      // eight lanes per proof: below one wavefront per SIMD the reduction is one lane's chain of additions
      ProfScope ps_(prof, KID_MSM_REDUCE, s);
      hipLaunchKernelGGL(k_g16_msm_reduce8, dim3((unsigned)((a.n + 31) / 32)), dim3(256), 0, s, (const int32_t*)a.msm_part, chunks, n, a.ws, a.status, a.k0);
    } else {
      BN_LAUNCH(KID_MSM_REDUCE, k_g16_msm_reduce, (const int32_t*)a.msm_part, chunks, n, a.ws, a.status, a.k0);
    }
  }
  if (ev) (void)hipEventRecord(ev[1], s);
  if (coop) {
    // small batch: cooperative layout (bn254_coop12.hip): public-input MSM, Miller loop of the three pairs, final exponentiation and the verdict in ONE launch
    hipError_t e;
    { ProfScope ps_(prof, KID_COOP_G16, s); e = bn254_coop12_miller_g16(a.ws, a.status, a.n, a.gtab, a.dtab, a.inputs, a.n_public, a.inputs_match_key, a.msm_tab, a.k0, wide ? 1 : 0, 1, a.target, s); }
    if (e != hipSuccess) return e;
    // the r-torsion test of B, the status precedence and the comparison with e(alpha, beta) are the tail of the same launch
    if (ev) { (void)hipEventRecord(ev[2], s); (void)hipEventRecord(ev[3], s); (void)hipEventRecord(ev[4], s); }
    return hipGetLastError();
  }
  LaunchOps ops{a.ws, n, a.status, grid, s, {a.gtab, a.dtab, nullptr}, prof};
  BN_LAUNCH(KID_VM_INIT, k_vm_init, a.ws, n, (const uint8_t*)a.status);
  if (form.form == G16_FORM_LATENCY) {
    // latency mode: Miller(A, B) on the launch stream, the two table-driven pairs as their own chains (accumulators in the free
    // slots VE_S1 / VE_S2) on two more streams; f = f_A f_B f_C afterwards.  Three times the squarings, 40 % less time at 4096.
    ops.f12_copy(VE_S1, VE_F); ops.f12_copy(VE_S2, VE_F);
    (void)hipEventRecord(a.split_ev[0], s);
    LaunchOps ob{a.ws, n, a.status, grid, a.split_streams[0], {a.gtab, a.dtab, nullptr}, nullptr};
    LaunchOps oc{a.ws, n, a.status, grid, a.split_streams[1], {a.gtab, a.dtab, nullptr}, nullptr};
    (void)hipStreamWaitEvent(ob.s, a.split_ev[0], 0); (void)hipStreamWaitEvent(oc.s, a.split_ev[0], 0);
    const uint8_t* kinds = step_kinds_host();
    for (int st_ = 0; st_ < BN_ATE_STEPS; st_++) {
      const int kind = kinds[st_];
      if (kind == 0) { if (st_ != 0) ops.miller_sqr_dbl_var(VE_T, VE_F, VE_AX); else ops.miller_dbl_var(VE_T, VE_F, VE_AX); }
      else ops.miller_add_var(VE_T, VE_B, kind - 1, VE_F, VE_AX);
      if (kind == 0 && st_ != 0) { ob.f12_sqr(VE_S1); oc.f12_sqr(VE_S2); }
      ob.f12_mul_line_fixed(VE_S1, 0, st_, VE_LX);
      oc.f12_mul_line_fixed(VE_S2, 1, st_, VE_CX);
    }
    (void)hipEventRecord(a.split_ev[1], ob.s); (void)hipEventRecord(a.split_ev[2], oc.s);
    (void)hipStreamWaitEvent(s, a.split_ev[1], 0); (void)hipStreamWaitEvent(s, a.split_ev[2], 0);
    ops.f12_mul(VE_F, VE_F, VE_S1); ops.f12_mul(VE_F, VE_F, VE_S2);
  } else {
    // Steps of the Miller loop per launch (k_miller_run: f never leaves LDS + registers inside a launch).  Large sub-batches take the whole loop in ONE
    // launch; sub-batches that are a single generation of workgroups run measurably better in a few shorter launches (batch 2^17 = two sub-batches of
    // 2^16: 11 steps per launch 5.75 M proofs/s, 22: 5.71, 44: 5.60, 88: 5.49; batch 2^19: 44 best; 2^20: 88 best by 0.7 %; profiles/r03_run_steps_sweep.txt);
    // a batch that is ONE sub-batch (up to 65 536 proofs) takes the whole loop in one launch: 14.23 ms against 14.29 ms with 11 steps at 65 536 (g16_launch_form)
    const int run_steps = form.run_steps;
    if (run_steps) vm_miller_program_runs(ops, run_steps);
    else vm_miller_program(ops, step_kinds_host(), true);
  }
  if (ev) (void)hipEventRecord(ev[2], s);
  // r-torsion test of B from the loop's final point; resolves the deferred statuses (C errors, input count)
  BN_LAUNCH(KID_SUBGROUP, k_g16_subgroup, n, a.ws, a.status, a.inputs_match_key, (int)VE_T);
  if (ev) (void)hipEventRecord(ev[3], s);
  vm_final_exp_program(ops);
  BN_LAUNCH(KID_COMPARE, k_g16_compare, a.ws, n, a.status, a.target, BN254_ST_REJECT);
  if (ev) (void)hipEventRecord(ev[4], s);
  return hipGetLastError();
}

// ---- RLC batch mode: per-proof stage, fold, group stage, scatter (bn254_rlc.h; orchestration of the fallback in bn254_capi.hip) ---------------------
hipError_t bn254_launch_g16_rlc(const G16LaunchArgs& a, const RlcLaunchArgs& r, hipStream_t s) {
  unsigned grid = grid_for(a.n);
  uint32_t n = (uint32_t)a.n;
  G16Prof* prof = nullptr;
  ChaChaKey key;
  for (int i = 0; i < 8; i++) key.k[i] = r.key[i];
  for (int i = 0; i < 3; i++) key.nonce[i] = r.key[8 + i];
  // parse + checks; the public-input MSM is skipped (done once per group): the wide flag of k_g16_prepare returns before it
  BN_LAUNCH(KID_PREPARE, k_g16_prepare, a.proofs, a.stride, a.inputs, a.n_public, n, a.ws, a.status, a.msm_tab, a.k0, a.inputs_match_key, 1);
  if (a.strict_scalars && a.n_public > 0) hipLaunchKernelGGL(k_g16_check_scalars, dim3(grid), dim3(256), 0, s, a.inputs, a.n_public, n, a.status);
  LaunchOps ops{a.ws, n, a.status, grid, s, {nullptr, nullptr, nullptr}, nullptr};
  BN_LAUNCH(KID_VM_INIT, k_vm_init, a.ws, n, (const uint8_t*)a.status);
  hipLaunchKernelGGL(k_rlc_scale, dim3(grid), dim3(256), 0, s, a.ws, n, (const uint8_t*)a.status, a.inputs, a.n_public, key, r.counter_base);
  const uint8_t* kinds = step_kinds_host();
  const int share = 1 << r.plan.pre;
  if (share == 1) {
    vm_miller_program(ops, kinds, false);
  } else {
    // shared accumulators: lane j < m walks proofs j, j + m, ..., j + (share - 1) m; one squaring of f per lane and step
    const uint32_t m = r.plan.lanes;
    const unsigned mgrid = grid_for(m);
    hipLaunchKernelGGL(k_rlc_init_f, dim3(mgrid), dim3(256), 0, s, a.ws, n, m);
    for (int st_ = 0; st_ < BN_ATE_STEPS; st_++) {
      const int kind = kinds[st_];
      if (kind == 0 && st_ != 0) hipLaunchKernelGGL(k_rlc_miller_multi<true>, dim3(mgrid), dim3(256), 0, s, a.ws, n, (const uint8_t*)a.status, kind, share, m);
      else hipLaunchKernelGGL(k_rlc_miller_multi<false>, dim3(mgrid), dim3(256), 0, s, a.ws, n, (const uint8_t*)a.status, kind, share, m);
    }
  }
  BN_LAUNCH(KID_SUBGROUP, k_g16_subgroup, n, a.ws, a.status, a.inputs_match_key, (int)VE_T);
  hipLaunchKernelGGL(k_rlc_neutral, dim3(grid), dim3(256), 0, s, a.ws, n, (const uint8_t*)a.status, a.n_public, share == 1 ? 1 : 0);
  uint32_t cur = n;
  for (int k = 0; k < r.plan.rounds; k++) {
    const uint32_t half = r.plan.half[k];
    hipLaunchKernelGGL(k_rlc_fold, dim3(grid_for(cur - half)), dim3(256), 0, s, a.ws, n, cur, half, a.n_public, k >= r.plan.pre ? 1 : 0);
    cur = half;
  }
  // group stage: lanes [0, groups) of the same workspace, their own status bytes
  const uint32_t groups = r.plan.groups;
  const unsigned ggrid = grid_for(groups);
  (void)hipMemsetAsync(r.grp_status, 0, ((size_t)groups + 255) / 256 * 256, s);
  hipLaunchKernelGGL(k_rlc_group_points, dim3(ggrid), dim3(256), 0, s, a.ws, n, r.grp_status, groups, a.n_public, r.rlc_tab, a.msm_tab);
  LaunchOps gops{a.ws, n, r.grp_status, ggrid, s, {a.gtab, a.dtab, r.btab}, nullptr};
  gops.inf_mask[0] = BN254_ST_LINF; gops.inf_mask[1] = BN254_ST_LINF2; gops.inf_mask[2] = BN254_ST_LINF3;
  for (int st_ = 0; st_ < BN_ATE_STEPS; st_++) {
    const int32_t *t0 = a.gtab + (size_t)st_ * FIXED_LINE_DWORDS, *t1 = a.dtab + (size_t)st_ * FIXED_LINE_DWORDS, *t2 = r.btab + (size_t)st_ * FIXED_LINE_DWORDS;
    if (kinds[st_] == 0 && st_ != 0) hipLaunchKernelGGL(k_rlc_group_step<true>, dim3(ggrid), dim3(256), 0, s, a.ws, n, (const uint8_t*)r.grp_status, (int)RLC_ACC, t0, t1, t2);
    else hipLaunchKernelGGL(k_rlc_group_step<false>, dim3(ggrid), dim3(256), 0, s, a.ws, n, (const uint8_t*)r.grp_status, (int)RLC_ACC, t0, t1, t2);
  }
  gops.f12_mul(VE_F, VE_F, RLC_ACC);
  vm_final_exp_program(gops);
  { unsigned grid = ggrid; BN_LAUNCH(KID_COMPARE, k_g16_compare, a.ws, n, r.grp_status, r.one, BN254_ST_REJECT); }
  hipLaunchKernelGGL(k_rlc_scatter, dim3(grid), dim3(256), 0, s, a.status, n, (const uint8_t*)r.grp_status, r.plan);
  return hipGetLastError();
}
hipError_t bn254_launch_gather_rows(uint8_t* dst, const uint8_t* src, size_t src_stride, uint32_t row_bytes, const uint32_t* idx, uint32_t m, hipStream_t s) {
  const size_t threads = (size_t)m * (row_bytes / 4);
  if (threads) hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, dst, src, src_stride, row_bytes, idx, m);
  return hipGetLastError();
}
hipError_t bn254_launch_scatter_status(uint8_t* status, const uint8_t* fb_status, const uint32_t* idx, uint32_t m, hipStream_t s) {
  if (m) hipLaunchKernelGGL(k_scatter_status, dim3(grid_for(m)), dim3(256), 0, s, status, fb_status, idx, m);
  return hipGetLastError();
}
// lane-level multiply-adds per second of the current device: the best launch of k_valu_peak at four wavefronts per SIMD, each about 2 ms long (a
// launch of 0.15 ms measured 20 % low: launch ramp and clocks), once `reps` launches in a row have not improved on it; 0 on failure
double bn254_measure_valu_peak(int reps) {
  hipDeviceProp_t p;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0.0;
  const int grid = p.multiProcessorCount * 4;             // 256 threads = one wavefront on each SIMD of a CU; four blocks per CU
  uint32_t* out = nullptr;
  if (hipMalloc((void**)&out, (size_t)grid * 256 * 4) != hipSuccess) return 0.0;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_valu_peak, dim3(grid), dim3(256), 0, nullptr, out, 1u, VALU_PEAK_ITERS);
  // The clocks of an idle GPU take tens of milliseconds of load to settle (the first five launches measured 29.7 T, the next five 32.2, then 33.2 against the
  // 34.8 T of a probe that runs for seconds): keep launching until the best has not improved for `reps` launches in a row, 200 launches (0.4 s) at most.
  float best = 1e30f;
  int since_best = 0;
  for (int r = 0; r < 200 && since_best < reps; r++) {
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(k_valu_peak, dim3(grid), dim3(256), 0, nullptr, out, (uint32_t)(r + 2), VALU_PEAK_ITERS);
    (void)hipEventRecord(e1, nullptr);
    if (hipEventSynchronize(e1) != hipSuccess) { best = 1e30f; break; }
    float ms = 0.f;
    since_best++;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f && ms < best * 0.998f) { best = ms; since_best = 0; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
  if (best > 1e29f) return 0.0;
  return (double)grid * 256.0 * (double)VALU_PEAK_ITERS * 16.0 / ((double)best * 1e-3);
}
// The same kernel back to back for about `ms_target` milliseconds, timed as ONE interval: the rate the box sustains over the length of the path's long kernels
// (k_miller_run runs for 50 .. 100 ms) -- boxes whose best 2 ms launch agrees to 1 % differ by 3 % here (profiles/r05_box_variance.txt).
double bn254_measure_valu_sustained(double ms_target) {
  hipDeviceProp_t p;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0.0;
  const int grid = p.multiProcessorCount * 4;
  uint32_t* out = nullptr;
  if (hipMalloc((void**)&out, (size_t)grid * 256 * 4) != hipSuccess) return 0.0;
  int launches = (int)(ms_target / 2.0);
  if (launches < 4) launches = 4;
  if (launches > 1000) launches = 1000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, nullptr);
  for (int r = 0; r < launches; r++) hipLaunchKernelGGL(k_valu_peak, dim3(grid), dim3(256), 0, nullptr, out, (uint32_t)(r + 1), VALU_PEAK_ITERS);
  (void)hipEventRecord(e1, nullptr);
  float ms = 0.f;
  const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
  return ok ? (double)launches * grid * 256.0 * (double)VALU_PEAK_ITERS * 16.0 / ((double)ms * 1e-3) : 0.0;
}
hipError_t bn254_launch_dbg_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_dbg_fp_mul, dim3(grid_for(n)), dim3(256), 0, s, a, b, o, n);
  return hipGetLastError();
}
// op: 0 mul 1 sqr 2 inv 3 cyclo_sqr(easy part) 4 frob1.  status: n scratch bytes on the device
hipError_t bn254_launch_dbg_fp12_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n, int32_t* ws, uint8_t* status, hipStream_t s) {
  unsigned g = grid_for(n);
  uint32_t nn = (uint32_t)n;
  LaunchOps ops{ws, nn, status, g, s, {nullptr, nullptr, nullptr}, nullptr};
  hipLaunchKernelGGL(k_dbg_load, dim3(g), dim3(256), 0, s, ws, nn, status, (int)VE_F, a, 0);
  if (op == 0) { hipLaunchKernelGGL(k_dbg_load, dim3(g), dim3(256), 0, s, ws, nn, status, (int)VE_S1, b, 0); ops.f12_mul(VE_S0, VE_F, VE_S1); }
  else if (op == 1) { ops.f12_sqr(VE_F); ops.f12_conj(VE_S0, VE_F); ops.f12_conj(VE_S0, VE_S0); }
  else if (op == 2) ops.f12_inv(VE_S0, VE_F);
  else if (op == 3) {
    ops.f12_inv(VE_S0, VE_F); ops.f12_conj(VE_S1, VE_F); ops.f12_mul(VE_S0, VE_S1, VE_S0); ops.f12_frob(VE_S1, VE_S0, 2);
    ops.f12_mul(VE_S0, VE_S1, VE_S0); ops.f12_cyclo_sqr(VE_S0, VE_S0);
  } else ops.f12_frob(VE_S0, VE_F, 1);
  hipLaunchKernelGGL(k_dbg_store, dim3(g), dim3(256), 0, s, ws, nn, (int)VE_S0, o);
  return hipGetLastError();
}
hipError_t bn254_launch_dbg_pairing(const uint8_t* g1, const uint8_t* g2, uint8_t* o, size_t n, int32_t* ws, uint8_t* status, hipStream_t s) {
  unsigned g = grid_for(n);
  uint32_t nn = (uint32_t)n;
  LaunchOps ops{ws, nn, status, g, s, {nullptr, nullptr, nullptr}, nullptr};
  hipLaunchKernelGGL(k_dbg_load, dim3(g), dim3(256), 0, s, ws, nn, status, 0, g1, 1);
  hipLaunchKernelGGL(k_dbg_load, dim3(g), dim3(256), 0, s, ws, nn, status, 0, g2, 2);
  hipLaunchKernelGGL(k_vm_init, dim3(g), dim3(256), 0, s, ws, nn, (const uint8_t*)status);
  vm_miller_program(ops, step_kinds_host(), false);
  vm_final_exp_program(ops);
  hipLaunchKernelGGL(k_dbg_store, dim3(g), dim3(256), 0, s, ws, nn, (int)VE_S0, o);
  return hipGetLastError();
}
// r-torsion test through the product path: T from the Miller program (variable pair only, A = any G1 point), then the ate relation
hipError_t bn254_launch_dbg_g2_ate(const uint8_t* g1, const uint8_t* g2, uint8_t* o, size_t n, int32_t* ws, uint8_t* status, hipStream_t s) {
  unsigned g = grid_for(n);
  uint32_t nn = (uint32_t)n;
  LaunchOps ops{ws, nn, status, g, s, {nullptr, nullptr, nullptr}, nullptr};
  hipLaunchKernelGGL(k_dbg_load, dim3(g), dim3(256), 0, s, ws, nn, status, 0, g1, 1);
  hipLaunchKernelGGL(k_dbg_load, dim3(g), dim3(256), 0, s, ws, nn, status, 0, g2, 2);
  hipLaunchKernelGGL(k_vm_init, dim3(g), dim3(256), 0, s, ws, nn, (const uint8_t*)status);
  vm_miller_program(ops, step_kinds_host(), false);
  hipLaunchKernelGGL(k_dbg_g2_ate, dim3(g), dim3(256), 0, s, ws, nn, (const uint8_t*)status, o);
  return hipGetLastError();
}
hipError_t bn254_launch_dbg_g2_subgroup(const uint8_t* g2, uint8_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_dbg_g2_subgroup, dim3(grid_for(n)), dim3(256), 0, s, g2, o, n);
  return hipGetLastError();
}

// ---- PlonK: the two-fixed-pair pairing check (the MSM stages: bn254_k_msm.hip) -------------------------------------------------------------------
// prod_t e(P_t, Q_t) == 1 for two key-side G2 points (line tables tab0, tab1) and per-item G1 points already in the workspace
// (P_0 at VE_LX, P_1 at VE_CX; identity flags BN254_ST_LINF / BN254_ST_LINF2 in the status byte): ACCEPT or reject_code
hipError_t bn254_launch_pairing2_fixed(int32_t* ws, uint8_t* status, size_t n, const int32_t* tab0, const int32_t* tab1, const int32_t* target_one,
                                       int reject_code, hipStream_t s, hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join) {
  unsigned grid = grid_for(n);
  uint32_t nn = (uint32_t)n;
  G16Prof* prof = nullptr;
  LaunchOps ops{ws, nn, status, grid, s, {tab0, tab1, nullptr}, nullptr};
  ops.inf_mask[0] = BN254_ST_LINF; ops.inf_mask[1] = BN254_ST_LINF2;
  static const bool coop_on = [] { const char* e = getenv("BN254_COOP"); return !e || atoi(e) != 0; }();
  static const size_t coop_fixed_max = [] { const char* e = getenv("BN254_COOP_FIXED_MAX"); return e ? (size_t)atol(e) : bn254_coop_max_proofs_fixed(); }();
  if (coop_on && n <= coop_fixed_max) {
    // small batch: the cooperative layout (bn254_coop12.hip), Miller loop of the two pairs, final exponentiation and the comparison in ONE launch
    return bn254_coop12_miller_fixed(ws, status, n, 2, tab0, tab1, tab0, VE_LX, VE_CX, VE_LX, BN254_ST_LINF, BN254_ST_LINF2, 0, 1, target_one, reject_code, s);
  }
  BN_LAUNCH(KID_VM_INIT, k_vm_init, ws, nn, (const uint8_t*)status);
  const uint8_t* kinds = step_kinds_host();
  if (aux && n <= G16_SPLIT_MAX_PROOFS) {
    // latency mode (as in bn254_launch_g16): the two pairs as two concurrent chains, multiplied at the end
    ops.f12_copy(VE_S1, VE_F);
    (void)hipEventRecord(ev_fork, s);
    LaunchOps ob = ops; ob.s = aux;
    (void)hipStreamWaitEvent(aux, ev_fork, 0);
    for (int st_ = 0; st_ < BN_ATE_STEPS; st_++) {
      if (kinds[st_] == 0 && st_ != 0) { ops.f12_sqr(VE_F); ob.f12_sqr(VE_S1); }
      ops.f12_mul_line_fixed(VE_F, 0, st_, VE_LX);
      ob.f12_mul_line_fixed(VE_S1, 1, st_, VE_CX);
    }
    (void)hipEventRecord(ev_join, aux);
    (void)hipStreamWaitEvent(s, ev_join, 0);
    ops.f12_mul(VE_F, VE_F, VE_S1);
  } else {
    // a large batch (throughput): the whole loop of the two pairs in one launch, the accumulator in flight (k_miller_run_fixed2); BN254_MILLER_RUN_STEPS=0 keeps
    // the one-launch-per-operation form
    static const int run_steps = [] { const char* e = getenv("BN254_MILLER_RUN_STEPS"); int v = e ? atoi(e) : BN_ATE_STEPS; return v < 0 ? BN_ATE_STEPS : v; }();
    if (run_steps > 0) {
      for (int s0 = 0; s0 < BN_ATE_STEPS; s0 += run_steps) ops.miller_run_fixed2(s0, s0 + run_steps < BN_ATE_STEPS ? s0 + run_steps : BN_ATE_STEPS, VE_F, VE_LX, VE_CX);
    } else {
      for (int st_ = 0; st_ < BN_ATE_STEPS; st_++) {
        if (kinds[st_] == 0 && st_ != 0) ops.f12_sqr(VE_F);
        ops.f12_mul_line_fixed2(VE_F, st_, VE_LX, VE_CX);
      }
    }
  }
  vm_final_exp_program(ops);
  BN_LAUNCH(KID_COMPARE, k_g16_compare, ws, nn, status, target_one, reject_code);
  return hipGetLastError();
}
