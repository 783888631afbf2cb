// bn254_kernels.hip -- the gfx950 kernels of the Groth16 batch verifier, one proof per lane.
//
// Pipeline per batch (DESIGN.md "Kernels"):
//   k_g16_prepare   parse 256 proof bytes (coalesced through LDS), range / on-curve checks of A, B, C, Montgomery
//                   conversion, L = K0 + sum x_i K_i by fixed-base 8-bit windows        (groth16/converter.rs:14-26, verify.rs:53-63)
//   k_g16_subgroup  r-torsion test of B, status precedence                              (converter.rs:152)
//   k_g16_miller    f = Miller(A,B) * lines_G(L) * lines_D(C), G/D tables shared by the batch (verify.rs:73-77)
//   k_g16_finalexp  f^((p^12-1)/r) == e(alpha,beta) -> status byte                      (verify.rs:77)
// Intermediate state lives in an SoA workspace in HBM: element e, limb l, proof i at ws[(e * 9 + l) * n + i], so
// that every load and store of a wave is one contiguous 256-byte segment.
#include <hip/hip_runtime.h>
#include "bn254_pairing.h"
#include "bn254_kernels.h"

namespace bn254 {

// ---- SoA workspace accessors -------------------------------------------------------------------------------------
struct Ws {
  int32_t* base;
  size_t n;
};
__device__ __forceinline__ Fp ws_ld(const Ws& w, int e, size_t i) {
  Fp r;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) r.v[l] = w.base[(size_t)(e * BN_NL + l) * w.n + i];
  return r;
}
__device__ __forceinline__ void ws_st(const Ws& w, int e, size_t i, const Fp& a) {
#pragma unroll
  for (int l = 0; l < BN_NL; l++) w.base[(size_t)(e * BN_NL + l) * w.n + i] = a.v[l];
}
__device__ __forceinline__ Fp2 ws_ld2(const Ws& w, int e, size_t i) { Fp2 r; r.c0 = ws_ld(w, e, i); r.c1 = ws_ld(w, e + 1, i); return r; }
__device__ __forceinline__ void ws_st2(const Ws& w, int e, size_t i, const Fp2& a) { ws_st(w, e, i, a.c0); ws_st(w, e + 1, i, a.c1); }
__device__ __forceinline__ Fp12 ws_ld12(const Ws& w, int e, size_t i) {
  Fp12 r;
  r.c0.c0 = ws_ld2(w, e, i); r.c0.c1 = ws_ld2(w, e + 2, i); r.c0.c2 = ws_ld2(w, e + 4, i);
  r.c1.c0 = ws_ld2(w, e + 6, i); r.c1.c1 = ws_ld2(w, e + 8, i); r.c1.c2 = ws_ld2(w, e + 10, i);
  return r;
}
__device__ __forceinline__ void ws_st12(const Ws& w, int e, size_t i, const Fp12& a) {
  ws_st2(w, e, i, a.c0.c0); ws_st2(w, e + 2, i, a.c0.c1); ws_st2(w, e + 4, i, a.c0.c2);
  ws_st2(w, e + 6, i, a.c1.c0); ws_st2(w, e + 8, i, a.c1.c1); ws_st2(w, e + 10, i, a.c1.c2);
}
// uniform (batch-constant) data: limbs stored contiguously per element
__device__ __forceinline__ Fp uni_ld(const int32_t* p) {
  Fp r;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) r.v[l] = p[l];
  return r;
}
__device__ __forceinline__ Fp2 uni_ld2(const int32_t* p) { Fp2 r; r.c0 = uni_ld(p); r.c1 = uni_ld(p + BN_NL); return r; }
__device__ __forceinline__ FixedLine uni_ld_line(const int32_t* tab, int idx) {
  FixedLine l;
  l.m = uni_ld2(tab + (size_t)idx * 4 * BN_NL);
  l.c = uni_ld2(tab + (size_t)idx * 4 * BN_NL + 2 * BN_NL);
  return l;
}

enum { E_AX = 0, E_AY = 1, E_BX = 2, E_BY = 4, E_CX = 6, E_CY = 7, E_LX = 8, E_LY = 9, E_LZ = 10, E_F = 11 };

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
k_g16_prepare(const uint8_t* __restrict__ proofs, size_t stride, const uint8_t* __restrict__ inputs, int n_public, size_t n,
              Ws ws, uint8_t* __restrict__ status, const int32_t* __restrict__ msm_tab, const int32_t* __restrict__ k0,
              int inputs_match_key) {
  __shared__ uint32_t lds[4 * 64 * PREP_LDS_ROW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t first = (size_t)blockIdx.x * 256 + (size_t)wave * 64;
  uint32_t* wl = lds + wave * 64 * PREP_LDS_ROW;
  const bool aligned = ((((uintptr_t)proofs) | stride) & 3) == 0;
  if (aligned) {
    // record j of this wave: one 256-byte contiguous segment per load instruction
    for (int j = 0; j < 64; j++) {
      size_t rec = first + j;
      uint32_t v = 0;
      if (rec < n) v = *(const uint32_t*)(proofs + rec * stride + (size_t)lane * 4);
      wl[j * PREP_LDS_ROW + lane] = v;
    }
  } else {
    for (int j = 0; j < 64; j++) {
      size_t rec = first + j;
      uint32_t v = 0;
      if (rec < n) {
        const uint8_t* p = proofs + rec * stride + (size_t)lane * 4;
        v = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
      }
      wl[j * PREP_LDS_ROW + lane] = v;
    }
  }
  __syncthreads();
  const size_t i = first + lane;
  const bool live = i < n;
  const size_t ii = live ? i : n - 1;
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
  if (live) { ws_st(ws, E_AX, i, A.x); ws_st(ws, E_AY, i, A.y); }

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
  if (live) { ws_st2(ws, E_BX, i, B.x); ws_st2(ws, E_BY, i, B.y); }

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
  if (live) { ws_st(ws, E_CX, i, C.x); ws_st(ws, E_CY, i, C.y); }

  // ---- L = K0 + sum_i x_i K_i, x_i taken as raw 256-bit integers (no range check, as bn::Fr::from_slice)
  G1Aff K0; K0.x = uni_ld(k0); K0.y = uni_ld(k0 + BN_NL);
  G1Proj L = g1_from_affine(K0);
  if (inputs_match_key) {
    for (int s = 0; s < n_public; s++) {
      const uint8_t* sp = inputs + (ii * (size_t)n_public + s) * 32;
      uint32_t sw[8];
      if ((((uintptr_t)inputs) & 3) == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) sw[k] = ((const uint32_t*)sp)[k];
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) sw[k] = (uint32_t)sp[4 * k] | (uint32_t)sp[4 * k + 1] << 8 | (uint32_t)sp[4 * k + 2] << 16 | (uint32_t)sp[4 * k + 3] << 24;
      }
      // byte j of the big-endian scalar is sw[j / 4] >> (8 (j % 4)); window w (weight 2^(8w)) is byte 31 - w
      for (int w = 0; w < 32; w++) {
        int j = 31 - w;
        uint32_t dig = (sw[j >> 2] >> (8 * (j & 3))) & 0xff;
        if (dig != 0) {
          const int32_t* e = msm_tab + ((size_t)(s * 32 + w) * 255 + (dig - 1)) * MSM_ENTRY_DWORDS;
          G1Aff q;
#pragma unroll
          for (int l = 0; l < BN_NL; l++) { q.x.v[l] = e[l]; q.y.v[l] = e[BN_NL + l]; }
          L = g1_add_mixed(L, q);
        }
      }
    }
  }
  if (live) { ws_st(ws, E_LX, i, fp_reduce(L.x)); ws_st(ws, E_LY, i, fp_reduce(L.y)); ws_st(ws, E_LZ, i, fp_reduce(L.z)); }
  if (live) status[i] = err ? (uint8_t)err : (uint8_t)(BN254_ST_PENDING | err_c);
}

// =====================================================================================================================
// k_g16_subgroup
// =====================================================================================================================
__global__ void __launch_bounds__(256, 2)
k_g16_subgroup(size_t n, Ws ws, uint8_t* __restrict__ status, int inputs_match_key) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  const size_t ii = live ? i : n - 1;
  uint8_t st = status[ii];
  // a wave whose proofs all failed earlier has nothing to do
  if (__builtin_amdgcn_ballot_w64((st & BN254_ST_PENDING) != 0) == 0) return;
  G2Aff B; B.x = ws_ld2(ws, E_BX, ii); B.y = ws_ld2(ws, E_BY, ii);
  bool ok = g2_in_subgroup(B);
  if (live && (st & BN254_ST_PENDING)) {
    uint8_t out;
    if (!ok) out = BN254_ST_NOT_IN_SUBGROUP;
    else if (st & 0x7f) out = st & 0x7f;                      // deferred error of C
    else if (!inputs_match_key) out = BN254_ST_INPUT_LEN;     // PrepareInputsFailed comes after every loader error
    else out = BN254_ST_PENDING;
    status[i] = out;
  }
}

// =====================================================================================================================
// k_g16_miller : (A, B) variable, (L, G) and (C, D) against the key's line tables
// =====================================================================================================================
__global__ void __launch_bounds__(256, 2)
k_g16_miller(size_t n, Ws ws, const uint8_t* __restrict__ status, const int32_t* __restrict__ gtab, const int32_t* __restrict__ dtab) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t ii = i < n ? i : n - 1;
  if (__builtin_amdgcn_ballot_w64(status[ii] == BN254_ST_PENDING) == 0) return;
  G1Aff A; A.x = ws_ld(ws, E_AX, ii); A.y = ws_ld(ws, E_AY, ii);
  G1Aff C; C.x = ws_ld(ws, E_CX, ii); C.y = ws_ld(ws, E_CY, ii);
  G1Proj L; L.x = ws_ld(ws, E_LX, ii); L.y = ws_ld(ws, E_LY, ii); L.z = ws_ld(ws, E_LZ, ii);
  G2Aff B; B.x = ws_ld2(ws, E_BX, ii); B.y = ws_ld2(ws, E_BY, ii);
  G2Aff nB = g2_neg(B);
  G2Proj T = g2_from_affine(B);
  Fp12 f = fp12_one();
  int idx = 0;
  for (int it = 1; it < BN_ATE_NAF_LEN; it++) {
    f = fp12_sqr(f);
    {
      G2Line l = g2_double_step(T);
      f = miller_mul_var(f, l, A);
      f = miller_mul_fixed_proj(f, uni_ld_line(gtab, idx), L);
      f = miller_mul_fixed_aff(f, uni_ld_line(dtab, idx), C);
      idx++;
    }
    int dgt = BN_ATE_NAF[it];
    if (dgt != 0) {  // public constant: wave-uniform
      G2Line l = g2_add_step(T, dgt > 0 ? B : nB);
      f = miller_mul_var(f, l, A);
      f = miller_mul_fixed_proj(f, uni_ld_line(gtab, idx), L);
      f = miller_mul_fixed_aff(f, uni_ld_line(dtab, idx), C);
      idx++;
    }
  }
  G2Aff Q1 = g2_psi_affine(B);
  G2Aff Q2 = g2_neg(g2_psi2_affine(B));
#pragma unroll 1
  for (int s = 0; s < 2; s++) {
    G2Line l = g2_add_step(T, s == 0 ? Q1 : Q2);
    f = miller_mul_var(f, l, A);
    f = miller_mul_fixed_proj(f, uni_ld_line(gtab, idx), L);
    f = miller_mul_fixed_aff(f, uni_ld_line(dtab, idx), C);
    idx++;
  }
  if (i < n) ws_st12(ws, E_F, i, fp12_reduce(f));
}

// =====================================================================================================================
// k_g16_finalexp
// =====================================================================================================================
__global__ void __launch_bounds__(256, 2)
k_g16_finalexp(size_t n, Ws ws, uint8_t* __restrict__ status, const int32_t* __restrict__ target) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t ii = i < n ? i : n - 1;
  uint8_t st = status[ii];
  if (__builtin_amdgcn_ballot_w64(st == BN254_ST_PENDING) == 0) return;
  Fp12 f = ws_ld12(ws, E_F, ii);
  Fp12 e = final_exponentiation(f);
  Fp12 t;
  t.c0.c0 = uni_ld2(target); t.c0.c1 = uni_ld2(target + 2 * BN_NL); t.c0.c2 = uni_ld2(target + 4 * BN_NL);
  t.c1.c0 = uni_ld2(target + 6 * BN_NL); t.c1.c1 = uni_ld2(target + 8 * BN_NL); t.c1.c2 = uni_ld2(target + 10 * BN_NL);
  bool acc = fp12_eq(e, t);
  if (i < n && st == BN254_ST_PENDING) status[i] = acc ? BN254_ST_ACCEPT : BN254_ST_REJECT;
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
__device__ Fp12 probe_ld_fp12(const uint8_t* p) {
  Fp12 r;
  Fp2* c[6] = {&r.c0.c0, &r.c0.c1, &r.c0.c2, &r.c1.c0, &r.c1.c1, &r.c1.c2};
  for (int k = 0; k < 6; k++) { c[k]->c0 = probe_ld_fp(p + 64 * k); c[k]->c1 = probe_ld_fp(p + 64 * k + 32); }
  return r;
}
__device__ void probe_st_fp12(uint8_t* p, const Fp12& a) {
  const Fp2* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  for (int k = 0; k < 6; k++) { probe_st_fp(p + 64 * k, c[k]->c0); probe_st_fp(p + 64 * k + 32, c[k]->c1); }
}
__global__ void __launch_bounds__(256, 2) k_dbg_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  probe_st_fp(o + 32 * i, fp_mul(probe_ld_fp(a + 32 * i), probe_ld_fp(b + 32 * i)));
}
__global__ void __launch_bounds__(256, 2) k_dbg_fp12_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Fp12 x = probe_ld_fp12(a + 384 * i), r = x;
  if (op == 0) r = fp12_mul(x, probe_ld_fp12(b + 384 * i));
  else if (op == 1) r = fp12_sqr(x);
  else if (op == 2) r = fp12_inv(x);
  else if (op == 3) r = fp12_cyclo_sqr(final_exp_easy(x));
  else if (op == 4) r = fp12_frob(x, 1);
  probe_st_fp12(o + 384 * i, r);
}
__global__ void __launch_bounds__(256, 2) k_dbg_pairing(const uint8_t* g1, const uint8_t* g2, uint8_t* o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  G1Aff p; p.x = probe_ld_fp(g1 + 64 * i); p.y = probe_ld_fp(g1 + 64 * i + 32);
  G2Aff q; q.x.c1 = probe_ld_fp(g2 + 128 * i); q.x.c0 = probe_ld_fp(g2 + 128 * i + 32);
  q.y.c1 = probe_ld_fp(g2 + 128 * i + 64); q.y.c0 = probe_ld_fp(g2 + 128 * i + 96);
  Fp12 f = miller_loop<0>(p, q, nullptr, nullptr);
  probe_st_fp12(o + 384 * i, final_exponentiation(f));
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

hipError_t bn254_launch_g16(const G16LaunchArgs& a, hipStream_t s, hipEvent_t* ev /* 5 events or nullptr */) {
  Ws ws{a.ws, a.n};
  unsigned g = grid_for(a.n);
  if (ev) (void)hipEventRecord(ev[0], s);
  hipLaunchKernelGGL(k_g16_prepare, dim3(g), dim3(256), 0, s, a.proofs, a.stride, a.inputs, a.n_public, a.n, ws, a.status,
                     a.msm_tab, a.k0, a.inputs_match_key);
  if (ev) (void)hipEventRecord(ev[1], s);
  hipLaunchKernelGGL(k_g16_subgroup, dim3(g), dim3(256), 0, s, a.n, ws, a.status, a.inputs_match_key);
  if (ev) (void)hipEventRecord(ev[2], s);
  hipLaunchKernelGGL(k_g16_miller, dim3(g), dim3(256), 0, s, a.n, ws, (const uint8_t*)a.status, a.gtab, a.dtab);
  if (ev) (void)hipEventRecord(ev[3], s);
  hipLaunchKernelGGL(k_g16_finalexp, dim3(g), dim3(256), 0, s, a.n, ws, a.status, a.target);
  if (ev) (void)hipEventRecord(ev[4], s);
  return hipGetLastError();
}
hipError_t bn254_launch_dbg_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_dbg_fp_mul, dim3(grid_for(n)), dim3(256), 0, s, a, b, o, n);
  return hipGetLastError();
}
hipError_t bn254_launch_dbg_fp12_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_dbg_fp12_op, dim3(grid_for(n)), dim3(256), 0, s, op, a, b, o, n);
  return hipGetLastError();
}
hipError_t bn254_launch_dbg_pairing(const uint8_t* g1, const uint8_t* g2, uint8_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_dbg_pairing, dim3(grid_for(n)), dim3(256), 0, s, g1, g2, o, n);
  return hipGetLastError();
}
hipError_t bn254_launch_dbg_g2_subgroup(const uint8_t* g2, uint8_t* o, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_dbg_g2_subgroup, dim3(grid_for(n)), dim3(256), 0, s, g2, o, n);
  return hipGetLastError();
}
