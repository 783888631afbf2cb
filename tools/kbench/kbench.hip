// kernel micro-benchmark: one Fp12-level op variant per build (-DV_xxx), timed at n = 2^20 lanes, 4-byte-stride SoA rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../snark-bn254-verifier_amd/csrc/bn254_pairing.h"
using namespace bn254;
struct Ws { __amdgpu_buffer_rsrc_t rsrc; uint32_t row_bytes; uint32_t voff;
  __device__ __forceinline__ Fp ld(int e) const { Fp r;
#ifdef NOLOAD
    uint32_t vo = threadIdx.x * 4u;   // every block reads block 0's lanes: cache hits
#else
    uint32_t vo = voff;
#endif
#pragma unroll
    for (int l = 0; l < 9; l++) r.v[l] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, (uint32_t)(e * 9 + l) * row_bytes, 0);
    return r; }
  __device__ __forceinline__ void st(int e, const Fp& a) const {
#ifdef NOSTORE
    if (a.v[8] != 0x7fffffff) return;  // never true for a normalised value, but keeps the computation alive
#endif
#pragma unroll
    for (int l = 0; l < 9; l++) __builtin_amdgcn_raw_buffer_store_b32(a.v[l], rsrc, voff, (uint32_t)(e * 9 + l) * row_bytes, 0); }
};
__device__ __forceinline__ Fp2 ld2(const Ws& w, int e) { Fp2 r; r.c0 = w.ld(e); r.c1 = w.ld(e + 1); return r; }
__device__ __forceinline__ void st2(const Ws& w, int e, const Fp2& a) { w.st(e, a.c0); w.st(e + 1, a.c1); }
#define NELEM 40
#define WS_SETUP Ws ws; ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, n * 4u * 9u * NELEM, 0x00020000); ws.row_bytes = n * 4u; ws.voff = (blockIdx.x * 256 + threadIdx.x) * 4u;
__global__ void k_fill(int32_t* base, uint32_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * 9 * NELEM) return;
  uint64_t x = i * 0x9E3779B97F4A7C15ull + 12345; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
  size_t row = i / n;
  int32_t d = (int32_t)(x & 0x1fffffff) - (1 << 28);
  if (row % 9 == 8) d = (int32_t)(x & 0x3fffff);  // top digit: value < p (22-bit top digit of a 254-bit number)
  base[i] = d;
}
#ifndef STAG_UNITS
#define STAG_UNITS 0
#endif
__device__ __forceinline__ void stagger() {
  // first-generation blocks start at different phases so that later generations do not load/compute in lock-step
  if (STAG_UNITS > 0 && blockIdx.x < 512) {
    uint32_t k = ((blockIdx.x * 2654435761u) >> 16) % STAG_UNITS;
    for (uint32_t i = 0; i < k; i++) __builtin_amdgcn_s_sleep(127);
  }
}
#if defined(V_SQR_OLD)
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  stagger();
  WS_SETUP
  Fp2 k0 = ld2(ws, 0), k1 = ld2(ws, 2), k2 = ld2(ws, 4), k3 = ld2(ws, 6), k4 = ld2(ws, 8), k5 = ld2(ws, 10);
  Fp2 x3 = fp2_mul_xi(k3), x4 = fp2_mul_xi(k4), x5 = fp2_mul_xi(k5);
  st2(ws, 0, fp2_dotp(pp(k0, k0), pp2(k1, x5), pp2(k2, x4), pp(k3, x3)));
  st2(ws, 2, fp2_dotp(pp2(k0, k1), pp2(k2, x5), pp2(k3, x4)));
  st2(ws, 4, fp2_dotp(pp2(k0, k2), pp(k1, k1), pp2(k3, x5), pp(k4, x4)));
  st2(ws, 6, fp2_dotp(pp2(k0, k3), pp2(k1, k2), pp2(k4, x5)));
  st2(ws, 8, fp2_dotp(pp2(k0, k4), pp2(k1, k3), pp(k2, k2), pp(k5, x5)));
  st2(ws, 10, fp2_dotp(pp2(k0, k5), pp2(k1, k4), pp2(k2, k3)));
}
#elif defined(V_SQR_K)
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  WS_SETUP
  Fp2 k0 = ld2(ws, 0), k1 = ld2(ws, 2), k2 = ld2(ws, 4), k3 = ld2(ws, 6), k4 = ld2(ws, 8), k5 = ld2(ws, 10);
  Fp2 x3 = fp2_mul_xi(k3), x4 = fp2_mul_xi(k4), x5 = fp2_mul_xi(k5);
  st2(ws, 0, fp2_dotk(ksq(k0), kp2(k1, x5), kp2(k2, x4), kp(k3, x3)));
  st2(ws, 2, fp2_dotk(kp2(k0, k1), kp2(k2, x5), kp2(k3, x4)));
  st2(ws, 4, fp2_dotk(kp2(k0, k2), ksq(k1), kp2(k3, x5), kp(k4, x4)));
  st2(ws, 6, fp2_dotk(kp2(k0, k3), kp2(k1, k2), kp2(k4, x5)));
  st2(ws, 8, fp2_dotk(kp2(k0, k4), kp2(k1, k3), ksq(k2), kp(k5, x5)));
  st2(ws, 10, fp2_dotk(kp2(k0, k5), kp2(k1, k4), kp2(k2, k3)));
}
#elif defined(V_LINE_OLD)
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  WS_SETUP
  Fp2 d0 = ld2(ws, 12), d3 = ld2(ws, 14), d4 = ld2(ws, 16);
  Fp2 x3 = fp2_mul_xi(d3), x4 = fp2_mul_xi(d4);
  Fp2 k0 = ld2(ws, 0), k1 = ld2(ws, 2), k2 = ld2(ws, 4), k3 = ld2(ws, 6), k4 = ld2(ws, 8), k5 = ld2(ws, 10);
  st2(ws, 0, fp2_dotp(pp(d0, k0), pp(x3, k5), pp(x4, k3)));
  st2(ws, 2, fp2_dotp(pp(d0, k1), pp(d3, k0), pp(x4, k4)));
  st2(ws, 4, fp2_dotp(pp(d0, k2), pp(d3, k1), pp(x4, k5)));
  st2(ws, 6, fp2_dotp(pp(d0, k3), pp(d3, k2), pp(d4, k0)));
  st2(ws, 8, fp2_dotp(pp(d0, k4), pp(d3, k3), pp(d4, k1)));
  st2(ws, 10, fp2_dotp(pp(d0, k5), pp(d3, k4), pp(d4, k2)));
}
#elif defined(V_LINE_K)
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  WS_SETUP
  Fp2 d0 = ld2(ws, 12), d3 = ld2(ws, 14), d4 = ld2(ws, 16);
  Fp2 x3 = fp2_mul_xi(d3), x4 = fp2_mul_xi(d4);
  Fp2 k0 = ld2(ws, 0), k1 = ld2(ws, 2), k2 = ld2(ws, 4), k3 = ld2(ws, 6), k4 = ld2(ws, 8), k5 = ld2(ws, 10);
  st2(ws, 0, fp2_dotk(kp(d0, k0), kp(x3, k5), kp(x4, k3)));
  st2(ws, 2, fp2_dotk(kp(d0, k1), kp(d3, k0), kp(x4, k4)));
  st2(ws, 4, fp2_dotk(kp(d0, k2), kp(d3, k1), kp(x4, k5)));
  st2(ws, 6, fp2_dotk(kp(d0, k3), kp(d3, k2), kp(d4, k0)));
  st2(ws, 8, fp2_dotk(kp(d0, k4), kp(d3, k3), kp(d4, k1)));
  st2(ws, 10, fp2_dotk(kp(d0, k5), kp(d3, k4), kp(d4, k2)));
}
#elif defined(V_LINEFP_OLD)
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  WS_SETUP
  Fp d0 = ws.ld(12); Fp2 d3 = ld2(ws, 14), d4 = ld2(ws, 16), x4 = ld2(ws, 18);
  Fp2 x3 = fp2_mul_xi(d3);
  Fp2 k0 = ld2(ws, 0), k1 = ld2(ws, 2), k2 = ld2(ws, 4), k3 = ld2(ws, 6), k4 = ld2(ws, 8), k5 = ld2(ws, 10);
  st2(ws, 0, fp2_dot_line(d0, k0, x3, k5, x4, k3));
  st2(ws, 2, fp2_dot_line(d0, k1, d3, k0, x4, k4));
  st2(ws, 4, fp2_dot_line(d0, k2, d3, k1, x4, k5));
  st2(ws, 6, fp2_dot_line(d0, k3, d3, k2, d4, k0));
  st2(ws, 8, fp2_dot_line(d0, k4, d3, k3, d4, k1));
  st2(ws, 10, fp2_dot_line(d0, k5, d3, k4, d4, k2));
}
#elif defined(V_LINEFP_K)
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  WS_SETUP
  Fp d0 = ws.ld(12); Fp2 d3 = ld2(ws, 14), d4 = ld2(ws, 16), x4 = ld2(ws, 18);
  Fp2 x3 = fp2_mul_xi(d3);
  Fp2 k0 = ld2(ws, 0), k1 = ld2(ws, 2), k2 = ld2(ws, 4), k3 = ld2(ws, 6), k4 = ld2(ws, 8), k5 = ld2(ws, 10);
  st2(ws, 0, fp2_dotk(kfp(k0, d0), kp(x3, k5), kp(x4, k3)));
  st2(ws, 2, fp2_dotk(kfp(k1, d0), kp(d3, k0), kp(x4, k4)));
  st2(ws, 4, fp2_dotk(kfp(k2, d0), kp(d3, k1), kp(x4, k5)));
  st2(ws, 6, fp2_dotk(kfp(k3, d0), kp(d3, k2), kp(d4, k0)));
  st2(ws, 8, fp2_dotk(kfp(k4, d0), kp(d3, k3), kp(d4, k1)));
  st2(ws, 10, fp2_dotk(kfp(k5, d0), kp(d3, k4), kp(d4, k2)));
}
#elif defined(V_CYC) || defined(V_CYC_K)
#ifndef WPE
#define WPE 2
#endif
#ifdef V_CYC_K
__device__ __forceinline__ void gs_pair_k(Fp2& za, Fp2& zb, const Fp2& a, const Fp2& b, const Fp2& sub, const Fp2& add, bool xi_on_cross) {
  Fp2 xb = fp2_mul_xi(b);
  Fp2 S = fp2_dotk(kp(xb, b), ksq(a));
  Fp2 T = xi_on_cross ? fp2_dotk(kp2(a, xb)) : fp2_dotk(kp2(a, b));
  za = fp2_lincomb_reduce(3, S, -2, sub);
  zb = fp2_lincomb_reduce(3, T, 2, add);
}
#define GSP gs_pair_k
#else
#define GSP gs_pair
#endif
__global__ void __launch_bounds__(256, WPE) k_op(int32_t* base, uint32_t n) {
  WS_SETUP
  {
    Fp2 k0 = ld2(ws, 0), k3 = ld2(ws, 6);
    Fp2 za, zb;
    GSP(za, zb, k0, k3, k0, k3, false);
    st2(ws, 0, za); st2(ws, 6, zb);
  }
  {
    Fp2 k1 = ld2(ws, 2), k2 = ld2(ws, 4), k4 = ld2(ws, 8), k5 = ld2(ws, 10);
    Fp2 z2, z5, z4, z1;
    GSP(z2, z5, k1, k4, k2, k5, false);
    GSP(z4, z1, k2, k5, k4, k1, true);
    st2(ws, 4, z2); st2(ws, 10, z5); st2(ws, 8, z4); st2(ws, 2, z1);
  }
}
#elif defined(V_SUBG_REG) || defined(V_SUBG_LD)
struct WsQ { const Ws& w; __device__ __forceinline__ G2Aff operator()() const { G2Aff q; q.x = ld2(w, 0); q.y = ld2(w, 2); return q; } };
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  WS_SETUP
#ifdef V_SUBG_REG
  G2Aff q; q.x = ld2(ws, 0); q.y = ld2(ws, 2);
  bool ok = g2_in_subgroup(q);
#else
  bool ok = g2_in_subgroup_ld(WsQ{ws});
#endif
  base[(size_t)n * 9 * 30 + blockIdx.x * 256 + threadIdx.x] = ok;
}
#elif defined(V_MUL_OLD) || defined(V_MUL_K) || defined(V_MUL_L)
__device__ __forceinline__ Fp6 ldh(const Ws& w, int e, int h) { Fp6 r; r.c0 = ld2(w, e + 2 * h); r.c1 = ld2(w, e + 4 + 2 * h); r.c2 = ld2(w, e + 8 + 2 * h); return r; }
__device__ __forceinline__ void sth(const Ws& w, int e, int h, const Fp6& a) { st2(w, e + 2 * h, a.c0); st2(w, e + 4 + 2 * h, a.c1); st2(w, e + 8 + 2 * h, a.c2); }
__device__ __forceinline__ Fp6 ld6(const Ws& w, int e) { Fp6 r; r.c0 = ld2(w, e); r.c1 = ld2(w, e + 2); r.c2 = ld2(w, e + 4); return r; }
__device__ __forceinline__ void st6(const Ws& w, int e, const Fp6& a) { st2(w, e, a.c0); st2(w, e + 2, a.c1); st2(w, e + 4, a.c2); }
#if defined(V_MUL_K) || defined(V_MUL_L)
__device__ __forceinline__ Fp6 fp6_mul_k(const Fp6& x, const Fp6& y) {
  Fp2 Y1 = fp2_mul_xi(y.c1), Y2 = fp2_mul_xi(y.c2);
  Fp6 r;
  r.c0 = fp2_dotk(kp(x.c0, y.c0), kp(x.c1, Y2), kp(x.c2, Y1));
  r.c1 = fp2_dotk(kp(x.c0, y.c1), kp(x.c1, y.c0), kp(x.c2, Y2));
  r.c2 = fp2_dotk(kp(x.c0, y.c2), kp(x.c1, y.c1), kp(x.c2, y.c0));
  return r;
}
#define F6MUL fp6_mul_k
#else
#define F6MUL fp6_mul
#endif
#ifdef V_MUL_L
__device__ __forceinline__ void lput(int32_t* lds, int slot, const Fp2& a) {
#pragma unroll
  for (int l = 0; l < 9; l++) { lds[(slot * 18 + l) * 256 + threadIdx.x] = a.c0.v[l]; lds[(slot * 18 + 9 + l) * 256 + threadIdx.x] = a.c1.v[l]; }
}
__device__ __forceinline__ Fp2 lget(const int32_t* lds, int slot) { Fp2 r;
#pragma unroll
  for (int l = 0; l < 9; l++) { r.c0.v[l] = lds[(slot * 18 + l) * 256 + threadIdx.x]; r.c1.v[l] = lds[(slot * 18 + 9 + l) * 256 + threadIdx.x]; }
  return r; }
#endif
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
#ifdef V_MUL_L
  __shared__ int32_t lds[54 * 256];
#endif
  WS_SETUP
  const int e_a = 0, e_b = 12, e_dst = 0, TA = 24, TB = 30;
#ifdef V_MUL_L
  { Fp6 v0 = F6MUL(ldh(ws, e_a, 0), ldh(ws, e_b, 0)); lput(lds, 0, v0.c0); lput(lds, 1, v0.c1); lput(lds, 2, v0.c2); }
#else
  { Fp6 v0 = F6MUL(ldh(ws, e_a, 0), ldh(ws, e_b, 0)); st6(ws, TA, v0); }
#endif
  { Fp6 v1 = F6MUL(ldh(ws, e_a, 1), ldh(ws, e_b, 1)); st6(ws, TB, v1); }
  Fp6 s;
  {
    Fp6 sa = fp6_add(ldh(ws, e_a, 0), ldh(ws, e_a, 1));
    Fp6 sb = fp6_add(ldh(ws, e_b, 0), ldh(ws, e_b, 1));
    s = F6MUL(sa, sb);
  }
#ifdef V_MUL_L
  Fp6 v0; v0.c0 = lget(lds, 0); v0.c1 = lget(lds, 1); v0.c2 = lget(lds, 2);
  Fp6 v1 = ld6(ws, TB);
#else
  Fp6 v0 = ld6(ws, TA), v1 = ld6(ws, TB);
#endif
  Fp6 c0, c1;
  c0.c0 = fp2_add(v0.c0, fp2_mul_xi(v1.c2)); c0.c1 = fp2_add(v0.c1, v1.c0); c0.c2 = fp2_add(v0.c2, v1.c1);
  c1.c0 = fp2_sub2(s.c0, v0.c0, v1.c0); c1.c1 = fp2_sub2(s.c1, v0.c1, v1.c1); c1.c2 = fp2_sub2(s.c2, v0.c2, v1.c2);
  sth(ws, e_dst, 0, c0); sth(ws, e_dst, 1, c1);
}
#elif defined(V_MUL_T3)
// the "third form" of the general Fp12 product.  Karatsuba over Fp6 as in the product's k_f12_mul (three Fp6 products, each nine Fp2
// products in three sums: the same multiply-adds), but a0 and a1 stay in registers from their ONE load, v0 = a0 b0 waits in LDS (54 KB per block), v1 = a1 b1 in
// registers, and only b0 is read a second time (for b0 + b1): 432 (a) + 432 (b) + 216 (b0 again) bytes read + 432 written = 1512 B per proof against the 2451 B the
// counters show for k_f12_mul (which re-reads a and b for the sums and passes v1 through the workspace).
__device__ __forceinline__ Fp6 ldh(const Ws& w, int e, int h) { Fp6 r; r.c0 = ld2(w, e + 2 * h); r.c1 = ld2(w, e + 4 + 2 * h); r.c2 = ld2(w, e + 8 + 2 * h); return r; }
__device__ __forceinline__ void sth(const Ws& w, int e, int h, const Fp6& a) { st2(w, e + 2 * h, a.c0); st2(w, e + 4 + 2 * h, a.c1); st2(w, e + 8 + 2 * h, a.c2); }
__device__ __forceinline__ Fp6 fp6_mul_k(const Fp6& x, const Fp6& y) {
  Fp2 Y1 = fp2_mul_xi(y.c1), Y2 = fp2_mul_xi(y.c2);
  Fp6 r;
  r.c0 = fp2_dotk(kp(x.c0, y.c0), kp(x.c1, Y2), kp(x.c2, Y1));
  r.c1 = fp2_dotk(kp(x.c0, y.c1), kp(x.c1, y.c0), kp(x.c2, Y2));
  r.c2 = fp2_dotk(kp(x.c0, y.c2), kp(x.c1, y.c1), kp(x.c2, y.c0));
  return r;
}
__device__ __forceinline__ void lput(int32_t* lds, int slot, const Fp2& a) {
#pragma unroll
  for (int l = 0; l < 9; l++) { lds[(slot * 18 + l) * 256 + threadIdx.x] = a.c0.v[l]; lds[(slot * 18 + 9 + l) * 256 + threadIdx.x] = a.c1.v[l]; }
}
__device__ __forceinline__ Fp2 lget(const int32_t* lds, int slot) { Fp2 r;
#pragma unroll
  for (int l = 0; l < 9; l++) { r.c0.v[l] = lds[(slot * 18 + l) * 256 + threadIdx.x]; r.c1.v[l] = lds[(slot * 18 + 9 + l) * 256 + threadIdx.x]; }
  return r; }
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  __shared__ int32_t lds[54 * 256];
  WS_SETUP
  const int e_a = 0, e_b = 12, e_dst = 0;
  Fp6 a0 = ldh(ws, e_a, 0), a1 = ldh(ws, e_a, 1);
  { Fp6 v0 = fp6_mul_k(a0, ldh(ws, e_b, 0)); lput(lds, 0, v0.c0); lput(lds, 1, v0.c1); lput(lds, 2, v0.c2); }
  BN_SCHED_FENCE();
  Fp6 b1 = ldh(ws, e_b, 1);
  const Fp6 v1 = fp6_mul_k(a1, b1);
  BN_SCHED_FENCE();
  Fp6 s;
  {
    const Fp6 sa = fp6_add(a0, a1);
    const Fp6 sb = fp6_add(ldh(ws, e_b, 0), b1);      // b0: the one second read
    s = fp6_mul_k(sa, sb);
  }
  BN_SCHED_FENCE();
  Fp6 v0; v0.c0 = lget(lds, 0); v0.c1 = lget(lds, 1); v0.c2 = lget(lds, 2);
  Fp6 c0, c1;
  c0.c0 = fp2_add(v0.c0, fp2_mul_xi(v1.c2)); c0.c1 = fp2_add(v0.c1, v1.c0); c0.c2 = fp2_add(v0.c2, v1.c1);
  c1.c0 = fp2_sub2(s.c0, v0.c0, v1.c0); c1.c1 = fp2_sub2(s.c1, v0.c1, v1.c1); c1.c2 = fp2_sub2(s.c2, v0.c2, v1.c2);
  sth(ws, e_dst, 0, c0); sth(ws, e_dst, 1, c1);
}
#elif defined(V_MUL_T3D)
// ... ordered so that v0, v1 and s are never live together: v0 -> LDS; v1; c0 = v0 + xi-shift(v1) leaves at once, t = v0 + v1 replaces v0 in LDS; then s, c1 = s - t.
// Same 1512 B, the register peak is a0, a1, one half of b and the product in flight (what k_f12_mul's first Fp6 product already holds).
__device__ __forceinline__ Fp6 ldh(const Ws& w, int e, int h) { Fp6 r; r.c0 = ld2(w, e + 2 * h); r.c1 = ld2(w, e + 4 + 2 * h); r.c2 = ld2(w, e + 8 + 2 * h); return r; }
__device__ __forceinline__ void sth(const Ws& w, int e, int h, const Fp6& a) { st2(w, e + 2 * h, a.c0); st2(w, e + 4 + 2 * h, a.c1); st2(w, e + 8 + 2 * h, a.c2); }
__device__ __forceinline__ Fp6 fp6_mul_k(const Fp6& x, const Fp6& y) {
  Fp2 Y1 = fp2_mul_xi(y.c1), Y2 = fp2_mul_xi(y.c2);
  Fp6 r;
  r.c0 = fp2_dotk(kp(x.c0, y.c0), kp(x.c1, Y2), kp(x.c2, Y1));
  r.c1 = fp2_dotk(kp(x.c0, y.c1), kp(x.c1, y.c0), kp(x.c2, Y2));
  r.c2 = fp2_dotk(kp(x.c0, y.c2), kp(x.c1, y.c1), kp(x.c2, y.c0));
  return r;
}
__device__ __forceinline__ void lput(int32_t* lds, int slot, const Fp2& a) {
#pragma unroll
  for (int l = 0; l < 9; l++) { lds[(slot * 18 + l) * 256 + threadIdx.x] = a.c0.v[l]; lds[(slot * 18 + 9 + l) * 256 + threadIdx.x] = a.c1.v[l]; }
}
__device__ __forceinline__ Fp2 lget(const int32_t* lds, int slot) { Fp2 r;
#pragma unroll
  for (int l = 0; l < 9; l++) { r.c0.v[l] = lds[(slot * 18 + l) * 256 + threadIdx.x]; r.c1.v[l] = lds[(slot * 18 + 9 + l) * 256 + threadIdx.x]; }
  return r; }
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  __shared__ int32_t lds[54 * 256];
  WS_SETUP
  const int e_a = 0, e_b = 12, e_dst = 24;
  Fp6 a0 = ldh(ws, e_a, 0), a1 = ldh(ws, e_a, 1);
  { Fp6 v0 = fp6_mul_k(a0, ldh(ws, e_b, 0)); lput(lds, 0, v0.c0); lput(lds, 1, v0.c1); lput(lds, 2, v0.c2); }
  BN_SCHED_FENCE();
  Fp6 b1 = ldh(ws, e_b, 1);
  {
    const Fp6 v1 = fp6_mul_k(a1, b1);
    BN_SCHED_FENCE();
    Fp6 v0; v0.c0 = lget(lds, 0); v0.c1 = lget(lds, 1); v0.c2 = lget(lds, 2);
    Fp6 c0;
    c0.c0 = fp2_add(v0.c0, fp2_mul_xi(v1.c2)); c0.c1 = fp2_add(v0.c1, v1.c0); c0.c2 = fp2_add(v0.c2, v1.c1);
    sth(ws, e_dst, 0, c0);
    lput(lds, 0, fp2_add(v0.c0, v1.c0)); lput(lds, 1, fp2_add(v0.c1, v1.c1)); lput(lds, 2, fp2_add(v0.c2, v1.c2));
  }
  BN_SCHED_FENCE();
  Fp6 s;
  {
    a0 = fp6_add(a0, a1);
    b1 = fp6_add(ldh(ws, e_b, 0), b1);      // b0: the one second read
    s = fp6_mul_k(a0, b1);
  }
  BN_SCHED_FENCE();
  Fp6 c1;
  c1.c0 = fp2_sub(s.c0, lget(lds, 0)); c1.c1 = fp2_sub(s.c1, lget(lds, 1)); c1.c2 = fp2_sub(s.c2, lget(lds, 2));
  sth(ws, e_dst, 1, c1);
}
#elif defined(V_MUL_T3E)
// ... T3D with a0 read again for the sum as well (1728 B): the register peak of every Fp6 product is its two operands, as in k_f12_mul -- no spill.
__device__ __forceinline__ Fp6 ldh(const Ws& w, int e, int h) { Fp6 r; r.c0 = ld2(w, e + 2 * h); r.c1 = ld2(w, e + 4 + 2 * h); r.c2 = ld2(w, e + 8 + 2 * h); return r; }
__device__ __forceinline__ void sth(const Ws& w, int e, int h, const Fp6& a) { st2(w, e + 2 * h, a.c0); st2(w, e + 4 + 2 * h, a.c1); st2(w, e + 8 + 2 * h, a.c2); }
__device__ __forceinline__ Fp6 fp6_mul_k(const Fp6& x, const Fp6& y) {
  Fp2 Y1 = fp2_mul_xi(y.c1), Y2 = fp2_mul_xi(y.c2);
  Fp6 r;
  r.c0 = fp2_dotk(kp(x.c0, y.c0), kp(x.c1, Y2), kp(x.c2, Y1));
  r.c1 = fp2_dotk(kp(x.c0, y.c1), kp(x.c1, y.c0), kp(x.c2, Y2));
  r.c2 = fp2_dotk(kp(x.c0, y.c2), kp(x.c1, y.c1), kp(x.c2, y.c0));
  return r;
}
__device__ __forceinline__ void lput(int32_t* lds, int slot, const Fp2& a) {
#pragma unroll
  for (int l = 0; l < 9; l++) { lds[(slot * 18 + l) * 256 + threadIdx.x] = a.c0.v[l]; lds[(slot * 18 + 9 + l) * 256 + threadIdx.x] = a.c1.v[l]; }
}
__device__ __forceinline__ Fp2 lget(const int32_t* lds, int slot) { Fp2 r;
#pragma unroll
  for (int l = 0; l < 9; l++) { r.c0.v[l] = lds[(slot * 18 + l) * 256 + threadIdx.x]; r.c1.v[l] = lds[(slot * 18 + 9 + l) * 256 + threadIdx.x]; }
  return r; }
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  __shared__ int32_t lds[54 * 256];
  WS_SETUP
  const int e_a = 0, e_b = 12, e_dst = 24;
  { Fp6 v0 = fp6_mul_k(ldh(ws, e_a, 0), ldh(ws, e_b, 0)); lput(lds, 0, v0.c0); lput(lds, 1, v0.c1); lput(lds, 2, v0.c2); }
  BN_SCHED_FENCE();
  Fp6 a1 = ldh(ws, e_a, 1), b1 = ldh(ws, e_b, 1);
  {
    const Fp6 v1 = fp6_mul_k(a1, b1);
    BN_SCHED_FENCE();
    Fp6 v0; v0.c0 = lget(lds, 0); v0.c1 = lget(lds, 1); v0.c2 = lget(lds, 2);
    Fp6 c0;
    c0.c0 = fp2_add(v0.c0, fp2_mul_xi(v1.c2)); c0.c1 = fp2_add(v0.c1, v1.c0); c0.c2 = fp2_add(v0.c2, v1.c1);
    sth(ws, e_dst, 0, c0);
    lput(lds, 0, fp2_add(v0.c0, v1.c0)); lput(lds, 1, fp2_add(v0.c1, v1.c1)); lput(lds, 2, fp2_add(v0.c2, v1.c2));
  }
  BN_SCHED_FENCE();
  Fp6 s;
  {
    a1 = fp6_add(ldh(ws, e_a, 0), a1);      // a0, b0: read a second time
    b1 = fp6_add(ldh(ws, e_b, 0), b1);
    s = fp6_mul_k(a1, b1);
  }
  BN_SCHED_FENCE();
  Fp6 c1;
  c1.c0 = fp2_sub(s.c0, lget(lds, 0)); c1.c1 = fp2_sub(s.c1, lget(lds, 1)); c1.c2 = fp2_sub(s.c2, lget(lds, 2));
  sth(ws, e_dst, 1, c1);
}
#elif defined(V_MUL_T3F)
// ... T3E that tolerates dst == a or dst == b (k_f12_mul's callers alias): c0 is held back in registers until a0 and b0 have been read the second time.
__device__ __forceinline__ Fp6 ldh(const Ws& w, int e, int h) { Fp6 r; r.c0 = ld2(w, e + 2 * h); r.c1 = ld2(w, e + 4 + 2 * h); r.c2 = ld2(w, e + 8 + 2 * h); return r; }
__device__ __forceinline__ void sth(const Ws& w, int e, int h, const Fp6& a) { st2(w, e + 2 * h, a.c0); st2(w, e + 4 + 2 * h, a.c1); st2(w, e + 8 + 2 * h, a.c2); }
__device__ __forceinline__ Fp6 fp6_mul_k(const Fp6& x, const Fp6& y) {
  Fp2 Y1 = fp2_mul_xi(y.c1), Y2 = fp2_mul_xi(y.c2);
  Fp6 r;
  r.c0 = fp2_dotk(kp(x.c0, y.c0), kp(x.c1, Y2), kp(x.c2, Y1));
  r.c1 = fp2_dotk(kp(x.c0, y.c1), kp(x.c1, y.c0), kp(x.c2, Y2));
  r.c2 = fp2_dotk(kp(x.c0, y.c2), kp(x.c1, y.c1), kp(x.c2, y.c0));
  return r;
}
__device__ __forceinline__ void lput(int32_t* lds, int slot, const Fp2& a) {
#pragma unroll
  for (int l = 0; l < 9; l++) { lds[(slot * 18 + l) * 256 + threadIdx.x] = a.c0.v[l]; lds[(slot * 18 + 9 + l) * 256 + threadIdx.x] = a.c1.v[l]; }
}
__device__ __forceinline__ Fp2 lget(const int32_t* lds, int slot) { Fp2 r;
#pragma unroll
  for (int l = 0; l < 9; l++) { r.c0.v[l] = lds[(slot * 18 + l) * 256 + threadIdx.x]; r.c1.v[l] = lds[(slot * 18 + 9 + l) * 256 + threadIdx.x]; }
  return r; }
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  __shared__ int32_t lds[54 * 256];
  WS_SETUP
  const int e_a = 0, e_b = 12, e_dst = 0;      // dst aliases a, as in exp-by-u
  { Fp6 v0 = fp6_mul_k(ldh(ws, e_a, 0), ldh(ws, e_b, 0)); lput(lds, 0, v0.c0); lput(lds, 1, v0.c1); lput(lds, 2, v0.c2); }
  BN_SCHED_FENCE();
  Fp6 a1 = ldh(ws, e_a, 1), b1 = ldh(ws, e_b, 1);
  Fp6 c0;
  {
    const Fp6 v1 = fp6_mul_k(a1, b1);
    BN_SCHED_FENCE();
    // one coefficient of v0 at a time: c0 = v0 + (xi v1.c2, v1.c0, v1.c1), t = v0 + v1 back into the same LDS slots
    { const Fp2 x = lget(lds, 0); c0.c0 = fp2_add(x, fp2_mul_xi(v1.c2)); lput(lds, 0, fp2_add(x, v1.c0)); }
    { const Fp2 x = lget(lds, 1); c0.c1 = fp2_add(x, v1.c0); lput(lds, 1, fp2_add(x, v1.c1)); }
    { const Fp2 x = lget(lds, 2); c0.c2 = fp2_add(x, v1.c1); lput(lds, 2, fp2_add(x, v1.c2)); }
  }
  BN_SCHED_FENCE();
  // a0, b0 a second time, straight into the sums; only then may dst (= a or b) be written
  a1 = fp6_add(ldh(ws, e_a, 0), a1);
  b1 = fp6_add(ldh(ws, e_b, 0), b1);
  sth(ws, e_dst, 0, c0);
  BN_SCHED_FENCE();
  const Fp6 s = fp6_mul_k(a1, b1);
  BN_SCHED_FENCE();
  Fp6 c1;
  c1.c0 = fp2_sub(s.c0, lget(lds, 0)); c1.c1 = fp2_sub(s.c1, lget(lds, 1)); c1.c2 = fp2_sub(s.c2, lget(lds, 2));
  sth(ws, e_dst, 1, c1);
}
#elif defined(V_MUL_T3B)
// ... the same with v1 in LDS as well and blocks of 128 lanes (2 x 27 KB per block): nothing but operands and the product in flight lives in registers
__device__ __forceinline__ Fp6 ldh(const Ws& w, int e, int h) { Fp6 r; r.c0 = ld2(w, e + 2 * h); r.c1 = ld2(w, e + 4 + 2 * h); r.c2 = ld2(w, e + 8 + 2 * h); return r; }
__device__ __forceinline__ void sth(const Ws& w, int e, int h, const Fp6& a) { st2(w, e + 2 * h, a.c0); st2(w, e + 4 + 2 * h, a.c1); st2(w, e + 8 + 2 * h, a.c2); }
__device__ __forceinline__ Fp6 fp6_mul_k(const Fp6& x, const Fp6& y) {
  Fp2 Y1 = fp2_mul_xi(y.c1), Y2 = fp2_mul_xi(y.c2);
  Fp6 r;
  r.c0 = fp2_dotk(kp(x.c0, y.c0), kp(x.c1, Y2), kp(x.c2, Y1));
  r.c1 = fp2_dotk(kp(x.c0, y.c1), kp(x.c1, y.c0), kp(x.c2, Y2));
  r.c2 = fp2_dotk(kp(x.c0, y.c2), kp(x.c1, y.c1), kp(x.c2, y.c0));
  return r;
}
__device__ __forceinline__ void lput(int32_t* lds, int slot, const Fp2& a) {
#pragma unroll
  for (int l = 0; l < 9; l++) { lds[(slot * 18 + l) * 256 + threadIdx.x] = a.c0.v[l]; lds[(slot * 18 + 9 + l) * 256 + threadIdx.x] = a.c1.v[l]; }
}
__device__ __forceinline__ Fp2 lget(const int32_t* lds, int slot) { Fp2 r;
#pragma unroll
  for (int l = 0; l < 9; l++) { r.c0.v[l] = lds[(slot * 18 + l) * 256 + threadIdx.x]; r.c1.v[l] = lds[(slot * 18 + 9 + l) * 256 + threadIdx.x]; }
  return r; }
__global__ void __launch_bounds__(256, 1) k_op(int32_t* base, uint32_t n) {
  __shared__ int32_t lds[108 * 256];
  WS_SETUP
  const int e_a = 0, e_b = 12, e_dst = 0;
  Fp6 a0 = ldh(ws, e_a, 0), a1 = ldh(ws, e_a, 1);
  Fp6 b0 = ldh(ws, e_b, 0), b1 = ldh(ws, e_b, 1);
  { Fp6 v0 = fp6_mul_k(a0, b0); lput(lds, 0, v0.c0); lput(lds, 1, v0.c1); lput(lds, 2, v0.c2); }
  BN_SCHED_FENCE();
  { Fp6 v1 = fp6_mul_k(a1, b1); lput(lds, 3, v1.c0); lput(lds, 4, v1.c1); lput(lds, 5, v1.c2); }
  BN_SCHED_FENCE();
  Fp6 s = fp6_mul_k(fp6_add(a0, a1), fp6_add(b0, b1));
  BN_SCHED_FENCE();
  Fp6 v0; v0.c0 = lget(lds, 0); v0.c1 = lget(lds, 1); v0.c2 = lget(lds, 2);
  Fp6 v1; v1.c0 = lget(lds, 3); v1.c1 = lget(lds, 4); v1.c2 = lget(lds, 5);
  Fp6 c0, c1;
  c0.c0 = fp2_add(v0.c0, fp2_mul_xi(v1.c2)); c0.c1 = fp2_add(v0.c1, v1.c0); c0.c2 = fp2_add(v0.c2, v1.c1);
  c1.c0 = fp2_sub2(s.c0, v0.c0, v1.c0); c1.c1 = fp2_sub2(s.c1, v0.c1, v1.c1); c1.c2 = fp2_sub2(s.c2, v0.c2, v1.c2);
  sth(ws, e_dst, 0, c0); sth(ws, e_dst, 1, c1);
}
#elif defined(V_MUL_STREAM)
// the STREAMING SCHOOLBOOK form of the general Fp12 product.  a is loaded once and stays in registers (108); for every output coefficient
// r_j = sum_t (xi if t > j) a_t b_((j - t) mod 6) the six coefficients of b are read again (from the caches: 6 x 432 B per lane) and the six Fp2 products are ONE sum
// of products (two 12-term fp_dot: 24 x 81 + 2 x 81 = 2106 multiply-adds per coefficient, 12.6 k per product against 8.3 k of the Karatsuba form); no Fp6-sized
// temporary goes through the workspace: operands read once from HBM (864 B) + result (432 B) = 1296 B against the 2451 B the counters show for k_f12_mul.
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  WS_SETUP
  const int e_a = 0, e_b = 12, e_dst = 24;
  const Fp2 a0 = ld2(ws, e_a), a1 = ld2(ws, e_a + 2), a2 = ld2(ws, e_a + 4), a3 = ld2(ws, e_a + 6), a4 = ld2(ws, e_a + 8), a5 = ld2(ws, e_a + 10);
#pragma unroll
  for (int j = 0; j < 6; j++) {
    // b_((j - t) mod 6) for t = 0..5, xi-multiplied where t > j
    Fp2 b0 = ld2(ws, e_b + 2 * ((j + 6 - 0) % 6)), b1 = ld2(ws, e_b + 2 * ((j + 6 - 1) % 6)), b2 = ld2(ws, e_b + 2 * ((j + 6 - 2) % 6));
    Fp2 b3 = ld2(ws, e_b + 2 * ((j + 6 - 3) % 6)), b4 = ld2(ws, e_b + 2 * ((j + 6 - 4) % 6)), b5 = ld2(ws, e_b + 2 * ((j + 6 - 5) % 6));
    if (1 > j) b1 = fp2_mul_xi(b1);
    if (2 > j) b2 = fp2_mul_xi(b2);
    if (3 > j) b3 = fp2_mul_xi(b3);
    if (4 > j) b4 = fp2_mul_xi(b4);
    if (5 > j) b5 = fp2_mul_xi(b5);
    st2(ws, e_dst + 2 * j, fp2_dotp(pp(a0, b0), pp(a1, b1), pp(a2, b2), pp(a3, b3), pp(a4, b4), pp(a5, b5)));
    BN_SCHED_FENCE();
  }
}
#elif defined(V_MUL_STREAM2)
// ... the same with every output coefficient as TWO sums of three products (the twelve-term form keeps 216 operand registers live and spills 1252 bytes per lane):
// a (108 registers) + three coefficients of b at a time; one more reduction per coefficient (13.6 k multiply-adds per product)
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  WS_SETUP
  const int e_a = 0, e_b = 12, e_dst = 24;
  const Fp2 a0 = ld2(ws, e_a), a1 = ld2(ws, e_a + 2), a2 = ld2(ws, e_a + 4), a3 = ld2(ws, e_a + 6), a4 = ld2(ws, e_a + 8), a5 = ld2(ws, e_a + 10);
#pragma unroll
  for (int j = 0; j < 6; j++) {
    Fp2 lo;
    {
      Fp2 b0 = ld2(ws, e_b + 2 * ((j + 6 - 0) % 6)), b1 = ld2(ws, e_b + 2 * ((j + 6 - 1) % 6)), b2 = ld2(ws, e_b + 2 * ((j + 6 - 2) % 6));
      if (1 > j) b1 = fp2_mul_xi(b1);
      if (2 > j) b2 = fp2_mul_xi(b2);
      lo = fp2_dotp(pp(a0, b0), pp(a1, b1), pp(a2, b2));
    }
    BN_SCHED_FENCE();
    {
      Fp2 b3 = ld2(ws, e_b + 2 * ((j + 6 - 3) % 6)), b4 = ld2(ws, e_b + 2 * ((j + 6 - 4) % 6)), b5 = ld2(ws, e_b + 2 * ((j + 6 - 5) % 6));
      if (3 > j) b3 = fp2_mul_xi(b3);
      if (4 > j) b4 = fp2_mul_xi(b4);
      if (5 > j) b5 = fp2_mul_xi(b5);
      st2(ws, e_dst + 2 * j, fp2_add(lo, fp2_dotp(pp(a3, b3), pp(a4, b4), pp(a5, b5))));
    }
    BN_SCHED_FENCE();
  }
}
#elif defined(V_MUL_STREAM3)
// ... and as THREE sums of two products (a + two coefficients of b at a time; 14.6 k multiply-adds per product)
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  WS_SETUP
  const int e_a = 0, e_b = 12, e_dst = 24;
  const Fp2 a0 = ld2(ws, e_a), a1 = ld2(ws, e_a + 2), a2 = ld2(ws, e_a + 4), a3 = ld2(ws, e_a + 6), a4 = ld2(ws, e_a + 8), a5 = ld2(ws, e_a + 10);
#pragma unroll
  for (int j = 0; j < 6; j++) {
    Fp2 acc;
    { Fp2 b0 = ld2(ws, e_b + 2 * ((j + 6 - 0) % 6)), b1 = ld2(ws, e_b + 2 * ((j + 6 - 1) % 6)); if (1 > j) b1 = fp2_mul_xi(b1); acc = fp2_dotp(pp(a0, b0), pp(a1, b1)); }
    BN_SCHED_FENCE();
    { Fp2 b2 = ld2(ws, e_b + 2 * ((j + 6 - 2) % 6)), b3 = ld2(ws, e_b + 2 * ((j + 6 - 3) % 6)); if (2 > j) b2 = fp2_mul_xi(b2); if (3 > j) b3 = fp2_mul_xi(b3); acc = fp2_add(acc, fp2_dotp(pp(a2, b2), pp(a3, b3))); }
    BN_SCHED_FENCE();
    { Fp2 b4 = ld2(ws, e_b + 2 * ((j + 6 - 4) % 6)), b5 = ld2(ws, e_b + 2 * ((j + 6 - 5) % 6)); if (4 > j) b4 = fp2_mul_xi(b4); if (5 > j) b5 = fp2_mul_xi(b5); st2(ws, e_dst + 2 * j, fp2_add(acc, fp2_dotp(pp(a4, b4), pp(a5, b5)))); }
    BN_SCHED_FENCE();
  }
}
#elif defined(V_COPY)
__global__ void __launch_bounds__(256, 2) k_op(int32_t* base, uint32_t n) {
  WS_SETUP
  Fp2 k0 = ld2(ws, 0), k1 = ld2(ws, 2), k2 = ld2(ws, 4), k3 = ld2(ws, 6), k4 = ld2(ws, 8), k5 = ld2(ws, 10);
  st2(ws, 0, k1); st2(ws, 2, k2); st2(ws, 4, k3); st2(ws, 6, k4); st2(ws, 8, k5); st2(ws, 10, k0);
}
#endif
int main(int argc, char** argv) {
  uint32_t n = 1u << 20;
  int reps = 20;
  int32_t* base;
  size_t words = (size_t)n * 9 * NELEM;
  if (hipMalloc((void**)&base, words * 4) != hipSuccess) { printf("malloc failed\n"); return 1; }
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, 0, base, n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_op, dim3(n / 256), dim3(256), 0, 0, base, n);
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_op, dim3(n / 256), dim3(256), 0, 0, base, n);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  printf("%s: %.1f us per launch (n=%u) %s\n", VNAME, ms * 1e3 / reps, n, e == hipSuccess ? "" : hipGetErrorString(e));
  return 0;
}
