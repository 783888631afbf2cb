// bn254_coop12.hip -- the cooperative layout for SMALL batches: TWELVE lanes per proof.
//
// A first generation (retired) spread a proof over six lanes, one Fp2 coefficient of every Fp12 value per lane: a batch of
// 4096 proofs was 410 wavefronts on 1024 SIMDs, and the time of the one launch one wavefront's serial instruction stream.  Here every Fp2 coefficient
// k_c = re + im i is split once more: lane (c, h) of a proof keeps ONE Fp number, h = 0: re, h = 1: im.  A product of two coefficients is
//     (a b)_h = a.re * b_h + a.im * (i b)_h          with  i b = (-b.im, b.re)
// so a sum of n coefficient products is one Fp dot product of 2 n terms per lane (bn254_fp.h::fp_dot: 162 n + 81 multiply-adds against
// 243 n + 162 of the Karatsuba form a six-lane layout runs), and the first operand is fetched whole while the second is fetched as the pair
// (b_h, (i b)_h): from the value itself and from an "i-image" the operation publishes first.  No exchange of partial products is needed.
// Five proofs per wavefront (lanes 60..63 idle): 4096 proofs are 820 wavefronts -- still one per SIMD -- with about 0.62 of the multiply-adds.
//
// LDS image per wavefront: img[slot][lane][12 dwords] (one Fp = 9 digits + 3 pad: three ds_read_b128; 12 * lane mod 64 puts 16 consecutive lanes
// on disjoint 4-bank groups), 13 slots = 39 KB, so FOUR wavefronts share a CU's 160 KB (the six-lane image was 66.5 KB: two).  Slots hold Fp12
// VALUES (slot numbers = the workspace map's, the final exponentiation is the same program template as everywhere else); "half h of coefficient i of value a"
// is a read at (a, group base + 2 i + h).
//
// Lock-step: a wavefront executes its LDS instructions in order, every operation reads all of its inputs before it writes
// its output slot.
#include <hip/hip_runtime.h>
#include <mutex>
#include "bn254_vm.h"
#include "bn254_kernels.h"

namespace bn254 {

static_assert(COOP_T_ELEM == VE_S2, "COOP_T_ELEM must name a workspace slot that neither VE_T nor the result slot VE_S0 overlays");
#define C12_STRIDE 12
#define C12_SLOT(e) (((e) - VE_F) / 12)   // VE_F 0, VE_S0 1, S1 2, S2 3, S3 4, S4 5, P3 6, (7: scratch), P5 8, P7 9
#define C12_X 10     // xi-multiples of an operand
#define C12_I 11     // i-multiples of an operand
#define C12_B 7      // conj(b) of a general product
#define C12_R1 7     // G2 step: products of a round; public-input MSM: the partial sums (X, Y, Z in R1, R2, R3)
#define C12_R2 11
#define C12_R3 12
#define C12_SLOTS 13
#define C12_WAVE_DWORDS (C12_SLOTS * 64 * C12_STRIDE)
#define C12_PER_WAVE 5

typedef __attribute__((address_space(3))) int32_t c12_lds_i32;   // LDS pointers keep their address space through the out-of-line operations (ds_* instead of flat_*)
typedef int c12_v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) c12_v4i c12_lds_i4;
struct Coop12 {
  c12_lds_i32* img;          // this wavefront's LDS image
  uint32_t lane, g0, c, h;   // lane in the wavefront, first lane of the proof's group (12 lanes), coefficient index, half (lanes 60..63: copies of group 4)
  __device__ __forceinline__ void put(int slot, const Fp& a) const {
    c12_lds_i4* q = (c12_lds_i4*)(img + ((size_t)slot * 64 + lane) * C12_STRIDE);
    q[0] = (c12_v4i){a.v[0], a.v[1], a.v[2], a.v[3]}; q[1] = (c12_v4i){a.v[4], a.v[5], a.v[6], a.v[7]}; q[2] = (c12_v4i){a.v[8], 0, 0, 0};
  }
  __device__ __forceinline__ Fp at(int slot, uint32_t ln) const {
    const c12_lds_i4* q = (const c12_lds_i4*)(img + ((size_t)slot * 64 + ln) * C12_STRIDE);
    const c12_v4i v0 = q[0], v1 = q[1], v2 = q[2];
    Fp a;
    a.v[0] = v0.x; a.v[1] = v0.y; a.v[2] = v0.z; a.v[3] = v0.w; a.v[4] = v1.x; a.v[5] = v1.y; a.v[6] = v1.z; a.v[7] = v1.w; a.v[8] = v2.x;
    return a;
  }
  __device__ __forceinline__ Fp own(int slot) const { return at(slot, lane); }
  __device__ __forceinline__ Fp half(int slot, uint32_t i) const { return at(slot, g0 + 2 * i + h); }            // my half of coefficient i
  __device__ __forceinline__ Fp2 coef(int slot, uint32_t i) const { Fp2 r; r.c0 = at(slot, g0 + 2 * i); r.c1 = at(slot, g0 + 2 * i + 1); return r; }
};

__device__ __forceinline__ Fp2 c12_scale(const Fp2& a, int32_t w) {  // w in {0, 1, 2}, digit-wise
  Fp2 r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) { r.c0.v[i] = a.c0.v[i] * w; r.c1.v[i] = a.c1.v[i] * w; }
  return r;
}
// the pair (b_h, (i b)_h) of a value every lane holds whole
struct C12H { Fp p, q; };
__device__ __forceinline__ C12H c12_halves(const Fp2& b, uint32_t h) { C12H r; r.p = fp_select(h != 0, b.c1, b.c0); r.q = fp_select(h != 0, b.c0, fp_neg(b.c1)); return r; }
// publish the images of the value in `slot`: xi * k (my half: 9 own -/+ partner) and / or i * k (my half: -im | re)
__device__ __forceinline__ void c12_publish_xi(const Coop12& co, int slot) {
  const Fp o = co.own(slot), pt = co.at(slot, co.lane ^ 1);
  co.put(C12_X, fp_lincomb_reduce(9, o, co.h ? 1 : -1, pt));
}
__device__ __forceinline__ void c12_publish_i(const Coop12& co, int slot) {
  const Fp pt = co.at(slot, co.lane ^ 1);
  co.put(C12_I, fp_select(co.h != 0, pt, fp_neg(pt)));
}

// ---- f <- f^2 (general squaring): r_j = sum_t w_t P_t Q_t with P from the value or its xi image (whole) and Q plain (halves) -----------------------------
//   r0 = k0 k0 + 2 xk5 k1 + 2 xk4 k2 + xk3 k3     r1 = 2 k1 k0 + 2 xk5 k2 + 2 xk4 k3             r2 = 2 k2 k0 + k1 k1 + 2 xk5 k3 + xk4 k4
//   r3 = 2 k3 k0 + 2 k2 k1 + 2 xk5 k4             r4 = 2 k4 k0 + 2 k3 k1 + k2 k2 + xk5 k5         r5 = 2 k5 k0 + 2 k4 k1 + 2 k3 k2
__constant__ int8_t C12_SQ_Q[6][4] = {{0, 1, 2, 3}, {0, 2, 3, 0}, {0, 1, 3, 4}, {0, 1, 4, 0}, {0, 1, 2, 5}, {0, 1, 2, 0}};
__constant__ int8_t C12_SQ_P[6][4] = {{0, 5, 4, 3}, {1, 5, 4, 0}, {2, 1, 5, 4}, {3, 2, 5, 0}, {4, 3, 2, 5}, {5, 4, 3, 0}};
__constant__ int8_t C12_SQ_X[6][4] = {{0, 1, 1, 1}, {0, 1, 1, 0}, {0, 0, 1, 1}, {0, 0, 1, 0}, {0, 0, 0, 1}, {0, 0, 0, 0}};
__constant__ int8_t C12_SQ_W[6][4] = {{1, 2, 2, 1}, {2, 2, 2, 0}, {2, 1, 2, 1}, {2, 2, 2, 0}, {2, 2, 1, 1}, {2, 2, 2, 0}};
__device__ __forceinline__ void c12_sqr(const Coop12& co, int s) {
  c12_publish_xi(co, s);
  c12_publish_i(co, s);
  const uint32_t c = co.c;
  const Fp2 p0 = c12_scale(co.coef(s, C12_SQ_P[c][0]), C12_SQ_W[c][0]);
  const Fp2 p1 = c12_scale(co.coef(C12_SQ_X[c][1] ? C12_X : s, C12_SQ_P[c][1]), C12_SQ_W[c][1]);
  const Fp2 p2 = c12_scale(co.coef(C12_SQ_X[c][2] ? C12_X : s, C12_SQ_P[c][2]), C12_SQ_W[c][2]);
  const Fp2 p3 = c12_scale(co.coef(C12_SQ_X[c][3] ? C12_X : s, C12_SQ_P[c][3]), C12_SQ_W[c][3]);
  const Fp q0 = co.half(s, C12_SQ_Q[c][0]), q1 = co.half(s, C12_SQ_Q[c][1]), q2 = co.half(s, C12_SQ_Q[c][2]), q3 = co.half(s, C12_SQ_Q[c][3]);
  const Fp j0 = co.half(C12_I, C12_SQ_Q[c][0]), j1 = co.half(C12_I, C12_SQ_Q[c][1]), j2 = co.half(C12_I, C12_SQ_Q[c][2]), j3 = co.half(C12_I, C12_SQ_Q[c][3]);
  co.put(s, fp_dot(dplus(p0.c0, q0), dplus(p0.c1, j0), dplus(p1.c0, q1), dplus(p1.c1, j1), dplus(p2.c0, q2), dplus(p2.c1, j2), dplus(p3.c0, q3), dplus(p3.c1, j3)));
}
// ---- f <- f * (d0 + d3 w + d4 w^3): r_j = k_j d0 + (xi?) k_(j-1) d3 + (xi?) k_(j-3) d4: the k whole from the value or its xi image, the d as halves ---------
// d0 in Fp (the lines of the table-driven pairs, y_P): its term is own * d0.  keep: leave f (line value 1)
__device__ __forceinline__ void c12_mul_line_fp(const Coop12& co, int s, const Fp& d0, const C12H& d3, const C12H& d4, bool keep) {
  c12_publish_xi(co, s);
  const uint32_t c = co.c;
  const Fp k = co.own(s);
  const Fp2 k1 = co.coef(c >= 1 ? s : C12_X, (c + 5) % 6), k3 = co.coef(c >= 3 ? s : C12_X, (c + 3) % 6);
  const Fp r = fp_dot(dplus(k, d0), dplus(k1.c0, d3.p), dplus(k1.c1, d3.q), dplus(k3.c0, d4.p), dplus(k3.c1, d4.q));
  co.put(s, fp_select(keep, k, r));
}
__device__ __forceinline__ void c12_mul_line_fp2(const Coop12& co, int s, const C12H& d0, const C12H& d3, const C12H& d4) {
  c12_publish_xi(co, s);
  const uint32_t c = co.c;
  const Fp2 k0 = co.coef(s, c);
  const Fp2 k1 = co.coef(c >= 1 ? s : C12_X, (c + 5) % 6), k3 = co.coef(c >= 3 ? s : C12_X, (c + 3) % 6);
  co.put(s, fp_dot(dplus(k0.c0, d0.p), dplus(k0.c1, d0.q), dplus(k1.c0, d3.p), dplus(k1.c1, d3.q), dplus(k3.c0, d4.p), dplus(k3.c1, d4.q)));
}
// ---- general product d <- a * (conj?) b:  r_j = sum_t (xi if t > j) a_t b_((j - t) mod 6): xi image of a, i image of (conj?) b --------------------------------
__device__ __noinline__ void c12_mul(const Coop12 co, int d, int a, int b, bool conj_b) {
  const uint32_t c = co.c;
  c12_publish_xi(co, a);
  int bp = b;
  if (conj_b) {   // conjugation negates the odd coefficients: a plain image of conj(b) in the scratch slot
    const Fp o = co.own(b);
    co.put(C12_B, (c & 1) ? fp_neg(o) : o);
    bp = C12_B;
  }
  c12_publish_i(co, bp);
  Fp acc;
  {
    const Fp2 a0 = co.coef(0 <= (int)c ? a : C12_X, 0), a1 = co.coef(1 <= c ? a : C12_X, 1), a2 = co.coef(2 <= c ? a : C12_X, 2);
    const uint32_t i0 = (c + 6 - 0) % 6, i1 = (c + 6 - 1) % 6, i2 = (c + 6 - 2) % 6;
    const Fp q0 = co.half(bp, i0), q1 = co.half(bp, i1), q2 = co.half(bp, i2), j0 = co.half(C12_I, i0), j1 = co.half(C12_I, i1), j2 = co.half(C12_I, i2);
    acc = fp_dot(dplus(a0.c0, q0), dplus(a0.c1, j0), dplus(a1.c0, q1), dplus(a1.c1, j1), dplus(a2.c0, q2), dplus(a2.c1, j2));
  }
  {
    const Fp2 a3 = co.coef(3 <= c ? a : C12_X, 3), a4 = co.coef(4 <= c ? a : C12_X, 4), a5 = co.coef(5 <= c ? a : C12_X, 5);
    const uint32_t i3 = (c + 6 - 3) % 6, i4 = (c + 6 - 4) % 6, i5 = (c + 6 - 5) % 6;
    const Fp q3 = co.half(bp, i3), q4 = co.half(bp, i4), q5 = co.half(bp, i5), j3 = co.half(C12_I, i3), j4 = co.half(C12_I, i4), j5 = co.half(C12_I, i5);
    acc = fp_add(acc, fp_dot(dplus(a3.c0, q3), dplus(a3.c1, j3), dplus(a4.c0, q4), dplus(a4.c1, j4), dplus(a5.c0, q5), dplus(a5.c1, j5)));
  }
  co.put(d, acc);
}
// ---- Granger-Scott squarings, `count` times (pairing of coefficients and roles: bn254_tower.h::fp12_cyclo_sqr) ------------------------------------------------
__constant__ int8_t C12_CY_A[6] = {0, 2, 1, 0, 2, 1};
__constant__ int8_t C12_CY_B[6] = {3, 5, 4, 3, 5, 4};
__device__ __noinline__ void c12_cyclo_sqr_n(const Coop12 co, int d, int s, int count) {
  const uint32_t c = co.c, h = co.h;
  const bool is_s = (c == 0) | (c == 2) | (c == 4);   // lanes that compute S = xi b^2 + a^2; the others T = 2 a b (xi on it for coefficient 1)
  const int32_t lin = is_s ? -2 : 2;
  Fp k = co.own(s);
  for (int it = 0; it < count; it++) {
    const int src = it == 0 ? s : d;
    const Fp2 a = co.coef(src, C12_CY_A[c]), b = co.coef(src, C12_CY_B[c]);
    // X = u1 v1 + u2 v2 with (u1, v1, u2, v2) = (xi b, b, a, a) or (2 a, b or xi b, 0, a)
    const Fp2 xb = fp2_mul_xi(b);
    const Fp2 u1 = fp2_select(is_s, xb, c12_scale(a, 2));
    const Fp2 v1 = fp2_select(is_s | (c == 1), is_s ? b : xb, b);
    const Fp2 u2 = c12_scale(a, is_s ? 1 : 0);
    const C12H h1 = c12_halves(v1, h), h2 = c12_halves(a, h);
    const Fp X = fp_dot(dplus(u1.c0, h1.p), dplus(u1.c1, h1.q), dplus(u2.c0, h2.p), dplus(u2.c1, h2.q));
    const Fp z = fp_lincomb_reduce(3, X, lin, k);
    co.put(d, z);
    k = z;
  }
}
__device__ __noinline__ void c12_conj(const Coop12 co, int d, int s) {
  const Fp k = co.own(s);
  co.put(d, (co.c & 1) ? fp_neg(k) : k);
}
__device__ __noinline__ void c12_frob(const Coop12 co, int d, int s, int j) {
  Fp2 k = co.coef(s, co.c);
  if (j & 1) k = fp2_conj(k);
  const int32_t(*t)[2][BN_NL] = j == 1 ? BN_FROB_G1 : j == 2 ? BN_FROB_G2 : BN_FROB_G3;
  Fp2 g;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { g.c0.v[l] = t[co.c][0][l]; g.c1.v[l] = t[co.c][1][l]; }
  const C12H gh = c12_halves(g, co.h);
  co.put(d, fp_dot(dplus(k.c0, gh.p), dplus(k.c1, gh.q)));   // coefficient 0: g = 1
}
// inverse: every lane gathers the whole value and runs the one-proof-per-lane inversion; lane (c, h) keeps its half of coefficient c
__device__ __noinline__ void c12_inv(const Coop12 co, int d, int s) {
  Fp12 f;
  K0(f) = co.coef(s, 0); K1(f) = co.coef(s, 1); K2(f) = co.coef(s, 2); K3(f) = co.coef(s, 3); K4(f) = co.coef(s, 4); K5(f) = co.coef(s, 5);
  Fp12 r = fp12_inv(f);
  const uint32_t c = co.c;
  const Fp2 o = fp2_select(c == 0, K0(r), fp2_select(c == 1, K1(r), fp2_select(c == 2, K2(r), fp2_select(c == 3, K3(r), fp2_select(c == 4, K4(r), K5(r))))));
  co.put(d, fp_select(co.h != 0, o.c1, o.c0));
}
struct Coop12Ops {
  const Coop12& co;
  __device__ __forceinline__ void f12_inv(int d, int a) { c12_inv(co, C12_SLOT(d), C12_SLOT(a)); }
  __device__ __forceinline__ void f12_conj(int d, int a) { c12_conj(co, C12_SLOT(d), C12_SLOT(a)); }
  __device__ __forceinline__ void f12_mul(int d, int a, int b, bool conj_b = false) { c12_mul(co, C12_SLOT(d), C12_SLOT(a), C12_SLOT(b), conj_b); }
  __device__ __forceinline__ void f12_frob(int d, int a, int j) { c12_frob(co, C12_SLOT(d), C12_SLOT(a), j); }
  __device__ __forceinline__ void f12_cyclo_sqr(int d, int a) { c12_cyclo_sqr_n(co, C12_SLOT(d), C12_SLOT(a), 1); }
  __device__ __forceinline__ void f12_cyclo_sqr_n(int d, int a, int count) { c12_cyclo_sqr_n(co, C12_SLOT(d), C12_SLOT(a), count); }
};

// ---- workspace access -----------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ Fp c12_ws_ld(const int32_t* ws, uint32_t n, uint32_t p, int e) {
  Fp r;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) r.v[l] = ws[((size_t)e * BN_NL + l) * n + p];
  return r;
}
__device__ __forceinline__ void c12_ws_st(int32_t* ws, uint32_t n, uint32_t p, int e, const Fp& a) {
#pragma unroll
  for (int l = 0; l < BN_NL; l++) ws[((size_t)e * BN_NL + l) * n + p] = a.v[l];
}
__device__ __forceinline__ Fp2 c12_ws_ld2(const int32_t* ws, uint32_t n, uint32_t p, int e) { Fp2 r; r.c0 = c12_ws_ld(ws, n, p, e); r.c1 = c12_ws_ld(ws, n, p, e + 1); return r; }
__device__ __forceinline__ FixedLine c12_line_entry(const int32_t* entry) {   // wave-uniform table entry: scalar loads
  FixedLine l;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) {
    l.m.c0.v[i] = entry[i]; l.m.c1.v[i] = entry[BN_NL + i]; l.c.c0.v[i] = entry[2 * BN_NL + i]; l.c.c1.v[i] = entry[3 * BN_NL + i];
    l.xc.c0.v[i] = entry[4 * BN_NL + i]; l.xc.c1.v[i] = entry[5 * BN_NL + i];
  }
  return l;
}
// the value in `slot` goes back to the workspace element e: lane (c, h) holds Fp number 2 c + h of it
__device__ __forceinline__ void c12_store_f12(const Coop12& co, int32_t* ws, uint32_t n, uint32_t p, int slot, int e, bool pending) {
  const Fp r = co.own(slot);
  if (pending) c12_ws_st(ws, n, p, e + 2 * (int)co.c + (int)co.h, r);
}

// the value in `slot` against a constant (12 Fp in k-order): every lane compares its own Fp number, the twelve lanes of a proof vote
__device__ __forceinline__ bool c12_eq_const(const Coop12& co, int slot, const int32_t* __restrict__ target, uint32_t pl) {
  Fp t;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) t.v[l] = target[(2 * (int)co.c + (int)co.h) * BN_NL + l];
  BN_SETB(t, 1.0, 0.5);
  const uint64_t m = __builtin_amdgcn_ballot_w64(fp_eq(co.own(slot), t));
  return ((m >> (pl * 12u)) & 0xfffull) == 0xfffull;
}

#define C12_PROLOGUE()                                                                                        \
  extern __shared__ __attribute__((aligned(16))) int32_t c12_lds[];                                           \
  const uint32_t lane = threadIdx.x & 63;                                                                     \
  const bool act = lane < 12 * C12_PER_WAVE;                                                                  \
  const uint32_t la = act ? lane : lane - 12;             /* idle lanes shadow the last group */             \
  const uint32_t pl = la / 12, c = (la - pl * 12) >> 1, h = la & 1;                                           \
  const uint32_t p = blockIdx.x * (uint32_t)C12_PER_WAVE + pl;                                                \
  const bool live = act && p < n;                                                                             \
  const uint32_t pc = p < n ? p : n - 1;                                                                      \
  const uint8_t st = status[pc];                                                                              \
  const bool pending = live && (st & BN254_ST_PENDING) != 0;                                                  \
  if (__builtin_amdgcn_ballot_w64(pending) == 0) return;                                                      \
  Coop12 co{(c12_lds_i32*)c12_lds, lane, pl * 12, c, h}

// ---- final exponentiation of VE_F (workspace) -> VE_S0 (workspace), the whole program in one launch ---------------------------------------------------------
__global__ void __launch_bounds__(64) k_coop12_final_exp(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status) {
  C12_PROLOGUE();
  co.put(C12_SLOT(VE_F), c12_ws_ld(ws, n, pc, VE_F + 2 * (int)c + (int)h));
  Coop12Ops ops{co};
  vm_final_exp_program(ops);
  c12_store_f12(co, ws, n, p, C12_SLOT(VE_S0), VE_S0, pending);
}

// ---- Miller loop of the table-driven pairs only (PlonK's two-pair check): f = prod_t Miller(P_t, Q_t), then (optionally) the final exponentiation -------------
__global__ void __launch_bounds__(64)
k_coop12_miller_fixed(int32_t* ws, uint32_t n, uint8_t* status, const uint8_t* __restrict__ kinds, int n_pairs,
                      const int32_t* __restrict__ tab0, const int32_t* __restrict__ tab1, const int32_t* __restrict__ tab2,
                      int e_p0, int e_p1, int e_p2, int inf0, int inf1, int inf2, int fuse_final_exp, const int32_t* __restrict__ target, int reject_code) {
  C12_PROLOGUE();
  const int F = C12_SLOT(VE_F);
  co.put(F, (c == 0 && h == 0) ? fp_one() : fp_zero());
  const int np = __builtin_amdgcn_readfirstlane(n_pairs);
  const Fp px0 = c12_ws_ld(ws, n, pc, e_p0), py0 = c12_ws_ld(ws, n, pc, e_p0 + 1), px1 = c12_ws_ld(ws, n, pc, e_p1), py1 = c12_ws_ld(ws, n, pc, e_p1 + 1);
  const Fp px2 = c12_ws_ld(ws, n, pc, e_p2), py2 = c12_ws_ld(ws, n, pc, e_p2 + 1);
  const bool i0 = (st & inf0) != 0, i1 = (st & inf1) != 0, i2 = (st & inf2) != 0;
  for (int s = 0; s < BN_ATE_STEPS; s++) {
    const int kind = __builtin_amdgcn_readfirstlane((int)kinds[s]);
    if (kind == 0 && s != 0) c12_sqr(co, F);
    // (m x_P)_h = m_h x_P and (i m x_P)_h = (i m)_h x_P: one Fp product each
    { const FixedLine l = c12_line_entry(tab0 + (size_t)s * FIXED_LINE_DWORDS); const C12H m = c12_halves(l.m, h);
      C12H d3; d3.p = fp_mul(m.p, px0); d3.q = fp_mul(m.q, px0); c12_mul_line_fp(co, F, py0, d3, c12_halves(l.c, h), i0); }
    if (np > 1) { const FixedLine l = c12_line_entry(tab1 + (size_t)s * FIXED_LINE_DWORDS); const C12H m = c12_halves(l.m, h);
      C12H d3; d3.p = fp_mul(m.p, px1); d3.q = fp_mul(m.q, px1); c12_mul_line_fp(co, F, py1, d3, c12_halves(l.c, h), i1); }
    if (np > 2) { const FixedLine l = c12_line_entry(tab2 + (size_t)s * FIXED_LINE_DWORDS); const C12H m = c12_halves(l.m, h);
      C12H d3; d3.p = fp_mul(m.p, px2); d3.q = fp_mul(m.q, px2); c12_mul_line_fp(co, F, py2, d3, c12_halves(l.c, h), i2); }
  }
  if (fuse_final_exp) {
    Coop12Ops ops{co};
    vm_final_exp_program(ops);
    if (target) {
      // the verdict in the same launch (k_g16_compare's rule): the result never reaches the workspace
      const bool acc = c12_eq_const(co, C12_SLOT(VE_S0), target, pl);
      if (pending && c == 0 && h == 0) status[p] = acc ? BN254_ST_ACCEPT : (uint8_t)reject_code;
    } else c12_store_f12(co, ws, n, p, C12_SLOT(VE_S0), VE_S0, pending);
  } else c12_store_f12(co, ws, n, p, F, VE_F, pending);
}

// ---- Groth16: public-input MSM, the shared Miller loop of (A, B) with the running G2 point and the two table-driven pairs, final exponentiation --------------
// The G2 step's independent Fp2 products are dealt to the six coefficient positions in ROUNDS; the two lanes of a position
// compute the two halves of its product.  T = (X, Y, Z) is kept whole by every lane.
__device__ __forceinline__ Fp2 c12_sel6(uint32_t c, const Fp2& v0, const Fp2& v1, const Fp2& v2, const Fp2& v3, const Fp2& v4, const Fp2& v5) {
  return fp2_select(c == 0, v0, fp2_select(c == 1, v1, fp2_select(c == 2, v2, fp2_select(c == 3, v3, fp2_select(c == 4, v4, v5)))));
}
__device__ __forceinline__ Fp2 c12_fp_as_fp2(const Fp& a) { Fp2 r; r.c0 = a; r.c1 = fp_zero(); return r; }
__device__ __forceinline__ Fp c12_prod(const Coop12& co, const Fp2& u, const Fp2& v) {   // my half of u v
  const C12H vh = c12_halves(v, co.h);
  return fp_dot(dplus(u.c0, vh.p), dplus(u.c1, vh.q));
}
struct C12Line { Fp2 d0, d3, d4, s1, s2, cz; };   // the variable pair's line at A; m1 X_L, m2 x_C and c1 Z_L of the two table-driven pairs (L projective)
__device__ __forceinline__ void c12_g2_double(const Coop12& co, G2Proj& t, const Fp& xa, const Fp& ya, const Fp2& m1, const Fp& xl, const Fp2& m2, const Fp& xc, const Fp2& c1,
                                              const Fp& zl, C12Line& out) {
  const uint32_t c = co.c;
  const Fp2 yz = fp2_add(t.y, t.z);
  // round 1: X Y, Y^2, Z^2, X^2, (Y + Z)^2, m1 xl
  co.put(C12_R1, c12_prod(co, c12_sel6(c, t.x, t.y, t.z, t.x, yz, m1), c12_sel6(c, t.y, t.y, t.z, t.x, yz, c12_fp_as_fp2(xl))));
  const Fp2 A = co.coef(C12_R1, 0), B = co.coef(C12_R1, 1), C = co.coef(C12_R1, 2), J = co.coef(C12_R1, 3), S = co.coef(C12_R1, 4);
  out.s1 = co.coef(C12_R1, 5);
  const Fp2 H = fp2_sub2(S, B, C);                       // 2 Y Z
  // round 2: B H, -, b3 C, H ya, J xa, m2 xc
  const Fp2 b3 = fp2_from_limbs(BN_TWIST_3B0, BN_TWIST_3B1);
  co.put(C12_R2, c12_prod(co, c12_sel6(c, B, B, b3, H, J, m2), c12_sel6(c, H, B, C, c12_fp_as_fp2(ya), c12_fp_as_fp2(xa), c12_fp_as_fp2(xc))));
  const Fp2 BH = co.coef(C12_R2, 0), E = co.coef(C12_R2, 2), Hy = co.coef(C12_R2, 3), Jx = co.coef(C12_R2, 4);
  out.s2 = co.coef(C12_R2, 5);
  const Fp2 F = fp2_mul_small(E, 3);
  const Fp2 BmF = fp2_sub(B, F), BF = fp2_add(B, F);
  // round 3: E^2, A (B - F), (B + F)^2, c1 zl
  co.put(C12_R3, c12_prod(co, c12_sel6(c, E, A, BF, c1, E, E), c12_sel6(c, E, BmF, BF, c12_fp_as_fp2(zl), E, E)));
  const Fp2 E2 = co.coef(C12_R3, 0), AX = co.coef(C12_R3, 1), BF2 = co.coef(C12_R3, 2);
  out.cz = co.coef(C12_R3, 3);
  t.x = fp2_dbl(AX);
  t.y = fp2_sub(BF2, fp2_mul_small(E2, 12));
  t.z = fp2_mul_small(BH, 4);
  out.d0 = fp2_neg(Hy);
  out.d3 = fp2_mul_small(Jx, 3);
  out.d4 = fp2_sub(E, B);
}
__device__ __forceinline__ void c12_g2_add(const Coop12& co, G2Proj& t, const Fp2& qx, const Fp2& qy, const Fp& xa, const Fp& ya, const Fp2& m1, const Fp& xl, const Fp2& m2,
                                           const Fp& xc, const Fp2& c1, const Fp& zl, C12Line& out) {
  const uint32_t c = co.c;
  // round 1: yQ Z, xQ Z, -, -, -, m1 xl
  co.put(C12_R1, c12_prod(co, c12_sel6(c, qy, qx, qx, qx, qx, m1), c12_sel6(c, t.z, t.z, t.z, t.z, t.z, c12_fp_as_fp2(xl))));
  const Fp2 O = fp2_sub(t.y, co.coef(C12_R1, 0)), L = fp2_sub(t.x, co.coef(C12_R1, 1));
  out.s1 = co.coef(C12_R1, 5);
  // round 2: O^2, L^2, xQ O, L yQ, L ya, m2 xc
  co.put(C12_R2, c12_prod(co, c12_sel6(c, O, L, qx, L, L, m2), c12_sel6(c, O, L, O, qy, c12_fp_as_fp2(ya), c12_fp_as_fp2(xc))));
  const Fp2 Cc = co.coef(C12_R2, 0), D = co.coef(C12_R2, 1), xqO = co.coef(C12_R2, 2), Lyq = co.coef(C12_R2, 3);
  out.d0 = co.coef(C12_R2, 4);
  out.s2 = co.coef(C12_R2, 5);
  // round 3: L D, Z C, X D, O xa, c1 zl
  co.put(C12_R3, c12_prod(co, c12_sel6(c, L, t.z, t.x, O, c1, O), c12_sel6(c, D, Cc, D, c12_fp_as_fp2(xa), c12_fp_as_fp2(zl), O)));
  const Fp2 E = co.coef(C12_R3, 0), Fz = co.coef(C12_R3, 1), G = co.coef(C12_R3, 2), Ox = co.coef(C12_R3, 3);
  out.cz = co.coef(C12_R3, 4);
  const Fp2 H = fp2_sub(fp2_add(E, Fz), fp2_dbl(G));
  const Fp2 GmH = fp2_sub(G, H);
  // round 4: L H, (G - H) O, Y E, E Z     (round-1 slot reused: its values are in registers by now)
  co.put(C12_R1, c12_prod(co, c12_sel6(c, L, GmH, t.y, E, E, E), c12_sel6(c, H, O, E, t.z, E, E)));
  t.x = co.coef(C12_R1, 0);
  t.y = fp2_sub(co.coef(C12_R1, 1), co.coef(C12_R1, 2));
  t.z = co.coef(C12_R1, 3);
  out.d3 = fp2_neg(Ox);
  out.d4 = fp2_sub(xqO, Lyq);
}
__device__ __forceinline__ G1Aff c12_msm_entry(const int32_t* __restrict__ msm_tab, size_t idx) {
  const int4* e = (const int4*)(msm_tab + idx * MSM_ENTRY_DWORDS);
  int4 v0 = e[0], v1 = e[1], v2 = e[2], v3 = e[3], v4 = e[4];
  G1Aff q;
  q.x.v[0] = v0.x; q.x.v[1] = v0.y; q.x.v[2] = v0.z; q.x.v[3] = v0.w; q.x.v[4] = v1.x; q.x.v[5] = v1.y; q.x.v[6] = v1.z; q.x.v[7] = v1.w;
  q.x.v[8] = v2.x; q.y.v[0] = v2.y; q.y.v[1] = v2.z; q.y.v[2] = v2.w; q.y.v[3] = v3.x; q.y.v[4] = v3.y; q.y.v[5] = v3.z; q.y.v[6] = v3.w;
  q.y.v[7] = v4.x; q.y.v[8] = v4.y;
  return q;
}
// L = K0 + sum_i x_i K_i (groth16/verify.rs:53-63) by the twelve lanes of a proof: lane l adds the table entries of the windows w = l, l + 12, ...
// (MSM_FW_WINDOWS = 20 windows of 13 bits per input, bn254_fw.h); the twelve partial sums are then added through LDS (lanes 0..5: own + lane l + 6; lanes 0, 1: l, l + 2, l + 4; 0 + 1).
// L stays PROJECTIVE: the line of the pair (L, g') is scaled by Z_L, an Fp factor the final exponentiation removes.
__device__ __noinline__ G1Proj c12_public_input_msm(const Coop12 co, const uint8_t* __restrict__ in /* this proof's inputs */, int n_public, bool use_inputs,
                                                    const int32_t* __restrict__ msm_tab, const int32_t* __restrict__ k0) {
  const uint32_t l12 = 2 * co.c + co.h;
  G1Proj acc = g1_identity();
  if (l12 == 0) { G1Aff K0; for (int l = 0; l < BN_NL; l++) { K0.x.v[l] = k0[l]; K0.y.v[l] = k0[BN_NL + l]; } acc = g1_from_affine(K0); }
  const int windows = use_inputs ? MSM_FW_WINDOWS * n_public : 0;
  for (int w = (int)l12; w < windows; w += 12) {
    const int sidx = w / MSM_FW_WINDOWS, wi = w - sidx * MSM_FW_WINDOWS;
    // window wi = bits 13 wi .. 13 wi + 12 of the big-endian scalar (bn254_fw.h): at most three of its bytes, byte 31 - k holding bits 8 k .. 8 k + 7
    const int bit = MSM_FW_BITS * wi, k0b = bit >> 3;
    const uint8_t* sc = in + (size_t)sidx * 32;
    const uint32_t three = (uint32_t)sc[31 - k0b] | (k0b + 1 < 32 ? (uint32_t)sc[30 - k0b] << 8 : 0u) | (k0b + 2 < 32 ? (uint32_t)sc[29 - k0b] << 16 : 0u);
    const uint32_t dig = (three >> (bit & 7)) & MSM_FW_ENTRIES;
    G1Proj nxt = g1_add_mixed(acc, c12_msm_entry(msm_tab, (size_t)(sidx * MSM_FW_WINDOWS + wi) * MSM_FW_ENTRIES + (dig ? dig - 1 : 0)));
    const bool take = dig != 0;
    acc.x = fp_select(take, nxt.x, acc.x); acc.y = fp_select(take, nxt.y, acc.y); acc.z = fp_select(take, nxt.z, acc.z);
  }
  auto publish = [&](const G1Proj& a) { co.put(C12_R1, fp_reduce(a.x)); co.put(C12_R2, fp_reduce(a.y)); co.put(C12_R3, fp_reduce(a.z)); };
  auto fetch = [&](uint32_t i) { G1Proj r; r.x = co.at(C12_R1, co.g0 + i); r.y = co.at(C12_R2, co.g0 + i); r.z = co.at(C12_R3, co.g0 + i); return r; };
  publish(acc);
  acc = g1_add(acc, fetch(l12 < 6 ? l12 + 6 : l12));          // lanes 6..11 add their own value (result unused)
  publish(acc);
  { const uint32_t b = l12 < 2 ? l12 : 0; acc = g1_add(g1_add(fetch(b), fetch(b + 2)), fetch(b + 4)); }
  publish(acc);
  acc = g1_add(fetch(0), fetch(1));
  publish(acc);
  return fetch(0);
}
__global__ void __launch_bounds__(64)
k_coop12_miller_g16(int32_t* ws, uint32_t n, uint8_t* status, const uint8_t* __restrict__ kinds, const int32_t* __restrict__ tab0,
                    const int32_t* __restrict__ tab1, const uint8_t* __restrict__ inputs, int n_public, int inputs_match_key,
                    const int32_t* __restrict__ msm_tab, const int32_t* __restrict__ k0, int l_from_ws, int fuse_final_exp, const int32_t* __restrict__ target) {
  C12_PROLOGUE();
  const int F = C12_SLOT(VE_F);
  // keys with many public inputs: L was computed by the wide MSM kernels (affine, in the workspace, identity flag in the status byte)
  G1Proj Lp;
  if (l_from_ws) { Lp.x = c12_ws_ld(ws, n, pc, VE_LX); Lp.y = c12_ws_ld(ws, n, pc, VE_LY); Lp.z = (st & BN254_ST_LINF) ? fp_zero() : fp_one(); }
  else Lp = c12_public_input_msm(co, inputs + (size_t)pc * (size_t)n_public * 32, n_public, inputs_match_key != 0, msm_tab, k0);
  const bool l_inf = fp_is_zero(Lp.z);
  const Fp xl = Lp.x, yl = fp_select(l_inf, fp_one(), Lp.y), zl = Lp.z;
  co.put(F, (c == 0 && h == 0) ? fp_one() : fp_zero());
  const Fp xa = c12_ws_ld(ws, n, pc, VE_AX), ya = c12_ws_ld(ws, n, pc, VE_AY);
  const Fp xc = c12_ws_ld(ws, n, pc, VE_CX), yc = c12_ws_ld(ws, n, pc, VE_CY);
  G2Aff q; q.x = c12_ws_ld2(ws, n, pc, VE_B); q.y = c12_ws_ld2(ws, n, pc, VE_B + 2);
  G2Proj t = g2_from_affine(q);
  for (int s = 0; s < BN_ATE_STEPS; s++) {
    const int kind = __builtin_amdgcn_readfirstlane((int)kinds[s]);
    if (kind == 0 && s != 0) c12_sqr(co, F);
    const FixedLine l0 = c12_line_entry(tab0 + (size_t)s * FIXED_LINE_DWORDS), l1 = c12_line_entry(tab1 + (size_t)s * FIXED_LINE_DWORDS);
    C12Line ln;
    if (kind == 0) {
      c12_g2_double(co, t, xa, ya, l0.m, xl, l1.m, xc, l0.c, zl, ln);
    } else {
      G2Aff b = q;
      if (kind == 2) b = g2_neg(q);
      else if (kind == 3) b = g2_psi_affine(q);
      else if (kind == 4) b = g2_neg(g2_psi2_affine(q));
      c12_g2_add(co, t, b.x, b.y, xa, ya, l0.m, xl, l1.m, xc, l0.c, zl, ln);
    }
    c12_mul_line_fp2(co, F, c12_halves(ln.d0, h), c12_halves(ln.d3, h), c12_halves(ln.d4, h));
    c12_mul_line_fp(co, F, yl, c12_halves(ln.s1, h), c12_halves(ln.cz, h), l_inf);      // (Y_L + m X_L w + c Z_L w^3): the line at L scaled by Z_L
    c12_mul_line_fp(co, F, yc, c12_halves(ln.s2, h), c12_halves(l1.c, h), false);
  }
  if (fuse_final_exp && target) {
    // The whole verdict in this launch (what k_g16_subgroup and k_g16_compare do for the lane kernels).  The r-torsion test of B from the loop's final
    // point (bn254_vm.h::vm_g2_ate_check: T == -psi^3(B) projectively) costs four Fp2 products: every lane holds T and B whole and computes it for itself.
    const Fp2 sx = fp2_mul(fp2_conj(q.x), frob_coeff(3, 2)), sy = fp2_neg(fp2_mul(fp2_conj(q.y), frob_coeff(3, 3)));
    const bool in_g2 = !fp2_is_zero(t.z) & fp2_eq(t.x, fp2_mul(sx, t.z)) & fp2_eq(t.y, fp2_mul(sy, t.z));
    Coop12Ops ops{co};
    vm_final_exp_program(ops);
    const bool acc = c12_eq_const(co, C12_SLOT(VE_S0), target, pl);
    if (pending && c == 0 && h == 0) {
      uint8_t out;
      if (!in_g2) out = BN254_ST_NOT_IN_SUBGROUP;
      else if (st & 0x3f) out = st & 0x3f;                      // deferred error of C
      else if (!inputs_match_key) out = BN254_ST_INPUT_LEN;     // PrepareInputsFailed comes after every loader error
      else out = acc ? BN254_ST_ACCEPT : BN254_ST_REJECT;
      status[p] = out;
    }
    return;
  }
  // the running point goes back to the workspace for the r-torsion test (k_g16_subgroup): lanes of coefficients 0..2 store X, Y, Z.  Not at
  // VE_T, which the result slot VE_S0 overlays: at COOP_T_ELEM
  if (pending && c < 3) {
    const Fp2 tc = c == 0 ? t.x : c == 1 ? t.y : t.z;
    c12_ws_st(ws, n, p, COOP_T_ELEM + 2 * (int)c + (int)h, h ? tc.c1 : tc.c0);
  }
  if (fuse_final_exp) {
    Coop12Ops ops{co};
    vm_final_exp_program(ops);
    c12_store_f12(co, ws, n, p, C12_SLOT(VE_S0), VE_S0, pending);
  } else c12_store_f12(co, ws, n, p, F, VE_F, pending);
}

}  // namespace bn254

using namespace bn254;
static inline unsigned c12_grid(size_t n) { return (unsigned)((n + C12_PER_WAVE - 1) / C12_PER_WAVE); }
// device copy of the step-kind table (88 bytes), created on first use per device
static const uint8_t* c12_kinds_dev() {
  static uint8_t* dev[64] = {nullptr};
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  int d = 0; (void)hipGetDevice(&d);
  if (d < 0 || d >= 64) return nullptr;
  if (!dev[d]) {
    uint8_t h[BN_ATE_STEPS];
    for (int i = 0; i < BN_ATE_STEPS; i++) h[i] = (uint8_t)miller_step_kind(i);
    if (hipMalloc((void**)&dev[d], BN_ATE_STEPS) != hipSuccess) return nullptr;
    (void)hipMemcpy(dev[d], h, BN_ATE_STEPS, hipMemcpyHostToDevice);
  }
  return dev[d];
}
hipError_t bn254_coop12_final_exp(int32_t* ws, uint8_t* status, size_t n, hipStream_t s) {
  const size_t lds = (size_t)C12_WAVE_DWORDS * 4;
  (void)hipFuncSetAttribute((const void*)k_coop12_final_exp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_coop12_final_exp, dim3(c12_grid(n)), dim3(64), lds, s, ws, (uint32_t)n, (const uint8_t*)status);
  return hipGetLastError();
}
hipError_t bn254_coop12_miller_fixed(int32_t* ws, uint8_t* status, size_t n, int n_pairs, const int32_t* tab0, const int32_t* tab1, const int32_t* tab2,
                                     int e_p0, int e_p1, int e_p2, int inf0, int inf1, int inf2, int fuse_final_exp, const int32_t* target, int reject_code, hipStream_t s) {
  const uint8_t* kinds = c12_kinds_dev();
  if (!kinds) return hipErrorOutOfMemory;
  const size_t lds = (size_t)C12_WAVE_DWORDS * 4;
  (void)hipFuncSetAttribute((const void*)k_coop12_miller_fixed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_coop12_miller_fixed, dim3(c12_grid(n)), dim3(64), lds, s, ws, (uint32_t)n, status, kinds, n_pairs, tab0, tab1, tab2,
                     e_p0, e_p1, e_p2, inf0, inf1, inf2, fuse_final_exp, target, reject_code);
  return hipGetLastError();
}
hipError_t bn254_coop12_miller_g16(int32_t* ws, uint8_t* status, size_t n, const int32_t* tab0, const int32_t* tab1, const uint8_t* inputs, int n_public,
                                   int inputs_match_key, const int32_t* msm_tab, const int32_t* k0, int l_from_ws, int fuse_final_exp, const int32_t* target, hipStream_t s) {
  const uint8_t* kinds = c12_kinds_dev();
  if (!kinds) return hipErrorOutOfMemory;
  const size_t lds = (size_t)C12_WAVE_DWORDS * 4;
  (void)hipFuncSetAttribute((const void*)k_coop12_miller_g16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_coop12_miller_g16, dim3(c12_grid(n)), dim3(64), lds, s, ws, (uint32_t)n, status, kinds, tab0, tab1, inputs, n_public,
                     inputs_match_key, msm_tab, k0, l_from_ws, fuse_final_exp, target);
  return hipGetLastError();
}
