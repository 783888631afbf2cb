// bn254_coop.hip -- the cooperative ("one pairing per wavefront") layout for SMALL batches: six lanes per proof, lane c of a group holds the
// Fp2 coefficient k_c of every Fp12 value  f = sum_c k_c w^c  (w^6 = xi), ten proofs per wavefront (lanes 60..63 idle).
//
// Why: with one proof per lane (bn254_kernels.hip) a batch of 4096 proofs is 64 wavefronts on 1024 SIMDs and every one of the ~420 launches
// lasts as long as ONE lane's serial work (11 ms per batch whatever its size below ~2^15).  Spreading a proof over six
// lanes shortens that serial chain about six times (tools/coop/coop_bench.hip: 7 us instead of 70 us per Miller step at 4096 proofs), and
// because the whole pairing then fits one kernel -- every Fp12 value of the computation lives in LDS -- the launch chain disappears as well.
//
// Data layout: per wavefront an LDS image  img[slot][lane][20 dwords]  (an Fp2 = 18 digits + 2 pad; 20 * lane mod 64 puts 16 consecutive
// lanes on disjoint 4-bank groups for ds_read_b128).  Slot s holds one Fp12 VALUE: lane c of a group keeps coefficient k_c at (s, lane).
// So "fetch coefficient i of value a" is a ds_read at (a, group_base + i): products of the form  r_j = sum_t a_t * b_(j-t)  need no
// separate exchange step, only the xi-multiples of one operand are published to a scratch slot first.  The slots double as the operand
// store of the final exponentiation (the element names of bn254_vm.h map to slots), which is driven by the SAME program template
// (vm_final_exp_program) as the one-proof-per-lane kernels and tests/hostsim: here its operations are small out-of-line device functions
// whose operands are slot numbers, so the kernel stays compact.
//
// Lock-step: a wavefront executes its LDS instructions in order, so a value written by every lane and then read by other lanes of the same
// wavefront needs no barrier; every operation reads all of its inputs before it writes its output slot (in-place operations are safe).
#include <hip/hip_runtime.h>
#include <mutex>
#include "bn254_vm.h"
#include "bn254_kernels.h"

namespace bn254 {

static_assert(COOP_T_ELEM == VE_S2, "COOP_T_ELEM must name a workspace slot that neither VE_T nor the result slot VE_S0 overlays");
#define CO_STRIDE 20
// slots: the Fp12 elements of bn254_vm.h (VE_F, VE_S0.., VE_P3.., 12 Fp apart from VE_F on) + scratch images
#define CO_SLOT(e) (((e) - VE_F) / 12)   // VE_F 0, VE_S0 1, S1 2, S2 3, S3 4, S4 5, P3 6, (TMPA/TMPB 7), P5 8, P7 9
#define CO_X 10      // xi-multiples of the second operand of a product
#define CO_R1 7      // G2 step: products of round 1 (the Fp6 temporaries of the lane layout are not needed here)
#define CO_R2 11
#define CO_R3 12
#define CO_SLOTS 13
#define CO_WAVE_DWORDS (CO_SLOTS * 64 * CO_STRIDE)

typedef __attribute__((address_space(3))) int32_t co_lds_i32;   // LDS pointers keep their address space through the out-of-line operations (ds_* instead of flat_*)
typedef int co_v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) co_v4i co_lds_i4;
struct Coop {
  co_lds_i32* img;       // this wavefront's LDS image
  uint32_t lane, g0, c;  // lane in the wavefront, first lane of the proof's group, coefficient index (lanes 60..63: a copy of group 9, never stored)
  __device__ __forceinline__ void put(int slot, const Fp2& a) const {
    co_lds_i4* q = (co_lds_i4*)(img + ((size_t)slot * 64 + lane) * CO_STRIDE);
    q[0] = (co_v4i){a.c0.v[0], a.c0.v[1], a.c0.v[2], a.c0.v[3]};
    q[1] = (co_v4i){a.c0.v[4], a.c0.v[5], a.c0.v[6], a.c0.v[7]};
    q[2] = (co_v4i){a.c0.v[8], a.c1.v[0], a.c1.v[1], a.c1.v[2]};
    q[3] = (co_v4i){a.c1.v[3], a.c1.v[4], a.c1.v[5], a.c1.v[6]};
    q[4] = (co_v4i){a.c1.v[7], a.c1.v[8], 0, 0};
  }
  __device__ __forceinline__ Fp2 at(int slot, uint32_t ln) const {
    const co_lds_i4* q = (const co_lds_i4*)(img + ((size_t)slot * 64 + ln) * CO_STRIDE);
    const co_v4i v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4];
    Fp2 a;
    a.c0.v[0] = v0.x; a.c0.v[1] = v0.y; a.c0.v[2] = v0.z; a.c0.v[3] = v0.w; a.c0.v[4] = v1.x; a.c0.v[5] = v1.y; a.c0.v[6] = v1.z; a.c0.v[7] = v1.w;
    a.c0.v[8] = v2.x; a.c1.v[0] = v2.y; a.c1.v[1] = v2.z; a.c1.v[2] = v2.w; a.c1.v[3] = v3.x; a.c1.v[4] = v3.y; a.c1.v[5] = v3.z; a.c1.v[6] = v3.w;
    a.c1.v[7] = v4.x; a.c1.v[8] = v4.y;
    return a;
  }
  __device__ __forceinline__ Fp2 own(int slot) const { return at(slot, lane); }
  __device__ __forceinline__ Fp2 coef(int slot, uint32_t i) const { return at(slot, g0 + i); }   // coefficient i of the value in `slot`
};

__device__ __forceinline__ Fp2 co_scale(const Fp2& a, int32_t w) {  // w in {0, 1, 2}, digit-wise
  Fp2 r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) { r.c0.v[i] = a.c0.v[i] * w; r.c1.v[i] = a.c1.v[i] * w; }
  return r;
}
__device__ __forceinline__ Fp2 co_add_r(const Fp2& a, const Fp2& b) { return fp2_reduce(fp2_add(a, b)); }

// ---- f <- f^2 (general squaring, Miller loop): bn254_tower.h::fp12_sqr, coefficient j on lane j --------------------------------------------------
//   r0 = k0 k0 + 2 k1 xk5 + 2 k2 xk4 + k3 xk3     r1 = 2 k0 k1 + 2 k2 xk5 + 2 k3 xk4             r2 = 2 k0 k2 + k1 k1 + 2 k3 xk5 + k4 xk4
//   r3 = 2 k0 k3 + 2 k1 k2 + 2 k4 xk5             r4 = 2 k0 k4 + 2 k1 k3 + k2 k2 + k5 xk5         r5 = 2 k0 k5 + 2 k1 k4 + 2 k2 k3
__constant__ int8_t CO_SQ_A[6][4] = {{0, 1, 2, 3}, {0, 2, 3, 0}, {0, 1, 3, 4}, {0, 1, 4, 0}, {0, 1, 2, 5}, {0, 1, 2, 0}};
__constant__ int8_t CO_SQ_B[6][4] = {{0, 5, 4, 3}, {1, 5, 4, 0}, {2, 1, 5, 4}, {3, 2, 5, 0}, {4, 3, 2, 5}, {5, 4, 3, 0}};
__constant__ int8_t CO_SQ_X[6][4] = {{0, 1, 1, 1}, {0, 1, 1, 0}, {0, 0, 1, 1}, {0, 0, 1, 0}, {0, 0, 0, 1}, {0, 0, 0, 0}};
__constant__ int8_t CO_SQ_W[6][4] = {{1, 2, 2, 1}, {2, 2, 2, 0}, {2, 1, 2, 1}, {2, 2, 2, 0}, {2, 2, 1, 1}, {2, 2, 2, 0}};
__device__ __forceinline__ void co_sqr(const Coop& co, int s) {
  const Fp2 k = co.own(s);
  co.put(CO_X, fp2_mul_xi(k));
  const uint32_t c = co.c;
  Fp2 x0 = co_scale(co.coef(s, CO_SQ_A[c][0]), CO_SQ_W[c][0]);
  Fp2 x1 = co_scale(co.coef(s, CO_SQ_A[c][1]), CO_SQ_W[c][1]);
  Fp2 x2 = co_scale(co.coef(s, CO_SQ_A[c][2]), CO_SQ_W[c][2]);
  Fp2 x3 = co_scale(co.coef(s, CO_SQ_A[c][3]), CO_SQ_W[c][3]);
  Fp2 y1 = co.coef(CO_SQ_X[c][1] ? CO_X : s, CO_SQ_B[c][1]);
  Fp2 y2 = co.coef(CO_SQ_X[c][2] ? CO_X : s, CO_SQ_B[c][2]);
  Fp2 y3 = co.coef(CO_SQ_X[c][3] ? CO_X : s, CO_SQ_B[c][3]);
  co.put(s, fp2_dotk(kp(x0, k), kp(x1, y1), kp(x2, y2), kp(x3, y3)));
}
// ---- f <- f * (d0 + d3 w + d4 w^3), d0 in Fp or Fp2: r_j = d0 k_j + d3 (xi?) k_(j-1) + d4 (xi?) k_(j-3); keep: leave f (line value 1) -------------
__device__ __forceinline__ void co_line_operands(const Coop& co, int s, const Fp2& k, Fp2& y1, Fp2& y3) {
  co.put(CO_X, fp2_mul_xi(k));
  const uint32_t c = co.c;
  y1 = co.coef(c >= 1 ? s : CO_X, (c + 5) % 6);
  y3 = co.coef(c >= 3 ? s : CO_X, (c + 3) % 6);
}
__device__ __forceinline__ void co_mul_line_fp(const Coop& co, int s, const Fp& d0, const Fp2& d3, const Fp2& d4, bool keep) {
  const Fp2 k = co.own(s);
  Fp2 y1, y3;
  co_line_operands(co, s, k, y1, y3);
  co.put(s, fp2_select(keep, k, fp2_dotk(kfp(k, d0), kp(d3, y1), kp(d4, y3))));
}
__device__ __forceinline__ void co_mul_line_fp2(const Coop& co, int s, const Fp2& d0, const Fp2& d3, const Fp2& d4) {
  const Fp2 k = co.own(s);
  Fp2 y1, y3;
  co_line_operands(co, s, k, y1, y3);
  co.put(s, fp2_dotk(kp(d0, k), kp(d3, y1), kp(d4, y3)));
}
// ---- general product d <- a * (conj?) b:  r_j = sum_t a_t (xi if t > j) b_((j - t) mod 6); conjugation negates the odd coefficients of b ------------
__device__ __noinline__ void co_mul(const Coop& co, int d, int a, int b, bool conj_b) {
  const uint32_t c = co.c;
  {
    Fp2 bk = co.own(b);
    if (conj_b && (c & 1)) bk = fp2_neg(bk);
    co.put(CO_X, fp2_mul_xi(bk));
  }
  // second operand of term t: coefficient (c - t) mod 6 of b, from the xi image when t > c; conjugation: sign of odd coefficients
  Fp2 lo, hi;
  {
    Fp2 a0 = co.coef(a, 0), a1 = co.coef(a, 1), a2 = co.coef(a, 2);
    Fp2 b0 = co.coef(0 <= (int)c ? b : CO_X, (c + 6 - 0) % 6), b1 = co.coef(1 <= c ? b : CO_X, (c + 6 - 1) % 6), b2 = co.coef(2 <= c ? b : CO_X, (c + 6 - 2) % 6);
    if (conj_b) {  // plain-image operands of odd index are negated here (the xi image was built from the conjugate already)
      if (((c + 6 - 0) % 6) & 1) b0 = fp2_neg(b0);
      if ((1 <= c) && (((c + 6 - 1) % 6) & 1)) b1 = fp2_neg(b1);
      if ((2 <= c) && (((c + 6 - 2) % 6) & 1)) b2 = fp2_neg(b2);
    }
    lo = fp2_dotk(kp(a0, b0), kp(a1, b1), kp(a2, b2));
  }
  {
    Fp2 a3 = co.coef(a, 3), a4 = co.coef(a, 4), a5 = co.coef(a, 5);
    Fp2 b3 = co.coef(3 <= c ? b : CO_X, (c + 6 - 3) % 6), b4 = co.coef(4 <= c ? b : CO_X, (c + 6 - 4) % 6), b5 = co.coef(5 <= c ? b : CO_X, (c + 6 - 5) % 6);
    if (conj_b) {
      if ((3 <= c) && (((c + 6 - 3) % 6) & 1)) b3 = fp2_neg(b3);
      if ((4 <= c) && (((c + 6 - 4) % 6) & 1)) b4 = fp2_neg(b4);
      if ((5 <= c) && (((c + 6 - 5) % 6) & 1)) b5 = fp2_neg(b5);
    }
    hi = fp2_dotk(kp(a3, b3), kp(a4, b4), kp(a5, b5));
  }
  co.put(d, fp2_add(lo, hi));
}
// ---- Granger-Scott squaring on the cyclotomic subgroup, `count` times (bn254_tower.h::fp12_cyclo_sqr).  Pairs (a, b) in w-power numbering:
//   (k0, k3), (k1, k4), (k2, k5);   z_a = 3 (xi b^2 + a^2) - 2 sub,   z_b = 3 (2 a b) + 2 add   with (sub, add) = (k0, k3), (k2, k5), (k4, k1) and xi on the
//   cross term of the third pair: z0 <- (k0,k3), z3;  z2 <- (k1,k4) with sub k2, z5 with add k5;  z4 <- (k2,k5) with sub k4, z1 = 3 xi (2 k2 k5) + 2 k1
// lane -> (a, b, role): 0: S(k0,k3) - k0 | 3: T(k0,k3) + k3 | 2: S(k1,k4) - k2 | 5: T(k1,k4) + k5 | 4: S(k2,k5) - k4 | 1: xi T(k2,k5) + k1
__constant__ int8_t CO_CY_A[6] = {0, 2, 1, 0, 2, 1};
__constant__ int8_t CO_CY_B[6] = {3, 5, 4, 3, 5, 4};
__device__ __noinline__ void co_cyclo_sqr_n(const Coop& co, int d, int s, int count) {
  const uint32_t c = co.c;
  const bool is_s = (c == 0) | (c == 2) | (c == 4);   // lanes that compute S = xi b^2 + a^2; the others T = 2 a b (xi on it for lane 1)
  const int32_t lin = is_s ? -2 : 2;
  Fp2 k = co.own(s);
  for (int it = 0; it < count; it++) {
    const int src = it == 0 ? s : d;
    const Fp2 a = co.coef(src, CO_CY_A[c]), b = co.coef(src, CO_CY_B[c]);
    // one instruction stream for both roles: X = u1 v1 + u2 v2 with (u1, v1, u2, v2) = (xi b, b, a, a) or (2 a, b or xi b, 0, a)
    const Fp2 xb = fp2_mul_xi(b);
    const Fp2 u1 = fp2_select(is_s, xb, co_scale(a, 2));
    const Fp2 v1 = fp2_select(is_s | (c == 1), is_s ? b : xb, b);
    const Fp2 u2 = co_scale(a, is_s ? 1 : 0);
    const Fp2 X = fp2_dotk(kp(u1, v1), kp(u2, a));
    const Fp2 z = fp2_lincomb_reduce(3, X, lin, k);
    co.put(d, z);
    k = z;
  }
}
__device__ __noinline__ void co_conj(const Coop& co, int d, int s) {
  Fp2 k = co.own(s);
  co.put(d, (co.c & 1) ? fp2_neg(k) : k);
}
__device__ __noinline__ void co_frob(const Coop& co, int d, int s, int j) {
  Fp2 k = co.own(s);
  if (j & 1) k = fp2_conj(k);
  const int32_t(*t)[2][BN_NL] = j == 1 ? BN_FROB_G1 : j == 2 ? BN_FROB_G2 : BN_FROB_G3;
  Fp2 g;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { g.c0.v[l] = t[co.c][0][l]; g.c1.v[l] = t[co.c][1][l]; }
  co.put(d, fp2_mul(k, g));   // coefficient 0: g = 1
}
// inverse: every lane gathers the whole value and runs the one-proof-per-lane inversion (one Fermat inversion in Fp dominates: it has no
// parallelism to distribute); lane c keeps coefficient c
__device__ __noinline__ void co_inv(const Coop& co, int d, int s) {
  Fp12 f;
  K0(f) = co.coef(s, 0); K1(f) = co.coef(s, 1); K2(f) = co.coef(s, 2); K3(f) = co.coef(s, 3); K4(f) = co.coef(s, 4); K5(f) = co.coef(s, 5);
  Fp12 r = fp12_inv(f);
  const uint32_t c = co.c;
  Fp2 o = fp2_select(c == 0, K0(r), fp2_select(c == 1, K1(r), fp2_select(c == 2, K2(r), fp2_select(c == 3, K3(r), fp2_select(c == 4, K4(r), K5(r))))));
  co.put(d, o);
}
// the operation set vm_final_exp_program expects, on slots
struct CoopOps {
  const Coop& co;
  __device__ __forceinline__ void f12_inv(int d, int a) { co_inv(co, CO_SLOT(d), CO_SLOT(a)); }
  __device__ __forceinline__ void f12_conj(int d, int a) { co_conj(co, CO_SLOT(d), CO_SLOT(a)); }
  __device__ __forceinline__ void f12_mul(int d, int a, int b, bool conj_b = false) { co_mul(co, CO_SLOT(d), CO_SLOT(a), CO_SLOT(b), conj_b); }
  __device__ __forceinline__ void f12_frob(int d, int a, int j) { co_frob(co, CO_SLOT(d), CO_SLOT(a), j); }
  __device__ __forceinline__ void f12_cyclo_sqr(int d, int a) { co_cyclo_sqr_n(co, CO_SLOT(d), CO_SLOT(a), 1); }
  __device__ __forceinline__ void f12_cyclo_sqr_n(int d, int a, int count) { co_cyclo_sqr_n(co, CO_SLOT(d), CO_SLOT(a), count); }
};

// ---- workspace access of a cooperative lane: coefficient c of the Fp12 element e of proof p = Fp elements e + 2c, e + 2c + 1 ----------------------------
__device__ __forceinline__ Fp co_ws_ld(const int32_t* ws, uint32_t n, uint32_t p, int e) {
  Fp r;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) r.v[l] = ws[((size_t)e * BN_NL + l) * n + p];
  return r;
}
__device__ __forceinline__ void co_ws_st(int32_t* ws, uint32_t n, uint32_t p, int e, const Fp& a) {
#pragma unroll
  for (int l = 0; l < BN_NL; l++) ws[((size_t)e * BN_NL + l) * n + p] = a.v[l];
}
__device__ __forceinline__ Fp2 co_ws_ld2(const int32_t* ws, uint32_t n, uint32_t p, int e) { Fp2 r; r.c0 = co_ws_ld(ws, n, p, e); r.c1 = co_ws_ld(ws, n, p, e + 1); return r; }
__device__ __forceinline__ FixedLine co_line_entry(const int32_t* entry) {   // wave-uniform table entry: scalar loads
  FixedLine l;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) {
    l.m.c0.v[i] = entry[i]; l.m.c1.v[i] = entry[BN_NL + i]; l.c.c0.v[i] = entry[2 * BN_NL + i]; l.c.c1.v[i] = entry[3 * BN_NL + i];
    l.xc.c0.v[i] = entry[4 * BN_NL + i]; l.xc.c1.v[i] = entry[5 * BN_NL + i];
  }
  return l;
}

#define CO_PROLOGUE()                                                                                         \
  extern __shared__ __attribute__((aligned(16))) int32_t co_lds[];                                            \
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                            \
  const uint32_t waves_per_block = blockDim.x >> 6;                                                           \
  const bool act = lane < 60;                                                                                 \
  const uint32_t pl = act ? lane / 6 : 9, c = act ? lane - pl * 6 : lane - 60;                                \
  const uint32_t p = (blockIdx.x * waves_per_block + wave) * 10u + pl;                                        \
  const bool live = act && p < n;                                                                             \
  const uint32_t pc = p < n ? p : n - 1;                                                                      \
  const uint8_t st = status[pc];                                                                              \
  const bool pending = live && (st & BN254_ST_PENDING) != 0;                                                  \
  if (__builtin_amdgcn_ballot_w64(pending) == 0) return;                                                      \
  Coop co{(co_lds_i32*)co_lds + (size_t)wave * CO_WAVE_DWORDS, lane, pl * 6, c}

// ---- final exponentiation of VE_F (workspace) -> VE_S0 (workspace), the whole program in one launch ---------------------------------------------------------
__global__ void __launch_bounds__(64) k_coop_final_exp(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status) {
  CO_PROLOGUE();
  co.put(CO_SLOT(VE_F), co_ws_ld2(ws, n, pc, VE_F + 2 * (int)c));
  CoopOps ops{co};
  vm_final_exp_program(ops);
  if (pending) { Fp2 r = co.own(CO_SLOT(VE_S0)); co_ws_st(ws, n, p, VE_S0 + 2 * (int)c, r.c0); co_ws_st(ws, n, p, VE_S0 + 2 * (int)c + 1, r.c1); }
}

// ---- Miller loop of the table-driven pairs only (PlonK's two-pair check, the group stage of the RLC mode): f = prod_t Miller(P_t, Q_t) -> VE_F ---------------
// e_p[t]: workspace element of the G1 point of pair t; inf_mask[t]: status bit that marks it as the identity
__global__ void __launch_bounds__(64)
k_coop_miller_fixed(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, const uint8_t* __restrict__ kinds, int n_pairs,
                    const int32_t* __restrict__ tab0, const int32_t* __restrict__ tab1, const int32_t* __restrict__ tab2,
                    int e_p0, int e_p1, int e_p2, int inf0, int inf1, int inf2, int fuse_final_exp) {
  CO_PROLOGUE();
  const int F = CO_SLOT(VE_F);
  { Fp2 one = c == 0 ? fp2_one() : fp2_zero(); co.put(F, one); }
  const int np = __builtin_amdgcn_readfirstlane(n_pairs);
  const Fp px0 = co_ws_ld(ws, n, pc, e_p0), py0 = co_ws_ld(ws, n, pc, e_p0 + 1), px1 = co_ws_ld(ws, n, pc, e_p1), py1 = co_ws_ld(ws, n, pc, e_p1 + 1);
  const Fp px2 = co_ws_ld(ws, n, pc, e_p2), py2 = co_ws_ld(ws, n, pc, e_p2 + 1);
  const bool i0 = (st & inf0) != 0, i1 = (st & inf1) != 0, i2 = (st & inf2) != 0;
  for (int s = 0; s < BN_ATE_STEPS; s++) {
    const int kind = __builtin_amdgcn_readfirstlane((int)kinds[s]);
    if (kind == 0 && s != 0) co_sqr(co, F);
    { FixedLine l = co_line_entry(tab0 + (size_t)s * FIXED_LINE_DWORDS); co_mul_line_fp(co, F, py0, fp2_mul_fp(l.m, px0), l.c, i0); }
    if (np > 1) { FixedLine l = co_line_entry(tab1 + (size_t)s * FIXED_LINE_DWORDS); co_mul_line_fp(co, F, py1, fp2_mul_fp(l.m, px1), l.c, i1); }
    if (np > 2) { FixedLine l = co_line_entry(tab2 + (size_t)s * FIXED_LINE_DWORDS); co_mul_line_fp(co, F, py2, fp2_mul_fp(l.m, px2), l.c, i2); }
  }
  if (fuse_final_exp) {
    CoopOps ops{co};
    vm_final_exp_program(ops);
    if (pending) { Fp2 r = co.own(CO_SLOT(VE_S0)); co_ws_st(ws, n, p, VE_S0 + 2 * (int)c, r.c0); co_ws_st(ws, n, p, VE_S0 + 2 * (int)c + 1, r.c1); }
  } else if (pending) {
    Fp2 r = co.own(F); co_ws_st(ws, n, p, VE_F + 2 * (int)c, r.c0); co_ws_st(ws, n, p, VE_F + 2 * (int)c + 1, r.c1);
  }
}


// ---- Groth16: the shared Miller loop of (A, B) with a running G2 point and the two table-driven pairs (L, g'), (C, d') -------------------------------------
// The G2 step has at most five independent Fp2 products at a time (Costello-Lange-Naehrig formulas of bn254_curve.h): they are dealt to the six
// lanes in ROUNDS -- lane c multiplies its pair of operands, publishes the product in a round slot, everyone fetches what the next round
// needs.  T = (X, Y, Z) is kept by every lane (the linear combinations between rounds are cheap and need no exchange); lanes that have no
// product of the G2 step in a round compute the scaled slopes m * x_P of the two table-driven lines instead.
__device__ __forceinline__ Fp2 co_sel6(uint32_t c, const Fp2& v0, const Fp2& v1, const Fp2& v2, const Fp2& v3, const Fp2& v4, const Fp2& v5) {
  return fp2_select(c == 0, v0, fp2_select(c == 1, v1, fp2_select(c == 2, v2, fp2_select(c == 3, v3, fp2_select(c == 4, v4, v5)))));
}
__device__ __forceinline__ Fp2 co_fp_as_fp2(const Fp& a) { Fp2 r; r.c0 = a; r.c1 = fp_zero(); return r; }
struct CoLine { Fp2 d0, d3, d4, s1, s2, cz; };   // the variable pair's line at A; m1 * X_L, m2 * x_C and c1 * Z_L of the two table-driven pairs (L projective)
// doubling step: T <- 2 T (bn254_curve.h::g2_double_step), line evaluated at A = (xa, ya); m1 xl, m2 xc in the idle lanes
__device__ __forceinline__ void co_g2_double(const Coop& co, G2Proj& t, const Fp& xa, const Fp& ya, const Fp2& m1, const Fp& xl, const Fp2& m2, const Fp& xc, const Fp2& c1,
                                          const Fp& zl, CoLine& out) {
  const uint32_t c = co.c;
  const Fp2 yz = fp2_add(t.y, t.z);
  {  // round 1: X Y, Y^2, Z^2, X^2, (Y + Z)^2, m1 xl
    const Fp2 u = co_sel6(c, t.x, t.y, t.z, t.x, yz, m1), v = co_sel6(c, t.y, t.y, t.z, t.x, yz, co_fp_as_fp2(xl));
    co.put(CO_R1, fp2_dotk(kp(u, v)));
  }
  const Fp2 A = co.coef(CO_R1, 0), B = co.coef(CO_R1, 1), C = co.coef(CO_R1, 2), J = co.coef(CO_R1, 3), S = co.coef(CO_R1, 4);
  out.s1 = co.coef(CO_R1, 5);
  const Fp2 H = fp2_sub2(S, B, C);                       // 2 Y Z
  {  // round 2: B H, -, b3 C, H ya, J xa, m2 xc
    const Fp2 b3 = fp2_from_limbs(BN_TWIST_3B0, BN_TWIST_3B1);
    const Fp2 u = co_sel6(c, B, B, b3, H, J, m2), v = co_sel6(c, H, B, C, co_fp_as_fp2(ya), co_fp_as_fp2(xa), co_fp_as_fp2(xc));
    co.put(CO_R2, fp2_dotk(kp(u, v)));
  }
  const Fp2 BH = co.coef(CO_R2, 0), E = co.coef(CO_R2, 2), Hy = co.coef(CO_R2, 3), Jx = co.coef(CO_R2, 4);
  out.s2 = co.coef(CO_R2, 5);
  const Fp2 F = fp2_mul_small(E, 3);
  const Fp2 BmF = fp2_sub(B, F), BF = fp2_add(B, F);
  {  // round 3: E^2, A (B - F), (B + F)^2, c1 zl
    const Fp2 u = co_sel6(c, E, A, BF, c1, E, E), v = co_sel6(c, E, BmF, BF, co_fp_as_fp2(zl), E, E);
    co.put(CO_R3, fp2_dotk(kp(u, v)));
  }
  const Fp2 E2 = co.coef(CO_R3, 0), AX = co.coef(CO_R3, 1), BF2 = co.coef(CO_R3, 2);
  out.cz = co.coef(CO_R3, 3);
  t.x = fp2_dbl(AX);
  t.y = fp2_sub(BF2, fp2_mul_small(E2, 12));
  t.z = fp2_mul_small(BH, 4);
  out.d0 = fp2_neg(Hy);
  out.d3 = fp2_mul_small(Jx, 3);
  out.d4 = fp2_sub(E, B);
}
// addition step: T <- T + Q (bn254_curve.h::g2_add_step), Q = (qx, qy) affine
__device__ __forceinline__ void co_g2_add(const Coop& co, G2Proj& t, const Fp2& qx, const Fp2& qy, const Fp& xa, const Fp& ya, const Fp2& m1, const Fp& xl, const Fp2& m2,
                                       const Fp& xc, const Fp2& c1, const Fp& zl, CoLine& out) {
  const uint32_t c = co.c;
  {  // round 1: yQ Z, xQ Z, -, -, -, m1 xl
    const Fp2 u = co_sel6(c, qy, qx, qx, qx, qx, m1), v = co_sel6(c, t.z, t.z, t.z, t.z, t.z, co_fp_as_fp2(xl));
    co.put(CO_R1, fp2_dotk(kp(u, v)));
  }
  const Fp2 O = fp2_sub(t.y, co.coef(CO_R1, 0)), L = fp2_sub(t.x, co.coef(CO_R1, 1));
  out.s1 = co.coef(CO_R1, 5);
  {  // round 2: O^2, L^2, xQ O, L yQ, L ya, m2 xc
    const Fp2 u = co_sel6(c, O, L, qx, L, L, m2), v = co_sel6(c, O, L, O, qy, co_fp_as_fp2(ya), co_fp_as_fp2(xc));
    co.put(CO_R2, fp2_dotk(kp(u, v)));
  }
  const Fp2 Cc = co.coef(CO_R2, 0), D = co.coef(CO_R2, 1), xqO = co.coef(CO_R2, 2), Lyq = co.coef(CO_R2, 3);
  out.d0 = co.coef(CO_R2, 4);
  out.s2 = co.coef(CO_R2, 5);
  {  // round 3: L D, Z C, X D, O xa, c1 zl
    const Fp2 u = co_sel6(c, L, t.z, t.x, O, c1, O), v = co_sel6(c, D, Cc, D, co_fp_as_fp2(xa), co_fp_as_fp2(zl), O);
    co.put(CO_R3, fp2_dotk(kp(u, v)));
  }
  const Fp2 E = co.coef(CO_R3, 0), Fz = co.coef(CO_R3, 1), G = co.coef(CO_R3, 2), Ox = co.coef(CO_R3, 3);
  out.cz = co.coef(CO_R3, 4);
  const Fp2 H = fp2_sub(fp2_add(E, Fz), fp2_dbl(G));
  const Fp2 GmH = fp2_sub(G, H);
  {  // round 4: L H, (G - H) O, Y E, E Z     (round-1 slot reused: its values are in registers by now)
    const Fp2 u = co_sel6(c, L, GmH, t.y, E, E, E), v = co_sel6(c, H, O, E, t.z, E, E);
    co.put(CO_R1, fp2_dotk(kp(u, v)));
  }
  t.x = co.coef(CO_R1, 0);
  t.y = fp2_sub(co.coef(CO_R1, 1), co.coef(CO_R1, 2));
  t.z = co.coef(CO_R1, 3);
  out.d3 = fp2_neg(Ox);
  out.d4 = fp2_sub(xqO, Lyq);
}
// table entry -> affine point (80-byte entries, five 16-byte loads; as bn254_kernels.hip::msm_entry)
__device__ __forceinline__ G1Aff co_msm_entry(const int32_t* __restrict__ msm_tab, size_t idx) {
  const int4* e = (const int4*)(msm_tab + idx * MSM_ENTRY_DWORDS);
  int4 v0 = e[0], v1 = e[1], v2 = e[2], v3 = e[3], v4 = e[4];
  G1Aff q;
  q.x.v[0] = v0.x; q.x.v[1] = v0.y; q.x.v[2] = v0.z; q.x.v[3] = v0.w; q.x.v[4] = v1.x; q.x.v[5] = v1.y; q.x.v[6] = v1.z; q.x.v[7] = v1.w;
  q.x.v[8] = v2.x; q.y.v[0] = v2.y; q.y.v[1] = v2.z; q.y.v[2] = v2.w; q.y.v[3] = v3.x; q.y.v[4] = v3.y; q.y.v[5] = v3.z; q.y.v[6] = v3.w;
  q.y.v[7] = v4.x; q.y.v[8] = v4.y;
  return q;
}
// L = K0 + sum_i x_i K_i (groth16/verify.rs:53-63) by the six lanes of a proof: lane c adds the table entries of the byte-windows w = c, c + 6, ...
// (32 windows per input), then the six partial sums are added through LDS (two slots per projective point).  L stays PROJECTIVE: the line of the
// pair (L, g') is then scaled by Z_L, an Fp factor the final exponentiation removes, and no inversion is needed.
__device__ __noinline__ G1Proj co_public_input_msm(const Coop& co, const uint8_t* __restrict__ in /* this proof's inputs */, int n_public, bool use_inputs,
                                                   const int32_t* __restrict__ msm_tab, const int32_t* __restrict__ k0) {
  const uint32_t c = co.c;
  G1Proj acc = g1_identity();
  if (c == 0) { G1Aff K0; for (int l = 0; l < BN_NL; l++) { K0.x.v[l] = k0[l]; K0.y.v[l] = k0[BN_NL + l]; } acc = g1_from_affine(K0); }
  const int windows = use_inputs ? 32 * n_public : 0;
  for (int w = (int)c; w < windows; w += 6) {
    const int sidx = w >> 5, wi = w & 31;
    const uint32_t dig = in[(size_t)sidx * 32 + (31 - wi)];          // byte j of the big-endian scalar is window 31 - j
    G1Proj nxt = g1_add_mixed(acc, co_msm_entry(msm_tab, (size_t)(sidx * 32 + wi) * 255 + (dig ? dig - 1 : 0)));
    const bool take = dig != 0;
    acc.x = fp_select(take, nxt.x, acc.x); acc.y = fp_select(take, nxt.y, acc.y); acc.z = fp_select(take, nxt.z, acc.z);
  }
  // tree over the group: 0 += 3, 1 += 4, 2 += 5; then 0 += 1; 0 += 2
  auto publish = [&](const G1Proj& a) { Fp2 xy; xy.c0 = fp_reduce(a.x); xy.c1 = fp_reduce(a.y); Fp2 z0; z0.c0 = fp_reduce(a.z); z0.c1 = fp_zero(); co.put(CO_R1, xy); co.put(CO_R2, z0); };
  auto fetch = [&](uint32_t i) { G1Proj r; Fp2 xy = co.coef(CO_R1, i), z0 = co.coef(CO_R2, i); r.x = xy.c0; r.y = xy.c1; r.z = z0.c0; return r; };
  publish(acc);
  acc = g1_add(acc, fetch(c < 3 ? c + 3 : c));          // lanes 3..5 add their own value (result unused)
  publish(acc);
  acc = g1_add(acc, fetch(1));                           // lane 0: + lane 1
  acc = g1_add(acc, fetch(2));                           //         + lane 2
  publish(acc);
  return fetch(0);
}
__global__ void __launch_bounds__(64)
k_coop_miller_g16(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, const uint8_t* __restrict__ kinds, const int32_t* __restrict__ tab0,
                  const int32_t* __restrict__ tab1, const uint8_t* __restrict__ inputs, int n_public, int inputs_match_key,
                  const int32_t* __restrict__ msm_tab, const int32_t* __restrict__ k0, int l_from_ws, int fuse_final_exp) {
  CO_PROLOGUE();
  const int F = CO_SLOT(VE_F);
  // keys with many public inputs: L was computed by the wide MSM kernels (affine, in the workspace, identity flag in the status byte)
  G1Proj Lp;
  if (l_from_ws) { Lp.x = co_ws_ld(ws, n, pc, VE_LX); Lp.y = co_ws_ld(ws, n, pc, VE_LY); Lp.z = (st & BN254_ST_LINF) ? fp_zero() : fp_one(); }
  else Lp = co_public_input_msm(co, inputs + (size_t)pc * (size_t)n_public * 32, n_public, inputs_match_key != 0, msm_tab, k0);
  const bool l_inf = fp_is_zero(Lp.z);
  const Fp xl = Lp.x, yl = fp_select(l_inf, fp_one(), Lp.y), zl = Lp.z;
  { Fp2 one = c == 0 ? fp2_one() : fp2_zero(); co.put(F, one); }
  const Fp xa = co_ws_ld(ws, n, pc, VE_AX), ya = co_ws_ld(ws, n, pc, VE_AY);
  const Fp xc = co_ws_ld(ws, n, pc, VE_CX), yc = co_ws_ld(ws, n, pc, VE_CY);
  G2Aff q; q.x = co_ws_ld2(ws, n, pc, VE_B); q.y = co_ws_ld2(ws, n, pc, VE_B + 2);
  G2Proj t = g2_from_affine(q);
  for (int s = 0; s < BN_ATE_STEPS; s++) {
    const int kind = __builtin_amdgcn_readfirstlane((int)kinds[s]);
    if (kind == 0 && s != 0) co_sqr(co, F);
    const FixedLine l0 = co_line_entry(tab0 + (size_t)s * FIXED_LINE_DWORDS), l1 = co_line_entry(tab1 + (size_t)s * FIXED_LINE_DWORDS);
    CoLine ln;
    if (kind == 0) {
      co_g2_double(co, t, xa, ya, l0.m, xl, l1.m, xc, l0.c, zl, ln);
    } else {
      G2Aff b = q;
      if (kind == 2) b = g2_neg(q);
      else if (kind == 3) b = g2_psi_affine(q);
      else if (kind == 4) b = g2_neg(g2_psi2_affine(q));
      co_g2_add(co, t, b.x, b.y, xa, ya, l0.m, xl, l1.m, xc, l0.c, zl, ln);
    }
    co_mul_line_fp2(co, F, ln.d0, ln.d3, ln.d4);
    co_mul_line_fp(co, F, yl, ln.s1, ln.cz, l_inf);      // (Y_L + m X_L w + c Z_L w^3): the line at L scaled by Z_L
    co_mul_line_fp(co, F, yc, ln.s2, l1.c, false);
  }
  // the running point goes back to the workspace for the r-torsion test (k_g16_subgroup): lanes 0..2 of a group store X, Y, Z.  Not at
  // VE_T, which the result slot VE_S0 overlays: at COOP_T_ELEM
  if (pending && c < 3) {
    const Fp2 tc = c == 0 ? t.x : c == 1 ? t.y : t.z;
    co_ws_st(ws, n, p, COOP_T_ELEM + 2 * (int)c, tc.c0); co_ws_st(ws, n, p, COOP_T_ELEM + 2 * (int)c + 1, tc.c1);
  }
  if (fuse_final_exp) {
    CoopOps ops{co};
    vm_final_exp_program(ops);
    if (pending) { Fp2 r = co.own(CO_SLOT(VE_S0)); co_ws_st(ws, n, p, VE_S0 + 2 * (int)c, r.c0); co_ws_st(ws, n, p, VE_S0 + 2 * (int)c + 1, r.c1); }
  } else if (pending) {
    Fp2 r = co.own(F); co_ws_st(ws, n, p, VE_F + 2 * (int)c, r.c0); co_ws_st(ws, n, p, VE_F + 2 * (int)c + 1, r.c1);
  }
}

}  // namespace bn254

using namespace bn254;
static inline unsigned co_grid(size_t n, int waves_per_block) { return (unsigned)((n + 10 * waves_per_block - 1) / (10 * waves_per_block)); }
// device copy of the step-kind table (88 bytes), created on first use per device
static const uint8_t* co_kinds_dev(hipStream_t s) {
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
  (void)s;
  return dev[d];
}
hipError_t bn254_coop_final_exp(int32_t* ws, uint8_t* status, size_t n, hipStream_t s) {
  const size_t lds = (size_t)CO_WAVE_DWORDS * 4;
  (void)hipFuncSetAttribute((const void*)k_coop_final_exp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_coop_final_exp, dim3(co_grid(n, 1)), dim3(64), lds, s, ws, (uint32_t)n, (const uint8_t*)status);
  return hipGetLastError();
}
hipError_t bn254_coop_miller_fixed(int32_t* ws, uint8_t* status, size_t n, int n_pairs, const int32_t* tab0, const int32_t* tab1, const int32_t* tab2,
                                   int e_p0, int e_p1, int e_p2, int inf0, int inf1, int inf2, int fuse_final_exp, hipStream_t s) {
  const uint8_t* kinds = co_kinds_dev(s);
  if (!kinds) return hipErrorOutOfMemory;
  const size_t lds = (size_t)CO_WAVE_DWORDS * 4;
  (void)hipFuncSetAttribute((const void*)k_coop_miller_fixed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_coop_miller_fixed, dim3(co_grid(n, 1)), dim3(64), lds, s, ws, (uint32_t)n, (const uint8_t*)status, kinds, n_pairs, tab0, tab1, tab2,
                     e_p0, e_p1, e_p2, inf0, inf1, inf2, fuse_final_exp);
  return hipGetLastError();
}
hipError_t bn254_coop_miller_g16(int32_t* ws, uint8_t* status, size_t n, const int32_t* tab0, const int32_t* tab1, const uint8_t* inputs, int n_public,
                                 int inputs_match_key, const int32_t* msm_tab, const int32_t* k0, int l_from_ws, int fuse_final_exp, hipStream_t s) {
  const uint8_t* kinds = co_kinds_dev(s);
  if (!kinds) return hipErrorOutOfMemory;
  const size_t lds = (size_t)CO_WAVE_DWORDS * 4;
  (void)hipFuncSetAttribute((const void*)k_coop_miller_g16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_coop_miller_g16, dim3(co_grid(n, 1)), dim3(64), lds, s, ws, (uint32_t)n, (const uint8_t*)status, kinds, tab0, tab1, inputs, n_public,
                     inputs_match_key, msm_tab, k0, l_from_ws, fuse_final_exp);
  return hipGetLastError();
}
