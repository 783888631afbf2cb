// bn254_k_msm.hip -- G1 multi-scalar multiplications of the PlonK path (plonk/verify.rs:253-284: the linearised-polynomial digest; plonk/kzg.rs:74-85,
// 161-175: the folded digests and quotients of the KZG check), one ROW of the plan per lane (bn254_msm.h).
//   k_g1_msm_rows    lane g = row * n_pad + item: the row's share of its item's sum -- a variable term over a range of joint bit positions (two-bit windows,
//                    15-entry table in the lane's scratch), a unit term, a slice of the fixed-base byte windows of the key-side terms -- as one projective point
//   k_g1_sum_affine  per item and sum: the rows added up (complete additions), to affine, into the workspace or out as canonical words
// Rows are wave-uniform (n_pad is a multiple of 64), so the kinds of work never diverge inside a wavefront.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "bn254_devws.h"
#include "bn254_msm.h"

namespace bn254 {

// a lane's 16 x 28 dwords of window-table scratch (global memory: one contiguous 108-byte read per step, issued before the doublings)
struct DevGlvTab {
  int32_t* base;
  size_t slot_stride;      // dwords between the same lane's tables in consecutive scratch rows (joint rows keep one table per term)
  __device__ __forceinline__ DevGlvTab slot(int j) const { return DevGlvTab{base + (size_t)j * slot_stride, slot_stride}; }
  __device__ __forceinline__ void put(int i, const G1Proj& p) const {
    const Fp x = fp_reduce(p.x), y = fp_reduce(p.y), z = fp_reduce(p.z);
    int4* q = (int4*)(base + i * 28);
    q[0] = make_int4(x.v[0], x.v[1], x.v[2], x.v[3]); q[1] = make_int4(x.v[4], x.v[5], x.v[6], x.v[7]); q[2] = make_int4(x.v[8], y.v[0], y.v[1], y.v[2]);
    q[3] = make_int4(y.v[3], y.v[4], y.v[5], y.v[6]); q[4] = make_int4(y.v[7], y.v[8], z.v[0], z.v[1]); q[5] = make_int4(z.v[2], z.v[3], z.v[4], z.v[5]);
    q[6] = make_int4(z.v[6], z.v[7], z.v[8], 0);
  }
  __device__ __forceinline__ G1Proj get(uint32_t i) const {
    const int4* q = (const int4*)(base + i * 28);
    const int4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6];
    G1Proj p;
    p.x.v[0] = a.x; p.x.v[1] = a.y; p.x.v[2] = a.z; p.x.v[3] = a.w; p.x.v[4] = b.x; p.x.v[5] = b.y; p.x.v[6] = b.z; p.x.v[7] = b.w; p.x.v[8] = c.x;
    p.y.v[0] = c.y; p.y.v[1] = c.z; p.y.v[2] = c.w; p.y.v[3] = d.x; p.y.v[4] = d.y; p.y.v[5] = d.z; p.y.v[6] = d.w; p.y.v[7] = e.x; p.y.v[8] = e.y;
    p.z.v[0] = e.z; p.z.v[1] = e.w; p.z.v[2] = f.x; p.z.v[3] = f.y; p.z.v[4] = f.z; p.z.v[5] = f.w; p.z.v[6] = g.x; p.z.v[7] = g.y; p.z.v[8] = g.z;
    BN_SETB(p.x, 1.01, 0.5); BN_SETB(p.y, 1.01, 0.5); BN_SETB(p.z, 1.01, 0.5);
    return p;
  }
  __device__ __forceinline__ void fence() const { __threadfence_block(); }
};
static_assert(G1_GLV_TAB_BYTES_PER_LANE == 16 * 28 * 4, "scratch layout");

// the item's terms (MsmTerm: 18 point digits + 8 scalar words each), its flag bytes and the key's window tables
struct DevMsmIO {
  const int32_t* terms_i; const uint8_t* flags_i; const int32_t* tabs;
  __device__ __forceinline__ void term(int t, G1Aff& P, uint32_t k1[4], uint32_t k2[4], uint32_t& fl) const {
    const int32_t* e = terms_i + (size_t)t * MSM_TERM_DWORDS;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { P.x.v[l] = e[l]; P.y.v[l] = e[BN_NL + l]; }
    BN_SETB(P.x, 1.0, 0.5); BN_SETB(P.y, 1.0, 0.5);
#pragma unroll
    for (int k = 0; k < 4; k++) { k1[k] = (uint32_t)e[18 + k]; k2[k] = (uint32_t)e[22 + k]; }
    fl = flags_i[t];
  }
  __device__ __forceinline__ uint32_t kword(int t, int half, int w) const { return (uint32_t)terms_i[(size_t)t * MSM_TERM_DWORDS + 18 + 4 * half + w]; }
  // digit w of the scalar (eight little-endian words): bits MSM_FW_BITS w .. MSM_FW_BITS (w + 1) - 1, zero beyond bit 255
  __device__ __forceinline__ uint32_t scalar_digit(int t, int w) const {
    const uint32_t* k = (const uint32_t*)(terms_i + (size_t)t * MSM_TERM_DWORDS + 18);
    const int bit = MSM_FW_BITS * w, i = bit >> 5, sh = bit & 31;
    const uint64_t two = (uint64_t)k[i] | ((uint64_t)(i + 1 < 8 ? k[i + 1 < 8 ? i + 1 : 7] : 0u) << 32);
    return (uint32_t)(two >> sh) & MSM_FW_ENTRIES;
  }
  __device__ __forceinline__ G1Aff entry(int tab, int w, uint32_t d) const {
    return msm_entry(tabs + (size_t)tab * ((size_t)MSM_FW_WINDOWS * MSM_FW_ENTRIES * MSM_ENTRY_DWORDS), (size_t)w * MSM_FW_ENTRIES + d);
  }
};

// The plan travels by value in the kernel arguments and is read THERE (scalar loads from the kernarg segment, indexed by the wave-uniform row): indexing
// the by-value copy would make the compiler spill the whole struct to scratch memory first.
__global__ void __launch_bounds__(256, 2)
k_g1_msm_rows(MsmPlan plan_arg, const int32_t* __restrict__ terms, const uint8_t* __restrict__ flags, uint32_t n, uint32_t n_pad, int n_terms,
              int32_t* __restrict__ part, int32_t* __restrict__ glv_tab, const int32_t* __restrict__ tabs) {
  typedef __attribute__((address_space(4))) const MsmPlan KernargPlan;
  const MsmPlan& plan = *(const MsmPlan*)(KernargPlan*)__builtin_amdgcn_kernarg_segment_ptr();     // plan_arg is the first argument: offset 0
  (void)plan_arg;
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  const int r = __builtin_amdgcn_readfirstlane((int)(g / n_pad));        // n_pad is a multiple of 64: uniform over the wavefront
  if (r >= plan.n_rows) return;
  const uint32_t i = g - (uint32_t)r * n_pad;
  const bool live = i < n;
  const uint32_t ii = live ? i : n - 1;
  DevMsmIO io{terms + (size_t)ii * (size_t)n_terms * MSM_TERM_DWORDS, flags + (size_t)ii * (size_t)n_terms, tabs};
  DevGlvTab tab{glv_tab + ((size_t)plan.row[r].glv_slot * n_pad + i) * (size_t)(G1_GLV_TAB_BYTES_PER_LANE / 4), (size_t)n_pad * (size_t)(G1_GLV_TAB_BYTES_PER_LANE / 4)};
  const G1Proj acc = msm_row_eval(plan, r, io, tab);
  if (!live) return;
  int32_t* o = part + (size_t)r * 27 * n + i;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { o[(size_t)l * n] = acc.x.v[l]; o[(size_t)(9 + l) * n] = acc.y.v[l]; o[(size_t)(18 + l) * n] = acc.z.v[l]; }
}

// per item: the sum of rows [first, first + count) of `part`, to affine.  out_words != nullptr: 16 little-endian words (x | y, canonical) and a flag byte
// (1 = identity) per item.  Otherwise the point goes to workspace elements (e_x, e_x + 1) as (x, y) or (0, 1) for the identity, whose flag bit `inf_bit`
// is OR-ed into the (pending) status byte.  count_b > 0: TWO sums per item in one launch (PlonK: P0 and P1 of the KZG check): the lanes from
// round_up(n, 64) items on form the second one (a wavefront never mixes the two) -- the launch lasts as long as one lane's chain, whatever the number of sums.
// lpi_log2 = 2: FOUR lanes per item and sum (small batches, where the launch is one lane's chain of additions): lane q of a quad adds rows q, q + 4, ...,
// two butterfly steps over the quad (27 dwords each way through the cross-lane network) leave the total in every lane, lane 0 writes: 14 rows are 4 + 2
// additions deep instead of 14.
__device__ __forceinline__ G1Proj g1_shfl_xor(const G1Proj& p, int mask) {
  G1Proj r;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { r.x.v[l] = __shfl_xor(p.x.v[l], mask); r.y.v[l] = __shfl_xor(p.y.v[l], mask); r.z.v[l] = __shfl_xor(p.z.v[l], mask); }
  BN_SETB(r.x, BN_VB(p.x), BN_LBD(p.x)); BN_SETB(r.y, BN_VB(p.y), BN_LBD(p.y)); BN_SETB(r.z, BN_VB(p.z), BN_LBD(p.z));
  return r;
}
__global__ void __launch_bounds__(256, 2)
k_g1_sum_affine(const int32_t* __restrict__ part, int first, int count, uint32_t n, uint32_t* __restrict__ out_words, uint8_t* __restrict__ out_inf, int32_t* ws,
                uint8_t* __restrict__ status, int e_x, int inf_bit, int first_b, int count_b, int e_x_b, int inf_bit_b, int lpi_log2) {
  const uint32_t gl = blockIdx.x * 256u + threadIdx.x;
  const uint32_t g = gl >> lpi_log2;                       // item-lane
  const int q = (int)(gl & ((1u << lpi_log2) - 1u)), lpi = 1 << lpi_log2;
  const uint32_t n_pad = (n + 63u) & ~63u;
  const bool second_sum = count_b > 0 && g >= n_pad;       // n_pad << lpi_log2 is a multiple of 64: uniform over a wavefront
  const uint32_t i = second_sum ? g - n_pad : (g < n ? g : 0xffffffffu);
  if (second_sum) { first = first_b; count = count_b; e_x = e_x_b; inf_bit = inf_bit_b; }
  const uint32_t ii = i < n ? i : n - 1;
  G1Proj L = g1_identity();
  for (int cc = q; cc < count; cc += lpi) {
    const int32_t* o = part + (size_t)(first + cc) * 27 * n + ii;
    G1Proj t;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { t.x.v[l] = o[(size_t)l * n]; t.y.v[l] = o[(size_t)(9 + l) * n]; t.z.v[l] = o[(size_t)(18 + l) * n]; }
    BN_SETB(t.x, 3.0, 0.5); BN_SETB(t.y, 3.0, 0.5); BN_SETB(t.z, 3.0, 0.5);
    L = g1_add(L, t);
  }
  for (int m = 1; m < lpi; m <<= 1) L = g1_add(L, g1_shfl_xor(L, m));
  bool l_inf = g1_is_identity(L);
  G1Aff La = g1_to_affine(L);
  const bool writer = i < n && q == 0;
  if (out_words) {
    if (writer) {
      uint32_t wx[8], wy[8];
      fp_to_words(wx, La.x); fp_to_words(wy, La.y);
#pragma unroll
      for (int k = 0; k < 8; k++) { out_words[(size_t)i * 16 + k] = wx[k]; out_words[(size_t)i * 16 + 8 + k] = wy[k]; }
      out_inf[i] = l_inf ? 1 : 0;
    }
  } else {
    DevWs w(ws, n, writer ? i : DEAD_LANE);
    La.y = fp_select(l_inf, fp_one(), La.y);
    w.st(e_x, La.x); w.st(e_x + 1, La.y);
    // the two sums of an item may both flag their point: different bits of the same status byte -> an atomic OR
    if (writer && l_inf) { if (status[i] & BN254_ST_PENDING) atomicOr((unsigned int*)(status + (i & ~3u)), (unsigned int)inf_bit << (8 * (i & 3u))); }
  }
}

// ---- BN254_FLAG_RLC on the PlonK path: the pairing checks of a pass batched over the 64 proofs of a wavefront ---------------------------------------------------
// Every pending proof i arrives with P0_i, P1_i already multiplied by its random weight (k_plonk_stage2).  One wavefront per group of 64 consecutive proofs: each
// lane takes its proof's point (the identity if the proof is decided already or its point is the identity), six butterfly steps of complete additions leave the
// group's sum in every lane, lane 0 writes it -- affine, in the pairing stage's layout -- to the GROUP workspace (n_groups "proofs") with the group's status byte:
// pending (+ identity flags), or decided (ACCEPT) when none of its proofs is pending.
__global__ void __launch_bounds__(256, 2)
k_plonk_group_sums(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int32_t* grp_ws, uint32_t n_groups, uint8_t* __restrict__ grp_status,
                   int e_p0, int inf0, int e_p1, int inf1) {
  const uint32_t gl = blockIdx.x * 256u + threadIdx.x;
  const uint32_t g = gl >> 6, lane = gl & 63u, i = g * 64u + lane;
  const uint8_t st = i < n ? status[i] : (uint8_t)0;
  const bool pend = (st & BN254_ST_PENDING) != 0;
  DevWs w(ws, n, i < n ? i : DEAD_LANE);
  DevWs wg(grp_ws, n_groups, (lane == 0 && g < n_groups) ? g : DEAD_LANE);
  uint32_t flags = 0;
#pragma unroll 1
  for (int which = 0; which < 2; which++) {
    G1Aff p; p.x = w.ld(which ? e_p1 : e_p0); p.y = w.ld((which ? e_p1 : e_p0) + 1);
    const bool skip = !pend || (st & (which ? inf1 : inf0)) != 0;
    G1Proj L = g1_from_affine(p);
    const G1Proj id = g1_identity();
    L.x = fp_select(skip, id.x, L.x); L.y = fp_select(skip, id.y, L.y); L.z = fp_select(skip, id.z, L.z);
    for (int m = 1; m < 64; m <<= 1) L = g1_add(L, g1_shfl_xor(L, m));
    const bool l_inf = g1_is_identity(L);
    G1Aff La = g1_to_affine(L);
    La.y = fp_select(l_inf, fp_one(), La.y);
    wg.st(which ? e_p1 : e_p0, La.x); wg.st((which ? e_p1 : e_p0) + 1, La.y);
    if (l_inf) flags |= (uint32_t)(which ? inf1 : inf0);
  }
  const bool any = __builtin_amdgcn_ballot_w64(pend) != 0;
  if (lane == 0 && g < n_groups) grp_status[g] = any ? (uint8_t)(BN254_ST_PENDING | flags) : (uint8_t)BN254_ST_ACCEPT;
}
// ... and back: a pending proof whose group passed is accepted; the proofs of a failed group stay pending (the exact per-proof check then runs on exactly
// those wavefronts) and the group counts once in *n_failed
__global__ void __launch_bounds__(256, 2)
k_plonk_group_scatter(uint8_t* __restrict__ status, uint32_t n, const uint8_t* __restrict__ grp_status, uint32_t* __restrict__ n_failed) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint8_t gs = grp_status[i >> 6];
  if (status[i] & BN254_ST_PENDING) {
    if (gs == BN254_ST_ACCEPT) status[i] = BN254_ST_ACCEPT;
  }
  if ((i & 63u) == 0 && gs != BN254_ST_ACCEPT) atomicAdd(n_failed, 1u);
}

}  // namespace bn254

using namespace bn254;
hipError_t bn254_launch_plonk_group_sums(int32_t* ws, const uint8_t* status, size_t n, int32_t* grp_ws, uint8_t* grp_status, int e_p0, int inf0, int e_p1, int inf1, hipStream_t s) {
  const size_t groups = (n + 63) / 64;
  hipLaunchKernelGGL(k_plonk_group_sums, dim3((unsigned)((groups * 64 + 255) / 256)), dim3(256), 0, s, ws, (uint32_t)n, status, grp_ws, (uint32_t)groups, grp_status, e_p0, inf0, e_p1, inf1);
  return hipGetLastError();
}
hipError_t bn254_launch_plonk_group_scatter(uint8_t* status, size_t n, const uint8_t* grp_status, uint32_t* n_failed, hipStream_t s) {
  hipLaunchKernelGGL(k_plonk_group_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, status, (uint32_t)n, grp_status, n_failed);
  return hipGetLastError();
}
// lanes of window-table scratch (G1_GLV_TAB_BYTES_PER_LANE each) a launch of this plan over n items needs
size_t bn254_g1_msm_scratch_lanes(const MsmPlan& plan, size_t n) { return (size_t)plan.n_var_rows * ((n + 63) & ~(size_t)63); }
// part: plan.n_rows * 27 * n dwords; glv_tab: bn254_g1_msm_scratch_lanes(plan, n) lanes; tabs: the key's window tables
hipError_t bn254_launch_g1_msm_rows(const MsmPlan& plan, const int32_t* terms, const uint8_t* flags, size_t n, int n_terms, int32_t* part, int32_t* glv_tab,
                                    const int32_t* tabs, hipStream_t s) {
  const size_t n_pad = (n + 63) & ~(size_t)63;
  const size_t lanes = (size_t)plan.n_rows * n_pad;
  hipLaunchKernelGGL(k_g1_msm_rows, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, plan, terms, flags, (uint32_t)n, (uint32_t)n_pad, n_terms, part, glv_tab, tabs);
  return hipGetLastError();
}
// the rows of sum 0 (and of sum 1 when the plan has two) added up: out_words / out_inf (canonical words for the host or the next stage) or the workspace
hipError_t bn254_launch_g1_sum_rows(const MsmPlan& plan, const int32_t* part, size_t n, uint32_t* out_words, uint8_t* out_inf, int32_t* ws, uint8_t* status, int e_x, int inf_bit,
                                    int e_x_b, int inf_bit_b, hipStream_t s) {
  const bool two = plan.count[1] > 0;
  // four lanes per item while the launch stays within one wavefront per SIMD (there it lasts as long as one lane's chain of additions); one lane per item beyond
  const int lpi_log2 = ((two ? 2 : 1) * n * 4 <= 65536 && plan.count[0] >= 4) ? 2 : 0;
  const size_t lanes = (two ? ((n + 63) & ~(size_t)63) + n : n) << lpi_log2;
  hipLaunchKernelGGL(k_g1_sum_affine, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, part, plan.first[0], plan.count[0], (uint32_t)n, out_words, out_inf, ws, status, e_x,
                     inf_bit, plan.first[1], two ? plan.count[1] : 0, e_x_b, inf_bit_b, lpi_log2);
  return hipGetLastError();
}
