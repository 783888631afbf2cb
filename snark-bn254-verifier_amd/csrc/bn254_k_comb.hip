// bn254_k_comb.hip -- the fixed-base tables of a key (the multiples of the points that enter a multi-scalar multiplication with per-proof scalars), built ON THE DEVICE that
// will use them.  Until round 5 the host built them (bn254_host.hpp::build_comb_table / build_window_table: 8191 / 8160 additions and a batch inversion per point -- 2.2 s on 8
// threads for a key with 1024 public inputs, 0.18 s for one with 16) and every device got a copy (671 MB for 1024 inputs).  The work is independent point additions: here it is
// three kinds of launches, milliseconds, and the host keeps 72 bytes per point.
//   Two table forms, both arrays of affine points of MSM_ENTRY_DWORDS dwords:
//   comb     (keys with more than 16 public inputs; read by k_g16_msm_partial_comb): entry[idx] = sum over the set bits t of idx of 2^(20 t) P, idx = 1 .. 8191 (13 teeth 20
//            bits apart); entry 0 is never read and holds P.  One block of 8192 entries per point.
//   windows  (byte windows: keys with up to 16 inputs, the key points of the PlonK MSMs, read by msm_entry(tab, (point * 32 + w) * 255 + d - 1)): entry[w][d - 1] =
//            d 2^(8 w) P, d = 1 .. 255, w = 0 .. 31.  32 blocks of 256 construction entries per point (d = 0 is a placeholder that is not written out), teeth 2^k P, k = 0 .. 255.
//   In both forms a block's entry e is the sum over the set bits t of e of the block's tooth t, so one construction serves:
//   k_tab_teeth       lane = point: the teeth as projective points (digit planes), `spacing` doublings apart
//   k_tab_normalize   lane = 8 consecutive construction entries: one inversion for the eight (prefix products in registers), affine digits out -- used for the teeth (18 dwords
//                     each) and, last, for the table itself
//   k_tab_level       level t, lane = (block, offset < 2^t): entry[2^t + offset] = entry[offset] + tooth_t of the block (projective, digit planes [27][blocks * entries])
#include <hip/hip_runtime.h>
#include "bn254_devws.h"
#include "bn254_kernels.h"

namespace bn254 {

#define TAB_GROUP 8

__device__ __forceinline__ G1Aff tab_ld_aff(const int32_t* __restrict__ p) {
  G1Aff a;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { a.x.v[l] = p[l]; a.y.v[l] = p[BN_NL + l]; }
  BN_SETB(a.x, 1.01, 0.5); BN_SETB(a.y, 1.01, 0.5);
  return a;
}
__device__ __forceinline__ void tab_st_aff(int32_t* p, const G1Aff& a) {     // the digits bn254_host.hpp::fp_to_limbs writes
  const Fp x = fp_reduce(fp_norm(a.x)), y = fp_reduce(fp_norm(a.y));
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { p[l] = x.v[l]; p[BN_NL + l] = y.v[l]; }
}
// projective points of the construction: digit l of coordinate c at plane[(c * 9 + l) * total + e] -- consecutive lanes, consecutive entries
__device__ __forceinline__ G1Proj tab_ld_proj(const int32_t* __restrict__ plane, size_t total, size_t e) {
  G1Proj p;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { p.x.v[l] = plane[(size_t)l * total + e]; p.y.v[l] = plane[(size_t)(BN_NL + l) * total + e]; p.z.v[l] = plane[(size_t)(2 * BN_NL + l) * total + e]; }
  BN_SETB(p.x, 1.01, 0.5); BN_SETB(p.y, 1.01, 0.5); BN_SETB(p.z, 1.01, 0.5);
  return p;
}
__device__ __forceinline__ void tab_st_proj(int32_t* plane, size_t total, size_t e, const G1Proj& p) {
  const Fp x = fp_reduce(p.x), y = fp_reduce(p.y), z = fp_reduce(p.z);
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { plane[(size_t)l * total + e] = x.v[l]; plane[(size_t)(BN_NL + l) * total + e] = y.v[l]; plane[(size_t)(2 * BN_NL + l) * total + e] = z.v[l]; }
}

// teeth of point i: tooth k = 2^(spacing k) P_i, k < teeth, at construction index i * teeth + k of `plane` (total = points * teeth)
__global__ void __launch_bounds__(64) k_tab_teeth(const int32_t* __restrict__ pts, uint32_t points, int teeth, int spacing, int32_t* __restrict__ plane) {
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i >= points) return;
  const size_t total = (size_t)points * (size_t)teeth;
  G1Proj t = g1_from_affine(tab_ld_aff(pts + (size_t)i * 2 * BN_NL));
  for (int k = 0; k < teeth; k++) {
    tab_st_proj(plane, total, (size_t)i * teeth + k, t);
    if (k + 1 < teeth) for (int d = 0; d < spacing; d++) t = g1_dbl(t);
  }
}
// level t of every block: entry[2^t + off] = entry[off] + tooth (off = 0: the tooth itself); teeth_aff: 18 dwords per tooth, tooth `level` of block b at b * teeth_per_block + level
__global__ void __launch_bounds__(256) k_tab_level(const int32_t* __restrict__ teeth_aff, int32_t* __restrict__ plane, size_t blocks, int log2_entries, int teeth_per_block, int level) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t per = (size_t)1 << level;
  if (g >= blocks * per) return;
  const size_t b = g >> level, off = g & (per - 1), total = blocks << log2_entries, base = b << log2_entries;
  const G1Aff tooth = tab_ld_aff(teeth_aff + (b * (size_t)teeth_per_block + level) * 2 * BN_NL);
  G1Proj r = g1_from_affine(tooth);
  if (off != 0) r = g1_add_mixed(tab_ld_proj(plane, total, base + off), tooth);
  tab_st_proj(plane, total, base + per + off, r);
  if (level == 0) tab_st_proj(plane, total, base, r);                    // entry 0 of the block (never read): a finite point, so that the shared inversion below stays non-zero
}
// construction entries e = 8 g .. 8 g + 7 to affine.  out_stride: dwords per output entry (2 x 9 digits, the rest zero).  skip_zero = 0: entry e goes to out[e]; skip_zero = 1
// (byte windows): entry d of block b goes to out[b * (entries - 1) + d - 1], d = 0 is not written.
__global__ void __launch_bounds__(256) k_tab_normalize(const int32_t* __restrict__ plane, size_t total, int32_t* __restrict__ out, int out_stride, int log2_entries, int skip_zero) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t e0 = g * TAB_GROUP;
  if (e0 >= total) return;
  // Montgomery's trick over the group: pre[k] = z_0 ... z_(k-1); one inversion; entry k gets inv(z_k) = inv(z_0 ... z_k) * pre[k]
  Fp pre[TAB_GROUP];
  Fp acc = fp_one();
#pragma unroll
  for (int k = 0; k < TAB_GROUP; k++) {
    Fp z = fp_one();
    if (e0 + k < total) {
#pragma unroll
      for (int l = 0; l < BN_NL; l++) z.v[l] = plane[(size_t)(2 * BN_NL + l) * total + e0 + k];
      BN_SETB(z, 1.01, 0.5);
    }
    pre[k] = acc;
    acc = fp_mul(acc, z);
  }
  Fp inv = fp_inv(acc);
  const size_t mask = ((size_t)1 << log2_entries) - 1;
#pragma unroll
  for (int k = TAB_GROUP - 1; k >= 0; k--) {
    if (e0 + k >= total) continue;                                         // (its z was taken as one: nothing to undo)
    const size_t e = e0 + k;
    const G1Proj p = tab_ld_proj(plane, total, e);
    const Fp zi = fp_mul(inv, pre[k]);
    inv = fp_mul(inv, p.z);
    if (skip_zero && (e & mask) == 0) continue;
    G1Aff a; a.x = fp_mul(p.x, zi); a.y = fp_mul(p.y, zi);
    const size_t oe = skip_zero ? (e >> log2_entries) * mask + (e & mask) - 1 : e;
    int32_t* o = out + oe * (size_t)out_stride;
    tab_st_aff(o, a);
    for (int l = 2 * BN_NL; l < out_stride; l++) o[l] = 0;
  }
}

}  // namespace bn254

using namespace bn254;
// pts: `points` affine points (18 dwords each) in device memory.  form 0: comb tables, points * 8192 entries; form 1: byte-window tables, points * 32 * 255 entries; form 2:
// MSM_FW_BITS-bit-window tables, points * MSM_FW_WINDOWS * (2^bits - 1) entries (all of MSM_ENTRY_DWORDS dwords).  Scratch: teeth_plane = 27 * points * T dwords, teeth_aff = 18 * points * T dwords (T = 13 | 256), plane = 27 * points * E dwords (E = 8192 | 32 * 256).
// form 2: MSM_FW_BITS-bit windows (the key points of the PlonK MSMs, bn254_fw.h): MSM_FW_WINDOWS blocks of 2^bits construction entries per point, `bits` teeth per
// block (tooth k = 2^k P, k < windows * bits), read by msm_entry(tab, (point * windows + w) * (2^bits - 1) + d - 1)
size_t bn254_tab_build_teeth(int form) { return form == 0 ? (size_t)G16_COMB_TEETH : form == 1 ? 256 : (size_t)MSM_FW_WINDOWS * MSM_FW_BITS; }
size_t bn254_tab_build_entries(int form) { return form == 0 ? ((size_t)1 << G16_COMB_TEETH) : form == 1 ? (size_t)32 * 256 : (size_t)MSM_FW_WINDOWS << MSM_FW_BITS; }
size_t bn254_tab_build_out_entries(int form) { return form == 0 ? ((size_t)1 << G16_COMB_TEETH) : form == 1 ? (size_t)32 * 255 : (size_t)MSM_FW_WINDOWS * MSM_FW_ENTRIES; }
hipError_t bn254_launch_tab_build(int form, const int32_t* pts, uint32_t points, int32_t* table, int32_t* teeth_plane, int32_t* teeth_aff, int32_t* plane, hipStream_t s) {
  if (points == 0) return hipSuccess;
  const int teeth = (int)bn254_tab_build_teeth(form), spacing = form == 0 ? G16_COMB_COLS : 1;
  const int log2_entries = form == 0 ? G16_COMB_TEETH : form == 1 ? 8 : MSM_FW_BITS, teeth_per_block = log2_entries;
  const size_t blocks = form == 0 ? (size_t)points : (size_t)points * (form == 1 ? 32 : MSM_FW_WINDOWS);
  hipLaunchKernelGGL(k_tab_teeth, dim3((points + 63) / 64), dim3(64), 0, s, pts, points, teeth, spacing, teeth_plane);
  const size_t n_teeth = (size_t)points * teeth;
  hipLaunchKernelGGL(k_tab_normalize, dim3((unsigned)(((n_teeth + TAB_GROUP - 1) / TAB_GROUP + 255) / 256)), dim3(256), 0, s, (const int32_t*)teeth_plane, n_teeth, teeth_aff, 2 * BN_NL, 0, 0);
  for (int level = 0; level < log2_entries; level++) {
    const size_t lanes = blocks << level;
    hipLaunchKernelGGL(k_tab_level, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, (const int32_t*)teeth_aff, plane, blocks, log2_entries, teeth_per_block, level);
  }
  const size_t total = blocks << log2_entries;
  hipLaunchKernelGGL(k_tab_normalize, dim3((unsigned)((total / TAB_GROUP + 255) / 256)), dim3(256), 0, s, (const int32_t*)plane, total, table, (int)MSM_ENTRY_DWORDS, log2_entries, form == 0 ? 0 : 1);
  return hipGetLastError();
}
