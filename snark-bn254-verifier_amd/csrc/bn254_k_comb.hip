// bn254_k_comb.hip -- the comb tables of a key with many public inputs (BASELINE configs[4]: 1024 inputs), built ON THE DEVICE that will use them.
//   Until round 5 the host built them (bn254_host.hpp::build_comb_table: 8191 additions and a batch inversion per input, 2.2 s on 16 threads for 1024 inputs) and every
//   device got a 671 MB copy.  The work is 8.4 M independent point additions: here it is three kinds of launches, a few milliseconds, and the host keeps 72 bytes per input.
//   Table of input i (groth16/verify.rs:53-63 multiplies K[i + 1] by a 256-bit scalar x): entry[idx] = sum over the set bits t of idx of 2^(20 t) K[i + 1], idx = 1 .. 8191,
//   as an affine point of MSM_ENTRY_DWORDS dwords -- what k_g16_msm_partial_comb (bn254_kernels.hip) reads; entry 0 is never read and holds the base point.
//   k_comb_teeth      lane = input: the 13 teeth 2^(20 t) K, each taken to affine (its own inversion: 13 per lane, the launch is latency-bound either way)
//   k_comb_level      level t = 0 .. 12, lane = (input, offset < 2^t): entry[2^t + offset] = entry[offset] + tooth_t (projective, digit planes [27][inputs * 8192])
//   k_comb_normalize  lane = 8 consecutive entries: one inversion for the eight (prefix products in registers), affine digits to the table
#include <hip/hip_runtime.h>
#include "bn254_devws.h"
#include "bn254_kernels.h"

namespace bn254 {

#define COMB_ENTRIES (1u << G16_COMB_TEETH)
#define COMB_GROUP 8

__device__ __forceinline__ G1Aff comb_ld_aff(const int32_t* __restrict__ p) {
  G1Aff a;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { a.x.v[l] = p[l]; a.y.v[l] = p[BN_NL + l]; }
  BN_SETB(a.x, 1.01, 0.5); BN_SETB(a.y, 1.01, 0.5);
  return a;
}
__device__ __forceinline__ void comb_st_aff(int32_t* p, const G1Aff& a) {
  const Fp x = fp_reduce(fp_norm(a.x)), y = fp_reduce(fp_norm(a.y));
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { p[l] = x.v[l]; p[BN_NL + l] = y.v[l]; }
}
// projective points of the construction: digit plane l of coordinate c at plane[(c * 9 + l) * total + e] -- consecutive lanes, consecutive entries
__device__ __forceinline__ G1Proj comb_ld_proj(const int32_t* __restrict__ plane, size_t total, size_t e) {
  G1Proj p;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { p.x.v[l] = plane[(size_t)l * total + e]; p.y.v[l] = plane[(size_t)(BN_NL + l) * total + e]; p.z.v[l] = plane[(size_t)(2 * BN_NL + l) * total + e]; }
  BN_SETB(p.x, 1.01, 0.5); BN_SETB(p.y, 1.01, 0.5); BN_SETB(p.z, 1.01, 0.5);
  return p;
}
__device__ __forceinline__ void comb_st_proj(int32_t* plane, size_t total, size_t e, const G1Proj& p) {
  const Fp x = fp_reduce(p.x), y = fp_reduce(p.y), z = fp_reduce(p.z);
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { plane[(size_t)l * total + e] = x.v[l]; plane[(size_t)(BN_NL + l) * total + e] = y.v[l]; plane[(size_t)(2 * BN_NL + l) * total + e] = z.v[l]; }
}

__global__ void __launch_bounds__(64) k_comb_teeth(const int32_t* __restrict__ kpts, uint32_t nb, int32_t* __restrict__ teeth) {
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i >= nb) return;
  G1Proj t = g1_from_affine(comb_ld_aff(kpts + (size_t)i * 2 * BN_NL));
  for (int k = 0; k < G16_COMB_TEETH; k++) {
    comb_st_aff(teeth + ((size_t)i * G16_COMB_TEETH + k) * 2 * BN_NL, g1_to_affine(t));
    for (int d = 0; d < G16_COMB_COLS; d++) t = g1_dbl(t);
  }
}
__global__ void __launch_bounds__(256) k_comb_level(const int32_t* __restrict__ teeth, int32_t* __restrict__ plane, uint32_t nb, int level) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t per = (size_t)1 << level;
  if (g >= (size_t)nb * per) return;
  const size_t i = g >> level, off = g & (per - 1), total = (size_t)nb * COMB_ENTRIES;
  const G1Aff tooth = comb_ld_aff(teeth + (i * G16_COMB_TEETH + level) * 2 * BN_NL);
  G1Proj r = g1_from_affine(tooth);
  if (off != 0) r = g1_add_mixed(comb_ld_proj(plane, total, i * COMB_ENTRIES + off), tooth);
  comb_st_proj(plane, total, i * COMB_ENTRIES + per + off, r);
  if (level == 0) comb_st_proj(plane, total, i * COMB_ENTRIES, r);        // entry 0 (never read): a finite point, so that the shared inversion below stays non-zero
}
__global__ void __launch_bounds__(256) k_comb_normalize(const int32_t* __restrict__ plane, size_t total, int32_t* __restrict__ out) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t e0 = g * COMB_GROUP;
  if (e0 >= total) return;                                               // total is a multiple of COMB_GROUP (8192 entries per input)
  // Montgomery's trick over the group: pre[k] = z_0 ... z_(k-1); one inversion; entry k gets inv(z_k) = inv(z_0 ... z_k) * pre[k]
  Fp pre[COMB_GROUP];
  Fp acc = fp_one();
#pragma unroll
  for (int k = 0; k < COMB_GROUP; k++) {
    Fp z;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) z.v[l] = plane[(size_t)(2 * BN_NL + l) * total + e0 + k];
    BN_SETB(z, 1.01, 0.5);
    pre[k] = acc;
    acc = fp_mul(acc, z);
  }
  Fp inv = fp_inv(acc);
#pragma unroll
  for (int k = COMB_GROUP - 1; k >= 0; k--) {
    const G1Proj p = comb_ld_proj(plane, total, e0 + k);
    const Fp zi = fp_mul(inv, pre[k]);
    inv = fp_mul(inv, p.z);
    G1Aff a; a.x = fp_mul(p.x, zi); a.y = fp_mul(p.y, zi);
    int32_t* o = out + (e0 + k) * MSM_ENTRY_DWORDS;
    comb_st_aff(o, a);
    o[2 * BN_NL] = 0; o[2 * BN_NL + 1] = 0;
  }
}

}  // namespace bn254

using namespace bn254;
// kpts: nb affine points (18 dwords each) in device memory; table: nb * 8192 * MSM_ENTRY_DWORDS dwords; scratch_teeth: nb * 13 * 18 dwords; scratch_plane: 27 * nb * 8192 dwords
hipError_t bn254_launch_comb_build(const int32_t* kpts, uint32_t nb, int32_t* table, int32_t* scratch_teeth, int32_t* scratch_plane, hipStream_t s) {
  if (nb == 0) return hipSuccess;
  hipLaunchKernelGGL(k_comb_teeth, dim3((nb + 63) / 64), dim3(64), 0, s, kpts, nb, scratch_teeth);
  for (int level = 0; level < G16_COMB_TEETH; level++) {
    const size_t lanes = (size_t)nb << level;
    hipLaunchKernelGGL(k_comb_level, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, (const int32_t*)scratch_teeth, scratch_plane, nb, level);
  }
  const size_t total = (size_t)nb * COMB_ENTRIES, groups = total / COMB_GROUP;
  hipLaunchKernelGGL(k_comb_normalize, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, (const int32_t*)scratch_plane, total, table);
  return hipGetLastError();
}
