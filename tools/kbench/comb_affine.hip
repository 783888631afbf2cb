// comb_affine.hip -- probe for BASELINE configs[4] (1024 public inputs x 4096 proofs): the public-input sum of one lane -- 16 inputs x 20 comb
// columns = 320 table points -- as it is computed today (complete mixed additions into a projective accumulator, 20 shared doublings: the loop of k_g16_msm_partial_comb)
// against IN-LANE BATCHED-AFFINE accumulation: per column the 16 points are added as a binary tree of affine additions; the additions of one tree level of all 20 columns
// (160, 80, 40, 20) are independent, so each level shares ONE field inversion through prefix products (Montgomery's trick inside the lane).  A lane has neither the
// registers nor the LDS for 160 prefix products, so they -- and every level's points -- live in a workspace slice in global memory, laid out [slot][digit][lane] so that
// every access of a wavefront is one contiguous run.  Then 20 doublings + 20 complete additions (Horner over the column sums).
// What the probe measures is the price of that staging against the multiply-adds it saves (6 products + a 160th of an inversion per addition against the complete
// formula's 1815 multiply-adds).  Exceptional cases (equal / opposite operands, absent digits) are NOT handled: a build would add them; they cost little.
// Table, digits and lane count are those of the real launch: 1024 x 8192 entries of 80 bytes (671 MB), 4096 proofs x 64 chunks of 16 inputs = 262 144 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../snark-bn254-verifier_amd/csrc/bn254_devws.h"
using namespace bn254;

#define COLS 20
#define PER 16
#define TEETH 13
__device__ __forceinline__ uint32_t digit_of(uint32_t lane, int col, int s) {           // stands in for the u16 digit array (2 bytes per addition: not what is being measured)
  uint32_t x = lane * 0x9E3779B9u + (uint32_t)(col * 1024 + s) * 0x85EBCA6Bu; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
  return (x & 8191u) | 1u;
}
__device__ __forceinline__ G1Aff entry_at(const int32_t* __restrict__ tab, int s_global, uint32_t idx) { return msm_entry(tab, ((size_t)s_global << TEETH) + idx); }

__global__ void k_fill(int32_t* tab, size_t entries) {
  size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= entries) return;
  uint64_t x = e * 0x9E3779B97F4A7C15ull + 777;
  for (int l = 0; l < 20; l++) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    int32_t d = (int32_t)(x & 0x1fffffff) - (1 << 28);
    if (l == 8 || l == 17) d = (int32_t)(x & 0x3fffff);
    if (l >= 18) d = 0;
    tab[e * 20 + l] = d;
  }
}

// ---- today's loop (bn254_kernels.hip::k_g16_msm_partial_comb without the digit array) ------------------------------------------------------------------------------
#ifndef LBW
#define LBW 2
#endif
__global__ void __launch_bounds__(256, LBW) k_cur(const int32_t* __restrict__ tab, uint32_t n, int chunks, int32_t* __restrict__ part) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= n * (uint32_t)chunks) return;
  const uint32_t c = g / n, i = g - c * n;
  const int s_begin = (int)(c * PER), s_end = s_begin + PER;
  G1Proj acc = g1_identity();
  int col1 = COLS - 1, s1 = s_begin;
  auto advance = [&](int& col, int& s) { if (++s == s_end) { s = s_begin; col--; } };
  G1Aff e_cur = entry_at(tab, s1, digit_of(g, col1, s1));
  int col0 = col1, s0 = s1;
  advance(col1, s1);
  while (col0 >= 0) {
    G1Aff e_nxt = e_cur;
    if (col1 >= 0) e_nxt = entry_at(tab, s1, digit_of(g, col1, s1));
    if (s0 == s_begin && col0 != COLS - 1) acc = g1_dbl(acc);
    acc = g1_add_mixed(acc, e_cur);
    col0 = col1; s0 = s1; e_cur = e_nxt;
    if (col1 >= 0) advance(col1, s1);
  }
  int32_t* o = part + (size_t)c * 27 * n + i;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { o[(size_t)l * n] = acc.x.v[l]; o[(size_t)(9 + l) * n] = acc.y.v[l]; o[(size_t)(18 + l) * n] = acc.z.v[l]; }
}

// ---- batched affine -------------------------------------------------------------------------------------------------------------------------------------------------
// workspace slice of a lane: slot q, digit l at scratch[(q * 9 + l) * lanes + lane].  Slots: [0, 160) prefix products; [160, 160 + 2 * 160) the points of the level being
// written (x, y), [480, 480 + 2 * 160) the points of the level being read (ping-pong).
struct Slice {
  int32_t* base; size_t lanes; uint32_t lane;
  __device__ __forceinline__ Fp ld(int q) const { Fp r;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) r.v[l] = base[((size_t)q * 9 + l) * lanes + lane];
    BN_SETB(r, 1.01, 0.5); return r; }
  __device__ __forceinline__ void st(int q, const Fp& a) const { const Fp c = fp_reduce(a);
#pragma unroll
    for (int l = 0; l < BN_NL; l++) base[((size_t)q * 9 + l) * lanes + lane] = c.v[l]; }
};
#define SL_PRE 0
#define SL_A 160
#define SL_B 480
#define SL_COUNT 800
__global__ void __launch_bounds__(256, 2) k_aff(const int32_t* __restrict__ tab, uint32_t n, int chunks, int32_t* __restrict__ part, int32_t* __restrict__ scratch) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lanes = n * (uint32_t)chunks;
  if (g >= lanes) return;
  const uint32_t c = g / n, i = g - c * n;
  const int s_begin = (int)(c * PER);
  const Slice w{scratch, lanes, g};
  // level 1: addition k = (column k / 8, pair k % 8) of table points 2p and 2p + 1
  auto leaf = [&](int k, int which) -> G1Aff { const int col = k >> 3, s = s_begin + 2 * (k & 7) + which; return entry_at(tab, s, digit_of(g, col, s)); };
  int wr = SL_A, rd = SL_B;
  for (int level = 0, m = 160; level < 4; level++, m >>= 1) {
    // forward: prefix products of the denominators x2 - x1
    Fp run = fp_one();
    for (int k = 0; k < m; k++) {
      Fp x1, x2;
      if (level == 0) { x1 = leaf(k, 0).x; x2 = leaf(k, 1).x; } else { x1 = w.ld(rd + 2 * (2 * k)); x2 = w.ld(rd + 2 * (2 * k + 1)); }
      w.st(SL_PRE + k, run);
      run = fp_mul(run, fp_sub(x2, x1));
    }
    Fp inv = fp_inv(run);
    // backward: one inverse each, the addition, the result to the level's output slots
    for (int k = m - 1; k >= 0; k--) {
      G1Aff p, q;
      if (level == 0) { p = leaf(k, 0); q = leaf(k, 1); }
      else { p.x = w.ld(rd + 2 * (2 * k)); p.y = w.ld(rd + 2 * (2 * k) + 1); q.x = w.ld(rd + 2 * (2 * k + 1)); q.y = w.ld(rd + 2 * (2 * k + 1) + 1); }
      const Fp d = fp_sub(q.x, p.x);
      const Fp idk = fp_mul(inv, w.ld(SL_PRE + k));
      inv = fp_mul(inv, d);
      const Fp lam = fp_mul(fp_sub(q.y, p.y), idk);
      const Fp x3 = fp_sub(fp_sub(fp_sqr(lam), p.x), q.x);
      const Fp y3 = fp_sub(fp_mul(lam, fp_sub(p.x, x3)), p.y);
      w.st(wr + 2 * k, x3); w.st(wr + 2 * k + 1, y3);
    }
    const int t = wr; wr = rd; rd = t;
  }
  // Horner over the 20 column sums (now in `rd`): acc = 2 acc + column
  G1Proj acc = g1_identity();
  for (int col = COLS - 1; col >= 0; col--) {
    if (col != COLS - 1) acc = g1_dbl(acc);
    G1Aff e; e.x = w.ld(rd + 2 * col); e.y = w.ld(rd + 2 * col + 1);
    acc = g1_add_mixed(acc, e);
  }
  int32_t* o = part + (size_t)c * 27 * n + i;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) { o[(size_t)l * n] = acc.x.v[l]; o[(size_t)(9 + l) * n] = acc.y.v[l]; o[(size_t)(18 + l) * n] = acc.z.v[l]; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const uint32_t n = 4096; const int chunks = 64;
  const size_t entries = (size_t)1024 << TEETH, lanes = (size_t)n * chunks;
  int32_t *tab, *part, *scratch;
  CK(hipMalloc((void**)&tab, entries * 80));
  CK(hipMalloc((void**)&part, lanes * 27 * 4));
  CK(hipMalloc((void**)&scratch, lanes * (size_t)SL_COUNT * 36));
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, 0, tab, entries);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((lanes + 255) / 256);
  for (int which = 0; which < 2; which++) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0, 0));
      if (which == 0) hipLaunchKernelGGL(k_cur, dim3(grid), dim3(256), 0, 0, tab, n, chunks, part);
      else hipLaunchKernelGGL(k_aff, dim3(grid), dim3(256), 0, 0, tab, n, chunks, part, scratch);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s: %.3f ms for %zu lanes (4096 proofs x 64 chunks of 16 inputs, 20 columns)%s\n", which == 0 ? "complete mixed additions (today)" : "batched affine, prefix products in a workspace slice",
             ms, lanes, rep == 0 ? "  [first launch]" : "");
    }
  }
  CK(hipGetLastError());
  return 0;
}
