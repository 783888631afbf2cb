// tools/coop/coop_bench.hip -- prototype of the north-star layout ("one pairing per wavefront, limbs staged in LDS") on the Fp12 microkernel
// that SURVEY.md section 7 asks for: the dependent chain  f <- f^2 ; f <- f * line  of a Miller loop, as
//   (a) k_coop_chain: 6 lanes per proof, lane c holds the Fp2 coefficient k_c of f = sum k_c w^c; every product needs other lanes' coefficients,
//       which travel through LDS ([slot][lane][20 dwords] image per wave, ds_read_b128); 10 proofs per wavefront (lanes 60..63 idle);
//   (b) k_lane_chain: the product's layout, one proof per lane, f in registers (reference for the result and for throughput).
// Both use the product's arithmetic headers.  Prints per-step times for several batch sizes and checks (a) == (b) bit for bit.
//   hipcc -O3 --offload-arch=gfx950 -I../../snark-bn254-verifier_amd/csrc coop_bench.hip -o coop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bn254_pairing.h"

using namespace bn254;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// deterministic inputs: proof p, value index v -> Montgomery form of a pseudo-random field element
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ Fp rnd_fp(uint32_t p, uint32_t v) {
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = mix(p * 977u + v * 131u + (uint32_t)i * 0x9e3779b9u + 12345u);
  w[7] &= 0x0fffffffu;   // < 2^252 < p
  return fp_from_words(w);
}
__device__ __forceinline__ void out_fp(uint32_t* o, const Fp& a) { fp_to_words(o, a); }

// ---------------------------------------------------------------- (b) one proof per lane
__global__ void __launch_bounds__(256) k_lane_chain(uint32_t* out, uint32_t n, int iters) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n) return;
  Fp12 f;
  Fp2* kc[6] = {&K0(f), &K1(f), &K2(f), &K3(f), &K4(f), &K5(f)};
  for (int c = 0; c < 6; c++) { kc[c]->c0 = rnd_fp(p, 2 * c); kc[c]->c1 = rnd_fp(p, 2 * c + 1); }
  Fp d0 = rnd_fp(p, 20);
  Fp2 d3, d4; d3.c0 = rnd_fp(p, 21); d3.c1 = rnd_fp(p, 22); d4.c0 = rnd_fp(p, 23); d4.c1 = rnd_fp(p, 24);
  Fp2 x4 = fp2_mul_xi(d4);
  for (int it = 0; it < iters; it++) {
    f = fp12_sqr(f);
    f = fp12_mul_by_034_fp(f, d0, d3, d4, x4);
  }
  for (int c = 0; c < 6; c++) { out_fp(out + ((size_t)p * 12 + 2 * c) * 8, kc[c]->c0); out_fp(out + ((size_t)p * 12 + 2 * c + 1) * 8, kc[c]->c1); }
}

// ---------------------------------------------------------------- (a) six lanes per proof
#define COOP_STRIDE 20   // dwords per Fp2 image: 18 + 2 pad; 20 * lane mod 64 puts 16 consecutive lanes on 16 disjoint 4-bank slots (ds_read_b128)
struct CoopLds {
  int32_t* base;   // this wave's image: [2 slots][64 lanes][COOP_STRIDE]
  __device__ __forceinline__ void put(int slot, uint32_t lane, const Fp2& a) const {
    int4* q = (int4*)(base + ((size_t)slot * 64 + lane) * COOP_STRIDE);
    q[0] = make_int4(a.c0.v[0], a.c0.v[1], a.c0.v[2], a.c0.v[3]);
    q[1] = make_int4(a.c0.v[4], a.c0.v[5], a.c0.v[6], a.c0.v[7]);
    q[2] = make_int4(a.c0.v[8], a.c1.v[0], a.c1.v[1], a.c1.v[2]);
    q[3] = make_int4(a.c1.v[3], a.c1.v[4], a.c1.v[5], a.c1.v[6]);
    q[4] = make_int4(a.c1.v[7], a.c1.v[8], 0, 0);
  }
  __device__ __forceinline__ Fp2 get(uint32_t slot_lane /* slot * 64 + lane */) const {
    const int4* q = (const int4*)(base + (size_t)slot_lane * COOP_STRIDE);
    int4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4];
    Fp2 a;
    a.c0.v[0] = v0.x; a.c0.v[1] = v0.y; a.c0.v[2] = v0.z; a.c0.v[3] = v0.w; a.c0.v[4] = v1.x; a.c0.v[5] = v1.y; a.c0.v[6] = v1.z; a.c0.v[7] = v1.w;
    a.c0.v[8] = v2.x; a.c1.v[0] = v2.y; a.c1.v[1] = v2.z; a.c1.v[2] = v2.w; a.c1.v[3] = v3.x; a.c1.v[4] = v3.y; a.c1.v[5] = v3.z; a.c1.v[6] = v3.w;
    a.c1.v[7] = v4.x; a.c1.v[8] = v4.y;
    return a;
  }
};
__device__ __forceinline__ Fp2 fp2_scale_small(const Fp2& a, int32_t w) {  // w in {0, 1, 2}: digit-wise (balanced digits need no carry for that)
  Fp2 r;
#pragma unroll
  for (int i = 0; i < BN_NL; i++) { r.c0.v[i] = a.c0.v[i] * w; r.c1.v[i] = a.c1.v[i] * w; }
  return r;
}
// squaring, coefficient j of  f^2  (bn254_tower.h::fp12_sqr):  up to four products  w * k_a * (xi?) k_b  per lane
//   r0 = k0 k0 + 2 k1 xk5 + 2 k2 xk4 + k3 xk3     r1 = 2 k0 k1 + 2 k2 xk5 + 2 k3 xk4             r2 = 2 k0 k2 + k1 k1 + 2 k3 xk5 + k4 xk4
//   r3 = 2 k0 k3 + 2 k1 k2 + 2 k4 xk5             r4 = 2 k0 k4 + 2 k1 k3 + k2 k2 + k5 xk5         r5 = 2 k0 k5 + 2 k1 k4 + 2 k2 k3
__constant__ int8_t SQ_A[6][4] = {{0, 1, 2, 3}, {0, 2, 3, 0}, {0, 1, 3, 4}, {0, 1, 4, 0}, {0, 1, 2, 5}, {0, 1, 2, 0}};
__constant__ int8_t SQ_B[6][4] = {{0, 5, 4, 3}, {1, 5, 4, 0}, {2, 1, 5, 4}, {3, 2, 5, 0}, {4, 3, 2, 5}, {5, 4, 3, 0}};
__constant__ int8_t SQ_X[6][4] = {{0, 1, 1, 1}, {0, 1, 1, 0}, {0, 0, 1, 1}, {0, 0, 1, 0}, {0, 0, 0, 1}, {0, 0, 0, 0}};   // second operand from the xi image
__constant__ int8_t SQ_W[6][4] = {{1, 2, 2, 1}, {2, 2, 2, 0}, {2, 1, 2, 1}, {2, 2, 2, 0}, {2, 2, 1, 1}, {2, 2, 2, 0}};
// line product (bn254_tower.h::fp12_mul_by_034_fp): r_j = d0 k_j + D3 (xi?) k_{j-1} + D4 (xi?) k_{j-3}
__global__ void __launch_bounds__(256) k_coop_chain(uint32_t* out, uint32_t n, int iters) {
  __shared__ __attribute__((aligned(16))) int32_t lds[4 * 2 * 64 * COOP_STRIDE];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t pl = lane / 6, c = lane - pl * 6;
  const bool act = lane < 60;
  const uint32_t p = (blockIdx.x * 4u + wave) * 10u + pl;
  const bool live = act && p < n;
  const uint32_t pp_ = live ? p : 0;
  CoopLds L{lds + (size_t)wave * 2 * 64 * COOP_STRIDE};
  const uint32_t g0 = act ? pl * 6 : 54;   // first lane of this proof's group (idle lanes 60..63 read group 9: in-bounds, results unused)
  Fp2 k; k.c0 = rnd_fp(pp_, 2 * c); k.c1 = rnd_fp(pp_, 2 * c + 1);
  const Fp d0 = rnd_fp(pp_, 20);
  Fp2 d3, d4; d3.c0 = rnd_fp(pp_, 21); d3.c1 = rnd_fp(pp_, 22); d4.c0 = rnd_fp(pp_, 23); d4.c1 = rnd_fp(pp_, 24);
  // per-lane operand plan of the squaring
  uint32_t sa[4], sb[4]; int32_t sw[4];
#pragma unroll
  for (int s = 0; s < 4; s++) { sa[s] = g0 + SQ_A[c][s]; sb[s] = (uint32_t)SQ_X[c][s] * 64u + g0 + SQ_B[c][s]; sw[s] = act ? SQ_W[c][s] : 0; }
  // line product: second operands k_{j-1}, k_{j-3} (indices mod 6), from the xi image when the index wraps
  const uint32_t l1 = (c >= 1 ? 0u : 64u) + g0 + (c + 5) % 6, l3 = (c >= 3 ? 0u : 64u) + g0 + (c + 3) % 6;
  for (int it = 0; it < iters; it++) {
    // ---- f <- f^2
    L.put(0, lane, k); L.put(1, lane, fp2_mul_xi(k));
    {
      Fp2 x0 = fp2_scale_small(L.get(sa[0]), sw[0]), x1 = fp2_scale_small(L.get(sa[1]), sw[1]), x2 = fp2_scale_small(L.get(sa[2]), sw[2]), x3 = fp2_scale_small(L.get(sa[3]), sw[3]);
      Fp2 y1 = L.get(sb[1]), y2 = L.get(sb[2]), y3 = L.get(sb[3]);
      k = fp2_dotk(kp(x0, k), kp(x1, y1), kp(x2, y2), kp(x3, y3));   // slot 0: the second operand is the lane's own coefficient
    }
    // ---- f <- f * (d0 + d3 w + d4 w^3)
    L.put(0, lane, k); L.put(1, lane, fp2_mul_xi(k));
    {
      Fp2 y1 = L.get(l1), y3 = L.get(l3);
      k = fp2_dotk(kfp(k, d0), kp(d3, y1), kp(d4, y3));
    }
  }
  if (live) { out_fp(out + ((size_t)p * 12 + 2 * c) * 8, k.c0); out_fp(out + ((size_t)p * 12 + 2 * c + 1) * 8, k.c1); }
}

static double time_kernel(void (*launch)(uint32_t*, uint32_t, int), uint32_t* out, uint32_t n, int iters, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(out, n, iters); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; r++) launch(out, n, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}
static void launch_lane(uint32_t* out, uint32_t n, int iters) { hipLaunchKernelGGL(k_lane_chain, dim3((n + 255) / 256), dim3(256), 0, 0, out, n, iters); }
static void launch_coop(uint32_t* out, uint32_t n, int iters) { hipLaunchKernelGGL(k_coop_chain, dim3((n + 39) / 40), dim3(256), 0, 0, out, n, iters); }

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 64;
  const uint32_t nmax = 1u << 18;
  uint32_t *o1, *o2;
  CK(hipMalloc((void**)&o1, (size_t)nmax * 96 * 4)); CK(hipMalloc((void**)&o2, (size_t)nmax * 96 * 4));
  // correctness: the two layouts must agree bit for bit
  {
    const uint32_t n = 4096 + 37;
    CK(hipMemset(o1, 0, (size_t)n * 96 * 4)); CK(hipMemset(o2, 0xff, (size_t)n * 96 * 4));
    launch_lane(o1, n, 7); launch_coop(o2, n, 7); CK(hipDeviceSynchronize());
    std::vector<uint32_t> h1((size_t)n * 96), h2((size_t)n * 96);
    CK(hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < h1.size(); i++) bad += h1[i] != h2[i];
    printf("parity (n = %u, 7 steps): %zu of %zu words differ -> %s\n", n, bad, h1.size(), bad ? "MISMATCH" : "identical");
    if (bad) return 1;
  }
  printf("chain of %d steps (f <- f^2 ; f <- f * line), time per step per batch and proof-steps per second\n", iters);
  printf("%10s %14s %14s %16s %16s\n", "proofs", "lane us/step", "coop us/step", "lane Msteps/s", "coop Msteps/s");
  for (uint32_t n : {1024u, 4096u, 16384u, 65536u, 262144u}) {
    double tl = time_kernel(launch_lane, o1, n, iters, 3), tc = time_kernel(launch_coop, o2, n, iters, 3);
    printf("%10u %14.2f %14.2f %16.1f %16.1f\n", n, tl * 1e3 / iters, tc * 1e3 / iters, n * (double)iters / tl / 1e3, n * (double)iters / tc / 1e3);
  }
  return 0;
}
