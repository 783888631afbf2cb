// tools/repro/fr_from_be32_standalone.hip -- STAND-ALONE attempt at the wrong Fiat-Shamir challenge (see plonk_challenge_repro.hip beside it, which includes the product
// header and DOES fail when built with -DBN254_FR_MUL_INLINE=1 -DBN254_FR_NO_BARRIER): SHA-256 of a short message with the pending block in LDS (addressed by offset, as
// bn254_plonk.hpp::pl_lane_lds does), the 32 digest bytes gathered into four 64-bit limbs, one 8 x 32-bit-word CIOS Montgomery product with R^2 mod r -- all inlined.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/repro/fr_from_be32_standalone.hip -o fr_repro && ./fr_repro      (ROCm 7.2.0)
// prints the lanes whose product differs from the host's 4 x 64-bit form of the same product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
struct FrM { uint64_t l[4]; };
struct Ctx { uint64_t m[4], inv; FrM r2; };
__host__ __device__ inline bool geq_m(const Ctx& F, const FrM& a) { for (int i = 3; i >= 0; i--) { if (a.l[i] > F.m[i]) return true; if (a.l[i] < F.m[i]) return false; } return true; }
__host__ __device__ inline FrM sub_m(const Ctx& F, const FrM& a) { FrM r; uint64_t br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)a.l[i] - F.m[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } return r; }
__host__ __device__ __attribute__((always_inline)) inline FrM mul_w32(const Ctx& F, const FrM& a, const FrM& b) {
  uint32_t aw[8], bw[8], mw[8];
#pragma unroll
  for (int i = 0; i < 4; i++) { aw[2 * i] = (uint32_t)a.l[i]; aw[2 * i + 1] = (uint32_t)(a.l[i] >> 32); bw[2 * i] = (uint32_t)b.l[i]; bw[2 * i + 1] = (uint32_t)(b.l[i] >> 32); mw[2 * i] = (uint32_t)F.m[i]; mw[2 * i + 1] = (uint32_t)(F.m[i] >> 32); }
  const uint32_t ninv = (uint32_t)F.inv;
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { const uint64_t s = (uint64_t)aw[j] * bw[i] + t[j] + c; t[j] = (uint32_t)s; c = s >> 32; }
    const uint64_t s8 = (uint64_t)t[8] + c; t[8] = (uint32_t)s8; t[9] = (uint32_t)(s8 >> 32);
    const uint32_t q = t[0] * ninv;
    c = ((uint64_t)q * mw[0] + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) { const uint64_t s = (uint64_t)q * mw[j] + t[j] + c; t[j - 1] = (uint32_t)s; c = s >> 32; }
    const uint64_t s9 = (uint64_t)t[8] + c; t[7] = (uint32_t)s9; t[8] = t[9] + (uint32_t)(s9 >> 32);
  }
  FrM r;
#pragma unroll
  for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)t[2 * i] | ((uint64_t)t[2 * i + 1] << 32);
  if (t[8] || geq_m(F, r)) r = sub_m(F, r);
  return r;
}
inline FrM mul_w64(const Ctx& F, const FrM& a, const FrM& b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    unsigned __int128 c = 0;
    for (int j = 0; j < 4; j++) { c += (unsigned __int128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t q = t[0] * F.inv;
    c = (unsigned __int128)q * F.m[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) { c += (unsigned __int128)q * F.m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  FrM r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || geq_m(F, r)) r = sub_m(F, r);
  return r;
}
struct Sha256 {
  uint32_t h[8]; uint8_t buf[64]; uint64_t len; size_t fill;
  __host__ __device__ uint8_t* slot() {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    return (uint8_t*)(lds_u8*)(uintptr_t)(16u + threadIdx.x * 68u);      // the lane's 64-byte block in LDS, by offset
#else
    return buf;
#endif
  }
  __host__ __device__ Sha256() { const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u}; memcpy(h, iv, sizeof h); len = 0; fill = 0; }
  __host__ __device__ static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  __host__ __device__ void block(const uint8_t* p) {
    static const uint32_t K[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
      0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
      0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
      0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
      0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
      0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
      uint32_t wi;
      if (i < 16) wi = w[i];
      else { const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15]; const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3), s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10); wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1; w[i & 15] = wi; }
      uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + wi;
      uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  __host__ __device__ void update(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data; len += n;
    while (n) { size_t k = 64 - fill < n ? 64 - fill : n; uint8_t* bb = slot(); for (size_t q = 0; q < k; q++) bb[fill + q] = p[q]; fill += k; p += k; n -= k; if (fill == 64) { block(bb); fill = 0; } }
  }
  __host__ __device__ void finish(uint8_t out[32]) {
    uint64_t bits = len * 8; uint8_t pad = 0x80; update(&pad, 1); uint8_t z = 0;
    while (fill != 56) update(&z, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
  }
};
__host__ __device__ inline FrM limbs_from_be32(const uint8_t* b) { FrM raw; for (int i = 0; i < 4; i++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v = v << 8 | b[(3 - i) * 8 + j]; raw.l[i] = v; } return raw; }
__global__ void __launch_bounds__(64) k(const Ctx* F, const uint8_t* msg, size_t n, FrM* out, uint8_t* digest) {
  extern __shared__ uint8_t dyn[];
  if (threadIdx.x == 0) *(uint32_t*)dyn = 68u;
  __syncthreads();
  uint8_t dg[32];
#if defined(NO_SHA)   // without the hash: eight words from memory, written out big-endian as Sha256::finish does (this form computes the RIGHT product)
  for (int i = 0; i < 8; i++) { const uint32_t h = ((const uint32_t*)msg)[i] + (uint32_t)n * 0x9e3779b9u; dg[4 * i] = (uint8_t)(h >> 24); dg[4 * i + 1] = (uint8_t)(h >> 16); dg[4 * i + 2] = (uint8_t)(h >> 8); dg[4 * i + 3] = (uint8_t)h; }
#else
  Sha256 s; s.update("gamma", 5); s.update(msg, n); s.finish(dg);
#endif
  out[threadIdx.x] = mul_w32(*F, limbs_from_be32(dg), F->r2);
  for (int i = 0; i < 32; i++) digest[32 * threadIdx.x + i] = dg[i];
}
int main() {
  Ctx F;
  const uint64_t R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  memcpy(F.m, R, 32);
  uint64_t x = 1; for (int i = 0; i < 6; i++) x *= 2 - F.m[0] * x; F.inv = ~x + 1;
  FrM t = {{1, 0, 0, 0}};
  for (int i = 0; i < 512; i++) { FrM r; uint64_t c = 0; for (int k = 0; k < 4; k++) { uint64_t v = t.l[k]; r.l[k] = (v << 1) | c; c = v >> 63; } if (c || geq_m(F, r)) r = sub_m(F, r); t = r; }
  F.r2 = t;
  uint8_t msg[704]; for (int i = 0; i < 704; i++) msg[i] = (uint8_t)(i * 131 + 7);
  uint8_t hd[32];
#if defined(NO_SHA)
  for (int i = 0; i < 8; i++) { uint32_t h; memcpy(&h, msg + 4 * i, 4); h += (uint32_t)sizeof msg * 0x9e3779b9u; hd[4 * i] = (uint8_t)(h >> 24); hd[4 * i + 1] = (uint8_t)(h >> 16); hd[4 * i + 2] = (uint8_t)(h >> 8); hd[4 * i + 3] = (uint8_t)h; }
#else
  { Sha256 s; s.update("gamma", 5); s.update(msg, sizeof msg); s.finish(hd); }
#endif
  const FrM want = mul_w64(F, limbs_from_be32(hd), F.r2), want32 = mul_w32(F, limbs_from_be32(hd), F.r2);      // the SAME mul_w32 source compiled for the host
  printf("host: the 32-bit-word form %s the 64-bit form\n", memcmp(&want, &want32, 32) == 0 ? "equals" : "DIFFERS FROM");
  Ctx* dF; uint8_t *dm, *dd; FrM* dout;
  if (hipMalloc(&dF, sizeof F) != hipSuccess) { printf("no device\n"); return 2; }
  hipMalloc(&dm, sizeof msg); hipMalloc(&dd, 64 * 32); hipMalloc(&dout, 64 * sizeof(FrM));
  hipMemcpy(dF, &F, sizeof F, hipMemcpyHostToDevice); hipMemcpy(dm, msg, sizeof msg, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 16 + 64 * 68, 0, dF, dm, sizeof msg, dout, dd);
  FrM o[64]; uint8_t dg[64 * 32];
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost); hipMemcpy(dg, dd, sizeof dg, hipMemcpyDeviceToHost);
  int bad = 0, dbad = 0;
  for (int i = 0; i < 64; i++) { bad += memcmp(&o[i], &want, 32) != 0; dbad += memcmp(dg + 32 * i, hd, 32) != 0; }
  printf("digest differs in %d lanes; product differs in %d lanes; device %016llx%016llx.. host %016llx%016llx..\n", dbad, bad, (unsigned long long)o[0].l[3], (unsigned long long)o[0].l[2],
         (unsigned long long)want.l[3], (unsigned long long)want.l[2]);
  return bad ? 1 : 0;
}
