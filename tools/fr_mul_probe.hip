// tools/fr_mul_probe.hip -- the scalar-field arithmetic of the PlonK device stages (csrc/bn254_plonk.hpp) in isolation on the GPU: the two forms of the
// Montgomery product (8 x 32-bit words: what the device stages run; 4 x 64-bit limbs through __int128: what the host runs) and the constant-time inversion,
// each compared bit for bit with the host's result of the same source, for Fr and Fp, on edge values (0, 1, m - 1, m, m + 1, 2^256 - 1, R^2, R) and random
// operands, full and partial wavefronts.  Build and run (GPU box):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I snark-bn254-verifier_amd/csrc tools/fr_mul_probe.hip -o /tmp/fr_mul_probe && /tmp/fr_mul_probe
// Round 3 reported that an earlier 32-bit form was "bit-exact in isolation" but gave a wrong opening check inside k_plonk_stage1 when the compiler chose the
// inlining; round 4 re-derived the form (mul_w32), checked the host build of this header under UBSan / ASan (clean) and runs the whole PlonK GPU suite
// through it.
#define BN254_PLONK_DEVICE_TU 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "bn254_plonk.hpp"
using namespace bn254host;
__global__ void kmul(FrM* o, const FrM* a, const FrM* b, int n, int field, int form) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const FrCtx& F = field ? fp64_ctx().F : fr_ctx();
  o[i] = form == 32 ? F.mul_w32(a[i], b[i]) : F.mul_w64(a[i], b[i]);
}
// a dependent chain, as the stages use the product: x <- x * x * b, 40 times (the compiler inlines and schedules it as it likes)
__global__ void kchain(FrM* o, const FrM* a, const FrM* b, int n, int field) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const FrCtx& F = field ? fp64_ctx().F : fr_ctx();
  FrM x = F.mul(a[i], F.r2);
  for (int k = 0; k < 40; k++) x = F.add(F.mul(F.mul(x, x), b[i]), F.one);
  o[i] = x;
}
// the shape of a Fiat-Shamir challenge (transcript.rs:68-107): eight 32-bit state words written out as 32 big-endian bytes (Sha256::finish), read back byte by
// byte into four 64-bit limbs and reduced with one Montgomery product (FrCtx::from_be32) -- everything inlined, as inside PlonkStage1::a
template <int FORM>
__global__ void kdigest(FrM* o, const uint32_t* hw, int n) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const FrCtx& F = fr_ctx();
  uint32_t h[8];
  for (int k = 0; k < 8; k++) h[k] = hw[8 * i + k] + (uint32_t)k * 0x9e3779b9u;          // (an addition in front, as the last step of the compression function)
  uint8_t out[32];
  for (int k = 0; k < 8; k++) { out[4 * k] = (uint8_t)(h[k] >> 24); out[4 * k + 1] = (uint8_t)(h[k] >> 16); out[4 * k + 2] = (uint8_t)(h[k] >> 8); out[4 * k + 3] = (uint8_t)h[k]; }
  FrM raw;
  for (int q = 0; q < 4; q++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v = v << 8 | out[(3 - q) * 8 + j]; raw.l[q] = v; }
  o[i] = FORM == 32 ? F.mul_w32(raw, F.r2) : F.mul_w64(raw, F.r2);
}
__global__ void kinv(FrM* o, const FrM* a, int n, int field) { int i = blockIdx.x * 64 + threadIdx.x; if (i < n) o[i] = (field ? fp64_ctx().F : fr_ctx()).inverse(a[i]); }
int main() {
  const FrCtx& F = fr_ctx(); const Fp64Ctx& C = fp64_ctx();
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_plonk_fr_raw), &F, sizeof(FrCtx)) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_plonk_fp64_raw), &C, sizeof(Fp64Ctx)) != hipSuccess) { printf("no device\n"); return 2; }
  const int n = 8192; std::mt19937_64 g(7);
  std::vector<FrM> a(n), b(n), o(n);
  int total_bad = 0;
  FrM *da, *db, *dd; hipMalloc(&da, n * 32); hipMalloc(&db, n * 32); hipMalloc(&dd, n * 32);
  for (int field = 0; field < 2; field++) {
    const FrCtx& X = field ? C.F : F;
    for (int i = 0; i < n; i++) { a[i] = {{g(), g(), g(), g()}}; b[i] = {{g(), g(), g(), g() >> 3}}; if (X.geq_m(b[i])) b[i] = X.sub_m(b[i]); if (i % 5 == 0) a[i].l[3] >>= 3; }
    FrM sp[8] = {{{0, 0, 0, 0}}, {{1, 0, 0, 0}}, {{X.m[0] - 1, X.m[1], X.m[2], X.m[3]}}, {{X.m[0], X.m[1], X.m[2], X.m[3]}}, {{X.m[0] + 1, X.m[1], X.m[2], X.m[3]}}, {{~0ull, ~0ull, ~0ull, ~0ull}}, X.r2, X.one};
    for (int i = 0; i < 64; i++) { a[i] = sp[i / 8]; b[i] = sp[i % 8]; if (X.geq_m(b[i])) b[i] = X.sub_m(b[i]); }
    hipMemcpy(da, a.data(), n * 32, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 32, hipMemcpyHostToDevice);
    for (int form : {32, 64}) {
      kmul<<<n / 64, 64>>>(dd, da, db, n, field, form); hipMemcpy(o.data(), dd, n * 32, hipMemcpyDeviceToHost);
      int bad = 0; for (int i = 0; i < n; i++) { FrM r = X.mul_w64(a[i], b[i]); if (memcmp(&r, &o[i], 32)) bad++; }
      printf("field %d product, %d-bit form on the device: %d mismatches of %d\n", field, form, bad, n); total_bad += bad;
    }
    kchain<<<n / 64, 64>>>(dd, da, db, n, field); hipMemcpy(o.data(), dd, n * 32, hipMemcpyDeviceToHost);
    { int bad = 0; for (int i = 0; i < n; i++) { FrM x = X.mul_w64(a[i], X.r2); for (int k = 0; k < 40; k++) x = X.add(X.mul_w64(X.mul_w64(x, x), b[i]), X.one); if (memcmp(&x, &o[i], 32)) bad++; }
      printf("field %d dependent chain (device default form, inlined): %d mismatches of %d\n", field, bad, n); total_bad += bad; }
    for (int m2 : {1, 3, 63}) {
      hipMemset(dd, 0, n * 32);
      kmul<<<1, 64>>>(dd, da + 100, db + 100, m2, field, 32); hipMemcpy(o.data(), dd, n * 32, hipMemcpyDeviceToHost);
      int bad = 0; for (int i = 0; i < m2; i++) { FrM r = X.mul_w64(a[100 + i], b[100 + i]); if (memcmp(&r, &o[i], 32)) bad++; }
      printf("field %d partial wavefront of %d lanes: %d mismatches\n", field, m2, bad); total_bad += bad;
    }
    kinv<<<n / 64, 64>>>(dd, db, n, field); hipMemcpy(o.data(), dd, n * 32, hipMemcpyDeviceToHost);
    { int bad = 0; for (int i = 0; i < n; i++) { FrM r = X.inverse_bgcd(b[i]); if (memcmp(&r, &o[i], 32)) bad++; }
      printf("field %d constant-time inverse on the device against the host's shift-and-subtract form: %d mismatches of %d\n", field, bad, n); total_bad += bad; }
  }
  {
    std::vector<uint32_t> hw(8 * n); for (auto& x : hw) x = (uint32_t)g();
    for (int k = 0; k < 8; k++) { hw[k] = 0xffffffffu - (uint32_t)k * 0x9e3779b9u; hw[8 + k] = 0u - (uint32_t)k * 0x9e3779b9u; }     // all-ones and all-zero digests
    uint32_t* dh; hipMalloc(&dh, hw.size() * 4); hipMemcpy(dh, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    for (int form : {32, 64}) {
      if (form == 32) kdigest<32><<<n / 64, 64>>>(dd, dh, n); else kdigest<64><<<n / 64, 64>>>(dd, dh, n);
      hipMemcpy(o.data(), dd, n * 32, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < n; i++) {
        uint8_t out[32];
        for (int k = 0; k < 8; k++) { uint32_t h = hw[8 * i + k] + (uint32_t)k * 0x9e3779b9u; out[4 * k] = (uint8_t)(h >> 24); out[4 * k + 1] = (uint8_t)(h >> 16); out[4 * k + 2] = (uint8_t)(h >> 8); out[4 * k + 3] = (uint8_t)h; }
        FrM r = F.from_be32(out);
        if (memcmp(&r, &o[i], 32)) bad++;
      }
      printf("digest words -> bytes -> limbs -> Montgomery product, %d-bit form: %d mismatches of %d\n", form, bad, n); total_bad += bad;
    }
  }
  printf("fr_mul_probe: %s\n", total_bad ? "MISMATCHES" : "ok");
  return total_bad != 0;
}
