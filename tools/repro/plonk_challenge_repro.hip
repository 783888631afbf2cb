// tools/repro/plonk_challenge_repro.hip -- towards a reduced test case for the wrong Fiat-Shamir challenge of round 4 (DESIGN.md "a wrong challenge on the device"):
// with FrCtx::mul_w32 inlined by force (-DBN254_FR_MUL_INLINE=1) and without the register barrier of FrCtx::from_be32 (-DBN254_FR_NO_BARRIER), k_plonk_stage1 derived
// gamma = from_be32(SHA-256 digest) wrongly from a CORRECT digest.  This file runs the same header code in kernels of decreasing size and compares every lane with the
// host's compile of the same source:
//   K_CHAIN  PlonkStage1::chain on one lane per proof (the four chained challenges, zeta^n, the denominators) -- what wavefront 0 of k_plonk_stage1 runs
//   K_GAMMA  the first challenge alone: SHA-256 over the key prefix, the public inputs and 192 proof bytes, then from_be32
//   K_DIGEST from_be32 of a digest that is handed in (no hashing in the kernel)
// The lanes' pending SHA-256 block lives in LDS exactly as in the product (bn254_plonk.hpp::pl_lane_lds: dynamic LDS, stride at offset 0).
// Build both ways and run on an MI355X (tools/gpu_repro.sh):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I snark-bn254-verifier_amd/csrc [-DBN254_FR_MUL_INLINE=1 -DBN254_FR_NO_BARRIER] tools/repro/plonk_challenge_repro.hip -o repro
//   ./repro tests/golden/plonk_vk.bin      -> one line per kernel: lanes that differ from the host, first differing value
#define BN254_PLONK_DEVICE_TU 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "bn254_plonk.hpp"
using namespace bn254host;

#define LANE_STRIDE 68u   // bytes per lane slot: a 64-byte SHA block + one dword (odd dword count, as the product's strides are)
__device__ __forceinline__ void lds_setup() {
  extern __shared__ uint8_t dyn[];
  if (threadIdx.x == 0) *(uint32_t*)dyn = LANE_STRIDE;
  __syncthreads();
}
__global__ void __launch_bounds__(64) k_chain(const PlonkKey* key, const uint8_t* proof, size_t len, const uint8_t* inputs, size_t n_pub, FrM* out /* 4 per lane */, int* st) {
  lds_setup();
  PlonkLayout lay;
  if (!plonk_proof_layout(lay, proof, len)) { st[threadIdx.x] = -1; return; }
  PlonkStage1 s; PlonkWork wk;
  st[threadIdx.x] = s.chain(*key, proof, inputs, n_pub, wk, lay.off_bsb, lay.n_bsb);
  out[4 * threadIdx.x] = s.gamma; out[4 * threadIdx.x + 1] = s.beta; out[4 * threadIdx.x + 2] = s.alpha; out[4 * threadIdx.x + 3] = s.zeta;
}
__global__ void __launch_bounds__(64) k_gamma(const PlonkKey* key, const uint8_t* proof, const uint8_t* inputs, size_t n_pub, FrM* out, uint8_t* digest) {
  lds_setup();
  uint8_t dg[32];
  Challenge cg(key->gamma_mid);
  cg.bind(inputs, 32 * n_pub);
  cg.bind(proof, 192);
  out[threadIdx.x] = cg.finish(dg);
  for (int i = 0; i < 32; i++) digest[32 * threadIdx.x + i] = dg[i];
}
__global__ void __launch_bounds__(64) k_digest(const uint8_t* digest, FrM* out) { out[threadIdx.x] = fr_ctx().from_be_reduce(digest + 32 * threadIdx.x, 32); }

// K_PI: the public-input sum of PlonkStage1::b (verify.rs:109-137) -- per input three products and FrCtx::from_be32 of 32 bytes that sit in LDS (as in k_plonk_stage1,
// which stages every proof's inputs there) or in global memory: the site at which the round-5 rebuild of the bad flags goes wrong inside the full kernel
template <bool FROM_LDS>
__global__ void __launch_bounds__(64) k_pi(const uint8_t* inputs, int n_pub, FrM zs, FrM iv, FrM gen, FrM* out) {
  extern __shared__ uint8_t dyn[];
  uint8_t* mine = dyn + 16 + 64 * LANE_STRIDE + threadIdx.x * (32 * 8 + 4);
  if (FROM_LDS) for (int k = 0; k < 32 * n_pub; k++) mine[k] = inputs[k];
  __syncthreads();
  const uint8_t* in = FROM_LDS ? mine : inputs;
  const FrCtx& F = fr_ctx();
  FrM pi = {{0, 0, 0, 0}}, accw = F.one;
  for (int i = 0; i < n_pub; i++) {
    FrM x = F.mul(F.mul(F.mul(zs, iv), accw), F.from_be32(in + 32 * i));
    accw = F.mul(accw, gen);
    pi = F.add(pi, x);
    iv = F.mul(iv, iv);
  }
  out[threadIdx.x] = pi;
}
static FrM host_pi(const uint8_t* in, int n_pub, FrM zs, FrM iv, FrM gen) {
  const FrCtx& F = fr_ctx();
  FrM pi = {{0, 0, 0, 0}}, accw = F.one;
  for (int i = 0; i < n_pub; i++) { FrM x = F.mul(F.mul(F.mul(zs, iv), accw), F.from_be32(in + 32 * i)); accw = F.mul(accw, gen); pi = F.add(pi, x); iv = F.mul(iv, iv); }
  return pi;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
static void hexs(char* o, const FrM& v) { uint8_t b[32]; fr_ctx().to_be(b, v); for (int i = 0; i < 32; i++) sprintf(o + 2 * i, "%02x", b[i]); }
int main(int argc, char** argv) {
  const FrCtx& F = fr_ctx(); const Fp64Ctx& C = fp64_ctx();
  std::vector<uint8_t> vkb;
  { FILE* f = fopen(argc > 1 ? argv[1] : "tests/golden/plonk_vk.bin", "rb"); if (!f) { printf("key file?\n"); return 2; } uint8_t buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) vkb.insert(vkb.end(), buf, buf + k); fclose(f); }
  PlonkKey key;
  if (parse_plonk_vk(key, vkb.data(), vkb.size()) != DEC_OK) { printf("key does not parse\n"); return 2; }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_plonk_fr_raw), &F, sizeof(FrCtx))); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_plonk_fp64_raw), &C, sizeof(Fp64Ctx)));
  const uint32_t q = key.n_qcp; const size_t n_pub = (size_t)key.nb_public, len = 516 + 32 * (size_t)(6 + q) + 100 + 64 * (size_t)q;
  std::vector<uint8_t> proof(len), inputs(32 * n_pub + 4);
  uint64_t sm = 0x5e1f7e57b254ull;
  auto next = [&sm] { uint64_t z = (sm += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
  for (auto& b : proof) b = (uint8_t)next();
  for (auto& b : inputs) b = (uint8_t)next();
  proof[512] = proof[513] = proof[514] = 0; proof[515] = (uint8_t)(6 + q);
  const size_t off_zs = 516 + 32 * (size_t)(6 + q);
  proof[off_zs + 96] = proof[off_zs + 97] = proof[off_zs + 98] = 0; proof[off_zs + 99] = (uint8_t)q;
  // host: the same source
  PlonkLayout lay; plonk_proof_layout(lay, proof.data(), len);
  PlonkStage1 hs; PlonkWork hw;
  const int hst = hs.chain(key, proof.data(), inputs.data(), n_pub, hw, lay.off_bsb, lay.n_bsb);
  uint8_t hdg[32];
  { Challenge cg(key.gamma_mid); cg.bind(inputs.data(), 32 * n_pub); cg.bind(proof.data(), 192); (void)cg.finish(hdg); }
  // device
  PlonkKey* dkey; uint8_t *dproof, *dinputs, *ddig; FrM* dout; int* dst;
  CK(hipMalloc(&dkey, sizeof key)); CK(hipMemcpy(dkey, &key, sizeof key, hipMemcpyHostToDevice));
  CK(hipMalloc(&dproof, len)); CK(hipMemcpy(dproof, proof.data(), len, hipMemcpyHostToDevice));
  CK(hipMalloc(&dinputs, inputs.size())); CK(hipMemcpy(dinputs, inputs.data(), inputs.size(), hipMemcpyHostToDevice));
  CK(hipMalloc(&ddig, 64 * 32)); CK(hipMalloc(&dout, 64 * 4 * sizeof(FrM))); CK(hipMalloc(&dst, 64 * sizeof(int)));
  const size_t lds = 16 + 64 * LANE_STRIDE;
  std::vector<FrM> o(256); std::vector<int> st(64); std::vector<uint8_t> dg(64 * 32);
  int bad_total = 0;
  auto report = [&](const char* name, int bad, const FrM& dev, const FrM& host) {
    char a[65], b[65]; hexs(a, dev); hexs(b, host);
    printf("%-9s %2d of 64 lanes differ from the host%s%s%s%s\n", name, bad, bad ? "; device " : "", bad ? a : "", bad ? " host " : "", bad ? b : "");
    bad_total += bad;
  };
  {
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), lds, 0, dkey, dproof, len, dinputs, n_pub, dout, dst);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(o.data(), dout, 256 * sizeof(FrM), hipMemcpyDeviceToHost)); CK(hipMemcpy(st.data(), dst, 64 * sizeof(int), hipMemcpyDeviceToHost));
    const FrM want[4] = {hs.gamma, hs.beta, hs.alpha, hs.zeta}; const char* nm[4] = {"K_CHAIN g", "K_CHAIN b", "K_CHAIN a", "K_CHAIN z"};
    for (int k = 0; k < 4; k++) { int bad = 0, first = -1; for (int i = 0; i < 64; i++) if (!F.eq(o[4 * i + k], want[k])) { bad++; if (first < 0) first = i; } report(nm[k], bad, o[4 * (first < 0 ? 0 : first) + k], want[k]); }
    int sb = 0; for (int i = 0; i < 64; i++) sb += st[i] != hst; printf("K_CHAIN   status differs in %d lanes (host %d)\n", sb, hst); bad_total += sb;
  }
  {
    hipLaunchKernelGGL(k_gamma, dim3(1), dim3(64), lds, 0, dkey, dproof, dinputs, n_pub, dout, ddig);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(o.data(), dout, 64 * sizeof(FrM), hipMemcpyDeviceToHost)); CK(hipMemcpy(dg.data(), ddig, 64 * 32, hipMemcpyDeviceToHost));
    int bad = 0, first = -1, dbad = 0; for (int i = 0; i < 64; i++) { if (!F.eq(o[i], hs.gamma)) { bad++; if (first < 0) first = i; } dbad += memcmp(&dg[32 * i], hdg, 32) != 0; }
    report("K_GAMMA", bad, o[first < 0 ? 0 : first], hs.gamma);
    printf("K_GAMMA   digest differs in %d lanes\n", dbad); bad_total += dbad;
    printf("K_GAMMA   digest = "); for (int i = 0; i < 32; i++) printf("%02x", hdg[i]); printf("\n");
  }
  {
    for (int i = 0; i < 64; i++) memcpy(&dg[32 * i], hdg, 32);
    CK(hipMemcpy(ddig, dg.data(), 64 * 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_digest, dim3(1), dim3(64), 0, 0, ddig, dout);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(o.data(), dout, 64 * sizeof(FrM), hipMemcpyDeviceToHost));
    int bad = 0, first = -1; for (int i = 0; i < 64; i++) if (!F.eq(o[i], hs.gamma)) { bad++; if (first < 0) first = i; }
    report("K_DIGEST", bad, o[first < 0 ? 0 : first], hs.gamma);
  }
  {
    const int np = 8;
    std::vector<uint8_t> pin(32 * np); for (auto& b : pin) b = (uint8_t)next();
    uint8_t* dpin; CK(hipMalloc(&dpin, pin.size())); CK(hipMemcpy(dpin, pin.data(), pin.size(), hipMemcpyHostToDevice));
    const FrM zs = hs.zeta_n, iv = hs.zh_zeta, gen = key.generator;
    const FrM want = host_pi(pin.data(), np, zs, iv, gen);
    const size_t lds2 = lds + 64 * (32 * 8 + 4);
    for (int from_lds = 1; from_lds >= 0; from_lds--) {
      if (from_lds) hipLaunchKernelGGL(k_pi<true>, dim3(1), dim3(64), lds2, 0, dpin, np, zs, iv, gen, dout);
      else hipLaunchKernelGGL(k_pi<false>, dim3(1), dim3(64), lds2, 0, dpin, np, zs, iv, gen, dout);
      CK(hipDeviceSynchronize()); CK(hipMemcpy(o.data(), dout, 64 * sizeof(FrM), hipMemcpyDeviceToHost));
      int bad = 0, first = -1; for (int i = 0; i < 64; i++) if (!F.eq(o[i], want)) { bad++; if (first < 0) first = i; }
      report(from_lds ? "K_PI lds" : "K_PI glob", bad, o[first < 0 ? 0 : first], want);
    }
  }
  printf("%s\n", bad_total ? "MISMATCH" : "all kernels agree with the host");
  return bad_total ? 1 : 0;
}
