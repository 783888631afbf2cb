// Microbenchmark: issue rate of the integer / fp64 VALU instructions a BN254 Montgomery
// multiply can be built from, on gfx950.  Calibrates the compute roofline.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 32768;     // loop trips
constexpr int UNROLL = 16;      // instructions per trip per chain-set

enum Op { MAD64 = 0, MUL_LO, MUL_HI, MAD_U24, MUL_U24, MULHI_U24, ADD_U32, ADDC_PAIR, FMA_F64, LSHL_ADD_U64,
          MAD64_DEP, MIX_MAD_ADD2, MIX_MAD_ADD4, FMA_F32, MAD64_4CHAIN, EMMART_LIMB, ADD_F64, NOPS };
static const char* op_name[] = { "v_mad_u64_u32 (16 indep)", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_u32_u24",
  "v_mul_hi_u32_u24", "v_add_u32", "v_add_co+v_addc pair", "v_fma_f64", "v_lshl_add_u64", "v_mad_u64_u32 (1 dep chain)",
  "mix 1 mad64 + 2 add", "mix 1 mad64 + 4 add", "v_fma_f32", "v_mad_u64_u32 (4 chains)",
  "52x52 limb product on FMAs (2 fma + f64 sub + 2 u64 add)", "v_add_f64" };

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, unsigned long long* cyc) {
  uint32_t tid = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t a = seed * 2654435761u + tid, b = (seed ^ 0x9e3779b9u) + tid * 7u;
  uint64_t acc[16];
  uint32_t r[16];
  double d[16];
  float f[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { acc[i] = (uint64_t)(a + i) << 7 | i; r[i] = a ^ (i * 0x01010101u); d[i] = 1.0 + i * 1e-9; f[i] = 1.0f + i; }
  double da = 1.0000001, db = 1e-12;
  float fa = 1.0000001f, fb = 1e-7f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) {
      if constexpr (OP == MAD64)        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      else if constexpr (OP == MAD64_DEP) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
      else if constexpr (OP == MAD64_4CHAIN) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b) : "vcc");
      else if constexpr (OP == MUL_LO)  asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      else if constexpr (OP == MUL_HI)  asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      else if constexpr (OP == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      else if constexpr (OP == MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      else if constexpr (OP == MULHI_U24) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      else if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      else if constexpr (OP == ADDC_PAIR) { if (i & 1) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
                                             else asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(a) : "vcc"); }
      else if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
      else if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fa), "v"(fb));
      else if constexpr (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 15]));
      else if constexpr (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
      else if constexpr (OP == EMMART_LIMB) {
        // one 52 x 52 -> 104-bit limb product in double precision (Emmart, Zheng, Weems 2018): hi = fma(a, b, C1) rounds to a multiple of 2^52, sub = C2 - hi,
        // lo = fma(a, b, sub) is the exact low part; the two halves are then accumulated into their columns as 64-bit integers (the magic constants align the
        // mantissas).  Five instructions per limb product; a 5 x 5 limb Montgomery product needs 2 x 25 of them.
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d[i]) : "v"(da), "v"(d[(i + 1) & 15]), "v"(db));
        asm volatile("v_add_f64 %0, %1, -%2" : "=v"(d[(i + 2) & 15]) : "v"(db), "v"(d[i]));
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d[(i + 3) & 15]) : "v"(da), "v"(d[(i + 1) & 15]), "v"(d[(i + 2) & 15]));
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(d[i]));
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[(i + 8) & 15]) : "v"(d[(i + 3) & 15]));
      }
      else if constexpr (OP == MIX_MAD_ADD2) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[(i + 8) & 15]) : "v"(b));
      } else if constexpr (OP == MIX_MAD_ADD4) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[(i + 4) & 15]) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[(i + 8) & 15]) : "v"(a));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[(i + 12) & 15]) : "v"(b));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) x ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32) ^ r[i] ^ (uint32_t)__double_as_longlong(d[i]) ^ __float_as_uint(f[i]);
  out[tid] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> void run(int waves_per_simd, uint32_t* dout, unsigned long long* dcyc, int ncu) {
  // 256-thread block = 4 waves = 1 wave per SIMD on one CU; k blocks per CU => k waves/SIMD
  int grid = ncu * waves_per_simd;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<OP><<<grid, 256>>>(dout, 1, dcyc); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CK(hipEventRecord(e0)); k<OP><<<grid, 256>>>(dout, rep + 2, dcyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  std::vector<unsigned long long> cyc(grid); CK(hipMemcpy(cyc.data(), dcyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double med = 0; { std::vector<unsigned long long> c = cyc; std::sort(c.begin(), c.end()); med = (double)c[c.size() / 2]; }
  int per_it = (OP == MIX_MAD_ADD2) ? 3 : (OP == MIX_MAD_ADD4 || OP == EMMART_LIMB) ? 5 : 1;
  double winstr = (double)ITERS * UNROLL * per_it;              // wave-instructions per wave
  double total = winstr * grid * 4;                              // over all waves
  // s_memtime ticks at a fixed 100 MHz-derived rate on some parts; report both wall-derived and tick-derived
  printf("%-30s waves/SIMD=%d  time=%8.3f ms  wave-instr/s=%8.2f G  => %6.2f ns/wave-instr/SIMD  ticks/instr/wave=%6.2f\n",
         op_name[OP], waves_per_simd, best, total / (best * 1e-3) / 1e9, (best * 1e6) / (winstr * waves_per_simd), med / winstr);
}

#include <algorithm>
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount;
  printf("device %s  CUs=%d  clock=%d kHz\n", p.gcnArchName, ncu, p.clockRate);
  uint32_t* dout; unsigned long long* dcyc;
  CK(hipMalloc(&dout, (size_t)ncu * 8 * 256 * 4)); CK(hipMalloc(&dcyc, (size_t)ncu * 8 * 8));
  for (int w : {1, 2, 4, 8}) {
    run<MAD64>(w, dout, dcyc, ncu); run<MAD64_4CHAIN>(w, dout, dcyc, ncu); run<MAD64_DEP>(w, dout, dcyc, ncu);
    run<MUL_LO>(w, dout, dcyc, ncu); run<MUL_HI>(w, dout, dcyc, ncu);
    run<MAD_U24>(w, dout, dcyc, ncu); run<MUL_U24>(w, dout, dcyc, ncu); run<MULHI_U24>(w, dout, dcyc, ncu);
    run<ADD_U32>(w, dout, dcyc, ncu); run<ADDC_PAIR>(w, dout, dcyc, ncu); run<FMA_F32>(w, dout, dcyc, ncu);
    run<FMA_F64>(w, dout, dcyc, ncu); run<LSHL_ADD_U64>(w, dout, dcyc, ncu);
    run<MIX_MAD_ADD2>(w, dout, dcyc, ncu); run<MIX_MAD_ADD4>(w, dout, dcyc, ncu);
    run<ADD_F64>(w, dout, dcyc, ncu); run<EMMART_LIMB>(w, dout, dcyc, ncu);
    printf("\n");
  }
  return 0;
}
