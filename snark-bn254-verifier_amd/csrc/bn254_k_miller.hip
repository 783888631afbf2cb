// bn254_k_miller.hip -- the Miller-loop step kernels of the large-batch (one proof per lane) Groth16 path: their own translation unit, because
// each instantiation is ~300 KB of straight-line code and the three of them dominate the library's compile time.
//   k_miller_step_dbl<DO_SQR>, k_miller_step_add: one whole step of the shared Miller loop (bn254_vm.h::vm_miller_step): [f <- f^2,] T <- 2T or
//   T + Q, f <- f * line_T(A) * line_gamma(L) * line_delta(C)   (replaces one iteration of bn's miller_loop_batch under groth16/verify.rs:73-77)
#include <cstdlib>
#include "bn254_devws.h"

namespace bn254 {

// the whole Miller step of the three pairs (bn254_vm.h::vm_miller_step): doubling steps (with the squaring of f, except the first)
// and addition steps as two kernels so that each carries only its own G2 formulas
template <bool DO_SQR>
__global__ void __launch_bounds__(256, 2) k_miller_step_dbl(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int e_t, int e, int e_pa,
                                                            const int32_t* __restrict__ entry0, int e_p0, int inf_mask0,
                                                            const int32_t* __restrict__ entry1, int e_p1, int inf_mask1) {
  __shared__ int32_t park_lds[72 * 256];
  VM_KERNEL_PROLOGUE();
  w.lds = park_lds;
  FixedLine l0, l1;
  l0.m = uni_ld2(entry0); l0.c = uni_ld2(entry0 + 2 * BN_NL); l0.xc = uni_ld2(entry0 + 4 * BN_NL);
  l1.m = uni_ld2(entry1); l1.c = uni_ld2(entry1 + 2 * BN_NL); l1.xc = uni_ld2(entry1 + 4 * BN_NL);
  vm_miller_step<DO_SQR>(w, 0, e_t, 0, e, e_pa, l0, e_p0, (st & inf_mask0) != 0, l1, e_p1, (st & inf_mask1) != 0);
}
__global__ void __launch_bounds__(256, 2) k_miller_step_add(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, int kind, int e_t, int e_b, int e, int e_pa,
                                                            const int32_t* __restrict__ entry0, int e_p0, int inf_mask0,
                                                            const int32_t* __restrict__ entry1, int e_p1, int inf_mask1) {
  __shared__ int32_t park_lds[72 * 256];
  VM_KERNEL_PROLOGUE();
  w.lds = park_lds;
  FixedLine l0, l1;
  l0.m = uni_ld2(entry0); l0.c = uni_ld2(entry0 + 2 * BN_NL); l0.xc = uni_ld2(entry0 + 4 * BN_NL);
  l1.m = uni_ld2(entry1); l1.c = uni_ld2(entry1 + 2 * BN_NL); l1.xc = uni_ld2(entry1 + 4 * BN_NL);
  int k = __builtin_amdgcn_readfirstlane(kind);
  vm_miller_step<false>(w, k < 1 ? 1 : k, e_t, e_b, e, e_pa, l0, e_p0, (st & inf_mask0) != 0, l1, e_p1, (st & inf_mask1) != 0);
}

// A RUN of steps in one launch (bn254_vm.h::vm_miller_run): steps [s_begin, s_end) of the loop -- by default the WHOLE loop.  f is loaded and stored
// once per run and travels in LDS + registers in between; the kind of a step and its line-table entries come from scalar loads indexed by the step.
struct DevLines {
  const int32_t* tab0; const int32_t* tab1;
  __device__ __forceinline__ FixedLine get(int t, int s) const {
    const int32_t* e = (t == 0 ? tab0 : tab1) + (size_t)s * FIXED_LINE_DWORDS;
    FixedLine l; l.m = uni_ld2(e); l.c = uni_ld2(e + 2 * BN_NL); l.xc = uni_ld2(e + 4 * BN_NL);
    return l;
  }
};
struct DevKinds {   // the step table (88 entries, 0..4), two steps per byte, by value in the kernel arguments
  MillerKinds k;
  __device__ __forceinline__ int get(int s) const { return __builtin_amdgcn_readfirstlane((k.nib[s >> 1] >> ((s & 1) * 4)) & 15); }
};
__global__ void __launch_bounds__(256, 2) k_miller_run(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, MillerKinds kinds, int s_begin, int s_end,
                                                       int e_t, int e_b, int e, int e_pa, const int32_t* __restrict__ tab0, int e_p0, int inf_mask0,
                                                       const int32_t* __restrict__ tab1, int e_p1, int inf_mask1) {
  __shared__ int32_t park_lds[72 * 256];
  VM_KERNEL_PROLOGUE();
  w.lds = park_lds;
  DevLines lines{uni_ptr(tab0), uni_ptr(tab1)};
  DevKinds dk{kinds};
  vm_miller_run(w, lines, dk, __builtin_amdgcn_readfirstlane(s_begin), __builtin_amdgcn_readfirstlane(s_end), e_t, e_b, e, e_pa, e_p0, (st & inf_mask0) != 0,
                e_p1, (st & inf_mask1) != 0);
}

// the loop of two table-driven pairs without a variable pair (bn254_vm.h::vm_miller_run_fixed2): the pairing check of a large PlonK batch
__global__ void __launch_bounds__(256, 2) k_miller_run_fixed2(int32_t* ws, uint32_t n, const uint8_t* __restrict__ status, MillerKinds kinds, int s_begin, int s_end, int e,
                                                              const int32_t* __restrict__ tab0, int e_p0, int inf_mask0, const int32_t* __restrict__ tab1, int e_p1, int inf_mask1) {
  __shared__ int32_t park_lds[72 * 256];
  VM_KERNEL_PROLOGUE();
  w.lds = park_lds;
  DevLines lines{uni_ptr(tab0), uni_ptr(tab1)};
  DevKinds dk{kinds};
  vm_miller_run_fixed2(w, lines, dk, __builtin_amdgcn_readfirstlane(s_begin), __builtin_amdgcn_readfirstlane(s_end), e, e_p0, (st & inf_mask0) != 0, e_p1, (st & inf_mask1) != 0);
}

}  // namespace bn254

using namespace bn254;
void bn254_launch_miller_run_fixed2(const MillerKinds& kinds, int s_begin, int s_end, int32_t* ws, uint32_t n, const uint8_t* status, unsigned grid, hipStream_t s, int e,
                                    const int32_t* tab0, int ep0, int inf0, const int32_t* tab1, int ep1, int inf1) {
  hipLaunchKernelGGL(k_miller_run_fixed2, dim3(grid), dim3(256), 0, s, ws, n, status, kinds, s_begin, s_end, e, tab0, ep0, inf0, tab1, ep1, inf1);
}
void bn254_launch_miller_run(const MillerKinds& kinds, int s_begin, int s_end, int32_t* ws, uint32_t n, const uint8_t* status, unsigned grid, hipStream_t s, int et, int eb,
                             int e, int epa, const int32_t* tab0, int ep0, int inf0, const int32_t* tab1, int ep1, int inf1) {
  hipLaunchKernelGGL(k_miller_run, dim3(grid), dim3(256), 0, s, ws, n, status, kinds, s_begin, s_end, et, eb, e, epa, tab0, ep0, inf0, tab1, ep1, inf1);
}
void bn254_launch_miller_step(bool do_sqr, int kind, int32_t* ws, uint32_t n, const uint8_t* status, unsigned grid, hipStream_t s, int et, int eb, int e, int epa,
                              const int32_t* t0, int ep0, int inf0, const int32_t* t1, int ep1, int inf1) {
  if (kind == 0 && do_sqr) hipLaunchKernelGGL(k_miller_step_dbl<true>, dim3(grid), dim3(256), 0, s, ws, n, status, et, e, epa, t0, ep0, inf0, t1, ep1, inf1);
  else if (kind == 0) hipLaunchKernelGGL(k_miller_step_dbl<false>, dim3(grid), dim3(256), 0, s, ws, n, status, et, e, epa, t0, ep0, inf0, t1, ep1, inf1);
  else hipLaunchKernelGGL(k_miller_step_add, dim3(grid), dim3(256), 0, s, ws, n, status, kind, et, eb, e, epa, t0, ep0, inf0, t1, ep1, inf1);
}
