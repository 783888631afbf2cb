// bn254_devws.h -- device-side workspace accessor shared by the kernel translation units (bn254_kernels.hip, bn254_k_miller.hip, bn254_k_rlc.hip).
// Workspace (bn254_vm.h element map): element e, digit l, proof i at dword (e * 9 + l) * n + i, accessed through ONE buffer descriptor: the row
// offset (e, l) is wave-uniform and travels in an SGPR (soffset), the lane offset i * 4 is one VGPR shared by every access, so no per-access address
// arithmetic exists and a wave-level access is one contiguous 256-byte segment (buffer_load_dword / buffer_store_dword).  Lanes past the end of the
// batch get an out-of-range offset: the descriptor's bounds check returns 0 for their loads and drops their stores.
#pragma once
#include <hip/hip_runtime.h>
#include "bn254_vm.h"
#include "bn254_kernels.h"

namespace bn254 {

// ---- workspace accessor ----------------------------------------------------------------------------------------------------------
struct DevWs {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t row_bytes;  // n * 4: one row per (element, limb)
  uint32_t voff;       // lane * 4
  int32_t* lds = nullptr;  // parking space of the operations that fuse two Fp12 products (72 dwords per lane, lane-interleaved)
  __device__ __forceinline__ void park(int slot, const Fp2& a) const {
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { lds[(slot * 18 + l) * 256 + threadIdx.x] = a.c0.v[l]; lds[(slot * 18 + BN_NL + l) * 256 + threadIdx.x] = a.c1.v[l]; }
  }
  __device__ __forceinline__ Fp2 unpark(int slot) const {
    Fp2 r;
#pragma unroll
    for (int l = 0; l < BN_NL; l++) { r.c0.v[l] = lds[(slot * 18 + l) * 256 + threadIdx.x]; r.c1.v[l] = lds[(slot * 18 + BN_NL + l) * 256 + threadIdx.x]; }
    BN_SETB(r.c0, 3.0, 0.5); BN_SETB(r.c1, 3.0, 0.5);
    return r;
  }
  __device__ __forceinline__ DevWs(int32_t* base, uint32_t n, uint32_t lane) {
    uint64_t b = (uint64_t)base;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    uint32_t nn = __builtin_amdgcn_readfirstlane(n);
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, nn * (uint32_t)(G16_WS_ELEMS * 36), 0x00020000);
    row_bytes = nn * 4u;
    voff = lane * 4u;
  }
  __device__ __forceinline__ Fp ld(int e) const {
    Fp r;
    uint32_t eu = __builtin_amdgcn_readfirstlane((uint32_t)e);
#pragma unroll
    for (int l = 0; l < BN_NL; l++) r.v[l] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (eu * (uint32_t)BN_NL + (uint32_t)l) * row_bytes, 0);
    return r;
  }
  __device__ __forceinline__ void st(int e, const Fp& a) const {
    uint32_t eu = __builtin_amdgcn_readfirstlane((uint32_t)e);
#pragma unroll
    for (int l = 0; l < BN_NL; l++) __builtin_amdgcn_raw_buffer_store_b32(a.v[l], rsrc, voff, (eu * (uint32_t)BN_NL + (uint32_t)l) * row_bytes, 0);
  }
};

// Lanes past the end of the batch get this lane index: lane * 4 lies beyond num_records, so the buffer bounds check makes their
// loads return 0 and drops their stores.  (They must NOT alias a live proof: different waves run the in-place operations at
// different times.)
#define DEAD_LANE (0xffffffffu / 4u)

// uniform (batch-constant) data: limbs stored contiguously per element; the pointer is wave-uniform -> scalar loads
__device__ __forceinline__ const int32_t* uni_ptr(const int32_t* p) {
  uint64_t b = (uint64_t)p;
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return (const int32_t*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ Fp uni_ld(const int32_t* p) {
  Fp r;
#pragma unroll
  for (int l = 0; l < BN_NL; l++) r.v[l] = p[l];
  return r;
}
__device__ __forceinline__ Fp2 uni_ld2(const int32_t* p) { Fp2 r; r.c0 = uni_ld(p); r.c1 = uni_ld(p + BN_NL); return r; }

// table entry -> affine point (80-byte entry, 16-byte aligned: five 16-byte loads)
__device__ __forceinline__ G1Aff msm_entry(const int32_t* __restrict__ msm_tab, size_t idx) {
  const int4* e = (const int4*)(msm_tab + idx * MSM_ENTRY_DWORDS);
  int4 v0 = e[0], v1 = e[1], v2 = e[2], v3 = e[3], v4 = e[4];
  G1Aff q;
  q.x.v[0] = v0.x; q.x.v[1] = v0.y; q.x.v[2] = v0.z; q.x.v[3] = v0.w; q.x.v[4] = v1.x; q.x.v[5] = v1.y; q.x.v[6] = v1.z; q.x.v[7] = v1.w;
  q.x.v[8] = v2.x; q.y.v[0] = v2.y; q.y.v[1] = v2.z; q.y.v[2] = v2.w; q.y.v[3] = v3.x; q.y.v[4] = v3.y; q.y.v[5] = v3.z; q.y.v[6] = v3.w;
  q.y.v[7] = v4.x; q.y.v[8] = v4.y;
  BN_SETB(q.x, 1.0, 0.5); BN_SETB(q.y, 1.0, 0.5);
  return q;
}

// ---- every VM operation (bn254_vm.h) is its own kernel ------------------------------------------------------------------------
// The VM programs (vm_miller_program, vm_final_exp_program) are host-compilable: the host walks them and enqueues one launch
// per operation (~210 per batch, all asynchronous on the sub-batch's stream, so launch overhead hides behind the previous kernel
// for any batch that matters).  No device-side function calls: each kernel gets exactly the registers it needs and no stack.
// A wave whose 64 proofs have all failed earlier checks exits at once.
#define VM_KERNEL_PROLOGUE()                                                                     \
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;                                           \
  const uint8_t st = status[i < n ? i : n - 1];                                                 \
  if (__builtin_amdgcn_ballot_w64((st & BN254_ST_PENDING) != 0) == 0) return;                   \
  DevWs w(ws, n, i < n ? i : DEAD_LANE)


}  // namespace bn254
