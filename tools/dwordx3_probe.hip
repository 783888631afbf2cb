// tools/dwordx3_probe.hip -- minimal reproduction attempt of the round-1 "lanes 12..15 of every 16, proof index >= 65 548" corruption seen with
// 12-byte-per-lane buffer_load/store_dwordx3 workspace rows.  Writes a known pattern through raw buffer dwordx3 stores
// (SGPR row offset + VGPR lane offset i * 12, as the round-1 kernels did), reads it back (a) through dwordx3 buffer loads and (b) through plain
// global loads, for several descriptor settings, and reports which lanes differ.
//   hipcc -O2 --offload-arch=gfx950 dwordx3_probe.hip -o dwordx3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int32_t i32x3 __attribute__((ext_vector_type(3)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(void* base, uint32_t bytes) {
  uint64_t b = (uint64_t)base;
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__global__ void k_store(int32_t* ws, uint32_t n, int rows, uint32_t num_records) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  __amdgpu_buffer_rsrc_t r = mk(ws, num_records);
  if (i >= n) return;   // no out-of-range lanes here: with num_records = 0xffffffff they would not be dropped
  const uint32_t voff = i * 12u;
  for (int row = 0; row < rows; row++) {
    i32x3 v = {(int32_t)(i * 3u + 0u + 1000003u * row), (int32_t)(i * 3u + 1u + 1000003u * row), (int32_t)(i * 3u + 2u + 1000003u * row)};
    __builtin_amdgcn_raw_buffer_store_b96(v, r, voff, (uint32_t)row * n * 12u, 0);
  }
}
__global__ void k_load_check(int32_t* ws, uint32_t n, int rows, uint32_t num_records, uint32_t* bad_buf, uint32_t* bad_glob) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  __amdgpu_buffer_rsrc_t r = mk(ws, num_records);
  if (i >= n) return;
  const uint32_t voff = i * 12u;
  for (int row = 0; row < rows; row++) {
    i32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r, voff, (uint32_t)row * n * 12u, 0);
    if (i < n) {
      const int32_t e0 = (int32_t)(i * 3u + 1000003u * row);
      if (v.x != e0 || v.y != e0 + 1 || v.z != e0 + 2) atomicOr(&bad_buf[i], 1u << row);
      const int32_t* g = ws + ((size_t)row * n + i) * 3;
      if (g[0] != e0 || g[1] != e0 + 1 || g[2] != e0 + 2) atomicOr(&bad_glob[i], 1u << row);
    }
  }
}
int main() {
  const uint32_t n = 196608; const int rows = 8;   // n: a multiple of 256, above 65 536
  int32_t* ws; uint32_t *b1, *b2;
  CK(hipMalloc((void**)&ws, (size_t)n * rows * 12 + 4096)); CK(hipMalloc((void**)&b1, n * 4)); CK(hipMalloc((void**)&b2, n * 4));
  for (uint32_t nr : {n * rows * 12u, 0xffffffffu}) {
    CK(hipMemset(ws, 0xee, (size_t)n * rows * 12)); CK(hipMemset(b1, 0, n * 4)); CK(hipMemset(b2, 0, n * 4));
    hipLaunchKernelGGL(k_store, dim3((n + 255) / 256), dim3(256), 0, 0, ws, n, rows, nr);
    hipLaunchKernelGGL(k_load_check, dim3((n + 255) / 256), dim3(256), 0, 0, ws, n, rows, nr, b1, b2);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h1(n), h2(n);
    CK(hipMemcpy(h1.data(), b1, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), b2, n * 4, hipMemcpyDeviceToHost));
    size_t c1 = 0, c2 = 0; uint32_t first1 = 0, first2 = 0; unsigned lanehist1[16] = {0}, lanehist2[16] = {0};
    for (uint32_t i = 0; i < n; i++) { if (h1[i]) { if (!c1) first1 = i; c1++; lanehist1[i & 15]++; } if (h2[i]) { if (!c2) first2 = i; c2++; lanehist2[i & 15]++; } }
    printf("num_records 0x%08x: dwordx3 read-back mismatches %zu (first %u), global read-back mismatches %zu (first %u)\n", nr, c1, first1, c2, first2);
    if (c1 || c2) { printf("  by lane mod 16 (buffer / global):"); for (int l = 0; l < 16; l++) printf(" %u/%u", lanehist1[l], lanehist2[l]); printf("\n"); }
  }
  fflush(stdout);
  return 0;
}
