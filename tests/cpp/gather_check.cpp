// tests/cpp/gather_check.cpp -- TEST: bn254_status_all_gather (the status gather of a multi-process job) as a C++ host would call it.
//   built with -DGATHER_MOCK: the executable exports its own ncclAllGather that plays `world` ranks inside one process (the library looks the symbol up
//   in the process first), so the shard arithmetic -- equal and ragged shards, padded in-place blocks, packing -- is checked on one GPU;
//   built without: a real one-rank RCCL communicator (ncclCommInitRank with nranks = 1), the library finds RCCL's ncclAllGather.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bn254_verify.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

#ifdef GATHER_MOCK
struct MockComm { int world, rank; size_t n; const uint8_t* whole; /* host: the statuses of the unsharded batch */ };
extern "C" __attribute__((visibility("default"))) int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t s) {
  const MockComm* c = (const MockComm*)comm;
  if (dtype != 1) return 99;                                        // ncclUint8
  int devs[64], k = 0; size_t first[64], cnt[64];
  if (bn254_shard_plan(c->n, c->world == 64 ? ~0ull : ((1ull << c->world) - 1), c->world, devs, first, cnt, &k)) return 98;
  for (int r = 0; r < c->world; r++) {
    uint8_t* dst = (uint8_t*)recv + (size_t)r * count;
    if (r == c->rank) { if (dst != send && hipMemcpyAsync(dst, send, count, hipMemcpyDeviceToDevice, s) != hipSuccess) return 97; continue; }
    // what rank r would send: its shard, padded to `count` with 0xAA (the padding must never reach the result)
    std::vector<uint8_t> blk(count, 0xAA);
    memcpy(blk.data(), c->whole + first[r], cnt[r]);
    if (hipMemcpyAsync(dst, blk.data(), count, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 96;
  }
  return 0;
}
#else
#include <rccl/rccl.h>
#endif

int main() {
  int fails = 0;
  hipStream_t s; CK(hipStreamCreate(&s));
#ifdef GATHER_MOCK
  const struct { int world; size_t n; } cases[] = {{2, 1000}, {2, 7}, {3, 1001}, {8, 1 << 20}, {8, (1 << 20) + 777}, {5, 3}, {4, 4}};
  for (auto& cs : cases) {
    std::vector<uint8_t> whole(cs.n);
    for (size_t i = 0; i < cs.n; i++) whole[i] = (uint8_t)((7 * i + 3) % 5);
    int devs[64], k = 0; size_t first[64], cnt[64];
    if (bn254_shard_plan(cs.n, (1ull << cs.world) - 1, cs.world, devs, first, cnt, &k)) { printf("plan failed\n"); return 1; }
    const size_t cap = (cs.n + cs.world - 1) / cs.world;
    for (int rank = 0; rank < cs.world; rank += (cs.world > 3 ? cs.world - 1 : 1)) {     // first and last rank of the large worlds, every rank of the small
      uint8_t *d_local, *d_full, *d_scr;
      CK(hipMalloc((void**)&d_local, cnt[rank] ? cnt[rank] : 1)); CK(hipMalloc((void**)&d_full, cs.n)); CK(hipMalloc((void**)&d_scr, cap * cs.world));
      CK(hipMemcpy(d_local, whole.data() + first[rank], cnt[rank], hipMemcpyHostToDevice));
      CK(hipMemset(d_full, 0xEE, cs.n));
      MockComm c{cs.world, rank, cs.n, whole.data()};
      int rc = bn254_status_all_gather(&c, cs.world, rank, d_local, cs.n, d_full, d_scr, s);
      CK(hipStreamSynchronize(s));
      std::vector<uint8_t> got(cs.n);
      CK(hipMemcpy(got.data(), d_full, cs.n, hipMemcpyDeviceToHost));
      const bool ok = rc == BN254_OK && got == whole;
      printf("mock world %d n %zu rank %d: %s\n", cs.world, cs.n, rank, ok ? "ok" : "MISMATCH");
      if (!ok) { fails++; printf("  rc %d %s\n", rc, bn254_last_error()); }
      if (cs.n % cs.world) {   // ragged shards without scratch: refused, nothing launched
        if (bn254_status_all_gather(&c, cs.world, rank, d_local, cs.n, d_full, nullptr, s) != BN254_E_BAD_ARG) { fails++; printf("  missing scratch not refused\n"); }
      }
      CK(hipFree(d_local)); CK(hipFree(d_full)); CK(hipFree(d_scr));
    }
  }
#else
  ncclUniqueId id; ncclComm_t comm;
  if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) { printf("RCCL communicator failed\n"); return 1; }
  const size_t n = 4133;
  std::vector<uint8_t> whole(n), got(n);
  for (size_t i = 0; i < n; i++) whole[i] = (uint8_t)(i % 7);
  uint8_t *d_local, *d_full;
  CK(hipMalloc((void**)&d_local, n)); CK(hipMalloc((void**)&d_full, n));
  CK(hipMemcpy(d_local, whole.data(), n, hipMemcpyHostToDevice)); CK(hipMemset(d_full, 0xEE, n));
  int rc = bn254_status_all_gather(comm, 1, 0, d_local, n, d_full, nullptr, s);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(got.data(), d_full, n, hipMemcpyDeviceToHost));
  printf("rccl one-rank gather: %s\n", (rc == BN254_OK && got == whole) ? "ok" : "MISMATCH");
  if (rc != BN254_OK || got != whole) { fails++; printf("  rc %d %s\n", rc, bn254_last_error()); }
  ncclCommDestroy(comm);
#endif
  if (bn254_status_all_gather(nullptr, 2, 0, nullptr, 10, nullptr, nullptr, s) != BN254_E_BAD_ARG) { fails++; printf("null communicator not refused\n"); }
  printf(fails ? "gather_check: %d failures\n" : "gather_check ok\n", fails);
  return fails ? 1 : 0;
}
