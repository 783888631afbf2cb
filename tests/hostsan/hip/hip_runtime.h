// tests/hostsan/hip/hip_runtime.h -- TEST HARNESS: a stand-in for the HIP runtime API backed by host memory, so that the HOST half of the product
// (snark-bn254-verifier_amd/csrc/bn254_capi.hip: parsers, key preparation, plans, the pinned ring of the host-buffer entry, context pools, host thread pool)
// compiles with g++ and runs under AddressSanitizer / UBSan in this GPU-less container (tests/hostsan/hostsan_main.cpp, tests/test_hostsan.py).  "Device" memory is
// malloc'ed, copies are memcpy, streams run in order at call time (every enqueue completes before it returns), events are time stamps.  Not part of the product.
#pragma once
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNoBinaryForGpu = 209, hipErrorInvalidDeviceFunction = 98, hipErrorNotReady = 600 };
struct FakeStream { int id; };
struct FakeEvent { double t; bool recorded; };
typedef FakeStream* hipStream_t;
typedef FakeEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
// (atomics: the library drives several "devices" from several host threads, and the harness is also built with -fsanitize=thread)
extern std::atomic<int> g_fake_device_count;          // tests set it (0: the no-device path)
extern std::atomic<size_t> g_fake_live_allocs;        // device + pinned allocations not yet freed
extern std::atomic<size_t> g_fake_fail_alloc_after;   // allocation number that fails (0: never): error paths
extern std::atomic<size_t> g_fake_alloc_counter;
extern std::atomic<int> g_fake_fail_device;           // every allocation made while this device is the calling thread's current one fails (-1: none): a GPU that is out of memory
extern thread_local int t_fake_current_device;        // hipSetDevice is per host thread, as in the real runtime
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorOutOfMemory ? "out of memory (fake)" : "error (fake)"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = g_fake_device_count; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= g_fake_device_count) return hipErrorInvalidDevice; t_fake_current_device = d; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = t_fake_current_device; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t fake_alloc(void** p, size_t n) {
  if (g_fake_fail_alloc_after && ++g_fake_alloc_counter == g_fake_fail_alloc_after) { *p = nullptr; return hipErrorOutOfMemory; }
  if (g_fake_fail_device >= 0 && t_fake_current_device == g_fake_fail_device) { *p = nullptr; return hipErrorOutOfMemory; }
  *p = malloc(n ? n : 1);                 // exact size: ASan sees every byte past the end
  if (!*p) return hipErrorOutOfMemory;
  g_fake_live_allocs++;
  return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { return fake_alloc(p, n); }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return fake_alloc((void**)p, n); }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return fake_alloc(p, n); }
inline hipError_t hipFree(void* p) { if (p) { free(p); g_fake_live_allocs--; } return hipSuccess; }
inline hipError_t hipHostFree(void* p) { return hipFree(p); }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new FakeStream{0}; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline double fake_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new FakeEvent{0, false}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = fake_now(); e->recorded = true; return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t e) { return e->recorded ? hipSuccess : hipErrorNotReady; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
