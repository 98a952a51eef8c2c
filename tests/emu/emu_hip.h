// emu_hip.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-process stand-in for the HIP runtime and the SIMT execution model so that the
// *same kernel sources* under circom_compat_amd/csrc can be compiled with g++ and stepped through
// on the CPU by the `-m "not gpu"` tests (there is no GPU in the build container).  It exists to
// catch indexing / algorithm bugs before a GPU run; it is never linked into, loaded by or reachable
// from the product library (libg16_amd.so), and nothing it produces is reported as a GPU result.
//
// Execution model: blocks run one after another; the threads of a block are ucontext fibers that
// run round-robin and switch at __syncthreads() / wave collectives.  Atomics are plain operations
// (single OS thread).  __shared__ becomes `static` (valid because blocks are sequential).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define G16_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
typedef struct emuStream* hipStream_t;
typedef struct emuEvent* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorPeerAccessUnsupported = 217,
       hipErrorPeerAccessAlreadyEnabled = 704 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
#define hipHostMallocPortable 1

namespace emu {
void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body);
void barrier();                 // __syncthreads
uint64_t wave_ballot(bool p);   // 64-lane collectives (all lanes of the wave must participate)
uint32_t wave_shfl(uint32_t v, int src);
extern unsigned char* dyn_smem;
}  // namespace emu

static inline void __syncthreads() { emu::barrier(); }
static inline unsigned long long __ballot(int p) { return emu::wave_ballot(p != 0); }
static inline uint32_t __shfl(uint32_t v, int src) { return emu::wave_shfl(v, src); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline unsigned __brev(unsigned x) {
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
  x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
  return (x >> 16) | (x << 16);
}

template <class T>
static inline T atomicAdd(T* p, T v) {
  T o = *p;
  *p = o + v;
  return o;
}
template <class T>
static inline T atomicMin(T* p, T v) {
  T o = *p;
  if (v < o) *p = v;
  return o;
}
template <class T>
static inline T atomicMax(T* p, T v) {
  T o = *p;
  if (v > o) *p = v;
  return o;
}

// ---- runtime API subset ----
static inline hipError_t hipMalloc(void** p, size_t n) {
  *p = n ? calloc(1, n + 64) : nullptr;  // zeroed: deterministic tests; real HBM is not zeroed
  if (n && !*p) return hipErrorOutOfMemory;
  if (*p) memset(*p, 0xA5, n);           // poison instead, to catch reliance on zero-init
  return hipSuccess;
}
template <class T>
static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n); return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = 0) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t = 0) {
  for (size_t r = 0; r < height; ++r) memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t = 0) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = 0) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
// G16_EMU_FREE_BYTES (tests only): the free device memory the emulator reports -- lets the CPU suite
// drive the ctx's memory plan (api.hip, plan_msm_configs) into its fewer-planes fallback
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) {
  const char* e = getenv("G16_EMU_FREE_BYTES");
  *f = *t = e ? (size_t)strtoull(e, nullptr, 0) : (size_t)1 << 34;
  return hipSuccess;
}
