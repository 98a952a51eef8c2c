// common.h -- launch / error plumbing shared by the kernel translation units.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <stdexcept>
#include <string>
#include <vector>

#ifdef G16_EMU
// tests/emu/emu_hip.h is force-included by the emulation build (tests only)
#include <tuple>
#define G16_LAUNCH(kern, grid, block, smem, stream, ...)                                  \
  do {                                                                                    \
    auto _g16_args = std::make_tuple(__VA_ARGS__); /* by value, like a real launch */     \
    emu::launch(dim3(grid), dim3(block), (smem), [&]() {                                  \
      std::apply([](auto&&... _a) { kern(_a...); }, _g16_args);                           \
    });                                                                                   \
  } while (0)
#define G16_DYN_SMEM(name) unsigned char* name = emu::dyn_smem
#else
#include <hip/hip_runtime.h>
#define G16_LAUNCH(kern, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (smem), (stream), __VA_ARGS__)
#define G16_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

#include "ec.h"

namespace g16 {

struct HipError : std::runtime_error {
  int code;
  HipError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define G16_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      char _b[512];                                                                           \
      snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
               __LINE__);                                                                     \
      throw g16::HipError((int)_e, _b);                                                       \
    }                                                                                         \
  } while (0)

static inline uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// simple RAII device buffer
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    release();
    n = count;
    if (count) G16_HIP(hipMalloc((void**)&p, count * sizeof(T)));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  size_t bytes() const { return n * sizeof(T); }
};

// ------------------------------------------------------------------------------------------------
// per-stage HIP-event timing (bench.py reads it through g16_stage_times; off by default)
// ------------------------------------------------------------------------------------------------
enum Stage {
  ST_WITNESS_MAP = 0,  // sparse mat-vec + the six NTTs + pointwise (aux stream)
  ST_MSM_SORT,
  ST_MSM_ACC_G1,  // k_bucket_accumulate<Fq>, one query per launch (L, H) -- the dominant kernel
  ST_MSM_ACC_G2,  // k_bucket_accumulate<Fq2>
  ST_MSM_REDUCE,
  ST_FINALIZE,
  ST_MSM_ACC_G1_PAIR,  // k_bucket_accumulate<Fq, 2, true>: A and B1 in one launch (interleaved pair)
  ST_MSM_FIXUP,        // k_acc_fixup + the exact kernel behind an optimistic G1 launch (early exit unless the list overflowed)
  ST_MSM_TABLE_G1,     // small keys: k_tbl_msm<Fq> + k_tbl_final (fixed-base tables, msm_table.h)
  ST_MSM_TABLE_G2,
  ST_COUNT
};

struct StageTimer {
  bool enabled = false;
  struct Span {
    int stage;
    hipEvent_t a, b;
  };
  std::vector<Span> spans;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e;
    G16_HIP(hipEventCreate(&e));
    return e;
  }
  int begin(int stage, hipStream_t s) {
    if (!enabled) return -1;
    Span sp{stage, get(), get()};
    G16_HIP(hipEventRecord(sp.a, s));
    spans.push_back(sp);
    return (int)spans.size() - 1;
  }
  void end(int id, hipStream_t s) {
    if (id < 0) return;
    G16_HIP(hipEventRecord(spans[id].b, s));
  }
  // call after the stream is synchronised; accumulates ms and launch counts, recycles events
  void collect(float* ms, uint32_t* counts) {
    for (auto& sp : spans) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) {
        ms[sp.stage] += t;
        counts[sp.stage] += 1;
      }
      pool.push_back(sp.a);
      pool.push_back(sp.b);
    }
    spans.clear();
  }
  ~StageTimer() {
    for (auto& sp : spans) {
      (void)hipEventDestroy(sp.a);
      (void)hipEventDestroy(sp.b);
    }
    for (auto e : pool) (void)hipEventDestroy(e);
  }
};

}  // namespace g16
