// instr_rates.hip -- gfx950 VALU issue-rate microbenchmarks for the instructions a 254-bit
// Montgomery multiplication can be built from.  Measurement only.
//   hipcc --offload-arch=gfx950 -O3 -o instr_rates instr_rates.hip && ./instr_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int UNROLL = 16;
constexpr int THREADS = 256;

// 16 independent chains; each asm statement is one instruction on its own accumulator
#define DEF_KERNEL(NAME, DECL, INIT, FOLD, ASM, ...)                                          \
  __global__ void __launch_bounds__(THREADS) NAME(uint64_t* out, uint32_t iters, uint32_t seed) { \
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;                            \
    DECL acc[UNROLL];                                                                     \
    uint32_t x = t * 2654435761u + seed, y = (t ^ 0x5bd1e995u) | 1u;                      \
    for (int k = 0; k < UNROLL; ++k) acc[k] = INIT;                                       \
    for (uint32_t i = 0; i < iters; ++i) {                                                \
      _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) asm volatile(ASM : __VA_ARGS__);      \
    }                                                                                     \
    uint64_t s = 0;                                                                       \
    for (int k = 0; k < UNROLL; ++k) s += FOLD;                                           \
    out[t] = s;                                                                           \
  }

DEF_KERNEL(k_mad_u64_u32, uint64_t, (uint64_t)(t + k), (acc[k]), "v_mad_u64_u32 %0, vcc, %1, %2, %0", "+v"(acc[k]) : "v"(x), "v"(y) : "vcc")
DEF_KERNEL(k_mul_lo_u32, uint32_t, (t + k), (uint64_t)acc[k], "v_mul_lo_u32 %0, %0, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_mul_hi_u32, uint32_t, (t + k) * 0x9e3779b9u, (uint64_t)acc[k], "v_mul_hi_u32 %0, %0, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_add_u32, uint32_t, (t + k), (uint64_t)acc[k], "v_add_u32 %0, %0, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_add_co_u32, uint32_t, (t + k), (uint64_t)acc[k], "v_add_co_u32 %0, vcc, %0, %1", "+v"(acc[k]) : "v"(y) : "vcc")
DEF_KERNEL(k_addc_co_u32, uint32_t, (t + k), (uint64_t)acc[k], "v_addc_co_u32 %0, vcc, %0, %1, vcc", "+v"(acc[k]) : "v"(y) : "vcc")
DEF_KERNEL(k_lshl_add_u64, uint64_t, (uint64_t)(t + k), (acc[k]), "v_lshl_add_u64 %0, %0, 0, %1", "+v"(acc[k]) : "v"((uint64_t)x))
DEF_KERNEL(k_mov_b32, uint32_t, (t + k), (uint64_t)acc[k], "v_mov_b32 %0, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_fma_f64, double, (double)(t + k), (uint64_t)acc[k], "v_fma_f64 %0, %0, %1, %2", "+v"(acc[k]) : "v"(1.0000001), "v"(0.5))
DEF_KERNEL(k_fma_f32, float, (float)(t + k), (uint64_t)acc[k], "v_fma_f32 %0, %0, %1, %2", "+v"(acc[k]) : "v"(1.0000001f), "v"(0.5f))
DEF_KERNEL(k_mul_u32_u24, uint32_t, (t + k), (uint64_t)acc[k], "v_mul_u32_u24 %0, %0, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_mad_u32_u24, uint32_t, (t + k), (uint64_t)acc[k], "v_mad_u32_u24 %0, %0, %1, %0", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_mul_hi_u32_u24, uint32_t, (t + k), (uint64_t)acc[k], "v_mul_hi_u32_u24 %0, %0, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_mad_u64_u32_sgprcarry, uint64_t, (uint64_t)(t + k), (acc[k]), "v_mad_u64_u32 %0, s[20:21], %1, %2, %0", "+v"(acc[k]) : "v"(x), "v"(y) : "s20", "s21")
DEF_KERNEL(k_cndmask, uint32_t, (t + k), (uint64_t)acc[k], "v_cndmask_b32 %0, %0, %1, vcc", "+v"(acc[k]) : "v"(y) : "vcc")
DEF_KERNEL(k_mul_f64, double, (double)(t + k), (uint64_t)acc[k], "v_mul_f64 %0, %0, %1", "+v"(acc[k]) : "v"(1.0000001))
DEF_KERNEL(k_add_f64, double, (double)(t + k), (uint64_t)acc[k], "v_add_f64 %0, %0, %1", "+v"(acc[k]) : "v"(1.0000001))
DEF_KERNEL(k_mad_i32_i24, uint32_t, (t + k), (uint64_t)acc[k], "v_mad_i32_i24 %0, %0, %1, %0", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_add3_u32, uint32_t, (t + k), (uint64_t)acc[k], "v_add3_u32 %0, %0, %1, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_alignbit, uint32_t, (t + k), (uint64_t)acc[k], "v_alignbit_b32 %0, %0, %1, 29", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_pk_mul_lo_u16, uint32_t, (t + k), (uint64_t)acc[k], "v_pk_mul_lo_u16 %0, %0, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_pk_mad_u16, uint32_t, (t + k), (uint64_t)acc[k], "v_pk_mad_u16 %0, %0, %1, %0", "+v"(acc[k]) : "v"(y))

DEF_KERNEL(k_ashrrev_i64, uint64_t, (uint64_t)(t + k) * 0x9e3779b97f4a7c15ull, (acc[k]), "v_ashrrev_i64 %0, 1, %0", "+v"(acc[k]))
DEF_KERNEL(k_lshrrev_b64, uint64_t, (uint64_t)(t + k) * 0x9e3779b97f4a7c15ull, (acc[k]), "v_lshrrev_b64 %0, 1, %0", "+v"(acc[k]))
DEF_KERNEL(k_mad_i64_i32, uint64_t, (uint64_t)(t + k), (acc[k]), "v_mad_i64_i32 %0, vcc, %1, %2, %0", "+v"(acc[k]) : "v"(x), "v"(y) : "vcc")
DEF_KERNEL(k_and_b32, uint32_t, (t + k), (uint64_t)acc[k], "v_and_b32 %0, %0, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_sub_u32, uint32_t, (t + k), (uint64_t)acc[k], "v_sub_u32 %0, %0, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_ashrrev_i32, uint32_t, (t + k), (uint64_t)acc[k], "v_ashrrev_i32 %0, 1, %0", "+v"(acc[k]))
DEF_KERNEL(k_lshl_or_b32, uint32_t, (t + k), (uint64_t)acc[k], "v_lshl_or_b32 %0, %0, 3, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_and_or_b32, uint32_t, (t + k), (uint64_t)acc[k], "v_and_or_b32 %0, %0, %1, %1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_bfe_u32, uint32_t, (t + k), (uint64_t)acc[k], "v_bfe_u32 %0, %0, 3, 29", "+v"(acc[k]))
DEF_KERNEL(k_add_lshl_u32, uint32_t, (t + k), (uint64_t)acc[k], "v_add_lshl_u32 %0, %0, %1, 1", "+v"(acc[k]) : "v"(y))
DEF_KERNEL(k_mad_u64_u32_sgpr, uint64_t, (uint64_t)(t + k), (acc[k]), "v_mad_u64_u32 %0, vcc, %1, s8, %0", "+v"(acc[k]) : "v"(x) : "vcc", "s8")

// mixes: 1 mad + n cheap ops on DIFFERENT registers (do the cheap ops hide behind the mads?)
#define MIX_KERNEL(NAME, NCHEAP)                                                          \
  __global__ void __launch_bounds__(THREADS) NAME(uint64_t* out, uint32_t iters, uint32_t seed) { \
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;                            \
    uint64_t acc[UNROLL];                                                                 \
    uint32_t c[UNROLL];                                                                   \
    uint32_t x = t * 2654435761u + seed, y = (t ^ 0x5bd1e995u) | 1u;                      \
    for (int k = 0; k < UNROLL; ++k) { acc[k] = t + k; c[k] = t ^ k; }                    \
    for (uint32_t i = 0; i < iters; ++i) {                                                \
      _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) {                                \
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(x), "v"(y) : "vcc"); \
        _Pragma("unroll") for (int j = 0; j < NCHEAP; ++j)                                \
          asm volatile("v_and_b32 %0, %0, %1" : "+v"(c[(k + j) % UNROLL]) : "v"(y));       \
      }                                                                                   \
    }                                                                                     \
    uint64_t s = 0;                                                                       \
    for (int k = 0; k < UNROLL; ++k) s += acc[k] + c[k];                                  \
    out[t] = s;                                                                           \
  }
MIX_KERNEL(k_mix_mad_1and, 1)
MIX_KERNEL(k_mix_mad_2and, 2)
MIX_KERNEL(k_mix_mad_4and, 4)

// dependent chain latency: one accumulator
__global__ void __launch_bounds__(64) k_lat_mad(uint64_t* out, uint32_t iters) {
  uint64_t acc = threadIdx.x;
  uint32_t x = threadIdx.x * 2654435761u + 1, y = threadIdx.x | 1;
  for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
  }
  out[threadIdx.x] = acc;
}

typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t);
struct Entry { const char* name; kern_t k; };

int main() {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, dev));
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int blocks = p.multiProcessorCount * 8;
  uint64_t* out;
  CK(hipMalloc(&out, (size_t)blocks * THREADS * 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  Entry es[] = {{"v_mov_b32", k_mov_b32}, {"v_add_u32", k_add_u32}, {"v_add_co_u32", k_add_co_u32},
                {"v_addc_co_u32", k_addc_co_u32}, {"v_add3_u32", k_add3_u32}, {"v_cndmask_b32", k_cndmask},
                {"v_alignbit_b32", k_alignbit}, {"v_lshl_add_u64", k_lshl_add_u64},
                {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
                {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mad_u64_u32(sgpr carry)", k_mad_u64_u32_sgprcarry},
                {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24},
                {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mad_i32_i24", k_mad_i32_i24},
                {"v_pk_mul_lo_u16", k_pk_mul_lo_u16}, {"v_pk_mad_u16", k_pk_mad_u16},
                {"v_ashrrev_i64", k_ashrrev_i64}, {"v_lshrrev_b64", k_lshrrev_b64}, {"v_mad_i64_i32", k_mad_i64_i32}, {"v_mad_u64_u32(sgpr op)", k_mad_u64_u32_sgpr},
                {"v_and_b32", k_and_b32}, {"v_sub_u32", k_sub_u32}, {"v_ashrrev_i32", k_ashrrev_i32}, {"v_lshl_or_b32", k_lshl_or_b32},
                {"v_and_or_b32", k_and_or_b32}, {"v_bfe_u32", k_bfe_u32}, {"v_add_lshl_u32", k_add_lshl_u32},
                {"mix: 1 mad + 1 and (per mad)", k_mix_mad_1and}, {"mix: 1 mad + 2 and (per mad)", k_mix_mad_2and}, {"mix: 1 mad + 4 and (per mad)", k_mix_mad_4and},
                {"v_fma_f32", k_fma_f32}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}};
  const uint32_t iters = 4096;
  printf("%-28s %12s %14s %s\n", "instruction", "ms", "Ginstr/s/lane", "cycles/wave-instr/SIMD @2.4GHz");
  for (auto& e : es) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(THREADS), 0, 0, out, iters, 7u);
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    const double lane_ops = (double)blocks * THREADS * iters * UNROLL;
    const double rate = lane_ops / (best * 1e-3);
    // wave-instructions per second per SIMD
    const double wi = rate / 64.0 / (p.multiProcessorCount * 4.0);
    printf("%-28s %12.3f %14.1f %8.2f\n", e.name, best, rate / 1e9, 2.4e9 / wi);
  }
  {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_lat_mad, dim3(1), dim3(64), 0, 0, out, 65536u);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("v_mad_u64_u32 dependent latency: %.2f ns/instr (%.1f cyc @2.4GHz)\n", ms * 1e6 / (65536.0 * 16), ms * 1e6 / (65536.0 * 16) * 2.4);
  }
  return 0;
}
