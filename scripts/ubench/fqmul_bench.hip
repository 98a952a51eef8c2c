// fqmul_bench.hip -- throughput of the two Fq Montgomery products on gfx950 (measurement only):
// field.h (8 x 32-bit saturated limbs, CIOS) vs field29.h (9 x 29-bit signed lazy limbs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../circom_compat_amd/csrc -o fqmul_bench fqmul_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "field29.h"
using namespace g16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_legacy(Fq* out, uint32_t iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = Fq::from_u32(t + 3), b = Fq::from_u32(t + 5), c = Fq::from_u32(t + 7), d = Fq::from_u32(t + 11);
  const Fq m = Fq::from_u32(0x9e3779b9u);
  for (uint32_t i = 0; i < iters; ++i) {
    a = a * m; b = b * a; c = c * b; d = d * c;
  }
  out[t] = a + b + c + d;
}
__global__ void __launch_bounds__(256) k_f29(Fq* out, uint32_t iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 a = Fq29::from_mont256(Fq::from_u32(t + 3)), b = Fq29::from_mont256(Fq::from_u32(t + 5)),
       c = Fq29::from_mont256(Fq::from_u32(t + 7)), d = Fq29::from_mont256(Fq::from_u32(t + 11));
  const Fq29 m = Fq29::from_mont256(Fq::from_u32(0x9e3779b9u));
  for (uint32_t i = 0; i < iters; ++i) {
    a = a * m; b = b * a; c = c * b; d = d * c;
  }
  out[t] = (a + b + c + d).to_mont256();
}
__global__ void __launch_bounds__(256) k_f29_sqr(Fq* out, uint32_t iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 a = Fq29::from_mont256(Fq::from_u32(t + 3)), b = Fq29::from_mont256(Fq::from_u32(t + 5)),
       c = Fq29::from_mont256(Fq::from_u32(t + 7)), d = Fq29::from_mont256(Fq::from_u32(t + 11));
  for (uint32_t i = 0; i < iters; ++i) {
    a = a.sqr(); b = b.sqr(); c = c.sqr(); d = d.sqr();
  }
  out[t] = (a + b + c + d).to_mont256();
}
__global__ void __launch_bounds__(256) k_legacy_sqr(Fq* out, uint32_t iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = Fq::from_u32(t + 3), b = Fq::from_u32(t + 5), c = Fq::from_u32(t + 7), d = Fq::from_u32(t + 11);
  for (uint32_t i = 0; i < iters; ++i) {
    a = a.sqr(); b = b.sqr(); c = c.sqr(); d = d.sqr();
  }
  out[t] = a + b + c + d;
}

int main() {
  CK(hipSetDevice(0));
  const int blocks = 4096, threads = 256;
  const uint32_t iters = 512;
  Fq *o1, *o2;
  CK(hipMalloc(&o1, sizeof(Fq) * blocks * threads));
  CK(hipMalloc(&o2, sizeof(Fq) * blocks * threads));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  auto time = [&](void (*k)(Fq*, uint32_t), Fq* o, const char* name) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a, 0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, o, iters);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    double ops = 4.0 * iters * blocks * threads;
    printf("%-12s %8.3f ms  %8.1f G mul/s\n", name, best, ops / best / 1e6);
  };
  time(k_legacy, o1, "legacy mul");
  time(k_f29, o2, "f29 mul");
  std::vector<Fq> h1((size_t)blocks * threads), h2(h1.size());
  CK(hipMemcpy(h1.data(), o1, h1.size() * sizeof(Fq), hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), o2, h2.size() * sizeof(Fq), hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < h1.size(); ++i) bad += !(h1[i] == h2[i]);
  printf("mul chains: %zu mismatches of %zu\n", bad, h1.size());
  time(k_legacy_sqr, o1, "legacy sqr");
  time(k_f29_sqr, o2, "f29 sqr");
  CK(hipMemcpy(h1.data(), o1, h1.size() * sizeof(Fq), hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), o2, h2.size() * sizeof(Fq), hipMemcpyDeviceToHost));
  bad = 0;
  for (size_t i = 0; i < h1.size(); ++i) bad += !(h1[i] == h2[i]);
  printf("sqr chains: %zu mismatches of %zu\n", bad, h1.size());
  return bad != 0;
}
