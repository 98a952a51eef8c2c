// fqmul_chain.hip -- round 6 experiment: the lazy 29-bit product with the column carry fed
// into the next column's FIRST multiply-add (inline-asm v_mad_i64_i32 chain: the compiler cannot re-associate it)
// against field29.h's product, where LLVM sums every column from zero and adds the carry with a v_lshl_add_u64 (17
// extra half-rate instructions per product, ~7 % of the accumulation kernel's issue slots).  The chained form is ONE
// dependent chain of 171 multiply-adds per product: it only pays if enough independent products are in flight.
// ILP = independent products per thread (1, 2, 4): the mixed addition offers 2-3.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../circom_compat_amd/csrc -o fqmul_chain fqmul_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "field29.h"
using namespace g16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void mad_vv(int64_t& acc, int32_t a, int32_t b) {
  uint64_t sd;
  asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sd) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mad_vs(int64_t& acc, int32_t a, int32_t b) {
  uint64_t sd;
  asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sd) : "v"(a), "s"(b));
}

__device__ __forceinline__ Fq29 mul_chain(const Fq29& a, const Fq29& b) {
  constexpr int N = f29::N;
  using C = Fq29::C;
  int64_t acc = 0;
  int32_t m[N];
  Fq29 r;
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) mad_vv(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; ++i) mad_vs(acc, m[i], C::MOD.v[k - i]);
    m[k] = (int32_t)(((uint32_t)acc * C::NINV) & f29::MASK);
    mad_vs(acc, m[k], C::MOD.v[0]);
    acc >>= 29;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
    for (int i = k - N + 1; i < N; ++i) mad_vv(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - N + 1; i < N; ++i) mad_vs(acc, m[i], C::MOD.v[k - i]);
    r.l[k - N] = (int32_t)((uint32_t)acc & f29::MASK);
    acc >>= 29;
  }
  r.l[N - 1] = (int32_t)acc;
  return r;
}

template <int ILP, bool CHAIN>
__global__ void __launch_bounds__(128) k_mul(Fq* out, uint32_t iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 x[ILP];
#pragma unroll
  for (int j = 0; j < ILP; ++j) x[j] = Fq29::from_mont256(Fq::from_u32(t * 4 + 3 + j));
  const Fq29 m = Fq29::from_mont256(Fq::from_u32(0x9e3779b9u));
  for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < ILP; ++j) x[j] = CHAIN ? mul_chain(x[j], m) : x[j] * m;
  }
  Fq29 s = x[0];
#pragma unroll
  for (int j = 1; j < ILP; ++j) s = (s + x[j]).carry();
  out[t] = s.to_mont256();
}

template <int ILP>
int run(Fq* o1, Fq* o2, int blocks) {
  const int threads = 128;
  const uint32_t iters = 512;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best[2] = {1e30f, 1e30f};
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_mul<ILP, false>), dim3(blocks), dim3(threads), 0, 0, o1, iters);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best[0]) best[0] = ms;
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_mul<ILP, true>), dim3(blocks), dim3(threads), 0, 0, o2, iters);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best[1]) best[1] = ms;
  }
  const size_t n = (size_t)blocks * threads;
  std::vector<Fq> h1(n), h2(n);
  CK(hipMemcpy(h1.data(), o1, n * sizeof(Fq), hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), o2, n * sizeof(Fq), hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) bad += h1[i] != h2[i];
  const double prods = (double)n * iters * ILP;
  printf("ILP %d grid %5d: field29.h %8.3f ms %7.1f G products/s | chained %8.3f ms %7.1f G products/s (%.2fx) mismatches %zu\n", ILP, blocks,
         best[0], prods / best[0] / 1e6, best[1], prods / best[1] / 1e6, best[0] / best[1], bad);
  return 0;
}

#include <vector>
int main() {
  CK(hipSetDevice(0));
  const int maxb = 8192;
  Fq *o1, *o2;
  CK(hipMalloc(&o1, sizeof(Fq) * maxb * 128));
  CK(hipMalloc(&o2, sizeof(Fq) * maxb * 128));
  for (int blocks : {2048, 6144, 8192}) {
    if (run<1>(o1, o2, blocks)) return 1;
    if (run<2>(o1, o2, blocks)) return 1;
    if (run<4>(o1, o2, blocks)) return 1;
  }
  return 0;
}
