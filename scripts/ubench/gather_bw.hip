// gather_bw.hip -- random-gather bandwidth of MI355X HBM for the access shapes of the MSM bucket
// kernel: every lane reads one 64-byte (G1 affine) or 128-byte (G2 affine) record at a
// pseudo-random index of a multi-GB table.  Measurement only.
//   hipcc --offload-arch=gfx950 -O3 -o gather_bw gather_bw.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int REC16>  // record size in 16-byte units
__global__ void __launch_bounds__(256) k_gather(const uint4* tbl, uint64_t nrec, uint32_t iters, uint4* out, int indep) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t x = t * 0x9E3779B97F4A7C15ull + 12345;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint32_t i = 0; i < iters; ++i) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    uint64_t idx = ((x >> 20) + (indep ? 0 : acc.x)) % nrec;  // indep == 0: dependent chain (latency)
    const uint4* p = tbl + idx * REC16;
#pragma unroll
    for (int k = 0; k < REC16; ++k) {
      uint4 v = p[k];
      acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
  }
  out[t] = acc;
}

int main() {
  CK(hipSetDevice(0));
  const size_t bytes = (size_t)4 << 30;
  uint4 *tbl, *out;
  CK(hipMalloc(&tbl, bytes));
  CK(hipMemset(tbl, 1, bytes));
  const int blocks = 256 * 12, threads = 256;
  CK(hipMalloc(&out, (size_t)blocks * threads * 16));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int rec = 64; rec <= 128; rec *= 2) {
    for (int indep = 1; indep >= 0; --indep) {
      const uint32_t iters = indep ? 256 : 64;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, 0));
        if (rec == 64) hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(threads), 0, 0, tbl, bytes / 64, iters, out, indep);
        else hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(threads), 0, 0, tbl, bytes / 128, iters, out, indep);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      const double n = (double)blocks * threads * iters;
      printf("record %3d B %s: %8.3f ms  %7.1f G records/s  %7.1f GB/s", rec, indep ? "independent" : "dependent  ", best, n / best / 1e6, n * rec / best / 1e6);
      if (!indep) printf("   (latency %.0f ns per dependent gather)", best * 1e6 / iters);
      printf("\n");
    }
  }
  return 0;
}
