// stream_conc.hip -- do kernels on different HIP streams overlap on this box? (measurement only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(long long cycles, int* out) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (out) out[0] = 1;
}
int main() {
  CK(hipSetDevice(0));
  const int NS = 6;
  hipStream_t s[NS];
  for (int i = 0; i < NS; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const long long cyc = 100000000LL / 10;  // wall_clock64 is 100 MHz: 10M ticks = 100 ms ... use 1M = 10 ms
  for (int n = 1; n <= NS; ++n) {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], 1000000LL, nullptr);
    CK(hipDeviceSynchronize());
    auto t1 = std::chrono::steady_clock::now();
    printf("%d streams x 10 ms single-block kernels: %.2f ms wall\n", n, std::chrono::duration<double, std::milli>(t1 - t0).count());
  }
  // pair (i, j): which stream pairs serialize?
  for (int i = 0; i < NS; ++i) for (int j = i + 1; j < NS; ++j) {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], 1000000LL, nullptr);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[j], 1000000LL, nullptr);
    CK(hipDeviceSynchronize());
    auto t1 = std::chrono::steady_clock::now();
    printf("pair (%d,%d): %.2f ms\n", i, j, std::chrono::duration<double, std::milli>(t1 - t0).count());
  }
  return 0;
}
