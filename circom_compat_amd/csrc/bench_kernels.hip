// bench_kernels.hip -- measurement-only kernels: the integer-ALU ceilings that bound the MSM and
// NTT kernels.  v_mad_u64_u32 issue rate is not in the local CDNA4 guides, so
// it is measured here; bench.py / scripts report MSM time against it next to the HBM roofline.
// Compiled only into a measurement build (make EXTRA=-DG16_DEBUG_ABI): the product library does not export it.
#ifdef G16_DEBUG_ABI
#include "../../include/g16_amd.h"
#include "common.h"
#include "ec29.h"

using namespace g16;

namespace {

// kind 0: dependent chains of Fq Montgomery multiplications (4 independent chains per thread)
__global__ void __launch_bounds__(256) k_bench_fqmul(Fq* out, uint32_t iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = Fq::from_u32(t + 3), b = Fq::from_u32(t + 5), c = Fq::from_u32(t + 7), d = Fq::from_u32(t + 11);
  const Fq m = Fq::from_u32(0x9e3779b9u);
  for (uint32_t i = 0; i < iters; ++i) {
    a = a * m;
    b = b * a;
    c = c * b;
    d = d * c;
  }
  out[t] = a + b + c + d;
}

// kind 1: raw v_mad_u64_u32 throughput, 8 independent accumulators per thread
__global__ void __launch_bounds__(256) k_bench_mad(uint64_t* out, uint32_t iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc[8];
  uint32_t x = t * 2654435761u + 12345u, y = t ^ 0x5bd1e995u;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = t + k;
  for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = (uint64_t)x * (uint32_t)(y + k) + acc[k];
    x += (uint32_t)acc[0];
    y ^= (uint32_t)(acc[7] >> 32);
  }
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += acc[k];
  out[t] = s;
}

// kind 2: G1 mixed additions (XYZZ += affine), the inner operation of k_bucket_accumulate
__global__ void __launch_bounds__(128) k_bench_madd(G1XYZZ* out, uint32_t iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  // arbitrary field values: madd's cost does not depend on the operands being on the curve
  G1Affine p{Fq::from_u32(t + 1), Fq::from_u32(t + 2)};
  G1XYZZ acc{Fq::from_u32(t + 3), Fq::from_u32(t + 4), Fq::from_u32(t + 5), Fq::from_u32(t + 6)};
  for (uint32_t i = 0; i < iters; ++i) {
    acc.madd(p);
    p.x = p.x + acc.zz;
  }
  out[t] = acc;
}

// kind 3: the same on the lazy 9 x 29-bit limbs (what k_bucket_accumulate<Fq> executes)
__global__ void __launch_bounds__(128) k_bench_madd29(G1XYZZ29* out, uint32_t iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Aff29<Fq29> p{Fq29::from_mont256(Fq::from_u32(t + 1)), Fq29::from_mont256(Fq::from_u32(t + 2)), false};
  G1XYZZ29 acc{Fq29::from_mont256(Fq::from_u32(t + 3)), Fq29::from_mont256(Fq::from_u32(t + 4)),
               Fq29::from_mont256(Fq::from_u32(t + 5)), Fq29::from_mont256(Fq::from_u32(t + 6))};
  for (uint32_t i = 0; i < iters; ++i) {
    acc.madd(p);
    p.x = (p.x + acc.zz).carry();
  }
  out[t] = acc;
}
// kind 4: G2 lazy mixed additions
__global__ void __launch_bounds__(128) k_bench_madd29_g2(G2XYZZ29* out, uint32_t iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  auto f = [&](uint32_t a) { return Fq2x29{Fq29::from_mont256(Fq::from_u32(t + a)), Fq29::from_mont256(Fq::from_u32(t + a + 100))}; };
  Aff29<Fq2x29> p{f(1), f(2), false};
  G2XYZZ29 acc{f(3), f(4), f(5), f(6)};
  for (uint32_t i = 0; i < iters; ++i) {
    acc.madd(p);
    p.x = (p.x + acc.zz).carry();
  }
  out[t] = acc;
}

}  // namespace

extern "C" g16_status g16_debug_alu_bench(int device, int kind, uint32_t blocks, uint32_t iters,
                                          double* seconds, double* ops) {
  if (!seconds || !ops) return G16_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return G16_ERR_NO_DEVICE;
  try {
    G16_HIP(hipSetDevice(device));
    hipEvent_t a, b;
    G16_HIP(hipEventCreate(&a));
    G16_HIP(hipEventCreate(&b));
    const uint32_t threads = kind >= 2 ? 128 : 256;
    DevBuf<uint8_t> buf;
    buf.alloc((size_t)blocks * threads * 320);
    for (int rep = 0; rep < 2; ++rep) {  // first repetition warms up
      G16_HIP(hipEventRecord(a, nullptr));
      if (kind == 0) G16_LAUNCH(k_bench_fqmul, blocks, threads, 0, nullptr, (Fq*)buf.p, iters);
      else if (kind == 1) G16_LAUNCH(k_bench_mad, blocks, threads, 0, nullptr, (uint64_t*)buf.p, iters);
      else if (kind == 2) G16_LAUNCH(k_bench_madd, blocks, threads, 0, nullptr, (G1XYZZ*)buf.p, iters);
      else if (kind == 3) G16_LAUNCH(k_bench_madd29, blocks, threads, 0, nullptr, (G1XYZZ29*)buf.p, iters);
      else G16_LAUNCH(k_bench_madd29_g2, blocks, threads, 0, nullptr, (G2XYZZ29*)buf.p, iters);
      G16_HIP(hipEventRecord(b, nullptr));
      G16_HIP(hipEventSynchronize(b));
    }
    float ms = 0.f;
    G16_HIP(hipEventElapsedTime(&ms, a, b));
    *seconds = ms * 1e-3;
    const double per_thread = kind == 0 ? 4.0 * iters : (kind == 1 ? 8.0 * iters : 1.0 * iters);
    *ops = per_thread * blocks * threads;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return G16_OK;
  } catch (const std::exception&) {
    return G16_ERR_HIP;
  }
}
#endif  // G16_DEBUG_ABI
