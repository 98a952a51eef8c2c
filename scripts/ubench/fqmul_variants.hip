// fqmul_variants.hip -- round-4 microbenchmark: is there a faster Fq Montgomery
// product on gfx950 than field29.h's 9 x 29-bit lazy limbs?  Measurement only, never linked into the
// library.  Variants, all as dependent-chain loops of 4 independent products per thread (the shape of
// scripts/ubench/fqmul_bench.hip, so the numbers are comparable with profiles/r01..r02):
//   f29      field29.h operator*  (the product the kernels use: 171 multiply-type instructions)
//   f29sqr   field29.h sqr()
//   legacy   field.h 8 x 32-bit CIOS
//   dpf52    5 x 52-bit limbs held in doubles, every limb product split into (hi, lo) halves by two
//            v_fma_f64 in round-toward-zero mode ("DPF": Emmart, Zheng, Weems, ARITH 2018), the halves
//            accumulated as 64-bit integers on the raw bit patterns; Montgomery radix 2^260
// The dpf52 product is checked against field.h on every thread (x y 2^-260 * 16 == x y 2^-256).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../circom_compat_amd/csrc -o fqmul_variants fqmul_variants.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "field29.h"
using namespace g16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// ---- 5 x 52-bit limbs in doubles -----------------------------------------------------------------
struct D5 {
  double l[5];
};
constexpr uint64_t M52 = (1ull << 52) - 1;
__device__ __forceinline__ uint64_t dbits(double x) { return (uint64_t)__double_as_longlong(x); }
__device__ __forceinline__ double bitsd(uint64_t x) { return __longlong_as_double((long long)x); }
// integer < 2^52 -> double: OR into the mantissa of 2^52, subtract 2^52 (exact)
__device__ __forceinline__ double u52_to_d(uint64_t v) { return bitsd(v | 0x4330000000000000ull) - 4503599627370496.0; }

struct DpfConst {
  double p[5];   // modulus limbs
  double ninv;   // -p^-1 mod 2^52
};

// FP64 operations as opaque instructions: LLVM's SIModeRegister pass re-asserts the DEFAULT rounding
// mode (s_setreg ... FP_ROUND, 0) in front of every FP64 instruction it can see once a kernel has
// touched the MODE register, which silently undoes set_rtz_f64() below (first version of this file:
// 65535 of 65536 products wrong).  Inline asm is invisible to that pass; `volatile` keeps the
// program order, so the independent operations of a row are written interleaved by hand.
__device__ __forceinline__ double dfma(double a, double b, double c) {
  double r;
  asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ double dsub(double a, double b) {  // a - b
  double r;
  asm volatile("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double u52_to_d_rtz(uint64_t v) { return dsub(bitsd(v | 0x4330000000000000ull), 4503599627370496.0); }

// a b / 2^260 mod p, result limbs < 2^52, value < 2p for inputs < 2^256.  Needs FP_ROUND(double) = RTZ.
__device__ __forceinline__ D5 dpf_mul(const D5& a, const D5& b, const DpfConst& K) {
  const double C1 = 20282409603651670423947251286016.0;              // 2^104
  const double C4 = 20282409603651670423947251286016.0 + 4503599627370496.0;  // 2^104 + 2^52 (53 significant bits)
  const uint64_t BH = 0x4670000000000000ull;  // bit pattern of 2^104
  const uint64_t BL = 0x4330000000000000ull;  // bit pattern of 2^52
  uint64_t acc[11];
  uint32_t nlo[11], nhi[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    acc[k] = 0;
    nlo[k] = nhi[k] = 0;
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    double hi[5], sb[5], lo[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) hi[j] = dfma(a.l[j], b.l[i], C1);       // floor(a b / 2^52) 2^52 + 2^104 (RTZ)
#pragma unroll
    for (int j = 0; j < 5; ++j) sb[j] = dsub(C4, hi[j]);
#pragma unroll
    for (int j = 0; j < 5; ++j) lo[j] = dfma(a.l[j], b.l[i], sb[j]);    // (a b mod 2^52) + 2^52
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      acc[i + j] += dbits(lo[j]);
      ++nlo[i + j];
      acc[i + j + 1] += dbits(hi[j]);
      ++nhi[i + j + 1];
    }
    // column i holds all of its a b terms and the q p terms of the earlier rows: q_i = -col / p mod 2^52
    const uint64_t col = acc[i] - (uint64_t)nlo[i] * BL - (uint64_t)nhi[i] * BH;
    const double tl = u52_to_d_rtz(col & M52);
    const double qh = dfma(tl, K.ninv, C1);
    const double q = dsub(dfma(tl, K.ninv, dsub(C4, qh)), 4503599627370496.0);  // low 52 bits of tl * ninv
#pragma unroll
    for (int j = 0; j < 5; ++j) hi[j] = dfma(q, K.p[j], C1);
#pragma unroll
    for (int j = 0; j < 5; ++j) sb[j] = dsub(C4, hi[j]);
#pragma unroll
    for (int j = 0; j < 5; ++j) lo[j] = dfma(q, K.p[j], sb[j]);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      acc[i + j] += dbits(lo[j]);
      ++nlo[i + j];
      acc[i + j + 1] += dbits(hi[j]);
      ++nhi[i + j + 1];
    }
    // column i is now 0 mod 2^52: carry the rest up
    const uint64_t col2 = acc[i] - (uint64_t)nlo[i] * BL - (uint64_t)nhi[i] * BH;
    acc[i + 1] += col2 >> 52;
  }
  D5 r;
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const uint64_t col = acc[5 + k] - (uint64_t)nlo[5 + k] * BL - (uint64_t)nhi[5 + k] * BH + carry;
    r.l[k] = u52_to_d_rtz(col & M52);
    carry = col >> 52;
  }
  return r;
}

__device__ __forceinline__ void set_rtz_f64() {
  // MODE register (hwreg id 1), FP_ROUND[3:2] = double/half rounding: 3 = toward zero
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
}

__device__ __forceinline__ D5 d5_from_words(const uint32_t (&w)[8]) {
  uint64_t q[4];
  for (int i = 0; i < 4; ++i) q[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  D5 r;
  r.l[0] = u52_to_d(q[0] & M52);
  r.l[1] = u52_to_d(((q[0] >> 52) | (q[1] << 12)) & M52);
  r.l[2] = u52_to_d(((q[1] >> 40) | (q[2] << 24)) & M52);
  r.l[3] = u52_to_d(((q[2] >> 28) | (q[3] << 36)) & M52);
  r.l[4] = u52_to_d(q[3] >> 16);
  return r;
}
__device__ __forceinline__ void d5_to_words(const D5& a, uint32_t (&w)[8]) {
  uint64_t l[5];
  for (int i = 0; i < 5; ++i) l[i] = (uint64_t)a.l[i];
  uint64_t q[4];
  q[0] = l[0] | (l[1] << 52);
  q[1] = (l[1] >> 12) | (l[2] << 40);
  q[2] = (l[2] >> 24) | (l[3] << 28);
  q[3] = (l[3] >> 36) | (l[4] << 16);
  for (int i = 0; i < 4; ++i) {
    w[2 * i] = (uint32_t)q[i];
    w[2 * i + 1] = (uint32_t)(q[i] >> 32);
  }
}

__global__ void __launch_bounds__(256) k_dpf(D5* out, uint32_t iters, DpfConst K) {
  set_rtz_f64();
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  D5 a, b, c, d, m;
  for (int i = 0; i < 5; ++i) {
    a.l[i] = (double)(t + 3 + i);
    b.l[i] = (double)(t + 5 + 2 * i);
    c.l[i] = (double)(t + 7 + 3 * i);
    d.l[i] = (double)(t + 11 + 5 * i);
    m.l[i] = (double)(0x9e3779b9u + i);
  }
  for (uint32_t i = 0; i < iters; ++i) {
    a = dpf_mul(a, m, K);
    b = dpf_mul(b, a, K);
    c = dpf_mul(c, b, K);
    d = dpf_mul(d, c, K);
  }
  D5 r;
  for (int i = 0; i < 5; ++i) r.l[i] = a.l[i] + b.l[i] + c.l[i] + d.l[i];
  out[t] = r;
}

// one product per thread, checked against field.h: dpf(x, y) * 16 == x y / 2^256 (both canonical)
__global__ void __launch_bounds__(256) k_dpf_check(const Fq* xs, const Fq* ys, uint32_t n, DpfConst K, uint32_t* bad) {
  set_rtz_f64();
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Fq x = xs[t], y = ys[t];  // the 8 words read as plain integers < q
  D5 z = dpf_mul(d5_from_words(x.v), d5_from_words(y.v), K);
  z = dpf_mul(z, z, K);            // a second, dependent product (inputs < 2p: the lazy range)
  Fq got;
  d5_to_words(z, got.v);           // value < 2q in 8 words
  got = Fq::reduce_once(got);
  // reference: ((x y / 2^256)^2 / 2^256); dpf gave ((x y / 2^260)^2 / 2^260) = ref / 2^12
  Fq ref = x * y;
  ref = ref * ref;
  Fq g = got;
  for (int i = 0; i < 12; ++i) g = g + g;
  if (g != ref) atomicAdd(bad, 1u);
}

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

static uint64_t neg_inv52(uint64_t p0) {
  uint64_t x = p0;
  for (int i = 0; i < 6; ++i) x *= 2 - p0 * x;
  return (0 - x) & M52;
}

int main() {
  CK(hipSetDevice(0));
  const int blocks = 4096, threads = 256;
  const uint32_t iters = 512;
  const size_t nthr = (size_t)blocks * threads;
  Fq* o1;
  D5* o2;
  CK(hipMalloc(&o1, sizeof(Fq) * nthr));
  CK(hipMalloc(&o2, sizeof(D5) * nthr));
  hipEvent_t ea, eb;
  CK(hipEventCreate(&ea));
  CK(hipEventCreate(&eb));

  DpfConst K;
  {
    uint64_t q[4];
    for (int i = 0; i < 4; ++i) q[i] = (uint64_t)FqParams::MOD[2 * i] | ((uint64_t)FqParams::MOD[2 * i + 1] << 32);
    const uint64_t l[5] = {q[0] & M52, ((q[0] >> 52) | (q[1] << 12)) & M52, ((q[1] >> 40) | (q[2] << 24)) & M52,
                           ((q[2] >> 28) | (q[3] << 36)) & M52, q[3] >> 16};
    for (int i = 0; i < 5; ++i) K.p[i] = (double)l[i];
    K.ninv = (double)neg_inv52(l[0]);
  }

  // correctness of the dpf product first
  {
    const uint32_t n = 1 << 16;
    std::vector<Fq> xs(n), ys(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
    for (uint32_t i = 0; i < n; ++i) {
      for (int k = 0; k < 8; ++k) { xs[i].v[k] = rnd(); ys[i].v[k] = rnd(); }
      xs[i].v[7] &= 0x1fffffffu;  // < q
      ys[i].v[7] &= 0x1fffffffu;
    }
    memset(&xs[0], 0, sizeof(Fq));                       // 0 * y
    for (int k = 0; k < 8; ++k) xs[1].v[k] = FqParams::MOD[k];
    xs[1].v[0] -= 1;                                     // q - 1
    ys[1] = xs[1];
    Fq *dx, *dy;
    uint32_t* dbad;
    CK(hipMalloc(&dx, n * sizeof(Fq)));
    CK(hipMalloc(&dy, n * sizeof(Fq)));
    CK(hipMalloc(&dbad, 4));
    CK(hipMemset(dbad, 0, 4));
    CK(hipMemcpy(dx, xs.data(), n * sizeof(Fq), hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, ys.data(), n * sizeof(Fq), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dpf_check, dim3(n / 256), dim3(256), 0, 0, (const Fq*)dx, (const Fq*)dy, n, K, dbad);
    uint32_t bad = 0;
    CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
    printf("dpf52 product vs field.h: %u mismatches of %u\n", bad, n);
  }

  auto report = [&](const char* name, float best) {
    const double ops = 4.0 * iters * nthr;
    printf("%-10s %8.3f ms  %8.1f G products/s\n", name, best, ops / best / 1e6);
  };
  auto time_fq = [&](void (*k)(Fq*, uint32_t), const char* name) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(ea, 0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, o1, iters);
      hipEventRecord(eb, 0);
      hipEventSynchronize(eb);
      float ms;
      hipEventElapsedTime(&ms, ea, eb);
      if (ms < best) best = ms;
    }
    report(name, best);
  };
  time_fq(k_legacy, "legacy");
  time_fq(k_f29, "f29");
  time_fq(k_f29_sqr, "f29sqr");
  {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(ea, 0);
      hipLaunchKernelGGL(k_dpf, dim3(blocks), dim3(threads), 0, 0, o2, iters, K);
      hipEventRecord(eb, 0);
      hipEventSynchronize(eb);
      float ms;
      hipEventElapsedTime(&ms, ea, eb);
      if (ms < best) best = ms;
    }
    report("dpf52", best);
  }
  return 0;
}
