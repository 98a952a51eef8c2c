// mfma_montgomery_bound.hip -- VERDICT r5 item 9 (gated experiment), measurement only: an UPPER BOUND on the rate of
// a lazy 29-bit Montgomery product whose constant half (m x p: 81 of the 164 multiply-adds) runs on the otherwise idle
// MFMA pipe as an int8 Toeplitz product.  The kernel below executes only the VALU work such a product cannot shed:
//   A. T = a x b, 17 columns of multiply-adds, carried into 18 limbs                  (81 multiply-adds + columns)
//   B. m = T_lo x p' mod 2^261 -- the quotient has to be COMPLETE before a matrix product can use it
//      ("separated operand" Montgomery; the interleaved form needs m_k column by column)   (45 multiply-adds)
//   C. m's 9 limbs packed into the 4-bytes-per-register form of an int8 MFMA operand       (pack())
//   E. the 65 int32 byte-columns of m x p (here: stand-in values derived from m, so that the compiler cannot drop
//      them) folded back into 29-bit limbs, added to T_hi, carried
// and NOTHING of: the cross-lane transposes into / out of the MFMA operand layout (a column spans 4 lanes; ~33 + ~145
// LDS operations per product), the 20 v_mfma_i32_16x16x64_i8 per 64 products, their issue slots and latency.  If this
// bound is not >= 1.15 x field29.h's product rate, the full scheme cannot be either (the gate of item 9).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../circom_compat_amd/csrc -o mfma_montgomery_bound mfma_montgomery_bound.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "field29.h"
using namespace g16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ Fq29 bound_product(const Fq29& a, const Fq29& b) {
  constexpr int N = f29::N;
  using C = Fq29::C;
  // A. full product, carried
  int32_t t[2 * N];
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; ++k) {
    const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
#pragma unroll
    for (int i = lo; i <= hi; ++i) acc += (int64_t)a.l[i] * (int64_t)b.l[k - i];
    t[k] = (int32_t)((uint32_t)acc & f29::MASK);
    acc >>= 29;
  }
  t[2 * N - 1] = (int32_t)acc;
  // B. m = T_lo * p' mod 2^261 (p' stands in as the NINV-seeded constant vector: same instruction count)
  int32_t m[N];
  acc = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (int64_t)t[i] * (int64_t)(C::MOD.v[k - i] ^ (int32_t)C::NINV);
    m[k] = (int32_t)((uint32_t)acc & f29::MASK);
    acc >>= 29;
  }
  // C. the MFMA operand form: 9 x 29 bits -> 9 words of 4 bytes
  uint32_t w[9];
  const uint32_t* u = reinterpret_cast<const uint32_t*>(m);
  w[0] = u[0] | (u[1] << 29);
  w[1] = (u[1] >> 3) | (u[2] << 26);
  w[2] = (u[2] >> 6) | (u[3] << 23);
  w[3] = (u[3] >> 9) | (u[4] << 20);
  w[4] = (u[4] >> 12) | (u[5] << 17);
  w[5] = (u[5] >> 15) | (u[6] << 14);
  w[6] = (u[6] >> 18) | (u[7] << 11);
  w[7] = (u[7] >> 21) | (u[8] << 8);
  w[8] = u[8] >> 24;
  // E. 65 byte-columns (stand-ins: a byte-column of m x p is < 33 * 255 * 255 < 2^22) folded into limbs: column k has
  //    weight 2^(8 k); limb j takes bits [29 j, 29 j + 29)
  int64_t lim[2 * N];
#pragma unroll
  for (int j = 0; j < 2 * N; ++j) lim[j] = t[j];
#pragma unroll
  for (int k = 0; k < 65; ++k) {
    const uint32_t col = ((w[k % 9] >> (8 * (k % 4))) * 2654435761u) >> 10;  // < 2^22, depends on m
    const int bit = 8 * k, j = bit / 29, sh = bit % 29;
    lim[j] += (int64_t)col << sh;  // spills into limb j + 1 through the carry below
  }
  Fq29 r;
  int64_t c = 0;
#pragma unroll
  for (int j = 0; j < 2 * N; ++j) {
    c += lim[j];
    if (j >= N) r.l[j - N] = (int32_t)((uint32_t)c & f29::MASK);
    c >>= 29;
  }
  r.l[N - 1] += (int32_t)(c << 29);
  return r;
}

template <int ILP, bool BOUND>
__global__ void __launch_bounds__(128) k_mul(Fq* out, uint32_t iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 x[ILP];
#pragma unroll
  for (int j = 0; j < ILP; ++j) x[j] = Fq29::from_mont256(Fq::from_u32(t * 4 + 3 + j));
  const Fq29 mm = Fq29::from_mont256(Fq::from_u32(0x9e3779b9u));
  for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < ILP; ++j) x[j] = BOUND ? bound_product(x[j], mm) : x[j] * mm;
  }
  Fq29 s = x[0];
#pragma unroll
  for (int j = 1; j < ILP; ++j) s = (s + x[j]).carry();
  Fq o;
  s.pack(o.v);
  out[t] = o;
}

int main() {
  CK(hipSetDevice(0));
  const int threads = 128;
  const uint32_t iters = 512;
  Fq* o;
  CK(hipMalloc(&o, sizeof(Fq) * 8192 * threads));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int blocks : {2048, 6144, 8192}) {
    float best[2] = {1e30f, 1e30f};
    for (int rep = 0; rep < 3; ++rep) {
      for (int v = 0; v < 2; ++v) {
        CK(hipEventRecord(a, 0));
        if (v == 0) hipLaunchKernelGGL((k_mul<4, false>), dim3(blocks), dim3(threads), 0, 0, o, iters);
        else hipLaunchKernelGGL((k_mul<4, true>), dim3(blocks), dim3(threads), 0, 0, o, iters);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best[v]) best[v] = ms;
      }
    }
    const double prods = (double)blocks * threads * iters * 4;
    printf("grid %5d (4 independent products per thread): field29.h %7.1f G products/s | VALU-only part of an MFMA-assisted product "
           "%7.1f G/s = %.2fx (gate: the WHOLE scheme >= 1.15x)\n", blocks, prods / best[0] / 1e6, prods / best[1] / 1e6, best[0] / best[1]);
  }
  return 0;
}
