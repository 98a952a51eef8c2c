// batch_affine.hip -- VERDICT r5 item 1(b), measurement only: the ARITHMETIC of a batched-affine bucket
// accumulation (affine + affine with Montgomery's shared inversion: 5M + 1S + the lane's share of one
// inversion per addition) against the mixed XYZZ addition the product kernels run (8M + 2S), both on the
// lazy 29-bit limbs of field29.h / ec29.h, both from a cache-resident point table -- i.e. the ceiling of the
// scheme BEFORE its memory traffic (two passes over the points + a prefix product per addition through HBM;
// priced in DESIGN.md section 9 from these rates).
//
// Layout of the batched kernel = what a level kernel of a pairwise bucket tree would run: every lane owns K
// independent additions (interleaved over the wave), pass 1 forms the running product of its denominators
// (prefixes to a global scratch, [i][lane]: coalesced), ONE Fermat inversion per lane (uniform control flow:
// no cross-lane step), pass 2 walks back, reloads the points, emits x3 | y3 (raw limbs, coalesced).
// A check kernel compares K additions done this way with XYZZ29::madd on the same pairs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../circom_compat_amd/csrc -o batch_affine batch_affine.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ec29.h"
using namespace g16;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int TBL = 4096;  // points of the table (256 KiB for G1, 512 KiB for G2: L2 resident)

template <class F>
__global__ void k_make_table(Affine<F> gen, Affine<F>* tbl) {
  using LF = typename Lazy<F>::type;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= TBL) return;
  const Aff29<LF> g = affine_from_mont256<F>(gen);
  XYZZ29<LF> acc = XYZZ29<LF>::infinity(), run = XYZZ29<LF>::from_affine(g);
  for (uint32_t k = t + 1; k; k >>= 1) {  // (t + 1) * gen
    if (k & 1u) acc.add(run);
    run.dbl_in_place();
  }
  tbl[t] = store_packed_affine<F>(acc.to_affine());
}

__device__ __forceinline__ uint32_t pick(uint32_t lane, uint32_t i, uint32_t salt) {
  uint32_t x = (lane * 2654435761u) ^ (i * 40503u + salt);
  x ^= x >> 15;
  x *= 2246822519u;
  x ^= x >> 13;
  return x & (TBL - 1);
}

// the product kernels' addition: acc += table[random], optimistic select form (msm_curve.inc.h)
template <class F>
__global__ void __launch_bounds__(128) k_xyzz(const Affine<F>* __restrict__ tbl, uint32_t iters, uint32_t* out) {
  using LF = typename Lazy<F>::type;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ29<LF> acc = XYZZ29<LF>::infinity();
  uint32_t bad = 0;
  for (uint32_t i = 0; i < iters; ++i) {
    const Aff29<LF> p = load_packed_affine<F>(tbl[pick(t, i, 1u)]);
    bool special;
    const XYZZ29<LF> r = XYZZ29<LF>::madd_select(acc, p, &special);
    if (special) ++bad;
    else acc = r;
  }
  out[t] = (uint32_t)acc.x.carry().limbs_all_zero() + bad;
}

template <class LF>
__device__ __forceinline__ void store_raw(int32_t* base, size_t stride, const LF& v) {
  const int32_t* l = reinterpret_cast<const int32_t*>(&v);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(LF) / 4); ++k) base[(size_t)k * stride] = l[k];
}
template <class LF>
__device__ __forceinline__ LF load_raw(const int32_t* base, size_t stride) {
  LF v;
  int32_t* l = reinterpret_cast<int32_t*>(&v);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(LF) / 4); ++k) l[k] = base[(size_t)k * stride];
  return v;
}

// K additions per lane and round; prefix: [K][limbs][nthreads] int32; outp: [K][2 * limbs][nthreads]
template <class F>
__global__ void __launch_bounds__(128) k_batch(const Affine<F>* __restrict__ tbl, uint32_t K, uint32_t rounds,
                                               int32_t* prefix, int32_t* outp, uint32_t* out) {
  using LF = typename Lazy<F>::type;
  constexpr int NL = (int)(sizeof(LF) / 4);
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthr = (size_t)gridDim.x * blockDim.x;
  uint32_t chk = 0;
  for (uint32_t r = 0; r < rounds; ++r) {
    // pass 1: running product of the denominators
    LF p = LF::one();
    for (uint32_t i = 0; i < K; ++i) {
      const Affine<F>& A = tbl[pick(t, i, r * 2u)];
      const Affine<F>& B = tbl[pick(t, i, r * 2u + 1u)];
      LF d = Lazy<F>::load_packed(B.x) - Lazy<F>::load_packed(A.x);
      if (d.maybe_zero_mod_p()) d = LF::one();  // equal x (doubling / cancellation): resolved in pass 2
      p = p * d;
      store_raw<LF>(prefix + ((size_t)i * NL) * nthr + t, nthr, p);
    }
    LF J = f29_inv(p);  // one inversion per lane, uniform over the wave
    // pass 2: back substitution, lambda, x3, y3
    for (uint32_t i = K; i-- > 0;) {
      const Aff29<LF> A = load_packed_affine<F>(tbl[pick(t, i, r * 2u)]);
      const Aff29<LF> B = load_packed_affine<F>(tbl[pick(t, i, r * 2u + 1u)]);
      LF d = B.x - A.x;
      const bool same_x = d.maybe_zero_mod_p();
      if (same_x) d = LF::one();
      const LF pe = i ? load_raw<LF>(prefix + ((size_t)(i - 1) * NL) * nthr + t, nthr) : LF::one();
      const LF inv = J * pe;
      J = J * d;
      const LF lam = (B.y - A.y) * inv;
      LF x3 = (lam.sqr() - A.x - B.x).carry();
      LF y3 = (lam * (A.x - x3) - A.y).carry();
      if (same_x) {  // (the level kernel would take the exact path here; the table never pairs equal points)
        x3 = A.x;
        y3 = A.y;
      }
      store_raw<LF>(outp + ((size_t)i * 2 * NL) * nthr + t, nthr, x3);
      store_raw<LF>(outp + ((size_t)(i * 2 + 1) * NL) * nthr + t, nthr, y3);
      chk += (uint32_t)reinterpret_cast<const int32_t*>(&x3)[0];
    }
  }
  out[t] = chk;
}

// one lane: K additions by the batched formulas == the same additions by XYZZ29 (cross-multiplied)
template <class F>
__global__ void k_check(const Affine<F>* __restrict__ tbl, uint32_t K, uint32_t* n_bad) {
  using LF = typename Lazy<F>::type;
  if (threadIdx.x || blockIdx.x) return;
  LF pre[64];
  LF p = LF::one();
  for (uint32_t i = 0; i < K; ++i) {
    LF d = Lazy<F>::load_packed(tbl[pick(7, i, 1u)].x) - Lazy<F>::load_packed(tbl[pick(7, i, 0u)].x);
    if (d.maybe_zero_mod_p()) d = LF::one();
    p = p * d;
    pre[i] = p;
  }
  LF J = f29_inv(p);
  uint32_t bad = 0;
  for (uint32_t i = K; i-- > 0;) {
    const Aff29<LF> A = load_packed_affine<F>(tbl[pick(7, i, 0u)]);
    const Aff29<LF> B = load_packed_affine<F>(tbl[pick(7, i, 1u)]);
    LF d = B.x - A.x;
    if (d.maybe_zero_mod_p()) { J = J * LF::one(); continue; }
    const LF inv = J * (i ? pre[i - 1] : LF::one());
    J = J * d;
    const LF lam = (B.y - A.y) * inv;
    const LF x3 = (lam.sqr() - A.x - B.x).carry();
    const LF y3 = (lam * (A.x - x3) - A.y).carry();
    XYZZ29<LF> s = XYZZ29<LF>::from_affine(A);
    s.madd(B);
    if (!((x3 * s.zz) - s.x).is_zero_mod_p() || !((y3 * s.zzz) - s.y).is_zero_mod_p()) ++bad;
  }
  *n_bad = bad;
}

template <class F>
int run(const char* name, Affine<F> gen, int mac_xyzz, int mac_ba) {
  Affine<F>* tbl;
  CK(hipMalloc(&tbl, sizeof(Affine<F>) * TBL));
  hipLaunchKernelGGL(k_make_table<F>, dim3(TBL / 64), dim3(64), 0, 0, gen, tbl);
  CK(hipDeviceSynchronize());
  uint32_t* nbad;
  CK(hipMalloc(&nbad, 4));
  hipLaunchKernelGGL(k_check<F>, dim3(1), dim3(64), 0, 0, (const Affine<F>*)tbl, 64u, nbad);
  uint32_t hb = 99;
  CK(hipMemcpy(&hb, nbad, 4, hipMemcpyDeviceToHost));
  printf("%s: check of 64 batched additions against XYZZ29::madd: %u mismatches\n", name, hb);
  using LF = typename Lazy<F>::type;
  constexpr int NL = (int)(sizeof(LF) / 4);
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  uint32_t* out;
  const int threads = 128;
  for (int blocks : {2048, 3072, 6144}) {
    CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    const uint32_t iters = sizeof(F) == sizeof(Fq) ? 512 : 192;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(k_xyzz<F>, dim3(blocks), dim3(threads), 0, 0, (const Affine<F>*)tbl, iters, out);
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    const double n = (double)blocks * threads * iters;
    const double xyzz_rate = n / best / 1e6;  // G additions/s
    printf("%s xyzz   grid %5d: %8.3f ms  %6.2f G additions/s  (%d multiply-adds each: %.1f T/s)\n", name, blocks, best,
           n / best / 1e6, mac_xyzz, n * mac_xyzz / best / 1e9);
    for (uint32_t K : {64u, 128u, 256u, 512u}) {
      const size_t nthr = (size_t)blocks * threads;
      int32_t *prefix, *outp;
      CK(hipMalloc(&prefix, nthr * K * NL * 4));
      CK(hipMalloc(&outp, nthr * K * 2 * NL * 4));
      const uint32_t budget = sizeof(F) == sizeof(Fq) ? 1024u : 512u;  // additions per lane and launch
      const uint32_t rounds = budget / K ? budget / K : 1u;
      best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_batch<F>, dim3(blocks), dim3(threads), 0, 0, (const Affine<F>*)tbl, K, rounds, prefix, outp, out);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      const double nb = (double)nthr * K * rounds;
      printf("%s batch  grid %5d K %3u: %8.3f ms  %6.2f G additions/s  (%.2fx xyzz; ~%d multiply-adds each + inversion / K)\n",
             name, blocks, K, best, nb / best / 1e6, (nb / best / 1e6) / xyzz_rate, mac_ba);
      CK(hipFree(prefix));
      CK(hipFree(outp));
    }
    CK(hipFree(out));
  }
  CK(hipFree(tbl));
  return 0;
}

int main() {
  CK(hipSetDevice(0));
  G1Affine g1{Fq::one(), Fq::one() + Fq::one()};
  static const uint32_t X0[8] = {0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu};
  static const uint32_t X1[8] = {0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u};
  static const uint32_t Y0[8] = {0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u};
  static const uint32_t Y1[8] = {0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u};
  auto fqc = [](const uint32_t* l) {
    U256 u;
    for (int i = 0; i < 8; ++i) u.v[i] = l[i];
    return Fq::from_canonical(u);
  };
  G2Affine g2{Fq2{fqc(X0), fqc(X1)}, Fq2{fqc(Y0), fqc(Y1)}};
  if (run<Fq>("G1", g1, 1557, 936)) return 1;
  if (run<Fq2>("G2", g2, 4878, 2835)) return 1;
  return 0;
}
