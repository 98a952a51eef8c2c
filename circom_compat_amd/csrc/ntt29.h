// ntt29.h -- the witness map's radix-2 NTT over BN254 Fr on the lazy 9 x 29-bit limbs of field29.h.
//
// Same role and pass structure as ntt.h (which stays for the key generator and as a second
// implementation the tests cross-check): replaces ark-poly Radix2EvaluationDomain::{ifft_in_place,
// fft_in_place, distribute_powers_and_mul_by_const} as called from reference
// src/circom/qap.rs:60-61,69-73,79-81.  inverse = DIF (natural in, bit-reversed out, 1/n and the
// omega_2n^i coset twist fused into the last pass), forward = DIT (bit-reversed in, natural out).
//
// Data layout in HBM: a vector of n field elements is stored as 9 planes of n int32 ("limb k of
// every element"), so every global access of a wave is a run of consecutive 4-byte words; values
// are lazy (field29.h): |value| < 16 r, limbs 0..7 within [-8, 2^29 + 8).  A batch of vectors is
// [vector][limb][n].
//
// Value bookkeeping (asserted by F29_CHECK in the emulator build):
//   DIF: the sum output x0 + x1 doubles per stage; every 6th stage of a pass multiplies the sum
//        by one() (passes of <= 6 stages need none: 2 r * 2^6 = 128 r) and every pass ends with a
//        multiplication of ALL elements (inter-pass twiddle, twist or one()), so |value| <= 128 r at
//        every product (the contract allows 169 r against a canonical twiddle).
//   DIT: x0 +- w x1 grows by at most 2 r per stage; every strided pass starts with a
//        multiplication of all elements (inter-pass twiddle), so |value| < 24 r throughout.
#pragma once
#include "field29.h"
#include "ntt.h"

namespace g16 {

constexpr int NTT29_LIMBS = f29::N;
constexpr int NTT29_FULL_TABLE_MAX_LOG = 24;   // 3 n x 32 B of tables: 1.5 GiB at 2^24

struct Ntt29Plan {
  NttPlan base;                  // pass schedule + host-computed tables in the storage form
  DevBuf<Fr> tlo[2], thi[2];     // the same tables in the PACKED INTERNAL form (canonical x * 2^261)
  DevBuf<Fr> twlo, twhi, loc[2];
  // Round 5: SINGLE-LEVEL tables (one product per twiddle instead of lo x hi, then the element) for
  // log n <= NTT29_FULL_TABLE_MAX_LOG.  ptab[d][i]: the inter-pass twiddles of strided pass i in the
  // order the pass touches them -- entry (kappa << lo) | c = omega_n^(+-(c kappa) << (k - hi)), 2^hi
  // entries: n for the top pass (32 B per element and pass of extra HBM reads, contiguous in c),
  // 2^16-ish (L2 resident) for the ones below.  twfull[i] = omega_2n^bitrev(i) / n: the coset twist of
  // the CircomReduction map at the position the last DIF pass stores to.  2 n + n entries of 32 B.
  DevBuf<Fr> ptab[2][4];
  DevBuf<Fr> twfull;
  bool full_tables = false;
  Fr n_inv_packed;               // 1/n, packed internal
  void build(int log_n, hipStream_t stream, bool want_full_tables = true);
  // two-level table of scale * base^j, j < n, in the packed internal form (lo: 2^h1 entries of
  // base^l, hi: scale * base^(h 2^h1)); the plan's own twlo / twhi are make_twist(omega_2n, 1/n)
  void make_twist(Fr base, Fr scale, DevBuf<Fr>& lo, DevBuf<Fr>& hi) const;
  size_t n() const { return base.n; }
};

// In-place transforms of `batch` vectors; vector v occupies planes data + v * vec_stride,
// plane k at + k * n (vec_stride >= 9 n, in int32 units).
// twlo / twhi (optional): twist tables to use with NTT_FUSE_TWIST_SCALE instead of the plan's
// omega_2n tables (see Ntt29Plan::make_twist)
void ntt29_dif(const Ntt29Plan& plan, int32_t* data, size_t vec_stride, int batch, bool inverse,
               NttFuse fuse, hipStream_t stream, const Fr* twlo = nullptr, const Fr* twhi = nullptr);
void ntt29_dit(const Ntt29Plan& plan, int32_t* data, size_t vec_stride, int batch, hipStream_t stream);

// storage form (Montgomery R = 2^256, 32 bytes per element) <-> planes
void ntt29_to_planes(const Fr* in, int32_t* planes, size_t n, hipStream_t stream);
void ntt29_from_planes(const int32_t* planes, Fr* out, size_t n, hipStream_t stream);
// out[i] = in[bitrev_k(i)] on planes (tests / debug only)
void ntt29_bitrev_planes(const int32_t* in, int32_t* out, int k, hipStream_t stream);

// plane access helpers shared with witness_map.hip
__device__ __forceinline__ Fr29 load_planes(const int32_t* __restrict__ base, size_t n, size_t i) {
  Fr29 x;
#pragma unroll
  for (int k = 0; k < NTT29_LIMBS; ++k) x.l[k] = base[(size_t)k * n + i];
  return x;
}
__device__ __forceinline__ void store_planes(int32_t* __restrict__ base, size_t n, size_t i,
                                             const Fr29& x) {
#pragma unroll
  for (int k = 0; k < NTT29_LIMBS; ++k) base[(size_t)k * n + i] = x.l[k];
}

}  // namespace g16
