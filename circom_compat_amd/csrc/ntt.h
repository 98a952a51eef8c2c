// ntt.h -- radix-2 NTT over BN254 Fr for the CircomReduction witness map.
//
// Replaces ark-poly Radix2EvaluationDomain::{ifft_in_place, fft_in_place,
// distribute_powers_and_mul_by_const} as called from reference src/circom/qap.rs:60-61,69-73,79-81.
// Those calls are natural-order-in / natural-order-out.  Here the pair (inverse, forward) is split as
//   inverse = decimation-in-frequency  : natural in  -> bit-reversed out, 1/n and the omega_{2n}^i
//             coset twist fused into its last pass,
//   forward = decimation-in-time       : bit-reversed in -> natural out,
// so no bit-reversal pass ever touches HBM and h comes out in natural order (what the H MSM needs).
#pragma once
#include <vector>

#include "common.h"

namespace g16 {

struct NttPass {
  int lo;    // lowest index bit this pass transforms
  int b;     // number of bits (local transform size R = 2^b held in LDS)
  int logT;  // log2 of independent columns / segments per workgroup tile
};

struct NttPlan {
  int k = 0;  // log2 n
  size_t n = 0;
  int h1 = 0;        // split of the two-level twiddle tables: exponent = hi * 2^h1 + lo
  int loc_bits = 0;  // local twiddle table holds omega_{2^loc_bits}^j
  std::vector<NttPass> passes;  // in DIF order (top bits first); DIT walks it backwards
  DevBuf<Fr> tlo[2], thi[2];    // [0] omega_n^e, [1] omega_n^-e
  DevBuf<Fr> twlo, twhi;        // omega_{2n}^j, j < n; twhi carries the 1/n factor
  DevBuf<Fr> loc[2];            // omega_{2^loc_bits}^{+-j}, j < 2^(loc_bits-1)
  Fr n_inv;

  void build(int log_n);  // host: computes tables (host Fr arithmetic), uploads
};

enum NttFuse { NTT_FUSE_NONE = 0, NTT_FUSE_TWIST_SCALE = 1, NTT_FUSE_SCALE = 2 };

// In-place transforms of `batch` vectors of n elements, vector v at data + v*stride.
//   ntt_dif: natural -> bit-reversed; inverse selects omega^-1; fuse applies after the last pass.
//   ntt_dit: bit-reversed -> natural (forward omega only).
void ntt_dif(const NttPlan& plan, Fr* data, size_t stride, int batch, bool inverse, NttFuse fuse,
             hipStream_t stream);
void ntt_dit(const NttPlan& plan, Fr* data, size_t stride, int batch, bool inverse,
             hipStream_t stream);
// out[i] = in[bitrev_k(i)]  (test / debug API only; never on the proving path)
void bitrev_copy(const Fr* in, Fr* out, int k, hipStream_t stream);

// host helpers (also used by the key generator)
Fr fr_root_of_unity(int log_n);  // omega_{2^log_n} = 5^((r-1)/2^28)^(2^(28-log_n)), Montgomery form
Fr fr_pow_u64(Fr base, uint64_t e);

}  // namespace g16
