// wm_dist.h -- CircomReduction::witness_map_from_matrices (reference src/circom/qap.rs:23-88)
// distributed over the G ranks of one node: the six length-n NTTs become "four-step" transforms
// whose two transposes are all-to-all exchanges over xGMI (RCCL, driven by the host framework).
//
// n = n1 * n2.  Rank g owns
//   * the constraint rows i = i1 * n2 + i2 with i2 in [g c2, (g+1) c2), c2 = n2 / G: it evaluates
//     a_i, b_i, c_i for those rows only (rows are independent: qap.rs:37-58);
//   * after the second exchange, h_e for exactly the same index set e = m2 + n2 * m1, m2 in its
//     i2-range -- so its shard of the H query is that strided point set, fixed at ctx creation.
//
// Per vector (a, b, c are batched):
//   phase 1  local inverse DIF over i1 (n1 points, per i2), twiddle omega_n^(-i2 j1), pack by
//            destination          -> exchange 1: rank d receives positions p(j1) in [d r1, (d+1) r1)
//   phase 2  local inverse DIF over i2 (n2 points, per j1), x 1/n * omega_2n^(j1 + n1 j2) (the coset
//            twist of qap.rs:63-70), local forward DIT over j2 (in place: the DIF's bit-reversed
//            output is the DIT's input), twiddle omega_n^(j1 m2), pack -> exchange 2
//   phase 3  local forward DIT over j1 (n1 points, per m2), h = a b - c (qap.rs:75,83-85).
// Exchange payload: 3 n / G elements of 9 x int32 per rank and exchange.
#pragma once
#include "witness_map.h"

namespace g16 {

struct WmDist {
  int rank = 0, world = 1;
  uint32_t m = 0, num_inputs = 0, n = 0;
  int k = 0, k1 = 0, k2 = 0;
  uint32_t n1 = 0, n2 = 0, c2 = 0, r1 = 0;  // c2 = n2 / G columns, r1 = n1 / G rows per rank
  Ntt29Plan planN, plan1, plan2;            // planN: only its omega_n / omega_2n tables are used
  CsrStore dA, dB;                          // cooked (spmv.h)
  SpmvPlan spmv;                            // classes of this rank's rows
  DevBuf<int32_t> bufA;  // [3][c2][9][n1]
  DevBuf<int32_t> bufB;  // [3][r1][9][n2]

  void init(const CsrHost& A, const CsrHost& B, uint32_t m, uint32_t num_inputs, int rank, int world);
  size_t exchange_ints() const { return (size_t)3 * c2 * n1 * NTT29_LIMBS; }  // per rank, per exchange
  // global evaluation index e of local h index t (the H-query shard of this rank)
  uint32_t global_index(uint32_t t) const { return (uint32_t)rank * c2 + t / n1 + n2 * (t % n1); }
  void phase1(const Fr* w_dev, int32_t* send, hipStream_t stream);
  void phase2(const int32_t* recv, int32_t* send, hipStream_t stream);
  void phase3(const int32_t* recv, U256* h_canon, hipStream_t stream);  // n / G canonical scalars
};

}  // namespace g16
