// msm_table.h -- fixed-base tables for SMALL keys: VariableBaseMSM::msm_bigint (ark-ec, reached from
// create_proof_with_assignment; call sites reference benches/groth16.rs:52-60, src/zkey.rs:903-911) as
// n * 32 independent table lookups + one tree sum, instead of sort -> bucket accumulate -> bucket
// reduce.
//
// Why (DESIGN.md "Small proofs"): below ~2^15 points a proof is not work, it is latency -- ~75
// dependent launches with single-wave tails (three bucket reductions of ~50 serial EC additions each,
// two 128-step GLV variable-base products, the affine conversions): 2.2-2.3 ms from 10^3 to 10^5
// constraints, the reference's own bench circuit (benches/groth16.rs:106, 10^4 constraints) among
// them.  With 288 GB of HBM a small key can spend capacity instead: for every query point P_i and
// every 8-bit window j the table holds k * 2^(8 j) * P_i, k = 1..128 (signed digits), in the packed
// affine form of the point planes -- 256 KiB per G1 point, 512 KiB per G2 point; 2.6 / 5.2 GB per query
// at 10^4 points.  An MSM is then:
//   k_tbl_msm    thread (i, g): the scalar's canonical bytes ARE the digits; 4 windows per thread, 4
//                mixed additions of looked-up points, then a block-level tree (LDS) -> one partial
//   k_tbl_final  one block per MSM: tree over the partials -> the sum
// -- ~21 dependent additions per MSM whatever n, no sort, no buckets, no running sum.  The two
// variable-base products of the finalisation disappear as well: s * MSM_A(w) = MSM_A(s w) and
// r * MSM_B1(w) = MSM_B1(r w) are two more table MSMs over the same tables (one Fr product per
// scalar), which turns g_c into fixed-base sums only (finalize.h: fin_fixed_dist / fin_final_dist,
// the tail the sharded provers already use).
#pragma once
#include "common.h"
#include "ec29.h"
#include "finalize.h"

namespace g16 {

constexpr int TBL_C = 8;                 // window bits = one byte of the canonical scalar
constexpr int TBL_W = 32;                // windows (the top byte of a scalar < r is < 0x31: no carry out)
constexpr int TBL_E = 1 << (TBL_C - 1);  // entries per window: k = 1..128
constexpr int TBL_WPT = 4;               // windows per thread
constexpr int TBL_TPP = TBL_W / TBL_WPT; // threads per point
constexpr int TBL_BLOCK = 256;     // G1: 256 x 144 B of LDS for the block tree
constexpr int TBL_BLOCK_G2 = 128;  // G2: 128 x 288 B
constexpr uint32_t TBL_MAX_POINTS = 1u << 14;  // per query (auto rule; 4 GiB G1 / 8 GiB G2 tables)

inline size_t tbl_bytes(size_t g1_points, size_t g2_points) {
  return (g1_points * 64 + g2_points * 128) * TBL_W * TBL_E;
}

// one MSM of a launch: table, scalars, where the partials and the sum go
struct TblJob {
  const void* table;    // Affine<F>[count][TBL_W][TBL_E], packed internal form
  const void* scalars;  // Fr[count] (Montgomery) or U256[count] (canonical)
  uint32_t count;
  int canonical;  // scalars are canonical integers (h_canon)
  int mul;        // 0: k = scalar; 1: k = r * scalar; 2: k = s * scalar   (rs_dev = r | s, Montgomery)
  void* partial;  // XYZZ29[blocks]
  void* sum;      // XYZZ29*
};
constexpr int TBL_MAX_JOBS = 5;
struct TblJobs {
  TblJob j[TBL_MAX_JOBS];
  int n;
};

struct TableSet {
  bool active = false;
  uint32_t len_w = 0, l_cnt = 0, l_idx_min = 0, len_h = 0;
  DevBuf<G1Affine> oA, oB1, oL, oH;  // owned tables
  DevBuf<G2Affine> oB2;
  const G1Affine *tA = nullptr, *tB1 = nullptr, *tL = nullptr, *tH = nullptr;  // views (a sibling borrows)
  const G2Affine* tB2 = nullptr;
  DevBuf<G1XYZZ29> part1;  // [5 jobs][blocks] witness-scalar G1 MSMs (A, B1, L, sA, rB1)
  DevBuf<G1XYZZ29> partH;
  DevBuf<G2XYZZ29> part2;
  uint32_t blocks_w = 0, blocks_w2 = 0, blocks_h = 0;

  // host pointers to storage-form points (possibly unaligned: zero-copy zkey views)
  void build(const uint8_t* a, const uint8_t* b1, const uint8_t* b2, const uint8_t* l, const uint8_t* h,
             uint32_t len_w, uint32_t l_idx_min, uint32_t l_cnt, uint32_t len_h, hipStream_t stream);
  void borrow(const TableSet& lender);
  void release() {  // a build that failed half way (automatic mode): no tables, the bucket path proves
    oA.release(); oB1.release(); oL.release(); oH.release(); oB2.release();
    part1.release(); partH.release(); part2.release();
    tA = tB1 = tL = tH = nullptr;
    tB2 = nullptr;
    active = false;
  }
  size_t table_bytes() const { return tbl_bytes((size_t)2 * len_w + l_cnt + len_h, len_w); }

  // A, B1, L, s*A, r*B1 over the witness (w1 = w_dev + 1: entry i pairs with w[1 + i])
  void run_g1_witness(const Fr* w1, const Fr* rs_dev, ProofSums* sums, hipStream_t stream);
  void run_g2_witness(const Fr* w1, ProofSums* sums, hipStream_t stream);
  void run_h(const U256* h_canon, ProofSums* sums, hipStream_t stream);
};

}  // namespace g16
