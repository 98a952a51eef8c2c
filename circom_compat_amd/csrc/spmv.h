// spmv.h -- row-length-adaptive sparse rows x vector over Fr.
//
// Replaces evaluate_constraint (ark-groth16 r1cs_to_qap, called at reference src/circom/qap.rs:37-44)
// for rows of ANY length: circom circuits hold one- and two-term rows (wire copies, products) next
// to linear sums of tens (Poseidon's folded MDS layers), hundreds (Num2Bits) or 10^5 terms, and the
// transposed matrices of the key generator hold the constant wire's column (one term per row).
// One thread per row runs a wave at the length of its longest row and reads every coefficient
// with a 36-byte stride; here rows are binned once per matrix set (SpmvPlan::build, host side):
//
//   short   every matrix holds <= SPMV_SHORT terms: one thread per row (the caller's own kernel,
//           spmv_row_thread)
//   medium  <= SPMV_TASK_TERMS terms per matrix: SPMV_G lanes per row (k_spmv_medium) -- lane l takes
//           terms s + l, s + l + G, ... so a group reads consecutive words; the lane sums are
//           tree-added with wave shuffles
//   huge    anything longer: cut into tasks of SPMV_TASK_TERMS terms (k_spmv_tasks, one group per
//           task, partial sums to HBM), one thread per row adds the partials (k_spmv_huge)
//
// Arithmetic: lazy 9 x 29-bit limbs (field29.h).  At plan time the coefficients are "cooked" in
// place (k_spmv_cook): coeff == 1 sets bit 31 of the column index and the coefficient is never read
// again (the reference skips the multiplication for it, qap.rs via evaluate_constraint; here it
// also skips the 32-byte load); any other coefficient becomes coeff * 2^266 mod r, canonical, in
// the 8 words of the Fr, so that  mul(cooked, w_mont256) = coeff * w * 2^261  is the product in the
// internal form with no conversion of the gathered witness value; a unit coefficient is the constant 2^266 mod r.
// Terms go through the reduction TWO at a time (mul2: (k1 w1 + k2 w2) / 2^261, 243 multiply-adds for two terms).
// The result of a row is the field element sum coeff * w[idx] -- identical to the reference's whatever the
// summation order.
#pragma once
#include <functional>

#include "common.h"
#include "field29.h"

namespace g16 {

constexpr uint32_t SPMV_ONE = 0x80000000u;  // col bit 31: coefficient == 1
constexpr uint32_t SPMV_SHORT = 4;
constexpr int SPMV_G = 4;
// terms a lane sums lazily: 8 pair products in (-r, 2r) = |value| < 16 r; a pair's operands -- cooked coefficient
// < r, witness word < 2^256 < 5.3 r even when it is not canonical -- keep |k1 w1| + |k2 w2| < 11 r^2 (field29.h mul2
// contract: 169).  SPMV_G = 4 lanes (measured on the 2^20 Poseidon chain, 9.3 / 17.7 terms per A / B row: 8 lanes
// 0.65 ms, most of it idle lanes; 4 lanes 0.58; rows sorted by trip count 0.51; pairs through mul2: see DESIGN.md)
constexpr uint32_t SPMV_LANE_TERMS = 16;
constexpr uint32_t SPMV_TASK_TERMS = SPMV_G * SPMV_LANE_TERMS;  // 64

struct SpmvDev {
  const uint32_t* rowptr;
  const uint32_t* col;  // bit 31: unit coefficient (after cooking)
  const Fr* val;        // cooked
};
struct SpmvTask {
  uint32_t s;      // first term
  uint32_t len_q;  // bits 0..15: terms (<= SPMV_TASK_TERMS); bits 16..: matrix
};

// Row classes of `nmat` matrices over the same rows (host side, once per key).
struct SpmvPlan {
  int nmat = 0;
  uint32_t rows = 0, n_med = 0, n_huge = 0, n_tasks = 0;
  DevBuf<uint32_t> med_rows, huge_rows;
  DevBuf<uint32_t> task_off;  // [n_huge * nmat + 1]: tasks of (huge row k, matrix q) = [task_off[k nmat + q], next)
  DevBuf<SpmvTask> tasks;
  DevBuf<int32_t> partial;    // [n_tasks][9]
  // rowptr_host[q]: host row pointers of matrix q ([rows + 1]); keep (optional): rows this device
  // evaluates (a rank of the distributed witness map owns a strided subset)
  void build(const uint32_t* const* rowptr_host, int nmat, uint32_t rows,
             const std::function<bool(uint32_t)>& keep = nullptr);
  size_t device_bytes() const {
    return med_rows.bytes() + huge_rows.bytes() + task_off.bytes() + tasks.bytes() + partial.bytes();
  }
};

// cooks col / val of one uploaded matrix in place (see the header comment)
void spmv_cook(uint32_t* col_dev, Fr* val_dev, size_t nnz, hipStream_t stream);

// ---- device side --------------------------------------------------------------------------------
#if defined(__HIPCC__) || defined(G16_EMU)

// One term as a pair of product operands: the cooked coefficient (c * 2^266; a unit coefficient is the constant
// 2^266 mod r and costs no load) and the gathered witness word as a raw Montgomery-256 integer.
struct SpmvTerm {
  Fr29 k, w;
};
__device__ __forceinline__ SpmvTerm spmv_term(const Fr* __restrict__ val, const Fr* __restrict__ x, uint32_t j,
                                              uint32_t cj, bool live) {
  SpmvTerm t{Fr29::zero(), Fr29::zero()};
  if (live) {
    t.w = Fr29::unpack(x[cj & ~SPMV_ONE].v);
    if (cj & SPMV_ONE) t.k = Fr29::from_limbs(Fr29::C::C266);
    else t.k = Fr29::unpack(val[j].v);
  }
  return t;
}

// terms s + lane, s + lane + stride, ... < e (at most SPMV_LANE_TERMS of them), TWO per reduction:
// (k1 w1 + k2 w2) / 2^261 is one mul2 (243 multiply-adds against 2 x 164 + a second set of column carries); a
// missing second term is a pair of zeros.  Returns the lazy sum: carried limbs, |value| < 2 r per pair, i.e.
// < SPMV_LANE_TERMS r.
__device__ __forceinline__ Fr29 spmv_lane_terms(const uint32_t* __restrict__ col, const Fr* __restrict__ val,
                                                const Fr* __restrict__ x, uint32_t s, uint32_t e, uint32_t lane,
                                                uint32_t stride) {
  Fr29 acc = Fr29::zero();
  uint32_t j = s + lane;
  // the next pair's column indices are in flight while this pair's witness words and coefficients arrive
  uint32_t c1 = j < e ? col[j] : 0u, c2 = j + stride < e ? col[j + stride] : 0u;
  while (j < e) {
    const uint32_t j2 = j + stride, jn = j2 + stride, jn2 = jn + stride;
    const uint32_t n1 = jn < e ? col[jn] : 0u, n2 = jn2 < e ? col[jn2] : 0u;
    const SpmvTerm a = spmv_term(val, x, j, c1, true);
    const SpmvTerm b = spmv_term(val, x, j2, c2, j2 < e);
    acc = (acc + Fr29::mul2(a.k, a.w, b.k, b.w)).carry();
    j = jn;
    c1 = n1;
    c2 = n2;
  }
  return acc;
}
// one thread, one short row (<= SPMV_SHORT = 4 terms): carried limbs, |value| < 4 r
__device__ __forceinline__ Fr29 spmv_row_thread(const SpmvDev& M, const Fr* __restrict__ x, uint32_t i) {
  return spmv_lane_terms(M.col, M.val, x, M.rowptr[i], M.rowptr[i + 1], 0u, 1u);
}
__device__ __forceinline__ bool spmv_row_is_short(const SpmvDev& M, uint32_t i) {
  return M.rowptr[i + 1] - M.rowptr[i] <= SPMV_SHORT;
}

__device__ __forceinline__ Fr29 spmv_shfl_xor(const Fr29& a, int lane, int off) {
  Fr29 r;
#pragma unroll
  for (int k = 0; k < f29::N; ++k) r.l[k] = (int32_t)__shfl((uint32_t)a.l[k], lane ^ off);
  return r;
}
// sum over the SPMV_G lanes of a group (every lane of the wave calls this): limbs carried, |value| < G x the lanes'
__device__ __forceinline__ Fr29 spmv_group_sum(Fr29 t) {
  const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
  for (int off = SPMV_G / 2; off > 0; off >>= 1) t = (t + spmv_shfl_xor(t, lane, off)).carry();
  return t;
}
// terms [s, e), e - s <= SPMV_TASK_TERMS, by the lanes of one group: carried limbs, |value| < G SPMV_LANE_TERMS r = 64 r
__device__ __forceinline__ Fr29 spmv_group_terms(const SpmvDev& M, const Fr* __restrict__ x, uint32_t s,
                                                 uint32_t e) {
  const uint32_t gl = threadIdx.x & (uint32_t)(SPMV_G - 1);
  return spmv_group_sum(spmv_lane_terms(M.col, M.val, x, s, e, gl, (uint32_t)SPMV_G));
}

// Out: struct with  __device__ void put(uint32_t row, const Fr29* v) const  -- v[q] = row of matrix q,
// carried limbs (0..7 within [0, 2^29 + 2)), |value| < 4 r
template <int NM>
struct SpmvMats {
  SpmvDev m[NM];
};

// medium rows: one group of SPMV_G lanes per row
template <int NM, class Out>
__global__ void __launch_bounds__(256) k_spmv_medium(SpmvMats<NM> M, const Fr* __restrict__ x,
                                                     const uint32_t* __restrict__ med_rows, uint32_t n_med,
                                                     Out out) {
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / SPMV_G;
  const bool live = g < n_med;  // dead groups still take part in the shuffles
  const uint32_t i = live ? med_rows[g] : 0u;
  Fr29 v[NM];
  auto row_of = [&](const SpmvDev& m) {
    const uint32_t s = live ? m.rowptr[i] : 0u, e = live ? m.rowptr[i + 1] : 0u;
    return spmv_group_terms(m, x, s, e) * Fr29::one();  // 64 r -> product class: |a b| < 4 r^2 for put()
  };
  v[0] = row_of(M.m[0]);  // written out: the unroller refuses a loop around the wave shuffles
  if constexpr (NM > 1) v[1] = row_of(M.m[1]);
  if constexpr (NM > 2) v[2] = row_of(M.m[2]);
  static_assert(NM <= 3, "k_spmv_medium: at most three matrices");
  if (live && (threadIdx.x & (SPMV_G - 1)) == 0) out.put(i, v);
}

// huge rows, stage 1: one group per task of <= SPMV_TASK_TERMS terms
template <int NM>
__global__ void __launch_bounds__(256) k_spmv_tasks(SpmvMats<NM> M, const Fr* __restrict__ x,
                                                    const SpmvTask* __restrict__ tasks, uint32_t n_tasks,
                                                    int32_t* __restrict__ partial) {
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / SPMV_G;
  const bool live = g < n_tasks;
  const SpmvTask t = live ? tasks[g] : SpmvTask{0u, 0u};
  const uint32_t q = t.len_q >> 16, len = t.len_q & 0xffffu;
  SpmvDev mq = M.m[0];  // the matrix is selected by compares: M stays in scalar registers
#pragma unroll
  for (int k = 1; k < NM; ++k)
    if ((uint32_t)k == q) mq = M.m[k];
  const Fr29 v = spmv_group_terms(mq, x, t.s, t.s + len) * Fr29::one();  // 64 r -> product class (-r, 2 r)
  if (live && (threadIdx.x & (SPMV_G - 1)) == 0) {
#pragma unroll
    for (int k = 0; k < f29::N; ++k) partial[(size_t)g * f29::N + k] = v.l[k];
  }
}

// huge rows, stage 2: one thread per row adds the partials of its tasks
template <int NM, class Out>
__global__ void __launch_bounds__(64) k_spmv_huge(const uint32_t* __restrict__ huge_rows,
                                                  const uint32_t* __restrict__ task_off, uint32_t n_huge,
                                                  const int32_t* __restrict__ partial, Out out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_huge) return;
  Fr29 v[NM];
#pragma unroll
  for (int q = 0; q < NM; ++q) {
    Fr29 acc = Fr29::zero();
    uint32_t pending = 0;
    for (uint32_t t = task_off[k * NM + q]; t < task_off[k * NM + q + 1]; ++t) {
      Fr29 p;
#pragma unroll
      for (int l = 0; l < f29::N; ++l) p.l[l] = partial[(size_t)t * f29::N + l];
      acc = (acc + p).carry();  // a partial is in (-r, 2 r)
      if (++pending == 64) {    // 2 r + 64 * 2 r = 130 r: back to the product class
        acc = acc * Fr29::one();
        pending = 0;
      }
    }
    v[q] = acc * Fr29::one();
  }
  out.put(huge_rows[k], v);
}

// launches the medium and huge stages of a plan (the caller's own kernel covers the short rows and
// whatever lies outside the matrices)
template <int NM, class Out>
void spmv_run_long(const SpmvPlan& P, const SpmvMats<NM>& M, const Fr* x, Out out, hipStream_t stream) {
  if (P.n_med)
    G16_LAUNCH((k_spmv_medium<NM, Out>), ceil_div((uint64_t)P.n_med * SPMV_G, 256), 256, 0, stream, M, x,
               (const uint32_t*)P.med_rows.p, P.n_med, out);
  if (P.n_huge) {
    G16_LAUNCH((k_spmv_tasks<NM>), ceil_div((uint64_t)P.n_tasks * SPMV_G, 256), 256, 0, stream, M, x,
               (const SpmvTask*)P.tasks.p, P.n_tasks, P.partial.p);
    G16_LAUNCH((k_spmv_huge<NM, Out>), ceil_div(P.n_huge, 64), 64, 0, stream, (const uint32_t*)P.huge_rows.p,
               (const uint32_t*)P.task_off.p, P.n_huge, (const int32_t*)P.partial.p, out);
  }
}

#endif  // device side

}  // namespace g16
