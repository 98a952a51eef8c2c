// witness_map.hip -- CircomReduction::witness_map_from_matrices on the GPU.
//
// Follows reference src/circom/qap.rs:23-88 step for step:
//   :37-58  a = A.w, b = B.w (rows >= m: a gets the first num_inputs witness values), c = a o b
//           -> k_spmv_abc (one pass, c fused)
//   :60-61,69-70,79-80  ifft + distribute_powers(omega_2n) -> ntt29_dif(inverse, TWIST_SCALE), batch of 3
//   :72-73,81           fft                                 -> ntt29_dit, batch of 3
//   :75,83-85           ab - c                              -> k_mul_sub
// Results are field elements, so they are bit-identical to the reference whatever the schedule.
#include "witness_map.h"

namespace g16 {

namespace {

__device__ __forceinline__ Fr row_dot(const uint32_t* rowptr, const uint32_t* col, const Fr* val,
                                      const Fr* w, uint32_t i) {
  // evaluate_constraint (ark-groth16 r1cs_to_qap, called at qap.rs:42-43): sum coeff * w[idx];
  // the multiply is skipped when coeff == 1, as upstream does.
  Fr acc = Fr::zero();
  const uint32_t e = rowptr[i + 1];
  const Fr one = Fr::one();
  for (uint32_t j = rowptr[i]; j < e; ++j) {
    Fr x = w[col[j]];
    Fr cf = val[j];
    if (cf != one) x = x * cf;
    acc = acc + x;
  }
  return acc;
}

// a, b, c = a o b of one row into the NTT's limb planes
struct AbcOut {
  int32_t* abc;
  uint32_t n;
  __device__ __forceinline__ void put(uint32_t i, const Fr29* v) const {
    const size_t vs = (size_t)NTT29_LIMBS * n;
    store_planes(abc, n, i, v[0]);
    store_planes(abc + vs, n, i, v[1]);
    store_planes(abc + 2 * vs, n, i, v[0] * v[1]);
  }
};

// Short rows (<= SPMV_SHORT terms in A and in B: what a chain of products or wire copies consists
// of) and the rows past the matrices; medium and huge rows belong to spmv_run_long (spmv.h).
__global__ void __launch_bounds__(256) k_spmv_abc(SpmvDev A, SpmvDev B, const Fr* w, uint32_t m,
                                                  uint32_t num_inputs, uint32_t n, AbcOut out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr29 v[2] = {Fr29::zero(), Fr29::zero()};
  if (i < m) {
    if (!spmv_row_is_short(A, i) || !spmv_row_is_short(B, i)) return;
    v[0] = spmv_row_thread(A, w, i);
    v[1] = spmv_row_thread(B, w, i);
  } else if (i < m + num_inputs) {
    v[0] = Fr29::from_mont256(w[i - m]);  // qap.rs:46-50
  }
  out.put(i, v);
}

// h = a * b - c (qap.rs:75,83-85), written as canonical integers and / or in the storage form
__global__ void __launch_bounds__(256) k_mul_sub(const int32_t* abc, U256* h_canon, Fr* h_mont,
                                                 uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t vs = (size_t)NTT29_LIMBS * n;
  const Fr29 one = Fr29::one();
  const Fr29 a = load_planes(abc, n, i) * one;  // forward-NTT outputs are < 24 r: bring one factor below 2 r
  const Fr29 b = load_planes(abc + vs, n, i);
  const Fr29 c = load_planes(abc + 2 * vs, n, i);
  const Fr29 h = Fr29::mul2(a, b, c.neg(), one);
  if (h_mont) h_mont[i] = h.to_mont256();
  if (h_canon) {
    // internal value x * 2^261 -> the integer x: one Montgomery reduction (multiplication by the
    // integer 1), then the unique representative in [0, r)
    f29::L9 uno{};
    uno.v[0] = 1;
    const Fr29 x = (h * Fr29::from_limbs(uno)).canonical();
    U256 u;
    x.pack(u.v);
    h_canon[i] = u;
  }
}

// LibsnarkReduction, after the coset transforms: vector 0 <- (a b - c) / Z_H(coset)
__global__ void __launch_bounds__(256) k_libsnark_quotient(int32_t* abc, Fr z_inv, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t vs = (size_t)NTT29_LIMBS * n;
  const Fr29 one = Fr29::one();
  const Fr29 a = load_planes(abc, n, i) * one;
  const Fr29 b = load_planes(abc + vs, n, i);
  const Fr29 c = load_planes(abc + 2 * vs, n, i);
  const Fr29 q = Fr29::mul2(a, b, c.neg(), one) * Fr29::unpack(z_inv.v);
  store_planes(abc, n, i, q);
}

// LibsnarkReduction, last step: the inverse coset DIF left coefficient j at position bitrev(j)
__global__ void __launch_bounds__(256) k_libsnark_finish(const int32_t* planes, int k, U256* h_canon,
                                                         Fr* h_mont, uint32_t n) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t pos = k ? (__brev(j) >> (32 - k)) : 0u;
  const Fr29 h = load_planes(planes, n, pos);
  if (h_mont) h_mont[j] = h.to_mont256();
  if (h_canon) {
    f29::L9 uno{};
    uno.v[0] = 1;
    const Fr29 x = (h * Fr29::from_limbs(uno)).canonical();
    U256 u;
    x.pack(u.v);
    h_canon[j] = u;
  }
}

// (A_i . w)(B_i . w) == C_i . w for every row; records the smallest failing row
__global__ void __launch_bounds__(256) k_check_rows(CsrDev A, CsrDev B, CsrDev C, const Fr* w,
                                                    uint32_t m, unsigned long long* first_bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const Fr a = row_dot(A.rowptr, A.col, A.val, w, i);
  const Fr b = row_dot(B.rowptr, B.col, B.val, w, i);
  const Fr c = row_dot(C.rowptr, C.col, C.val, w, i);
  if (a * b != c) atomicMin(first_bad, (unsigned long long)i);
}

}  // namespace

// The satisfiability check CircomBuilder::build runs in debug builds (reference
// src/circom/builder.rs:101-114, ConstraintSystem::is_satisfied in tests at circuit.rs:92-107)
// as one kernel.  Returns the first unsatisfied row, or -1.
long long check_satisfied(const CsrHost& A, const CsrHost& B, const CsrHost& C, uint32_t m,
                          const Fr* w_host, size_t n_vars) {
  CsrStore d[3];
  const CsrHost* hs[3] = {&A, &B, &C};
  for (int k = 0; k < 3; ++k) {
    d[k].rowptr.alloc((size_t)m + 1);
    d[k].col.alloc(hs[k]->nnz ? hs[k]->nnz : 1);
    d[k].val.alloc(hs[k]->nnz ? hs[k]->nnz : 1);
    G16_HIP(hipMemcpy(d[k].rowptr.p, hs[k]->rowptr, ((size_t)m + 1) * 4, hipMemcpyHostToDevice));
    if (hs[k]->nnz) {
      G16_HIP(hipMemcpy(d[k].col.p, hs[k]->col, hs[k]->nnz * 4, hipMemcpyHostToDevice));
      G16_HIP(hipMemcpy(d[k].val.p, hs[k]->val, hs[k]->nnz * sizeof(Fr), hipMemcpyHostToDevice));
    }
  }
  DevBuf<Fr> w;
  w.alloc(n_vars ? n_vars : 1);
  G16_HIP(hipMemcpy(w.p, w_host, n_vars * sizeof(Fr), hipMemcpyHostToDevice));
  DevBuf<unsigned long long> bad;
  bad.alloc(1);
  const unsigned long long none = ~0ull;
  G16_HIP(hipMemcpy(bad.p, &none, 8, hipMemcpyHostToDevice));
  if (m) {
    CsrDev a{d[0].rowptr.p, d[0].col.p, d[0].val.p}, b{d[1].rowptr.p, d[1].col.p, d[1].val.p},
        c{d[2].rowptr.p, d[2].col.p, d[2].val.p};
    G16_LAUNCH(k_check_rows, ceil_div(m, 256), 256, 0, nullptr, a, b, c, (const Fr*)w.p, m, bad.p);
  }
  unsigned long long out = none;
  G16_HIP(hipDeviceSynchronize());
  G16_HIP(hipMemcpy(&out, bad.p, 8, hipMemcpyDeviceToHost));
  return out == none ? -1 : (long long)out;
}

void WitnessMap::init(const CsrHost& A, const CsrHost& B, uint32_t m_, uint32_t num_inputs_,
                      int reduction_) {
  m = m_;
  num_inputs = num_inputs_;
  reduction = reduction_;
  uint64_t need = (uint64_t)m + num_inputs;
  int k = 0;
  while (((uint64_t)1 << k) < need) ++k;
  // qap.rs:31,66: both the size-n and the size-2n domain must exist (Fr two-adicity 28)
  if (k + 1 > 28) throw std::runtime_error("PolynomialDegreeTooLarge");
  plan.build(k, nullptr);
  n = (uint32_t)plan.n();
  if (reduction == 1) {
    // ark-groth16 LibsnarkReduction: coset g * H with g = Fr::GENERATOR = 5
    const Fr g = Fr::from_u32(5), gi = g.inv();
    plan.make_twist(g, plan.base.n_inv, cs_lo, cs_hi);
    plan.make_twist(gi, plan.base.n_inv, ci_lo, ci_hi);
    const Fr z = fr_pow_u64(g, n) - Fr::one();  // Z_H on the coset: g^n - 1
    Fr29::from_mont256(z.inv()).pack_internal(z_inv_packed.v);
  }
  auto up = [&](const CsrHost& h, CsrStore& d) {
    d.rowptr.alloc((size_t)m + 1);
    d.col.alloc(h.nnz ? h.nnz : 1);
    d.val.alloc(h.nnz ? h.nnz : 1);
    G16_HIP(hipMemcpy(d.rowptr.p, h.rowptr, ((size_t)m + 1) * 4, hipMemcpyHostToDevice));
    if (h.nnz) {
      G16_HIP(hipMemcpy(d.col.p, h.col, h.nnz * 4, hipMemcpyHostToDevice));
      G16_HIP(hipMemcpy(d.val.p, h.val, h.nnz * sizeof(Fr), hipMemcpyHostToDevice));
    }
  };
  up(A, dA);
  up(B, dB);
  spmv_cook(dA.col.p, dA.val.p, A.nnz, nullptr);
  spmv_cook(dB.col.p, dB.val.p, B.nnz, nullptr);
  const uint32_t* rps[2] = {A.rowptr, B.rowptr};
  spmv.build(rps, 2, m);
  abc.alloc((size_t)3 * NTT29_LIMBS * n);
}

void WitnessMap::run(const Fr* w_dev, U256* h_canon, Fr* h_mont, hipStream_t stream) {
  const SpmvMats<2> M{{SpmvDev{dA.rowptr.p, dA.col.p, dA.val.p}, SpmvDev{dB.rowptr.p, dB.col.p, dB.val.p}}};
  const size_t vs = (size_t)NTT29_LIMBS * n;
  const AbcOut out{abc.p, n};
  G16_LAUNCH(k_spmv_abc, ceil_div(n, 256), 256, 0, stream, M.m[0], M.m[1], w_dev, m, num_inputs, n, out);
  spmv_run_long<2, AbcOut>(spmv, M, w_dev, out, stream);
  if (reduction == 1) {
    // LibsnarkReduction::witness_map_from_matrices (ark-groth16; call sites reference
    // tests/groth16.rs:25-35): ifft, coset fft with g = 5, (a b - c) / Z, inverse coset fft
    ntt29_dif(plan, abc.p, vs, 3, /*inverse=*/true, NTT_FUSE_TWIST_SCALE, stream, cs_lo.p, cs_hi.p);
    ntt29_dit(plan, abc.p, vs, 3, stream);
    G16_LAUNCH(k_libsnark_quotient, ceil_div(n, 256), 256, 0, stream, abc.p, z_inv_packed, n);
    ntt29_dif(plan, abc.p, vs, 1, /*inverse=*/true, NTT_FUSE_TWIST_SCALE, stream, ci_lo.p, ci_hi.p);
    G16_LAUNCH(k_libsnark_finish, ceil_div(n, 256), 256, 0, stream, (const int32_t*)abc.p, plan.base.k,
               h_canon, h_mont, n);
    return;
  }
  ntt29_dif(plan, abc.p, vs, 3, /*inverse=*/true, NTT_FUSE_TWIST_SCALE, stream);
  ntt29_dit(plan, abc.p, vs, 3, stream);
  G16_LAUNCH(k_mul_sub, ceil_div(n, 256), 256, 0, stream, (const int32_t*)abc.p, h_canon, h_mont, n);
}

}  // namespace g16
