// witness_map.hip -- CircomReduction::witness_map_from_matrices on the GPU.
//
// Follows reference src/circom/qap.rs:23-88 step for step:
//   :37-58  a = A.w, b = B.w (rows >= m: a gets the first num_inputs witness values), c = a o b
//           -> k_spmv_abc (one pass, c fused)
//   :60-61,69-70,79-80  ifft + distribute_powers(omega_2n) -> ntt_dif(inverse, TWIST_SCALE), batch of 3
//   :72-73,81           fft                                 -> ntt_dit, batch of 3
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

__global__ void __launch_bounds__(256) k_spmv_abc(CsrDev A, CsrDev B, const Fr* w, uint32_t m,
                                                  uint32_t num_inputs, uint32_t n, Fr* a, Fr* b,
                                                  Fr* c) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr ai = Fr::zero(), bi = Fr::zero(), ci = Fr::zero();
  if (i < m) {
    ai = row_dot(A.rowptr, A.col, A.val, w, i);
    bi = row_dot(B.rowptr, B.col, B.val, w, i);
    ci = ai * bi;
  } else if (i < m + num_inputs) {
    ai = w[i - m];  // qap.rs:46-50
  }
  a[i] = ai;
  b[i] = bi;
  c[i] = ci;
}

__global__ void __launch_bounds__(256) k_mul_sub(const Fr* a, const Fr* b, const Fr* c, Fr* h,
                                                 uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  h[i] = a[i] * b[i] - c[i];
}

}  // namespace

void WitnessMap::init(const CsrHost& A, const CsrHost& B, uint32_t m_, uint32_t num_inputs_) {
  m = m_;
  num_inputs = num_inputs_;
  uint64_t need = (uint64_t)m + num_inputs;
  int k = 0;
  while (((uint64_t)1 << k) < need) ++k;
  // qap.rs:31,66: both the size-n and the size-2n domain must exist (Fr two-adicity 28)
  if (k + 1 > 28) throw std::runtime_error("PolynomialDegreeTooLarge");
  plan.build(k);
  n = (uint32_t)plan.n;
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
  abc.alloc((size_t)3 * n);
}

void WitnessMap::run(const Fr* w_dev, Fr* h_dev, hipStream_t stream) {
  Fr* a = abc.p;
  Fr* b = abc.p + n;
  Fr* c = abc.p + 2 * (size_t)n;
  CsrDev A{dA.rowptr.p, dA.col.p, dA.val.p};
  CsrDev B{dB.rowptr.p, dB.col.p, dB.val.p};
  G16_LAUNCH(k_spmv_abc, ceil_div(n, 256), 256, 0, stream, A, B, w_dev, m, num_inputs, n, a, b, c);
  ntt_dif(plan, abc.p, n, 3, /*inverse=*/true, NTT_FUSE_TWIST_SCALE, stream);
  ntt_dit(plan, abc.p, n, 3, /*inverse=*/false, stream);
  G16_LAUNCH(k_mul_sub, ceil_div(n, 256), 256, 0, stream, (const Fr*)a, (const Fr*)b,
             (const Fr*)c, h_dev, n);
}

}  // namespace g16
