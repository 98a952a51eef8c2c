// witness_map.h -- device-resident state for CircomReduction::witness_map_from_matrices
// (reference src/circom/qap.rs:23-88).  The CSR matrices are the A and B of the reference's
// ConstraintMatrices (src/zkey.rs:151-196): row-major sparse rows of (coeff, index).
#pragma once
#include "ntt29.h"
#include "spmv.h"

namespace g16 {

struct CsrHost {  // host pointers handed over the C ABI
  const uint32_t* rowptr;  // [m+1]
  const uint32_t* col;     // [nnz]
  const Fr* val;           // [nnz] Montgomery
  size_t nnz;
};
struct CsrDev {
  const uint32_t* rowptr;
  const uint32_t* col;
  const Fr* val;
};
struct CsrStore {
  DevBuf<uint32_t> rowptr, col;
  DevBuf<Fr> val;
};

struct WitnessMap {
  uint32_t m = 0, num_inputs = 0, n = 0;
  int reduction = 0;  // G16_REDUCTION_*: CircomReduction (0) or ark-groth16's LibsnarkReduction (1)
  Ntt29Plan plan;
  DevBuf<Fr> cs_lo, cs_hi, ci_lo, ci_hi;  // libsnark: coset tables g^j / n and g^-j / n (g = 5)
  Fr z_inv_packed;                        // libsnark: 1 / (g^n - 1), packed internal
  CsrStore dA, dB;   // cooked (spmv.h): unit coefficients flagged in col, the others as c * 2^266
  SpmvPlan spmv;     // row classes of (A, B): short rows in k_spmv_abc, longer ones in spmv.h's kernels
  DevBuf<int32_t> abc;  // a | b | c as limb planes (ntt29.h): [3][9][n] int32

  void init(const CsrHost& A, const CsrHost& B, uint32_t m, uint32_t num_inputs, int reduction = 0);
  // w_dev: full assignment (>= max column index + 1 elements, Montgomery).
  // h_canon (optional): h as canonical integers (ark-ff into_bigint form) -- what the H-query MSM
  //                     consumes; h_mont (optional): h in the storage form (Montgomery), the value
  //                     CircomReduction::witness_map_from_matrices returns.
  void run(const Fr* w_dev, U256* h_canon, Fr* h_mont, hipStream_t stream);
};

// first row i with (A_i.w)(B_i.w) != C_i.w, or -1 (host pointers; uploads, checks, frees)
long long check_satisfied(const CsrHost& A, const CsrHost& B, const CsrHost& C, uint32_t m,
                          const Fr* w_host, size_t n_vars);

}  // namespace g16
