// witness_map.h -- device-resident state for CircomReduction::witness_map_from_matrices
// (reference src/circom/qap.rs:23-88).  The CSR matrices are the A and B of the reference's
// ConstraintMatrices (src/zkey.rs:151-196): row-major sparse rows of (coeff, index).
#pragma once
#include "ntt.h"

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
  NttPlan plan;
  CsrStore dA, dB;
  DevBuf<Fr> abc;  // a | b | c, n elements each

  void init(const CsrHost& A, const CsrHost& B, uint32_t m, uint32_t num_inputs);
  // w_dev: full assignment (>= max column index + 1 elements, Montgomery); h_dev: n elements out
  void run(const Fr* w_dev, Fr* h_dev, hipStream_t stream);
};

}  // namespace g16
