// finalize.h -- the O(1) tail of ark-groth16's create_proof_with_assignment (equations in
// SURVEY.md section 3.1; reached from reference src/zkey.rs:903-911): r/s blinding, C assembly and the
// three into_affine conversions.
#pragma once
#include "common.h"

namespace g16 {

struct KeyHeaderDev {  // device-resident copy of the O(1) key points
  G1Affine alpha1, beta1, delta1;
  G1Affine a0, b1_0;  // a_query[0], b_g1_query[0]
  G2Affine beta2, delta2, b2_0;
};

struct ProofSums {  // MSM outputs (device)
  G1XYZZ A, B1, L, H;
  G2XYZZ B2;
};

// proof_dev: 256 bytes A|B|C affine.  rs_dev: r, s (Montgomery Fr).
void finalize_proof(const KeyHeaderDev* key, const ProofSums* sums, const Fr* rs_dev,
                    uint8_t* proof_dev, hipStream_t stream);
// ProofSums -> 384-byte affine record A|B1|B2|L|H (one rank's contribution)
void sums_to_partial(const ProofSums* sums, uint8_t* partial_dev, hipStream_t stream);
// world x 384-byte records -> ProofSums (local EC adds: the "all-reduce" tail)
void partials_to_sums(const uint8_t* partials_dev, int world, ProofSums* sums, hipStream_t stream);

}  // namespace g16
