// finalize.hip -- proof assembly (see finalize.h).  O(1) group operations per proof:
//   g_a  = r*delta1 + a_query[0] + MSM_A + alpha1
//   g1_b = s*delta1 + b_g1_query[0] + MSM_B1 + beta1
//   g2_b = s*delta2 + b_g2_query[0] + MSM_B2 + beta2
//   g_c  = s*g_a + r*g1_b - (r*s)*delta1 + MSM_L + MSM_H
// The independent scalar multiplications run in different lanes (G1) / a different wave (G2) of
// one small workgroup.
#include "finalize.h"

namespace g16 {

namespace {

__global__ void __launch_bounds__(128) k_finalize(const KeyHeaderDev* key, const ProofSums* sums,
                                                  const Fr* rs, uint8_t* proof) {
  __shared__ G1XYZZ sh[4];  // 0: r*delta1 -> s*g_a, 1: s*delta1 -> r*g1_b, 2: rs*delta1
  const int t = threadIdx.x;
  const Fr r = rs[0], s = rs[1];
  if (t < 3) {
    const U256 k = (t == 0 ? r : (t == 1 ? s : r * s)).to_canonical();
    sh[t] = G1XYZZ::from_affine(key->delta1).mul(k);
  } else if (t == 64) {
    G2XYZZ b = G2XYZZ::from_affine(key->delta2).mul(s.to_canonical());
    b.madd(key->b2_0);
    b.add(sums->B2);
    b.madd(key->beta2);
    *reinterpret_cast<G2Affine*>(proof + 64) = b.to_affine();
  }
  __syncthreads();
  if (t < 2) {
    G1XYZZ g = sh[t];
    g.madd(t == 0 ? key->a0 : key->b1_0);
    g.add(t == 0 ? sums->A : sums->B1);
    g.madd(t == 0 ? key->alpha1 : key->beta1);
    if (t == 0) *reinterpret_cast<G1Affine*>(proof) = g.to_affine();
    sh[t] = g.mul((t == 0 ? s : r).to_canonical());
  }
  __syncthreads();
  if (t == 0) {
    G1XYZZ c = sh[0];
    c.add(sh[1]);
    c.add(sh[2].neg());
    c.add(sums->L);
    c.add(sums->H);
    *reinterpret_cast<G1Affine*>(proof + 192) = c.to_affine();
  }
}

__global__ void __launch_bounds__(128) k_sums_to_partial(const ProofSums* sums, uint8_t* out) {
  const int t = threadIdx.x;
  if (t == 0) *reinterpret_cast<G1Affine*>(out) = sums->A.to_affine();
  if (t == 1) *reinterpret_cast<G1Affine*>(out + 64) = sums->B1.to_affine();
  if (t == 2) *reinterpret_cast<G1Affine*>(out + 256) = sums->L.to_affine();
  if (t == 3) *reinterpret_cast<G1Affine*>(out + 320) = sums->H.to_affine();
  if (t == 64) *reinterpret_cast<G2Affine*>(out + 128) = sums->B2.to_affine();
}

__global__ void __launch_bounds__(128) k_partials_to_sums(const uint8_t* parts, int world,
                                                          ProofSums* sums) {
  const int t = threadIdx.x;
  if (t < 4) {
    const int off = t == 0 ? 0 : (t == 1 ? 64 : (t == 2 ? 256 : 320));
    G1XYZZ acc = G1XYZZ::infinity();
    for (int k = 0; k < world; ++k)
      acc.madd(*reinterpret_cast<const G1Affine*>(parts + (size_t)k * 384 + off));
    (t == 0 ? sums->A : (t == 1 ? sums->B1 : (t == 2 ? sums->L : sums->H))) = acc;
  } else if (t == 64) {
    G2XYZZ acc = G2XYZZ::infinity();
    for (int k = 0; k < world; ++k)
      acc.madd(*reinterpret_cast<const G2Affine*>(parts + (size_t)k * 384 + 128));
    sums->B2 = acc;
  }
}

}  // namespace

void finalize_proof(const KeyHeaderDev* key, const ProofSums* sums, const Fr* rs_dev,
                    uint8_t* proof_dev, hipStream_t stream) {
  G16_LAUNCH(k_finalize, 1, 128, 0, stream, key, sums, rs_dev, proof_dev);
}
void sums_to_partial(const ProofSums* sums, uint8_t* partial_dev, hipStream_t stream) {
  G16_LAUNCH(k_sums_to_partial, 1, 128, 0, stream, sums, partial_dev);
}
void partials_to_sums(const uint8_t* partials_dev, int world, ProofSums* sums, hipStream_t stream) {
  G16_LAUNCH(k_partials_to_sums, 1, 128, 0, stream, partials_dev, world, sums);
}

}  // namespace g16
