// finalize.h -- the O(1) tail of ark-groth16's create_proof_with_assignment (equations in
// SURVEY.md section 3.1; reached from reference src/zkey.rs:903-911): r/s blinding, C assembly and the
// three into_affine conversions.
//
//   g_a  = r*delta1 + a_query[0] + MSM_A + alpha1
//   g1_b = s*delta1 + b_g1_query[0] + MSM_B1 + beta1
//   g2_b = s*delta2 + b_g2_query[0] + MSM_B2 + beta2
//   g_c  = s*g_a + r*g1_b - (r*s)*delta1 + MSM_L + MSM_H
//
// Split in four kernels so that nothing but a few point additions and one affine conversion sit
// on the critical path of a proof:
//   fin_fixed  needs only (r, s): r*delta1, s*delta1, rs*delta1, s*delta2 from per-key tables of
//              2^i * delta (one table entry per scalar bit, tree-summed by a workgroup).  Launched
//              on the side stream when the proof starts.
//   fin_var    needs MSM_A, MSM_B1: forms g_a, g1_b, writes A, and runs the two variable-base
//              multiplications s*g_a, r*g1_b (4-bit windows, one wave each) on the side stream
//              while the L / B2 / H MSMs run on the main stream.
//   fin_b      needs s*delta2 and MSM_B2: g2_b -> B, on its own stream as soon as the B2 reduction
//              is done (the Fq2 inversion hides under the H MSM).
//   fin_final  needs everything: g_c assembly + its affine conversion.
#pragma once
#include "common.h"
#include "ec29.h"

namespace g16 {

struct KeyHeaderDev {  // device-resident copy of the O(1) key points (storage form, as uploaded)
  G1Affine alpha1, beta1, delta1;
  G1Affine a0, b1_0;  // a_query[0], b_g1_query[0]
  G2Affine beta2, delta2, b2_0;
};

struct ProofSums {  // MSM outputs (device), lazy internal form
  G1XYZZ29 A, B1, L, H;
  G2XYZZ29 B2;
  G1XYZZ29 sA, rB1;  // sharded provers only: s * A and r * B1 of this rank's shard
};

struct FinTables {  // per key: 2^i * P, i < 256, for the fixed bases of the finalisation
  G1XYZZ29 d1[256];  // delta1
  G2XYZZ29 d2[256];  // delta2
  G1XYZZ29 ta[256];  // a_query[0] + alpha1
  G1XYZZ29 tb[256];  // b_g1_query[0] + beta1
};

struct FinScratch {  // per proof
  G1XYZZ29 rd1, sd1, rsd1;  // r*delta1, s*delta1, rs*delta1
  G2XYZZ29 sd2;             // s*delta2
  G1XYZZ29 sga, rgb;        // s*g_a, r*g1_b
  G1XYZZ29 sta, rtb;        // s*(a0 + alpha1), r*(b1_0 + beta1)   (sharded provers)
};

void fin_build_tables(const KeyHeaderDev* key, FinTables* tab, hipStream_t stream);
void fin_fixed(const FinTables* tab, const Fr* rs_dev, FinScratch* scr, hipStream_t stream);
void fin_var(const KeyHeaderDev* key, const ProofSums* sums, const Fr* rs_dev, FinScratch* scr,
             uint8_t* proof_dev, hipStream_t stream);
void fin_b(const KeyHeaderDev* key, const ProofSums* sums, const FinScratch* scr, uint8_t* proof_dev,
           hipStream_t stream);
void fin_final(const KeyHeaderDev* key, const ProofSums* sums, const FinScratch* scr,
               uint8_t* proof_dev, hipStream_t stream);
// fin_final with C left in XYZZ form at proj_dev + FIN_PROJ_C (below); fin_host_affine_c divides on the host and
// writes proof[192, 256): the single-device prover's tail (g16_prove / g16_prove_dev)
void fin_final_proj(const KeyHeaderDev* key, const ProofSums* sums, const FinScratch* scr, uint8_t* proj_dev,
                    hipStream_t stream);
void fin_host_affine_c(const uint8_t* proj_host, uint8_t* proof_out);

// ---- sharded provers (one process per GPU) ----------------------------------------------------
// The variable-base products of the finalisation are linear in the MSM sums, so every rank
// multiplies ITS partial sums while its remaining MSMs run:
//   s*g_a = s*r*delta1 + s*(a0 + alpha1) + sum_ranks s*A_rank        (and likewise r*g1_b)
// which leaves only fixed-base products (table tree-sums) and three affine conversions after the
// all-gather.
void fin_partial_var(ProofSums* sums, const Fr* rs_dev, hipStream_t stream);  // sA, rB1 from A, B1
void fin_fixed_dist(const FinTables* tab, const Fr* rs_dev, FinScratch* scr, hipStream_t stream);
void fin_final_dist(const KeyHeaderDev* key, const ProofSums* sums, const FinScratch* scr,
                    uint8_t* proof_dev, hipStream_t stream);
// ---- small keys through fixed-base tables (msm_table.h): the same equations in stages, each enqueued
// behind the sums it needs, so that one addition + one affine conversion follow the LAST sum of A, B, C:
//   fin_tab_pre  (after fin_fixed_dist)   a' = r delta1 + a0 + alpha1 (-> scr.sga), b' = s delta2 + b2_0 + beta2 (-> scr.sd2)
//   fin_tab_ac   (after A, B1, L, sA, rB1) A = a' + MSM_A -> proj; c' = sA + rB1 + sta + rtb + rs delta1 + L (-> scr.rgb)
//   fin_tab_c    (after H)                 C = c' + MSM_H -> proj
//   fin_tab_b    (after B2)                B = b' + MSM_B2 -> proj
// A, B, C leave the device in XYZZ form (storage Montgomery, FIN_PROJ_BYTES at proj_dev) and the host
// divides (fin_tab_host_affine: three field inversions by binary Euclid, ~5 us each on a CPU core against
// 150-250 us on one GPU lane; the host waits for these bytes anyway).  Same field elements, same bytes.
constexpr int FIN_PROJ_A = 0, FIN_PROJ_B = 128, FIN_PROJ_C = 384, FIN_PROJ_BYTES = 512;
void fin_tab_pre(const KeyHeaderDev* key, FinScratch* scr, hipStream_t stream);
void fin_tab_ac(const ProofSums* sums, FinScratch* scr, uint8_t* proj_dev, hipStream_t stream);
void fin_tab_c(const ProofSums* sums, const FinScratch* scr, uint8_t* proj_dev, hipStream_t stream);
void fin_tab_b(const ProofSums* sums, const FinScratch* scr, uint8_t* proj_dev, hipStream_t stream);
void fin_tab_host_affine(const uint8_t* proj_host, uint8_t* proof_out);  // host: FIN_PROJ_BYTES -> 256 proof bytes
constexpr int FIN_PARTIAL_BYTES = 1024;  // A | B1 | B2 | L | H | sA | rB1 in XYZZ storage form (G1 128 B, G2 256 B)
// ProofSums -> one rank's record (projective: no inversion before the all-gather)
void sums_to_partial(const ProofSums* sums, uint8_t* partial_dev, hipStream_t stream);
// A -> out[0,64), B2 -> out[128,256), affine storage form (g16_msm_g1 / g16_msm_g2)
void sums_to_affine(const ProofSums* sums, uint8_t* out_dev, hipStream_t stream);
// world x FIN_PARTIAL_BYTES records -> ProofSums (local EC adds: the "all-reduce" tail)
void partials_to_sums(const uint8_t* partials_dev, int world, ProofSums* sums, hipStream_t stream);

}  // namespace g16
