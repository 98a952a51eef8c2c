// finalize.hip -- proof assembly (see finalize.h).  All arithmetic on the lazy limbs of
// field29.h / ec29.h; results are converted to the storage form (affine, Montgomery R = 2^256,
// the zkey point encoding of reference src/zkey.rs:340-360) only when written to the proof.
#include "finalize.h"

namespace g16 {

namespace {

constexpr int FIN_T = 128;

template <class A>
__device__ __forceinline__ A tree_sum(A v, A* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int off = FIN_T / 2; off > 0; off >>= 1) {
    if (t < off) {
      A a = sh[t];
      a.add(sh[t + off]);
      sh[t] = a;
    }
    __syncthreads();
  }
  return sh[0];
}

template <class F>
__device__ __forceinline__ Affine<F> to_storage_affine(const XYZZ29<typename Lazy<F>::type>& a) {
  const auto p = a.template to_affine<true>();  // one lane works here: binary-GCD inversion
  if (p.inf) return Affine<F>::infinity();
  return Affine<F>{p.x.to_mont256(), p.y.to_mont256()};
}

// one block per table; thread 0 doubles 255 times (ctx-create only)
__global__ void k_fin_tables(const KeyHeaderDev* key, FinTables* tab) {
  if (threadIdx.x != 0) return;
  if (blockIdx.x == 1) {
    G2XYZZ29 q = G2XYZZ29::from_affine(affine_from_mont256<Fq2>(key->delta2));
    for (int i = 0; i < 256; ++i) {
      tab->d2[i] = q;
      q.dbl_in_place();
    }
    return;
  }
  G1XYZZ29 q;
  G1XYZZ29* dst;
  if (blockIdx.x == 0) {
    q = G1XYZZ29::from_affine(affine_from_mont256<Fq>(key->delta1));
    dst = tab->d1;
  } else if (blockIdx.x == 2) {
    q = G1XYZZ29::from_affine(affine_from_mont256<Fq>(key->a0));
    q.madd(affine_from_mont256<Fq>(key->alpha1));
    dst = tab->ta;
  } else {
    q = G1XYZZ29::from_affine(affine_from_mont256<Fq>(key->b1_0));
    q.madd(affine_from_mont256<Fq>(key->beta1));
    dst = tab->tb;
  }
  for (int i = 0; i < 256; ++i) {
    dst[i] = q;
    q.dbl_in_place();
  }
}

__device__ __forceinline__ bool bit_of(const U256& k, int i) { return (k.v[i >> 5] >> (i & 31)) & 1u; }

// blocks 0..2: k * delta1 for k = r, s, rs; block 3: s * delta2.  Thread t owns bits t and t+128.
__global__ void __launch_bounds__(FIN_T) k_fin_fixed(const FinTables* tab, const Fr* rs,
                                                     FinScratch* scr) {
  G16_DYN_SMEM(smem_raw);
  const int t = threadIdx.x;
  const Fr r = rs[0], s = rs[1];
  const int b = blockIdx.x;
  const U256 k = (b == 0 ? r : (b == 1 || b == 3 ? s : r * s)).to_canonical();
  if (b < 3) {
    G1XYZZ29* sh = reinterpret_cast<G1XYZZ29*>(smem_raw);
    G1XYZZ29 v = G1XYZZ29::infinity();
    if (bit_of(k, t)) v = tab->d1[t];
    if (bit_of(k, t + FIN_T)) v.add(tab->d1[t + FIN_T]);
    G1XYZZ29 tot = tree_sum(v, sh);
    if (t == 0) (b == 0 ? scr->rd1 : (b == 1 ? scr->sd1 : scr->rsd1)) = tot;
  } else {
    G2XYZZ29* sh = reinterpret_cast<G2XYZZ29*>(smem_raw);
    G2XYZZ29 v = G2XYZZ29::infinity();
    if (bit_of(k, t)) v = tab->d2[t];
    if (bit_of(k, t + FIN_T)) v.add(tab->d2[t + FIN_T]);
    G2XYZZ29 tot = tree_sum(v, sh);
    if (t == 0) scr->sd2 = tot;
  }
}

// ---- variable-base k * P with the GLV endomorphism of BN254 G1 -------------------------------------
// phi(x, y) = (beta x, y) = lambda (x, y) with beta^3 = 1 in Fq, lambda^3 = 1 in Fr.  k is split as
// k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^127 (lattice basis (A1, -B1), (A2, B2) of
// {(a, b): a + b lambda = 0 mod r}, rounded quotients through 2^256-scaled reciprocals), so the
// doubling chain is 128 long instead of 254.  One lane works (latency-bound by design: this runs
// beside the big MSMs); constants derived and cross-checked with the oracle
// (tests: every proof byte depends on them).
namespace glv {
constexpr uint32_t G1[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u};                              // round(2^256 B2 / r)
constexpr uint32_t G2[5] = {0x391eb18eu, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u};    // round(2^256 |B1| / r)
constexpr uint32_t A1[2] = {0x94d213e3u, 0x89d32568u};
constexpr uint32_t A2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
constexpr uint32_t NB1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};                // -B1 > 0
constexpr uint32_t B2[2] = {0x94d213e3u, 0x89d32568u};
constexpr uint32_t BETA[8] = {0x77fffffeu, 0x57634731u, 0xacdb5c4fu, 0xd4f263f1u,
                              0xa0d48bacu, 0x59e26bceu, 0x00000000u, 0x00000000u};

// out[0..NA+NB) = a * b (schoolbook, 32-bit limbs)
template <int NA, int NB>
__device__ __forceinline__ void mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  for (int i = 0; i < NA + NB; ++i) out[i] = 0;
  for (int i = 0; i < NA; ++i) {
    uint64_t c = 0;
    for (int j = 0; j < NB; ++j) {
      c += (uint64_t)a[i] * b[j] + out[i + j];
      out[i + j] = (uint32_t)c;
      c >>= 32;
    }
    out[i + NB] = (uint32_t)c;
  }
}
// acc (8 limbs, mod 2^256) += / -= x (n limbs)
__device__ __forceinline__ void add_to(uint32_t* acc, const uint32_t* x, int n) {
  uint64_t c = 0;
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)acc[i] + (i < n ? x[i] : 0u);
    acc[i] = (uint32_t)c;
    c >>= 32;
  }
}
__device__ __forceinline__ void sub_from(uint32_t* acc, const uint32_t* x, int n) {
  uint64_t br = 0;
  for (int i = 0; i < 8; ++i) {
    const uint64_t t = (uint64_t)acc[i] - (i < n ? x[i] : 0u) - br;
    acc[i] = (uint32_t)t;
    br = (t >> 32) & 1;
  }
}
// |acc| and its sign (two's complement, 256 bits)
__device__ __forceinline__ bool abs_in_place(uint32_t* acc) {
  if (!(acc[7] >> 31)) return false;
  uint64_t c = 1;
  for (int i = 0; i < 8; ++i) {
    c += (uint32_t)~acc[i];
    acc[i] = (uint32_t)c;
    c >>= 32;
  }
  return true;
}

struct Split {
  uint32_t k1[8], k2[8];  // magnitudes (< 2^127)
  bool neg1, neg2;
};

__device__ Split split(const U256& k) {
  uint32_t t1[11], t2[13];
  mul<8, 3>(k.v, G1, t1);
  mul<8, 5>(k.v, G2, t2);
  // c = (k * G + 2^255) >> 256: add the rounding bit at limb 7, keep limbs >= 8
  uint32_t c1[3], c2[5];
  {
    uint64_t c = (uint64_t)t1[7] + 0x80000000u;
    c >>= 32;
    for (int i = 0; i < 3; ++i) {
      c += t1[8 + i];
      c1[i] = (uint32_t)c;
      c >>= 32;
    }
    c = (uint64_t)t2[7] + 0x80000000u;
    c >>= 32;
    for (int i = 0; i < 5; ++i) {
      c += t2[8 + i];
      c2[i] = (uint32_t)c;
      c >>= 32;
    }
  }
  Split s;
  uint32_t p[9];
  // k1 = k - c1 A1 - c2 A2
  for (int i = 0; i < 8; ++i) s.k1[i] = k.v[i];
  mul<3, 2>(c1, A1, p);
  sub_from(s.k1, p, 5);
  mul<5, 4>(c2, A2, p);
  sub_from(s.k1, p, 8);
  // k2 = c1 |B1| - c2 B2
  for (int i = 0; i < 8; ++i) s.k2[i] = 0;
  mul<3, 4>(c1, NB1, p);
  add_to(s.k2, p, 7);
  mul<5, 2>(c2, B2, p);
  sub_from(s.k2, p, 7);
  s.neg1 = abs_in_place(s.k1);
  s.neg2 = abs_in_place(s.k2);
  return s;
}
}  // namespace glv

// k * P: 4-bit fixed windows over (k1, k2) jointly, MSB first; multiples 1..15 of P in LDS, the
// multiples of phi(P) are phi of those (one product by beta)
__device__ G1XYZZ29 var_mul(const G1XYZZ29& P, const U256& k, G1XYZZ29* tbl) {
  tbl[0] = G1XYZZ29::infinity();
  tbl[1] = P;
  for (int i = 2; i < 16; ++i) {
    G1XYZZ29 q = tbl[i - 1];
    q.add(P);
    tbl[i] = q;
  }
  const glv::Split sp = glv::split(k);
  U256 bu;
  for (int i = 0; i < 8; ++i) bu.v[i] = glv::BETA[i];
  const Fq29 beta = Fq29::from_mont256(Fq::from_canonical(bu));
  G1XYZZ29 acc = G1XYZZ29::infinity();
  for (int w = 31; w >= 0; --w) {
    for (int d = 0; d < 4; ++d) acc.dbl_in_place();
    const uint32_t n1 = (sp.k1[w >> 3] >> ((w & 7) * 4)) & 15u;
    const uint32_t n2 = (sp.k2[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (n1) {
      G1XYZZ29 t = tbl[n1];
      if (sp.neg1) t = t.neg();
      acc.add(t);
    }
    if (n2) {
      G1XYZZ29 t = tbl[n2];
      t.x = t.x * beta;  // phi
      if (sp.neg2) t = t.neg();
      acc.add(t);
    }
  }
  return acc;
}

// block 0: g_a, A, s*g_a      block 1: g1_b, r*g1_b
__global__ void __launch_bounds__(64) k_fin_var(const KeyHeaderDev* key, const ProofSums* sums,
                                                const Fr* rs, FinScratch* scr, uint8_t* proof) {
  __shared__ G1XYZZ29 tbl[16];
  if (threadIdx.x != 0) return;
  const int b = blockIdx.x;
  G1XYZZ29 g = b == 0 ? scr->rd1 : scr->sd1;
  g.madd(affine_from_mont256<Fq>(b == 0 ? key->a0 : key->b1_0));
  g.add(b == 0 ? sums->A : sums->B1);
  g.madd(affine_from_mont256<Fq>(b == 0 ? key->alpha1 : key->beta1));
  if (b == 0) *reinterpret_cast<G1Affine*>(proof) = to_storage_affine<Fq>(g);
  const U256 k = (b == 0 ? rs[1] : rs[0]).to_canonical();
  (b == 0 ? scr->sga : scr->rgb) = var_mul(g, k, tbl);
}

// g_c -> C (needs every sum)
__global__ void __launch_bounds__(64) k_fin_final(const KeyHeaderDev* key, const ProofSums* sums,
                                                  const FinScratch* scr, uint8_t* proof) {
  if (threadIdx.x != 0) return;
  G1XYZZ29 c = scr->sga;
  c.add(scr->rgb);
  c.add(scr->rsd1.neg());
  c.add(sums->L);
  c.add(sums->H);
  *reinterpret_cast<G1Affine*>(proof + 192) = to_storage_affine<Fq>(c);
}

// The same with C left in XYZZ form (storage Montgomery) at proj + FIN_PROJ_C: the one field inversion of the affine
// conversion is 0.2 ms on one GPU lane -- on the critical path of EVERY proof, behind the last reduction -- and ~5 us
// on the host, which waits for these bytes anyway (round 6; the table path has done so since round 5).
__global__ void __launch_bounds__(64) k_fin_final_proj(const KeyHeaderDev* key, const ProofSums* sums,
                                                       const FinScratch* scr, uint8_t* proj) {
  if (threadIdx.x != 0) return;
  G1XYZZ29 c = scr->sga;
  c.add(scr->rgb);
  c.add(scr->rsd1.neg());
  c.add(sums->L);
  c.add(sums->H);
  *reinterpret_cast<XYZZ<Fq>*>(proj + FIN_PROJ_C) = xyzz_to_mont256<Fq>(c);
}

// g2_b -> B (needs only s * delta2 and the B2 sum: runs as soon as the B2 reduction is done,
// underneath the H MSM)
__global__ void __launch_bounds__(64) k_fin_b(const KeyHeaderDev* key, const ProofSums* sums,
                                              const FinScratch* scr, uint8_t* proof) {
  if (threadIdx.x != 0) return;
  G2XYZZ29 b = scr->sd2;
  b.madd(affine_from_mont256<Fq2>(key->b2_0));
  b.add(sums->B2);
  b.madd(affine_from_mont256<Fq2>(key->beta2));
  *reinterpret_cast<G2Affine*>(proof + 64) = to_storage_affine<Fq2>(b);
}

// ---- sharded provers ---------------------------------------------------------------------------
// block 0: sA = s * A      block 1: rB1 = r * B1     (this rank's partial sums)
__global__ void __launch_bounds__(64) k_fin_partial_var(ProofSums* sums, const Fr* rs) {
  __shared__ G1XYZZ29 tbl[16];
  if (threadIdx.x != 0) return;
  if (blockIdx.x == 0) sums->sA = var_mul(sums->A, rs[1].to_canonical(), tbl);
  else sums->rB1 = var_mul(sums->B1, rs[0].to_canonical(), tbl);
}

// blocks 0..2: k * delta1 (k = r, s, rs); 3: s * delta2; 4: s * (a0 + alpha1); 5: r * (b1_0 + beta1)
__global__ void __launch_bounds__(FIN_T) k_fin_fixed_dist(const FinTables* tab, const Fr* rs,
                                                          FinScratch* scr) {
  G16_DYN_SMEM(smem_raw);
  const int t = threadIdx.x;
  const Fr r = rs[0], s = rs[1];
  const int b = blockIdx.x;
  const U256 k = (b == 0 || b == 5 ? r : (b == 2 ? r * s : s)).to_canonical();
  if (b == 3) {
    G2XYZZ29* sh = reinterpret_cast<G2XYZZ29*>(smem_raw);
    G2XYZZ29 v = G2XYZZ29::infinity();
    if (bit_of(k, t)) v = tab->d2[t];
    if (bit_of(k, t + FIN_T)) v.add(tab->d2[t + FIN_T]);
    G2XYZZ29 tot = tree_sum(v, sh);
    if (t == 0) scr->sd2 = tot;
    return;
  }
  const G1XYZZ29* base = b <= 2 ? tab->d1 : (b == 4 ? tab->ta : tab->tb);
  G1XYZZ29* sh = reinterpret_cast<G1XYZZ29*>(smem_raw);
  G1XYZZ29 v = G1XYZZ29::infinity();
  if (bit_of(k, t)) v = base[t];
  if (bit_of(k, t + FIN_T)) v.add(base[t + FIN_T]);
  G1XYZZ29 tot = tree_sum(v, sh);
  if (t == 0) {
    G1XYZZ29* dst = b == 0 ? &scr->rd1 : (b == 1 ? &scr->sd1 : (b == 2 ? &scr->rsd1 : (b == 4 ? &scr->sta : &scr->rtb)));
    *dst = tot;
  }
}

// block 0: A      block 1: B      block 2: C   (sums already reduced over the ranks)
__global__ void __launch_bounds__(64) k_fin_final_dist(const KeyHeaderDev* key, const ProofSums* sums,
                                                       const FinScratch* scr, uint8_t* proof) {
  if (threadIdx.x != 0) return;
  if (blockIdx.x == 0) {
    G1XYZZ29 a = scr->rd1;
    a.madd(affine_from_mont256<Fq>(key->a0));
    a.add(sums->A);
    a.madd(affine_from_mont256<Fq>(key->alpha1));
    *reinterpret_cast<G1Affine*>(proof) = to_storage_affine<Fq>(a);
  } else if (blockIdx.x == 1) {
    G2XYZZ29 b = scr->sd2;
    b.madd(affine_from_mont256<Fq2>(key->b2_0));
    b.add(sums->B2);
    b.madd(affine_from_mont256<Fq2>(key->beta2));
    *reinterpret_cast<G2Affine*>(proof + 64) = to_storage_affine<Fq2>(b);
  } else {
    // g_c = s g_a + r g1_b - rs delta1 + L + H with s g_a = rs delta1 + s (a0 + alpha1) + s A, ...
    G1XYZZ29 c = sums->sA;
    c.add(sums->rB1);
    c.add(scr->sta);
    c.add(scr->rtb);
    c.add(scr->rsd1);
    c.add(sums->L);
    c.add(sums->H);
    *reinterpret_cast<G1Affine*>(proof + 192) = to_storage_affine<Fq>(c);
  }
}

// ---- small keys (fixed-base tables, msm_table.h): the tail in stages, so that ONE addition and one
// affine conversion follow the last sum of each proof element ------------------------------------------
// after fin_fixed_dist (side stream).  block 0: a' = r delta1 + a0 + alpha1 -> scr->sga;
// block 1: b' = s delta2 + b2_0 + beta2 -> scr->sd2 (in place)
__global__ void __launch_bounds__(64) k_fin_tab_pre(const KeyHeaderDev* key, FinScratch* scr) {
  if (threadIdx.x != 0) return;
  if (blockIdx.x == 0) {
    G1XYZZ29 a = scr->rd1;
    a.madd(affine_from_mont256<Fq>(key->a0));
    a.madd(affine_from_mont256<Fq>(key->alpha1));
    scr->sga = a;
  } else {
    G2XYZZ29 b = scr->sd2;
    b.madd(affine_from_mont256<Fq2>(key->b2_0));
    b.madd(affine_from_mont256<Fq2>(key->beta2));
    scr->sd2 = b;
  }
}
// The three results leave the device in XYZZ form (storage Montgomery): the one field inversion each
// needs is ~150-250 us on one GPU lane and ~5 us on the host (fin_tab_host_affine), and the host waits for
// these bytes anyway.
// after the witness-scalar G1 sums (main stream).  block 0: A = a' + MSM_A -> proj;
// block 1: c' = s A + r B1 + s (a0 + alpha1) + r (b1_0 + beta1) + rs delta1 + L -> scr->rgb
__global__ void __launch_bounds__(64) k_fin_tab_ac(const ProofSums* sums, FinScratch* scr, uint8_t* proj) {
  if (threadIdx.x != 0) return;
  if (blockIdx.x == 0) {
    G1XYZZ29 a = scr->sga;
    a.add(sums->A);
    *reinterpret_cast<XYZZ<Fq>*>(proj + FIN_PROJ_A) = xyzz_to_mont256<Fq>(a);
  } else {
    G1XYZZ29 c = sums->sA;
    c.add(sums->rB1);
    c.add(scr->sta);
    c.add(scr->rtb);
    c.add(scr->rsd1);
    c.add(sums->L);
    scr->rgb = c;
  }
}
// after the H sum: C = c' + MSM_H -> proof
__global__ void __launch_bounds__(64) k_fin_tab_c(const ProofSums* sums, const FinScratch* scr, uint8_t* proj) {
  if (threadIdx.x != 0) return;
  G1XYZZ29 c = scr->rgb;
  c.add(sums->H);
  *reinterpret_cast<XYZZ<Fq>*>(proj + FIN_PROJ_C) = xyzz_to_mont256<Fq>(c);
}
// after the B2 sum: B = b' + MSM_B2 -> proof
__global__ void __launch_bounds__(64) k_fin_tab_b(const ProofSums* sums, const FinScratch* scr, uint8_t* proj) {
  if (threadIdx.x != 0) return;
  G2XYZZ29 b = scr->sd2;
  b.add(sums->B2);
  *reinterpret_cast<XYZZ<Fq2>*>(proj + FIN_PROJ_B) = xyzz_to_mont256<Fq2>(b);
}

__device__ __forceinline__ G1XYZZ29* g1_sum_slot(ProofSums* s, int b) {
  switch (b) {
    case 0: return &s->A;
    case 1: return &s->B1;
    case 2: return &s->L;
    case 3: return &s->H;
    case 4: return &s->sA;
    default: return &s->rB1;
  }
}
__device__ __forceinline__ int g1_sum_offset(int b) {  // byte offset inside a partial record
  const int off[6] = {0, 128, 512, 640, 768, 896};
  return off[b];
}

// A partial record carries the sums in XYZZ (projective) storage form: no field inversion sits
// between a rank's last MSM and the all-gather.  blocks 0..5: the six G1 sums, block 6: B2;
// lanes 0..3 convert one coordinate each.
template <class F>
__device__ __forceinline__ void store_xyzz(uint8_t* out, const XYZZ29<typename Lazy<F>::type>& p, int coord) {
  F* dst = reinterpret_cast<F*>(out) + coord;
  if (p.is_inf()) {
    *dst = F::zero();
    return;
  }
  const auto& c = coord == 0 ? p.x : (coord == 1 ? p.y : (coord == 2 ? p.zz : p.zzz));
  *dst = c.to_mont256();
}
template <class F>
__device__ __forceinline__ XYZZ29<typename Lazy<F>::type> load_xyzz(const uint8_t* in) {
  using LF = typename Lazy<F>::type;
  const F* src = reinterpret_cast<const F*>(in);
  if (src[2].is_zero()) return XYZZ29<LF>::infinity();
  return XYZZ29<LF>{LF::from_mont256(src[0]), LF::from_mont256(src[1]), LF::from_mont256(src[2]),
                    LF::from_mont256(src[3])};
}

__global__ void __launch_bounds__(64) k_sums_to_partial(ProofSums* sums, uint8_t* out) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= 4) return;
  if (b < 6) store_xyzz<Fq>(out + g1_sum_offset(b), *g1_sum_slot(sums, b), t);
  else store_xyzz<Fq2>(out + 256, sums->B2, t);
}

// affine conversion of two sums for the single-MSM entry points: A -> out[0,64), B2 -> out[128,256)
__global__ void __launch_bounds__(64) k_sums_to_affine(const ProofSums* sums, uint8_t* out) {
  if (threadIdx.x != 0) return;
  if (blockIdx.x == 0) *reinterpret_cast<G1Affine*>(out) = to_storage_affine<Fq>(sums->A);
  else *reinterpret_cast<G2Affine*>(out + 128) = to_storage_affine<Fq2>(sums->B2);
}

// one block per sum; lane k converts rank k's record, then a log2(world)-deep tree (ranks beyond
// 64 are folded in by strided lanes first)
__global__ void __launch_bounds__(64) k_partials_to_sums(const uint8_t* parts, int world,
                                                         ProofSums* sums) {
  G16_DYN_SMEM(smem_raw);
  const int b = blockIdx.x, t = threadIdx.x;
  if (b < 6) {
    G1XYZZ29* sh = reinterpret_cast<G1XYZZ29*>(smem_raw);
    const int off = g1_sum_offset(b);
    G1XYZZ29 acc = G1XYZZ29::infinity();
    for (int k = t; k < world; k += 64) acc.add(load_xyzz<Fq>(parts + (size_t)k * FIN_PARTIAL_BYTES + off));
    sh[t] = acc;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
      if (t < o && t + o < world) {
        G1XYZZ29 a = sh[t];
        a.add(sh[t + o]);
        sh[t] = a;
      }
      __syncthreads();
    }
    if (t == 0) *g1_sum_slot(sums, b) = sh[0];
  } else {
    G2XYZZ29* sh = reinterpret_cast<G2XYZZ29*>(smem_raw);
    G2XYZZ29 acc = G2XYZZ29::infinity();
    for (int k = t; k < world; k += 64) acc.add(load_xyzz<Fq2>(parts + (size_t)k * FIN_PARTIAL_BYTES + 256));
    sh[t] = acc;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
      if (t < o && t + o < world) {
        G2XYZZ29 a = sh[t];
        a.add(sh[t + o]);
        sh[t] = a;
      }
      __syncthreads();
    }
    if (t == 0) sums->B2 = sh[0];
  }
}

}  // namespace

void fin_build_tables(const KeyHeaderDev* key, FinTables* tab, hipStream_t stream) {
  G16_LAUNCH(k_fin_tables, 4, 64, 0, stream, key, tab);
}
void fin_fixed(const FinTables* tab, const Fr* rs_dev, FinScratch* scr, hipStream_t stream) {
  G16_LAUNCH(k_fin_fixed, 4, FIN_T, FIN_T * sizeof(G2XYZZ29), stream, tab, rs_dev, scr);
}
void fin_var(const KeyHeaderDev* key, const ProofSums* sums, const Fr* rs_dev, FinScratch* scr,
             uint8_t* proof_dev, hipStream_t stream) {
  G16_LAUNCH(k_fin_var, 2, 64, 0, stream, key, sums, rs_dev, scr, proof_dev);
}
void fin_final(const KeyHeaderDev* key, const ProofSums* sums, const FinScratch* scr,
               uint8_t* proof_dev, hipStream_t stream) {
  G16_LAUNCH(k_fin_final, 1, 64, 0, stream, key, sums, scr, proof_dev);
}
void fin_final_proj(const KeyHeaderDev* key, const ProofSums* sums, const FinScratch* scr, uint8_t* proj_dev,
                    hipStream_t stream) {
  G16_LAUNCH(k_fin_final_proj, 1, 64, 0, stream, key, sums, scr, proj_dev);
}
void fin_b(const KeyHeaderDev* key, const ProofSums* sums, const FinScratch* scr, uint8_t* proof_dev,
           hipStream_t stream) {
  G16_LAUNCH(k_fin_b, 1, 64, 0, stream, key, sums, scr, proof_dev);
}
void fin_partial_var(ProofSums* sums, const Fr* rs_dev, hipStream_t stream) {
  G16_LAUNCH(k_fin_partial_var, 2, 64, 0, stream, sums, rs_dev);
}
void fin_fixed_dist(const FinTables* tab, const Fr* rs_dev, FinScratch* scr, hipStream_t stream) {
  G16_LAUNCH(k_fin_fixed_dist, 6, FIN_T, FIN_T * sizeof(G2XYZZ29), stream, tab, rs_dev, scr);
}
void fin_final_dist(const KeyHeaderDev* key, const ProofSums* sums, const FinScratch* scr,
                    uint8_t* proof_dev, hipStream_t stream) {
  G16_LAUNCH(k_fin_final_dist, 3, 64, 0, stream, key, sums, scr, proof_dev);
}
// ---- host side of the table path: x = X / ZZ, y = Y / ZZZ for the three proof elements ----------------
namespace {
struct U256h {
  uint64_t v[4];
};
inline bool is_one(const U256h& a) { return a.v[0] == 1 && !(a.v[1] | a.v[2] | a.v[3]); }
inline bool geq(const U256h& a, const U256h& b) {
  for (int i = 3; i >= 0; --i)
    if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
  return true;
}
inline uint64_t sub_to(U256h& a, const U256h& b) {  // a -= b, returns the borrow
  unsigned __int128 br = 0;
  for (int i = 0; i < 4; ++i) {
    const unsigned __int128 d = (unsigned __int128)a.v[i] - b.v[i] - (uint64_t)br;
    a.v[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  return (uint64_t)br;
}
inline uint64_t add_to(U256h& a, const U256h& b) {  // a += b, returns the carry
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (unsigned __int128)a.v[i] + b.v[i];
    a.v[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
inline void shr1(U256h& a, uint64_t top) {
  for (int i = 0; i < 3; ++i) a.v[i] = (a.v[i] >> 1) | (a.v[i + 1] << 63);
  a.v[3] = (a.v[3] >> 1) | (top << 63);
}
// z^-1 as a field element (Montgomery in, Montgomery out); 0 -> 0.  Binary extended Euclid on the stored
// integer (Guide to ECC, Alg. 2.22) -- the same algorithm as f29_inv_vartime, on 64-bit host words.
Fq host_inv(const Fq& z) {
  U256h u, v, x1{{1, 0, 0, 0}}, x2{{0, 0, 0, 0}}, p;
  for (int i = 0; i < 4; ++i) {
    u.v[i] = (uint64_t)z.v[2 * i] | ((uint64_t)z.v[2 * i + 1] << 32);
    p.v[i] = (uint64_t)FqParams::MOD[2 * i] | ((uint64_t)FqParams::MOD[2 * i + 1] << 32);
  }
  if (!(u.v[0] | u.v[1] | u.v[2] | u.v[3])) return Fq::zero();
  v = p;
  auto halve = [&](U256h& x) {  // x / 2 mod p
    uint64_t top = 0;
    if (x.v[0] & 1) top = add_to(x, p);
    shr1(x, top);
  };
  auto sub_mod = [&](U256h& a, const U256h& b) {
    if (sub_to(a, b)) add_to(a, p);
  };
  for (int guard = 0; guard < 2048 && !is_one(u) && !is_one(v); ++guard) {
    while (!(u.v[0] & 1)) {
      shr1(u, 0);
      halve(x1);
    }
    while (!(v.v[0] & 1)) {
      shr1(v, 0);
      halve(x2);
    }
    if (geq(u, v)) {
      sub_to(u, v);
      sub_mod(x1, x2);
    } else {
      sub_to(v, u);
      sub_mod(x2, x1);
    }
  }
  const U256h& r = is_one(u) ? x1 : x2;  // I = (z R)^-1 as an integer
  U256 c;
  for (int i = 0; i < 4; ++i) {
    c.v[2 * i] = (uint32_t)r.v[i];
    c.v[2 * i + 1] = (uint32_t)(r.v[i] >> 32);
  }
  // from_canonical(I) has the value I = z^-1 / R; times the element of value R (representation R^2)
  return Fq::from_canonical(c) * Fq::r2();
}
Fq2 host_inv(const Fq2& z) {
  const Fq n = host_inv(z.c0.sqr() + z.c1.sqr());
  return Fq2{z.c0 * n, (z.c1 * n).neg()};
}
template <class F>
Affine<F> host_affine(const XYZZ<F>& a) {
  if (a.is_inf()) return Affine<F>::infinity();
  const F iz3 = host_inv(a.zzz);
  const F iz2 = iz3.sqr() * a.zz.sqr();  // 1 / zz = zz^2 / zzz^2
  return Affine<F>{a.x * iz2, a.y * iz3};
}
}  // namespace

void fin_tab_host_affine(const uint8_t* proj, uint8_t* proof) {
  XYZZ<Fq> a, c;
  XYZZ<Fq2> b;
  memcpy((void*)&a, proj + FIN_PROJ_A, sizeof a);
  memcpy((void*)&b, proj + FIN_PROJ_B, sizeof b);
  memcpy((void*)&c, proj + FIN_PROJ_C, sizeof c);
  const G1Affine pa = host_affine<Fq>(a), pc = host_affine<Fq>(c);
  const G2Affine pb = host_affine<Fq2>(b);
  memcpy(proof, (const void*)&pa, 64);
  memcpy(proof + 64, (const void*)&pb, 128);
  memcpy(proof + 192, (const void*)&pc, 64);
}

void fin_host_affine_c(const uint8_t* proj, uint8_t* proof) {
  XYZZ<Fq> c;
  memcpy((void*)&c, proj + FIN_PROJ_C, sizeof c);
  const G1Affine pc = host_affine<Fq>(c);
  memcpy(proof + 192, (const void*)&pc, 64);
}

void fin_tab_pre(const KeyHeaderDev* key, FinScratch* scr, hipStream_t stream) {
  G16_LAUNCH(k_fin_tab_pre, 2, 64, 0, stream, key, scr);
}
void fin_tab_ac(const ProofSums* sums, FinScratch* scr, uint8_t* proof_dev, hipStream_t stream) {
  G16_LAUNCH(k_fin_tab_ac, 2, 64, 0, stream, sums, scr, proof_dev);
}
void fin_tab_c(const ProofSums* sums, const FinScratch* scr, uint8_t* proof_dev, hipStream_t stream) {
  G16_LAUNCH(k_fin_tab_c, 1, 64, 0, stream, sums, scr, proof_dev);
}
void fin_tab_b(const ProofSums* sums, const FinScratch* scr, uint8_t* proof_dev, hipStream_t stream) {
  G16_LAUNCH(k_fin_tab_b, 1, 64, 0, stream, sums, scr, proof_dev);
}
void sums_to_partial(const ProofSums* sums, uint8_t* partial_dev, hipStream_t stream) {
  G16_LAUNCH(k_sums_to_partial, 7, 64, 0, stream, const_cast<ProofSums*>(sums), partial_dev);
}
void sums_to_affine(const ProofSums* sums, uint8_t* out_dev, hipStream_t stream) {
  G16_LAUNCH(k_sums_to_affine, 2, 64, 0, stream, sums, out_dev);
}
void partials_to_sums(const uint8_t* partials_dev, int world, ProofSums* sums, hipStream_t stream) {
  G16_LAUNCH(k_partials_to_sums, 7, 64, 64 * sizeof(G2XYZZ29), stream, partials_dev, world, sums);
}

}  // namespace g16
