// keygen.hip -- trapdoor (known toxic waste) circom/snarkjs-style Groth16 setup on the GPU.
//
// Not on the proving path: it mints the synthetic proving keys that the BASELINE configurations
// (2^20 .. 2^24 constraints) need, since the reference ships no large .zkey and circom/snarkjs/ptau
// are not available offline.  The math is what
// Groth16::generate_random_parameters_with_reduction::<CircomReduction> computes
// (reference tests/groth16.rs:25 for the call shape; CircomReduction::h_query_scalars,
// src/circom/qap.rs:90-105, for the H basis; SURVEY.md Appendix C.2 for the rest):
//   L_j(tau) = (iNTT of the powers of tau)_j
//   u_i = sum_j A[j][i] L_j (+ L_{m+i} for i < num_inputs), v_i, w_i likewise for B, C
//   A_query[i] = u_i G1, B_query[i] = v_i G1 / G2,
//   IC_i = (beta u_i + alpha v_i + w_i)/gamma G1 (i < num_inputs), L_i = (...)/delta G1 (rest)
//   H_i  = (iNTT over the 2n domain of delta^-1 tau^j, j < 2n-1)[2i+1] G1
// Batched fixed-base scalar multiplication uses an 8-bit windowed table of the generator.
#include "../../include/g16_amd.h"
#include "msm.h"
#include "ntt.h"
#include "witness_map.h"

using namespace g16;

namespace {

struct PowTable {
  Fr p[28];  // base^(2^j)
};

__global__ void __launch_bounds__(256) k_powers(PowTable T, Fr scale, Fr* out, uint32_t count,
                                                uint32_t total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (i >= count) {
    out[i] = Fr::zero();
    return;
  }
  Fr r = scale;
  for (int j = 0; j < 28; ++j)
    if ((i >> j) & 1) r = r * T.p[j];
  out[i] = r;
}

// out[i] = row i of M times x, storage form.  One thread per SHORT row; the long ones (the constant
// wire's column of a transposed circom matrix holds a term per constraint that uses a constant: 10^5 ..
// 10^6 terms, which one thread walked for 0.9 s on the 2^20 Poseidon chain) go through the row classes of
// spmv.h.
struct FrOut {
  Fr* out;
  __device__ __forceinline__ void put(uint32_t i, const Fr29* v) const { out[i] = v[0].to_mont256(); }
};
__global__ void __launch_bounds__(256) k_spmv(SpmvDev M, const Fr* x, FrOut out, uint32_t rows) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  if (!spmv_row_is_short(M, i)) return;
  const Fr29 v = spmv_row_thread(M, x, i);
  out.put(i, &v);
}

// k[i] = (beta u + alpha v + w) * (i < num_inputs ? 1/gamma : 1/delta)
__global__ void __launch_bounds__(256) k_lin_scalars(const Fr* u, const Fr* v, const Fr* w, Fr alpha,
                                                     Fr beta, Fr ginv, Fr dinv, uint32_t num_inputs,
                                                     uint32_t n, Fr* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr t = beta * u[i] + alpha * v[i] + w[i];
  out[i] = t * (i < num_inputs ? ginv : dinv);
}

// hk[i] = S_natural[2i+1] where S is held bit-reversed (output of the DIF inverse NTT, size 2^k2)
__global__ void __launch_bounds__(256) k_gather_odd(const Fr* s_br, int k2, Fr* hk, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = 2 * i + 1;
  hk[i] = s_br[__brev(j) >> (32 - k2)];
}

constexpr int FB_WIN = 8, FB_NWIN = 32, FB_ROW = 255;

// table[w][d-1] = d * 2^(8w) * G, affine
template <class F>
__global__ void __launch_bounds__(64) k_fb_table(Affine<F> gen, Affine<F>* table) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= FB_NWIN) return;
  XYZZ<F> b = XYZZ<F>::from_affine(gen);
  for (int i = 0; i < FB_WIN * w; ++i) b.dbl_in_place();
  const Affine<F> base = b.to_affine();
  XYZZ<F> acc = XYZZ<F>::infinity();
  for (int d = 1; d <= FB_ROW; ++d) {
    acc.madd(base);
    table[w * FB_ROW + d - 1] = acc.to_affine();
  }
}

template <class F>
__global__ void __launch_bounds__(128) k_fb_mul(const Affine<F>* __restrict__ table,
                                                const Fr* __restrict__ scalars, uint32_t n,
                                                Affine<F>* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U256 k = scalars[i].to_canonical();
  XYZZ<F> acc = XYZZ<F>::infinity();
#pragma unroll 1
  for (int limb = 0; limb < 8; ++limb) {
    const uint32_t v = k.v[limb];
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
      const uint32_t d = (v >> (8 * b)) & 0xffu;
      if (d) acc.madd(table[(limb * 4 + b) * FB_ROW + d - 1]);
    }
  }
  out[i] = acc.to_affine();
}

Fr fr_from_limbs(const uint64_t* p) {
  Fr a;
  memcpy(a.v, p, 32);
  return a;
}

template <class T>
void download(std::vector<uint8_t>& dst, const T* dev, size_t count) {
  dst.resize(count * sizeof(T) ? count * sizeof(T) : 1);
  if (count) G16_HIP(hipMemcpy(dst.data(), dev, count * sizeof(T), hipMemcpyDeviceToHost));
}

}  // namespace

struct g16_setup {
  uint32_t n_vars = 0, n_public = 0, domain = 0;
  std::vector<uint8_t> a, b1, b2, l, h, ic;
  uint8_t alpha1[64], beta1[64], delta1[64], beta2[128], gamma2[128], delta2[128];
  std::string err;
};

namespace {
thread_local std::string t_setup_err;
}

extern "C" {

g16_status g16_setup_create(int device, const g16_csr* at, const g16_csr* bt, const g16_csr* ct,
                            uint32_t n_vars, uint32_t n_public, uint32_t num_constraints,
                            const uint64_t* toxic, g16_setup** out) {
  return g16_setup_create_ex(device, at, bt, ct, n_vars, n_public, num_constraints, toxic,
                             G16_REDUCTION_CIRCOM, out);
}

g16_status g16_setup_create_ex(int device, const g16_csr* at, const g16_csr* bt, const g16_csr* ct,
                               uint32_t n_vars, uint32_t n_public, uint32_t num_constraints,
                               const uint64_t* toxic, int reduction, g16_setup** out) {
  if (!at || !bt || !ct || !toxic || !out) return G16_ERR_INVALID;
  if (reduction != G16_REDUCTION_CIRCOM && reduction != G16_REDUCTION_LIBSNARK) return G16_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return G16_ERR_NO_DEVICE;
  g16_setup* S = new g16_setup();
  try {
    G16_HIP(hipSetDevice(device));
    hipStream_t s = nullptr;
    const uint32_t num_inputs = n_public + 1, m = num_constraints, N = n_vars;
    int k = 0;
    while (((uint64_t)1 << k) < (uint64_t)m + num_inputs) ++k;
    if (k + 1 > 28) throw std::runtime_error("PolynomialDegreeTooLarge");  // the prover's own limit (qap.rs:63-68): n <= 2^27
    const uint32_t n = 1u << k;
    S->n_vars = N;
    S->n_public = n_public;
    S->domain = n;
    const Fr tau = fr_from_limbs(toxic), alpha = fr_from_limbs(toxic + 4),
             beta = fr_from_limbs(toxic + 8), gamma = fr_from_limbs(toxic + 12),
             delta = fr_from_limbs(toxic + 16);
    const Fr ginv = gamma.inv(), dinv = delta.inv();

    NttPlan plan_n, plan_2n;
    plan_n.build(k);
    plan_2n.build(k + 1);
    PowTable T;
    {
      Fr x = tau;
      for (int j = 0; j < 28; ++j) {
        T.p[j] = x;
        x = x.sqr();
      }
    }
    // ---- L_j(tau), natural order
    DevBuf<Fr> tmp, L;
    tmp.alloc((size_t)2 * n);
    L.alloc(n);
    G16_LAUNCH(k_powers, ceil_div(n, 256), 256, 0, s, T, Fr::one(), tmp.p, n, n);
    ntt_dif(plan_n, tmp.p, n, 1, true, NTT_FUSE_SCALE, s);
    bitrev_copy(tmp.p, L.p, k, s);

    // ---- u, v, w per wire
    auto upload = [&](const g16_csr* c, CsrStore& d) {
      d.rowptr.alloc((size_t)N + 1);
      d.col.alloc(c->nnz ? c->nnz : 1);
      d.val.alloc(c->nnz ? c->nnz : 1);
      G16_HIP(hipMemcpy(d.rowptr.p, c->row_ptr, ((size_t)N + 1) * 4, hipMemcpyHostToDevice));
      if (c->nnz) {
        G16_HIP(hipMemcpy(d.col.p, c->col, c->nnz * 4, hipMemcpyHostToDevice));
        G16_HIP(hipMemcpy(d.val.p, c->coeff, c->nnz * 32, hipMemcpyHostToDevice));
      }
    };
    DevBuf<Fr> uvw, lin, hk;
    uvw.alloc((size_t)3 * N);
    lin.alloc(N);
    hk.alloc(n);
    {
      CsrStore dA, dB, dC;
      upload(at, dA);
      upload(bt, dB);
      upload(ct, dC);
      const uint32_t g = ceil_div(N, 256);
      const g16_csr* hm[3] = {at, bt, ct};
      CsrStore* dm[3] = {&dA, &dB, &dC};
      for (int q = 0; q < 3; ++q) {
        spmv_cook(dm[q]->col.p, dm[q]->val.p, hm[q]->nnz, s);
        SpmvPlan plan;
        const uint32_t* rp = hm[q]->row_ptr;
        plan.build(&rp, 1, N);
        const SpmvMats<1> M{{SpmvDev{dm[q]->rowptr.p, dm[q]->col.p, dm[q]->val.p}}};
        const FrOut out{uvw.p + (size_t)q * N};
        G16_LAUNCH(k_spmv, g, 256, 0, s, M.m[0], (const Fr*)L.p, out, N);
        spmv_run_long<1, FrOut>(plan, M, (const Fr*)L.p, out, s);
        G16_HIP(hipStreamSynchronize(s));  // the plan's buffers are released at the end of this scope
      }
      G16_LAUNCH(k_lin_scalars, g, 256, 0, s, (const Fr*)uvw.p, (const Fr*)(uvw.p + N),
                 (const Fr*)(uvw.p + 2 * (size_t)N), alpha, beta, ginv, dinv, num_inputs, N, lin.p);
      G16_HIP(hipDeviceSynchronize());
    }
    if (reduction == G16_REDUCTION_LIBSNARK) {
      // ---- H scalars of ark-groth16's LibsnarkReduction::h_query_scalars(n - 1, tau, Z(tau), 1/delta):
      // Z(tau)/delta * tau^i for i < n - 1; entry n - 1 is padded with 0 (-> point at infinity)
      const Fr zt = fr_pow_u64(tau, n) - Fr::one();
      G16_LAUNCH(k_powers, ceil_div(n, 256), 256, 0, s, T, zt * dinv, hk.p, n - 1, n);
    } else {
      // ---- H scalars (qap.rs:90-105 with max_power = n - 1)
      G16_LAUNCH(k_powers, ceil_div(2 * n, 256), 256, 0, s, T, dinv, tmp.p, 2 * n - 1, 2 * n);
      ntt_dif(plan_2n, tmp.p, (size_t)2 * n, 1, true, NTT_FUSE_SCALE, s);
      G16_LAUNCH(k_gather_odd, ceil_div(n, 256), 256, 0, s, (const Fr*)tmp.p, k + 1, hk.p, n);
    }

    // ---- fixed-base tables
    G1Affine g1{Fq::one(), Fq::one() + Fq::one()};
    static const uint32_t X0[8] = {0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu};
    static const uint32_t X1[8] = {0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u};
    static const uint32_t Y0[8] = {0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u};
    static const uint32_t Y1[8] = {0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u};
    auto fqc = [](const uint32_t* l) {
      U256 u;
      memcpy(u.v, l, 32);
      return Fq::from_canonical(u);
    };
    G2Affine g2{Fq2{fqc(X0), fqc(X1)}, Fq2{fqc(Y0), fqc(Y1)}};  // reference src/zkey.rs:443-463
    DevBuf<G1Affine> tab1;
    DevBuf<G2Affine> tab2;
    tab1.alloc(FB_NWIN * FB_ROW);
    tab2.alloc(FB_NWIN * FB_ROW);
    G16_LAUNCH((k_fb_table<Fq>), 1, 64, 0, s, g1, tab1.p);
    G16_LAUNCH((k_fb_table<Fq2>), 1, 64, 0, s, g2, tab2.p);

    // ---- batched fixed-base multiplications
    const size_t maxn = N > n ? N : n;
    DevBuf<G1Affine> o1;
    DevBuf<G2Affine> o2;
    o1.alloc(maxn);
    o2.alloc(N);
    auto fb1 = [&](const Fr* sc, uint32_t cnt) {
      if (cnt) G16_LAUNCH((k_fb_mul<Fq>), ceil_div(cnt, 128), 128, 0, s, (const G1Affine*)tab1.p, sc, cnt, o1.p);
      G16_HIP(hipDeviceSynchronize());
    };
    fb1(uvw.p, N);
    download(S->a, o1.p, N);
    fb1(uvw.p + N, N);
    download(S->b1, o1.p, N);
    G16_LAUNCH((k_fb_mul<Fq2>), ceil_div(N, 128), 128, 0, s, (const G2Affine*)tab2.p, (const Fr*)(uvw.p + N), N, o2.p);
    G16_HIP(hipDeviceSynchronize());
    download(S->b2, o2.p, N);
    fb1(lin.p, N);
    {
      std::vector<uint8_t> all;
      download(all, o1.p, N);
      S->ic.assign(all.begin(), all.begin() + (size_t)num_inputs * 64);
      S->l.assign(all.begin() + (size_t)num_inputs * 64, all.end());
      if (S->l.empty()) S->l.resize(1);
    }
    fb1(hk.p, n);
    download(S->h, o1.p, n);
    // header points: alpha, beta, delta in G1; beta, gamma, delta in G2
    {
      Fr hs[4] = {alpha, beta, delta, gamma};
      DevBuf<Fr> dhs;
      dhs.alloc(4);
      G16_HIP(hipMemcpy(dhs.p, hs, sizeof hs, hipMemcpyHostToDevice));
      fb1(dhs.p, 3);
      G1Affine h1[3];
      G16_HIP(hipMemcpy(h1, o1.p, sizeof h1, hipMemcpyDeviceToHost));
      memcpy(S->alpha1, &h1[0], 64);
      memcpy(S->beta1, &h1[1], 64);
      memcpy(S->delta1, &h1[2], 64);
      G16_LAUNCH((k_fb_mul<Fq2>), 1, 128, 0, s, (const G2Affine*)tab2.p, (const Fr*)dhs.p, 4u, o2.p);
      G16_HIP(hipDeviceSynchronize());
      G2Affine h2[4];
      G16_HIP(hipMemcpy(h2, o2.p, sizeof h2, hipMemcpyDeviceToHost));
      memcpy(S->beta2, &h2[1], 128);
      memcpy(S->delta2, &h2[2], 128);
      memcpy(S->gamma2, &h2[3], 128);
    }
    *out = S;
    return G16_OK;
  } catch (const HipError& e) {
    t_setup_err = e.what();
    delete S;
    return G16_ERR_HIP;
  } catch (const std::exception& e) {
    t_setup_err = e.what();
    const bool dom = t_setup_err.find("PolynomialDegreeTooLarge") != std::string::npos;
    delete S;
    return dom ? G16_ERR_DOMAIN_TOO_LARGE : G16_ERR_INTERNAL;
  }
}

void g16_setup_destroy(g16_setup* s) { delete s; }

g16_status g16_setup_key(g16_setup* s, g16_key_desc* key, const uint8_t** ic, uint32_t* ic_count,
                         uint8_t gamma_g2[128]) {
  if (!s || !key) return G16_ERR_INVALID;
  memset(key, 0, sizeof *key);
  key->n_vars = s->n_vars;
  key->n_public = s->n_public;
  key->domain_size = s->domain;
  key->a_query = s->a.data();
  key->b_g1_query = s->b1.data();
  key->b_g2_query = s->b2.data();
  key->l_query = s->l.data();
  key->h_query = s->h.data();
  memcpy(key->alpha_g1, s->alpha1, 64);
  memcpy(key->beta_g1, s->beta1, 64);
  memcpy(key->delta_g1, s->delta1, 64);
  memcpy(key->beta_g2, s->beta2, 128);
  memcpy(key->delta_g2, s->delta2, 128);
  if (ic) *ic = s->ic.data();
  if (ic_count) *ic_count = s->n_public + 1;
  if (gamma_g2) memcpy(gamma_g2, s->gamma2, 128);
  return G16_OK;
}

}  // extern "C"
