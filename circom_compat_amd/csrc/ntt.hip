// ntt.hip -- LDS-staged multi-pass radix-2 NTT over Fr (see ntt.h for the role on the path).
//
// Decomposition (validated index-for-index by tests against the oracle's plain radix-2 NTT):
// the k index bits are split into passes; a pass owning bits [lo, lo+b) performs, for every
// residual index, a size-R=2^b transform held entirely in LDS, preceded (DIT) or followed (DIF) by
// the Cooley-Tukey inter-pass twiddle  omega_{2^(lo+b)}^(c * bitrev_b(rho))  where rho is the row
// position inside the tile and c the low `lo` index bits.  Strided passes (lo > 0) move tiles of
// R rows x T consecutive elements so every HBM access is a T*32-byte contiguous run; the lo == 0
// pass moves fully contiguous 32*R*T-byte tiles.  Local-stage twiddles are staged in LDS once per
// workgroup; inter-pass / coset twiddles come from two 2^(k/2)-entry tables that stay L2-resident.
//
// Cost per element per pass: 64 B of HBM traffic, b/2 butterfly multiplies, <= 2 twiddle
// multiplies.  A 254-bit Montgomery multiply is ~130 v_mad_u64_u32, so the kernel is integer-ALU
// bound, not HBM bound.
#include "ntt.h"

namespace g16 {

namespace {

constexpr int NTT_THREADS = 256;
constexpr int NTT_CONTIG_BITS = 10;  // lo == 0 pass: R*T = 1024 elements (32 KiB) per workgroup
constexpr int NTT_STRIDED_BITS = 6;  // strided passes: R <= 64 rows, T = 1024 / R columns
constexpr int NTT_TILE_ELEMS = 1024;

struct NttPassArgs {
  Fr* data;
  uint64_t stride;
  int k, lo, b, logT;
  const Fr* loc;   // omega_{2^loc_bits}^(+-j)
  int loc_shift;   // loc_bits - b
  const Fr* tlo;   // inter-pass twiddle tables (direction already chosen)
  const Fr* thi;
  int h1;
  const Fr* twlo;  // coset twist tables
  const Fr* twhi;
  int fuse;
  Fr scale;        // 1/n for NTT_FUSE_SCALE
};

__device__ __forceinline__ Fr two_level(const Fr* tlo, const Fr* thi, int h1, uint32_t e) {
  const uint32_t l = e & ((1u << h1) - 1u);
  const uint32_t h = e >> h1;
  Fr a = tlo[l];
  if (h == 0) return a;
  return a * thi[h];
}

template <bool DIT>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_pass(NttPassArgs A) {
  G16_DYN_SMEM(smem_raw);
  Fr* s = reinterpret_cast<Fr*>(smem_raw);
  const int R = 1 << A.b;
  const int T = 1 << A.logT;
  const int E = R * T;
  Fr* stw = s + E;  // R/2 local twiddles
  const int tid = threadIdx.x;
  const int hi = A.lo + A.b;
  Fr* data = A.data + (uint64_t)blockIdx.y * A.stride;

  // tile -> global index mapping
  const uint32_t q = blockIdx.x;
  uint32_t base;  // global index of tile element (rho = 0, t = 0)
  int rs, cs;     // LDS strides of the row / column coordinate
  uint32_t c_base = 0;
  if (A.lo == 0) {
    base = q * (uint32_t)E;
    rs = 1;
    cs = R;
  } else {
    const uint32_t groups = 1u << (A.lo - A.logT);
    const uint32_t hipart = q / groups;
    const uint32_t lowgrp = q % groups;
    c_base = lowgrp << A.logT;
    base = (hipart << hi) | c_base;
    rs = T;
    cs = 1;
  }

  for (int j = tid; j < R / 2; j += NTT_THREADS) stw[j] = A.loc[(uint32_t)j << A.loc_shift];

  // ---- load (DIT: apply the inter-pass twiddle on the way in)
  for (int e = tid; e < E; e += NTT_THREADS) {
    uint32_t gidx;
    int rho, t;
    if (A.lo == 0) {
      gidx = base + e;
      rho = e & (R - 1);
      t = e >> A.b;
    } else {
      t = e & (T - 1);
      rho = e >> A.logT;
      gidx = base | ((uint32_t)rho << A.lo) | (uint32_t)t;
    }
    Fr x = data[gidx];
    if (DIT && A.lo != 0) {
      const uint32_t c = c_base | (uint32_t)t;
      const uint32_t kap = __brev((uint32_t)rho) >> (32 - A.b);
      const uint32_t ex = (c * kap) << (A.k - hi);
      if (ex != 0) x = x * two_level(A.tlo, A.thi, A.h1, ex);
    }
    s[e] = x;
    (void)rs;
  }
  __syncthreads();

  // ---- local radix-2 stages in LDS
  const int half = E / 2;
  for (int st = 0; st < A.b; ++st) {
    const int lm = DIT ? st : (A.b - 1 - st);  // log2 of the half-size m'
    const int m = 1 << lm;
    for (int u = tid; u < half; u += NTT_THREADS) {
      int j, t;
      if (A.lo == 0) {
        j = u & (R / 2 - 1);
        t = u >> (A.b - 1);
      } else {
        t = u & (T - 1);
        j = u >> A.logT;
      }
      const int jm = j & (m - 1);
      const int rho0 = ((j >> lm) << (lm + 1)) | jm;
      const int a0 = rho0 * rs + t * cs;
      const int a1 = a0 + m * rs;
      Fr x0 = s[a0];
      Fr x1 = s[a1];
      if (DIT) {
        if (lm != 0) x1 = x1 * stw[jm << (A.b - 1 - lm)];
        s[a0] = x0 + x1;
        s[a1] = x0 - x1;
      } else {
        Fr d = x0 - x1;
        if (lm != 0) d = d * stw[jm << (A.b - 1 - lm)];
        s[a0] = x0 + x1;
        s[a1] = d;
      }
    }
    __syncthreads();
  }

  // ---- store (DIF: inter-pass twiddle, or the fused 1/n * omega_2n^i twist after the last pass)
  for (int e = tid; e < E; e += NTT_THREADS) {
    uint32_t gidx;
    int rho, t;
    if (A.lo == 0) {
      gidx = base + e;
      rho = e & (R - 1);
      t = e >> A.b;
    } else {
      t = e & (T - 1);
      rho = e >> A.logT;
      gidx = base | ((uint32_t)rho << A.lo) | (uint32_t)t;
    }
    Fr x = s[e];
    if (!DIT) {
      if (A.lo != 0) {
        const uint32_t c = c_base | (uint32_t)t;
        const uint32_t kap = __brev((uint32_t)rho) >> (32 - A.b);
        const uint32_t ex = (c * kap) << (A.k - hi);
        if (ex != 0) x = x * two_level(A.tlo, A.thi, A.h1, ex);
      } else if (A.fuse == NTT_FUSE_TWIST_SCALE) {
        const uint32_t j = __brev(gidx) >> (32 - A.k);
        const uint32_t l = j & ((1u << A.h1) - 1u);
        x = x * (A.twlo[l] * A.twhi[j >> A.h1]);  // twhi[0] = 1/n, so no shortcut for h == 0
      } else if (A.fuse == NTT_FUSE_SCALE) {
        x = x * A.scale;
      }
    }
    data[gidx] = x;
  }
}

__global__ void k_bitrev_copy(const Fr* in, Fr* out, int k) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << k)) return;
  const uint32_t j = k ? (__brev(i) >> (32 - k)) : 0;
  out[i] = in[j];
}

void run_pass(const NttPlan& P, const NttPass& ps, bool dit, bool inverse, Fr* data, size_t stride,
              int batch, int fuse, hipStream_t stream) {
  NttPassArgs A;
  A.data = data;
  A.stride = stride;
  A.k = P.k;
  A.lo = ps.lo;
  A.b = ps.b;
  A.logT = ps.logT;
  const int d = inverse ? 1 : 0;
  A.loc = P.loc[d].p;
  A.loc_shift = P.loc_bits - ps.b;
  A.tlo = P.tlo[d].p;
  A.thi = P.thi[d].p;
  A.h1 = P.h1;
  A.twlo = P.twlo.p;
  A.twhi = P.twhi.p;
  A.fuse = fuse;
  A.scale = P.n_inv;
  const size_t E = (size_t)1 << (ps.b + ps.logT);
  const uint32_t tiles = (uint32_t)(P.n / E);
  const size_t smem = (E + ((size_t)1 << ps.b) / 2 + 1) * sizeof(Fr);
  if (dit)
    G16_LAUNCH((k_ntt_pass<true>), dim3(tiles, batch), NTT_THREADS, smem, stream, A);
  else
    G16_LAUNCH((k_ntt_pass<false>), dim3(tiles, batch), NTT_THREADS, smem, stream, A);
}

}  // namespace

Fr fr_pow_u64(Fr base, uint64_t e) {
  Fr r = Fr::one();
  while (e) {
    if (e & 1) r = r * base;
    base = base.sqr();
    e >>= 1;
  }
  return r;
}

Fr fr_root_of_unity(int log_n) {
  // TWO_ADIC_ROOT = 5^((r-1)/2^28); omega_{2^log_n} = ROOT^(2^(28-log_n))
  static bool init = false;
  static Fr root;
  if (!init) {
    uint32_t e[8];
    for (int i = 0; i < 8; ++i) e[i] = FrParams::MOD[i];
    e[0] -= 1;  // r - 1
    // shift right by 28 bits
    uint32_t s[8];
    for (int i = 0; i < 8; ++i) {
      uint64_t lo = e[i];
      uint64_t hi = (i + 1 < 8) ? e[i + 1] : 0;
      s[i] = (uint32_t)(((hi << 32) | lo) >> 28);
    }
    root = Fr::from_u32(5).pow(s);
    init = true;
  }
  Fr w = root;
  for (int i = log_n; i < 28; ++i) w = w.sqr();
  return w;
}

void NttPlan::build(int log_n) {
  // k = 28 (the whole two-adic subgroup of Fr) serves the key generator's one transform of size 2n at
  // n = 2^27 -- the largest domain the reference accepts; such a plan has no omega_2n coset twist
  // (no 2^29-th root exists) and ntt_dif refuses NTT_FUSE_TWIST_SCALE on it.  The witness map needs the
  // twist and therefore stops at k = 27 (witness_map.hip: PolynomialDegreeTooLarge, as qap.rs:63-68).
  if (log_n < 0 || log_n > 28) throw std::runtime_error("NTT size out of range (Fr has two-adicity 28)");
  k = log_n;
  n = (size_t)1 << k;
  passes.clear();
  if (k <= NTT_CONTIG_BITS) {
    passes.push_back(NttPass{0, k, 0});
  } else {
    const int rem = k - NTT_CONTIG_BITS;
    const int np = (rem + NTT_STRIDED_BITS - 1) / NTT_STRIDED_BITS;
    int hi = k;
    for (int i = 0; i < np; ++i) {
      const int b = rem / np + (i < rem % np ? 1 : 0);
      int logT = 10 - b;  // R*T = NTT_TILE_ELEMS
      if (logT > hi - b) logT = hi - b;
      passes.push_back(NttPass{hi - b, b, logT});
      hi -= b;
    }
    passes.push_back(NttPass{0, NTT_CONTIG_BITS, 0});
  }
  // the lo == 0 pass with small k: pack several segments per workgroup? (k <= 10: one tile, T = 1)
  loc_bits = 1;
  for (auto& p : passes)
    if (p.b > loc_bits) loc_bits = p.b;
  h1 = (k + 1) / 2;
  const size_t nlo = (size_t)1 << h1, nhi = (size_t)1 << (k - h1);

  const Fr w = fr_root_of_unity(k);
  const Fr winv = w.inv();
  const Fr w2n = k < 28 ? fr_root_of_unity(k + 1) : Fr::one();
  Fr nn = Fr::from_u32(1);
  {  // n as a field element
    U256 u;
    for (int i = 0; i < 8; ++i) u.v[i] = 0;
    u.v[k / 32] = 1u << (k % 32);
    nn = Fr::from_canonical(u);
  }
  n_inv = nn.inv();

  std::vector<Fr> hlo(nlo), hhi(nhi);
  auto fill = [&](Fr g, Fr first_hi) {
    Fr x = Fr::one();
    for (size_t i = 0; i < nlo; ++i) {
      hlo[i] = x;
      x = x * g;
    }
    // x == g^(2^h1) now
    Fr step = x;
    Fr y = first_hi;
    for (size_t i = 0; i < nhi; ++i) {
      hhi[i] = y;
      y = y * step;
    }
  };
  for (int d = 0; d < 2; ++d) {
    fill(d ? winv : w, Fr::one());
    tlo[d].alloc(nlo);
    thi[d].alloc(nhi);
    G16_HIP(hipMemcpy(tlo[d].p, hlo.data(), nlo * sizeof(Fr), hipMemcpyHostToDevice));
    G16_HIP(hipMemcpy(thi[d].p, hhi.data(), nhi * sizeof(Fr), hipMemcpyHostToDevice));
  }
  if (k < 28) {
    fill(w2n, n_inv);
    twlo.alloc(nlo);
    twhi.alloc(nhi);
    G16_HIP(hipMemcpy(twlo.p, hlo.data(), nlo * sizeof(Fr), hipMemcpyHostToDevice));
    G16_HIP(hipMemcpy(twhi.p, hhi.data(), nhi * sizeof(Fr), hipMemcpyHostToDevice));
  }

  const size_t nloc = (size_t)1 << (loc_bits - 1);
  std::vector<Fr> hl(nloc);
  for (int d = 0; d < 2; ++d) {
    Fr g = d ? winv : w;  // omega_n^(+-1) -> omega_{2^loc_bits} = g^(2^(k-loc_bits))
    for (int i = loc_bits; i < k; ++i) g = g.sqr();
    Fr x = Fr::one();
    for (size_t i = 0; i < nloc; ++i) {
      hl[i] = x;
      x = x * g;
    }
    loc[d].alloc(nloc);
    G16_HIP(hipMemcpy(loc[d].p, hl.data(), nloc * sizeof(Fr), hipMemcpyHostToDevice));
  }
}

void ntt_dif(const NttPlan& P, Fr* data, size_t stride, int batch, bool inverse, NttFuse fuse,
             hipStream_t stream) {
  if (fuse == NTT_FUSE_TWIST_SCALE && !P.twlo.p) throw std::runtime_error("NTT plan of size 2^28 has no coset twist");
  if (P.k == 0) return;
  for (size_t i = 0; i < P.passes.size(); ++i) {
    const bool last = (i + 1 == P.passes.size());
    run_pass(P, P.passes[i], false, inverse, data, stride, batch, last ? (int)fuse : 0, stream);
  }
}

void ntt_dit(const NttPlan& P, Fr* data, size_t stride, int batch, bool inverse,
             hipStream_t stream) {
  if (P.k == 0) return;
  for (size_t i = P.passes.size(); i-- > 0;)
    run_pass(P, P.passes[i], true, inverse, data, stride, batch, 0, stream);
}

void bitrev_copy(const Fr* in, Fr* out, int k, hipStream_t stream) {
  const uint32_t n = 1u << k;
  G16_LAUNCH(k_bitrev_copy, ceil_div(n, 256), 256, 0, stream, in, out, k);
}

}  // namespace g16
