// ntt29.hip -- LDS-staged multi-pass radix-2 NTT over Fr on lazy 29-bit limbs (see ntt29.h).
//
// Index decomposition identical to ntt.hip (validated index-for-index against the oracle's plain
// radix-2 NTT): a pass owning index bits [lo, lo+b) performs, for every residual index, a size
// R = 2^b transform held in LDS, preceded (DIT) or followed (DIF) by the Cooley-Tukey inter-pass
// twiddle omega_{2^(lo+b)}^(c * bitrev_b(rho)).  What changes is the arithmetic: a butterfly is one
// 171-multiply-add lazy product plus 18 + 26 single-cycle limb operations instead of a CIOS
// product with two carried modular add/subs, and global memory holds limb planes (coalesced
// 4-byte runs) instead of 32-byte elements.
//
// Per element per pass: 72 B of HBM traffic, b/2 butterfly products + 1 twiddle product
// (+ 1/2 product per 5 DIF stages for the value normalisation).  Integer-ALU bound.
#include "ntt29.h"

namespace g16 {

namespace {

constexpr int NTT_THREADS = 256;

struct Ntt29Args {
  int32_t* data;
  uint64_t vec_stride;
  uint64_t n;
  int k, lo, b, logT;
  const Fr* loc;  // omega_{2^loc_bits}^(+-j), packed internal
  int loc_shift;
  const Fr* tlo;  // inter-pass twiddle tables (direction already chosen), packed internal
  const Fr* thi;
  int h1;
  const Fr* twlo;  // coset twist tables (twhi carries 1/n)
  const Fr* twhi;
  const Fr* ptab;    // single-level inter-pass twiddles of THIS pass ((kappa << lo) | c), or null
  const Fr* twfull;  // single-level coset twist at the stored position, or null
  int fuse;
  Fr scale;  // 1/n, packed internal
};

__device__ __forceinline__ Fr29 unpack_tw(const Fr& raw) { return Fr29::unpack(raw.v); }

__device__ __forceinline__ Fr29 two_level(const Fr* tlo, const Fr* thi, int h1, uint32_t e) {
  const uint32_t l = e & ((1u << h1) - 1u);
  const uint32_t h = e >> h1;
  Fr29 a = unpack_tw(tlo[l]);
  if (h == 0) return a;
  return a * unpack_tw(thi[h]);
}

template <bool DIT>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt29_pass(Ntt29Args A) {
  G16_DYN_SMEM(smem_raw);
  Fr29* s = reinterpret_cast<Fr29*>(smem_raw);
  const int R = 1 << A.b;
  const int T = 1 << A.logT;
  const int E = R * T;
  Fr29* stw = s + E;  // R/2 local twiddles
  const int tid = threadIdx.x;
  const int hi = A.lo + A.b;
  int32_t* data = A.data + (uint64_t)blockIdx.y * A.vec_stride;

  const uint32_t q = blockIdx.x;
  uint32_t base;
  int rs, cs;
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

  for (int j = tid; j < R / 2; j += NTT_THREADS) stw[j] = unpack_tw(A.loc[(uint32_t)j << A.loc_shift]);

  // ---- load (DIT: inter-pass twiddle on the way in; it also renormalises the values)
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
    Fr29 x = load_planes(data, A.n, gidx);
    if (DIT && A.lo != 0) {
      const uint32_t c = c_base | (uint32_t)t;
      const uint32_t kap = __brev((uint32_t)rho) >> (32 - A.b);
      if (A.ptab) {
        x = x * unpack_tw(A.ptab[(kap << A.lo) | c]);
      } else {
        const uint32_t ex = (c * kap) << (A.k - hi);
        x = x * two_level(A.tlo, A.thi, A.h1, ex);  // ex == 0 multiplies by one(): value back below 2 r
      }
    }
    s[e] = x;
    (void)rho;
  }
  __syncthreads();

  // ---- local radix-2 stages in LDS
  const int half = E / 2;
  for (int st = 0; st < A.b; ++st) {
    const int lm = DIT ? st : (A.b - 1 - st);  // log2 of the half-size
    const int m = 1 << lm;
    const bool renorm = !DIT && ((st + 1) % 6 == 0) && (st + 1 < A.b);
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
      Fr29 x0 = s[a0];
      Fr29 x1 = s[a1];
      if (DIT) {
        if (lm != 0) x1 = x1 * stw[jm << (A.b - 1 - lm)];
        s[a0] = (x0 + x1).carry();
        s[a1] = (x0 - x1).carry();
      } else {
        Fr29 d = x0 - x1;
        if (lm != 0) d = d * stw[jm << (A.b - 1 - lm)];
        else d = d.carry();
        Fr29 sum = (x0 + x1).carry();
        if (renorm) sum = sum * Fr29::one();
        s[a0] = sum;
        s[a1] = d;
      }
    }
    __syncthreads();
  }

  // ---- store (DIF: inter-pass twiddle, the fused 1/n * omega_2n^i twist, or plain one())
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
    Fr29 x = s[e];
    if (!DIT) {
      if (A.lo != 0) {
        const uint32_t c = c_base | (uint32_t)t;
        const uint32_t kap = __brev((uint32_t)rho) >> (32 - A.b);
        if (A.ptab) {
          x = x * unpack_tw(A.ptab[(kap << A.lo) | c]);
        } else {
          const uint32_t ex = (c * kap) << (A.k - hi);
          x = x * two_level(A.tlo, A.thi, A.h1, ex);
        }
      } else if (A.fuse == NTT_FUSE_TWIST_SCALE) {
        if (A.twfull) {
          x = x * unpack_tw(A.twfull[gidx]);
        } else {
          const uint32_t j = A.k ? (__brev(gidx) >> (32 - A.k)) : 0u;
          const uint32_t l = j & ((1u << A.h1) - 1u);
          x = x * (unpack_tw(A.twlo[l]) * unpack_tw(A.twhi[j >> A.h1]));  // twhi carries 1/n
        }
      } else if (A.fuse == NTT_FUSE_SCALE) {
        x = x * unpack_tw(A.scale);
      } else {
        x = x * Fr29::one();
      }
    }
    store_planes(data, A.n, gidx, x);
    (void)rho;
  }
}

// storage-form table -> packed internal form
__global__ void k_table_to_internal(const Fr* in, Fr* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr r;
  Fr29::from_mont256(in[i]).pack_internal(r.v);
  out[i] = r;
}

// single-level inter-pass table of one strided pass: out[(kappa << lo) | c] = omega^((c kappa) << shift)
__global__ void k_build_pass_table(const Fr* tlo, const Fr* thi, int h1, int lo, int shift, uint32_t count,
                                   Fr* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t c = i & ((1u << lo) - 1u), kap = i >> lo;
  Fr r;
  two_level(tlo, thi, h1, (c * kap) << shift).pack_internal(r.v);
  out[i] = r;
}
// single-level coset twist: out[i] = twlo[l] * twhi[h] for j = bitrev_k(i) = h 2^h1 + l
__global__ void k_build_twist_table(const Fr* twlo, const Fr* twhi, int h1, int k, uint32_t count, Fr* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t j = k ? (__brev(i) >> (32 - k)) : 0u;
  Fr r;
  (unpack_tw(twlo[j & ((1u << h1) - 1u)]) * unpack_tw(twhi[j >> h1])).pack_internal(r.v);
  out[i] = r;
}

__global__ void k_to_planes(const Fr* in, int32_t* planes, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_planes(planes, n, i, Fr29::from_mont256(in[i]));
}
__global__ void k_from_planes(const int32_t* planes, Fr* out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = load_planes(planes, n, i).to_mont256();
}
__global__ void k_bitrev_planes(const int32_t* in, int32_t* out, int k) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = 1u << k;
  if (i >= n) return;
  const uint32_t j = k ? (__brev(i) >> (32 - k)) : 0;
  for (int l = 0; l < NTT29_LIMBS; ++l) out[(size_t)l * n + i] = in[(size_t)l * n + j];
}

void run_pass(const Ntt29Plan& P, size_t pass_index, bool dit, bool inverse, int32_t* data,
              size_t vec_stride, int batch, int fuse, hipStream_t stream, const Fr* twlo = nullptr,
              const Fr* twhi = nullptr) {
  const NttPass& ps = P.base.passes[pass_index];
  Ntt29Args A;
  A.data = data;
  A.vec_stride = vec_stride;
  A.n = P.base.n;
  A.k = P.base.k;
  A.lo = ps.lo;
  A.b = ps.b;
  A.logT = ps.logT;
  const int d = inverse ? 1 : 0;
  A.loc = P.loc[d].p;
  A.loc_shift = P.base.loc_bits - ps.b;
  A.tlo = P.tlo[d].p;
  A.thi = P.thi[d].p;
  A.h1 = P.base.h1;
  A.twlo = twlo ? twlo : P.twlo.p;
  A.twhi = twhi ? twhi : P.twhi.p;
  A.ptab = (P.full_tables && ps.lo != 0) ? P.ptab[d][pass_index].p : nullptr;
  A.twfull = (P.full_tables && !twlo) ? P.twfull.p : nullptr;  // a caller's own twist tables stay two-level
  A.fuse = fuse;
  A.scale = P.n_inv_packed;
  const size_t E = (size_t)1 << (ps.b + ps.logT);
  const uint32_t tiles = (uint32_t)(P.base.n / E);
  const size_t smem = (E + ((size_t)1 << ps.b) / 2 + 1) * sizeof(Fr29);
  if (dit)
    G16_LAUNCH((k_ntt29_pass<true>), dim3(tiles, batch), NTT_THREADS, smem, stream, A);
  else
    G16_LAUNCH((k_ntt29_pass<false>), dim3(tiles, batch), NTT_THREADS, smem, stream, A);
}

void convert_table(const DevBuf<Fr>& in, DevBuf<Fr>& out, hipStream_t stream) {
  out.alloc(in.n ? in.n : 1);
  if (in.n)
    G16_LAUNCH(k_table_to_internal, ceil_div(in.n, 256), 256, 0, stream, (const Fr*)in.p, out.p,
               (uint32_t)in.n);
}

}  // namespace

void Ntt29Plan::build(int log_n, hipStream_t stream, bool want_full_tables) {
  base.build(log_n);
  full_tables = false;
  for (int d = 0; d < 2; ++d) {
    convert_table(base.tlo[d], tlo[d], stream);
    convert_table(base.thi[d], thi[d], stream);
    convert_table(base.loc[d], loc[d], stream);
  }
  convert_table(base.twlo, twlo, stream);
  convert_table(base.twhi, twhi, stream);
  Fr29::from_mont256(base.n_inv).pack_internal(n_inv_packed.v);  // host arithmetic
  // G16_NTT_TWO_LEVEL=1 (diagnostic, tested): the two-level tables every plan above 2^24 uses, at any size
  const char* two_level_only = getenv("G16_NTT_TWO_LEVEL");
  if (two_level_only && two_level_only[0] == '1') want_full_tables = false;
  if (want_full_tables && base.k >= 1 && base.k <= NTT29_FULL_TABLE_MAX_LOG && base.passes.size() <= 4) {
    for (int d = 0; d < 2; ++d)
      for (size_t i = 0; i < base.passes.size(); ++i) {
        const NttPass& ps = base.passes[i];
        if (ps.lo == 0) continue;
        const int hi = ps.lo + ps.b;
        const uint32_t count = 1u << hi;
        ptab[d][i].alloc(count);
        G16_LAUNCH(k_build_pass_table, ceil_div(count, 256), 256, 0, stream, (const Fr*)tlo[d].p,
                   (const Fr*)thi[d].p, base.h1, ps.lo, base.k - hi, count, ptab[d][i].p);
      }
    const uint32_t n = (uint32_t)base.n;
    twfull.alloc(n);
    G16_LAUNCH(k_build_twist_table, ceil_div(n, 256), 256, 0, stream, (const Fr*)twlo.p, (const Fr*)twhi.p,
               base.h1, base.k, n, twfull.p);
    full_tables = true;
  }
  G16_HIP(hipStreamSynchronize(stream));
}

void ntt29_dif(const Ntt29Plan& P, int32_t* data, size_t vec_stride, int batch, bool inverse,
               NttFuse fuse, hipStream_t stream, const Fr* twlo, const Fr* twhi) {
  if (P.base.k == 0) return;
  for (size_t i = 0; i < P.base.passes.size(); ++i) {
    const bool last = (i + 1 == P.base.passes.size());
    run_pass(P, i, false, inverse, data, vec_stride, batch, last ? (int)fuse : 0, stream, twlo, twhi);
  }
}

void Ntt29Plan::make_twist(Fr g, Fr scale, DevBuf<Fr>& lo, DevBuf<Fr>& hi) const {
  const size_t nlo = (size_t)1 << base.h1, nhi = (size_t)1 << (base.k - base.h1);
  std::vector<Fr> hlo(nlo), hhi(nhi);
  auto pack = [](const Fr& x) {
    Fr r;
    Fr29::from_mont256(x).pack_internal(r.v);  // host arithmetic
    return r;
  };
  Fr x = Fr::one();
  for (size_t i = 0; i < nlo; ++i) {
    hlo[i] = pack(x);
    x = x * g;
  }
  const Fr step = x;  // g^(2^h1)
  Fr y = scale;
  for (size_t i = 0; i < nhi; ++i) {
    hhi[i] = pack(y);
    y = y * step;
  }
  lo.alloc(nlo);
  hi.alloc(nhi);
  G16_HIP(hipMemcpy(lo.p, hlo.data(), nlo * sizeof(Fr), hipMemcpyHostToDevice));
  G16_HIP(hipMemcpy(hi.p, hhi.data(), nhi * sizeof(Fr), hipMemcpyHostToDevice));
}

void ntt29_dit(const Ntt29Plan& P, int32_t* data, size_t vec_stride, int batch, hipStream_t stream) {
  if (P.base.k == 0) return;
  for (size_t i = P.base.passes.size(); i-- > 0;)
    run_pass(P, i, true, false, data, vec_stride, batch, 0, stream);
}

void ntt29_to_planes(const Fr* in, int32_t* planes, size_t n, hipStream_t stream) {
  G16_LAUNCH(k_to_planes, ceil_div(n, 256), 256, 0, stream, in, planes, (uint64_t)n);
}
void ntt29_from_planes(const int32_t* planes, Fr* out, size_t n, hipStream_t stream) {
  G16_LAUNCH(k_from_planes, ceil_div(n, 256), 256, 0, stream, planes, out, (uint64_t)n);
}
void ntt29_bitrev_planes(const int32_t* in, int32_t* out, int k, hipStream_t stream) {
  const uint32_t n = 1u << k;
  G16_LAUNCH(k_bitrev_planes, ceil_div(n, 256), 256, 0, stream, in, out, k);
}

}  // namespace g16
