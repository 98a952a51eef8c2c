// wm_dist.hip -- distributed witness map (see wm_dist.h).  Every kernel here is element-wise or a
// batched local NTT of ntt29.hip; the two exchanges in between belong to the host framework.
#include "wm_dist.h"

namespace g16 {

namespace {

__device__ __forceinline__ Fr29 omega_pow(const Fr* tlo, const Fr* thi, int h1, uint32_t e) {
  const uint32_t l = e & ((1u << h1) - 1u);
  const uint32_t h = e >> h1;
  Fr29 a = Fr29::unpack(tlo[l].v);
  if (h == 0) return a;
  return a * Fr29::unpack(thi[h].v);
}

__device__ __forceinline__ uint32_t brev_bits(uint32_t x, int bits) {
  return bits ? (__brev(x) >> (32 - bits)) : 0u;
}

struct DistGeom {
  uint32_t m, num_inputs, n, n1, n2, c2, r1;
  int k, k1, k2, rank, world, h1;
};

// a, b, c = a o b of row i = i1 n2 + rank c2 + i2l of this rank -> bufA[(v, i2l)][limb][i1]
struct DistAbcOut {
  int32_t* bufA;
  DistGeom G;
  __device__ __forceinline__ void put(uint32_t i, const Fr29* v) const {
    const uint32_t i1 = i / G.n2, i2l = i % G.n2 - (uint32_t)G.rank * G.c2;
    const size_t vs = (size_t)NTT29_LIMBS * G.n1;
    store_planes(bufA + ((size_t)0 * G.c2 + i2l) * vs, G.n1, i1, v[0]);
    store_planes(bufA + ((size_t)1 * G.c2 + i2l) * vs, G.n1, i1, v[1]);
    store_planes(bufA + ((size_t)2 * G.c2 + i2l) * vs, G.n1, i1, v[0] * v[1]);
  }
};

// short rows of this rank and its rows past the matrices (the longer ones: spmv_run_long, spmv.h)
__global__ void __launch_bounds__(256) k_dist_spmv(SpmvDev A, SpmvDev B, const Fr* w, DistGeom G,
                                                   DistAbcOut out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G.c2 * G.n1) return;
  const uint32_t i2l = t / G.n1, i1 = t % G.n1;
  const uint32_t i = i1 * G.n2 + (uint32_t)G.rank * G.c2 + i2l;
  Fr29 v[2] = {Fr29::zero(), Fr29::zero()};
  if (i < G.m) {
    if (!spmv_row_is_short(A, i) || !spmv_row_is_short(B, i)) return;
    v[0] = spmv_row_thread(A, w, i);
    v[1] = spmv_row_thread(B, w, i);
  } else if (i < G.m + G.num_inputs) {
    v[0] = Fr29::from_mont256(w[i - G.m]);  // qap.rs:46-50
  }
  out.put(i, v);
}

// bufA[(v, i2l)][.][p] * omega_n^(-i2 j1)  ->  send[d][v][pl][limb][i2l],  p = d r1 + pl, j1 = bitrev(p)
__global__ void __launch_bounds__(256) k_dist_pack1(const int32_t* bufA, DistGeom G, const Fr* tlo,
                                                    const Fr* thi, int32_t* send) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t per_v = G.c2 * G.n1;
  if (t >= 3 * per_v) return;
  const uint32_t v = t / per_v, rem = t % per_v;
  const uint32_t p = rem / G.c2, i2l = rem % G.c2;  // i2l fastest: coalesced writes
  const size_t vs = (size_t)NTT29_LIMBS * G.n1;
  Fr29 x = load_planes(bufA + ((size_t)v * G.c2 + i2l) * vs, G.n1, p);
  const uint32_t j1 = brev_bits(p, G.k1);
  const uint32_t i2 = (uint32_t)G.rank * G.c2 + i2l;
  const uint32_t e = (uint32_t)(((uint64_t)i2 * j1) & (G.n - 1));
  x = x * omega_pow(tlo, thi, G.h1, e);
  const uint32_t d = p / G.r1, pl = p % G.r1;
  int32_t* dst = send + (((size_t)d * 3 + v) * G.r1 + pl) * NTT29_LIMBS * G.c2;
#pragma unroll
  for (int l = 0; l < NTT29_LIMBS; ++l) dst[(size_t)l * G.c2 + i2l] = x.l[l];
}

// recv[src][v][pl][limb][i2l] -> bufB[(v, pl)][limb][src c2 + i2l]
__global__ void __launch_bounds__(256) k_dist_unpack1(const int32_t* recv, DistGeom G, int32_t* bufB) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)G.world * 3 * G.r1 * NTT29_LIMBS * G.c2;
  if (t >= total) return;
  const uint32_t i2l = t % G.c2;
  size_t q = t / G.c2;
  const uint32_t l = q % NTT29_LIMBS;
  q /= NTT29_LIMBS;
  const uint32_t pl = q % G.r1;
  q /= G.r1;
  const uint32_t v = q % 3, src = (uint32_t)(q / 3);
  bufB[(((size_t)v * G.r1 + pl) * NTT29_LIMBS + l) * G.n2 + (size_t)src * G.c2 + i2l] = recv[t];
}

// coefficient (j1, j2) at bufB[(v, pl)][.][q], q = bitrev(j2):  x 1/n omega_2n^(j1 + n1 j2)
__global__ void __launch_bounds__(256) k_dist_twist(int32_t* bufB, DistGeom G, const Fr* twlo,
                                                    const Fr* twhi) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t per_v = G.r1 * G.n2;
  if (t >= 3 * per_v) return;
  const uint32_t v = t / per_v, rem = t % per_v;
  const uint32_t pl = rem / G.n2, q = rem % G.n2;
  const uint32_t j1 = brev_bits((uint32_t)G.rank * G.r1 + pl, G.k1);
  const uint32_t j2 = brev_bits(q, G.k2);
  const uint32_t c = j1 + G.n1 * j2;  // < n
  int32_t* vec = bufB + ((size_t)v * G.r1 + pl) * NTT29_LIMBS * G.n2;
  Fr29 x = load_planes(vec, G.n2, q);
  x = x * (Fr29::unpack(twlo[c & ((1u << G.h1) - 1u)].v) * Fr29::unpack(twhi[c >> G.h1].v));
  store_planes(vec, G.n2, q, x);
}

// bufB[(v, pl)][.][m2] * omega_n^(j1 m2) -> send[d][v][m2l][limb][pl],  m2 = d c2 + m2l
__global__ void __launch_bounds__(256) k_dist_pack2(const int32_t* bufB, DistGeom G, const Fr* tlo,
                                                    const Fr* thi, int32_t* send) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t per_v = G.r1 * G.n2;
  if (t >= 3 * per_v) return;
  const uint32_t v = t / per_v, rem = t % per_v;
  const uint32_t m2 = rem / G.r1, pl = rem % G.r1;  // pl fastest: coalesced writes
  const int32_t* vec = bufB + ((size_t)v * G.r1 + pl) * NTT29_LIMBS * G.n2;
  Fr29 x = load_planes(vec, G.n2, m2);
  const uint32_t j1 = brev_bits((uint32_t)G.rank * G.r1 + pl, G.k1);
  const uint32_t e = (uint32_t)(((uint64_t)j1 * m2) & (G.n - 1));
  x = x * omega_pow(tlo, thi, G.h1, e);
  const uint32_t d = m2 / G.c2, m2l = m2 % G.c2;
  int32_t* dst = send + (((size_t)d * 3 + v) * G.c2 + m2l) * NTT29_LIMBS * G.r1;
#pragma unroll
  for (int l = 0; l < NTT29_LIMBS; ++l) dst[(size_t)l * G.r1 + pl] = x.l[l];
}

// recv[src][v][m2l][limb][pl] -> bufA[(v, m2l)][limb][src r1 + pl]
__global__ void __launch_bounds__(256) k_dist_unpack2(const int32_t* recv, DistGeom G, int32_t* bufA) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)G.world * 3 * G.c2 * NTT29_LIMBS * G.r1;
  if (t >= total) return;
  const uint32_t pl = t % G.r1;
  size_t q = t / G.r1;
  const uint32_t l = q % NTT29_LIMBS;
  q /= NTT29_LIMBS;
  const uint32_t m2l = q % G.c2;
  q /= G.c2;
  const uint32_t v = q % 3, src = (uint32_t)(q / 3);
  bufA[(((size_t)v * G.c2 + m2l) * NTT29_LIMBS + l) * G.n1 + (size_t)src * G.r1 + pl] = recv[t];
}

// h[m2l n1 + m1] = a b - c as canonical integers
__global__ void __launch_bounds__(256) k_dist_mul_sub(const int32_t* bufA, DistGeom G, U256* h_canon) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G.c2 * G.n1) return;
  const uint32_t m2l = t / G.n1, m1 = t % G.n1;
  const size_t vs = (size_t)NTT29_LIMBS * G.n1;
  const Fr29 one = Fr29::one();
  const Fr29 a = load_planes(bufA + ((size_t)0 * G.c2 + m2l) * vs, G.n1, m1) * one;
  const Fr29 b = load_planes(bufA + ((size_t)1 * G.c2 + m2l) * vs, G.n1, m1);
  const Fr29 c = load_planes(bufA + ((size_t)2 * G.c2 + m2l) * vs, G.n1, m1);
  const Fr29 h = Fr29::mul2(a, b, c.neg(), one);
  f29::L9 uno{};
  uno.v[0] = 1;
  const Fr29 x = (h * Fr29::from_limbs(uno)).canonical();
  U256 u;
  x.pack(u.v);
  h_canon[t] = u;
}

DistGeom geom(const WmDist& d) {
  DistGeom G;
  G.m = d.m;
  G.num_inputs = d.num_inputs;
  G.n = d.n;
  G.n1 = d.n1;
  G.n2 = d.n2;
  G.c2 = d.c2;
  G.r1 = d.r1;
  G.k = d.k;
  G.k1 = d.k1;
  G.k2 = d.k2;
  G.rank = d.rank;
  G.world = d.world;
  G.h1 = d.planN.base.h1;
  return G;
}

}  // namespace

void WmDist::init(const CsrHost& A, const CsrHost& B, uint32_t m_, uint32_t num_inputs_, int rank_,
                  int world_) {
  m = m_;
  num_inputs = num_inputs_;
  rank = rank_;
  world = world_;
  uint64_t need = (uint64_t)m + num_inputs;
  k = 0;
  while (((uint64_t)1 << k) < need) ++k;
  if (k + 1 > 28) throw std::runtime_error("PolynomialDegreeTooLarge");  // qap.rs:31,66
  if (world & (world - 1)) throw std::runtime_error("distributed witness map needs a power-of-two world");
  k1 = k / 2;
  k2 = k - k1;
  n = 1u << k;
  n1 = 1u << k1;
  n2 = 1u << k2;
  if (n1 < (uint32_t)world || n2 < (uint32_t)world)
    throw std::runtime_error("domain too small for a distributed witness map on this many ranks");
  c2 = n2 / world;
  r1 = n1 / world;
  planN.build(k, nullptr, /*want_full_tables=*/false);  // only its two-level omega tables are read
  plan1.build(k1, nullptr);
  plan2.build(k2, nullptr);
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
  spmv_cook(dA.col.p, dA.val.p, A.nnz, nullptr);
  spmv_cook(dB.col.p, dB.val.p, B.nnz, nullptr);
  const uint32_t* rps[2] = {A.rowptr, B.rowptr};
  const uint32_t lo = (uint32_t)rank * c2, hi = lo + c2, mask = n2 - 1;
  spmv.build(rps, 2, m, [=](uint32_t i) { return (i & mask) >= lo && (i & mask) < hi; });
  bufA.alloc(exchange_ints());
  bufB.alloc(exchange_ints());
}

void WmDist::phase1(const Fr* w_dev, int32_t* send, hipStream_t s) {
  const DistGeom G = geom(*this);
  const SpmvMats<2> M{{SpmvDev{dA.rowptr.p, dA.col.p, dA.val.p}, SpmvDev{dB.rowptr.p, dB.col.p, dB.val.p}}};
  const DistAbcOut out{bufA.p, G};
  G16_LAUNCH(k_dist_spmv, ceil_div((uint64_t)c2 * n1, 256), 256, 0, s, M.m[0], M.m[1], w_dev, G, out);
  spmv_run_long<2, DistAbcOut>(spmv, M, w_dev, out, s);
  ntt29_dif(plan1, bufA.p, (size_t)NTT29_LIMBS * n1, 3 * (int)c2, /*inverse=*/true, NTT_FUSE_NONE, s);
  G16_LAUNCH(k_dist_pack1, ceil_div((uint64_t)3 * c2 * n1, 256), 256, 0, s, (const int32_t*)bufA.p, G,
             (const Fr*)planN.tlo[1].p, (const Fr*)planN.thi[1].p, send);
}

void WmDist::phase2(const int32_t* recv, int32_t* send, hipStream_t s) {
  const DistGeom G = geom(*this);
  G16_LAUNCH(k_dist_unpack1, ceil_div(exchange_ints(), 256), 256, 0, s, recv, G, bufB.p);
  ntt29_dif(plan2, bufB.p, (size_t)NTT29_LIMBS * n2, 3 * (int)r1, /*inverse=*/true, NTT_FUSE_NONE, s);
  G16_LAUNCH(k_dist_twist, ceil_div((uint64_t)3 * r1 * n2, 256), 256, 0, s, bufB.p, G,
             (const Fr*)planN.twlo.p, (const Fr*)planN.twhi.p);
  ntt29_dit(plan2, bufB.p, (size_t)NTT29_LIMBS * n2, 3 * (int)r1, s);
  G16_LAUNCH(k_dist_pack2, ceil_div((uint64_t)3 * r1 * n2, 256), 256, 0, s, (const int32_t*)bufB.p, G,
             (const Fr*)planN.tlo[0].p, (const Fr*)planN.thi[0].p, send);
}

void WmDist::phase3(const int32_t* recv, U256* h_canon, hipStream_t s) {
  const DistGeom G = geom(*this);
  G16_LAUNCH(k_dist_unpack2, ceil_div(exchange_ints(), 256), 256, 0, s, recv, G, bufA.p);
  ntt29_dit(plan1, bufA.p, (size_t)NTT29_LIMBS * n1, 3 * (int)c2, s);
  G16_LAUNCH(k_dist_mul_sub, ceil_div((uint64_t)c2 * n1, 256), 256, 0, s, (const int32_t*)bufA.p, G,
             h_canon);
}

}  // namespace g16
