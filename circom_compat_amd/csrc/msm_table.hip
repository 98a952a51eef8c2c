// msm_table.hip -- fixed-base table MSMs for small keys (see msm_table.h).
#include "msm_table.h"

namespace g16 {

namespace {

template <class F>
using LazyT = typename Lazy<F>::type;

// ---- table construction (once per key) -----------------------------------------------------------
// bases[i][j] = 2^(8 j) * P_i, packed affine.  One thread per point: 31 x (8 doublings + one affine
// conversion by the uniform Fermat inversion).
template <class F>
__global__ void __launch_bounds__(64) k_tbl_bases(const Affine<F>* pts, uint32_t count, Affine<F>* bases) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Aff29<LazyT<F>> p = affine_from_mont256<F>(pts[i]);
  for (int j = 0; j < TBL_W; ++j) {
    bases[(size_t)i * TBL_W + j] = store_packed_affine<F>(p);
    if (j + 1 == TBL_W) break;
    XYZZ29<LazyT<F>> a = XYZZ29<LazyT<F>>::from_affine(p);
    for (int d = 0; d < TBL_C; ++d) a.dbl_in_place();
    p = a.template to_affine<false>();
  }
}

// table[(i W + j) E + k - 1] = k * bases[i][j], k = 1..128.  One thread per (point, window): a chain
// of 127 mixed additions, every partial result converted to affine on its own.
template <class F>
__global__ void __launch_bounds__(64) k_tbl_entries(const Affine<F>* bases, uint32_t pairs, Affine<F>* table) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= pairs) return;
  const Aff29<LazyT<F>> b = load_packed_affine<F>(bases[g]);
  Affine<F>* out = table + (size_t)g * TBL_E;
  if (b.inf) {
    for (int k = 0; k < TBL_E; ++k) out[k] = Affine<F>::infinity();
    return;
  }
  out[0] = bases[g];
  XYZZ29<LazyT<F>> acc = XYZZ29<LazyT<F>>::from_affine(b);
  for (int k = 1; k < TBL_E; ++k) {
    acc.madd(b);  // k = 1 -> 2 is a doubling: madd's exact P = Q path
    out[k] = store_packed_affine<F>(acc.template to_affine<false>());
  }
}

// ---- the MSM ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t byte_of(const U256& k, int j) { return (k.v[j >> 2] >> (8 * (j & 3))) & 0xffu; }

template <class A>
__device__ __forceinline__ A block_tree_sum(A v, A* sh, int nthreads) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int off = nthreads / 2; off > 0; off >>= 1) {
    if (t < off) {
      A a = sh[t];
      a.add(sh[t + off]);
      sh[t] = a;
    }
    __syncthreads();
  }
  return sh[0];
}

// grid = (blocks, jobs).  Thread (i, g) of job y: windows 4 g .. 4 g + 3 of scalar i.
template <class F, int NT>
__global__ void __launch_bounds__(NT) k_tbl_msm(TblJobs J, const Fr* rs) {
  G16_DYN_SMEM(smem_raw);
  using X = XYZZ29<LazyT<F>>;
  X* sh = reinterpret_cast<X*>(smem_raw);
  const TblJob job = J.j[blockIdx.y];
  const uint32_t gid = blockIdx.x * NT + threadIdx.x;
  const uint32_t i = gid / TBL_TPP, g = gid % TBL_TPP;
  X acc = X::infinity();
  if (i < job.count) {
    U256 k;
    if (job.canonical) {
      k = reinterpret_cast<const U256*>(job.scalars)[i];
    } else {
      Fr v = reinterpret_cast<const Fr*>(job.scalars)[i];
      if (job.mul) v = v * rs[job.mul - 1];
      k = v.to_canonical();
    }
    // signed 8-bit digits: t = byte + carry; t > 128 -> digit t - 256, carry 1.  The carry into this
    // thread's first window comes from walking the lower bytes (at most 28 trivial steps).
    uint32_t carry = 0;
    for (int j = 0; j < TBL_WPT * (int)g; ++j) carry = (byte_of(k, j) + carry) > 128u ? 1u : 0u;
    const Affine<F>* tab = reinterpret_cast<const Affine<F>*>(job.table) + ((size_t)i * TBL_W + TBL_WPT * g) * TBL_E;
#pragma unroll 1
    for (int q = 0; q < TBL_WPT; ++q) {
      const uint32_t t = byte_of(k, TBL_WPT * g + q) + carry;
      carry = t > 128u ? 1u : 0u;
      const int d = carry ? (int)t - 256 : (int)t;
      if (d != 0) {
        const uint32_t m = (uint32_t)(d < 0 ? -d : d);
        Aff29<LazyT<F>> p = load_packed_affine<F>(tab[(size_t)q * TBL_E + (m - 1)]);
        if (d < 0) p.y = p.y.neg().carry();
        acc.madd(p);
      }
    }
  }
  const X tot = block_tree_sum(acc, sh, NT);
  if (threadIdx.x == 0) reinterpret_cast<X*>(job.partial)[blockIdx.x] = tot;
}

// one block per job: strided sums of the partials, then the tree
template <class F, int NT>
__global__ void __launch_bounds__(NT) k_tbl_final(TblJobs J, uint32_t blocks) {
  G16_DYN_SMEM(smem_raw);
  using X = XYZZ29<LazyT<F>>;
  X* sh = reinterpret_cast<X*>(smem_raw);
  const TblJob job = J.j[blockIdx.x];
  const X* part = reinterpret_cast<const X*>(job.partial);
  X acc = X::infinity();
  for (uint32_t b = threadIdx.x; b < blocks; b += NT) acc.add(part[b]);
  const X tot = block_tree_sum(acc, sh, NT);
  if (threadIdx.x == 0) *reinterpret_cast<X*>(job.sum) = tot;
}

template <class F>
void build_one(const uint8_t* host_pts, uint32_t count, DevBuf<Affine<F>>& table, hipStream_t s) {
  table.alloc((size_t)(count ? count : 1) * TBL_W * TBL_E);
  if (!count) return;
  DevBuf<Affine<F>> pts, bases;
  pts.alloc(count);
  bases.alloc((size_t)count * TBL_W);
  G16_HIP(hipMemcpyAsync(pts.p, host_pts, (size_t)count * sizeof(Affine<F>), hipMemcpyHostToDevice, s));
  G16_LAUNCH((k_tbl_bases<F>), ceil_div(count, 64), 64, 0, s, (const Affine<F>*)pts.p, count, bases.p);
  const uint32_t pairs = count * TBL_W;
  G16_LAUNCH((k_tbl_entries<F>), ceil_div(pairs, 64), 64, 0, s, (const Affine<F>*)bases.p, pairs, table.p);
  G16_HIP(hipStreamSynchronize(s));  // the temporaries die here
}

uint32_t blocks_for(uint32_t count, int nt) { return ceil_div((uint64_t)(count ? count : 1) * TBL_TPP, nt); }

}  // namespace

void TableSet::build(const uint8_t* a, const uint8_t* b1, const uint8_t* b2, const uint8_t* l, const uint8_t* h,
                     uint32_t len_w_, uint32_t l_idx_min_, uint32_t l_cnt_, uint32_t len_h_, hipStream_t s) {
  len_w = len_w_;
  l_idx_min = l_idx_min_;
  l_cnt = l_cnt_;
  len_h = len_h_;
  build_one<Fq>(a, len_w, oA, s);
  build_one<Fq>(b1, len_w, oB1, s);
  build_one<Fq2>(b2, len_w, oB2, s);
  build_one<Fq>(l, l_cnt, oL, s);
  build_one<Fq>(h, len_h, oH, s);
  tA = oA.p;
  tB1 = oB1.p;
  tB2 = oB2.p;
  tL = oL.p;
  tH = oH.p;
  blocks_w = blocks_for(len_w, TBL_BLOCK);
  blocks_w2 = blocks_for(len_w, TBL_BLOCK_G2);
  blocks_h = blocks_for(len_h, TBL_BLOCK);
  part1.alloc((size_t)TBL_MAX_JOBS * blocks_w);
  part2.alloc(blocks_w2);
  partH.alloc(blocks_h);
  active = true;
}

void TableSet::borrow(const TableSet& o) {
  len_w = o.len_w;
  l_idx_min = o.l_idx_min;
  l_cnt = o.l_cnt;
  len_h = o.len_h;
  tA = o.tA;
  tB1 = o.tB1;
  tB2 = o.tB2;
  tL = o.tL;
  tH = o.tH;
  blocks_w = o.blocks_w;
  blocks_w2 = o.blocks_w2;
  blocks_h = o.blocks_h;
  part1.alloc((size_t)TBL_MAX_JOBS * blocks_w);
  part2.alloc(blocks_w2);
  partH.alloc(blocks_h);
  active = o.active;
}

void TableSet::run_g1_witness(const Fr* w1, const Fr* rs_dev, ProofSums* S, hipStream_t s) {
  TblJobs J{};
  J.n = 5;
  auto job = [&](int k, const G1Affine* tab, const Fr* sc, uint32_t cnt, int mul, G1XYZZ29* sum) {
    J.j[k] = TblJob{tab, sc, cnt, 0, mul, part1.p + (size_t)k * blocks_w, sum};
  };
  job(0, tA, w1, len_w, 0, &S->A);
  job(1, tB1, w1, len_w, 0, &S->B1);
  job(2, tL, w1 + l_idx_min, l_cnt, 0, &S->L);
  job(3, tA, w1, len_w, 2, &S->sA);    // s * A   = MSM_A(s w)
  job(4, tB1, w1, len_w, 1, &S->rB1);  // r * B1  = MSM_B1(r w)
  const size_t smem = (size_t)TBL_BLOCK * sizeof(G1XYZZ29);
  G16_LAUNCH((k_tbl_msm<Fq, TBL_BLOCK>), dim3(blocks_w, J.n), TBL_BLOCK, smem, s, J, rs_dev);
  G16_LAUNCH((k_tbl_final<Fq, TBL_BLOCK>), J.n, TBL_BLOCK, smem, s, J, blocks_w);
}

void TableSet::run_g2_witness(const Fr* w1, ProofSums* S, hipStream_t s) {
  TblJobs J{};
  J.n = 1;
  J.j[0] = TblJob{tB2, w1, len_w, 0, 0, part2.p, &S->B2};
  const size_t smem = (size_t)TBL_BLOCK_G2 * sizeof(G2XYZZ29);  // 36 KiB, as the G1 launches
  G16_LAUNCH((k_tbl_msm<Fq2, TBL_BLOCK_G2>), dim3(blocks_w2, 1), TBL_BLOCK_G2, smem, s, J, (const Fr*)nullptr);
  G16_LAUNCH((k_tbl_final<Fq2, TBL_BLOCK_G2>), 1, TBL_BLOCK_G2, smem, s, J, blocks_w2);
}

void TableSet::run_h(const U256* h_canon, ProofSums* S, hipStream_t s) {
  TblJobs J{};
  J.n = 1;
  J.j[0] = TblJob{tH, h_canon, len_h, 1, 0, partH.p, &S->H};
  const size_t smem = (size_t)TBL_BLOCK * sizeof(G1XYZZ29);
  G16_LAUNCH((k_tbl_msm<Fq, TBL_BLOCK>), dim3(blocks_h, 1), TBL_BLOCK, smem, s, J, (const Fr*)nullptr);
  G16_LAUNCH((k_tbl_final<Fq, TBL_BLOCK>), 1, TBL_BLOCK, smem, s, J, blocks_h);
}

}  // namespace g16
