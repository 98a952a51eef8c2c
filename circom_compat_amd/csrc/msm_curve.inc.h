// msm_curve.inc.h -- field-generic MSM kernels; instantiated for G1 (Fq) in msm_g1.hip and for
// G2 (Fq2) in msm_g2.hip so the two heavy translation units build in parallel.
//
// Stages (all launched on one stream, no host round trips):
//   k_precompute_planes   (ctx_create only) plane j of point P = 2^(c*D*j) * P, affine
//   k_bucket_accumulate   THE hot kernel: one task = <= cfg.chunk sorted entries of one bucket,
//                         mixed XYZZ additions of gathered affine points
//   k_combine_small/large buckets that were split into several tasks get their partials summed
//   k_bucket_reduce       sum_b (b+1) * S_b over chunks of MSM_RED_CHUNK buckets (running sums)
//   k_set_sum             tree-sum of the chunk contributions of one bucket set
//   k_horner              only when D > 1: fold the D bucket sets with c doublings in between
#pragma once
#include <stdlib.h>

#include "msm.h"

namespace g16 {

namespace {

constexpr int ACC_THREADS = 128;
constexpr int COMB_THREADS = 64;
constexpr int SUM_THREADS = 128;

// plane 0 arrives in the storage form (Montgomery-256, as uploaded from the key); every plane is
// left in the packed internal form.  ctx-create only.
template <class F>
__global__ void __launch_bounds__(128) k_precompute_planes(Affine<F>* pts, uint32_t count, int Pn,
                                                           int shift_bits) {
  using LF = typename Lazy<F>::type;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const Aff29<LF> p0 = affine_from_mont256<F>(pts[i]);
  pts[i] = store_packed_affine<F>(p0);
  XYZZ29<LF> q = XYZZ29<LF>::from_affine(p0);
  for (int j = 1; j < Pn; ++j) {
    for (int s = 0; s < shift_bits; ++s) q.dbl_in_place();
    pts[(size_t)j * count + i] = store_packed_affine<F>(q.to_affine());
  }
}

template <class F>
__global__ void __launch_bounds__(ACC_THREADS)
    k_bucket_accumulate(const Affine<F>* __restrict__ pts, uint32_t npts, uint32_t idx_min,
                        const uint32_t* __restrict__ entries, const uint32_t* __restrict__ offset,
                        const uint32_t* __restrict__ count, const MsmTask* __restrict__ tasks,
                        const uint32_t* __restrict__ ntask_off, uint32_t nb,
                        MsmAcc<F>* __restrict__ partial) {
  using LF = typename Lazy<F>::type;
  const uint32_t total = ntask_off[nb];
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const MsmTask tk = tasks[t];
    // the bucket's cnt entries are split EVENLY over its nt tasks (lengths differ by <= 1), so the
    // lanes of a wave run near-equal trip counts
    const uint32_t cnt = count[tk.g];
    const uint32_t nt = ntask_off[tk.g + 1] - ntask_off[tk.g];
    const uint32_t base = cnt / nt, rem = cnt - base * nt;
    const uint32_t first = tk.k * base + (tk.k < rem ? tk.k : rem);
    const uint32_t len = base + (tk.k < rem ? 1u : 0u);
    const uint32_t* e = entries + offset[tk.g] + first;
    XYZZ29<LF> acc = XYZZ29<LF>::infinity();
    // software pipeline: the gather of entry j+1 (a random 64/128-byte HBM read, ~2 us under load)
    // is in flight while the ~1600 multiply-adds of entry j execute
    auto fetch = [&](uint32_t en, Affine<F>& raw) {
      const uint32_t idx = en & MSM_IDX_MASK;
      if (idx >= idx_min) {
        const uint32_t plane = (en >> MSM_IDX_BITS) & 31u;
        raw = pts[(size_t)plane * npts + (idx - idx_min)];
      } else {
        raw = Affine<F>::infinity();  // entry below this query's range (public inputs of L)
      }
    };
    // entries are read two iterations ahead, points one iteration ahead: neither latency is exposed
    uint32_t en_next = len ? e[0] : 0u;
    uint32_t en_next2 = len > 1 ? e[1] : 0u;
    Affine<F> raw_next = Affine<F>::infinity();
    if (len) fetch(en_next, raw_next);
    for (uint32_t j = 0; j < len; ++j) {
      // consume what the previous iteration fetched, THEN issue the next fetches, THEN compute:
      // all waits happen on loads that have had a whole madd to complete
      const uint32_t en = en_next;
      Aff29<LF> p = load_packed_affine<F>(raw_next);
      if (en >> 31) p.y = p.y.neg().carry();
      en_next = en_next2;
      if (j + 2 < len) en_next2 = e[j + 2];
      if (j + 1 < len) fetch(en_next, raw_next);
      acc.madd(p);
    }
    partial[t] = acc;
  }
}

// buckets split into 2..MSM_SMALL_MULTI tasks: one thread sums the partials into the first slot
template <class F>
__global__ void __launch_bounds__(COMB_THREADS)
    k_combine_small(uint32_t nb, const uint32_t* __restrict__ ntask_off, MsmAcc<F>* partial) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  const uint32_t first = ntask_off[g], nt = ntask_off[g + 1] - first;
  if (nt < 2 || nt > (uint32_t)MSM_SMALL_MULTI) return;
  MsmAcc<F> acc = partial[first];
  for (uint32_t k = 1; k < nt; ++k) acc.add(partial[first + k]);
  partial[first] = acc;
}

// block-wide tree sum through LDS; result valid in thread 0
template <class F, int T>
__device__ __forceinline__ MsmAcc<F> block_sum(MsmAcc<F> v, MsmAcc<F>* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int off = T / 2; off > 0; off >>= 1) {
    if (t < off) {
      MsmAcc<F> a = sh[t];
      a.add(sh[t + off]);
      sh[t] = a;
    }
    __syncthreads();
  }
  MsmAcc<F> r = sh[0];
  __syncthreads();
  return r;
}

// hot buckets (> MSM_SMALL_MULTI partials): a whole workgroup per bucket
template <class F>
__global__ void __launch_bounds__(COMB_THREADS)
    k_combine_large(const uint32_t* __restrict__ list, const uint32_t* __restrict__ meta,
                    const uint32_t* __restrict__ ntask_off, MsmAcc<F>* partial) {
  G16_DYN_SMEM(smem_raw);
  MsmAcc<F>* sh = reinterpret_cast<MsmAcc<F>*>(smem_raw);
  const uint32_t n = meta[1];
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const uint32_t g = list[i];
    const uint32_t first = ntask_off[g], nt = ntask_off[g + 1] - first;
    MsmAcc<F> acc = MsmAcc<F>::infinity();
    for (uint32_t k = threadIdx.x; k < nt; k += COMB_THREADS) acc.add(partial[first + k]);
    MsmAcc<F> tot = block_sum<F, COMB_THREADS>(acc, sh);
    if (threadIdx.x == 0) partial[first] = tot;
    __syncthreads();
  }
}

// contribution of buckets [lo, lo+L) of one set: sum (b+1) S_b = sum (b-lo+1) S_b + lo * sum S_b
template <class F>
__global__ void __launch_bounds__(64)
    k_bucket_reduce(const MsmAcc<F>* __restrict__ partial, const uint32_t* __restrict__ ntask_off,
                    uint32_t B, uint32_t chunks_per_set, uint32_t nchunks, MsmAcc<F>* contrib) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nchunks) return;
  const uint32_t set = q / chunks_per_set;
  const uint32_t lo = (q % chunks_per_set) * (uint32_t)MSM_RED_CHUNK;
  uint32_t hi = lo + MSM_RED_CHUNK;
  if (hi > B) hi = B;
  MsmAcc<F> run = MsmAcc<F>::infinity(), acc = MsmAcc<F>::infinity();
  for (uint32_t b = hi; b-- > lo;) {
    const uint32_t g = set * B + b;
    const uint32_t first = ntask_off[g];
    if (ntask_off[g + 1] != first) run.add(partial[first]);
    acc.add(run);
  }
  if (lo != 0 && !run.is_inf()) {
    MsmAcc<F> m = MsmAcc<F>::infinity();
    for (int bit = 31 - __clz(lo); bit >= 0; --bit) {
      m.dbl_in_place();
      if ((lo >> bit) & 1) m.add(run);
    }
    acc.add(m);
  }
  contrib[q] = acc;
}

// tree-sum: block (set, blk) of a (sets x nblk) grid sums its slice of the `per_set` inputs of the set
template <class F>
__global__ void __launch_bounds__(SUM_THREADS)
    k_set_sum(const MsmAcc<F>* __restrict__ in, uint32_t per_set, uint32_t nblk, MsmAcc<F>* out) {
  G16_DYN_SMEM(smem_raw);
  MsmAcc<F>* sh = reinterpret_cast<MsmAcc<F>*>(smem_raw);
  const uint32_t set = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const MsmAcc<F>* c = in + (size_t)set * per_set;
  MsmAcc<F> acc = MsmAcc<F>::infinity();
  for (uint32_t k = blk * SUM_THREADS + threadIdx.x; k < per_set; k += nblk * SUM_THREADS)
    acc.add(c[k]);
  MsmAcc<F> tot = block_sum<F, SUM_THREADS>(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// total = sum_d 2^(c*d) wsum[d]   (D == 1: plain copy)
template <class F>
__global__ void k_horner(const MsmAcc<F>* wsum, int D, int c, MsmAcc<F>* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  MsmAcc<F> t = wsum[D - 1];
  for (int d = D - 2; d >= 0; --d) {
    for (int s = 0; s < c; ++s) t.dbl_in_place();
    t.add(wsum[d]);
  }
  out[0] = t;
}

}  // namespace

template <class F>
void MsmPoints<F>::init_from_device(const Affine<F>* dev_points, uint32_t n, const MsmConfig& c,
                                    hipStream_t stream) {
  cfg = c;
  count = n;
  pts.alloc((size_t)cfg.Pn * (n ? n : 1));
  if (!n) return;
  G16_HIP(hipMemcpyAsync(pts.p, dev_points, (size_t)n * sizeof(Affine<F>), hipMemcpyDeviceToDevice,
                         stream));
  // always: plane 0 is converted from the storage form to the packed internal form
  G16_LAUNCH((k_precompute_planes<F>), ceil_div(n, 128), 128, 0, stream, pts.p, n, cfg.Pn,
             cfg.c * cfg.D);
}

template <class F>
void MsmPoints<F>::init(const Affine<F>* host_points, uint32_t n, const MsmConfig& c,
                        hipStream_t stream) {
  cfg = c;
  count = n;
  pts.alloc((size_t)cfg.Pn * (n ? n : 1));
  if (!n) return;
  G16_HIP(hipMemcpyAsync(pts.p, host_points, (size_t)n * sizeof(Affine<F>), hipMemcpyHostToDevice,
                         stream));
  // always: plane 0 is converted from the storage form to the packed internal form
  G16_LAUNCH((k_precompute_planes<F>), ceil_div(n, 128), 128, 0, stream, pts.p, n, cfg.Pn,
             cfg.c * cfg.D);
}

template <class F>
void MsmWork<F>::init(uint32_t max_tasks, uint32_t n_contrib, int max_sets) {
  partial.alloc(max_tasks ? max_tasks : 1);
  contrib.alloc(n_contrib ? n_contrib : 1);
  bsum.alloc((size_t)256 * max_sets);
  wsum.alloc(max_sets);
}

template <class F>
void msm_run(const MsmSort& s, const MsmPoints<F>& P, uint32_t idx_min, MsmWork<F>& work,
             MsmAcc<F>* out_dev, hipStream_t stream, StageTimer* tm) {
  const MsmConfig& cfg = s.cfg;
  const uint32_t nb = cfg.nb();
  const int acc_stage = sizeof(F) == sizeof(Fq) ? ST_MSM_ACC_G1 : ST_MSM_ACC_G2;
  // enough workgroups to fill 256 CUs several times over; tasks are grid-strided
  uint32_t grid = ceil_div(s.max_tasks, ACC_THREADS);
  static const uint32_t grid_cap = [] {
    const char* e = getenv("G16_ACC_GRID");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? (uint32_t)v : 8192u;
  }();
  if (grid > grid_cap) grid = grid_cap;
  if (grid < 1) grid = 1;
  int id = tm ? tm->begin(acc_stage, stream) : -1;
  G16_LAUNCH((k_bucket_accumulate<F>), grid, ACC_THREADS, 0, stream,
             (const Affine<F>*)P.pts.p, P.count, idx_min, (const uint32_t*)s.entries.p,
             (const uint32_t*)s.offset.p, (const uint32_t*)s.count.p, (const MsmTask*)s.tasks.p,
             (const uint32_t*)s.ntask_off.p, nb, work.partial.p);
  if (tm) tm->end(id, stream);

  id = tm ? tm->begin(ST_MSM_REDUCE, stream) : -1;
  G16_LAUNCH((k_combine_small<F>), ceil_div(nb, COMB_THREADS), COMB_THREADS, 0, stream, nb,
             (const uint32_t*)s.ntask_off.p, work.partial.p);
  G16_LAUNCH((k_combine_large<F>), 1024, COMB_THREADS, COMB_THREADS * sizeof(MsmAcc<F>), stream,
             (const uint32_t*)s.multi_l.p, (const uint32_t*)s.meta.p,
             (const uint32_t*)s.ntask_off.p, work.partial.p);
  const uint32_t cps = ceil_div(cfg.B, MSM_RED_CHUNK);
  const uint32_t nchunks = cps * (uint32_t)cfg.D;
  G16_LAUNCH((k_bucket_reduce<F>), ceil_div(nchunks, 64), 64, 0, stream,
             (const MsmAcc<F>*)work.partial.p, (const uint32_t*)s.ntask_off.p, cfg.B, cps, nchunks,
             work.contrib.p);
  // two-level tree: cps contributions -> nblk block sums -> 1 per set
  uint32_t nblk = ceil_div(cps, SUM_THREADS * 2);
  if (nblk > 256) nblk = 256;
  G16_LAUNCH((k_set_sum<F>), (uint32_t)cfg.D * nblk, SUM_THREADS, SUM_THREADS * sizeof(MsmAcc<F>),
             stream, (const MsmAcc<F>*)work.contrib.p, cps, nblk, work.bsum.p);
  G16_LAUNCH((k_set_sum<F>), (uint32_t)cfg.D, SUM_THREADS, SUM_THREADS * sizeof(MsmAcc<F>), stream,
             (const MsmAcc<F>*)work.bsum.p, nblk, 1u, work.wsum.p);
  G16_LAUNCH((k_horner<F>), 1, 64, 0, stream, (const MsmAcc<F>*)work.wsum.p, cfg.D, cfg.c, out_dev);
  if (tm) tm->end(id, stream);
}

}  // namespace g16
