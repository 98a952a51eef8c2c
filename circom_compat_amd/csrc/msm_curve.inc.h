// msm_curve.inc.h -- field-generic MSM kernels; instantiated for G1 (Fq) in msm_g1.hip and for
// G2 (Fq2) in msm_g2.hip so the two heavy translation units build in parallel.
//
// Stages (all launched on one stream, no host round trips):
//   k_precompute_planes   (ctx_create only) plane j of point P = 2^(c*D*j) * P, affine
//   k_bucket_accumulate   THE hot kernel: one lane = one equal segment of the sorted entry list,
//                         mixed XYZZ additions of gathered affine points, one partial per bucket touched
//   k_combine_large       hot buckets (> MSM_SMALL_MULTI partials) get their partials tree-summed
//   k_bucket_reduce       sum_b (b+1) * S_b over chunks of MSM_RED_CHUNK buckets (running sums);
//                         S_b = sum of the bucket's <= MSM_SMALL_MULTI partials
//   k_set_sum             tree-sum of the chunk contributions of one bucket set
//   k_horner              only when D > 1: fold the D bucket sets with c doublings in between
#pragma once
#include <stdlib.h>

#include "msm.h"

namespace g16 {

namespace {

constexpr int ACC_THREADS = MSM_ACC_THREADS;
constexpr uint32_t MSM_EXACT_BLOCKS = 256;  // workgroups of the exact kernel behind an optimistic launch

// G16_ACC_FAST=0: the exact in-kernel redo only (A/B partner of the optimistic G1 kernel)
inline bool acc_fast() {
  static const bool v = [] { const char* e = getenv("G16_ACC_FAST"); return !e || atoi(e) != 0; }();
  return v;
}
// (the optimistic kernel for the G2 launch was built and measured in round 3: 13.5-13.7 against
// 12.2-12.3 ms per launch, DESIGN.md section 5 -- G2 keeps the exact kernel, the variant is not compiled)
template <class F>
constexpr bool acc_is_g1() { return sizeof(F) == sizeof(Fq); }
constexpr int COMB_THREADS = 64;
constexpr int SUM_THREADS = 128;

// plane 0 arrives in the storage form (Montgomery-256, as uploaded from the key); every plane is
// left in the packed internal form.  ctx-create only.
template <class F>
__global__ void __launch_bounds__(128) k_precompute_planes(Affine<F>* pts, uint32_t count, int Pn,
                                                           int shift_bits, uint32_t stride,
                                                           uint32_t off) {
  using LF = typename Lazy<F>::type;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  pts += off;
  const Aff29<LF> p0 = affine_from_mont256<F>(pts[(size_t)i * stride]);
  pts[(size_t)i * stride] = store_packed_affine<F>(p0);
  XYZZ29<LF> q = XYZZ29<LF>::from_affine(p0);
  for (int j = 1; j < Pn; ++j) {
    for (int s = 0; s < shift_bits; ++s) q.dbl_in_place();
    pts[((size_t)j * count + i) * stride] = store_packed_affine<F>(q.to_affine());
  }
}

// index of the bucket that holds sorted position `pos`: the g with offset[g] <= pos < offset[g+1]
__device__ __forceinline__ uint32_t bucket_of(const uint32_t* __restrict__ offset, uint32_t nb,
                                              uint32_t pos) {
  uint32_t lo = 0, hi = nb;  // invariant: offset[lo] <= pos < offset[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (offset[mid] <= pos) lo = mid;
    else hi = mid;
  }
  return lo;
}

// per-segment state of the accumulation kernel (plain struct of scalars + registers: no arrays, no
// address-taken locals, so everything stays in VGPRs)
template <class F>
struct AccWay {
  using LF = typename Lazy<F>::type;
  uint32_t seg, pos, end, g, bend, bnext, en_next, en_next2;  // bend = offset[g+1], bnext = offset[g+2]
  bool live;
  XYZZ29<LF> acc;
  Affine<F> raw_next;
};

// G16_DEBUG_GATHER_MASK (measurement only, wrong results): point index &= mask, i.e. every gather
// falls into a cache-resident set of points -- separates the cost of the random HBM gathers from
// the arithmetic of the accumulation kernel
__device__ uint32_t g_gather_mask = 0xffffffffu;

// pts already points at this query's half of an interleaved pair; PS = record stride in points
template <class F, int PS>
// (round 4: non-temporal loads for these single-use gathers -- __builtin_nontemporal_load per 16-byte piece --
// were measured on a variant build: 39.17 / 38.96 / 39.24 against 37.64 / 37.78 / 37.41 ms per 2^22 proof, the G2
// launch 14.3 against 12.7 ms: the pieces of a point stop sharing the line their first miss brought in.  Not kept.)
__device__ __forceinline__ void acc_fetch(const Affine<F>* __restrict__ pts, uint32_t npts,
                                          uint32_t idx_min, uint32_t idx_bits, uint32_t en,
                                          Affine<F>& raw) {
  const uint32_t idx = en & ((1u << idx_bits) - 1u);  // idx_bits is wave-uniform: mask and shift live in SGPRs
  if (idx >= idx_min) {
    const uint32_t plane = (en & 0x7fffffffu) >> idx_bits;
#ifdef G16_DEBUG_GATHER
    raw = pts[((size_t)(plane & g_gather_mask) * npts + ((idx - idx_min) & g_gather_mask)) * PS];
#else
    raw = pts[((size_t)plane * npts + (idx - idx_min)) * PS];
#endif
  } else {
    raw = Affine<F>::infinity();  // entry below this query's range (public inputs of L)
  }
}

template <class F, int PS>
__device__ __forceinline__ void acc_way_init(AccWay<F>& w, uint32_t seg, uint32_t S, uint32_t M,
                                             const Affine<F>* __restrict__ pts, uint32_t npts,
                                             uint32_t idx_min, uint32_t idx_bits,
                                             const uint32_t* __restrict__ entries,
                                             const uint32_t* __restrict__ offset, uint32_t nb) {
  w.seg = seg;
  w.pos = seg * S;
  w.live = w.pos < M;
  w.end = w.live ? (w.pos + S < M ? w.pos + S : M) : w.pos;
  w.acc = XYZZ29<typename Lazy<F>::type>::infinity();
  w.raw_next = Affine<F>::infinity();
  w.g = 0;
  w.bend = w.bnext = 0;
  w.en_next = w.en_next2 = 0;
  if (w.live) {
    w.g = bucket_of(offset, nb, w.pos);
    w.bend = offset[w.g + 1];
    w.bnext = offset[w.g + 2 < nb ? w.g + 2 : nb];
    w.en_next = entries[w.pos];
    w.en_next2 = w.pos + 1 < w.end ? entries[w.pos + 1] : 0u;
    acc_fetch<F, PS>(pts, npts, idx_min, idx_bits, w.en_next, w.raw_next);
  }
}

// consume the prefetched point + bucket-boundary bookkeeping + issue the next fetches.
// Order matters for the memory counter (vmcnt is in order and counts stores too): the point
// prefetched one iteration ago is unpacked FIRST, so that the wait for it sits in front of the
// boundary block; that block's partial store and its look-ahead load of the next bucket end are then
// only waited for one iteration later (at the next point's unpack), when they have long completed.
// With the boundary block first, every wave drained its fresh stores + a dependent offset[] load
// whenever one of its lanes crossed a bucket boundary (~half of the iterations).
template <class F, int PS>
__device__ __forceinline__ Aff29<typename Lazy<F>::type> acc_way_prepare(
    AccWay<F>& w, bool* step, const Affine<F>* __restrict__ pts, uint32_t npts, uint32_t idx_min,
    uint32_t idx_bits, const uint32_t* __restrict__ entries, const uint32_t* __restrict__ offset,
    uint32_t nb, MsmAcc<F>* __restrict__ partial) {
  using LF = typename Lazy<F>::type;
  *step = w.live && w.pos < w.end;
  const uint32_t en = w.en_next;
  Aff29<LF> p = load_packed_affine<F>(w.raw_next);
  if (en >> 31) p.y = p.y.neg().carry();
  if (!*step) p.inf = true;
  if (*step && w.pos == w.bend) {  // crossed into the next non-empty bucket: emit, restart
    partial[w.g + w.seg] = w.acc;
    w.acc = XYZZ29<LF>::infinity();
    ++w.g;
    w.bend = w.bnext;  // fetched during the previous iteration
    while (w.bend == w.pos) {  // runs of empty buckets (rare): chase the array
      ++w.g;
      w.bend = offset[w.g + 1];
    }
  }
  // look-ahead for the next crossing, EVERY iteration (a conditional load would need a copy into
  // the loop-carried register at the end of the boundary block, i.e. a wait for it right there)
  if (*step) w.bnext = offset[w.g + 2 < nb ? w.g + 2 : nb];
  w.en_next = w.en_next2;
  if (*step && w.pos + 2 < w.end) w.en_next2 = entries[w.pos + 2];
  if (*step && w.pos + 1 < w.end) acc_fetch<F, PS>(pts, npts, idx_min, idx_bits, w.en_next, w.raw_next);
  return p;
}

// FAST: the rare addition whose x-coordinates may coincide (2^-23 filter; true coincidences need
// repeated points) is not redone here -- an exact madd() drags an out-of-line call, its scratch frame
// and ~90 VGPRs into the hot kernel -- but appended to a list: (slot of the partial this accumulator
// will be stored to, entry).  k_acc_fixup adds the listed points to the stored partials afterwards.
template <class F, bool FAST>
__device__ __forceinline__ void acc_way_commit(AccWay<F>& w, bool step,
                                               const XYZZ29<typename Lazy<F>::type>& res, bool special,
                                               const Aff29<typename Lazy<F>::type>& p, uint32_t en,
                                               uint32_t half, MsmFixList* fix) {
  if (FAST) {
    if (special) {
      const uint32_t k = atomicAdd(&fix->count, 1u);
      if (k < MSM_FIX_CAP) {
        fix->slot[k] = (w.g + w.seg) | (half << 31);
        fix->entry[k] = en;
      }
    } else {
      w.acc = res;
    }
  } else {
    if (special) w.acc.madd(p);  // x-coordinates may coincide: redo exactly (doubling / cancellation)
    else w.acc = res;
  }
  if (step) ++w.pos;
}

// One lane = one equal segment of the sorted entry list (persistent grid: `lanes` segments over
// gridDim.x * blockDim.x threads, normally one each).  The dependent multiply-add chains of one
// mixed addition cannot keep the half-rate multiplier busy on their own (~11 cycles of latency per
// ~8-cycle issue); two or three waves per SIMD do: the exact variant (219 VGPRs) runs two, the
// optimistic G1 variant (149 VGPRs, FAST below) three -- 3072 x 128 threads = two rounds of resident
// workgroups (msm.h, MSM_ACC_BLOCKS), the G2 kernel one wave per SIMD over 2048 workgroups.  Interleaving
// two segments per lane in one basic block was measured and lost: ~450 VGPRs (1 wave per SIMD),
// 21.4 vs 17.5 ms for the four G1 MSMs of a 2^22 proof; capped at 256 VGPRs it spills.
// PS = record stride in points (2: this query is one half of an interleaved pair,
// MsmPoints::init_pair).  PAIR: both halves in one launch -- the two waves of a workgroup walk the SAME 64 segments, wave h over half h
// of every 128-byte record into partial set h -- identical control flow, so the waves stay within an
// iteration or two of each other and the later one finds the line in the cache.
// FAST (G1): optimistic kernel, see acc_way_commit; !FAST with a list: the exact kernel launched
// behind it, which returns at once unless the list overflowed (fix->overflow, set by k_acc_fixup).
#ifdef G16_ACC_WAVES2  // experiment: cap the kernel at 256 VGPRs (two waves per SIMD for the G2 optimistic variant)
#define G16_ACC_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define G16_ACC_ATTR
#endif
template <class F, int PS, bool PAIR, bool FAST>
__global__ void __launch_bounds__(ACC_THREADS) G16_ACC_ATTR
    k_bucket_accumulate(const Affine<F>* __restrict__ pts, uint32_t npts, uint32_t idx_min,
                        uint32_t idx_bits, const uint32_t* __restrict__ entries,
                        const uint32_t* __restrict__ offset, uint32_t nb, uint32_t lanes,
                        MsmAcc<F>* __restrict__ partial, size_t slot_stride, MsmFixList* fix) {
  using LF = typename Lazy<F>::type;
  if (!FAST && fix && fix->overflow == 0) return;
  const uint32_t M = offset[nb];
  const uint32_t S = msm_seg_len(M, lanes);
  uint32_t nthreads = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t half = 0;
  if (PAIR) {
    half = threadIdx.x / (ACC_THREADS / 2);
    pts += half;
    partial += (size_t)half * slot_stride;
    nthreads /= 2;
    t0 = blockIdx.x * (ACC_THREADS / 2) + threadIdx.x % (ACC_THREADS / 2);
  }
  for (; t0 < lanes; t0 += nthreads) {
    AccWay<F> w;
    acc_way_init<F, PS>(w, t0, S, M, pts, npts, idx_min, idx_bits, entries, offset, nb);
    if (!w.live) break;  // segments are handed out in order: nothing left for later threads either
    for (uint32_t it = 0; it < S; ++it) {
      bool step, special;
      const uint32_t en = w.en_next;  // the entry whose point acc_way_prepare unpacks
      const Aff29<LF> p =
          acc_way_prepare<F, PS>(w, &step, pts, npts, idx_min, idx_bits, entries, offset, nb, partial);
      const XYZZ29<LF> r = XYZZ29<LF>::madd_select(w.acc, p, &special);
      acc_way_commit<F, FAST>(w, step, r, special, p, en, half, fix);
    }
    partial[w.g + w.seg] = w.acc;
  }
}

// Adds the deferred points (MsmFixList) to their stored partials with the exact madd().  One block;
// items that share a slot are taken one per round (the list is short: ~6 false positives of the
// filter per 2^22 MSM).  An overflowed list is left to the exact kernel: overflow = 1.
template <class F, int PS>
__global__ void __launch_bounds__(256)
    k_acc_fixup(const Affine<F>* __restrict__ pts, uint32_t npts, uint32_t idx_min, uint32_t idx_bits,
                MsmFixList* fix, MsmAcc<F>* partial, size_t slot_stride) {
  __shared__ uint32_t sl[MSM_FIX_CAP];
  __shared__ uint32_t done[MSM_FIX_CAP];
  __shared__ uint32_t left;
  const uint32_t n = fix->count;
  __syncthreads();
  if (threadIdx.x == 0) {
    fix->overflow = n > MSM_FIX_CAP ? 1u : 0u;
    fix->count = 0;
  }
  if (n == 0 || n > MSM_FIX_CAP) return;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    sl[i] = fix->slot[i];
    done[i] = 0;
  }
  if (threadIdx.x == 0) left = n;
  __syncthreads();
  while (left != 0) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      if (done[i]) continue;
      bool first = true;  // no earlier pending item on the same slot
      for (uint32_t j = 0; j < i; ++j)
        if (done[j] != 1 && sl[j] == sl[i]) first = false;  // 0: pending, 2: being added in this round
      if (!first) continue;
      const uint32_t en = fix->entry[i], half = sl[i] >> 31, slot = sl[i] & 0x7fffffffu;
      Affine<F> raw;
      acc_fetch<F, PS>(pts + half, npts, idx_min, idx_bits, en, raw);
      Aff29<typename Lazy<F>::type> p = load_packed_affine<F>(raw);
      if (en >> 31) p.y = p.y.neg().carry();
      MsmAcc<F>* dst = partial + (size_t)half * slot_stride + slot;
      MsmAcc<F> a = *dst;
      a.madd(p);
      *dst = a;
      done[i] = 2;  // finished in this round
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t l = 0;
      for (uint32_t i = 0; i < n; ++i) {
        if (done[i] == 2) done[i] = 1;
        if (!done[i]) ++l;
      }
      left = l;
    }
    __syncthreads();
  }
}

// bucket g's partials live in slots [g + first_lane, g + last_lane]
__device__ __forceinline__ void bucket_slots(const uint32_t* __restrict__ offset, uint32_t g,
                                             uint32_t S, uint32_t* first, uint32_t* n) {
  const uint32_t lo = offset[g], hi = offset[g + 1];
  if (hi == lo) {
    *first = 0;
    *n = 0;
    return;
  }
  const uint32_t l0 = lo / S, l1 = (hi - 1) / S;
  *first = g + l0;
  *n = l1 - l0 + 1;
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

// hot buckets (> MSM_SMALL_MULTI partials): a whole workgroup per bucket sums into the first slot
template <class F>
__global__ void __launch_bounds__(COMB_THREADS)
    k_combine_large(const uint32_t* __restrict__ list, const uint32_t* __restrict__ meta,
                    const uint32_t* __restrict__ offset, uint32_t nb, uint32_t lanes,
                    MsmAcc<F>* partial, size_t slot_stride) {
  G16_DYN_SMEM(smem_raw);
  MsmAcc<F>* sh = reinterpret_cast<MsmAcc<F>*>(smem_raw);
  partial += (size_t)blockIdx.y * slot_stride;  // batch of MSMs over the same sort
  const uint32_t n = meta[1];
  const uint32_t S = msm_seg_len(offset[nb], lanes);
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    uint32_t first, nt;
    bucket_slots(offset, list[i], S, &first, &nt);
    MsmAcc<F> acc = MsmAcc<F>::infinity();
    for (uint32_t k = threadIdx.x; k < nt; k += COMB_THREADS) acc.add(partial[first + k]);
    MsmAcc<F> tot = block_sum<F, COMB_THREADS>(acc, sh);
    if (threadIdx.x == 0) partial[first] = tot;
    __syncthreads();
  }
}

// contribution of buckets [lo, lo+L) of one set: sum (b+1) S_b = sum (b-lo+1) S_b + lo * sum S_b
// range != nullptr (bucket-range sharding, MsmSort::range): only the chunks of this rank's run of
// global bucket ids [range[4], range[5]) are reduced -- thread t takes chunk range[4] / red_chunk + t
// (the run starts on a chunk boundary: msm_red_chunk) and the threads past the run exit at once;
// contrib keeps its global layout, k_set_sum reads the same run.
template <class F>
__global__ void __launch_bounds__(64)
    k_bucket_reduce(const MsmAcc<F>* __restrict__ partial, const uint32_t* __restrict__ offset,
                    uint32_t nb, uint32_t lanes, uint32_t B, uint32_t red_chunk,
                    uint32_t chunks_per_set, uint32_t nchunks, MsmAcc<F>* contrib,
                    size_t slot_stride, const uint32_t* __restrict__ range) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (range) {
    const uint32_t q_lo = range[4] / red_chunk, q_hi = (range[5] + red_chunk - 1) / red_chunk;
    if (q >= q_hi - q_lo || q_hi <= q_lo) return;
    q += q_lo;
  }
  if (q >= nchunks) return;
  partial += (size_t)blockIdx.y * slot_stride;
  contrib += (size_t)blockIdx.y * nchunks;
  const uint32_t S = msm_seg_len(offset[nb], lanes);
  const uint32_t set = q / chunks_per_set;
  const uint32_t lo = (q % chunks_per_set) * red_chunk;
  uint32_t hi = lo + red_chunk;
  if (hi > B) hi = B;
  // Running sums over the thread's buckets, highest first: run += every partial of the bucket, then
  // acc += run.  Flattened so that every iteration of every lane is exactly ONE addition (buckets
  // own 1..3 partials: a per-bucket inner loop would run each wave at the maximum over its lanes).
  MsmAcc<F> run = MsmAcc<F>::infinity(), acc = MsmAcc<F>::infinity();
  uint32_t b = hi, first = 0, left = 0;
  bool pending = false;  // the acc += run of bucket b is still to do
  for (;;) {
    if (left == 0 && !pending) {
      if (b == lo) break;
      --b;
      bucket_slots(offset, set * B + b, S, &first, &left);
      if (left > (uint32_t)MSM_SMALL_MULTI) left = 1;  // pre-summed into the first slot by k_combine_large
      pending = true;
    }
    const bool is_slot = left > 0;
    MsmAcc<F> x = is_slot ? run : acc, y = run;
    if (is_slot) y = partial[first + --left];
    x.add(y);  // the one addition of this iteration, operands selected
    if (is_slot) {
      run = x;
    } else {
      acc = x;
      pending = false;
    }
  }
  if (lo != 0 && !run.is_inf()) {
    // acc += lo * run, radix-4 signed digits {-1, 0, 1, 2} of lo, most significant first: the same
    // trip count (nd, from the largest offset of the grid) and the same 2 doublings + 1 addition
    // per digit in every lane of the wave (a binary or NAF ladder adds whenever ANY lane's bit is
    // set, i.e. always)
    uint64_t digits = 0;  // 3 bits per digit: value + 1
    uint32_t v = lo;
    int nd = 0;
    for (uint32_t m = (chunks_per_set - 1) * red_chunk; m != 0 || nd == 0; m >>= 2) ++nd;
    ++nd;  // room for the last carry
    for (int d = 0; d < nd; ++d) {
      uint32_t t = v & 3u;
      v >>= 2;
      if (t == 3u) {
        ++v;  // 3 = 4 - 1
        t = 0u;  // encodes -1
      } else {
        ++t;
      }
      digits |= (uint64_t)t << (3 * d);
    }
    contrib[q] = acc;  // parked in its destination: one point less to keep in registers (Fq2)
    MsmAcc<F> run2 = run;
    run2.dbl_in_place();
    MsmAcc<F> m = MsmAcc<F>::infinity();
    for (int d = nd - 1; d >= 0; --d) {
      m.dbl_in_place();
      m.dbl_in_place();
      const uint32_t t = (uint32_t)(digits >> (3 * d)) & 7u;  // digit + 1
      if (t != 1u) {
        MsmAcc<F> y = t == 3u ? run2 : run;
        if (t == 0u) y = y.neg();
        m.add(y);
      }
    }
    acc = contrib[q];
    acc.add(m);
  }
  contrib[q] = acc;
}

// tree-sum: block (set, blk) of a (sets x nblk) grid sums its slice of the `per_set` inputs of the set
// range != nullptr: only the chunk contributions of the rank's bucket run exist (k_bucket_reduce)
template <class F>
__global__ void __launch_bounds__(SUM_THREADS)
    k_set_sum(const MsmAcc<F>* __restrict__ in, uint32_t per_set, uint32_t nblk, MsmAcc<F>* out,
              size_t in_stride, size_t out_stride, const uint32_t* __restrict__ range,
              uint32_t red_chunk) {
  G16_DYN_SMEM(smem_raw);
  MsmAcc<F>* sh = reinterpret_cast<MsmAcc<F>*>(smem_raw);
  in += (size_t)blockIdx.y * in_stride;
  out += (size_t)blockIdx.y * out_stride;
  const uint32_t set = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const MsmAcc<F>* c = in + (size_t)set * per_set;
  uint32_t k_lo = 0, k_hi = per_set;
  if (range) {
    const uint64_t q_lo = range[4] / red_chunk, q_hi = (range[5] + red_chunk - 1) / red_chunk;
    const uint64_t s_lo = (uint64_t)set * per_set, s_hi = s_lo + per_set;
    const uint64_t a = q_lo > s_lo ? q_lo : s_lo, b = q_hi < s_hi ? q_hi : s_hi;
    k_lo = b > a ? (uint32_t)(a - s_lo) : 0u;
    k_hi = b > a ? (uint32_t)(b - s_lo) : 0u;
  }
  MsmAcc<F> acc = MsmAcc<F>::infinity();
  for (uint32_t k = k_lo + blk * SUM_THREADS + threadIdx.x; k < k_hi; k += nblk * SUM_THREADS)
    acc.add(c[k]);
  MsmAcc<F> tot = block_sum<F, SUM_THREADS>(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// total = sum_d 2^(c*d) wsum[d]   (D == 1: plain copy)
template <class F>
__global__ void k_horner(const MsmAcc<F>* wsum, int stride, int D, int c, MsmAcc<F>* out) {
  if (threadIdx.x != 0) return;
  wsum += (size_t)blockIdx.x * stride;  // one block per MSM of the batch
  out += blockIdx.x;
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
             cfg.c * cfg.D, 1u, 0u);
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
             cfg.c * cfg.D, 1u, 0u);
}

template <class F>
void MsmPoints<F>::init_pair(MsmPoints<F>& a, MsmPoints<F>& b, const Affine<F>* host_a,
                             const Affine<F>* host_b, uint32_t n, const MsmConfig& c,
                             hipStream_t stream) {
  a.cfg = b.cfg = c;
  a.count = b.count = n;
  a.stride = b.stride = 2;
  a.off = 0;
  b.off = 1;
  a.pts.alloc((size_t)2 * c.Pn * (n ? n : 1));
  a.view = nullptr;
  b.view = a.pts.p;
  if (!n) return;
  const size_t rec = sizeof(Affine<F>);
  G16_HIP(hipMemcpy2DAsync(a.pts.p, 2 * rec, host_a, rec, rec, n, hipMemcpyHostToDevice, stream));
  G16_HIP(hipMemcpy2DAsync(a.pts.p + 1, 2 * rec, host_b, rec, rec, n, hipMemcpyHostToDevice, stream));
  for (uint32_t h = 0; h < 2; ++h)
    G16_LAUNCH((k_precompute_planes<F>), ceil_div(n, 128), 128, 0, stream, a.pts.p, n, c.Pn,
               c.c * c.D, 2u, h);
}

template <class F>
void MsmWork<F>::init(uint32_t n_slots, uint32_t n_contrib, int max_sets, int batch_) {
  slots = n_slots ? n_slots : 1;
  ncontrib = n_contrib ? n_contrib : 1;
  sets = max_sets;
  batch = batch_;
  partial.alloc((size_t)batch * slots);
  contrib.alloc((size_t)batch * ncontrib);
  bsum.alloc((size_t)batch * 256 * sets);
  wsum.alloc((size_t)batch * sets);
  fix.alloc((size_t)batch);
  G16_HIP(hipMemset(fix.p, 0, (size_t)batch * sizeof(MsmFixList)));
}

template <class F>
void msm_accumulate(const MsmSort& s, const MsmPoints<F>& P, uint32_t idx_min, MsmWork<F>& work,
                    int slot, hipStream_t stream, StageTimer* tm, bool fixup) {
  const MsmConfig& cfg = s.cfg;
  const uint32_t nb = cfg.nb();
  const int acc_stage = sizeof(F) == sizeof(Fq) ? ST_MSM_ACC_G1 : ST_MSM_ACC_G2;
  if (slot < 0 || slot >= work.batch) throw std::runtime_error("msm_accumulate: bad workspace slot");
  // persistent grid: `lanes` lanes, each owning an equal segment of the sorted entry list
  const uint32_t lanes = s.lanes_of(sizeof(F) != sizeof(Fq));
  const uint32_t grid = lanes / ACC_THREADS;
#ifdef G16_DEBUG_GATHER
  static const bool mask_set = [] {
    if (const char* e = getenv("G16_DEBUG_GATHER_MASK")) {
      const uint32_t m = (uint32_t)strtoul(e, nullptr, 0);
      G16_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gather_mask), &m, sizeof m));
    }
    return true;
  }();
  (void)mask_set;
#endif
  int id = tm ? tm->begin(acc_stage, stream) : -1;
  MsmAcc<F>* out = work.partial.p + (size_t)slot * work.slots;
  const uint32_t* en = (const uint32_t*)s.entries.p;
  const uint32_t* of = (const uint32_t*)s.offset.p;
  if constexpr (acc_is_g1<F>()) {
    if (acc_fast()) {
      // optimistic kernel; the deferred exact additions follow (msm_fixup), here or on the reducing stream
      MsmFixList* fix = work.fix.p + slot;
      if (P.stride == 2)
        G16_LAUNCH((k_bucket_accumulate<F, 2, false, true>), grid, ACC_THREADS, 0, stream, P.data() + P.off,
                   P.count, idx_min, cfg.idx_bits, en, of, nb, lanes, out, (size_t)0, fix);
      else
        G16_LAUNCH((k_bucket_accumulate<F, 1, false, true>), grid, ACC_THREADS, 0, stream, P.data(), P.count,
                   idx_min, cfg.idx_bits, en, of, nb, lanes, out, (size_t)0, fix);
      if (tm) tm->end(id, stream);
      if (fixup) msm_fixup<F>(s, P, idx_min, work, slot, stream, tm);
      return;
    }
  }
  if (P.stride == 2)
    G16_LAUNCH((k_bucket_accumulate<F, 2, false, false>), grid, ACC_THREADS, 0, stream, P.data() + P.off,
               P.count, idx_min, cfg.idx_bits, en, of, nb, lanes, out, (size_t)0, (MsmFixList*)nullptr);
  else
    G16_LAUNCH((k_bucket_accumulate<F, 1, false, false>), grid, ACC_THREADS, 0, stream, P.data(), P.count,
               idx_min, cfg.idx_bits, en, of, nb, lanes, out, (size_t)0, (MsmFixList*)nullptr);
  if (tm) tm->end(id, stream);
}

template <class F>
void msm_fixup(const MsmSort& s, const MsmPoints<F>& P, uint32_t idx_min, MsmWork<F>& work, int slot,
               hipStream_t stream, StageTimer* tm) {
  if constexpr (!acc_is_g1<F>()) {
    return;  // the G2 launch is the exact kernel: nothing was set aside
  } else {
  if (!acc_fast()) return;
  const int tid = tm ? tm->begin(ST_MSM_FIXUP, stream) : -1;
  const MsmConfig& cfg = s.cfg;
  const uint32_t nb = cfg.nb();
  const uint32_t lanes = s.lanes_of(sizeof(F) != sizeof(Fq));
  const uint32_t grid = lanes / ACC_THREADS;
  MsmAcc<F>* out = work.partial.p + (size_t)slot * work.slots;
  const uint32_t* en = (const uint32_t*)s.entries.p;
  const uint32_t* of = (const uint32_t*)s.offset.p;
  MsmFixList* fix = work.fix.p + slot;
  // exact additions of the listed points, then the exact kernel (returns at once unless the list
  // overflowed).  The exact kernel is launched as a SMALL persistent grid (its threads loop over the
  // segments): all of its workgroups have to be placed just to read one word, and with 219 VGPRs
  // they only fit where an accumulation wave has retired -- 6144 workgroups of the pair launch took
  // 0.78 ms to drain through a running G2 accumulation on a sharded rank's `red` stream
  // (profiles/r03_rank8_timeline_k22_points_fixup_inline.txt's successor); a degenerate key that
  // overflows the list pays for the narrow grid, nobody else.
  const uint32_t grid_x = grid < MSM_EXACT_BLOCKS ? grid : MSM_EXACT_BLOCKS;
  if (P.stride == 2) {
    G16_LAUNCH((k_acc_fixup<F, 2>), 1, 256, 0, stream, P.data() + P.off, P.count, idx_min, cfg.idx_bits, fix, out, (size_t)0);
    G16_LAUNCH((k_bucket_accumulate<F, 2, false, false>), grid_x, ACC_THREADS, 0, stream, P.data() + P.off,
               P.count, idx_min, cfg.idx_bits, en, of, nb, lanes, out, (size_t)0, fix);
  } else {
    G16_LAUNCH((k_acc_fixup<F, 1>), 1, 256, 0, stream, P.data(), P.count, idx_min, cfg.idx_bits, fix, out, (size_t)0);
    G16_LAUNCH((k_bucket_accumulate<F, 1, false, false>), grid_x, ACC_THREADS, 0, stream, P.data(), P.count,
               idx_min, cfg.idx_bits, en, of, nb, lanes, out, (size_t)0, fix);
  }
  if (tm) tm->end(tid, stream);
  }
}

template <class F>
void msm_accumulate_pair(const MsmSort& s, const MsmPoints<F>& A, const MsmPoints<F>& B,
                         MsmWork<F>& work, int slot, hipStream_t stream, StageTimer* tm, bool fixup) {
  const MsmConfig& cfg = s.cfg;
  const uint32_t nb = cfg.nb();
  const int acc_stage = ST_MSM_ACC_G1_PAIR;
  if (slot < 0 || slot + 2 > work.batch) throw std::runtime_error("msm_accumulate_pair: bad workspace slot");
  if (A.stride != 2 || B.stride != 2 || A.data() != B.data() || A.off != 0 || B.off != 1 ||
      A.count != B.count)
    throw std::runtime_error("msm_accumulate_pair: the queries are not an interleaved pair");
  // twice the workgroups of a single launch: each one covers 64 segments with both of its waves
  const uint32_t grid = cfg.lanes / (ACC_THREADS / 2);
  int id = tm ? tm->begin(acc_stage, stream) : -1;
  const uint32_t* en = (const uint32_t*)s.entries.p;
  const uint32_t* of = (const uint32_t*)s.offset.p;
  MsmAcc<F>* out = work.partial.p + (size_t)slot * work.slots;
  if constexpr (sizeof(F) == sizeof(Fq)) {
    if (acc_fast()) {
      G16_LAUNCH((k_bucket_accumulate<F, 2, true, true>), grid, ACC_THREADS, 0, stream, A.data(), A.count, 0u,
                 cfg.idx_bits, en, of, nb, cfg.lanes, out, (size_t)work.slots, work.fix.p + slot);
      if (tm) tm->end(id, stream);
      if (fixup) msm_fixup_pair<F>(s, A, B, work, slot, stream, tm);
      return;
    }
  }
  G16_LAUNCH((k_bucket_accumulate<F, 2, true, false>), grid, ACC_THREADS, 0, stream, A.data(), A.count, 0u,
             cfg.idx_bits, en, of, nb, cfg.lanes, out, (size_t)work.slots, (MsmFixList*)nullptr);
  if (tm) tm->end(id, stream);
}

template <class F>
void msm_fixup_pair(const MsmSort& s, const MsmPoints<F>& A, const MsmPoints<F>& B, MsmWork<F>& work,
                    int slot, hipStream_t stream, StageTimer* tm) {
  (void)B;
  if constexpr (sizeof(F) == sizeof(Fq)) {
    if (!acc_fast()) return;
    const int tid = tm ? tm->begin(ST_MSM_FIXUP, stream) : -1;
    const MsmConfig& cfg = s.cfg;
    const uint32_t nb = cfg.nb();
    const uint32_t grid = cfg.lanes / (ACC_THREADS / 2);
    MsmAcc<F>* out = work.partial.p + (size_t)slot * work.slots;
    MsmFixList* fix = work.fix.p + slot;
    const uint32_t grid_x = grid < MSM_EXACT_BLOCKS ? grid : MSM_EXACT_BLOCKS;  // see msm_fixup
    G16_LAUNCH((k_acc_fixup<F, 2>), 1, 256, 0, stream, A.data(), A.count, 0u, cfg.idx_bits, fix, out, (size_t)work.slots);
    G16_LAUNCH((k_bucket_accumulate<F, 2, true, false>), grid_x, ACC_THREADS, 0, stream, A.data(), A.count, 0u,
               cfg.idx_bits, (const uint32_t*)s.entries.p, (const uint32_t*)s.offset.p, nb, cfg.lanes, out,
               (size_t)work.slots, fix);
    if (tm) tm->end(tid, stream);
  }
}

// Bucket reduction of `nbatch` MSMs (workspace slots first_slot ...) that were accumulated over
// the SAME sort: the reduction is a chain of dependent EC additions (~0.3-0.5 ms of latency
// whatever the size), so MSMs sharing a sort pay it once.  out_dev: nbatch consecutive sums.
template <class F>
void msm_reduce(const MsmSort& s, MsmWork<F>& work, int first_slot, int nbatch, MsmAcc<F>* out_dev,
                hipStream_t stream, StageTimer* tm, bool hidden) {
  const MsmConfig& cfg = s.cfg;
  const uint32_t nb = cfg.nb();
  if (first_slot < 0 || nbatch < 1 || first_slot + nbatch > work.batch)
    throw std::runtime_error("msm_reduce: bad workspace slots");
  MsmAcc<F>* partial = work.partial.p + (size_t)first_slot * work.slots;
  int id = tm ? tm->begin(ST_MSM_REDUCE, stream) : -1;
  const bool g2 = sizeof(F) != sizeof(Fq);
  const uint32_t lanes = s.lanes_of(g2);
  G16_LAUNCH((k_combine_large<F>), dim3(1024, nbatch), COMB_THREADS, COMB_THREADS * sizeof(MsmAcc<F>),
             stream, s.large_list(g2), s.large_meta(g2),
             (const uint32_t*)s.offset.p, nb, lanes, partial, (size_t)work.slots);
  // buckets per thread: the running sums are a dependent chain of EC additions (~10 us each on
  // one lane): see msm_red_chunk for the thread count this aims at
  const uint32_t* rng = s.range_dev();  // bucket-range sharding: this rank's run of the bucket set
  const uint32_t red_chunk = msm_red_chunk(cfg, (uint32_t)nbatch, (uint32_t)s.world, hidden);
  const uint32_t cps = ceil_div(cfg.B, red_chunk);
  const uint32_t nchunks = cps * (uint32_t)cfg.D;
  if (nchunks > work.ncontrib) throw std::runtime_error("msm_reduce: contribution buffer too small");
  // the scratch of slot k starts at k * (its per-slot size): reductions of DIFFERENT slots may run on different
  // streams at the same time (round 6: the L reduction of a mid-sized proof runs on the side stream)
  MsmAcc<F>* contrib = work.contrib.p + (size_t)first_slot * work.ncontrib;
  MsmAcc<F>* bsum = work.bsum.p + (size_t)first_slot * 256 * work.sets;
  MsmAcc<F>* wsum = work.wsum.p + (size_t)first_slot * work.sets;
  G16_LAUNCH((k_bucket_reduce<F>), dim3(ceil_div(nchunks, 64), nbatch), 64, 0, stream,
             (const MsmAcc<F>*)partial, (const uint32_t*)s.offset.p, nb, lanes, cfg.B, red_chunk,
             cps, nchunks, contrib, (size_t)work.slots, rng);
  // two-level tree: cps contributions -> nblk block sums -> 1 per set
  uint32_t nblk = ceil_div(cps / (uint32_t)s.world, SUM_THREADS * 2);
  if (nblk > 256) nblk = 256;
  if (nblk < 1) nblk = 1;
  G16_LAUNCH((k_set_sum<F>), dim3((uint32_t)cfg.D * nblk, nbatch), SUM_THREADS,
             SUM_THREADS * sizeof(MsmAcc<F>), stream, (const MsmAcc<F>*)contrib, cps, nblk,
             bsum, (size_t)nchunks, (size_t)256 * work.sets, rng, red_chunk);
  G16_LAUNCH((k_set_sum<F>), dim3((uint32_t)cfg.D, nbatch), SUM_THREADS,
             SUM_THREADS * sizeof(MsmAcc<F>), stream, (const MsmAcc<F>*)bsum, nblk, 1u,
             wsum, (size_t)256 * work.sets, (size_t)work.sets, (const uint32_t*)nullptr, 1u);
  G16_LAUNCH((k_horner<F>), nbatch, 64, 0, stream, (const MsmAcc<F>*)wsum, work.sets, cfg.D,
             cfg.c, out_dev);
  if (tm) tm->end(id, stream);
}

template <class F>
void msm_run(const MsmSort& s, const MsmPoints<F>& P, uint32_t idx_min, MsmWork<F>& work,
             MsmAcc<F>* out_dev, hipStream_t stream, StageTimer* tm) {
  msm_accumulate<F>(s, P, idx_min, work, 0, stream, tm);
  msm_reduce<F>(s, work, 0, 1, out_dev, stream, tm);
}

}  // namespace g16
