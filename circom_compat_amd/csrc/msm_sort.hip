// msm_sort.hip -- digit decomposition + counting sort of (bucket, point) pairs for the MSM.
//
// Stands in for ark-ec's make_digits / per-window bucket fill loop (VariableBaseMSM::msm_bigint,
// reached from the call sites cited in msm.h).  One run per *scalar vector*: the witness-based
// queries A, B1, B2, L share one sorted list, the H query has its own.
#include "msm.h"

#include <stdlib.h>

#include <algorithm>

namespace g16 {

MsmConfig msm_make_config(size_t len, int c_override, int planes_override) {
  MsmConfig cfg;
  // window size from a cost model: len * W(c) mixed additions in the bucket kernel plus ~8
  // mixed-addition equivalents per bucket in the reduction.  Windows whose TOP digit has few
  // significant bits (254 - c (W - 1) <= 9: c = 18, 19, 21) are skipped for large inputs: all len
  // top digits land in <= 512 buckets, each spanning more than MSM_SMALL_MULTI lane segments (hot
  // LDS-atomic bins in the sort, a k_combine_large tree per bucket).  Measured sweeps:
  // profiles/r01_window_sweep.txt.
  int c = 3;
  double best = 1e300;
  for (int t = 3; t <= 22; ++t) {
    const int Wt = (255 + t - 1) / t;
    if (len >= ((size_t)1 << 18) && 254 - t * (Wt - 1) <= 9) continue;
    const double cost = (double)len * (double)Wt + 8.0 * (double)((size_t)1 << (t - 1));
    if (cost < best) {
      best = cost;
      c = t;
    }
  }
  // Latency regime (a few thousand to ~2^16 scalars: every stage is a dependent chain, the model's
  // addition counts say nothing): measured on one box at 2^12..2^15 and on the reference bench's
  // 10 000-constraint circuit, c = 15 (top digit 14 bits, no hot top-window buckets) beats the
  // model's c = 11..13 by 4-18 % per proof and every other candidate (scripts/gpu_run22/23.sh).
  // Below that (2^11 proofs) c = 13 wins by 20 %; smaller inputs are within noise of each other.
  // At 2^17 scalars the model's tie between 15 and 16 goes to 16 (3.62 vs 3.80 ms per proof).
  if (len >= 100000 && c < 16) c = 16;
  else if (len >= 2048 && c < 15) c = 15;
  else if (len >= 1024 && c < 13) c = 13;
  if (c_override > 0) c = c_override;
  if (c < 2) c = 2;
  if (c > 24) c = 24;
  cfg.c = c;
  cfg.W = (255 + c - 1) / c;
  int pn = planes_override > 0 ? planes_override : cfg.W;
  if (pn > cfg.W) pn = cfg.W;
  if (pn > 32) pn = 32;  // 5 plane bits in an entry at most
  if (len > ((size_t)1 << MSM_IDX_BITS_NARROW) && pn > 16) pn = 16;  // 27 index bits leave 4 plane bits
  cfg.D = (cfg.W + pn - 1) / pn;
  cfg.Pn = (cfg.W + cfg.D - 1) / cfg.D;
  cfg.B = 1u << (c - 1);
  cfg.idx_bits = cfg.Pn <= 16 ? MSM_IDX_BITS_WIDE : MSM_IDX_BITS_NARROW;
  // G1 segment count: 3072 workgroups (two full rounds of the optimistic kernel's three waves per
  // SIMD) only when a segment still holds ~100 entries; with shorter segments every bucket is cut
  // into more partials than the wider grid is worth (same box: 2^22 proof 37.3 -> 37.2 ms with 3072,
  // 2^20 proof 11.7 -> 12.3 ms)
  if ((uint64_t)len * (uint64_t)cfg.W < (uint64_t)96 * MSM_ACC_BLOCKS * MSM_ACC_THREADS)
    cfg.lanes = (uint32_t)MSM_ACC_BLOCKS_G2 * MSM_ACC_THREADS;
  // G2 segment count: HALF the workgroups (1024 x 128 lanes: exactly one round of the G2 kernel's one wave per
  // SIMD) once a segment still holds >= 256 entries -- fewer partial sums per bucket for the reduction, fewer
  // bucket crossings per lane.  Same box, 2^22, three alternating pairs (profiles/r06_g2_grid_ab.txt):
  // 36.67 / 36.59 / 36.59 ms per proof at 2048 workgroups, 36.04 / 36.15 / 35.98 at 1024; at 2^20 (120 entries
  // per segment at 1024) 11.61 against 11.84: the long segments are what pays.
  if ((uint64_t)len * (uint64_t)cfg.W >= (uint64_t)256 * (MSM_ACC_BLOCKS_G2 / 2) * MSM_ACC_THREADS)
    cfg.lanes2 = (uint32_t)(MSM_ACC_BLOCKS_G2 / 2) * MSM_ACC_THREADS;
  if (const char* e = getenv("G16_ACC_GRID")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 65536) cfg.lanes = (uint32_t)v * MSM_ACC_THREADS;
  }
  if (const char* e = getenv("G16_ACC_GRID_G2")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 65536) cfg.lanes2 = (uint32_t)v * MSM_ACC_THREADS;
  }
  return cfg;
}

namespace {


// Signed c-bit digits of canonical 256-bit scalars, streamed: the eight words are pushed into a
// 64-bit bit buffer in order (static word indices: a runtime index would move the scalar to scratch
// memory, i.e. one or two ~1 us round trips per digit) and a window is cut off whenever c bits are
// there.  c <= 24, so buffer occupancy stays below 24 + 32 bits.  The window schedule depends on c
// only: every lane of the grid runs the same control flow, and several scalars can advance in lock
// step (their LDS ranks / stores overlap).  ~10 instructions per digit; the previous form (select
// the two words of every window with eight compares) cost ~40, and the level-1 passes are bound by
// exactly this instruction count once a rank walks ALL scalars to keep an eighth of the digits.
//   push(k, avail): OR word k of every scalar in flight into its buffer at bit `avail`
//   step(w):        consume window w (the low c bits) of every buffer
template <class Push, class Step>
__device__ __forceinline__ void walk_windows(int c, int W, Push push, Step step) {
  int avail = 0, w = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    push(k, avail);
    avail += 32;
    while (avail >= c && w < W) {
      step(w);
      avail -= c;
      ++w;
    }
  }
  if (w < W) step(w);  // W c >= 255 > 256 - c: at most one window is left, its missing top bits are zero
}

struct DigitState {
  uint64_t buf = 0;
  uint32_t carry = 0;
  // window w of scalar i (the low c bits of buf): false for a zero digit
  __device__ __forceinline__ bool take(int w, uint32_t i, int c, int D, uint32_t B, uint32_t idx_bits,
                                       uint32_t* g, uint32_t* entry) {
    const uint32_t half = 1u << (c - 1);
    const uint32_t raw = ((uint32_t)buf & ((1u << c) - 1u)) + carry;
    buf >>= c;
    uint32_t mag, neg;
    if (raw > half) {
      mag = (1u << c) - raw;
      neg = 1;
      carry = 1;
    } else {
      mag = raw;
      neg = 0;
      carry = 0;
    }
    if (!mag) return false;
    const uint32_t d = (uint32_t)(w % D), j = (uint32_t)(w / D);
    *g = d * B + (mag - 1);
    *entry = i | (j << idx_bits) | (neg << 31);
    return true;
  }
};

// Walks the signed c-bit digits of canonical scalar i; emit(g, entry) for every non-zero digit.
template <class Emit>
__device__ __forceinline__ void for_each_digit(const U256& scalar, uint32_t i, int c, int W, int D,
                                               uint32_t B, uint32_t idx_bits, Emit emit) {
  DigitState st;
  walk_windows(
      c, W, [&](int k, int avail) { st.buf |= (uint64_t)scalar.v[k] << avail; },
      [&](int w) {
        uint32_t g, e;
        if (st.take(w, i, c, D, B, idx_bits, &g, &e)) emit(g, e);
      });
}

// ---- two-level counting sort of the (bucket, entry) pairs ---------------------------------------
// A single-level sort needs one global atomic per entry and per pass (2 x 58.7 M at n = 2^22:
// 8.7 ms).  Here every pass aggregates in LDS first:
//   level 1: partition by the top bits of the bucket id (<= 1024 partitions); no global atomics: pass
//            A leaves per-block partition histograms, their exclusive scan in [partition][block]
//            order IS every block's write position, pass B ranks its digits with LDS cursors;
//   level 2: inside a partition-ordered array a chunk of 16 K entries spans a contiguous range of
//            <= 4096 buckets: LDS histogram / LDS ranks again, one global atomic per (chunk, bucket).
// Order inside a bucket is arbitrary (EC addition commutes), which is what makes atomics-based
// ranks admissible.  The outputs (count, offset, entries) are those of the single-level sort.
constexpr int P1_THREADS = 256, P1_PER_THREAD = 8, P1_TILE = P1_THREADS * P1_PER_THREAD;
constexpr int P1_MAX_BINS = 1 << MSM_PART_BITS_MAX;  // LDS histogram / cursor array of the level-1 passes
constexpr int P2_THREADS = 256, P2_CHUNK = 16384, P2_BINS = 4096;

struct SortGeom {
  int c, W, D;
  uint32_t B;
  uint32_t idx_bits;  // MsmConfig::idx_bits
  int sh;          // partition = bucket >> sh
  uint32_t bins1;  // number of partitions
};

template <bool MONT>
__device__ __forceinline__ U256 load_scalar(const void* scalars, uint32_t i) {
  if (MONT) return reinterpret_cast<const Fr*>(scalars)[i].to_canonical();  // ark-ff into_bigint
  return reinterpret_cast<const U256*>(scalars)[i];
}

// Level 1, pass A: histogram of the partitions over the tiles of ONE block, stored (not added) at
// blk_hist[partition][block].  The exclusive scan of that array in this order IS the write position
// of every (partition, block) run, so pass B needs neither a tile histogram nor a global cursor:
// the earlier scheme reserved runs with one global atomic per (tile, partition), i.e. ~2000 atomics
// WITH return on each of <= 1024 addresses, which the L2 serialises (0.3 ms of a 0.34 ms kernel at
// 2^22 whether a rank kept all digits or an eighth of them).
template <bool MONT>
__global__ void __launch_bounds__(P1_THREADS) k_part_count(const void* scalars, uint32_t n,
                                                           SortGeom G, uint32_t* blk_hist) {
  __shared__ uint32_t h[P1_MAX_BINS];
  const int tid = threadIdx.x;
  for (uint32_t b = tid; b < G.bins1; b += P1_THREADS) h[b] = 0;
  __syncthreads();
  const uint32_t ntiles = (n + P1_TILE - 1) / P1_TILE;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int k = 0; k < P1_PER_THREAD; ++k) {
      const uint32_t i = tile * P1_TILE + k * P1_THREADS + tid;
      if (i >= n) continue;
      const U256 sc = load_scalar<MONT>(scalars, i);
      for_each_digit(sc, i, G.c, G.W, G.D, G.B, G.idx_bits, [&](uint32_t g, uint32_t) { atomicAdd(&h[g >> G.sh], 1u); });
    }
  }
  __syncthreads();
  for (uint32_t b = tid; b < G.bins1; b += P1_THREADS) blk_hist[(size_t)b * gridDim.x + blockIdx.x] = h[b];
}

// part_off[p] = first position of partition p = blk_off[p][block 0]; part_off[bins1] = total
__global__ void __launch_bounds__(256) k_part_offsets(const uint32_t* __restrict__ blk_off, uint32_t nblk,
                                                      uint32_t bins1, uint32_t* part_off) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p <= bins1) part_off[p] = blk_off[(size_t)p * nblk];  // p = bins1: the scan's total
}

// Cuts the partition-ordered list into `world` contiguous runs of whole partitions with (nearly)
// equal entry counts and leaves rank's run in range[] (MsmSort::range).  Every rank runs this on the
// same histogram, so the cuts agree without a message.  Cut g (the start of rank g's run) is the
// partition boundary whose prefix count is closest to g M / world -- independent of the other cuts,
// so each is one binary search (a serial walk over <= 1024 partitions was 0.1 ms of single-lane LDS
// latency in front of every rank's first accumulation).  A partition larger than a fair share (the
// value-1 bucket of a 0/1-heavy witness) ends up alone in its run, and a neighbouring run may be empty.
__global__ void __launch_bounds__(64) k_pick_range(const uint32_t* __restrict__ part_off, uint32_t bins1,
                                                   int sh, uint32_t nb, int rank, int world,
                                                   uint32_t* range) {
  __shared__ uint32_t cut[2];
  const uint32_t t = threadIdx.x;
  if (t < 2) {
    const int g = rank + (int)t;  // t = 0: start of the run, t = 1: its end
    uint32_t p;
    if (g <= 0) {
      p = 0;
    } else if (g >= world) {
      p = bins1;
    } else {
      const uint64_t M = part_off[bins1];
      const uint64_t tgt = M * (uint64_t)g / (uint64_t)world;
      uint32_t lo = 0, hi = bins1;  // part_off[lo] <= tgt < part_off[hi] (or lo = hi - 1 at the top)
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (part_off[mid] <= tgt) lo = mid;
        else hi = mid;
      }
      // the closer of the two boundaries; ties go down, so equal prefixes (empty partitions) give
      // every rank the same answer
      p = (tgt - part_off[lo] <= part_off[hi] - tgt) ? lo : hi;
    }
    cut[t] = p;
  }
  __syncthreads();
  if (t != 0) return;
  const uint32_t lo = cut[0], hi = cut[1] > cut[0] ? cut[1] : cut[0];
  range[0] = lo;
  range[1] = hi;
  range[2] = part_off[lo];
  range[3] = part_off[hi] - part_off[lo];
  const uint64_t b_lo = (uint64_t)lo << sh, b_hi = (uint64_t)hi << sh;
  range[4] = (uint32_t)(b_lo < nb ? b_lo : nb);
  range[5] = (uint32_t)(b_hi < nb ? b_hi : nb);
}

// Level 1, pass B: every block walks the SAME tiles as in pass A; its write cursor of partition p
// starts at blk_off[p][block] (LDS copy) and a digit's rank is one LDS atomic -- a single digit walk,
// no global atomics.  range != nullptr (bucket-range sharding): only the pairs of partitions
// [range[0], range[1]) are kept, written at their position minus range[2].
template <bool MONT>
__global__ void __launch_bounds__(P1_THREADS) k_part_scatter(const void* scalars, uint32_t n,
                                                             SortGeom G,
                                                             const uint32_t* __restrict__ blk_off,
                                                             MsmPair* part,
                                                             const uint32_t* __restrict__ range) {
  __shared__ uint32_t h[P1_MAX_BINS];
  const int tid = threadIdx.x;
  const uint32_t ntiles = (n + P1_TILE - 1) / P1_TILE;
  const uint32_t p_lo = range ? range[0] : 0u, p_hi = range ? range[1] : G.bins1;
  const uint32_t base = range ? range[2] : 0u;
  if (p_lo >= p_hi) return;
  for (uint32_t b = p_lo + tid; b < p_hi; b += P1_THREADS) h[b] = blk_off[(size_t)b * gridDim.x + blockIdx.x] - base;
  __syncthreads();
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    U256 sc[P1_PER_THREAD];
    uint32_t idx[P1_PER_THREAD];
    bool live[P1_PER_THREAD];
#pragma unroll
    for (int k = 0; k < P1_PER_THREAD; ++k) {
      idx[k] = tile * P1_TILE + (uint32_t)k * P1_THREADS + tid;
      live[k] = idx[k] < n;
      if (live[k]) sc[k] = load_scalar<MONT>(scalars, idx[k]);
      else sc[k] = U256{};
    }
    // four scalars advance window by window together: their four LDS ranks are issued back to
    // back, then the four stores (one scalar at a time the rank -> store chain is pure latency)
    constexpr int U = 4;
    static_assert(P1_PER_THREAD % U == 0, "P1_PER_THREAD must be a multiple of the unroll");
#pragma unroll
    for (int k0 = 0; k0 < P1_PER_THREAD; k0 += U) {
      DigitState wk[U];
      walk_windows(
          G.c, G.W,
          [&](int k, int avail) {
#pragma unroll
            for (int u = 0; u < U; ++u) wk[u].buf |= (uint64_t)sc[k0 + u].v[k] << avail;
          },
          [&](int w) {
            uint32_t g[U], e[U], pos[U];
            bool v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              v[u] = wk[u].take(w, idx[k0 + u], G.c, G.D, G.B, G.idx_bits, &g[u], &e[u]) && live[k0 + u];
              if (v[u]) {
                const uint32_t pb = g[u] >> G.sh;
                v[u] = pb >= p_lo && pb < p_hi;
                if (v[u]) pos[u] = atomicAdd(&h[pb], 1u);
              }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (v[u]) part[pos[u]] = MsmPair{e[u], g[u]};
          });
    }
  }
}

// level 2, pass A: per-bucket counts.  total = part_off[bins1] (device side).
// keep (optional): one bit per point index; pairs of points whose bit is clear are left out of the view
// (the B queries of real circom keys hold a point at infinity for every wire that appears in no B row)
__device__ __forceinline__ bool pair_kept(const uint32_t* __restrict__ keep, uint32_t idx_mask, uint32_t entry) {
  if (!keep) return true;
  const uint32_t idx = entry & idx_mask;
  return (keep[idx >> 5] >> (idx & 31u)) & 1u;
}

template <bool FILTER>
__global__ void __launch_bounds__(P2_THREADS) k_bucket_count(const MsmPair* part,
                                                             const uint32_t* total_ptr, int sh,
                                                             uint32_t* count, const uint32_t* keep,
                                                             uint32_t idx_mask) {
  __shared__ uint32_t h[P2_BINS];
  const int tid = threadIdx.x;
  const uint32_t total = *total_ptr;
  const uint32_t nchunks = (total + P2_CHUNK - 1) / P2_CHUNK;
  for (uint32_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const uint32_t lo = ch * P2_CHUNK;
    const uint32_t hi = lo + P2_CHUNK < total ? lo + P2_CHUNK : total;
    const uint32_t gmin = (part[lo].y >> sh) << sh;
    const uint32_t range = (((part[hi - 1].y >> sh) + 1u) << sh) - gmin;
    if (range <= (uint32_t)P2_BINS) {
      for (uint32_t b = tid; b < range; b += P2_THREADS) h[b] = 0;
      __syncthreads();
      if (FILTER) {
        // the view: whole pairs, eight in flight, then the bitmap lookups, then the LDS counts (the
        // register pattern of k_bucket_scatter below)
        constexpr uint32_t NONE = 0xffffffffu;
        for (uint32_t j0 = lo + tid; j0 < hi; j0 += 8 * P2_THREADS) {
          MsmPair pe[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const uint32_t j = j0 + (uint32_t)u * P2_THREADS;
            pe[u] = j < hi ? part[j] : MsmPair{0u, NONE};
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (pe[u].y != NONE && !pair_kept(keep, idx_mask, pe[u].x)) pe[u].y = NONE;
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (pe[u].y != NONE) atomicAdd(&h[pe[u].y - gmin], 1u);
        }
      } else {
        for (uint32_t j0 = lo + tid; j0 < hi; j0 += 8 * P2_THREADS) {  // eight loads in flight
          uint32_t y[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const uint32_t j = j0 + (uint32_t)u * P2_THREADS;
            y[u] = j < hi ? part[j].y : 0xffffffffu;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (y[u] != 0xffffffffu) atomicAdd(&h[y[u] - gmin], 1u);
        }
      }
      __syncthreads();
      for (uint32_t b = tid; b < range; b += P2_THREADS)
        if (h[b]) atomicAdd(&count[gmin + b], h[b]);
      __syncthreads();
    } else {  // sparse buckets (tiny inputs or huge windows): plain global atomics
      for (uint32_t j = lo + tid; j < hi; j += P2_THREADS) {
        const MsmPair pe = part[j];
        if (!FILTER || pair_kept(keep, idx_mask, pe.x)) atomicAdd(&count[pe.y], 1u);
      }
    }
  }
}

// level 2, pass B: final placement.  A chunk's pairs are loaded ONCE into registers (P2S_PER per
// thread, all loads in flight together), counted into the LDS histogram, and scattered from the
// registers after the chunk's runs have been reserved.
constexpr int P2S_PER = 32, P2S_CHUNK = P2_THREADS * P2S_PER, P2S_GROUP = 8;
template <bool FILTER>
__global__ void __launch_bounds__(P2_THREADS) k_bucket_scatter(const MsmPair* part,
                                                               const uint32_t* total_ptr, int sh,
                                                               uint32_t* cursor, uint32_t* entries,
                                                               const uint32_t* keep, uint32_t idx_mask) {
  __shared__ uint32_t h[P2_BINS];
  const int tid = threadIdx.x;
  const uint32_t total = *total_ptr;
  const uint32_t nchunks = (total + P2S_CHUNK - 1) / P2S_CHUNK;
  constexpr uint32_t NONE = 0xffffffffu;  // never a bucket id
  for (uint32_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const uint32_t lo = ch * P2S_CHUNK;
    const uint32_t hi = lo + P2S_CHUNK < total ? lo + P2S_CHUNK : total;
    const uint32_t gmin = (part[lo].y >> sh) << sh;
    const uint32_t range = (((part[hi - 1].y >> sh) + 1u) << sh) - gmin;
    if (range <= (uint32_t)P2_BINS) {
      for (uint32_t b = tid; b < range; b += P2_THREADS) h[b] = 0;
      MsmPair pe[P2S_PER];
#pragma unroll
      for (int u = 0; u < P2S_PER; ++u) {
        const uint32_t j = lo + (uint32_t)u * P2_THREADS + tid;
        pe[u] = j < hi ? part[j] : MsmPair{0u, NONE};
        if (FILTER && pe[u].y != NONE && !pair_kept(keep, idx_mask, pe[u].x)) pe[u].y = NONE;
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < P2S_PER; ++u)
        if (pe[u].y != NONE) atomicAdd(&h[pe[u].y - gmin], 1u);
      __syncthreads();
      // reserve this chunk's run in every bucket; h[] becomes the running write cursor
      for (uint32_t b = tid; b < range; b += P2_THREADS) {
        const uint32_t cnt = h[b];
        h[b] = cnt ? atomicAdd(&cursor[gmin + b], cnt) : 0u;
      }
      __syncthreads();
#pragma unroll
      for (int u0 = 0; u0 < P2S_PER; u0 += P2S_GROUP) {  // ranks of a group back to back, then its stores
        uint32_t pos[P2S_GROUP];
#pragma unroll
        for (int u = 0; u < P2S_GROUP; ++u)
          if (pe[u0 + u].y != NONE) pos[u] = atomicAdd(&h[pe[u0 + u].y - gmin], 1u);
#pragma unroll
        for (int u = 0; u < P2S_GROUP; ++u)
          if (pe[u0 + u].y != NONE) entries[pos[u]] = pe[u0 + u].x;
      }
      __syncthreads();
    } else {  // sparse buckets (tiny inputs or huge windows): plain global atomics
      for (uint32_t j = lo + tid; j < hi; j += P2_THREADS) {
        const MsmPair pe = part[j];
        if (!FILTER || pair_kept(keep, idx_mask, pe.x)) entries[atomicAdd(&cursor[pe.y], 1u)] = pe.x;
      }
    }
  }
}

// ---- exclusive scan of L u32 values (mode = 0) or of ceil(x / mode) (mode = chunk), 1024 values per block
constexpr int SCAN_T = 256, SCAN_V = 4, SCAN_TILE = SCAN_T * SCAN_V;

__device__ __forceinline__ uint32_t scan_xform(uint32_t x, int mode) {
  return mode ? (x + (uint32_t)mode - 1) / (uint32_t)mode : x;
}

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, total in *tot
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* sh, uint32_t* tot) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int off = 1; off < SCAN_T; off <<= 1) {
    uint32_t x = (t >= off) ? sh[t - off] : 0;
    __syncthreads();
    sh[t] += x;
    __syncthreads();
  }
  const uint32_t incl = sh[t];
  *tot = sh[SCAN_T - 1];
  __syncthreads();
  return incl - v;
}

__global__ void __launch_bounds__(SCAN_T) k_scan_block_sums(const uint32_t* in, uint32_t L, int mode,
                                                            uint32_t* bsum) {
  __shared__ uint32_t sh[SCAN_T];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_V;
  uint32_t s = 0;
  for (int v = 0; v < SCAN_V; ++v)
    if (base + v < L) s += scan_xform(in[base + v], mode);
  uint32_t tot;
  (void)block_excl_scan(s, sh, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_T) k_scan_top(uint32_t* bsum, uint32_t nblk, uint32_t* total) {
  __shared__ uint32_t sh[SCAN_T];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nblk; base += SCAN_T) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nblk ? bsum[i] : 0;
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, sh, &tot);
    if (i < nblk) bsum[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(SCAN_T) k_scan_apply(const uint32_t* in, uint32_t L, int mode,
                                                       const uint32_t* bsum, uint32_t* out,
                                                       uint32_t* out2) {
  __shared__ uint32_t sh[SCAN_T];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_V;
  uint32_t x[SCAN_V];
  uint32_t s = 0;
  for (int v = 0; v < SCAN_V; ++v) {
    x[v] = (base + v < L) ? scan_xform(in[base + v], mode) : 0;
    s += x[v];
  }
  uint32_t tot;
  uint32_t run = block_excl_scan(s, sh, &tot) + bsum[blockIdx.x];
  for (int v = 0; v < SCAN_V; ++v) {
    if (base + v < L) {
      out[base + v] = run;
      if (out2) out2[base + v] = run;
    }
    run += x[v];
  }
}

// out[0..L) = exclusive scan, out[L] = total.  tmp needs ceil(L/1024) words.
void scan_exclusive(const uint32_t* in, uint32_t L, int mode, uint32_t* out, uint32_t* out2,
                    uint32_t* tmp, hipStream_t s) {
  const uint32_t nblk = ceil_div(L, SCAN_TILE);
  G16_LAUNCH(k_scan_block_sums, nblk, SCAN_T, 0, s, in, L, mode, tmp);
  G16_LAUNCH(k_scan_top, 1, SCAN_T, 0, s, tmp, nblk, out + L);
  G16_LAUNCH(k_scan_apply, nblk, SCAN_T, 0, s, in, L, mode, (const uint32_t*)tmp, out, out2);
}

// buckets whose entries span more than MSM_SMALL_MULTI lane segments (hot buckets of a skewed
// witness): listed for k_combine_large.  Rare, so the atomics do not matter.
__global__ void __launch_bounds__(256) k_find_large(const uint32_t* offset, uint32_t nb,
                                                    uint32_t lanes, uint32_t* multi_l,
                                                    uint32_t* meta) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  const uint32_t lo = offset[g], hi = offset[g + 1];
  if (hi == lo) return;
  const uint32_t S = msm_seg_len(offset[nb], lanes);
  const uint32_t np = (hi - 1) / S - lo / S + 1;
  if (np > (uint32_t)MSM_SMALL_MULTI) multi_l[atomicAdd(&meta[1], 1u)] = g;
}

}  // namespace


// workgroups of the sort kernels
static uint32_t sort_grid_cap() {
#ifdef G16_EMU
  return 32u;  // emulator: every thread of a launch is stepped through
#else
  return 2048u;
#endif
}

void MsmSort::set_shard(int rank_, int world_) {
  if (world_ < 1 || rank_ < 0 || rank_ >= world_) throw std::runtime_error("MsmSort::set_shard: bad rank/world");
  rank = rank_;
  world = world_;
  if (world > 1) range.alloc(8);
}

void MsmSort::init(uint32_t capacity, const MsmConfig& c) {
  cfg = c;
  cap = capacity;
  // an entry holds idx_bits index bits and 31 - idx_bits plane bits (msm.h); 2^27 points is also the
  // largest domain the reference accepts (witness_map.hip: PolynomialDegreeTooLarge above it)
  if (capacity > cfg.max_points() || (uint32_t)cfg.Pn > (1u << (31 - cfg.idx_bits)))
    throw std::runtime_error("MSM slice does not fit the sort entry (2^27 points x 16 planes, or 2^26 x 32)");
  const uint32_t nb = cfg.nb();
  const uint64_t M = (uint64_t)cap * cfg.W;
  if (M >= ((uint64_t)1 << 32)) throw std::runtime_error("MSM entry count exceeds 2^32");
  part.alloc(M ? M : 1);
  const size_t bins1 = (size_t)1 << msm_part_bits(nb);
  part_off.alloc(bins1 + 1);
  blk_hist.alloc(bins1 * sort_grid_cap() + 1);
  blk_off.alloc(bins1 * sort_grid_cap() + 1);
  count.alloc((size_t)nb + 1);
  offset.alloc((size_t)nb + 1);
  cursor.alloc((size_t)nb + 1);
  entries.alloc(M ? M : 1);
  // a bucket is 'large' when it spans > MSM_SMALL_MULTI segments of >= MSM_MIN_SEG entries
  multi_l.alloc((size_t)(M / ((uint64_t)MSM_MIN_SEG * MSM_SMALL_MULTI)) + 2);
  meta.alloc(4);
  multi_l2.alloc((size_t)(M / ((uint64_t)MSM_MIN_SEG * MSM_SMALL_MULTI)) + 2);
  meta2.alloc(4);
  const uint64_t scan_len = std::max<uint64_t>((uint64_t)nb + 1, (uint64_t)bins1 * sort_grid_cap() + 1);
  scan_tmp.alloc(ceil_div(scan_len, SCAN_TILE) + 1);
}

size_t MsmSort::bytes_for(uint32_t capacity, const MsmConfig& cfg) {
  const uint64_t M = std::max<uint64_t>((uint64_t)capacity * cfg.W, 1);
  const uint64_t nb = cfg.nb();
  const uint64_t bins1 = (uint64_t)1 << msm_part_bits((uint32_t)nb);
  const uint64_t hist = bins1 * sort_grid_cap() + 1;
  const uint64_t large = M / ((uint64_t)MSM_MIN_SEG * MSM_SMALL_MULTI) + 2;
  return (size_t)(M * sizeof(MsmPair) + M * 4 + 3 * (nb + 1) * 4 + 2 * hist * 4 + 2 * large * 4 +
                  (ceil_div(std::max<uint64_t>(nb + 1, hist), SCAN_TILE) + 1) * 4 + (bins1 + 1) * 4 + 64);
}

size_t MsmSort::device_bytes() const {
  return part.bytes() + count.bytes() + offset.bytes() + cursor.bytes() + entries.bytes() +
         multi_l.bytes();
}

void MsmSort::run(const void* scalars, uint32_t n, bool mont, hipStream_t s) {
  if (n > cap) throw std::runtime_error("MsmSort::run: more scalars than capacity");
  len = n;
  const uint32_t nb = cfg.nb();
  SortGeom G;
  G.c = cfg.c;
  G.W = cfg.W;
  G.D = cfg.D;
  G.B = cfg.B;
  G.idx_bits = cfg.idx_bits;
  G.sh = msm_part_shift(nb);
  G.bins1 = ((nb - 1) >> G.sh) + 1;
  G16_HIP(hipMemsetAsync(count.p, 0, ((size_t)nb + 1) * 4, s));
  G16_HIP(hipMemsetAsync(meta.p, 0, 16, s));
  G16_HIP(hipMemsetAsync(meta2.p, 0, 16, s));
  const uint32_t grid_cap = sort_grid_cap();
  uint32_t grid1 = ceil_div(n, P1_TILE);
  if (grid1 > grid_cap) grid1 = grid_cap;
  if (grid1 < 1) grid1 = 1;
  uint32_t grid2 = ceil_div((uint64_t)n * cfg.W, P2_CHUNK);
  if (grid2 > grid_cap) grid2 = grid_cap;
  if (grid2 < 1) grid2 = 1;
  // level 1: per-block partition histograms -> scan (= every block's write positions) -> partition
  const uint32_t L1 = G.bins1 * grid1;
  if (mont) G16_LAUNCH((k_part_count<true>), grid1, P1_THREADS, 0, s, scalars, n, G, blk_hist.p);
  else G16_LAUNCH((k_part_count<false>), grid1, P1_THREADS, 0, s, scalars, n, G, blk_hist.p);
  scan_exclusive(blk_hist.p, L1, 0, blk_off.p, nullptr, scan_tmp.p, s);
  G16_LAUNCH(k_part_offsets, ceil_div(G.bins1 + 1, 256), 256, 0, s, (const uint32_t*)blk_off.p, grid1,
             G.bins1, part_off.p);
  const uint32_t* rng = range_dev();
  if (rng)
    G16_LAUNCH(k_pick_range, 1, 64, 0, s, (const uint32_t*)part_off.p, G.bins1, G.sh, nb, rank, world,
               range.p);
  if (mont)
    G16_LAUNCH((k_part_scatter<true>), grid1, P1_THREADS, 0, s, scalars, n, G, (const uint32_t*)blk_off.p,
               part.p, rng);
  else
    G16_LAUNCH((k_part_scatter<false>), grid1, P1_THREADS, 0, s, scalars, n, G, (const uint32_t*)blk_off.p,
               part.p, rng);
  // level 2: bucket sizes, offsets, final placement (over this rank's pairs only when sharded)
  const uint32_t* total = rng ? rng + 3 : part_off.p + G.bins1;
  G16_LAUNCH((k_bucket_count<false>), grid2, P2_THREADS, 0, s, (const MsmPair*)part.p, total, G.sh, count.p,
             (const uint32_t*)nullptr, 0u);
  scan_exclusive(count.p, nb, 0, offset.p, cursor.p, scan_tmp.p, s);
  uint32_t grid3 = ceil_div((uint64_t)n * cfg.W, P2S_CHUNK);
  if (grid3 > 2 * grid_cap) grid3 = 2 * grid_cap;
  if (grid3 < 1) grid3 = 1;
  G16_LAUNCH((k_bucket_scatter<false>), grid3, P2_THREADS, 0, s, (const MsmPair*)part.p, total, G.sh, cursor.p,
             entries.p, (const uint32_t*)nullptr, 0u);
  G16_LAUNCH(k_find_large, ceil_div(nb, 256), 256, 0, s, (const uint32_t*)offset.p, nb, cfg.lanes,
             multi_l.p, meta.p);
  G16_LAUNCH(k_find_large, ceil_div(nb, 256), 256, 0, s, (const uint32_t*)offset.p, nb, cfg.lanes2,
             multi_l2.p, meta2.p);
}

// ---- a filtered view of another sort: level 2 again over `src`'s level-1 pairs, leaving out the
// points whose bit in `keep` is clear.  Same bucket ids, same configuration, own offsets / entries /
// hot-bucket lists: everything the accumulation and reduction kernels read from an MsmSort.
void MsmSort::init_view(uint32_t capacity, const MsmConfig& c) {
  cfg = c;
  cap = capacity;
  const uint32_t nb = cfg.nb();
  const uint64_t M = (uint64_t)cap * cfg.W;
  count.alloc((size_t)nb + 1);
  offset.alloc((size_t)nb + 1);
  cursor.alloc((size_t)nb + 1);
  entries.alloc(M ? M : 1);
  multi_l.alloc((size_t)(M / ((uint64_t)MSM_MIN_SEG * MSM_SMALL_MULTI)) + 2);
  meta.alloc(4);
  multi_l2.alloc((size_t)(M / ((uint64_t)MSM_MIN_SEG * MSM_SMALL_MULTI)) + 2);
  meta2.alloc(4);
  scan_tmp.alloc(ceil_div((uint64_t)nb + 1, SCAN_TILE) + 1);
}

void MsmSort::run_view(const MsmSort& src, const uint32_t* keep_bits, hipStream_t s) {
  if (src.world > 1) throw std::runtime_error("MsmSort::run_view: the source sort is sharded");
  len = src.len;
  const uint32_t nb = cfg.nb(), n = src.len;
  const int sh = msm_part_shift(nb);
  const uint32_t bins1 = ((nb - 1) >> sh) + 1;
  const uint32_t idx_mask = (1u << cfg.idx_bits) - 1u;
  G16_HIP(hipMemsetAsync(count.p, 0, ((size_t)nb + 1) * 4, s));
  G16_HIP(hipMemsetAsync(meta.p, 0, 16, s));
  G16_HIP(hipMemsetAsync(meta2.p, 0, 16, s));
  const uint32_t grid_cap = sort_grid_cap();
  uint32_t grid2 = ceil_div((uint64_t)n * cfg.W, P2_CHUNK);
  if (grid2 > grid_cap) grid2 = grid_cap;
  if (grid2 < 1) grid2 = 1;
  const uint32_t* total = src.part_off.p + bins1;
  G16_LAUNCH((k_bucket_count<true>), grid2, P2_THREADS, 0, s, (const MsmPair*)src.part.p, total, sh, count.p,
             keep_bits, idx_mask);
  scan_exclusive(count.p, nb, 0, offset.p, cursor.p, scan_tmp.p, s);
  uint32_t grid3 = ceil_div((uint64_t)n * cfg.W, P2S_CHUNK);
  if (grid3 > 2 * grid_cap) grid3 = 2 * grid_cap;
  if (grid3 < 1) grid3 = 1;
  G16_LAUNCH((k_bucket_scatter<true>), grid3, P2_THREADS, 0, s, (const MsmPair*)src.part.p, total, sh, cursor.p,
             entries.p, keep_bits, idx_mask);
  G16_LAUNCH(k_find_large, ceil_div(nb, 256), 256, 0, s, (const uint32_t*)offset.p, nb, cfg.lanes, multi_l.p,
             meta.p);
  G16_LAUNCH(k_find_large, ceil_div(nb, 256), 256, 0, s, (const uint32_t*)offset.p, nb, cfg.lanes2, multi_l2.p,
             meta2.p);
}

}  // namespace g16
