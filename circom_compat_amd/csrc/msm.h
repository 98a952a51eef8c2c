// msm.h -- BN254 variable-base multi-scalar multiplication (Pippenger bucket method) on gfx950.
//
// Replaces ark-ec VariableBaseMSM::msm_bigint as invoked (through ark-groth16
// create_proof_with_assignment) for the A, B1, L, H queries in G1 and the B2 query in G2; the
// reference call sites are src/zkey.rs:903-911 and benches/groth16.rs:52-60, the query arrays are
// the ones read_zkey produces (src/zkey.rs:103-133).
//
// MI355X-first design:
//  * signed c-bit digits (2^(c-1) buckets per bucket set), zero digits skipped;
//  * the 288 GB of HBM is spent on precomputed multiples 2^(c*D*j) * P_i stored as extra affine
//    "planes": window w = j*D + d reads plane j and feeds bucket set d, so with full
//    precomputation (D = 1) ALL windows share one bucket set -- one bucket reduction per MSM and
//    no 254-step doubling chain at the end;
//  * one counting sort of (bucket, point) pairs per *scalar vector*; A, B1, B2 and L reuse the same
//    sorted list because they share the witness as scalars;
//  * bucket filling is balanced by ENTRIES, not by buckets: the sorted entry list is cut into
//    cfg.lanes equal contiguous segments, one per lane of a persistent grid; a lane emits one partial
//    sum per bucket it touches (slot = bucket + lane, unique and contiguous per bucket).  Every lane
//    runs the same trip count whatever the bucket sizes, so skewed witnesses (most circom wires are
//    0/1) neither serialise on a hot bucket nor leave lanes idle; a hot bucket simply spans many
//    lanes and its partials are tree-summed by a workgroup.
#pragma once
#include <stdlib.h>
#include "common.h"
#include "ec29.h"

namespace g16 {

constexpr int MSM_ACC_THREADS = 128;    // workgroup of the accumulation kernel
#ifdef G16_EMU  // the CPU SIMT emulator steps through every thread of a launch: tests use small grids
constexpr int MSM_ACC_BLOCKS = 48;
constexpr int MSM_ACC_BLOCKS_G2 = 32;
#else
constexpr int MSM_ACC_BLOCKS = 3072;    // G1: segments = 3072 x 128 -- two rounds of the optimistic kernel's 3 waves per SIMD on 256 CUs (G16_ACC_GRID overrides)
constexpr int MSM_ACC_BLOCKS_G2 = 2048; // G2: 2048 x 128 (one wave per SIMD, four rounds; G16_ACC_GRID_G2 overrides).  Same box, 2^22,
                                        // ms per launch at 2048 / 3072 workgroups: A|B1 pair 9.69 / 9.38, L or H 4.73 / 4.49, B2 12.1 / 12.8
#endif
constexpr int MSM_MIN_SEG = 8;          // shortest per-lane segment
constexpr int MSM_SMALL_MULTI = 32;     // buckets with <= this many partials are summed inside the reduce
constexpr int MSM_RED_CHUNK = 16;   // max buckets per thread in the weighted bucket reduction
// A sort entry packs (point index, plane, sign) into 32 bits: idx | plane << idx_bits | neg << 31.
// MsmConfig::idx_bits = 27 (4 plane bits) whenever the configuration stores <= 16 planes -- every
// window c >= 16, i.e. every large input: slices of up to 2^27 points, the largest domain the
// reference accepts (Fr has two-adicity 28 and qap.rs:63-68 needs the 2n domain) -- and 26 (5 plane
// bits, <= 32 planes) for the small windows of tiny inputs.
constexpr uint32_t MSM_IDX_BITS_WIDE = 27, MSM_IDX_BITS_NARROW = 26;

struct MsmConfig {
  int c = 0;       // window bits
  int W = 0;       // number of windows, W*c >= 255
  int Pn = 1;      // stored multiples (planes) per point
  int D = 0;       // bucket sets = ceil(W / Pn)
  uint32_t B = 0;  // buckets per set = 2^(c-1)
  uint32_t idx_bits = MSM_IDX_BITS_NARROW;  // point-index bits of a sort entry (31 - idx_bits plane bits)
  uint32_t max_points() const { return 1u << idx_bits; }
  uint32_t lanes = MSM_ACC_BLOCKS * MSM_ACC_THREADS;        // segments the entry list is cut into by the G1 launches
  uint32_t lanes2 = MSM_ACC_BLOCKS_G2 * MSM_ACC_THREADS;    // ... by the G2 launch over the same sort
  uint32_t nb() const { return (uint32_t)D * B; }
  uint32_t max_lanes() const { return lanes > lanes2 ? lanes : lanes2; }
};

// level-1 partition of the counting sort = bucket >> msm_part_shift(nb): 2^10 partitions (every (sort
// block, partition) pair is one write stream of the level-1 scatter; 256 partitions: level 1 0.62 -> 0.48 ms,
// level 2 0.39 -> 0.61 ms at 2^22, profiles/r03_sort_partition_ab.txt) -- more only where level 2 needs
// them: its LDS histogram holds 4096 buckets per chunk, and a partition that spans more falls back to one
// global atomic per ENTRY (seen at 2^27 on one GPU: 3 bucket sets of 2^21 = 6144 buckets per partition,
// both sorts 346 ms instead of ~90).  So 2^11 / 2^12 partitions once the bucket sets exceed 2^22 / 2^23.
constexpr int MSM_PART_BITS_MAX = 12;
inline int msm_part_bits(uint32_t nb) {
  int bits = 10;
  while (bits < MSM_PART_BITS_MAX && (((uint64_t)nb + ((uint64_t)1 << bits) - 1) >> bits) > 4096u) ++bits;
  return bits;
}
inline int msm_part_shift(uint32_t nb) {
  int bits = 0;
  while (((uint64_t)1 << bits) < nb) ++bits;
  const int pb = msm_part_bits(nb);
  return bits > pb ? bits - pb : 0;
}

// buckets per thread of k_bucket_reduce.  The kernel is a serial chain of ~3 EC additions per bucket
// plus one lo * run product (~21 addition-equivalents) per thread, each addition ~9 us of one wave's
// issue slots: one wave per SIMD (65536 threads) minimises chain x waves-per-SIMD.  Measured at
// 2^22 (2^19 buckets): 8 per thread 5.0 ms of reductions per proof vs 6.35 ms at 4, 8.7 ms at 2.
// `world` > 1 (bucket-range sharding, MsmSort::set_shard): this rank reduces ~1/world of the bucket
// set, so the chunk shrinks with it (the threads outside the rank's range exit at once); a chunk
// never exceeds a level-1 sort partition, whose boundaries the rank ranges are cut at.
// `hidden`: the reduction runs on the `red` stream underneath the next accumulation.  There its
// additions cost accumulation issue slots (65536 threads of one bucket do 25 additions per bucket,
// 8192 threads of 8 buckets 5.3), but its latency still matters: the reductions of a proof queue up
// on that one stream and the variable-base products wait for the first of them.  Measured on one box
// (scripts/gpu_r3_run13.sh; 2^20 proof / one rank of 8 at 2^22, ms): 65536 threads 11.4-11.8 / 7.8-8.0,
// 16384 threads 11.5-11.9 / 7.7-8.1, 8192 threads 12.4-12.5 / 8.9-9.1 -- the narrow reductions finish
// after the last accumulation (round 3's schedule; round 6 re-measured below).
inline uint32_t msm_red_chunk(const MsmConfig& cfg, uint32_t nbatch = 1, uint32_t world = 1, bool hidden = false) {
  // Hidden reductions (red / side streams, underneath an accumulation): HALF the exposed width.  Round 3 measured
  // 65536 against 16384 threads as a tie; with round 6's schedule, same box (profiles/r06_schedule_ab.txt, block
  // 3), 65536 / 32768 / 16384 threads: 2^19 6.77 / 6.35 / 6.53 ms, 2^20 11.44 / 11.07 / 10.97, dense-skewed 2^20 7.27 /
  // 7.11 / 7.44, 2^18 4.51 / 4.50 / 4.61 -- the narrower launch takes fewer issue slots from the accumulation it
  // hides under and still finishes before it.  Exposed reductions keep 65536 (2^22: 35.45 ms; 131072: 36.18;
  // 32768: 36.90).
  // Only for SMALL bucket sets (<= 2^17 buckets per launch and rank): with more, half the threads means chains of
  // 16 buckets and the reductions outlast what they hide under (one rank of 2 at 2^22: 20.4 -> 21.3 ms; one
  // bucket-sharded rank of 8 at 2^24: 21.1 -> 23.8 ms, profiles/r06_proj_k22.json / _k24.json first runs).
  const uint64_t per_launch = (uint64_t)nbatch * cfg.D * cfg.B / (world ? world : 1);
  const uint32_t target = (hidden && per_launch <= (1u << 17)) ? 32768u : 65536u;
  uint32_t ch = (uint32_t)(((uint64_t)nbatch * cfg.D * cfg.B) / ((uint64_t)target * (world ? world : 1)));
  if (ch < 1) ch = 1;
  if (ch > (uint32_t)MSM_RED_CHUNK) ch = (uint32_t)MSM_RED_CHUNK;
  if (world > 1) {  // largest power of two <= ch, <= 2^(partition shift)
    uint32_t p2 = 1;
    while (p2 * 2 <= ch) p2 *= 2;
    ch = p2;
    const uint32_t part = 1u << msm_part_shift(cfg.nb());
    if (ch > part) ch = part;
  }
  return ch;
}

// c_override / planes_override <= 0 selects the defaults for `len` scalars.
MsmConfig msm_make_config(size_t len, int c_override, int planes_override);

struct alignas(8) MsmPair {  // level-1 sort record
  uint32_t x;  // entry
  uint32_t y;  // global bucket id
};

// entries per lane for a list of M entries cut over `lanes` lanes (same formula on host and device)
G16_HD uint32_t msm_seg_len(uint32_t M, uint32_t lanes) {
  uint32_t S = (uint32_t)(((uint64_t)M + lanes - 1) / lanes);
  return S < (uint32_t)MSM_MIN_SEG ? (uint32_t)MSM_MIN_SEG : S;
}

// Sorted (bucket -> entries) view of one scalar vector.  entry = idx | plane << cfg.idx_bits | neg << 31.
struct MsmSort {
  MsmConfig cfg;
  uint32_t cap = 0, len = 0;
  DevBuf<MsmPair> part;  // level-1 output: (entry, bucket) pairs ordered by partition
  // multi_l / meta: hot buckets of the G1 segmentation (cfg.lanes), multi_l2 / meta2: of the G2 one; meta[1] = their number
  DevBuf<uint32_t> count, offset, cursor, entries, multi_l, meta, multi_l2, meta2, scan_tmp;
  uint32_t lanes_of(bool g2) const { return g2 ? cfg.lanes2 : cfg.lanes; }
  const uint32_t* large_list(bool g2) const { return g2 ? multi_l2.p : multi_l.p; }
  const uint32_t* large_meta(bool g2) const { return g2 ? meta2.p : meta.p; }
  DevBuf<uint32_t> part_off;            // level-1 partition offsets (+ total)
  DevBuf<uint32_t> blk_hist, blk_off;   // [partition][sort block]: per-block counts / their exclusive scan
  // Bucket-range sharding (set_shard, world > 1): every rank walks ALL `n` scalars but keeps only
  // the (bucket, point) pairs of ITS contiguous run of level-1 partitions, chosen on the device from
  // the partition histogram so that the ranks hold equal shares of the sorted entry list (every rank
  // computes the same histogram from the same scalars, hence the same cut).  Bucket ids stay global:
  // count / offset span all nb buckets (zero outside the rank's run), entries holds the rank's
  // M_local = offset[nb] entries.  range = {p_lo, p_hi, entry_base, M_local, b_lo, b_hi}.
  int rank = 0, world = 1;
  DevBuf<uint32_t> range;
  const uint32_t* range_dev() const { return world > 1 ? range.p : nullptr; }

  void init(uint32_t capacity, const MsmConfig& cfg);
  void set_shard(int rank, int world);
  // scalars: `n` field elements (Montgomery Fr when mont, else canonical U256) in device memory
  void run(const void* scalars, uint32_t n, bool mont, hipStream_t stream);
  // A filtered VIEW of another (unsharded) sort over the same scalars: level 2 re-run over src's level-1
  // pairs without the points whose bit in keep_bits is clear (msm_sort.hip).  The B queries of a real
  // circom key hold the point at infinity for every wire that appears in no B row: a third of the wires
  // of a Poseidon chain; the shared witness sort would spend a full G2 mixed addition on each.
  void init_view(uint32_t capacity, const MsmConfig& cfg);
  void release_view() {  // a view whose allocation failed half way: back to the empty state
    count.release(); offset.release(); cursor.release(); entries.release(); multi_l.release();
    meta.release(); multi_l2.release(); meta2.release(); scan_tmp.release();
    cap = len = 0;
  }
  void run_view(const MsmSort& src, const uint32_t* keep_bits, hipStream_t stream);
  static size_t view_bytes_for(uint32_t capacity, const MsmConfig& cfg) {
    const uint64_t M = (uint64_t)capacity * cfg.W;
    return (size_t)(M * 4 + 3 * ((uint64_t)cfg.nb() + 1) * 4 + 2 * (M / ((uint64_t)MSM_MIN_SEG * MSM_SMALL_MULTI) + 2) * 4 + 4096);
  }
  size_t device_bytes() const;
  // what init(capacity, cfg) allocates (the memory plan of g16_ctx_create: api.hip, plan_msm_configs)
  static size_t bytes_for(uint32_t capacity, const MsmConfig& cfg);
};

// accumulator type of the kernels: XYZZ over the lazy 9 x 29-bit limbs (field29.h / ec29.h)
template <class F>
using MsmAcc = XYZZ29<typename Lazy<F>::type>;

template <class F>
struct MsmPoints {
  MsmConfig cfg;
  uint32_t count = 0;
  // [Pn][count][stride], PACKED INTERNAL form (Lazy<F>): canonical x * 2^261 mod q in the 8 words of
  // an Fq.  stride 1: this query alone.  stride 2: two queries over the same scalars interleaved
  // point by point (A_i | B1_i = ONE 128-byte line per gather instead of two half-used ones);
  // `off` selects this query's half, the second query is a view (`view`) of the first one's buffer.
  DevBuf<Affine<F>> pts;
  const Affine<F>* view = nullptr;
  uint32_t stride = 1, off = 0;
  const Affine<F>* data() const { return view ? view : pts.p; }
  // uploads `count` affine points (packed Montgomery x|y, all-zero = infinity) and fills the planes
  void init(const Affine<F>* host_points, uint32_t count, const MsmConfig& cfg, hipStream_t stream);
  // two queries of the same length interleaved in a's buffer (b becomes a view of it)
  static void init_pair(MsmPoints& a, MsmPoints& b, const Affine<F>* host_a, const Affine<F>* host_b,
                        uint32_t count, const MsmConfig& cfg, hipStream_t stream);
  // same, from points already in device memory (key generator)
  void init_from_device(const Affine<F>* dev_points, uint32_t count, const MsmConfig& cfg,
                        hipStream_t stream);
};

// Mixed additions the optimistic accumulation kernel set aside (its x-coordinate filter fired:
// acc = +-P is possible): (slot of the partial sum the addition belongs to, sort entry).  A list that
// overflows marks the launch for the exact kernel.
// ~2^-22.7 of the mixed additions are set aside (the filter's false positives): 6 per launch at 2^22, 242 per
// single-query launch and 484 per pair launch at 2^27 -- where a 512-entry list overflowed in one proof of
// two and the exact kernel redid a 270 ms launch (profiles/r04_bench_chain27.json, first run)
#ifdef G16_EMU
constexpr uint32_t MSM_FIX_CAP = 512;  // emulator: keeps the overflow test small
#else
constexpr uint32_t MSM_FIX_CAP = 2048;
#endif
struct MsmFixList {
  uint32_t count;     // appended by k_bucket_accumulate<.., FAST>, reset by k_acc_fixup
  uint32_t overflow;  // written by k_acc_fixup: the exact kernel behind it redoes the whole launch
  uint32_t slot[MSM_FIX_CAP];  // bit 31: second half of an interleaved pair
  uint32_t entry[MSM_FIX_CAP];
};

template <class F>
struct MsmWork {
  int batch = 1;                    // MSMs whose partials can be alive at once (workspace slots)
  uint32_t slots = 0, ncontrib = 0;  // per slot
  int sets = 1;
  DevBuf<MsmAcc<F>> partial;  // [batch][slots]; slot index = bucket + lane
  DevBuf<MsmAcc<F>> contrib;  // [batch][ncontrib]: one per reduction chunk
  DevBuf<MsmAcc<F>> bsum;     // [batch][256 * sets]: intermediate tree level
  DevBuf<MsmAcc<F>> wsum;     // [batch][sets]: one per bucket set
  DevBuf<MsmFixList> fix;     // [batch]: deferred exact additions of the accumulation into each slot
  // sized for the larger of several sorts that will share this workspace
  void init(uint32_t n_slots, uint32_t n_contrib, int max_sets, int batch = 1);
};

// what MsmWork<F>::init(n_slots, n_contrib, max_sets, batch) allocates
template <class F>
inline size_t msm_work_bytes(uint32_t n_slots, uint32_t n_contrib, int max_sets, int batch) {
  return (size_t)batch * ((size_t)n_slots + n_contrib + (size_t)257 * max_sets) * sizeof(MsmAcc<F>) +
         (size_t)batch * sizeof(MsmFixList);
}

// out_dev[0] = sum_i scalar_i * P_{i - idx_min} over the entries of `s` with idx >= idx_min
// (lazy internal form; finalize.hip converts to the storage form when it writes the proof).
template <class F>
void msm_run(const MsmSort& s, const MsmPoints<F>& pts, uint32_t idx_min, MsmWork<F>& work,
             MsmAcc<F>* out_dev, hipStream_t stream, StageTimer* tm = nullptr);
// The two halves of msm_run, for MSMs that share a sort (A, B1, L over the witness scalars): their
// accumulations go into different workspace slots and ONE batched reduction finishes all of them.
// fixup = false (G1 only): just the optimistic kernel; the caller runs msm_fixup on the stream that
// reduces this slot, so that the single-block fix-up never sits between two accumulations
template <class F>
void msm_accumulate(const MsmSort& s, const MsmPoints<F>& pts, uint32_t idx_min, MsmWork<F>& work,
                    int slot, hipStream_t stream, StageTimer* tm = nullptr, bool fixup = true);
// the deferred exact additions of the accumulation(s) into `slot` (an interleaved pair: both halves,
// slots `slot` and `slot + 1`): k_acc_fixup + the exact kernel that redoes an overflowed launch
template <class F>
void msm_fixup(const MsmSort& s, const MsmPoints<F>& pts, uint32_t idx_min, MsmWork<F>& work, int slot,
               hipStream_t stream, StageTimer* tm = nullptr);
// a, b: the two halves of an interleaved pair (MsmPoints::init_pair).  ONE launch: the two waves of
// a workgroup walk the same 64 segments, wave 0 adding a's points into `slot`, wave 1 b's points into
// `slot + 1` -- the second wave finds the 128-byte line its neighbour just pulled in the cache.
template <class F>
void msm_accumulate_pair(const MsmSort& s, const MsmPoints<F>& a, const MsmPoints<F>& b,
                         MsmWork<F>& work, int slot, hipStream_t stream, StageTimer* tm = nullptr,
                         bool fixup = true);
template <class F>
void msm_fixup_pair(const MsmSort& s, const MsmPoints<F>& a, const MsmPoints<F>& b, MsmWork<F>& work,
                    int slot, hipStream_t stream, StageTimer* tm = nullptr);
template <class F>
void msm_reduce(const MsmSort& s, MsmWork<F>& work, int first_slot, int nbatch, MsmAcc<F>* out_dev,
                hipStream_t stream, StageTimer* tm = nullptr, bool hidden = false);



}  // namespace g16
