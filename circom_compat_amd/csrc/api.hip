// api.hip -- the C ABI declared in include/g16_amd.h: context (device-resident key, matrices,
// tables, workspaces) and the per-proof drivers.  Host-side orchestration only; all arithmetic
// runs in the HIP kernels of ntt.hip / witness_map.hip / msm_*.hip / finalize.hip.
#include "../../include/g16_amd.h"

#include <stddef.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>

#include "ctx.h"

using namespace g16;

static_assert(offsetof(g16::ProofSums, B1) == sizeof(g16::G1XYZZ29) && offsetof(g16::ProofSums, L) == 2 * sizeof(g16::G1XYZZ29),
              "ProofSums must keep A, B1, L adjacent (batched reduction writes them as an array)");
static_assert(G16_PARTIAL_BYTES == g16::FIN_PARTIAL_BYTES, "partial record size out of sync");

namespace {
std::mutex g_err_mu;
std::string g_create_error;
}  // namespace


namespace {

g16_status fail(g16_ctx* ctx, g16_status code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  else {
    std::lock_guard<std::mutex> g(g_err_mu);
    g_create_error = msg;
  }
  return code;
}

template <class Fn>
g16_status guarded(g16_ctx* ctx, Fn fn) {
  try {
    if (ctx) G16_HIP(hipSetDevice(ctx->device));
    return fn();
  } catch (const HipError& e) {
    return fail(ctx, G16_ERR_HIP, e.what());
  } catch (const std::bad_alloc&) {
    return fail(ctx, G16_ERR_INTERNAL, "host allocation failed");
  } catch (const std::exception& e) {
    const bool dom = std::string(e.what()).find("PolynomialDegreeTooLarge") != std::string::npos;
    return fail(ctx, dom ? G16_ERR_DOMAIN_TOO_LARGE : G16_ERR_INTERNAL, e.what());
  }
}

void shard(uint32_t len, int rank, int world, uint32_t* lo, uint32_t* hi) {
  *lo = (uint32_t)((uint64_t)len * rank / world);
  *hi = (uint32_t)((uint64_t)len * (rank + 1) / world);
}

// work1 keeps three partial-sum slots alive (A, B1, L accumulated before one batched reduction, or
// reduced off the main stream) for small bucket sets and sharded ranks, two otherwise
int work1_batch(const MsmConfig& cw, uint32_t wr, bool sharded) {
  return (cw.nb() / wr < (1u << 18) || sharded) ? 3 : 2;
}

// Device bytes the MSM state of one ctx takes for the configurations (cw: the four witness-scalar
// queries, ch: the H query): point planes, both sorts, the partial-sum workspaces -- the same
// arithmetic as the allocations in ctx_create_impl (MsmSort::bytes_for, msm_work_bytes).
// own_w / own_h: false when the planes are borrowed from another ctx.
size_t msm_state_bytes(const MsmConfig& cw, const MsmConfig& ch, uint32_t lw, uint32_t l_cnt, uint32_t lh,
                       uint32_t wr, bool sharded, bool own_w, bool own_h, bool view_w) {
  size_t b = 0;
  if (view_w) b += MsmSort::view_bytes_for(lw, cw);  // the filtered view of the witness sort (sparse B queries)
  if (own_w) b += (size_t)cw.Pn * ((size_t)lw * (64 * 2 + 128) + (size_t)l_cnt * 64);
  if (own_h) b += (size_t)ch.Pn * lh * 64;
  b += MsmSort::bytes_for(lw, cw) + MsmSort::bytes_for(lh, ch);
  const uint32_t nc_w = ceil_div(cw.B, msm_red_chunk(cw, 1, wr)) * cw.D;
  const uint32_t nc_h = ceil_div(ch.B, msm_red_chunk(ch)) * ch.D;
  const uint32_t slots_w = cw.nb() + cw.max_lanes(), slots_h = ch.nb() + ch.max_lanes();
  b += msm_work_bytes<Fq>(slots_w, nc_w, cw.D, work1_batch(cw, wr, sharded)) + msm_work_bytes<Fq2>(slots_w, nc_w, cw.D, 1) +
       msm_work_bytes<Fq>(slots_h, nc_h, ch.D, 1);
  return b;
}

// The memory plan of a ctx: full plane precomputation (D = 1: one bucket set per MSM) when it fits
// what is free on the device, otherwise the pair of plane counts (witness queries: 320 B per point and
// plane; H query: 64 B) that fits with the fewest bucket sets to reduce per proof -- 5 D_w + D_h: the
// four witness-scalar MSMs (the G2 one counted twice) against the one H MSM; D > 1 sets are folded by
// k_horner.  Every byte the MSM state allocates is in the estimate (round 3 budgeted the planes
// against 70 % of the free memory and nothing else); what stays out is a margin of 2 GiB + 2 % for
// allocator granularity and the runtime's own needs.  Domains the reference accepts (n <= 2^27,
// qap.rs:30-32,63-68) are refused for memory only when not even ONE plane per point fits.
void plan_msm_configs(const g16_options& o, uint32_t lw, uint32_t l_cnt, uint32_t lh, uint32_t wr, bool sharded,
                      bool own_w, bool own_h, bool view_w, MsmConfig* cw, MsmConfig* ch) {
  if (own_w) *cw = msm_make_config(lw ? lw : 1, o.window_bits, o.planes);
  if (own_h) *ch = msm_make_config(lh ? lh : 1, o.window_bits, o.planes);
  if (o.planes > 0) return;  // the caller's choice: allocation failures are reported as such
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) return;
  // margin: 2 GiB + 2 % on a device with room, never more than a quarter of what is free (a box whose
  // memory is mostly taken -- the host framework's caching allocator, other ctxs -- still serves a small key)
  const size_t margin = std::min(((size_t)2 << 30) + fr / 50, fr / 4);
  const size_t budget = fr - margin;
  if (msm_state_bytes(*cw, *ch, lw, l_cnt, lh, wr, sharded, own_w, own_h, view_w) <= budget) return;  // full planes fit
  // the distinct (D, Pn) layouts of each side, most planes first
  auto layouts = [&](const MsmConfig& full, uint32_t len, bool own) {
    std::vector<MsmConfig> v{full};
    for (int pn = full.Pn - 1; own && pn >= 1; --pn) {
      const MsmConfig c = msm_make_config(len ? len : 1, o.window_bits, pn);
      if (c.Pn != v.back().Pn) v.push_back(c);
    }
    return v;
  };
  const std::vector<MsmConfig> vw = layouts(*cw, lw, own_w), vh = layouts(*ch, lh, own_h);
  long best = -1;
  size_t best_bytes = 0;
  for (const MsmConfig& a : vw)
    for (const MsmConfig& b : vh) {
      const size_t need = msm_state_bytes(a, b, lw, l_cnt, lh, wr, sharded, own_w, own_h, view_w);
      if (need > budget) continue;
      const long score = 5L * a.D + b.D;
      if (best < 0 || score < best || (score == best && need > best_bytes)) {
        best = score;
        best_bytes = need;
        *cw = a;
        *ch = b;
      }
    }
  if (best < 0) {
    if (!own_w && !own_h)  // a borrower owns no planes: what does not fit is its own sort / workspace state
      throw std::runtime_error("the sort and partial-sum state of this ctx does not fit the device memory that is free (" +
                               std::to_string(fr >> 20) + " MiB)");
    throw std::runtime_error("the proving key does not fit this device's memory even with one plane per point (" +
                             std::to_string(fr >> 20) + " MiB free)");
  }
}

// G16_SCHED_R5=1 (diagnostic, A/B): the round-5 schedule -- witness map always started beside the witness sort, the
// L reduction and B's assembly of a mid-sized proof queued on the `red` stream behind each other
inline bool sched_r5() {
  static const bool v = [] { const char* e = getenv("G16_SCHED_R5"); return e && atoi(e) != 0; }();
  return v;
}

void collect_times(g16_ctx* c) {
  if (c->timer.enabled) c->timer.collect(c->st_ms, c->st_cnt);
}

// A and B1 accumulations into work1 slots 0 and 1: one launch over the interleaved pair, or two
void accumulate_ab(g16_ctx* c, hipStream_t s, StageTimer* tm, bool fixup = true) {
  if (c->ptsA.stride == 2) {
    msm_accumulate_pair<Fq>(c->sort_w, c->ptsA, c->ptsB1, c->work1, 0, s, tm, fixup);
  } else {
    msm_accumulate<Fq>(c->sort_w, c->ptsA, 0, c->work1, 0, s, tm, fixup);
    msm_accumulate<Fq>(c->sort_w, c->ptsB1, 0, c->work1, 1, s, tm, fixup);
  }
}
// the deferred exact additions of accumulate_ab(fixup = false), on the stream that reduces A and B1
void fixup_ab(g16_ctx* c, hipStream_t q) {
  StageTimer* tm = c->timer.enabled ? &c->timer : nullptr;
  if (c->ptsA.stride == 2) {
    msm_fixup_pair<Fq>(c->sort_w, c->ptsA, c->ptsB1, c->work1, 0, q, tm);
  } else {
    msm_fixup<Fq>(c->sort_w, c->ptsA, 0, c->work1, 0, q, tm);
    msm_fixup<Fq>(c->sort_w, c->ptsB1, 0, c->work1, 1, q, tm);
  }
}

// main stream: witness-scalar sort, then the A, B1, L, B2 MSMs (ALU bound).  `after_ab` is called
// once the A and B1 sums are enqueued: the provers fork the variable-base part of the
// finalisation onto the side stream there.
void enqueue_witness_sort(g16_ctx* c, const Fr* w_dev) {
  hipStream_t s = c->stream;
  StageTimer* tm = c->timer.enabled ? &c->timer : nullptr;
  int id = tm ? tm->begin(ST_MSM_SORT, s) : -1;
  c->sort_w.run(w_dev + 1 + c->w_lo, c->w_hi - c->w_lo, /*mont=*/true, s);
  if (tm) tm->end(id, s);
  if (c->sparse_b) {
    // the filtered view is built beside the A | B1 accumulation (which reads the full sort): on the `red`
    // stream; the B2 launch waits for ev_view (wait_for_b_view)
    hipStream_t v = c->overlap ? c->red : s;
    G16_HIP(hipEventRecord(c->ev_view, s));
    G16_HIP(hipStreamWaitEvent(v, c->ev_view, 0));
    id = tm ? tm->begin(ST_MSM_SORT, v) : -1;
    c->sort_b.run_view(c->sort_w, c->keep_b, v);
    if (tm) tm->end(id, v);
    G16_HIP(hipEventRecord(c->ev_view, v));
  }
}
// before the first launch that reads the filtered B view
void wait_for_b_view(g16_ctx* c, hipStream_t s) {
  if (c->sparse_b) G16_HIP(hipStreamWaitEvent(s, c->ev_view, 0));
}

template <class Hook, class Hook2>
void enqueue_witness_msms(g16_ctx* c, const Fr* w_dev, Hook after_ab, Hook2 after_b2, bool sorted = false) {
  hipStream_t s = c->stream;
  StageTimer* tm = c->timer.enabled ? &c->timer : nullptr;
  ProofSums* S = c->sums_dev.p;
  if (!sorted) enqueue_witness_sort(c, w_dev);
  // buckets this ctx reduces per MSM: 1/world of the set under bucket-range sharding
  const uint32_t nb_eff = c->cfg_w.nb() / (c->shard_buckets ? (uint32_t)c->world : 1u);
  const bool small = nb_eff < (1u << 18);
  const uint32_t b2_limit = c->world > 1 ? (1u << 18) : (1u << 16);
  const bool b2_off = c->overlap && nb_eff < b2_limit;
  // Sharded ranks (either cut): the reductions always leave the main stream, whatever the size of the
  // bucket set.  A rank's chip is kept full by the distributed witness map on the aux stream, so a
  // reduction on the main stream is exposed latency there while its work costs the same issue slots
  // either way: one rank of 8 at 2^24 (2^19 buckets, point ranges), same box, medians of 7 proofs:
  // 21.03 / 20.91 / 21.18 ms on the main stream, 20.57 / 20.64 ms off it (profiles/r04_proj_k24_knob_sweep2.json);
  // one batched reduction of A, B1, L instead: 22.0 ms.
  const bool sharded = c->world > 1 || c->dist_wm;
  const bool mid = (nb_eff >= (1u << 15) && small) || (sharded && nb_eff >= (1u << 15));
  if (mid && c->overlap && c->work1.batch >= 3) {
    // Mid-sized bucket sets (2^15..2^17: 2^18..2^20-constraint proofs, ranks of a sharded 2^22
    // one): every reduction is a latency-bound chain long enough to matter and short enough to
    // hide, so none stays on the main stream -- it only accumulates (A, B1, B2, L, then H) and the
    // `red` stream reduces each result underneath the next accumulation.  Order: A and B1 first
    // (the variable-base products of the finalisation need them early), B2 next (the longest
    // reduction), L last.  Same-box A/B, ms per proof without / with: 2^17 4.21 / 4.29 (smaller
    // sets: three reductions in a row outlast the accumulations, the batched path below wins),
    // 2^18 5.25 / 5.06, 2^19 8.03 / 7.46, 2^20 13.46 / 12.49, 2^21 23.1 / 23.1 (larger sets: the
    // reductions take issue slots from a saturated accumulation).
    // The single-block fix-up of the optimistic G1 kernel goes with the reduction: on the main
    // stream it would sit between two accumulations and wait for a wave slot of a chip that the
    // red / aux streams keep busy (0.4 ms in a rank's timeline).  Same box, fix-up inline / on the
    // reducing stream: 2^20 proof 11.6-11.8 / 11.4 ms, one point-sharded rank of 8 at 2^22 7.9-8.0 / 7.8 ms
    // (round 3, same box).
    hipStream_t q = c->red;
    accumulate_ab(c, s, tm, /*fixup=*/false);
    G16_HIP(hipEventRecord(c->ev_acc[0], s));
    G16_HIP(hipStreamWaitEvent(q, c->ev_acc[0], 0));
    fixup_ab(c, q);
    msm_reduce<Fq>(c->sort_w, c->work1, 0, 2, &S->A, q, tm, /*hidden=*/true);
    after_ab(q);
    wait_for_b_view(c, s);
    msm_accumulate<Fq2>(c->sort_for_b(), c->ptsB2, 0, c->work2, 0, s, tm);
    G16_HIP(hipEventRecord(c->ev_acc[1], s));
    G16_HIP(hipStreamWaitEvent(q, c->ev_acc[1], 0));
    msm_reduce<Fq2>(c->sort_for_b(), c->work2, 0, 1, &S->B2, q, tm, /*hidden=*/true);
    // Round 6 (2^20 timeline, profiles/r06_timeline_k20.txt): B's assembly (fin_b, 0.25 ms on one lane) used to
    // queue on `red` BEHIND the L reduction and ended 0.3 ms after the last reduction of the proof -- fin_final
    // waited for it.  Now it follows the B2 reduction at once and the L reduction takes the side stream (idle
    // since the variable-base products finished), so the three tails -- B, L, H -- run side by side.
    const bool r5 = sched_r5();
    hipStream_t ql = r5 ? q : c->side;
    if (!r5) after_b2();
    msm_accumulate<Fq>(c->sort_w, c->ptsL, c->l_idx_min, c->work1, 2, s, tm, /*fixup=*/false);
    G16_HIP(hipEventRecord(c->ev_acc[2], s));
    G16_HIP(hipStreamWaitEvent(ql, c->ev_acc[2], 0));
    msm_fixup<Fq>(c->sort_w, c->ptsL, c->l_idx_min, c->work1, 2, ql, tm);
    msm_reduce<Fq>(c->sort_w, c->work1, 2, 1, &S->L, ql, tm, /*hidden=*/true);
    if (r5) after_b2();
    G16_HIP(hipEventRecord(c->ev_b2, q));
    G16_HIP(hipStreamWaitEvent(c->side, c->ev_b2, 0));  // the side stream joins: one event to wait on
    G16_HIP(hipEventRecord(c->ev_side, c->side));
    return;
  }
  if (small && c->work1.batch >= 3) {
    // A, B1, L share the witness sort: three accumulations, ONE batched bucket reduction.  With
    // few buckets the reduction is pure latency (~0.4 ms of dependent EC additions whatever the
    // size): paying it once instead of three times is worth 20 % of a 2^16 proof and of a rank's
    // share of a sharded 2^22 proof.  ProofSums keeps A, B1, L adjacent.
    accumulate_ab(c, s, tm);
    msm_accumulate<Fq>(c->sort_w, c->ptsL, c->l_idx_min, c->work1, 2, s, tm);
    msm_reduce<Fq>(c->sort_w, c->work1, 0, 3, &S->A, s, tm);
    after_ab(s);
  } else {
    // large bucket sets: the reduction is throughput bound, and reducing A and B1 at once lets
    // the variable-base part of the finalisation start ~10 ms earlier (measured at 2^22: 41.3 vs
    // 43.0 ms per proof)
    accumulate_ab(c, s, tm);
    msm_reduce<Fq>(c->sort_w, c->work1, 0, 2, &S->A, s, tm);  // ProofSums keeps A, B1 adjacent
    after_ab(s);
    // (round 4, VERDICT r3 item 5: taking the L reduction off the main stream -- beside the H
    // accumulation, or beside the H reduction -- was built and measured: 37.67 / 37.50 ms shipped vs
    // 37.56 / 37.59 and 37.77 / 37.46, same box; profiles/r04_defer_l_reduction_ab.txt.  Not kept.)
    msm_run<Fq>(c->sort_w, c->ptsL, c->l_idx_min, c->work1, &S->L, s, tm);
  }
  // B2: accumulate here; with small bucket sets its reduction (a latency-bound chain of Fq2 point
  // additions) runs on its own stream underneath the H MSM -- with large ones it only takes VALU
  // slots from it.  Measured on one box, batched reduction / own stream for the B2 reduction:
  //   2^14: 4.60 ms neither, 3.84 batched, 3.64 both;  2^18: 6.74 / 5.61 / 5.23;
  //   2^20: 13.56 / 13.05 / 13.21;  2^22: 40.1 / 41.9 / -.   ev_side = "everything forked is done".
  wait_for_b_view(c, s);
  msm_accumulate<Fq2>(c->sort_for_b(), c->ptsB2, 0, c->work2, 0, s, tm);
  // sharded ranks: the main stream is the critical path (the witness-map phases and exchanges hide
  // under it), so the B2 reduction leaves it whenever the bucket set is small
  hipStream_t rs = b2_off ? c->red : s;
  G16_HIP(hipEventRecord(c->ev_b2, s));
  G16_HIP(hipStreamWaitEvent(rs, c->ev_b2, 0));
  msm_reduce<Fq2>(c->sort_for_b(), c->work2, 0, 1, &S->B2, rs, tm);
  // what only needs the B2 sum continues on the `red` stream (never on the main stream: it is a
  // single-lane chain)
  G16_HIP(hipEventRecord(c->ev_b2, rs));
  if (rs != c->red) G16_HIP(hipStreamWaitEvent(c->red, c->ev_b2, 0));
  after_b2();
  G16_HIP(hipEventRecord(c->ev_b2, c->red));
  G16_HIP(hipStreamWaitEvent(c->side, c->ev_b2, 0));  // the side stream joins it: one event to wait on
  G16_HIP(hipEventRecord(c->ev_side, c->side));
}

// main stream: H MSM once the aux stream has produced (and sorted) this rank's h scalars.
// (Keeping the whole H chain on the aux stream, in parallel with the witness-scalar chain, was
// measured on one box at 2^14..2^22: 1-5 % slower at small sizes, neutral at large ones.)
void enqueue_h_msm(g16_ctx* c) {
  hipStream_t s = c->stream;
  StageTimer* tm = c->timer.enabled ? &c->timer : nullptr;
  G16_HIP(hipStreamWaitEvent(s, c->ev_h, 0));
  msm_run<Fq>(c->sort_h, c->ptsH, 0, c->workH, &c->sums_dev.p->H, s, tm);
}

// MSMs of one proof on this ctx's shard; results left in sums_dev.
// aux stream: witness map (integer-ALU bound) then the H-query sort (atomics/HBM bound), beside
// the main stream's work.
template <class Hook, class Hook2>
void run_msms(g16_ctx* c, const Fr* w_dev, Hook after_ab, Hook2 after_b2) {
  hipStream_t s = c->stream, x = c->overlap ? c->aux : c->stream;
  StageTimer* tm = c->timer.enabled ? &c->timer : nullptr;
  // Round 6: with LARGE bucket sets (>= 2^18: 2^21-constraint proofs and up, the reductions stay on the main
  // stream) the witness sort goes FIRST and the witness map waits for it -- the high-priority aux stream's NTT
  // passes otherwise take the chip from the sort kernels and the first accumulation starts later; h is not needed
  // before the last MSM.  Same box, map first / sort first (profiles/r06_schedule_ab.txt): 2^22 36.47, 36.30 /
  // 35.79, 36.11 ms; 2^21 equal.  Mid-sized proofs keep the map first: behind the sort it runs wholly under the
  // accumulations and costs them more than the earlier start returns (2^19 6.6 / 6.9, 2^20 11.2 / 11.4, Poseidon
  // 2^20 10.9 / 11.2 ms).
  const uint32_t nb_eff = c->cfg_w.nb() / (c->shard_buckets ? (uint32_t)c->world : 1u);
  // Up to 2^24: above, the witness map is what the H MSM waits for and must not start later
  // (2^25: 265.3 ms map first, 267.8 sort first; 2^23 / 2^24 equal: profiles/r06_schedule_ab.txt block 5).
  const bool sort_first = !sched_r5() && c->overlap && c->world == 1 && nb_eff >= (1u << 18) && c->n <= (1u << 24);
  if (sort_first) enqueue_witness_sort(c, w_dev);
  G16_HIP(hipEventRecord(c->ev_w, s));  // w is resident (upload enqueued on the main stream) [and sorted]
  G16_HIP(hipStreamWaitEvent(x, c->ev_w, 0));
  int id = tm ? tm->begin(ST_WITNESS_MAP, x) : -1;
  c->wm.run(w_dev, c->h_canon.p, nullptr, x);
  if (tm) tm->end(id, x);
  id = tm ? tm->begin(ST_MSM_SORT, x) : -1;
  c->sort_h.run(c->h_canon.p + c->h_lo, c->h_hi - c->h_lo, /*mont=*/false, x);
  if (tm) tm->end(id, x);
  G16_HIP(hipEventRecord(c->ev_h, x));
  enqueue_witness_msms(c, w_dev, after_ab, after_b2, /*sorted=*/sort_first);
  enqueue_h_msm(c);
}

template <class Hook>
void run_msms(g16_ctx* c, const Fr* w_dev, Hook after_ab) {
  run_msms(c, w_dev, after_ab, [] {});
}

// sharded provers: upload (r, s) and start the r/s-only fixed-base sums on the side stream
void begin_sharded(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4]) {
  hipStream_t s = c->stream;
  memcpy(c->fixed_rs, r, 32);
  memcpy(c->fixed_rs + 4, s_, 32);
  G16_HIP(hipMemcpyAsync(c->rs_dev.p, c->fixed_rs, 64, hipMemcpyHostToDevice, s));
  G16_HIP(hipEventRecord(c->ev_start, s));
  G16_HIP(hipStreamWaitEvent(c->side, c->ev_start, 0));
  fin_fixed_dist(c->fin_tab.p, c->rs_dev.p, c->fin_scr.p, c->side);
  c->fixed_ready = true;
}

g16_status check_w(g16_ctx* c, size_t n_vars) {
  if (n_vars != c->N) return fail(c, G16_ERR_INVALID, "witness length != n_vars of the key");
  return G16_OK;
}

}  // namespace

namespace g16 {

void rank_collect_times(g16_ctx* c) { collect_times(c); }

// Small keys: the same proof through the fixed-base tables (msm_table.h).  Main stream: the five
// witness-scalar G1 sums (A, B1, L, s A, r B1); aux: witness map -> H; red: B2; side: the (r, s)-only
// fixed-base sums; then the finalisation the sharded provers use (no variable-base product left).
static void enqueue_prove_tables(g16_ctx* c, const Fr* w_dev) {
  hipStream_t s = c->stream;
  // q carries the G2 sum -- the longest dependent chain of the proof (Fq2 additions) -- on the ctx's
  // HIGH-priority stream (aux), so that its workgroups are placed before the G1 launches' when the chip
  // is full; the witness map -> H chain takes the red stream here
  hipStream_t x = c->overlap ? c->red : s, q = c->overlap ? c->aux : s, sd = c->overlap ? c->side : s;
  StageTimer* tm = c->timer.enabled ? &c->timer : nullptr;
  ProofSums* S = c->sums_dev.p;
  G16_HIP(hipMemcpyAsync(c->rs_dev.p, c->pin_io, 64, hipMemcpyHostToDevice, s));
  G16_HIP(hipEventRecord(c->ev_start, s));  // (r, s) and the witness are resident
  // the G2 sum first
  G16_HIP(hipStreamWaitEvent(q, c->ev_start, 0));
  int id = tm ? tm->begin(ST_MSM_TABLE_G2, q) : -1;
  c->tbl.run_g2_witness(w_dev + 1, S, q);
  if (tm) tm->end(id, q);
  // witness map, then the H sum
  G16_HIP(hipStreamWaitEvent(x, c->ev_start, 0));
  id = tm ? tm->begin(ST_WITNESS_MAP, x) : -1;
  c->wm.run(w_dev, c->h_canon.p, nullptr, x);
  if (tm) tm->end(id, x);
  id = tm ? tm->begin(ST_MSM_TABLE_G1, x) : -1;
  c->tbl.run_h(c->h_canon.p, S, x);
  if (tm) tm->end(id, x);
  G16_HIP(hipEventRecord(c->ev_h, x));
  // side: the (r, s)-only fixed-base sums and what can be added up front
  G16_HIP(hipStreamWaitEvent(sd, c->ev_start, 0));
  fin_fixed_dist(c->fin_tab.p, c->rs_dev.p, c->fin_scr.p, sd);
  fin_tab_pre(c->key_dev.p, c->fin_scr.p, sd);
  G16_HIP(hipEventRecord(c->ev_fixed, sd));
  // behind the G2 sum: B = b' + MSM_B2
  G16_HIP(hipStreamWaitEvent(q, c->ev_fixed, 0));
  fin_tab_b(S, c->fin_scr.p, c->part_dev(), q);
  G16_HIP(hipEventRecord(c->ev_b2, q));
  // main: the five witness-scalar G1 sums, A, the H-free part of C; then C behind the H sum
  id = tm ? tm->begin(ST_MSM_TABLE_G1, s) : -1;
  c->tbl.run_g1_witness(w_dev + 1, c->rs_dev.p, S, s);
  if (tm) tm->end(id, s);
  G16_HIP(hipStreamWaitEvent(s, c->ev_fixed, 0));
  id = tm ? tm->begin(ST_FINALIZE, s) : -1;
  fin_tab_ac(S, c->fin_scr.p, c->part_dev(), s);
  G16_HIP(hipStreamWaitEvent(s, c->ev_h, 0));
  fin_tab_c(S, c->fin_scr.p, c->part_dev(), s);
  if (tm) tm->end(id, s);
  G16_HIP(hipStreamWaitEvent(s, c->ev_b2, 0));
  // A, B, C in XYZZ form (the partial-record slot of out_dev is free on a world = 1 ctx); proof_from_pin() divides
  G16_HIP(hipMemcpyAsync(c->pin_io + 64 + G16_PROOF_BYTES, c->part_dev(), FIN_PROJ_BYTES, hipMemcpyDeviceToHost, s));
}

// after the main stream is synchronised: the proof bytes of the last enqueue_prove
static void proof_from_pin(g16_ctx* c, uint8_t* proof_out) {
  if (c->tbl.active) {
    fin_tab_host_affine(c->pin_io + 64 + G16_PROOF_BYTES, proof_out);
  } else {
    memcpy(proof_out, c->pin_io + 64, G16_PROOF_BYTES);  // A and B were converted on the device, off the critical path
    fin_host_affine_c(c->pin_io + 64 + G16_PROOF_BYTES, proof_out);  // C: the host divides (fin_final_proj)
  }
}

// one whole single-device proof, enqueue only: (r, s) from and the proof to the ctx's pinned buffer
static void enqueue_prove(g16_ctx* c, const Fr* w_dev) {
  if (c->tbl.active) return enqueue_prove_tables(c, w_dev);
  hipStream_t s = c->stream;
  G16_HIP(hipMemcpyAsync(c->rs_dev.p, c->pin_io, 64, hipMemcpyHostToDevice, s));
  // fork: the (r, s)-only part of the finalisation runs beside the witness map / MSMs
  G16_HIP(hipEventRecord(c->ev_start, s));
  G16_HIP(hipStreamWaitEvent(c->side, c->ev_start, 0));
  fin_fixed(c->fin_tab.p, c->rs_dev.p, c->fin_scr.p, c->side);
  G16_HIP(hipEventRecord(c->ev_fixed, c->side));
  run_msms(
      c, w_dev,
      [&](hipStream_t from) {
        // A and B1 are enqueued: g_a, g1_b and the two variable-base products overlap L / B2 / H
        G16_HIP(hipEventRecord(c->ev_ab, from));
        G16_HIP(hipStreamWaitEvent(c->side, c->ev_ab, 0));
        fin_var(c->key_dev.p, c->sums_dev.p, c->rs_dev.p, c->fin_scr.p, c->out_dev.p, c->side);
      },
      [&] {
        // the B2 sum is there: B (assembly + Fq2 inversion) hides under the H MSM
        G16_HIP(hipStreamWaitEvent(c->red, c->ev_fixed, 0));
        fin_b(c->key_dev.p, c->sums_dev.p, c->fin_scr.p, c->out_dev.p, c->red);
      });
  G16_HIP(hipStreamWaitEvent(s, c->ev_side, 0));  // join
  int id = c->timer.enabled ? c->timer.begin(ST_FINALIZE, s) : -1;
  // C leaves the device in XYZZ form (the partial-record slot of out_dev is free on a world = 1 ctx): the host divides
  fin_final_proj(c->key_dev.p, c->sums_dev.p, c->fin_scr.p, c->part_dev(), s);
  c->timer.end(id, s);
  G16_HIP(hipMemcpyAsync(c->pin_io + 64, c->out_dev.p, G16_PROOF_BYTES, hipMemcpyDeviceToHost, s));
  G16_HIP(hipMemcpyAsync(c->pin_io + 64 + G16_PROOF_BYTES + FIN_PROJ_C, c->part_dev() + FIN_PROJ_C, sizeof(XYZZ<Fq>),
                         hipMemcpyDeviceToHost, s));
}

// this rank's r*A-side products: s*A and r*B1 (variable-base, one wave each) overlap its L / B2 / H MSMs
static void fork_partial_var(g16_ctx* c, hipStream_t from) {
  G16_HIP(hipEventRecord(c->ev_ab, from));
  G16_HIP(hipStreamWaitEvent(c->side, c->ev_ab, 0));
  fin_partial_var(c->sums_dev.p, c->rs_dev.p, c->side);
}

void rank_partial_enqueue(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4], const Fr* w_dev) {
  G16_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  begin_sharded(c, r, s_);
  run_msms(c, w_dev, [&](hipStream_t from) { fork_partial_var(c, from); });
  G16_HIP(hipStreamWaitEvent(s, c->ev_side, 0));
  sums_to_partial(c->sums_dev.p, c->part_dev(), s);
  G16_HIP(hipEventRecord(c->ev_part, s));
}

void rank_phase1_enqueue(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4], const Fr* w_dev,
                         int32_t* send_dev) {
  G16_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream, x = c->aux;
  begin_sharded(c, r, s_);
  G16_HIP(hipEventRecord(c->ev_w, s));
  G16_HIP(hipStreamWaitEvent(x, c->ev_w, 0));
  c->wd.phase1(w_dev, send_dev, x);
  G16_HIP(hipEventRecord(c->ev_send, x));
  // The witness-scalar MSMs of this rank run on the main stream during both exchanges and phases
  // 2-3.  A persistent accumulation grid only hands wave slots to the aux stream when one of its
  // rounds retires, so phases 2-3 stretch fourfold underneath (profiles/r03_rank8_timeline_*).
  // (Holding the first accumulation back until phase 2 is through was measured in round 3 -- 7.98 vs
  // 7.76 ms per rank at 2^22 / 8, equal at 2^24, profiles/r03_rank_schedule_ab.txt -- and removed.)
  enqueue_witness_msms(c, w_dev, [&](hipStream_t from) { fork_partial_var(c, from); }, [] {});
}

void rank_phase2_enqueue(g16_ctx* c, const int32_t* recv_dev, int32_t* send_dev) {
  G16_HIP(hipSetDevice(c->device));
  c->wd.phase2(recv_dev, send_dev, c->aux);
  G16_HIP(hipEventRecord(c->ev_send, c->aux));
}

void rank_phase3_enqueue(g16_ctx* c, const int32_t* recv_dev) {
  G16_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream, x = c->aux;
  c->wd.phase3(recv_dev, c->h_canon.p, x);
  c->sort_h.run(c->h_canon.p, c->h_hi - c->h_lo, /*mont=*/false, x);
  G16_HIP(hipEventRecord(c->ev_h, x));
  enqueue_h_msm(c);
  G16_HIP(hipStreamWaitEvent(s, c->ev_side, 0));
  sums_to_partial(c->sums_dev.p, c->part_dev(), s);
  G16_HIP(hipEventRecord(c->ev_part, s));
}

void rank_finish_enqueue(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4], int world) {
  G16_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  uint64_t rs[8];
  memcpy(rs, r, 32);
  memcpy(rs + 4, s_, 32);
  partials_to_sums(c->gathered_dev(), world, c->sums_dev.p, s);
  // the r/s-only sums were started by this ctx's partial / phase-1 call when (r, s) match (side
  // stream; ev_side was joined by the main stream before the record was written)
  if (!(c->fixed_ready && memcmp(c->fixed_rs, rs, 64) == 0)) {
    memcpy(c->fixed_rs, rs, 64);
    G16_HIP(hipMemcpyAsync(c->rs_dev.p, c->fixed_rs, 64, hipMemcpyHostToDevice, s));
    fin_fixed_dist(c->fin_tab.p, c->rs_dev.p, c->fin_scr.p, s);
  }
  c->fixed_ready = false;
  fin_final_dist(c->key_dev.p, c->sums_dev.p, c->fin_scr.p, c->out_dev.p, s);
}

}  // namespace g16

extern "C" {

const char* g16_last_error(const g16_ctx* ctx) {
  if (ctx) return ctx->err.c_str();
  std::lock_guard<std::mutex> g(g_err_mu);
  static thread_local std::string copy;
  copy = g_create_error;
  return copy.c_str();
}

}  // extern "C"

namespace g16 {


g16_status ctx_create_impl(const g16_key_desc* key, const g16_csr* a, const g16_csr* b,
                           uint32_t num_constraints, const g16_options* opt, g16_ctx* share_from,
                           g16_ctx** out, std::string* err) {
  auto bad = [&](g16_status code, const std::string& msg) {
    if (err) *err = msg;
    return code;
  };
  if (!key || !a || !b || !out) return bad(G16_ERR_INVALID, "null argument");
  *out = nullptr;
  g16_options o{};
  if (opt) o = *opt;
  if (o.world <= 0) o.world = 1;
  if (o.rank < 0 || o.rank >= o.world) return bad(G16_ERR_INVALID, "bad rank/world");
  if (o.shard < G16_SHARD_AUTO || o.shard > G16_SHARD_BUCKETS) return bad(G16_ERR_INVALID, "unknown shard mode");
  if ((uint64_t)key->n_vars < (uint64_t)key->n_public + 1)
    return bad(G16_ERR_INVALID, "n_vars < n_public+1");
  // the kernels index w[col[j]] and col/coeff[row_ptr[i] .. row_ptr[i+1]) unchecked: validate once
  // here (the reference panics on an out-of-bounds wire index in evaluate_constraint)
  for (const g16_csr* mtx : {a, b}) {
    if (!mtx->row_ptr || (mtx->nnz && (!mtx->col || !mtx->coeff)))
      return bad(G16_ERR_INVALID, "matrix with null arrays");
    if (mtx->row_ptr[0] != 0 || mtx->row_ptr[num_constraints] != mtx->nnz)
      return bad(G16_ERR_INVALID, "matrix row_ptr does not span [0, nnz]");
    for (uint32_t i = 0; i < num_constraints; ++i)
      if (mtx->row_ptr[i] > mtx->row_ptr[i + 1])
        return bad(G16_ERR_INVALID, "matrix row_ptr is not monotone");
    for (uint64_t j = 0; j < mtx->nnz; ++j)
      if (mtx->col[j] >= key->n_vars)
        return bad(G16_ERR_INVALID, "matrix wire index >= n_vars");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return bad(G16_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
  if (o.device < 0 || o.device >= ndev) return bad(G16_ERR_INVALID, "bad device ordinal");
  // a lender's planes are whole-key planes: either a bucket-sharded rank's (multi-device ctx with
  // repeated ordinals) or a plain single-device ctx's (g16_ctx_create_sibling: two proofs in flight)
  if (share_from && (share_from->device != o.device || !share_from->has_key || share_from->multi ||
                     !(share_from->shard_buckets || (share_from->world == 1 && o.world == 1 && !share_from->dist_wm && o.dist_wm <= 0))))
    return bad(G16_ERR_INVALID, "planes can only be shared with a whole-key ctx on the same device");

  g16_ctx* c = new (std::nothrow) g16_ctx();
  if (!c) return bad(G16_ERR_INTERNAL, "host allocation failed");
  c->device = o.device;
  c->rank = o.rank;
  c->world = o.world;
  g16_status st = guarded(c, [&]() -> g16_status {
    // (a CU mask that keeps the main stream off some CUs for the witness map was measured in rounds 2-3
    // and changes nothing: the chip is issue bound, reserved CUs are lost to the accumulations)
    G16_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    G16_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    G16_HIP(hipStreamCreateWithFlags(&c->red, hipStreamNonBlocking));
    {
      // the aux stream carries the memory/atomic-bound work that should slip in beside the
      // ALU-bound MSM kernels: give its workgroups dispatch priority
      int lo = 0, hi = 0;
      G16_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      (void)lo;
      G16_HIP(hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, hi));
    }
    if (const char* e = getenv("G16_NO_OVERLAP")) c->overlap = !(e[0] == '1');
    G16_HIP(hipEventCreateWithFlags(&c->ev_w, hipEventDisableTiming));
    G16_HIP(hipEventCreateWithFlags(&c->ev_h, hipEventDisableTiming));
    G16_HIP(hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming));
    G16_HIP(hipEventCreateWithFlags(&c->ev_ab, hipEventDisableTiming));
    G16_HIP(hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
    G16_HIP(hipEventCreateWithFlags(&c->ev_b2, hipEventDisableTiming));
    for (auto& e : c->ev_acc) G16_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    G16_HIP(hipEventCreateWithFlags(&c->ev_fixed, hipEventDisableTiming));
    G16_HIP(hipEventCreateWithFlags(&c->ev_view, hipEventDisableTiming));
    G16_HIP(hipEventCreateWithFlags(&c->ev_send, hipEventDisableTiming));
    G16_HIP(hipEventCreateWithFlags(&c->ev_part, hipEventDisableTiming));
    G16_HIP(hipEventCreateWithFlags(&c->ev_user, hipEventDisableTiming));
    hipStream_t s = c->stream;
    c->N = key->n_vars;
    c->p = key->n_public;
    c->num_inputs = c->p + 1;
    c->m = num_constraints;
    c->nnz_a = a->nnz;
    c->nnz_b = b->nnz;
    CsrHost A{a->row_ptr, a->col, (const Fr*)a->coeff, (size_t)a->nnz};
    CsrHost B{b->row_ptr, b->col, (const Fr*)b->coeff, (size_t)b->nnz};
    if (o.reduction != G16_REDUCTION_CIRCOM && o.reduction != G16_REDUCTION_LIBSNARK)
      throw std::runtime_error("unknown reduction");
    // (world = 1 with dist_wm > 0 is the degenerate one-rank case of the phase API: every exchange is
    // a copy onto itself -- what a single-process run of the host framework's collectives exercises)
    c->dist_wm = o.dist_wm > 0;
    if (c->dist_wm && o.reduction != G16_REDUCTION_CIRCOM)
      throw std::runtime_error("the distributed witness map implements CircomReduction only");
    if (c->dist_wm) {
      c->wd.init(A, B, c->m, c->num_inputs, c->rank, c->world);
      c->n = c->wd.n;
    } else {
      c->wm.init(A, B, c->m, c->num_inputs, o.reduction);
      c->n = c->wm.n;
    }
    if (key->domain_size != c->n)
      throw std::runtime_error("key domain_size does not match num_constraints + num_inputs");

    const uint32_t len_w = c->N - 1;
    c->has_key = key->a_query != nullptr;
    // AUTO = point ranges.  Measured on one MI355X, one rank of 8 alone (scripts/dist_projection.py,
    // profiles/r03_proj_k22.json / _k24.json): the two cuts tie -- 7.6 / 7.8 ms per rank at 2^22, 21.0 /
    // 21.4 ms at 2^24 -- because a bucket-sharded rank walks ALL n scalars to keep an eighth of the
    // digits and that front costs what the single-GPU window saves; point ranges need 1/world of the
    // key per device instead of all of it.  DESIGN.md section 7.
    c->shard_buckets = c->world > 1 && c->has_key && o.shard == G16_SHARD_BUCKETS;  // one rank: nothing to cut
    // (a bucket-sharded rank holds ALL points of the witness queries: plan_msm_configs below plans its
    // planes like any other ctx's -- full when they fit, fewer otherwise -- round 4's separate pre-check
    // refused keys the plan can serve)
    c->share_from = (c->shard_buckets || c->world == 1) ? share_from : nullptr;
    if (c->share_from) ++c->share_from->borrowers;
    // the H query is ALWAYS cut by point range: its scalars are born sharded (the distributed
    // witness map leaves rank g the n / world evaluations e = global_index(t)), so every rank
    // multiplies its own slice; only the witness-scalar queries (A, B1, B2, L: the witness is
    // resident everywhere anyway) are cut by bucket range under G16_SHARD_BUCKETS
    shard(len_w, c->rank, c->world, &c->w_lo, &c->w_hi);
    shard(c->n, c->rank, c->world, &c->h_lo, &c->h_hi);
    if (c->dist_wm) {  // the distributed witness map leaves n / world scalars, local order
      c->h_lo = 0;
      c->h_hi = c->n / (uint32_t)c->world;
    }
    if (c->shard_buckets) {  // every rank holds every witness-query point; the sort keeps 1/world of the entries
      c->w_lo = 0;
      c->w_hi = len_w;
    }
    const uint32_t lw = c->w_hi - c->w_lo, lh = c->h_hi - c->h_lo;
    const uint32_t wr = c->shard_buckets ? (uint32_t)c->world : 1u;  // ranks sharing one bucket set

    c->w_dev.alloc(c->N);
    c->h_dev.alloc(c->n);
    c->h_canon.alloc(c->n);
    if (!c->has_key) {
      G16_HIP(hipStreamSynchronize(s));
      return G16_OK;
    }
    if (!key->b_g1_query || !key->b_g2_query || !key->h_query || (!key->l_query && c->N > c->p + 1))
      throw std::runtime_error("key descriptor has null query arrays");
    c->rs_dev.alloc(2);
    c->key_dev.alloc(1);
    c->sums_dev.alloc(1);
    c->fin_tab.alloc(1);
    c->fin_scr.alloc(1);
    c->out_dev.alloc(G16_PROOF_BYTES + G16_PARTIAL_BYTES * (size_t)(c->world + 1));
    {
      void* pin = nullptr;
      G16_HIP(hipHostMalloc(&pin, 64 + G16_PROOF_BYTES + FIN_PROJ_BYTES, 0));
      c->pin_io = (uint8_t*)pin;
    }

    g16_ctx* lender = c->share_from;
    // point planes lent by another ctx of the same key on this device (ranks of a multi-device ctx
    // that repeat a device ordinal): same configuration, views of its arrays
    auto borrow = [](auto& mine, const auto& theirs) {
      mine.cfg = theirs.cfg;
      mine.count = theirs.count;
      mine.stride = theirs.stride;
      mine.off = theirs.off;
      mine.view = theirs.data();
    };
    // MSM configurations: the four witness queries share one sort, hence one (c, W, planes); both
    // configurations are planned together against the memory that is free now (plan_msm_configs)
    const bool lend_h = lender && c->world == 1;  // a sibling of a single-device ctx: same H planes too
    // L pairs l_query[j] with w[num_inputs + j], i.e. entry index i = p + j
    const uint32_t l_first = c->w_lo > c->p ? c->w_lo : c->p;  // first global entry with an L point
    const uint32_t l_cnt = c->w_hi > l_first ? c->w_hi - l_first : 0;
    if (lender) c->cfg_w = lender->cfg_w;
    if (lend_h) c->cfg_h = lender->cfg_h;
    // Sparse B queries: count the wires whose B points are BOTH the point at infinity (b_g1_query[1 + i] and
    // b_g2_query[1 + i]: b_i(tau) times a generator each, so a well-formed key has the same pattern in both;
    // a term is only dropped when both are -- a key with a finite G2 point beside an infinite G1 one keeps
    // it, as the reference does, which reads both arrays as given).  From 1/8 of the wires on, the B2 MSM --
    // the G2 one, a third of a proof -- runs over a filtered view of the witness sort (ctx.h: sort_b):
    // single-device proving ctxs of >= 2^15 wires; G16_SPARSE_B=0 / 1 (diagnostic, tested) overrides the
    // rule.  B1 stays in the A | B1 pair launch over the full sort (un-pairing it and reducing it on its own
    // cost more launches than its third of additions is worth: profiles/r05_sparse_b_ab.txt).  Decided
    // BEFORE the memory plan, which budgets the view's entries (MsmSort::view_bytes_for).
    if (lender) {
      c->sparse_b = lender->sparse_b && c->world == 1;
      c->keep_b = lender->keep_b;
      c->b_inf_points = lender->b_inf_points;
    } else if (c->world == 1 && !c->dist_wm && lw) {
      std::vector<uint32_t> bits((lw + 31) / 32, 0u);
      uint32_t inf = 0;
      const uint8_t* b1 = key->b_g1_query + 64;
      const uint8_t* b2 = key->b_g2_query + 128;
      auto all_zero = [](const uint8_t* q, int n) {
        for (int k = 0; k < n; ++k)
          if (q[k]) return false;
        return true;
      };
      for (uint32_t i = 0; i < lw; ++i) {
        if (all_zero(b1 + (size_t)i * 64, 64) && all_zero(b2 + (size_t)i * 128, 128)) ++inf;
        else bits[i >> 5] |= 1u << (i & 31);
      }
      c->b_inf_points = inf;
      const char* e = getenv("G16_SPARSE_B");
      c->sparse_b = e ? atoi(e) != 0 : (lw >= (1u << 15) && (uint64_t)inf * 8 >= lw);
      if (c->sparse_b) {
        c->keep_b_own.alloc(bits.size());
        G16_HIP(hipMemcpyAsync(c->keep_b_own.p, bits.data(), bits.size() * 4, hipMemcpyHostToDevice, s));
        G16_HIP(hipStreamSynchronize(s));  // `bits` is read by the async upload
        c->keep_b = c->keep_b_own.p;
      }
    }
    const bool sharded = c->world > 1 || c->dist_wm;
    plan_msm_configs(o, lw, l_cnt, lh, wr, sharded, !lender, !lend_h, c->sparse_b, &c->cfg_w, &c->cfg_h);
    c->sort_w.init(lw, c->cfg_w);
    c->sort_w.set_shard(c->shard_buckets ? c->rank : 0, (int)wr);
    // A and B1 are gathered by the same (scalar, digit, bucket) entries: interleaved point by point,
    // one 128-byte line serves both (G16_NO_PAIR_AB=1: separate arrays, for A/B measurements)
    static const bool no_pair = [] { const char* e = getenv("G16_NO_PAIR_AB"); return e && atoi(e) != 0; }();
    if (c->sparse_b) {
      // the view is an optimisation: a device that cannot hold it after all (a caller-fixed plane count
      // bypasses the plan) proves over the shared sort instead
      try {
        c->sort_b.init_view(lw, c->cfg_w);
      } catch (const HipError&) {
        (void)hipGetLastError();
        c->sort_b.release_view();
        c->sparse_b = false;
      }
    }
    if (lender) {
      borrow(c->ptsA, lender->ptsA);
      borrow(c->ptsB1, lender->ptsB1);
      borrow(c->ptsB2, lender->ptsB2);
      borrow(c->ptsL, lender->ptsL);
      c->l_idx_min = lender->l_idx_min;
    } else {
      if (no_pair) {
        c->ptsA.init((const G1Affine*)key->a_query + 1 + c->w_lo, lw, c->cfg_w, s);
        c->ptsB1.init((const G1Affine*)key->b_g1_query + 1 + c->w_lo, lw, c->cfg_w, s);
      } else {
        MsmPoints<Fq>::init_pair(c->ptsA, c->ptsB1, (const G1Affine*)key->a_query + 1 + c->w_lo,
                                 (const G1Affine*)key->b_g1_query + 1 + c->w_lo, lw, c->cfg_w, s);
      }
      c->ptsB2.init((const G2Affine*)key->b_g2_query + 1 + c->w_lo, lw, c->cfg_w, s);
      c->l_idx_min = l_first - c->w_lo;
      c->ptsL.init((const G1Affine*)key->l_query + (l_first - c->p), l_cnt, c->cfg_w, s);
    }
    c->sort_h.init(lh, c->cfg_h);
    if (lend_h) {
      borrow(c->ptsH, lender->ptsH);
    } else if (c->dist_wm) {
      // this rank's h scalars are the evaluations e = global_index(t): gather the matching points
      std::vector<G1Affine> mine(lh);
      const G1Affine* hq = (const G1Affine*)key->h_query;
      // byte copies: the caller's arrays carry no alignment (zero-copy views of zkey sections start
      // at arbitrary file offsets) and G1Affine is an over-aligned type
      for (uint32_t t = 0; t < lh; ++t)
        memcpy((void*)&mine[t], (const uint8_t*)hq + (size_t)c->wd.global_index(t) * sizeof(G1Affine), sizeof(G1Affine));
      c->ptsH.init(mine.data(), lh, c->cfg_h, s);
      G16_HIP(hipStreamSynchronize(s));  // `mine` is read by the async upload
    } else {
      c->ptsH.init((const G1Affine*)key->h_query + c->h_lo, lh, c->cfg_h, s);
    }

    // Small keys (g16_options.fixed_tables): fixed-base tables, the path g16_prove takes when they
    // exist; the planes / sorts above stay for the single-MSM entry points and cost nothing at this size.
    {
      const bool can = c->world == 1 && !c->dist_wm && c->has_key;
      if (o.fixed_tables > 0 && !can)
        throw std::runtime_error("fixed_tables: only a single-device proving ctx (world = 1, no distributed witness map) can use them");
      if (lender && lend_h) {
        if (lender->tbl.active && o.fixed_tables >= 0) c->tbl.borrow(lender->tbl);
        else if (o.fixed_tables > 0) throw std::runtime_error("fixed_tables: the donor has no tables to lend");
      } else if (can && !lender && o.fixed_tables >= 0) {
        const size_t need = tbl_bytes((size_t)2 * lw + l_cnt + lh, lw);
        size_t fr = 0, tot = 0;
        G16_HIP(hipMemGetInfo(&fr, &tot));
        const bool fits = need <= fr / 3;
        const bool small = lw <= TBL_MAX_POINTS && lh <= TBL_MAX_POINTS;
        if (o.fixed_tables > 0 && !fits)
          throw std::runtime_error("fixed_tables: the tables need " + std::to_string(need >> 20) + " MiB, more than a third of the free device memory");
        if (o.fixed_tables > 0 || (small && fits)) {
          try {
            c->tbl.build(key->a_query + 64, key->b_g1_query + 64, key->b_g2_query + 128,
                         key->l_query + (size_t)(l_first - c->p) * 64, key->h_query, lw, l_first - c->w_lo, l_cnt, lh, s);
          } catch (const HipError&) {
            // automatic mode only: a device that runs out of memory while the tables are built (their
            // temporaries are not in the one-third rule) keeps the bucket path; an explicit request fails
            if (o.fixed_tables > 0) throw;
            (void)hipGetLastError();
            (void)hipStreamSynchronize(s);
            c->tbl.release();
          }
        }
      }
    }

    // workspaces: G1 over the witness sort (3 slots when the reductions are batched), G1 over the h
    // sort, G2 over the witness sort
    {
      const uint32_t nc_w = ceil_div(c->cfg_w.B, msm_red_chunk(c->cfg_w, 1, wr)) * c->cfg_w.D;
      const uint32_t nc_h = ceil_div(c->cfg_h.B, msm_red_chunk(c->cfg_h)) * c->cfg_h.D;
      const uint32_t slots_w = c->cfg_w.nb() + c->cfg_w.max_lanes(), slots_h = c->cfg_h.nb() + c->cfg_h.max_lanes();
      c->work1.init(slots_w, nc_w, c->cfg_w.D, work1_batch(c->cfg_w, wr, sharded));
      c->workH.init(slots_h, nc_h, c->cfg_h.D, 1);
      c->work2.init(slots_w, nc_w, c->cfg_w.D);
    }

    KeyHeaderDev kh;
    memcpy(&kh.alpha1, key->alpha_g1, 64);
    memcpy(&kh.beta1, key->beta_g1, 64);
    memcpy(&kh.delta1, key->delta_g1, 64);
    memcpy(&kh.a0, key->a_query, 64);
    memcpy(&kh.b1_0, key->b_g1_query, 64);
    memcpy(&kh.beta2, key->beta_g2, 128);
    memcpy(&kh.delta2, key->delta_g2, 128);
    memcpy(&kh.b2_0, key->b_g2_query, 128);
    G16_HIP(hipMemcpyAsync(c->key_dev.p, &kh, sizeof kh, hipMemcpyHostToDevice, s));
    G16_HIP(hipMemsetAsync(c->out_dev.p, 0, c->out_dev.bytes(), s));
    G16_HIP(hipMemsetAsync(c->sums_dev.p, 0, sizeof(ProofSums), s));  // all sums = infinity
    G16_HIP(hipMemsetAsync(c->fin_scr.p, 0, sizeof(FinScratch), s));
    fin_build_tables(c->key_dev.p, c->fin_tab.p, s);
    G16_HIP(hipStreamSynchronize(s));
    return G16_OK;
  });
  if (st != G16_OK) {
    if (err) *err = c->err;
    g16_ctx_destroy(c);
    return st;
  }
  *out = c;
  return G16_OK;
}

}  // namespace g16

extern "C" {

g16_status g16_ctx_create(const g16_key_desc* key, const g16_csr* a, const g16_csr* b,
                          uint32_t num_constraints, const g16_options* opt, g16_ctx** out) {
  std::string err;
  const g16_status st = ctx_create_impl(key, a, b, num_constraints, opt, nullptr, out, &err);
  if (st != G16_OK) return fail(nullptr, st, err);
  return G16_OK;
}

g16_status g16_ctx_create_sibling(g16_ctx* donor, const g16_key_desc* key, const g16_csr* a, const g16_csr* b,
                                  uint32_t num_constraints, const g16_options* opt, g16_ctx** out) {
  if (!donor) return fail(nullptr, G16_ERR_INVALID, "null argument");
  g16_options o{};
  if (opt) o = *opt;
  o.device = donor->device;
  o.rank = 0;
  o.world = 1;
  o.dist_wm = 0;
  if (donor->multi || donor->share_from || !donor->has_key)
    return fail(nullptr, G16_ERR_INVALID, "sibling: the donor must be a plain single-device proving ctx that owns its planes");
  if (key && (key->n_vars != donor->N || key->n_public != donor->p || key->domain_size != donor->n))
    return fail(nullptr, G16_ERR_INVALID, "sibling: not the donor's key");
  // the planes (and with them the window and plane count) are the donor's: a conflicting request is an
  // error, not something to ignore; so is a matrix pair of another shape (same sizes, other circuit)
  if ((o.window_bits > 0 && o.window_bits != donor->cfg_w.c) || (o.planes > 0 && o.planes != donor->cfg_w.Pn))
    return fail(nullptr, G16_ERR_INVALID, "sibling: window_bits / planes differ from the donor's configuration");
  if (num_constraints != donor->m || !a || !b || a->nnz != donor->nnz_a || b->nnz != donor->nnz_b)
    return fail(nullptr, G16_ERR_INVALID, "sibling: not the donor's constraint matrices");
  std::string err;
  const g16_status st = ctx_create_impl(key, a, b, num_constraints, &o, donor, out, &err);
  if (st != G16_OK) return fail(nullptr, st, err);
  return G16_OK;
}

// A ctx that still lends its point planes (g16_ctx_create_sibling; ranks of a multi-device ctx that
// repeat a device ordinal) is not freed under its borrowers: the handle dies for the caller, the state
// stays until the last borrower is destroyed (round 5; rounds 3-4 printed a warning and freed anyway).
static std::mutex g_lend_mu;

void g16_ctx_destroy(g16_ctx* c) {
  if (!c) return;
  g16_ctx* lender = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_lend_mu);
    if (c->borrowers.load() > 0) {
      c->zombie = true;  // the last borrower's destroy comes back here
      return;
    }
    if (c->share_from && c->share_from->borrowers.fetch_sub(1) == 1 && c->share_from->zombie) lender = c->share_from;
    c->share_from = nullptr;
  }
  struct Then {  // the lender, if this was its last borrower and its owner has already let go of it
    g16_ctx* l;
    ~Then() {
      if (l) g16_ctx_destroy(l);
    }
  } then{lender};
  if (c->multi) {  // parent of a multi-device prover: the children own all device state
    multi_destroy(c->multi);
    if (c->pinned_w) (void)hipHostFree(c->pinned_w);
    delete c;
    return;
  }
  (void)hipSetDevice(c->device);
  if (c->side) {
    (void)hipStreamSynchronize(c->side);
    (void)hipStreamDestroy(c->side);
  }
  if (c->aux) {
    (void)hipStreamSynchronize(c->aux);
    (void)hipStreamDestroy(c->aux);
  }
  if (c->red) {
    (void)hipStreamSynchronize(c->red);
    (void)hipStreamDestroy(c->red);
  }
  if (c->xs_own) {  // g16_dist_attach_rccl (the communicator itself is the host's)
    (void)hipStreamSynchronize(c->xs_own);
    (void)hipStreamDestroy(c->xs_own);
  }
  if (c->ev_w) (void)hipEventDestroy(c->ev_w);
  if (c->ev_h) (void)hipEventDestroy(c->ev_h);
  if (c->stream) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
  }
  if (c->ev_start) (void)hipEventDestroy(c->ev_start);
  if (c->ev_ab) (void)hipEventDestroy(c->ev_ab);
  if (c->ev_side) (void)hipEventDestroy(c->ev_side);
  if (c->ev_b2) (void)hipEventDestroy(c->ev_b2);
  for (auto e : c->ev_acc)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_fixed) (void)hipEventDestroy(c->ev_fixed);
  if (c->ev_view) (void)hipEventDestroy(c->ev_view);
  if (c->ev_send) (void)hipEventDestroy(c->ev_send);
  if (c->ev_part) (void)hipEventDestroy(c->ev_part);
  if (c->ev_user) (void)hipEventDestroy(c->ev_user);
  if (c->pinned_w) (void)hipHostFree(c->pinned_w);
  if (c->pin_io) (void)hipHostFree(c->pin_io);
  delete c;
}

g16_status g16_witness_map(g16_ctx* c, const uint64_t* w, size_t n_vars, uint64_t* h_out) {
  if (!c || !w || !h_out) return fail(c, G16_ERR_INVALID, "null argument");
  if (c->multi) return fail(c, G16_ERR_INVALID, "multi-device ctx proves only (g16_prove)");
  if (c->dist_wm) return fail(c, G16_ERR_INVALID, "dist_wm ctx holds 1/world of the witness map");
  if (check_w(c, n_vars) != G16_OK) return G16_ERR_INVALID;
  return guarded(c, [&]() -> g16_status {
    hipStream_t s = c->stream;
    G16_HIP(hipMemcpyAsync(c->w_dev.p, w, (size_t)c->N * 32, hipMemcpyHostToDevice, s));
    c->wm.run(c->w_dev.p, nullptr, c->h_dev.p, s);
    G16_HIP(hipMemcpyAsync(h_out, c->h_dev.p, (size_t)c->n * 32, hipMemcpyDeviceToHost, s));
    G16_HIP(hipStreamSynchronize(s));
    return G16_OK;
  });
}

g16_status g16_witness_map_dev(g16_ctx* c, const void* w_dev, size_t n_vars, void* h_dev_out) {
  if (!c || !w_dev || !h_dev_out) return fail(c, G16_ERR_INVALID, "null argument");
  if (c->multi) return fail(c, G16_ERR_INVALID, "multi-device ctx proves only (g16_prove)");
  if (c->dist_wm) return fail(c, G16_ERR_INVALID, "dist_wm ctx holds 1/world of the witness map");
  if (check_w(c, n_vars) != G16_OK) return G16_ERR_INVALID;
  return guarded(c, [&]() -> g16_status {
    c->wm.run((const Fr*)w_dev, nullptr, (Fr*)h_dev_out, c->stream);
    G16_HIP(hipStreamSynchronize(c->stream));
    return G16_OK;
  });
}

static g16_status msm_common(g16_ctx* c, int which, bool g2, const uint64_t* scalars, size_t len,
                             uint8_t* out, bool on_device = false) {
  if (!c || !scalars || !out) return fail(c, G16_ERR_INVALID, "null argument");
  if (c->multi) return fail(c, G16_ERR_INVALID, "multi-device ctx proves only (g16_prove)");
  if (!c->has_key) return fail(c, G16_ERR_INVALID, "witness-map-only ctx has no resident key");
  if (c->world != 1) return fail(c, G16_ERR_INVALID, "g16_msm_* needs a world == 1 ctx");
  return guarded(c, [&]() -> g16_status {
    hipStream_t s = c->stream;
    MsmSort* sort = &c->sort_w;
    uint32_t idx_min = 0;
    const MsmPoints<Fq>* P1 = nullptr;
    size_t maxlen = 0;
    Fr* stage = c->w_dev.p;  // scalar staging: reuse the witness / h buffers
    if (g2) {
      maxlen = c->ptsB2.count;
    } else if (which == G16_QUERY_A) {
      P1 = &c->ptsA;
      maxlen = P1->count;
    } else if (which == G16_QUERY_B1) {
      P1 = &c->ptsB1;
      maxlen = P1->count;
    } else if (which == G16_QUERY_L) {
      P1 = &c->ptsL;
      maxlen = P1->count;
    } else if (which == G16_QUERY_H) {
      P1 = &c->ptsH;
      maxlen = P1->count;
      sort = &c->sort_h;
      stage = c->h_dev.p;
    } else {
      return fail(c, G16_ERR_INVALID, "unknown query id");
    }
    if (len > maxlen) return fail(c, G16_ERR_INVALID, "more scalars than resident points");
    if (on_device) stage = (Fr*)const_cast<uint64_t*>(scalars);
    else G16_HIP(hipMemcpyAsync(stage, scalars, len * 32, hipMemcpyHostToDevice, s));
    sort->run(stage, (uint32_t)len, true, s);
    (void)idx_min;
    if (g2) {
      msm_run<Fq2>(*sort, c->ptsB2, 0, c->work2, &c->sums_dev.p->B2, s, nullptr);
    } else {
      // for L the entry index is already the l_query index here (scalars pair with l_query[i])
      msm_run<Fq>(*sort, *P1, 0, sort == &c->sort_h ? c->workH : c->work1, &c->sums_dev.p->A, s, nullptr);
    }
    sums_to_affine(c->sums_dev.p, c->out_dev.p, s);  // A -> bytes [0,64), B2 -> [128,256)
    G16_HIP(hipMemcpyAsync(out, c->out_dev.p + (g2 ? 128 : 0), g2 ? 128 : 64,
                           hipMemcpyDeviceToHost, s));
    G16_HIP(hipStreamSynchronize(s));
    return G16_OK;
  });
}

g16_status g16_msm_g1(g16_ctx* c, int which, const uint64_t* scalars, size_t len, uint8_t out[64]) {
  return msm_common(c, which, false, scalars, len, out);
}
g16_status g16_msm_g2(g16_ctx* c, const uint64_t* scalars, size_t len, uint8_t out[128]) {
  return msm_common(c, 0, true, scalars, len, out);
}
g16_status g16_msm_g1_dev(g16_ctx* c, int which, const void* scalars_dev, size_t len, uint8_t out[64]) {
  return msm_common(c, which, false, (const uint64_t*)scalars_dev, len, out, true);
}
g16_status g16_msm_g2_dev(g16_ctx* c, const void* scalars_dev, size_t len, uint8_t out[128]) {
  return msm_common(c, 0, true, (const uint64_t*)scalars_dev, len, out, true);
}

g16_status g16_prove_dev(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4], const void* w_dev,
                         size_t n_vars, uint8_t proof_out[G16_PROOF_BYTES]) {
  if (!c || !r || !s_ || !w_dev || !proof_out) return fail(c, G16_ERR_INVALID, "null argument");
  if (c->multi) {  // sharded transparently over the devices of g16_ctx_create_multi
    if (check_w(c, n_vars) != G16_OK) return G16_ERR_INVALID;
    return multi_prove(c, r, s_, w_dev, /*w_on_device=*/true, proof_out);
  }
  if (!c->has_key) return fail(c, G16_ERR_INVALID, "witness-map-only ctx has no resident key");
  if (c->world != 1) return fail(c, G16_ERR_INVALID, "g16_prove needs world == 1; use partial/finish");
  if (c->dist_wm) return fail(c, G16_ERR_INVALID, "dist_wm ctx: use the g16_prove_dist_phase* calls");
  if (check_w(c, n_vars) != G16_OK) return G16_ERR_INVALID;
  return guarded(c, [&]() -> g16_status {
    hipStream_t s = c->stream;
    memcpy(c->pin_io, r, 32);
    memcpy(c->pin_io + 32, s_, 32);
    enqueue_prove(c, (const Fr*)w_dev);
    G16_HIP(hipStreamSynchronize(s));
    proof_from_pin(c, proof_out);
    collect_times(c);
    return G16_OK;
  });
}

g16_status g16_prove(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4], const uint64_t* w,
                     size_t n_vars, uint8_t proof_out[G16_PROOF_BYTES]) {
  if (!c || !w) return fail(c, G16_ERR_INVALID, "null argument");
  if (check_w(c, n_vars) != G16_OK) return G16_ERR_INVALID;
  if (c->multi) {
    if (!r || !s_ || !proof_out) return fail(c, G16_ERR_INVALID, "null argument");
    return multi_prove(c, r, s_, w, /*w_on_device=*/false, proof_out);
  }
  g16_status st = guarded(c, [&]() -> g16_status {
    G16_HIP(hipMemcpyAsync(c->w_dev.p, w, (size_t)c->N * 32, hipMemcpyHostToDevice, c->stream));
    return G16_OK;
  });
  if (st != G16_OK) return st;
  return g16_prove_dev(c, r, s_, c->w_dev.p, n_vars, proof_out);
}

g16_status g16_prove_partial_dev(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4],
                                 const void* w_dev, size_t n_vars,
                                 uint8_t partial_out[G16_PARTIAL_BYTES]) {
  if (!c || !r || !s_ || !w_dev || !partial_out) return fail(c, G16_ERR_INVALID, "null argument");
  if (c->multi) return fail(c, G16_ERR_INVALID, "multi-device ctx: use g16_prove");
  if (!c->has_key) return fail(c, G16_ERR_INVALID, "witness-map-only ctx has no resident key");
  if (check_w(c, n_vars) != G16_OK) return G16_ERR_INVALID;
  if (c->dist_wm) return fail(c, G16_ERR_INVALID, "dist_wm ctx: use the g16_prove_dist_phase* calls");
  return guarded(c, [&]() -> g16_status {
    hipStream_t s = c->stream;
    rank_partial_enqueue(c, r, s_, (const Fr*)w_dev);
    G16_HIP(hipMemcpyAsync(partial_out, c->part_dev(), G16_PARTIAL_BYTES, hipMemcpyDeviceToHost, s));
    G16_HIP(hipStreamSynchronize(s));
    collect_times(c);
    return G16_OK;
  });
}

g16_status g16_prove_partial(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4],
                             const uint64_t* w, size_t n_vars,
                             uint8_t partial_out[G16_PARTIAL_BYTES]) {
  if (!c || !w) return fail(c, G16_ERR_INVALID, "null argument");
  if (c->multi) return fail(c, G16_ERR_INVALID, "multi-device ctx: use g16_prove");
  if (check_w(c, n_vars) != G16_OK) return G16_ERR_INVALID;
  g16_status st = guarded(c, [&]() -> g16_status {
    G16_HIP(hipMemcpyAsync(c->w_dev.p, w, (size_t)c->N * 32, hipMemcpyHostToDevice, c->stream));
    return G16_OK;
  });
  if (st != G16_OK) return st;
  return g16_prove_partial_dev(c, r, s_, c->w_dev.p, n_vars, partial_out);
}

g16_status g16_prove_finish(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4],
                            const uint8_t* partials, int world,
                            uint8_t proof_out[G16_PROOF_BYTES]) {
  if (!c || !r || !s_ || !partials || !proof_out) return fail(c, G16_ERR_INVALID, "null argument");
  if (c->multi) return fail(c, G16_ERR_INVALID, "multi-device ctx: use g16_prove");
  if (!c->has_key) return fail(c, G16_ERR_INVALID, "witness-map-only ctx has no resident key");
  if (world != c->world) return fail(c, G16_ERR_INVALID, "world does not match the ctx");
  return guarded(c, [&]() -> g16_status {
    hipStream_t s = c->stream;
    G16_HIP(hipMemcpyAsync(c->gathered_dev(), partials, (size_t)world * G16_PARTIAL_BYTES,
                           hipMemcpyHostToDevice, s));
    rank_finish_enqueue(c, r, s_, world);
    G16_HIP(hipMemcpyAsync(proof_out, c->out_dev.p, G16_PROOF_BYTES, hipMemcpyDeviceToHost, s));
    G16_HIP(hipStreamSynchronize(s));
    return G16_OK;
  });
}

void* g16_partial_buffer(g16_ctx* c) { return (c && !c->multi && c->has_key) ? (void*)c->part_dev() : nullptr; }
void* g16_gather_buffer(g16_ctx* c) { return (c && !c->multi && c->has_key) ? (void*)c->gathered_dev() : nullptr; }

g16_status g16_prove_finish_dev(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4],
                                uint8_t proof_out[G16_PROOF_BYTES]) {
  if (!c || !r || !s_ || !proof_out) return fail(c, G16_ERR_INVALID, "null argument");
  if (c->multi) return fail(c, G16_ERR_INVALID, "multi-device ctx: use g16_prove");
  if (!c->has_key) return fail(c, G16_ERR_INVALID, "witness-map-only ctx has no resident key");
  return guarded(c, [&]() -> g16_status {
    hipStream_t s = c->stream;
    if (c->have_xstream) {  // the all-gather into gathered_dev() was enqueued on the exchange stream
      G16_HIP(hipEventRecord(c->ev_user, c->xstream));
      G16_HIP(hipStreamWaitEvent(s, c->ev_user, 0));
    }
    rank_finish_enqueue(c, r, s_, c->world);
    G16_HIP(hipMemcpyAsync(proof_out, c->out_dev.p, G16_PROOF_BYTES, hipMemcpyDeviceToHost, s));
    G16_HIP(hipStreamSynchronize(s));
    collect_times(c);
    return G16_OK;
  });
}

g16_status g16_dist_set_exchange_stream(g16_ctx* c, void* hip_stream, int enabled) {
  if (!c || c->multi) return fail(c, G16_ERR_INVALID, "not a per-rank ctx");
  c->xstream = (hipStream_t)hip_stream;
  c->have_xstream = enabled != 0;
  return G16_OK;
}

size_t g16_dist_exchange_bytes(const g16_ctx* c) {
  return (c && c->dist_wm) ? c->wd.exchange_ints() * sizeof(int32_t) : 0;
}

// Hand-off to the host framework's collective.  With an exchange stream registered the caller's
// stream is made to wait for the event (the host never blocks); without one the call blocks until
// the buffer is complete, which is what a framework that cannot share a stream needs.
static void handoff(g16_ctx* c, hipEvent_t ev, hipStream_t producer) {
  if (c->have_xstream) G16_HIP(hipStreamWaitEvent(c->xstream, ev, 0));
  else G16_HIP(hipStreamSynchronize(producer));
}
// ... and back: what the caller enqueued on the exchange stream so far precedes `consumer`
static void handback(g16_ctx* c, hipStream_t consumer) {
  if (!c->have_xstream) return;  // blocking collectives: already complete when the call is made
  G16_HIP(hipEventRecord(c->ev_user, c->xstream));
  G16_HIP(hipStreamWaitEvent(consumer, c->ev_user, 0));
}

g16_status g16_prove_dist_phase1(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4],
                                 const void* w_dev, size_t n_vars, void* send_dev) {
  if (!c || !r || !s_ || !w_dev || !send_dev) return fail(c, G16_ERR_INVALID, "null argument");
  if (c->multi) return fail(c, G16_ERR_INVALID, "multi-device ctx: use g16_prove");
  if (!c->dist_wm || !c->has_key) return fail(c, G16_ERR_INVALID, "not a dist_wm proving ctx");
  if (check_w(c, n_vars) != G16_OK) return G16_ERR_INVALID;
  return guarded(c, [&]() -> g16_status {
    handback(c, c->stream);  // w_dev may have been produced on the caller's stream
    rank_phase1_enqueue(c, r, s_, (const Fr*)w_dev, (int32_t*)send_dev);
    handoff(c, c->ev_send, c->aux);  // send buffer complete; the main stream keeps running
    return G16_OK;
  });
}

g16_status g16_prove_dist_phase2(g16_ctx* c, const void* recv_dev, void* send_dev) {
  if (!c || !recv_dev || !send_dev) return fail(c, G16_ERR_INVALID, "null argument");
  if (c->multi) return fail(c, G16_ERR_INVALID, "multi-device ctx: use g16_prove");
  if (!c->dist_wm) return fail(c, G16_ERR_INVALID, "not a dist_wm ctx");
  return guarded(c, [&]() -> g16_status {
    handback(c, c->aux);
    rank_phase2_enqueue(c, (const int32_t*)recv_dev, (int32_t*)send_dev);
    handoff(c, c->ev_send, c->aux);
    return G16_OK;
  });
}

g16_status g16_prove_dist_phase3(g16_ctx* c, const void* recv_dev,
                                 uint8_t partial_out[G16_PARTIAL_BYTES]) {
  if (!c || !recv_dev) return fail(c, G16_ERR_INVALID, "null argument");
  if (c->multi) return fail(c, G16_ERR_INVALID, "multi-device ctx: use g16_prove");
  if (!c->dist_wm || !c->has_key) return fail(c, G16_ERR_INVALID, "not a dist_wm proving ctx");
  if (!partial_out && !c->have_xstream)
    return fail(c, G16_ERR_INVALID, "partial_out == NULL needs an exchange stream (g16_partial_buffer hand-off)");
  return guarded(c, [&]() -> g16_status {
    hipStream_t s = c->stream;
    handback(c, c->aux);
    rank_phase3_enqueue(c, (const int32_t*)recv_dev);
    if (partial_out) {
      G16_HIP(hipMemcpyAsync(partial_out, c->part_dev(), G16_PARTIAL_BYTES, hipMemcpyDeviceToHost, s));
      G16_HIP(hipStreamSynchronize(s));
      collect_times(c);
    } else {
      // device-side hand-off: the record stays in g16_partial_buffer(); the caller's all-gather
      // (enqueued on the exchange stream) waits for it
      G16_HIP(hipStreamWaitEvent(c->xstream, c->ev_part, 0));
    }
    return G16_OK;
  });
}

g16_status g16_set_profiling(g16_ctx* c, int enabled) {
  if (!c) return G16_ERR_INVALID;
  for (int g = 0; g < multi_size(c); ++g) g16_set_profiling(multi_child(c, g), enabled);
  c->timer.enabled = enabled != 0;
  for (int i = 0; i < ST_COUNT; ++i) {
    c->st_ms[i] = 0.f;
    c->st_cnt[i] = 0;
  }
  return G16_OK;
}

g16_status g16_stage_times(g16_ctx* c, float ms[G16_N_STAGES], uint32_t launches[G16_N_STAGES]) {
  if (!c || !ms || !launches) return G16_ERR_INVALID;
  static_assert(G16_N_STAGES == ST_COUNT, "stage table out of sync with the header");
  if (c->multi) {  // per stage: the slowest device (the one the proof waited for)
    for (int i = 0; i < ST_COUNT; ++i) {
      ms[i] = 0.f;
      launches[i] = 0;
    }
    for (int g = 0; g < multi_size(c); ++g) {
      float m1[ST_COUNT];
      uint32_t l1[ST_COUNT];
      g16_stage_times(multi_child(c, g), m1, l1);
      for (int i = 0; i < ST_COUNT; ++i)
        if (m1[i] >= ms[i]) {
          ms[i] = m1[i];
          launches[i] = l1[i];
        }
    }
    return G16_OK;
  }
  for (int i = 0; i < ST_COUNT; ++i) {
    ms[i] = c->st_ms[i];
    launches[i] = c->st_cnt[i];
    c->st_ms[i] = 0.f;
    c->st_cnt[i] = 0;
  }
  return G16_OK;
}

const char* g16_stage_name(int stage) {
  static const char* names[ST_COUNT] = {"witness_map",       "msm_sort",   "msm_accumulate_g1",
                                        "msm_accumulate_g2", "msm_reduce", "finalize",
                                        "msm_accumulate_g1_pair", "msm_fixup", "msm_table_g1", "msm_table_g2"};
  return (stage >= 0 && stage < ST_COUNT) ? names[stage] : "?";
}

g16_status g16_ctx_info(const g16_ctx* c, uint32_t out[16]) {
  if (!c || !out) return G16_ERR_INVALID;
  if (c->multi) {  // the shards are alike: report rank 0's configuration and the device count
    g16_status st = g16_ctx_info(multi_child(const_cast<g16_ctx*>(c), 0), out);
    out[12] = (uint32_t)multi_size(c);
    out[14] = c->peer_state;
    return st;
  }
  memset(out, 0, 16 * sizeof(uint32_t));
  out[12] = 1;
  out[0] = c->cfg_w.c; out[1] = c->cfg_w.W; out[2] = c->cfg_w.Pn; out[3] = c->cfg_w.D;
  out[4] = c->cfg_h.c; out[5] = c->cfg_h.W; out[6] = c->cfg_h.Pn; out[7] = c->cfg_h.D;
  out[8] = c->n;
  out[9] = (uint32_t)(c->dist_wm ? c->wd.k : c->wm.plan.base.k);
  out[10] = c->w_hi - c->w_lo;
  out[11] = c->h_hi - c->h_lo;
  out[13] = c->world > 1 ? (c->shard_buckets ? G16_SHARD_BUCKETS : G16_SHARD_POINTS) : 0;
  out[15] = (c->tbl.active ? 1u : 0u) | (c->sparse_b ? 2u : 0u);
  return G16_OK;
}

g16_status g16_multi_links(const g16_ctx* c, float* gbps, float* echo_us, int cap, uint64_t* probe_bytes) {
  if (!c || cap < 0) return G16_ERR_INVALID;
  size_t pb = 0;
  const int g = multi_links(c, gbps, echo_us, cap, &pb);
  if (g < 0) return fail(const_cast<g16_ctx*>(c), G16_ERR_INVALID, "g16_multi_links: not a multi-device ctx");
  if (probe_bytes) *probe_bytes = pb;
  return G16_OK;
}

void* g16_witness_buffer(g16_ctx* c) {
  if (c && c->multi) c = multi_child(c, 0);
  return c ? (void*)c->w_dev.p : nullptr;
}

g16_status g16_witness_upload(g16_ctx* c, const uint64_t* w, size_t n_vars) {
  if (!c || !w) return fail(c, G16_ERR_INVALID, "null argument");
  if (check_w(c, n_vars) != G16_OK) return G16_ERR_INVALID;
  if (c->multi) return multi_witness_upload(c, w);
  return guarded(c, [&]() -> g16_status {
    G16_HIP(hipMemcpyAsync(c->w_dev.p, w, (size_t)c->N * 32, hipMemcpyHostToDevice, c->stream));
    G16_HIP(hipStreamSynchronize(c->stream));
    return G16_OK;
  });
}

void* g16_witness_host_buffer(g16_ctx* c) {
  if (!c) return nullptr;
  if (!c->pinned_w) {
    (void)hipSetDevice(c->device);
    void* p = nullptr;
    // portable: a multi-device ctx uploads the same staging buffer to every device
    if (hipHostMalloc(&p, (size_t)(c->N ? c->N : 1) * 32, hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    c->pinned_w = p;
  }
  return c->pinned_w;
}

g16_status g16_ctx_create_multi(const g16_key_desc* key, const g16_csr* a, const g16_csr* b,
                                uint32_t num_constraints, const int* device_ids, int n_dev,
                                const g16_options* opt, g16_ctx** out) {
  if (!key || !a || !b || !out || !device_ids) return fail(nullptr, G16_ERR_INVALID, "null argument");
  *out = nullptr;
  if (n_dev < 1 || n_dev > 64) return fail(nullptr, G16_ERR_INVALID, "n_dev must be in [1, 64]");
  if (!key->a_query) return fail(nullptr, G16_ERR_INVALID, "a multi-device ctx needs the proving key");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, G16_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
  for (int i = 0; i < n_dev; ++i)
    if (device_ids[i] < 0 || device_ids[i] >= ndev) return fail(nullptr, G16_ERR_INVALID, "bad device ordinal");
  if (n_dev == 1) {
    g16_options o{};
    if (opt) o = *opt;
    o.device = device_ids[0];
    o.rank = 0;
    o.world = 1;
    o.dist_wm = 0;
    return g16_ctx_create(key, a, b, num_constraints, &o, out);
  }
  std::string err;
  const g16_status st = multi_create(key, a, b, num_constraints, device_ids, n_dev, opt, out, &err);
  if (st != G16_OK) return fail(nullptr, st, err);
  return G16_OK;
}

g16_status g16_check_satisfied(int device, const g16_csr* a, const g16_csr* b, const g16_csr* c_,
                               uint32_t num_constraints, const uint64_t* w, size_t n_vars,
                               int64_t* first_unsatisfied) {
  if (!a || !b || !c_ || !w || !first_unsatisfied) return fail(nullptr, G16_ERR_INVALID, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, G16_ERR_NO_DEVICE, "no HIP device visible");
  for (const g16_csr* m : {a, b, c_})
    for (uint64_t j = 0; j < m->nnz; ++j)
      if (m->col[j] >= n_vars) return fail(nullptr, G16_ERR_INVALID, "wire index beyond the witness");
  try {
    G16_HIP(hipSetDevice(device));
    CsrHost A{a->row_ptr, a->col, (const Fr*)a->coeff, (size_t)a->nnz};
    CsrHost B{b->row_ptr, b->col, (const Fr*)b->coeff, (size_t)b->nnz};
    CsrHost Cm{c_->row_ptr, c_->col, (const Fr*)c_->coeff, (size_t)c_->nnz};
    *first_unsatisfied = check_satisfied(A, B, Cm, num_constraints, (const Fr*)w, n_vars);
    return G16_OK;
  } catch (const HipError& e) {
    return fail(nullptr, G16_ERR_HIP, e.what());
  } catch (const std::exception& e) {
    return fail(nullptr, G16_ERR_INTERNAL, e.what());
  }
}

g16_status g16_fft_in_place(int device, uint64_t* data, int log_n, int inverse, int algo) {
  if (!data || log_n < 0) return fail(nullptr, G16_ERR_INVALID, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, G16_ERR_NO_DEVICE, "no HIP device visible");
  try {
    G16_HIP(hipSetDevice(device));
    NttPlan plan;
    plan.build(log_n);
    const size_t n = plan.n;
    DevBuf<Fr> a, b;
    a.alloc(n);
    b.alloc(n);
    G16_HIP(hipMemcpy(a.p, data, n * 32, hipMemcpyHostToDevice));
    Fr* res = a.p;
    if (algo == 0) {
      ntt_dif(plan, a.p, n, 1, inverse != 0, inverse ? NTT_FUSE_SCALE : NTT_FUSE_NONE, nullptr);
      bitrev_copy(a.p, b.p, log_n, nullptr);
      res = b.p;
    } else if (algo == 1) {
      if (inverse) return fail(nullptr, G16_ERR_INVALID, "algo 1 (DIT) is forward-only");
      bitrev_copy(a.p, b.p, log_n, nullptr);
      ntt_dit(plan, b.p, n, 1, false, nullptr);
      res = b.p;
    } else {  // 2: lazy-limb DIF + bit reversal, 3: bit reversal + lazy-limb DIT (forward only)
      if (algo == 3 && inverse) return fail(nullptr, G16_ERR_INVALID, "algo 3 (DIT) is forward-only");
      Ntt29Plan p29;
      p29.build(log_n, nullptr);
      DevBuf<int32_t> pa, pb;
      pa.alloc((size_t)NTT29_LIMBS * n);
      pb.alloc((size_t)NTT29_LIMBS * n);
      ntt29_to_planes(a.p, pa.p, n, nullptr);
      if (algo == 2) {
        ntt29_dif(p29, pa.p, (size_t)NTT29_LIMBS * n, 1, inverse != 0,
                  inverse ? NTT_FUSE_SCALE : NTT_FUSE_NONE, nullptr);
        ntt29_bitrev_planes(pa.p, pb.p, log_n, nullptr);
      } else {
        ntt29_bitrev_planes(pa.p, pb.p, log_n, nullptr);
        ntt29_dit(p29, pb.p, (size_t)NTT29_LIMBS * n, 1, nullptr);
      }
      ntt29_from_planes(pb.p, b.p, n, nullptr);
      res = b.p;
    }
    G16_HIP(hipDeviceSynchronize());
    G16_HIP(hipMemcpy(data, res, n * 32, hipMemcpyDeviceToHost));
    return G16_OK;
  } catch (const HipError& e) {
    return fail(nullptr, G16_ERR_HIP, e.what());
  } catch (const std::exception& e) {
    return fail(nullptr, G16_ERR_INTERNAL, e.what());
  }
}

}  // extern "C"
