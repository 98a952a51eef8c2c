// ctx.h -- private definition of g16_ctx (the opaque handle of include/g16_amd.h), shared by
// api.hip (per-device drivers) and multi.hip (single-process multi-device orchestration).
#pragma once
#include <atomic>
#include "../../include/g16_amd.h"

#include <memory>
#include <string>

#include "finalize.h"
#include "msm_table.h"
#include "msm.h"
#include "witness_map.h"
#include "wm_dist.h"

namespace g16 {
struct Multi;  // multi.hip
}

struct g16_ctx {
  int device = 0, rank = 0, world = 1;
  uint32_t N = 0, p = 0, n = 0, m = 0, num_inputs = 0;
  bool has_key = false;  // false: witness-map-only context (a_query == NULL at create)
  bool overlap = true;   // G16_NO_OVERLAP=1: everything on one stream (A/B measurements)
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;  // finalize stages that overlap the MSMs
  hipStream_t aux = nullptr;   // witness map + H-query sort, beside the witness-scalar MSMs
  hipStream_t red = nullptr;   // G2 bucket reduction, underneath the H MSM
  hipEvent_t ev_start = nullptr, ev_ab = nullptr, ev_side = nullptr, ev_w = nullptr, ev_h = nullptr,
             ev_b2 = nullptr, ev_fixed = nullptr, ev_view = nullptr;  // ev_view: the filtered B view is built
  hipEvent_t ev_acc[3] = {nullptr, nullptr, nullptr};
  // sharded provers: hand-off points of the exchanges (no host synchronisation in between)
  hipEvent_t ev_send = nullptr;  // aux stream: the send buffer of the last phase is complete
  hipEvent_t ev_part = nullptr;  // main stream: this rank's partial record is complete (part_dev())
  hipEvent_t ev_user = nullptr;  // recorded on the caller's exchange stream at phase entry
  // g16_dist_set_exchange_stream: the stream the host framework enqueues its collectives on; when
  // set, the phase calls order themselves against it with events instead of blocking the host
  hipStream_t xstream = nullptr;
  bool have_xstream = false;
  // g16_dist_attach_rccl: the communicator of the host's RCCL (opaque ncclComm_t), its size, the library's own
  // high-priority exchange stream and the two exchange buffers g16_prove_dist moves with ncclAllToAll
  void* nccl_comm = nullptr;
  int nccl_ranks = 0;
  hipStream_t xs_own = nullptr;
  g16::DevBuf<uint8_t> xsend, xrecv;
  std::string err;

  g16::WitnessMap wm;
  g16::WmDist wd;  // distributed witness map (options.dist_wm, world > 1)
  bool dist_wm = false;
  // sharded provers: r/s-only finalisation sums already enqueued on the side stream by the
  // partial / phase-1 call for these (r, s)
  bool fixed_ready = false;
  uint64_t fixed_rs[8] = {0};
  // Sharding of the MSMs over the ranks (options.shard):
  //   point ranges  -- rank g holds the points [w_lo, w_hi) / [h_lo, h_hi) of every query and its own
  //                    (smaller) window configuration;
  //   bucket ranges -- the witness-scalar queries (A, B1, B2, L): every rank holds ALL their points
  //                    (w_lo = 0, w_hi = N - 1) with the single-GPU window and keeps 1/world of the
  //                    sorted (bucket, point) list (MsmSort::set_shard): the accumulation AND the
  //                    bucket reduction shrink by world, nothing but the 1 KiB record leaves the
  //                    device.  The H query stays cut by point range: its scalars are born sharded.
  bool shard_buckets = false;
  // shard of the assignment-index space [0, N-1) (entry i <-> w[1+i]) and of [0, n) for H
  uint32_t w_lo = 0, w_hi = 0, h_lo = 0, h_hi = 0;
  uint32_t l_idx_min = 0;  // entries below this local index have no L point (public inputs)
  g16::MsmConfig cfg_w, cfg_h;
  g16::MsmSort sort_w, sort_h;
  // Sparse B queries (real circom keys: wires that appear in no B row have the point at infinity in
  // b_g1_query / b_g2_query): sort_b = the witness sort without those points (MsmSort::run_view); the B2
  // (G2) MSM accumulates and reduces over it.  keep_b: one bit per entry i (wire i + 1), set when the point is finite.
  g16::MsmSort sort_b;
  g16::DevBuf<uint32_t> keep_b_own;
  const uint32_t* keep_b = nullptr;
  bool sparse_b = false;
  uint32_t b_inf_points = 0;  // points at infinity among b_g1_query[1..] (reported by g16_ctx_info)
  const g16::MsmSort& sort_for_b() const { return sparse_b ? sort_b : sort_w; }
  g16::MsmPoints<g16::Fq> ptsA, ptsB1, ptsL, ptsH;
  g16::MsmPoints<g16::Fq2> ptsB2;
  g16::MsmWork<g16::Fq> work1, workH;  // witness-scalar G1 MSMs (A, B1, L) / H MSM
  g16::MsmWork<g16::Fq2> work2;
  g16::TableSet tbl;  // small keys: fixed-base tables (msm_table.h); g16_prove goes through them when active

  g16::DevBuf<g16::Fr> w_dev, h_dev, rs_dev;  // h_dev: storage form (g16_witness_map / g16_msm_g1 staging)
  g16::DevBuf<g16::U256> h_canon;             // h as canonical integers: scalars of the H-query MSM
  g16::DevBuf<g16::KeyHeaderDev> key_dev;
  g16::DevBuf<g16::ProofSums> sums_dev;
  g16::DevBuf<g16::FinTables> fin_tab;
  g16::DevBuf<g16::FinScratch> fin_scr;
  g16::DevBuf<uint8_t> out_dev;  // proof (256) | this rank's partial (1024) | gathered partials
  void* pinned_w = nullptr;      // g16_witness_host_buffer: page-locked staging for the witness
  uint8_t* pin_io = nullptr;     // page-locked (r, s) [64 B] | proof [256 B] of g16_prove_dev

  g16::StageTimer timer;
  float st_ms[g16::ST_COUNT] = {0};
  uint32_t st_cnt[g16::ST_COUNT] = {0};

  // parent of a single-process multi-device prover (g16_ctx_create_multi): holds no device state
  // of its own, only the per-device children and the exchange plumbing between them
  g16::Multi* multi = nullptr;
  g16_ctx* share_from = nullptr;  // lender of the point planes (kept alive by `borrowers` below)
  std::atomic<int> borrowers{0};  // live ctxs that borrow this one's planes (incremented from parallel create threads)
  bool zombie = false;            // g16_ctx_destroy was called while borrowers > 0: freed by the last borrower's destroy
  uint64_t nnz_a = 0, nnz_b = 0;  // shape of the constraint matrices (g16_ctx_create_sibling checks them)
  uint32_t peer_state = 0;        // multi-device parent: 1 = every peer pair has direct access, 2 = some copies are staged

  uint8_t* part_dev() { return out_dev.p + G16_PROOF_BYTES; }
  uint8_t* gathered_dev() { return out_dev.p + G16_PROOF_BYTES + G16_PARTIAL_BYTES; }
};

namespace g16 {

// ---- asynchronous cores of the sharded provers (api.hip): enqueue only, never block the host ----
// All of them set the ctx's device, throw HipError / std::exception on failure, and leave the
// hand-off events (ev_send / ev_part) recorded.
void rank_phase1_enqueue(g16_ctx* c, const uint64_t r[4], const uint64_t s[4], const Fr* w_dev,
                         int32_t* send_dev);
void rank_phase2_enqueue(g16_ctx* c, const int32_t* recv_dev, int32_t* send_dev);
void rank_phase3_enqueue(g16_ctx* c, const int32_t* recv_dev);  // ... -> part_dev(), ev_part
// internal create: `share_from` (same device, same key, bucket-range sharding) lends its point
// planes; errors come back through *err (never through the process-global message)
g16_status ctx_create_impl(const g16_key_desc* key, const g16_csr* a, const g16_csr* b,
                           uint32_t num_constraints, const g16_options* opt, g16_ctx* share_from,
                           g16_ctx** out, std::string* err);
// true when the full-precomputation planes of the WHOLE key fit `device` (bucket-range sharding)
void rank_partial_enqueue(g16_ctx* c, const uint64_t r[4], const uint64_t s[4], const Fr* w_dev);
// gathered: world x G16_PARTIAL_BYTES already in c->gathered_dev() (ordered on the main stream)
void rank_finish_enqueue(g16_ctx* c, const uint64_t r[4], const uint64_t s[4], int world);
void rank_collect_times(g16_ctx* c);

// ---- single-process multi-device prover (multi.hip) -------------------------------------------
g16_status multi_create(const g16_key_desc* key, const g16_csr* a, const g16_csr* b,
                        uint32_t num_constraints, const int* device_ids, int n_dev,
                        const g16_options* opt, g16_ctx** out, std::string* err);
void multi_destroy(Multi* m);
// w: host pointer (w_on_device = false: every device uploads its own copy over its own PCIe link)
// or a device pointer on children[0]'s device (true: peer-broadcast)
g16_status multi_prove(g16_ctx* parent, const uint64_t r[4], const uint64_t s[4], const void* w,
                       bool w_on_device, uint8_t proof_out[G16_PROOF_BYTES]);
// host witness -> every device's staging buffer; a later multi_prove(w = children[0]->w_dev.p, on
// device) then moves nothing
g16_status multi_witness_upload(g16_ctx* parent, const uint64_t* w);
g16_ctx* multi_child(g16_ctx* parent, int index);  // nullptr when out of range / not a parent
int multi_size(const g16_ctx* parent);
// create-time link probe of a multi-device parent: [src * G + dst] tables; returns G, -1 if not a parent
int multi_links(const g16_ctx* parent, float* gbps, float* echo_us, int cap, size_t* probe_bytes);

}  // namespace g16
