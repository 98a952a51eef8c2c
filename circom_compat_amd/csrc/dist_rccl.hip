// dist_rccl.hip -- RCCL collectives issued by the library itself (g16_dist_attach_rccl / g16_prove_dist).
//
// north_star names "a final RCCL all-reduce of partial bucket sums over xGMI"; until round 6 the only RCCL
// call sites were the host framework's (bench.py through torch.distributed).  A Rust host has no
// torch.distributed: it creates the communicator and hands it over.  RCCL is NOT linked: the entry points are
// resolved from the RCCL the host already loaded (it created the communicator with it) or, failing that, from
// librccl.so -- a process that never attaches never touches it.
#include <dlfcn.h>

#include "ctx.h"

using namespace g16;

namespace {

using nccl_comm_t = void*;
struct Rccl {
  int (*all_to_all)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*comm_count)(nccl_comm_t, int*) = nullptr;
  int (*comm_user_rank)(nccl_comm_t, int*) = nullptr;
  const char* (*error_string)(int) = nullptr;
  std::string why;
  bool ok = false;
};
constexpr int NCCL_UINT8 = 1;  // ncclUint8 (rccl.h ncclDataType_t)

Rccl load_rccl() {
  Rccl r;
  void* h = nullptr;
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(RTLD_DEFAULT, name);  // the host's own RCCL, if it is in the process already
    if (!p && h) p = dlsym(h, name);
    return p;
  };
  if (!dlsym(RTLD_DEFAULT, "ncclAllToAll")) {
    for (const char* lib : {"librccl.so", "librccl.so.1"}) {
      h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) {
      r.why = "RCCL is neither loaded in this process nor found as librccl.so";
      return r;
    }
  }
  r.all_to_all = (decltype(r.all_to_all))sym("ncclAllToAll");
  r.all_gather = (decltype(r.all_gather))sym("ncclAllGather");
  r.comm_count = (decltype(r.comm_count))sym("ncclCommCount");
  r.comm_user_rank = (decltype(r.comm_user_rank))sym("ncclCommUserRank");
  r.error_string = (decltype(r.error_string))sym("ncclGetErrorString");
  r.ok = r.all_to_all && r.all_gather && r.comm_count && r.comm_user_rank;
  if (!r.ok) r.why = "the RCCL in this process lacks ncclAllToAll / ncclAllGather / ncclCommCount";
  return r;
}
Rccl& rccl() {
  static Rccl r = load_rccl();
  return r;
}
void nccl_check(int st, const char* what) {
  if (st == 0) return;
  const Rccl& R = rccl();
  throw std::runtime_error(std::string(what) + " failed: " + (R.error_string ? R.error_string(st) : "RCCL error ") +
                           " (" + std::to_string(st) + ")");
}
g16_status fail_ctx(g16_ctx* c, g16_status code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

}  // namespace

extern "C" {

g16_status g16_dist_attach_rccl(g16_ctx* c, void* nccl_comm) {
  if (!c || !nccl_comm) return fail_ctx(c, G16_ERR_INVALID, "null argument");
  if (c->multi || !c->dist_wm || !c->has_key)
    return fail_ctx(c, G16_ERR_INVALID, "g16_dist_attach_rccl: a per-rank proving ctx with dist_wm = 1 is needed");
  Rccl& R = rccl();
  if (!R.ok) return fail_ctx(c, G16_ERR_INVALID, "g16_dist_attach_rccl: " + R.why);
  try {
    int n = 0, me = -1;
    nccl_check(R.comm_count(nccl_comm, &n), "ncclCommCount");
    nccl_check(R.comm_user_rank(nccl_comm, &me), "ncclCommUserRank");
    if (n != c->world || me != c->rank)
      return fail_ctx(c, G16_ERR_INVALID, "g16_dist_attach_rccl: the communicator is rank " + std::to_string(me) + " of " +
                                              std::to_string(n) + ", the ctx rank " + std::to_string(c->rank) + " of " +
                                              std::to_string(c->world));
    G16_HIP(hipSetDevice(c->device));
    if (!c->xs_own) {
      int lo = 0, hi = 0;
      G16_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      (void)lo;
      // high priority: the collectives must not queue behind the ctx's accumulation launches
      G16_HIP(hipStreamCreateWithPriority(&c->xs_own, hipStreamNonBlocking, hi));
    }
    const size_t bytes = c->wd.exchange_ints() * sizeof(int32_t);
    c->xsend.alloc(bytes);
    c->xrecv.alloc(bytes);
    c->nccl_comm = nccl_comm;
    c->nccl_ranks = n;
    return G16_OK;
  } catch (const HipError& e) {
    return fail_ctx(c, G16_ERR_HIP, e.what());
  } catch (const std::exception& e) {
    return fail_ctx(c, G16_ERR_INTERNAL, e.what());
  }
}

int g16_dist_rccl_ranks(const g16_ctx* c) { return c ? c->nccl_ranks : 0; }

g16_status g16_prove_dist(g16_ctx* c, const uint64_t r[4], const uint64_t s_[4], const void* w_dev, size_t n_vars,
                          uint8_t proof_out[G16_PROOF_BYTES]) {
  if (!c || !r || !s_ || !w_dev || !proof_out) return fail_ctx(c, G16_ERR_INVALID, "null argument");
  if (!c->nccl_comm) return fail_ctx(c, G16_ERR_INVALID, "g16_prove_dist: no communicator attached (g16_dist_attach_rccl)");
  Rccl& R = rccl();
  // the phase calls order themselves against the exchange stream with events: the library's own stream here
  const hipStream_t saved = c->xstream;
  const bool had = c->have_xstream;
  c->xstream = c->xs_own;
  c->have_xstream = true;
  g16_status st = G16_OK;
  auto restore = [&] {
    c->xstream = saved;
    c->have_xstream = had;
  };
  try {
    const size_t bytes = c->xsend.bytes(), per_peer = bytes / (size_t)c->world;
    st = g16_prove_dist_phase1(c, r, s_, w_dev, n_vars, c->xsend.p);
    if (st == G16_OK) {
      nccl_check(R.all_to_all(c->xsend.p, c->xrecv.p, per_peer, NCCL_UINT8, c->nccl_comm, c->xs_own), "ncclAllToAll (exchange 1)");
      st = g16_prove_dist_phase2(c, c->xrecv.p, c->xsend.p);
    }
    if (st == G16_OK) {
      nccl_check(R.all_to_all(c->xsend.p, c->xrecv.p, per_peer, NCCL_UINT8, c->nccl_comm, c->xs_own), "ncclAllToAll (exchange 2)");
      st = g16_prove_dist_phase3(c, c->xrecv.p, nullptr);  // the record stays on the device, xs_own waits for it
    }
    if (st == G16_OK) {
      nccl_check(R.all_gather(c->part_dev(), c->gathered_dev(), G16_PARTIAL_BYTES, NCCL_UINT8, c->nccl_comm, c->xs_own),
                 "ncclAllGather (partial records)");
      st = g16_prove_finish_dev(c, r, s_, proof_out);
    }
  } catch (const std::exception& e) {
    st = fail_ctx(c, G16_ERR_HIP, e.what());
  }
  restore();
  return st;
}

}  // extern "C"
