// multi.hip -- single-process multi-device Groth16 prover: g16_ctx_create_multi / g16_prove on a
// parent ctx whose children are one sharded rank per device.
//
// What is sharded (north_star: "MSM shards by point-range across the 8 GPUs ... partial sums over
// xGMI"): the MSMs -- every query array cut by contiguous point range at create time (G16_SHARD_POINTS,
// what AUTO selects), or the four witness-scalar queries cut by BUCKET range (G16_SHARD_BUCKETS: every
// device holds all their points and the single-GPU window and keeps 1/G of the sorted bucket list;
// ranks that repeat a device ordinal borrow the first one's planes) -- and, when the device count is
// a power of two, the witness map as well (wm_dist.h: four-step NTTs whose two transposes are
// all-to-all exchanges).  Nothing of a proof touches the host between the witness upload and the
// 256-byte download:
//   * exchanges are PUSHED with hipMemcpyPeerAsync, one copy stream per destination so that all
//     seven xGMI links of a device carry one chunk each at the same time (xGMI is point-to-point:
//     an all-to-all IS seven independent peer copies per device; there is no ring to build);
//   * hand-offs are events: a consumer stream waits for the `arrived` events of its G producers,
//     the 1 KiB partial records are peer-copied to device 0 behind each rank's ev_part, and the
//     "all-reduce of partial sums" is a gather + local EC additions there (EC addition is not an
//     ncclRedOp; the payload is 8 KiB, i.e. latency only);
//   * one host thread per device enqueues that device's ~100 launches (a single thread would
//     serialise 8 x 0.5 ms of launch overhead in front of a 5-7 ms rank); the threads meet at a
//     host barrier only so that an event is RECORDED before a peer's stream is told to wait for it.
// The emulator build (tests only) has no threads: the same stages run device after device.
#include "ctx.h"

#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <vector>

#include <mutex>
#ifndef G16_EMU
#include <condition_variable>
#include <thread>
#endif

namespace g16 {

namespace {

// ---- stage runner: stage k of every device completes (on the host) before stage k+1 starts ----
struct StageRunner {
  int n = 0;
  std::string first_error;
  int first_code = G16_OK;
#ifndef G16_EMU
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  const std::vector<std::function<void(int)>>* stages = nullptr;
  uint64_t gen = 0;      // job generation the workers wait for
  int done = 0;          // workers that finished the current job
  bool stop = false;
  int bcount = 0;        // barrier state
  uint64_t bgen = 0;
  bool failed = false;
#endif

  void start(int count) {
    n = count;
#ifndef G16_EMU
    for (int g = 1; g < n; ++g) th.emplace_back([this, g] { worker(g); });
#endif
  }
  void shutdown() {
#ifndef G16_EMU
    {
      std::lock_guard<std::mutex> l(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : th) t.join();
    th.clear();
#endif
  }

  void note(int code, const std::string& msg) {
#ifndef G16_EMU
    std::lock_guard<std::mutex> l(mu);
    failed = true;
#endif
    if (first_code == G16_OK) {
      first_code = code;
      first_error = msg;
    }
  }
  bool call(const std::function<void(int)>& f, int g) {
    try {
      f(g);
      return true;
    } catch (const HipError& e) {
      note(G16_ERR_HIP, e.what());
    } catch (const std::bad_alloc&) {
      note(G16_ERR_INTERNAL, "host allocation failed");
    } catch (const std::exception& e) {
      const bool dom = std::string(e.what()).find("PolynomialDegreeTooLarge") != std::string::npos;
      note(dom ? G16_ERR_DOMAIN_TOO_LARGE : G16_ERR_INTERNAL, e.what());
    }
    return false;
  }

#ifndef G16_EMU
  void barrier() {
    std::unique_lock<std::mutex> l(mu);
    const uint64_t my = bgen;
    if (++bcount == n) {
      bcount = 0;
      ++bgen;
      cv.notify_all();
    } else {
      cv.wait(l, [&] { return bgen != my; });
    }
  }
  void run_as(int g) {
    for (size_t k = 0; k < stages->size(); ++k) {
      bool skip;
      {
        std::lock_guard<std::mutex> l(mu);
        skip = failed;
      }
      if (!skip) call((*stages)[k], g);
      barrier();  // every thread passes every barrier, failed or not
    }
  }
  void worker(int g) {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return stop || gen != seen; });
        if (stop) return;
        seen = gen;
      }
      run_as(g);
      {
        std::lock_guard<std::mutex> l(mu);
        ++done;
      }
      cv.notify_all();
    }
  }
#endif

  // returns G16_OK or the first failure (message in first_error)
  int run(const std::vector<std::function<void(int)>>& st) {
    first_code = G16_OK;
    first_error.clear();
#ifdef G16_EMU
    for (auto& f : st)
      for (int g = 0; g < n && first_code == G16_OK; ++g) call(f, g);
#else
    {
      std::lock_guard<std::mutex> l(mu);
      stages = &st;
      failed = false;
      done = 0;
      ++gen;
    }
    cv.notify_all();
    run_as(0);
    {
      std::unique_lock<std::mutex> l(mu);
      cv.wait(l, [&] { return done == n - 1; });
      stages = nullptr;
    }
#endif
    return first_code;
  }
};

}  // namespace

struct Multi {
  int G = 0;
  std::vector<g16_ctx*> ch;
  bool dist = false;    // fully sharded: the witness map is distributed too (power-of-two G)
  bool buckets = false;  // MSMs sharded by bucket range (every device holds the whole key), else by point range
  size_t chunk_ints = 0;  // int32 per (source, destination) pair and exchange
  struct Dev {
    DevBuf<int32_t> send[2], recv[2];
    std::vector<hipStream_t> cs;            // one copy stream per destination
    std::vector<hipEvent_t> arrived[2];     // [exchange][dst]: my chunk has landed in dst's recv buffer
  };
  std::vector<std::unique_ptr<Dev>> dv;  // DevBuf is not movable
  bool witness_resident = false;  // g16_witness_upload: every child's w_dev holds the current witness
  // create-time probe of every ordered (source, destination) pair (multi_link_probe): GB/s of one large
  // peer copy, microseconds of a 4 KiB there-and-back; [src * G + dst]
  std::vector<float> link_gbps, link_echo_us;
  size_t link_probe_bytes = 0;
  bool links_probed = false;  // the probe runs on the first g16_multi_links call, not at create
  StageRunner pool;
};

namespace {

// Direct peer access for every ordered pair of distinct devices.  Returns 1 when all of them have
// it, 2 when at least one does not: hipMemcpyPeerAsync still works then, but the runtime stages
// the copy (through host memory), which turns every exchange of a proof into two PCIe crossings.
// That is reported (g16_ctx_info out[14]), never hidden; G16_REQUIRE_PEER_ACCESS=1 makes it an error.
uint32_t enable_peers(const std::vector<g16_ctx*>& ch, std::string* why) {
  uint32_t state = 1;
  for (auto* a : ch)
    for (auto* b : ch) {
      if (a->device == b->device) continue;
      G16_HIP(hipSetDevice(a->device));
      int can = 0;
      hipError_t e = hipDeviceCanAccessPeer(&can, a->device, b->device);
      if (e == hipSuccess && can) {
        e = hipDeviceEnablePeerAccess(b->device, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
      } else if (e == hipSuccess) {
        e = hipErrorPeerAccessUnsupported;
      }
      (void)hipGetLastError();
      if (e != hipSuccess) {
        state = 2;
        if (why && why->empty())
          *why = "no direct peer access from device " + std::to_string(a->device) + " to device " +
                 std::to_string(b->device) + " (" + hipGetErrorString(e) + ")";
      }
    }
  return state;
}

// G16_DEBUG_SELFTEST_CORRUPT="x:src:dst" (tests only): the create-time self-test's copy of exchange x
// (0, 1; 2 = the record gather) from rank src to rank dst reads the wrong source chunk -- what a
// misrouted peer copy would deliver.  tests/test_kernels.py checks that g16_ctx_create_multi then fails.
bool selftest_corrupt(int x, int src, int dst) {
  const char* e = getenv("G16_DEBUG_SELFTEST_CORRUPT");
  int cx = -1, cs = -1, cd = -1;
  return e && sscanf(e, "%d:%d:%d", &cx, &cs, &cd) == 3 && cx == x && cs == src && cd == dst;
}

// exchange x of rank g: its G chunks go to the G recv buffers (own chunk included), each on its
// own stream behind the producer's ev_send.  ints < chunk_ints: only the head of every chunk (the
// create-time self-test); selftest: the corruption hook above applies.
void push_chunks(Multi& M, int g, int x, size_t ints = 0, bool selftest = false) {
  g16_ctx* c = M.ch[g];
  Multi::Dev& me = *M.dv[g];
  const size_t bytes = (ints ? ints : M.chunk_ints) * sizeof(int32_t);
  for (int k = 0; k < M.G; ++k) {
    const int d = (g + k) % M.G;  // start with the own chunk, then rotate: no destination is hit by all at once
    hipStream_t st = me.cs[d];
    const int from = selftest && selftest_corrupt(x, g, d) ? (d + 1) % M.G : d;
    G16_HIP(hipStreamWaitEvent(st, c->ev_send, 0));
    G16_HIP(hipMemcpyPeerAsync(M.dv[d]->recv[x].p + (size_t)g * M.chunk_ints, M.ch[d]->device,
                               me.send[x].p + (size_t)from * M.chunk_ints, c->device, bytes, st));
    G16_HIP(hipEventRecord(me.arrived[x][d], st));
  }
}

void await_chunks(Multi& M, int g, int x, hipStream_t consumer) {
  for (int src = 0; src < M.G; ++src)
    G16_HIP(hipStreamWaitEvent(consumer, M.dv[src]->arrived[x][g], 0));
}

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

// ---- create-time self-test of the peer paths --------------------------------------------------
// A proof's exchanges are peer copies and cross-device event waits that nothing on a single-GPU box
// ever executes between two DISTINCT devices.  So g16_ctx_create_multi runs one small all-to-all
// echo through exactly the code a proof uses -- push_chunks / await_chunks on the aux stream behind
// ev_send, both exchange buffers, every (source, destination) pair incl. the local one -- and one
// gather of the 1 KiB partial records to the first device behind ev_part, all with known patterns:
// a broken or misrouted peer path is an error at create (naming the pair), not a wrong proof later.
namespace {

constexpr uint32_t ST_WORDS = 1024;   // 4 KiB per (source, destination) pair
constexpr uint32_t ST_ECHO = 0x5a5a5a5au;

__host__ __device__ inline uint32_t selftest_word(uint32_t src, uint32_t dst, uint32_t i, uint32_t salt) {
  uint32_t x = (src + 1u) * 0x9e3779b1u ^ (dst + 1u) * 0x85ebca77u ^ (i + 1u) * 0xc2b2ae3du ^ salt;
  x ^= x >> 15;
  x *= 0x2c1b3c6du;
  x ^= x >> 12;
  return x;
}

// buf[d * stride + i] = word(me, d, i) for d < G, i < T
__global__ void __launch_bounds__(256) k_selftest_fill(int32_t* buf, size_t stride, uint32_t T, uint32_t G,
                                                       uint32_t me, uint32_t salt) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G * T) return;
  const uint32_t d = idx / T, i = idx % T;
  buf[(size_t)d * stride + i] = (int32_t)selftest_word(me, d, i, salt);
}

// mode 0: recv[src * stride + i] must be word(src, me, i); the echo send2[src * stride + i] = that ^ ST_ECHO
// mode 1: recv[src * stride + i] must be word(me, src, i) ^ ST_ECHO (my own words, back from src)
// mode 2: recv[src * stride + i] must be word(src, 0, i) (the gathered records)
// bad[src] counts the mismatching words of source src
__global__ void __launch_bounds__(256) k_selftest_check(const int32_t* recv, int32_t* send2, size_t stride, uint32_t T,
                                                        uint32_t G, uint32_t me, uint32_t salt, int mode,
                                                        uint32_t* bad) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G * T) return;
  const uint32_t src = idx / T, i = idx % T;
  const uint32_t got = (uint32_t)recv[(size_t)src * stride + i];
  uint32_t want;
  if (mode == 0) want = selftest_word(src, me, i, salt);
  else if (mode == 1) want = selftest_word(me, src, i, salt) ^ ST_ECHO;
  else want = selftest_word(src, 0u, i, salt);
  if (got != want) atomicAdd(&bad[src], 1u);
  if (mode == 0) send2[(size_t)src * stride + i] = (int32_t)(got ^ ST_ECHO);
}

// What the links deliver, measured with the copies a proof makes (hipMemcpyPeerAsync from the source's
// exchange buffer into the destination's, on the source's copy stream for that destination): one copy
// of up to 64 MiB timed by events on that stream, and a 4 KiB copy there and back timed on the host.
// One pair at a time, on the first g16_multi_links call (a measurement must neither lengthen nor be
// able to fail the creation of a ctx that would otherwise work: its errors leave zeros): the figures are the uncontended per-link rates that
// scripts/dist_projection.py assumes (48 GB/s per xGMI link) -- on the first multi-GPU box they are a
// measurement instead (g16_multi_links; scripts/hardware_day.sh prints the table).
void multi_link_probe(Multi& M) {
  const int G = M.G;
  M.link_gbps.assign((size_t)G * G, 0.f);
  M.link_echo_us.assign((size_t)G * G, 0.f);
  if (!M.dist) return;  // no exchange buffers: the replicated witness map moves 1 KiB records only
  const size_t bytes = std::min<size_t>(M.dv[0]->send[0].bytes(), (size_t)64 << 20);
  M.link_probe_bytes = bytes;
  if (bytes < 4096) return;
  for (int a = 0; a < G; ++a) {
    g16_ctx* ca = M.ch[a];
    G16_HIP(hipSetDevice(ca->device));
    hipEvent_t e0, e1;
    G16_HIP(hipEventCreate(&e0));
    G16_HIP(hipEventCreate(&e1));
    for (int b = 0; b < G; ++b) {
      if (a == b) continue;  // no link to itself
      g16_ctx* cb = M.ch[b];
      hipStream_t cs = M.dv[a]->cs[b];
      // warm the path, then time one copy
      G16_HIP(hipMemcpyPeerAsync(M.dv[b]->recv[0].p, cb->device, M.dv[a]->send[0].p, ca->device, 4096, cs));
      G16_HIP(hipStreamSynchronize(cs));
      G16_HIP(hipEventRecord(e0, cs));
      G16_HIP(hipMemcpyPeerAsync(M.dv[b]->recv[0].p, cb->device, M.dv[a]->send[0].p, ca->device, bytes, cs));
      G16_HIP(hipEventRecord(e1, cs));
      G16_HIP(hipStreamSynchronize(cs));
      float ms = 0.f;
      G16_HIP(hipEventElapsedTime(&ms, e0, e1));
      M.link_gbps[(size_t)a * G + b] = ms > 0.f ? (float)((double)bytes / (ms * 1e-3) / 1e9) : 0.f;
      const auto t0 = std::chrono::steady_clock::now();
      const int reps = 8;
      for (int r = 0; r < reps; ++r) {
        G16_HIP(hipMemcpyPeerAsync(M.dv[b]->recv[0].p, cb->device, M.dv[a]->send[0].p, ca->device, 4096, cs));
        G16_HIP(hipStreamSynchronize(cs));
        G16_HIP(hipMemcpyPeerAsync(M.dv[a]->recv[1].p, ca->device, M.dv[b]->recv[0].p, cb->device, 4096, cs));
        G16_HIP(hipStreamSynchronize(cs));
      }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      M.link_echo_us[(size_t)a * G + b] = (float)(us / reps);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
}

void multi_selftest(Multi& M) {
  const int G = M.G;
  const uint32_t T = M.dist ? (uint32_t)std::min<size_t>(M.chunk_ints, ST_WORDS) : 0u;
  const uint32_t TR = G16_PARTIAL_BYTES / 4;  // one partial record
  const uint32_t salt = 0x67313661u;
  std::vector<std::unique_ptr<DevBuf<uint32_t>>> bad;
  for (int g = 0; g < G; ++g) bad.emplace_back(new DevBuf<uint32_t>());
  std::vector<std::vector<uint32_t>> host(G, std::vector<uint32_t>(3 * (size_t)G, 0u));
  std::vector<std::function<void(int)>> st;
  st.push_back([&](int g) {
    g16_ctx* c = M.ch[g];
    G16_HIP(hipSetDevice(c->device));
    bad[g]->alloc(3 * (size_t)G);
    G16_HIP(hipMemsetAsync(bad[g]->p, 0, 3 * (size_t)G * 4, c->aux));
    if (T)
      G16_LAUNCH(k_selftest_fill, ceil_div((uint64_t)G * T, 256), 256, 0, c->aux, M.dv[g]->send[0].p, M.chunk_ints, T,
                 (uint32_t)G, (uint32_t)g, salt);
    G16_HIP(hipEventRecord(c->ev_send, c->aux));
    if (T) push_chunks(M, g, 0, T, /*selftest=*/true);
    // this rank's "partial record": the pattern word(g, 0, i), complete behind ev_part on the main stream
    G16_HIP(hipStreamWaitEvent(c->stream, c->ev_send, 0));
    G16_LAUNCH(k_selftest_fill, ceil_div(TR, 256), 256, 0, c->stream, (int32_t*)c->part_dev(), (size_t)TR, TR, 1u,
               (uint32_t)g, salt ^ 2u);
    G16_HIP(hipEventRecord(c->ev_part, c->stream));
  });
  if (T)
    st.push_back([&](int g) {
      g16_ctx* c = M.ch[g];
      G16_HIP(hipSetDevice(c->device));
      await_chunks(M, g, 0, c->aux);
      G16_LAUNCH(k_selftest_check, ceil_div((uint64_t)G * T, 256), 256, 0, c->aux, (const int32_t*)M.dv[g]->recv[0].p,
                 M.dv[g]->send[1].p, M.chunk_ints, T, (uint32_t)G, (uint32_t)g, salt, 0, bad[g]->p);
      G16_HIP(hipEventRecord(c->ev_send, c->aux));
      push_chunks(M, g, 1, T, /*selftest=*/true);
    });
  st.push_back([&](int g) {
    g16_ctx* c = M.ch[g];
    G16_HIP(hipSetDevice(c->device));
    if (T) {
      await_chunks(M, g, 1, c->aux);
      G16_LAUNCH(k_selftest_check, ceil_div((uint64_t)G * T, 256), 256, 0, c->aux, (const int32_t*)M.dv[g]->recv[1].p,
                 (int32_t*)nullptr, M.chunk_ints, T, (uint32_t)G, (uint32_t)g, salt, 1, bad[g]->p + G);
    }
    if (g == 0) {  // the record gather of stage_gather_finish, checked instead of summed
      hipStream_t s = c->stream;
      for (int src = 0; src < G; ++src) {
        G16_HIP(hipStreamWaitEvent(s, M.ch[src]->ev_part, 0));
        const size_t off = selftest_corrupt(2, src, 0) ? 4 : 0;
        G16_HIP(hipMemcpyPeerAsync(c->gathered_dev() + (size_t)src * G16_PARTIAL_BYTES, c->device,
                                   M.ch[src]->part_dev() + off, M.ch[src]->device, G16_PARTIAL_BYTES - off, s));
      }
      G16_HIP(hipStreamWaitEvent(s, c->ev_send, 0));  // bad[] was zeroed on aux
      G16_LAUNCH(k_selftest_check, ceil_div((uint64_t)G * TR, 256), 256, 0, s, (const int32_t*)c->gathered_dev(),
                 (int32_t*)nullptr, (size_t)TR, TR, (uint32_t)G, 0u, salt ^ 2u, 2, bad[0]->p + 2 * G);
      G16_HIP(hipStreamSynchronize(s));
    }
    G16_HIP(hipStreamSynchronize(c->aux));
    G16_HIP(hipMemcpy(host[g].data(), bad[g]->p, 3 * (size_t)G * 4, hipMemcpyDeviceToHost));
  });
  // a stage of its own: the first device has finished gathering (it synchronised in the stage above)
  // before any rank clears the record it was read from -- with the clean-up inside the gather stage a
  // fast rank zeroed its record under the gather's peer copy (seen once on the GPU, never on the
  // emulator, whose stages run device after device)
  st.push_back([&](int g) {
    g16_ctx* c = M.ch[g];
    G16_HIP(hipSetDevice(c->device));
    // leave the buffers as a fresh ctx has them
    G16_HIP(hipMemsetAsync(c->out_dev.p, 0, c->out_dev.bytes(), c->stream));
    G16_HIP(hipStreamSynchronize(c->stream));
    bad[g]->release();
  });
  const int code = M.pool.run(st);
  if (code != G16_OK) throw std::runtime_error("multi-device self-test: " + M.pool.first_error);
  static const char* what[3] = {"all-to-all exchange 0", "all-to-all exchange 1 (echo)", "partial-record gather"};
  for (int x = 0; x < 3; ++x)  // the earliest failing stage first: a bad exchange 0 also spoils its echo
    for (int g = 0; g < G; ++g)
      for (int src = 0; src < G; ++src)
        if (host[g][(size_t)x * G + src])
          throw std::runtime_error(std::string("multi-device self-test failed: ") + what[x] + ", rank " + std::to_string(src) +
                                   " (device " + std::to_string(M.ch[src]->device) + ") -> rank " + std::to_string(g) +
                                   " (device " + std::to_string(M.ch[g]->device) + "): " +
                                   std::to_string(host[g][(size_t)x * G + src]) + " words differ -- the peer path between these devices "
                                   "does not deliver what was sent");
}

}  // namespace

g16_status multi_witness_upload(g16_ctx* parent, const uint64_t* w) {
  Multi& M = *parent->multi;
  const size_t wbytes = (size_t)parent->N * 32;
  M.witness_resident = false;
  std::vector<std::function<void(int)>> st;
  st.push_back([&](int g) {
    g16_ctx* c = M.ch[g];
    G16_HIP(hipSetDevice(c->device));
    G16_HIP(hipMemcpyAsync(c->w_dev.p, w, wbytes, hipMemcpyHostToDevice, c->stream));
    G16_HIP(hipStreamSynchronize(c->stream));
  });
  const int code = M.pool.run(st);
  if (code != G16_OK) parent->err = M.pool.first_error;
  else M.witness_resident = true;
  return code;
}

g16_ctx* multi_child(g16_ctx* parent, int index) {
  if (!parent || !parent->multi || index < 0 || index >= parent->multi->G) return nullptr;
  return parent->multi->ch[index];
}
int multi_links(const g16_ctx* parent, float* gbps, float* echo_us, int cap, size_t* probe_bytes) {
  if (!parent || !parent->multi) return -1;
  Multi& M = *parent->multi;
  if (!M.links_probed) {
    M.links_probed = true;
    try {
      multi_link_probe(M);
    } catch (const std::exception&) {  // a measurement: what could not be timed reads 0
      (void)hipGetLastError();
    }
  }
  const int n = M.G * M.G;
  for (int i = 0; i < n && i < cap; ++i) {
    if (gbps) gbps[i] = i < (int)M.link_gbps.size() ? M.link_gbps[i] : 0.f;
    if (echo_us) echo_us[i] = i < (int)M.link_echo_us.size() ? M.link_echo_us[i] : 0.f;
  }
  if (probe_bytes) *probe_bytes = M.link_probe_bytes;
  return M.G;
}

int multi_size(const g16_ctx* parent) { return (parent && parent->multi) ? parent->multi->G : 0; }

void multi_destroy(Multi* M) {
  if (!M) return;
  M->pool.shutdown();
  for (int g = 0; g < (int)M->dv.size(); ++g) {
    if (g < (int)M->ch.size() && M->ch[g]) (void)hipSetDevice(M->ch[g]->device);
    for (auto s : M->dv[g]->cs)
      if (s) {
        (void)hipStreamSynchronize(s);
        (void)hipStreamDestroy(s);
      }
    for (auto& v : M->dv[g]->arrived)
      for (auto e : v)
        if (e) (void)hipEventDestroy(e);
    for (auto& b : M->dv[g]->send) b.release();
    for (auto& b : M->dv[g]->recv) b.release();
  }
  for (auto it = M->ch.rbegin(); it != M->ch.rend(); ++it) g16_ctx_destroy(*it);  // borrowers before their lenders
  delete M;
}

g16_status multi_create(const g16_key_desc* key, const g16_csr* a, const g16_csr* b,
                        uint32_t num_constraints, const int* device_ids, int n_dev,
                        const g16_options* opt, g16_ctx** out, std::string* err) {
  Multi* M = new Multi();
  M->G = n_dev;
  M->ch.assign(n_dev, nullptr);
  for (int g = 0; g < n_dev; ++g) M->dv.emplace_back(new Multi::Dev());
  // fully sharded when the four-step split exists for this many ranks (wm_dist.h): world a power
  // of two and n1, n2 >= world; otherwise the witness map is replicated and only the MSMs shard
  uint64_t need = (uint64_t)num_constraints + key->n_public + 1;
  int k = 0;
  while (((uint64_t)1 << k) < need) ++k;
  const bool libsnark = opt && opt->reduction == G16_REDUCTION_LIBSNARK;
  M->dist = is_pow2(n_dev) && n_dev > 1 && !libsnark && (1u << (k / 2)) >= (uint32_t)n_dev &&
            !(opt && opt->dist_wm < 0);
  // bucket-range sharding of the witness-scalar MSMs when asked for; AUTO = point ranges (the two
  // cuts tie in rank time, point ranges hold 1/n_dev of the key per device: api.hip)
  M->buckets = opt && opt->shard == G16_SHARD_BUCKETS;
  M->pool.start(n_dev);
  // ranks that repeat a device ordinal (functional runs of an N-rank prover on fewer GPUs) borrow
  // the point planes of the first rank on that device under bucket-range sharding, where every
  // rank's planes are the whole key: created in a second stage, after their lenders
  std::vector<int> lender(n_dev, -1);
  for (int g = 0; g < n_dev; ++g)
    for (int f = 0; f < g && M->buckets; ++f)
      if (device_ids[f] == device_ids[g]) {
        lender[g] = f;
        break;
      }
  std::mutex dev_mu[64];
  auto make_child = [&](int g) {
    g16_options o{};
    if (opt) o = *opt;
    o.device = device_ids[g];
    o.rank = g;
    o.world = n_dev;
    o.dist_wm = M->dist ? 1 : 0;
    o.shard = M->buckets ? G16_SHARD_BUCKETS : G16_SHARD_POINTS;
    g16_ctx* c = nullptr;
    std::string cerr;
    g16_status s;
    {
      // children that share a device ordinal (functional runs) are created one after the other: each plans
      // its planes against the memory that is free at that moment (plan_msm_configs), and two planners
      // reading the same figure would both commit it.  Distinct devices still build in parallel.
      std::lock_guard<std::mutex> per_device(dev_mu[device_ids[g] & 63]);
      s = ctx_create_impl(key, a, b, num_constraints, &o, lender[g] >= 0 ? M->ch[lender[g]] : nullptr, &c, &cerr);
    }
    if (s != G16_OK)
      throw std::runtime_error("device " + std::to_string(device_ids[g]) + ": " + cerr +
                               (s == G16_ERR_DOMAIN_TOO_LARGE ? " PolynomialDegreeTooLarge" : ""));
    M->ch[g] = c;
    Multi::Dev& d = *M->dv[g];
    G16_HIP(hipSetDevice(c->device));
    // Copy streams are created with the witness-map (aux) stream's HIGH priority on purpose: HIP
    // maps streams onto a few hardware queues per priority level (4 by default, GPU_MAX_HW_QUEUES)
    // and streams that share a queue serialise.  A copy stream that lands on the queue of the main /
    // side / red stream would sit behind the whole MSM chain (measured: the exchange then starts
    // after the last accumulation, profiles/r02_rank8_timeline_before.txt); sharing a queue with aux
    // or with another copy stream costs nothing, they are one dependency chain anyway.
    int prio_lo = 0, prio_hi = 0;
    G16_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    d.cs.assign(n_dev, nullptr);
    for (auto& s2 : d.cs) G16_HIP(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, prio_hi));
    for (int x = 0; x < 2; ++x) {
      d.arrived[x].assign(n_dev, nullptr);
      for (auto& e : d.arrived[x]) G16_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (M->dist) {
      const size_t ints = c->wd.exchange_ints();
      for (int x = 0; x < 2; ++x) {
        d.send[x].alloc(ints);
        d.recv[x].alloc(ints);
      }
    }
  };
  std::vector<std::function<void(int)>> st;
  st.push_back([&](int g) {
    if (lender[g] < 0) make_child(g);
  });
  st.push_back([&](int g) {
    if (lender[g] >= 0) make_child(g);
  });
  int code = M->pool.run(st);
  uint32_t peer_state = 0;
  if (code == G16_OK) {
    try {
      if (M->dist) M->chunk_ints = M->ch[0]->wd.exchange_ints() / (size_t)n_dev;
      std::string why;
      peer_state = enable_peers(M->ch, &why);
      const char* req = getenv("G16_REQUIRE_PEER_ACCESS");
      if (peer_state != 1 && req && atoi(req) != 0) throw std::runtime_error("G16_REQUIRE_PEER_ACCESS: " + why);
      if (peer_state != 1)
        fprintf(stderr, "libg16_amd: %s -- the exchanges of every proof will be staged by the runtime\n", why.c_str());
      multi_selftest(*M);
    } catch (const std::exception& e) {
      code = G16_ERR_HIP;
      M->pool.first_error = e.what();
    }
  }
  if (code != G16_OK) {
    if (err) *err = M->pool.first_error;
    multi_destroy(M);
    return code;
  }
  g16_ctx* parent = new g16_ctx();
  parent->multi = M;
  parent->device = M->ch[0]->device;
  parent->world = 1;  // from the caller's point of view: g16_prove just works
  parent->N = M->ch[0]->N;
  parent->p = M->ch[0]->p;
  parent->n = M->ch[0]->n;
  parent->m = M->ch[0]->m;
  parent->num_inputs = M->ch[0]->num_inputs;
  parent->has_key = true;
  parent->peer_state = peer_state;
  *out = parent;
  return G16_OK;
}

g16_status multi_prove(g16_ctx* parent, const uint64_t r[4], const uint64_t s_[4], const void* w,
                       bool w_on_device, uint8_t proof_out[G16_PROOF_BYTES]) {
  Multi& M = *parent->multi;
  const size_t wbytes = (size_t)parent->N * 32;
  std::vector<const Fr*> wp(M.G, nullptr);

  // witness already in every device's staging buffer (g16_witness_upload): nothing to move
  const bool resident = w_on_device && M.witness_resident && w == (const void*)M.ch[0]->w_dev.p;
  if (!resident) M.witness_resident = false;  // the staging buffers are about to be overwritten
  auto stage_witness = [&](int g) {
    g16_ctx* c = M.ch[g];
    G16_HIP(hipSetDevice(c->device));
    if (resident) {
      wp[g] = c->w_dev.p;
    } else if (!w_on_device) {
      // every device pulls its own copy over its own PCIe link
      G16_HIP(hipMemcpyAsync(c->w_dev.p, w, wbytes, hipMemcpyHostToDevice, c->stream));
      wp[g] = c->w_dev.p;
    } else if (g == 0) {
      wp[g] = (const Fr*)w;
    } else {
      G16_HIP(hipMemcpyPeerAsync(c->w_dev.p, c->device, w, M.ch[0]->device, wbytes, c->stream));
      wp[g] = c->w_dev.p;
    }
  };
  auto stage_gather_finish = [&](int g) {
    g16_ctx* c = M.ch[g];
    G16_HIP(hipSetDevice(c->device));
    if (g != 0) {
      if (c->timer.enabled) {  // stage timers only: the proof itself needs no host wait here
        G16_HIP(hipStreamSynchronize(c->stream));
        rank_collect_times(c);
      }
      return;
    }
    hipStream_t s = c->stream;
    for (int src = 0; src < M.G; ++src) {
      G16_HIP(hipStreamWaitEvent(s, M.ch[src]->ev_part, 0));
      G16_HIP(hipMemcpyPeerAsync(c->gathered_dev() + (size_t)src * G16_PARTIAL_BYTES, c->device,
                                 M.ch[src]->part_dev(), M.ch[src]->device, G16_PARTIAL_BYTES, s));
    }
    rank_finish_enqueue(c, r, s_, M.G);
    G16_HIP(hipMemcpyAsync(proof_out, c->out_dev.p, G16_PROOF_BYTES, hipMemcpyDeviceToHost, s));
    G16_HIP(hipStreamSynchronize(s));
    rank_collect_times(c);
  };

  std::vector<std::function<void(int)>> st;
  if (M.dist) {
    st.push_back([&](int g) {
      stage_witness(g);
      rank_phase1_enqueue(M.ch[g], r, s_, wp[g], M.dv[g]->send[0].p);
      push_chunks(M, g, 0);
    });
    st.push_back([&](int g) {
      G16_HIP(hipSetDevice(M.ch[g]->device));
      await_chunks(M, g, 0, M.ch[g]->aux);
      rank_phase2_enqueue(M.ch[g], M.dv[g]->recv[0].p, M.dv[g]->send[1].p);
      push_chunks(M, g, 1);
    });
    st.push_back([&](int g) {
      G16_HIP(hipSetDevice(M.ch[g]->device));
      await_chunks(M, g, 1, M.ch[g]->aux);
      rank_phase3_enqueue(M.ch[g], M.dv[g]->recv[1].p);
    });
  } else {
    st.push_back([&](int g) {
      stage_witness(g);
      rank_partial_enqueue(M.ch[g], r, s_, wp[g]);
    });
  }
  st.push_back(stage_gather_finish);
  const int code = M.pool.run(st);
  if (code != G16_OK) {
    parent->err = M.pool.first_error;
    // leave no work in flight behind a failed proof
    for (auto* c : M.ch) {
      (void)hipSetDevice(c->device);
      (void)hipDeviceSynchronize();
    }
  }
  return code;
}

}  // namespace g16
