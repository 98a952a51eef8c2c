// emu_hip.cpp -- TEST INFRASTRUCTURE ONLY: fiber scheduler behind emu_hip.h (see the header).
#include "emu_hip.h"

#include <stdio.h>
#include <sys/mman.h>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {

unsigned char* dyn_smem = nullptr;

namespace {
constexpr size_t kStack = 256 * 1024;
enum Wait { RUN = 0, BLOCK_BAR = 1, WAVE_BAR = 2, DONE = 3 };
struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  int wait = DONE;
};
std::vector<Fiber> fibers;
ucontext_t sched;
int cur = -1;
const std::function<void()>* body = nullptr;
std::vector<uint8_t> pred;       // per thread: ballot predicate
std::vector<uint32_t> shval;     // per thread: shuffle payload
std::vector<unsigned char> dyn_buf;

void trampoline() {
  (*body)();
  fibers[cur].wait = DONE;
  pred[cur] = 0;
  swapcontext(&fibers[cur].ctx, &sched);
}

void yield_as(int w) {
  fibers[cur].wait = w;
  swapcontext(&fibers[cur].ctx, &sched);
}
}  // namespace

void barrier() { yield_as(BLOCK_BAR); }
static void wave_sync() { yield_as(WAVE_BAR); }

uint64_t wave_ballot(bool p) {
  pred[cur] = p ? 1 : 0;
  wave_sync();
  int base = (cur / 64) * 64;
  uint64_t m = 0;
  for (int l = 0; l < 64 && base + l < (int)fibers.size(); ++l)
    if (pred[base + l]) m |= (uint64_t)1 << l;
  wave_sync();
  return m;
}

uint32_t wave_shfl(uint32_t v, int src) {
  shval[cur] = v;
  wave_sync();
  int base = (cur / 64) * 64;
  int s = base + (src & 63);
  uint32_t r = (s < (int)fibers.size()) ? shval[s] : v;
  wave_sync();
  return r;
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& fn) {
  const int nt = (int)(block.x * block.y * block.z);
  if ((int)fibers.size() < nt) {
    size_t old = fibers.size();
    fibers.resize(nt);
    for (size_t i = old; i < fibers.size(); ++i) {
      fibers[i].stack = (char*)mmap(nullptr, kStack, PROT_READ | PROT_WRITE,
                                    MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (fibers[i].stack == MAP_FAILED) { perror("emu mmap"); abort(); }
    }
  }
  pred.assign(fibers.size(), 0);
  shval.assign(fibers.size(), 0);
  if (dyn_buf.size() < smem + 64) dyn_buf.resize(smem + 64);
  dyn_smem = (unsigned char*)(((uintptr_t)dyn_buf.data() + 63) & ~(uintptr_t)63);
  body = &fn;
  blockDim = block;
  gridDim = grid;
  std::vector<Fiber> saved;  // keep only the first nt fibers active
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        for (int t = 0; t < nt; ++t) {
          Fiber& f = fibers[t];
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = kStack;
          f.ctx.uc_link = &sched;
          makecontext(&f.ctx, trampoline, 0);
          f.wait = RUN;
          pred[t] = 0;
        }
        for (;;) {
          bool any_run = false;
          for (int t = 0; t < nt; ++t) {
            if (fibers[t].wait != RUN) continue;
            any_run = true;
            cur = t;
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            swapcontext(&sched, &fibers[t].ctx);
          }
          if (any_run) continue;
          // nobody runnable: release wave barriers whose alive lanes have all arrived
          bool released = false;
          for (int w = 0; w * 64 < nt; ++w) {
            int lo = w * 64, hi = lo + 64 < nt ? lo + 64 : nt;
            bool all = true, some = false;
            for (int t = lo; t < hi; ++t) {
              if (fibers[t].wait == WAVE_BAR) some = true;
              else if (fibers[t].wait != DONE) all = false;
            }
            if (some && all) {
              for (int t = lo; t < hi; ++t)
                if (fibers[t].wait == WAVE_BAR) fibers[t].wait = RUN;
              released = true;
            }
          }
          if (released) continue;
          bool all_bar = true, some_bar = false;
          for (int t = 0; t < nt; ++t) {
            if (fibers[t].wait == BLOCK_BAR) some_bar = true;
            else if (fibers[t].wait != DONE) all_bar = false;
          }
          if (some_bar && all_bar) {
            for (int t = 0; t < nt; ++t)
              if (fibers[t].wait == BLOCK_BAR) fibers[t].wait = RUN;
            continue;
          }
          if (some_bar) { fprintf(stderr, "emu: deadlock (mixed barriers)\n"); abort(); }
          break;  // all DONE
        }
      }
  cur = -1;
}

}  // namespace emu
