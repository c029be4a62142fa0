// TEST INFRASTRUCTURE -- runtime of the host-side wavefront emulator (see wave_emu.h).
// One OS thread, one fiber per lane, hand-rolled x86-64 context switch.
#include <cstring>
#include <cstdlib>
#include <utility>
#include <vector>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <vector>

#include "wave_emu.h"

namespace emu {

Dim3 g_threadIdx{0, 0, 0}, g_blockIdx{0, 0, 0}, g_blockDim{1, 1, 1};

enum State { READY, WAIT_WAVE, WAIT_BLOCK, PARKED, DONE };

struct Fiber {
  void* sp = nullptr;
  void* stack = nullptr;
  unsigned tid = 0;            // thread index inside its workgroup
  unsigned wg = 0;             // workgroup (index into the grid being run)
  State state = READY;
  unsigned long nops = 0;      // cross-lane exchanges done (parity selects the exchange buffer)
};

static const size_t kStack = 512 * 1024;
static std::vector<Fiber> g_fibers;
static Fiber* g_cur = nullptr;
static void* g_sched_sp = nullptr;
static KernelFn g_fn = nullptr;
static void* g_args = nullptr;
static std::vector<int> g_xbuf;          // [workgroup][wave][2][64]
static unsigned long g_ops = 0;
static unsigned g_threads = 0, g_waves = 0, g_block0 = 0;
static std::vector<std::vector<unsigned char>> g_lds;      // per workgroup of the grid (wg_lds)
static unsigned long long g_ticks = 0;                     // scheduler rounds: the emulated clock

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch,.-emu_switch
)");

static void to_scheduler() { emu_switch(&g_cur->sp, g_sched_sp); }

static void trampoline() {
  g_fn(g_args);
  g_cur->state = DONE;
  to_scheduler();
  abort();   // a finished fiber is never resumed
}

int lane_id() { return (int)(g_cur->tid & 63u); }

int* wave_exchange(int value) {
  Fiber* f = g_cur;
  const unsigned wave = f->tid >> 6, lane = f->tid & 63u;
  int* buf = &g_xbuf[(((size_t)f->wg * g_waves + wave) * 2 + (f->nops & 1)) * 64];
  buf[lane] = value;
  ++f->nops;
  if (f->tid == 0 && f->wg == 0) ++g_ops;
  f->state = WAIT_WAVE;
  to_scheduler();
  return buf;
}

void block_barrier() {
  g_cur->state = WAIT_BLOCK;
  to_scheduler();
}

// a lane polling memory another wavefront writes: let everybody else run, then look again
void spin_yield() {
  g_cur->state = READY;
  to_scheduler();
}

// The lanes of a wavefront that left a divergent region early (a `return` out of a unit function whose other lanes still
// exchange values) wait here until the rest of the wavefront arrives: what the EXEC mask does on the hardware.  While they
// wait they do not count as participants of the others' cross-lane operations.
void wave_reconverge() {
  g_cur->state = PARKED;
  to_scheduler();
}

void* wg_lds(size_t bytes) {
  std::vector<unsigned char>& v = g_lds[g_cur->wg];
  if (v.size() < bytes) v.resize(bytes, 0xCD);
  return v.data();
}

unsigned long long ticks() { return g_ticks; }

unsigned long cross_lane_ops() { return g_ops; }

void run_grid(KernelFn fn, void* args, unsigned threads, unsigned nblocks, unsigned block0) {
  g_fn = fn;
  g_args = args;
  g_threads = threads;
  g_block0 = block0;
  g_blockDim = Dim3{threads, 1, 1};
  const unsigned waves = (threads + 63) / 64;
  g_waves = waves;
  g_xbuf.assign((size_t)nblocks * waves * 2 * 64, 0);
  g_lds.assign(nblocks, {});
  const unsigned total = threads * nblocks;
  g_fibers.assign(total, Fiber{});
  for (unsigned i = 0; i < total; ++i) {
    Fiber& f = g_fibers[i];
    f.tid = i % threads;
    f.wg = i / threads;
    f.stack = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (f.stack == MAP_FAILED) { perror("mmap"); abort(); }
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // keeps the entry frame 16-byte aligned like a real call
    *--sp = (void*)&trampoline;      // `ret` target of the first switch
    for (int k = 0; k < 6; ++k) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
    f.sp = sp;
  }
  // The order in which the lanes run between two cross-lane operations is the emulator's choice; the hardware runs them
  // instruction by instruction together.  A kernel whose lanes communicate through memory inside such an interval gives
  // different results for different orders: ZPQ_EMU_ORDER=reverse | shuffle[:seed] lets the tests try others.
  static const char* order_env = getenv("ZPQ_EMU_ORDER");
  static const int order_mode = !order_env ? 0 : (!strncmp(order_env, "reverse", 7) ? 1 : (!strncmp(order_env, "shuffle", 7) ? 2 : 0));
  static unsigned long long order_rng = order_env && strchr(order_env, ':') ? strtoull(strchr(order_env, ':') + 1, nullptr, 10) * 2654435761ull + 1 : 12345;
  std::vector<unsigned> order(total);
  for (unsigned t = 0; t < total; ++t) order[t] = order_mode == 1 ? total - 1 - t : t;
  for (;;) {
    bool ran = false;
    ++g_ticks;
    if (order_mode == 2)
      for (unsigned t = total; t > 1; --t) {
        order_rng = order_rng * 6364136223846793005ull + 1442695040888963407ull;
        std::swap(order[t - 1], order[(unsigned)((order_rng >> 33) % t)]);
      }
    for (unsigned oi = 0; oi < total; ++oi) {
      Fiber& f = g_fibers[order[oi]];
      if (f.state != READY) continue;
      g_cur = &f;
      g_threadIdx = Dim3{f.tid, 0, 0};
      g_blockIdx = Dim3{block0 + f.wg, 0, 0};
      emu_switch(&g_sched_sp, f.sp);
      ran = true;
    }
    unsigned done_all = 0;
    for (unsigned wg = 0; wg < nblocks; ++wg) {
      const unsigned base = wg * threads;
      // release wavefronts whose live lanes have all arrived at the same exchange (parked lanes stand aside), and parked
      // wavefronts once every lane that has not finished is parked
      unsigned done = 0, at_barrier = 0;
      for (unsigned w = 0; w < waves; ++w) {
        unsigned waiting = 0, live = 0, parked = 0, notdone = 0;
        const unsigned lo = w * 64, hi = lo + 64 < threads ? lo + 64 : threads;
        for (unsigned t = lo; t < hi; ++t) {
          const State st = g_fibers[base + t].state;
          if (st == DONE) { ++done; continue; }
          ++notdone;
          if (st == PARKED) { ++parked; continue; }
          ++live;
          if (st == WAIT_WAVE) ++waiting;
          if (st == WAIT_BLOCK) ++at_barrier;
        }
        if (live && waiting == live) {
          unsigned long nops = 0;
          for (unsigned t = lo; t < hi; ++t)
            if (g_fibers[base + t].state == WAIT_WAVE) { g_fibers[base + t].state = READY; nops = g_fibers[base + t].nops; }
          if (parked) {      // parked lanes are inactive: they contribute nothing to this exchange (ballot: 0)
            int* buf = &g_xbuf[(((size_t)wg * waves + w) * 2 + ((nops - 1) & 1)) * 64];
            for (unsigned t = lo; t < hi; ++t)
              if (g_fibers[base + t].state == PARKED) buf[t - lo] = 0;
          }
          ran = true;
        } else if (notdone && parked == notdone) {
          unsigned long nops = 0;        // the lanes rejoin with one exchange count (the ones that left early did fewer)
          for (unsigned t = lo; t < hi; ++t)
            if (g_fibers[base + t].state == PARKED && g_fibers[base + t].nops > nops) nops = g_fibers[base + t].nops;
          for (unsigned t = lo; t < hi; ++t)
            if (g_fibers[base + t].state == PARKED) { g_fibers[base + t].state = READY; g_fibers[base + t].nops = nops; }
          ran = true;
        }
      }
      done_all += done;
      if (at_barrier && at_barrier + done == threads) {
        for (unsigned t = 0; t < threads; ++t)
          if (g_fibers[base + t].state == WAIT_BLOCK) g_fibers[base + t].state = READY;
        ran = true;
      }
    }
    if (done_all == total) break;
    if (!ran) { fprintf(stderr, "wave_emu: deadlock -- the lanes of a wavefront disagree about the next cross-lane operation "
                           "(one sits in divergent control flow) or a barrier is not reached by every thread\n"); abort(); }
  }
  for (Fiber& f : g_fibers) munmap(f.stack, kStack);
  g_fibers.clear();
  g_lds.clear();
}

void run_workgroup(KernelFn fn, void* args, unsigned threads, unsigned bx) { run_grid(fn, args, threads, 1, bx); }

}  // namespace emu
