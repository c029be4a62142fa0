// TEST INFRASTRUCTURE -- not part of the product, never linked into libzpaq_amd.so.
//
// A host-side wavefront emulator for the specialised coder's kernel template
// (zpaq_amd/csrc/device/spec_kernel.h + the source host/codegen.cpp generates).
// It lets the CPU test-suite (-m "not gpu") execute the SAME device source the
// GPU runs -- lane-parallel model, DPP/readlane cross-lane traffic, LDS layout,
// dummy-slot addressing, HCOMP translation -- and compare its output bit for bit
// with the oracle, without a GPU.  It checks logic, not timing.
//
// How: a workgroup is 64*W cooperative fibers on one OS thread (one per lane).
// A fiber runs until it reaches a cross-lane operation or a barrier, publishes
// its operand, and yields; when all 64 lanes of its wavefront have arrived the
// wavefront continues and every lane reads what it needs.  This is exact as long
// as cross-lane operations sit in wave-uniform control flow, which the kernel
// guarantees (and the hardware needs for the same reason).
//
// The emulated primitives follow the gfx9 ISA definitions:
//   v_readlane / v_readfirstlane, ds_bpermute (__shfl), DPP row_shr:n,
//   row_bcast:15, row_bcast:31, wave_shr:1 with row_mask / bank_mask / bound_ctrl.
#pragma once
#define __HIPCC_RTC__ 1   // makes layout.h / spec_kernel.h skip <hip/hip_runtime.h>
#define ZPQ_EMU 1

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace __hip_internal {
using ::int8_t; using ::int16_t; using ::int32_t; using ::int64_t;
using ::uint8_t; using ::uint16_t; using ::uint32_t; using ::uint64_t;
}

#define __global__
#define __device__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(x)
#define __shared__ static

struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }

namespace emu {

struct Dim3 { unsigned x, y, z; };
extern Dim3 g_threadIdx, g_blockIdx, g_blockDim;   // of the fiber that is running

// block until all 64 lanes of this fiber's wavefront are here (returns the lane's exchange row)
int* wave_exchange(int value);          // publish `value`, wait, return pointer to the 64 published values
void block_barrier();                   // __syncthreads
void spin_yield();                      // inside a poll loop on LDS / memory another wavefront writes
void wave_reconverge();                 // lanes that returned early out of a divergent region wait for the rest of their wavefront
void* wg_lds(size_t bytes);             // the running workgroup's LDS (kernels that run as a GRID of live workgroups: run_grid)
unsigned long long ticks();             // the emulated clock: scheduler rounds
int lane_id();

typedef void (*KernelFn)(void* args);
// run `fn(args)` once per thread of a workgroup of `threads` threads, block index `bx`
void run_workgroup(KernelFn fn, void* args, unsigned threads, unsigned bx);
// all workgroups block0 .. block0 + nblocks - 1 alive together (a persistent launch whose workgroups wait for each other)
void run_grid(KernelFn fn, void* args, unsigned threads, unsigned nblocks, unsigned block0);
unsigned long cross_lane_ops();         // statistics: exchanges executed by lane 0 of wave 0

}  // namespace emu

#define threadIdx emu::g_threadIdx
#define blockIdx emu::g_blockIdx
#define blockDim emu::g_blockDim

static inline void __syncthreads() { emu::block_barrier(); }

static inline int __builtin_amdgcn_readlane(int v, int lane) { return emu::wave_exchange(v)[lane & 63]; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return emu::wave_exchange(v)[0]; }   // all lanes active here
static inline int __shfl(int v, int src) { return emu::wave_exchange(v)[src & 63]; }
static inline unsigned __shfl(unsigned v, int src) { return (unsigned)emu::wave_exchange((int)v)[src & 63]; }
namespace emu {
static inline bool wave_any(bool x) {        // true if x holds in any lane that reaches this point
  const int* v = wave_exchange(x ? 1 : 0);
  // lanes that already left the kernel keep a stale slot; they published 0 or 1 at an earlier exchange --
  // the pipe kernels call this before any lane can exit, so all 64 slots are current
  for (int i = 0; i < 64; ++i) if (v[i]) return true;
  return false;
}
}  // namespace emu
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool x) {      // (every lane of the wavefront is here)
  const int* v = emu::wave_exchange(x ? 1 : 0);
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) if (v[i]) m |= 1ull << i;
  return m;
}
static inline int __mul24(int a, int b) {
  const int x = (int)((unsigned)a << 8) >> 8, y = (int)((unsigned)b << 8) >> 8;
  return (int)((unsigned)x * (unsigned)y);
}

static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }

// v_mov_b32_dpp semantics (gfx9): returns the new value of the destination whose previous content is `old`
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int* v = emu::wave_exchange(src);
  const int lane = emu::lane_id();
  const int row = lane >> 4, bank = (lane >> 2) & 3, in_row = lane & 15;
  if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) return old;   // lane not enabled for writing
  int from = -1;                                                            // -1: no valid source lane
  if (ctrl >= 0x111 && ctrl <= 0x11F) {                                     // row_shr:n
    const int n = ctrl - 0x110;
    if (in_row >= n) from = lane - n;
  } else if (ctrl >= 0x101 && ctrl <= 0x10F) {                              // row_shl:n
    const int n = ctrl - 0x100;
    if (in_row + n < 16) from = lane + n;
  } else if (ctrl >= 0 && ctrl <= 0xFF) {                                   // quad_perm:[a,b,c,d]
    from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  } else if (ctrl == 0x140) {                                               // row_mirror
    from = (lane & ~15) | (15 - in_row);
  } else if (ctrl == 0x141) {                                               // row_half_mirror
    from = (lane & ~7) | (7 - (lane & 7));
  } else if (ctrl == 0x138) {                                               // wave_shr:1
    if (lane >= 1) from = lane - 1;
  } else if (ctrl == 0x130) {                                               // wave_shl:1
    if (lane < 63) from = lane + 1;
  } else if (ctrl == 0x142) {                                               // row_bcast:15 (lane 15 of each row -> next row)
    if (row >= 1) from = row * 16 - 1;
  } else if (ctrl == 0x143) {                                               // row_bcast:31 (lane 31 -> rows 2 and 3)
    if (row >= 2) from = 31;
  } else {
    fprintf(stderr, "wave_emu: DPP control 0x%x not modelled\n", ctrl);
    abort();
  }
  if (from < 0) return bound_ctrl ? 0 : old;
  return v[from];
}
