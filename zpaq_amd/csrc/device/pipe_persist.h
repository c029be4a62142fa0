// PERSISTENT form of the pipelined encoder: the units of pipe_kernel.h inside ONE launch.
//
// pipe_kernel.h's six kernels are launched once per step (chunk of PIPE_C input bytes): 12 384 dispatches for the
// 1024 x 1 MiB headline, and -- what costs more -- a kernel of a step lasts as long as its SLOWEST workgroup.  The
// placement trace of round 3 (profiles/r03/call2_summary.txt) shows what that means: the MIX workgroups of a step take
// 1.2 ms on average and 2.4 ms at worst (who shares a SIMD, who queues behind whom in the memory system), the kernel
// 1.76 ms, the step 2.07 ms.  Every unit of every group pays for the unluckiest workgroup of the whole GPU, every step.
//
// Here a unit (one component of one group of PIPE_G blocks) is a wavefront that lives for the whole sequence and walks
// its chunks in a loop.  What it needs from other units it waits for by itself:
//   * before chunk c it polls the progress counters of the units whose streams it reads (they must have finished chunk c)
//     and of the units that read ITS streams (they must have finished chunk c - PIPE_S: the ring slot it overwrites);
//   * after chunk c it drains its stores and bumps its own counter.
// A unit is delayed only by the units of ITS OWN group it really depends on, a slow chunk is made up for by the ring's
// slack, and the sequence runs at the pace of the mean, not of the per-step maximum over 2 000 workgroups.
//
// Visibility between workgroups (MI355X_MICROARCH.md, "inter-workgroup visibility"): stream elements are stored
// write-through (sc1: PipeLane::put_*), the producer drains them (s_waitcnt vmcnt(0)) and then publishes its counter with
// a relaxed agent-scope atomic; the consumer polls relaxed, takes ONE agent-scope acquire (buffer_inv sc1: drops this CU's
// L1) and reads with plain loads.  Nothing depends on where a workgroup runs.  Everything else a unit touches (its tables
// in the arena, its state words) is touched by that wavefront alone for the whole launch.
//
// Residency.  Units spin, so every workgroup of the launch must be resident at the same time.  The generator packs the
// units of a group into PS_WPG workgroups of PS_WAVES wavefronts whose LDS (shared constant tables + the private tables of
// the HCOMP / ICM / ISSE units among them) fits a CU; the engine asks the occupancy API how many such workgroups the device
// holds and never launches more (a larger batch runs in several rounds; a chain that cannot be packed runs on the six
// kernels).  Should a workgroup nevertheless not get a CU, the pollers give up after PipeArgs::timeout_ticks without
// progress, raise the abort word and exit; the engine then codes the batch with the six kernels.  The launch cannot hang.
//
// Blocks-to-XCD: with PipeArgs::spread = 8 the workgroups of a group are 8 apart in the grid, so they share an XCD (and its L2)
// when the dispatcher deals workgroups round-robin over the XCDs, as it is observed to do (pipe_persist_body).  A speed choice only.
#pragma once
#include "pipe_kernel.h"

#if !defined(ZPQ_EMU)
#ifndef ZPQ_PERSIST_LDS_BYTES
#error "the generated source defines ZPQ_PERSIST_LDS_BYTES before it includes pipe_persist.h"
#endif
namespace zpq { __shared__ __attribute__((aligned(16))) unsigned char persist_lds[ZPQ_PERSIST_LDS_BYTES]; }
#endif

namespace zpq {

// constant tables every unit of a workgroup shares (loaded once per workgroup)
struct PipeRO {
  int dt[1024];
  unsigned short dt2k[256];
  PipeSquash squash;
  PipeStretch stretch;
  unsigned char ns[1024];
};

#ifdef ZPQ_EMU
#define ZPQ_PERSIST_LDS(bytes) ((unsigned char*)emu::wg_lds(bytes))
__device__ __forceinline__ unsigned pipe_prog_load(const unsigned* p) { return *(volatile const unsigned*)p; }
__device__ __forceinline__ void pipe_prog_add(unsigned* p) { *(volatile unsigned*)p = *(volatile unsigned*)p + 1u; }
__device__ __forceinline__ void pipe_flag_set(unsigned* p, unsigned v) { *(volatile unsigned*)p = v; }
__device__ __forceinline__ void pipe_drain_stores() {}
__device__ __forceinline__ void pipe_acquire() {}
__device__ __forceinline__ unsigned long long pipe_clock() { return emu::ticks(); }
__device__ __forceinline__ void pipe_nap() { emu::spin_yield(); }
__device__ __forceinline__ void pipe_reconverge() { emu::wave_reconverge(); }
#else
// the workgroup's LDS is ONE array at namespace scope: the unit functions are real calls (one function per unit keeps the
// compile time of the launch near that of the six kernels), and a callee that names the array itself knows it is LDS
#define ZPQ_PERSIST_LDS(bytes) (zpq::persist_lds)
__device__ __forceinline__ unsigned pipe_prog_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pipe_prog_add(unsigned* p) { (void)__hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pipe_flag_set(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pipe_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pipe_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ unsigned long long pipe_clock() { return __builtin_amdgcn_s_memrealtime(); }      // 100 MHz
__device__ __forceinline__ void pipe_nap() { __builtin_amdgcn_s_sleep(32); }
__device__ __forceinline__ void pipe_reconverge() {}
#endif

// Light units whose table lives in the LDS inside the persistent launch: words per block (0: the table stays in the arena).
// host/codegen.cpp plan_persistent reserves the same amount (x PIPE_G lanes x 4 bytes).
static const unsigned kPipeLightLdsMaxWords = 512u;
template <class Chain>
__device__ __forceinline__ constexpr unsigned pipe_light_lds_words(int lk, int I) {
  const CompK c = Chain::comp[I];
  if (lk == PK_CM && c.mask0 + 1u <= kPipeLightLdsMaxWords && c.mask0 >= 3u) return c.mask0 + 1u;
  if (lk == PK_MIX2 && c.mask0 != 0u && c.mask0 + 1u <= kPipeLightLdsMaxWords && c.mask0 >= 3u) return c.mask0 + 1u;
  return 0u;
}

// what host/codegen.cpp plan_persistent decided for the chain (absent in older generated sources: off)
template <class Chain, class = void> struct PipeCoderFast { static constexpr bool value = false; };
template <class Chain> struct PipeCoderFast<Chain, decltype((void)Chain::PS_CODER_FAST)> { static constexpr bool value = Chain::PS_CODER_FAST; };
template <class Chain, class = void> struct PipeAhead { static constexpr int value = 0; };          // stream elements read value + 1 bytes ahead
template <class Chain> struct PipeAhead<Chain, decltype((void)Chain::PS_AHEAD)> { static constexpr int value = Chain::PS_AHEAD; };
template <class Chain, class = void> struct PipeRowRing { static constexpr bool value = false; };   // lane-per-block ROW units two bytes ahead in the table
template <class Chain> struct PipeRowRing<Chain, decltype((void)Chain::PS_ROW_RING)> { static constexpr bool value = Chain::PS_ROW_RING; };
template <class Chain, class = void> struct PipeSmallChain { static constexpr bool value = false; };
template <class Chain> struct PipeSmallChain<Chain, decltype((void)Chain::PS_SMALL)> { static constexpr bool value = Chain::PS_SMALL; };

// ROW unit `role` with a lane per nibble (pipe_row_halves): small chains (PS_ROW_HALVES, where the generated source says otherwise:
// variant 3 of a larger chain has the small chains' LDS-rich maps but not this), groups of 32 blocks, tables of 8 KiB or more
// ICM maps with the whole stretch table (PS_ICM_FULL where the generated source says otherwise: variant 1 of a larger chain keeps the compact one)
template <class Chain, class = void> struct PipeIcmFull { static constexpr bool value = PipeSmallChain<Chain>::value; };
template <class Chain> struct PipeIcmFull<Chain, decltype((void)Chain::PS_ICM_FULL)> { static constexpr bool value = Chain::PS_ICM_FULL; };
template <class Chain, class = void> struct PipeRowHalves { static constexpr bool value = PipeSmallChain<Chain>::value; };
template <class Chain> struct PipeRowHalves<Chain, decltype((void)Chain::PS_ROW_HALVES)> { static constexpr bool value = Chain::PS_ROW_HALVES; };
template <class Chain>
__device__ __forceinline__ constexpr bool pipe_row_in_halves(int role) {
  return PipeRowHalves<Chain>::value && Chain::PIPE_G == 32u && Chain::comp[Chain::ROW_COMP[role]].mask1 + 1u >= 8192u;
}

// a pointer every lane holds the same value of, as a value the compiler knows to be wave-uniform (function arguments and
// what is loaded through them arrive in vector registers; a buffer descriptor built from one would be waterfalled)
template <class T>
__device__ __forceinline__ T* pipe_uniform(T* p) {
#ifdef ZPQ_EMU
  return p;
#else
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
#endif
}

// Wait until every unit slot SLOT depends on is far enough for chunk c.  Lane d looks after dependency d.  false: the
// launch was aborted (by this wavefront's own watchdog or by another's).
template <class Chain>
__device__ __forceinline__ bool pipe_wait(const PipeArgs& a, const unsigned* prog, unsigned c, int lane, int SLOT) {
  const int nd = Chain::PS_NDEP[SLOT], d0 = Chain::PS_DEP0[SLOT];
  for (int base = 0; base < nd; base += 64) {
    const int d = base + lane;
    const bool mine = d < nd;
    const int di = d0 + (mine ? d : 0);
    const unsigned unit = (unsigned)Chain::PS_DEP_UNIT[di];
    const int need = mine ? Chain::PS_DEP_MULT[di] * ((int)c + 1 - Chain::PS_DEP_LAG[di]) : 0;
    bool ok = need <= 0;
    if (!ok) ok = (int)pipe_prog_load(prog + unit) >= need;
    if (!pipe_any(!ok)) continue;
    const unsigned long long t0 = pipe_clock();
    unsigned spins = 0;
    for (;;) {
      pipe_nap();
      if (!ok) ok = (int)pipe_prog_load(prog + unit) >= need;
      if (!pipe_any(!ok)) break;
      if ((++spins & 63u) == 0u) {
        bool stop = pipe_prog_load(a.ctl) != 0u;
        if (!stop && pipe_clock() - t0 > (unsigned long long)a.timeout_ticks) {
          if (lane == 0) { pipe_flag_set(a.ctl, 1u); pipe_flag_set(a.ctl + 1, (unsigned)SLOT); pipe_flag_set(a.ctl + 2, c); }
          stop = true;
        }
        if (pipe_any(stop)) return false;
      }
    }
  }
  pipe_acquire();
  return true;
}

// Arrival handshake.  The units spin on each other, so a launch makes progress only if ALL its workgroups are resident at the
// same time.  The engine sizes the grid from the occupancy API, but it cannot know what else holds compute units (another
// process, the host application's own kernels): a workgroup left in the queue would let every unit that depends on it spin
// until the watchdog (seconds), with the arenas half coded.  So every workgroup first says it is there (ctl[3]) and no
// wavefront touches anything before all of them have: if the count stops short for arrive_ticks, the launch is given up
// with flag 2 -- nothing has been written, the engine hands the batch to the step kernels at once (no re-initialisation).
// false: do not start.
__device__ __forceinline__ bool pipe_arrived(const PipeArgs& a, int lane) {
  if (!a.arrive_need) return true;
  unsigned seen = pipe_prog_load(a.ctl + 3);
  if (seen >= a.arrive_need) return true;
  unsigned long long t0 = pipe_clock();
  for (;;) {
    pipe_nap();
    const unsigned now = pipe_prog_load(a.ctl + 3);
    if (now >= a.arrive_need) return true;
    if (pipe_prog_load(a.ctl) != 0u) return false;
    if (now != seen) { seen = now; t0 = pipe_clock(); }
    else if (pipe_clock() - t0 > (unsigned long long)a.arrive_ticks) {
      if (lane == 0) pipe_flag_set(a.ctl, 2u);
      return false;
    }
  }
}

template <class Chain>
__device__ __forceinline__ void pipe_publish(unsigned* prog, int unit, int lane) {
  pipe_drain_stores();
  if (lane == 0) pipe_prog_add(prog + unit);
}

// The whole life of one unit wavefront: unit type `kind` (0 hcomp, 1 row, 2 light, 3 icm, 4 isse, 5 mix), `role` = which of
// that type's units (for the code: the wavefronts of a unit that has several -- MIX lane groups, a light unit's bit-lane
// workgroups -- share ONE function and differ in `sub`, read from the slot tables).
template <class Chain, int kind, int role>
__device__ __attribute__((noinline)) void pipe_persist_unit(const PipeArgs& a, unsigned g, int lane, int SLOT) {
  unsigned char* const lds = ZPQ_PERSIST_LDS(Chain::PS_LDS_BYTES);
  const PipeRO& ro = *(const PipeRO*)lds;
  const int sub = Chain::PS_SUB[SLOT], unit = Chain::PS_UNIT[SLOT];
  constexpr unsigned G = Chain::PIPE_G;
  unsigned* const prog = pipe_uniform(a.prog + (unsigned long long)g * (unsigned)Chain::PS_NUNIT);
  const unsigned nchunks = a.group_chunks[g];
  unsigned char* const priv = lds + Chain::PS_LDS[SLOT];
  PipeLane<Chain> L;
  unsigned q = 0, B = 0, mix_bl = 0;
  // lane -> block
  if constexpr (kind == 0) {
    constexpr int HL = Chain::HCOMP_LANES < (int)G ? Chain::HCOMP_LANES : (int)G;
    const bool mine = lane < HL;
    L.bind(a, g, (unsigned)sub * HL + (mine ? (unsigned)lane : 0u), mine);
  } else if constexpr (kind == 5) {
    constexpr int QL = Chain::MIX_QL[role];
    if constexpr (Chain::MIX_BITS != 0) {
      constexpr int BPW = 64 / QL / 8;
      const unsigned pair = (unsigned)lane / QL;
      q = (unsigned)lane % QL; B = pair & 7u;
      L.bind(a, g, pipe_opaque((unsigned)sub * BPW + (pair >> 3)), true);
    } else {
      constexpr int NH = Chain::PS_MIX_NH;                           // lane groups per block: 2 = one for bits 0 .. 3, one for bits 4 .. 7
      constexpr int BPW = 64 / (QL * NH) < (int)G ? 64 / (QL * NH) : (int)G;       // all 64 lanes: 64 / (QL NH) blocks per wavefront
      const unsigned bl = (unsigned)lane / (QL * NH);
      q = (unsigned)lane % QL; B = ((unsigned)lane / QL) % NH;       // (B: the half)
      mix_bl = bl;
      const bool okl = bl < (unsigned)BPW;
      L.bind(a, g, (unsigned)sub * BPW + (okl ? bl : 0u), okl);
    }
  } else if constexpr (kind == 2 && Chain::LIGHT_KIND[role < 0 ? 0 : role] >= PK_CM_BITS) {
    B = (unsigned)lane & 7u;
    const unsigned gl = (unsigned)sub * 8u + ((unsigned)lane >> 3);
    L.bind(a, g, gl < G ? gl : 0u, gl < G);
  } else if constexpr (kind == 1 && pipe_row_in_halves<Chain>(role < 0 ? 0 : role)) {
    L.bind(a, g, (unsigned)lane & 31u, true);                       // lanes b and 32 + b: the two nibbles of block b's bytes
  } else {
    const bool okl = (unsigned)lane < G;
    L.bind(a, g, okl ? (unsigned)lane : 0u, okl);
  }
  if (!L.live) L.idle();
  L.gb = pipe_uniform(L.gb);
  // a.trace (ZPAQ_AMD_PERSIST_PROF=<file>): per (group, slot) four words -- ticks spent waiting, ticks spent working, chunks, slot
  unsigned long long t_wait = 0, t_work = 0, t_mark = a.trace ? pipe_clock() : 0ull;
  for (unsigned c = 0; c < nchunks; ++c) {
    if (!pipe_wait<Chain>(a, prog, c, lane, SLOT)) return;
    if (a.trace) { const unsigned long long t = pipe_clock(); t_wait += t - t_mark; t_mark = t; }
    L.at_chunk((int)c);
#if defined(ZPQ_EMU) && defined(ZPQ_PERSIST_DEBUG)
    if (lane == 0) fprintf(stderr, "[tick %llu] slot %d kind %d role %d unit %d chunk %u nb %u g %u\n", emu::ticks(), SLOT, kind, role, unit, c, L.nb, g);
#endif
    if constexpr (kind == 0) {
      pipe_hcomp_unit<Chain>(L, (unsigned*)priv, lane, c == 0, false);
    } else if constexpr (kind == 1) {
      if constexpr (pipe_row_in_halves<Chain>(role)) { if (pipe_any(L.nb > 0)) pipe_row_halves<Chain, Chain::ROW_COMP[role]>(L, ro.ns, (unsigned)lane >> 5); }
      else if constexpr (PipeRowRing<Chain>::value) { if (pipe_any(L.nb > 0)) pipe_row_ring<Chain, Chain::ROW_COMP[role]>(L, ro.ns); }
      else if (pipe_any(L.nb > 0)) pipe_row<Chain, Chain::ROW_COMP[role]>(L, ro.ns);
    } else if constexpr (kind == 3) {
      if constexpr (PipeIcmFull<Chain>::value) {
        short* const st = (short*)(priv + 256u * G * 4u);          // the whole stretch table behind the side table
        if (c == 0) {
          for (int i = lane; i < 16384; i += 64) ((unsigned*)st)[i] = ((const unsigned*)a.tb->stretch)[i];
          (void)pipe_any(true);    // (a table every lane reads: the lanes meet here)
        }
        if (pipe_any(L.nb > 0)) pipe_icm_unit<Chain, Chain::ICM_COMP[role], PipeStretchFull, PipeAhead<Chain>::value>(L, (unsigned*)priv, PipeStretchFull{st}, lane, c == 0, false);
      } else {
        if (pipe_any(L.nb > 0)) pipe_icm_unit<Chain, Chain::ICM_COMP[role], PipeStretch, PipeAhead<Chain>::value>(L, (unsigned*)priv, ro.stretch, lane, c == 0, false);
      }
    } else if constexpr (kind == 4) {
      if constexpr (PipeSmallChain<Chain>::value) {
        // a chain of a few units (PS_SMALL): LDS to spare, so the weight pairs stay as two words and squash is the whole table
        unsigned short* const sq = (unsigned short*)(priv + 512u * G * 4u);
        if (c == 0) {
          for (int i = lane; i < 4096; i += 64) sq[i] = a.tb->squash[i];
          (void)pipe_any(true);    // (a table every lane reads: the lanes meet here)
        }
        if (pipe_any(L.nb > 0)) pipe_isse_unit<Chain, Chain::ISSE_COMP[role], PipeSquashFull, PipeAhead<Chain>::value>(L, (unsigned*)priv, PipeSquashFull{sq}, lane, c == 0, false);
      } else {
        if (pipe_any(L.nb > 0)) pipe_isse_packed_unit<Chain, Chain::ISSE_COMP[role], PipeSquash, PipeAhead<Chain>::value>(L, (unsigned*)priv, ro.squash, lane, c == 0);
      }
    } else if constexpr (kind == 5) {
      if (pipe_any(L.nb > 0)) {
        if constexpr (Chain::MIX_BITS != 0) pipe_mix_bits_unit<Chain, role>(L, q, B, ro.squash);
        else if constexpr (PipeMixPacked<Chain>::of(role)) {
          constexpr int QL = Chain::MIX_QL[role], NH = Chain::PS_MIX_NH, BPW = 64 / (QL * NH) < (int)G ? 64 / (QL * NH) : (int)G;
          pipe_mix_packed_unit<Chain, role, PipeMixLdsRows<Chain>::of(role), BPW, NH>(L, q, B, mix_bl < (unsigned)BPW ? mix_bl : 0u, (unsigned*)priv, c == 0, ro.squash);
        }
        else pipe_mix_unit<Chain, role, Chain::PS_MIX_NH>(L, q, B, ro.squash);
      }
    } else {
      constexpr int lk = Chain::LIGHT_KIND[role], I = Chain::LIGHT_COMP[role];
      if constexpr (lk == PK_CODER) {
        if constexpr (PipeCoderFast<Chain>::value) pipe_coder_fast<Chain, (PipeAhead<Chain>::value > 0 ? PipeAhead<Chain>::value : 1)>(L, a, (unsigned*)priv, lane, c == 0);
        else pipe_coder<Chain>(L, a, ro.squash);
      } else if (pipe_any(L.nb > 0)) {
        if constexpr (lk == PK_CONS) pipe_cons<Chain, I>(L);
        else if constexpr (lk == PK_CM) {
          if constexpr (pipe_light_lds_words<Chain>(lk, I) != 0u) pipe_cm_lds<Chain, I>(L, (unsigned*)priv, ro.stretch, ro.dt, lane, c == 0);
          else pipe_cm<Chain, I>(L, ro.stretch, ro.dt);
        }
        else if constexpr (lk == PK_MATCH) pipe_match_any<Chain, I>(L, ro.stretch, ro.dt2k);
        else if constexpr (lk == PK_AVG) pipe_avg<Chain, I>(L);
        else if constexpr (lk == PK_MIX2) {
          if constexpr (pipe_light_lds_words<Chain>(lk, I) != 0u) pipe_mix2_lds<Chain, I>(L, (unsigned*)priv, ro.squash, lane, c == 0);
          else pipe_mix2<Chain, I>(L, ro.squash);
        }
        else if constexpr (lk == PK_SSE) pipe_sse<Chain, I>(L, ro.stretch, ro.dt);
        else if constexpr (lk == PK_CM_BITS) pipe_cm_bits<Chain, I>(L, B, ro.stretch, ro.dt);
        else if constexpr (lk == PK_MIX2_BITS) pipe_mix2_bits<Chain, I>(L, B, ro.squash);
        else if constexpr (lk == PK_SSE_BITS) pipe_sse_bits<Chain, I>(L, B, ro.stretch, ro.dt);
      }
    }
    pipe_reconverge();
    pipe_publish<Chain>(prog, unit, lane);
    if (a.trace) { const unsigned long long t = pipe_clock(); t_work += t - t_mark; t_mark = t; }
  }
  if (a.trace && lane == 0) {
    unsigned long long* rec = a.trace + 4ull * ((unsigned long long)g * (unsigned)Chain::PS_NSLOT + (unsigned)SLOT);
    rec[0] = t_wait; rec[1] = t_work; rec[2] = nchunks; rec[3] = ((unsigned long long)kind << 32) | ((unsigned long long)(unsigned)role << 16) | (unsigned)unit;
  }
}

template <class Chain>
__device__ __forceinline__ void pipe_persist_body(const PipeArgs& a) {
  unsigned char* const lds = ZPQ_PERSIST_LDS(Chain::PS_LDS_BYTES);
  PipeRO& ro = *(PipeRO*)lds;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 1024; i += (int)blockDim.x) ro.dt[i] = a.tb->dt[i];
  for (int i = tid; i < 256; i += (int)blockDim.x) ro.dt2k[i] = (unsigned short)a.tb->dt2k[i];
  for (int i = tid; i < 256; i += (int)blockDim.x) ((unsigned*)ro.ns)[i] = ((const unsigned*)a.tb->ns)[i];
  ro.squash.load(a.tb, tid);
  ro.stretch.load(a.tb, tid);
  if (tid == 0 && a.arrive_need) pipe_prog_add(a.ctl + 3);
  __syncthreads();
  // workgroup -> (group, flavour).  The dispatcher deals the workgroups of a launch round-robin over the XCDs (from where the
  // launch before left off: which XCD a residue class lands on is not known, that a class shares one is): with spread = 8 the
  // workgroups of a group are one class, for as many groups as make whole sets of 8; the remaining groups' workgroups follow
  // one after the other, so that every XCD gets the same number of workgroups of a launch whatever its group count (a second
  // run beside this one finds the same room everywhere) and the grid has no idle workgroups.
  const unsigned b = blockIdx.x, nx = a.spread ? a.spread : 1u, W = (unsigned)Chain::PS_WPG;
  const unsigned whole = a.ngroups_here - a.ngroups_here % nx;
  unsigned g, flavour;
  if (b < whole * W) { const unsigned j = b / nx; g = b % nx + nx * (j / W); flavour = j % W; }
  else { const unsigned r = b - whole * W; g = whole + r / W; flavour = r % W; }
  g += a.group0;
  if (g >= a.group0 + a.ngroups_here) return;
  const int slot = (int)flavour * Chain::PS_WAVES + wave;
  if (!pipe_arrived(a, lane)) return;
  static_for<0, Chain::PS_NSLOT>([&](auto sc) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value;
    if constexpr (Chain::PS_KIND[S] >= 0) {
      if (slot == S) pipe_persist_unit<Chain, Chain::PS_KIND[S], Chain::PS_ROLE[S]>(a, g, lane, S);
    }
  });
}

}  // namespace zpq
