// Specialised wave-parallel coder: the kernel template that a GENERATED
// translation unit instantiates for ONE block header (the GPU analogue of the
// reference's per-header x86 JIT, libzpaq.cpp:3824-4583 / 3231-3811).
//
// The generator (host/codegen.cpp) emits a `Chain` type holding the COMP list as
// constexpr data plus the HCOMP program translated to straight-line HIP C++;
// everything below is ordinary hand-written HIP for gfx950 that the compiler
// folds against those constants.
//
// Mapping: one ZPAQ block per wavefront, component i on lane i (n <= 64),
// Chain::WAVES (4 or 8) blocks per workgroup.  The ideas, in the order they
// appear below (DESIGN.md section 4.1 has the measurements behind each):
//   * small hot tables in LDS: half of stretch (mirrored), the non-trivial part
//     of squash, dt, the state table, and -- per block -- HCOMP's H array and as
//     many ICM/ISSE bit-history->probability side tables as fit;
//   * the 16-byte bit-history row of every ICM/ISSE is cached in 4 VGPRs for the
//     4 bits of a nibble: one probe (3 x 16 B in one 64-B line) + one 16-B
//     write-back per nibble instead of a byte load/store per bit;
//   * every global word of a bit (CM word, MIX2 weight, MIX rows, SSE row) has an
//     address that depends only on (h, c8, hmap4): both candidates of the NEXT
//     bit are fetched during this one; what predict read stays in registers for
//     update (no re-load);
//   * the 8 bits of a byte are unrolled with the bit position as a compile-time
//     tag, so position-dependent control flow disappears;
//   * lane-class selects are bit blends with per-lane constant masks in VGPRs;
//     lanes without a table of some kind work on one shared dummy line, so no
//     per-bit code needs the exec mask;
//   * the dependent chain (ISSE/AVG/MIX2/MIX/SSE) is unrolled at compile time
//     with literal lane numbers (v_readlane / DPP), no descriptor fetches;
//   * HCOMP runs as compiled code, not through an interpreter; the arithmetic
//     coder runs on the scalar unit.
// Integer arithmetic is bit-exact with Predictor::predict0/update0.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "layout.h"

// Compiler fences for register values (no instructions): KEEP = "this value is used" (keeps a prefetch
// load alive), OPAQUE = "forget what you know about this value".  The host-side wavefront emulator
// (tests/emu, test infrastructure) defines ZPQ_EMU and needs neither.
#ifndef ZPQ_EMU
#define ZPQ_KEEP2(a, b) asm volatile("" ::"v"(a), "v"(b))
#define ZPQ_KEEP3(a, b, c) asm volatile("" ::"v"(a), "v"(b), "v"(c))
#define ZPQ_KEEP4(a, b, c, d) asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d))
#define ZPQ_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define ZPQ_KEEP2(a, b) ((void)(a), (void)(b))
#define ZPQ_KEEP3(a, b, c) ((void)(a), (void)(b), (void)(c))
#define ZPQ_KEEP4(a, b, c, d) ((void)(a), (void)(b), (void)(c), (void)(d))
#define ZPQ_OPAQUE(x) ((void)(x))
#endif

namespace zpq {

// ---- compile-time description of one component (emitted by the generator) ----
struct CompK {
  unsigned type, a1, a2, a3, a4, a5;
  unsigned limit, mask0, mask1;
  unsigned long long t0, t1;   // arena byte offsets
  int lds;                     // byte offset of the side table in the wave's LDS region, or -1
  int slot;                    // ordinal among MIX (resp. SSE) components, else -1
  unsigned stride = 0;         // MIX: words between weight rows (layout.h mix_row_stride)
};

struct SpecTables {                            // shared by the waves of a workgroup
  int16_t stretch_hi[16384];                   // stretch(x) for x >= 16384; mirrored below
  uint16_t squash_mid[1344];                   // squash index 1376..2719
  int32_t dt[1024];
  uint16_t dt2k[256];
  uint8_t ns[1024];
};
static_assert(sizeof(SpecTables) == kSpecTablesBytes, "host codegen and device disagree on LDS tables");

__device__ __forceinline__ int sp_stretch(const SpecTables& T, unsigned x) {   // x in 0..32767
  const bool hi = x >= 16384u;
  const int v = T.stretch_hi[hi ? x - 16384u : 16383u - x];
  return hi ? v : -v;
}
__device__ __forceinline__ int sp_squash(const SpecTables& T, int p) {         // p in -2048..2047
  const int i = p + 2048 - 1376;
  const int v = T.squash_mid[min(max(i, 0), 1343)];
  return i < 0 ? 0 : (i > 1343 ? 32767 : v);
}
__device__ __forceinline__ int sp_clamp2k(int x) { return min(max(x, -2048), 2047); }
__device__ __forceinline__ int sp_clamp512k(int x) { return min(max(x, -(1 << 19)), (1 << 19) - 1); }

__device__ __forceinline__ int sp_rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ unsigned sp_rlu(unsigned v, int lane) { return (unsigned)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ unsigned sp_uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
// HCOMP's condition flag: wave-uniform by construction; telling the compiler so keeps the VM's
// branches scalar.  (The emulator runs HCOMP on one lane only -- no cross-lane traffic there.)
__device__ __forceinline__ unsigned vm_flag(bool c) {
#if !defined(ZPQ_EMU) && !defined(ZPQ_LANE_VM)   // ZPQ_LANE_VM: one HCOMP machine per lane (pipe_kernel.h), the flag is per lane
  return sp_uni(c ? 1u : 0u);
#else
  return c ? 1u : 0u;
#endif
}
__device__ __forceinline__ unsigned long long sp_uni64(unsigned long long v) {
  return (unsigned long long)sp_uni((unsigned)(v >> 32)) << 32 | sp_uni((unsigned)v);
}

// sum of lanes 0..LANES-1 (the other lanes hold 0); result wave-uniform.  Row scans by DPP, then one
// readlane per 16-lane row that can hold an input.
template <int LANES>
__device__ __forceinline__ int sp_lanes_sum(int x) {
  int v = x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
  int s = sp_rl(v, 15);
  if constexpr (LANES > 16) s += sp_rl(v, 31);
  if constexpr (LANES > 32) s += sp_rl(v, 47);
  if constexpr (LANES > 48) s += sp_rl(v, 63);
  return s;
}

// LDS-qualified views: loads/stores through these are ds_read/ds_write, never flat.
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) unsigned lds_u32;

// Explicit global-memory views of the arena: pointers loaded from the job
// descriptor are generic to the compiler, and flat_* accesses would tie vmcnt
// and lgkmcnt together (every LDS wait would also wait for HBM).
typedef __attribute__((address_space(1))) unsigned char g_u8;
typedef __attribute__((address_space(1))) unsigned g_u32;
typedef __attribute__((address_space(1))) int g_i32;
typedef __attribute__((address_space(1))) uint4 g_u128;

// byte `idx` (0..15, wave-uniform) of a 16-byte row held in 4 registers
__device__ __forceinline__ unsigned row_get(unsigned r0, unsigned r1, unsigned r2, unsigned r3, int idx) {
  const int sel = idx >> 2;
  const unsigned w = sel == 0 ? r0 : (sel == 1 ? r1 : (sel == 2 ? r2 : r3));
  return (w >> ((idx & 3) * 8)) & 255u;
}
__device__ __forceinline__ void row_set(unsigned& r0, unsigned& r1, unsigned& r2, unsigned& r3, int idx, unsigned v) {
  const int sel = idx >> 2, sh = (idx & 3) * 8;
  const unsigned m = ~(255u << sh), b = v << sh;
  r0 = sel == 0 ? ((r0 & m) | b) : r0;
  r1 = sel == 1 ? ((r1 & m) | b) : r1;
  r2 = sel == 2 ? ((r2 & m) | b) : r2;
  r3 = sel == 3 ? ((r3 & m) | b) : r3;
}

// The same with the bit's position in its nibble known (NB = 0..3): the slot index is 1, 2..3, 4..7 or
// 8..15, so the register holding it is known except for the last position (NB < 0: unknown, generic).
template <int NB>
__device__ __forceinline__ unsigned row_get_nb(unsigned r0, unsigned r1, unsigned r2, unsigned r3, int idx) {
  if constexpr (NB < 0) return row_get(r0, r1, r2, r3, idx);
  const unsigned w = NB <= 1 ? r0 : (NB == 2 ? r1 : ((idx & 4) ? r3 : r2));
  return (w >> ((idx & 3) * 8)) & 255u;
}
template <int NB>
__device__ __forceinline__ void row_set_nb(unsigned& r0, unsigned& r1, unsigned& r2, unsigned& r3, int idx, unsigned v) {
  if constexpr (NB < 0) { row_set(r0, r1, r2, r3, idx, v); return; }
  const int sh = (idx & 3) * 8;
  const unsigned m = ~(255u << sh), b = v << sh;
  if constexpr (NB <= 1) r0 = (r0 & m) | b;
  else if constexpr (NB == 2) r1 = (r1 & m) | b;
  else {
    r2 = (idx & 4) ? r2 : ((r2 & m) | b);
    r3 = (idx & 4) ? ((r3 & m) | b) : r3;
  }
}

template <class Chain, int I>
struct Dep;   // forward

// Per-lane constants of the dependent components, as VGPR masks / offsets (see lane_mask in the kernel
// body): on gfx950 a v_cmp that writes an SGPR mask needs two wait states before a v_cndmask may read
// it, so "lane == I ? a : b" costs three issue slots; a blend with a ready mask costs one.
template <int N, int NM, int NS>
struct LaneK {
  unsigned is[N];       // all ones in lane i
  unsigned mixin[NM];   // all ones in the lanes that hold an input / weight of MIX slot k (lane < m)
  unsigned mixst[NM];   // where the lane stores its weight of the selected row: table base + 4 lane, or the dummy
  unsigned ssest[NS];   // lane 0: SSE table base, other lanes: the dummy
  unsigned lane0;       // all ones in lane 0
};
__device__ __forceinline__ unsigned sp_blend(unsigned m, unsigned a, unsigned b) { return (a & m) | (b & ~m); }

// compile-time loop: f(IC<B>{}), f(IC<B+1>{}), ...
template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}

// May a MIX / SSE row be fetched one bit early?  Only when consecutive bits of a byte always
// select different rows (then the early copy cannot miss the previous bit's update).
constexpr bool mix_pf(const CompK& c) { return c.a5 == 255u && c.mask0 >= 255u; }
constexpr bool sse_pf(const CompK& c) { return c.mask0 >= 32u * 256u - 1u; }

// 24-bit multiply-add (v_mad_i32_i24): exact whenever both factors fit in 24 signed bits, and the
// low 32 bits of the product wrap exactly like a 32-bit multiply
__device__ __forceinline__ int sp_mad24(int a, int b, int c) { return __mul24(a, b) + c; }

// value of the lane to the left (lane 0 receives 0): DPP wave_shr:1
__device__ __forceinline__ int sp_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }

// ISSE fast path: every ISSE takes its input from the lane on its left, and that lane is a
// context-only component or another ISSE.  Then all ISSE chains of the model are resolved together
// by DEPTH rounds of (shift right one lane, multiply-add, clamp) instead of one round per component.
template <class Chain>
constexpr bool isse_left_fed() {
  for (int i = 0; i < Chain::N; ++i) {
    if (Chain::comp[i].type != C_ISSE) continue;
    if (i == 0 || Chain::comp[i].a2 != (unsigned)(i - 1)) return false;
    const unsigned t = Chain::comp[i - 1].type;
    if (!(t == C_CONS || t == C_CM || t == C_ICM || t == C_MATCH || t == C_ISSE)) return false;
  }
  return true;
}
template <class Chain>
constexpr int isse_depth() {
  int best = 0, run = 0;
  for (int i = 0; i < Chain::N; ++i) {
    run = Chain::comp[i].type == C_ISSE ? run + 1 : 0;
    if (run > best) best = run;
  }
  return best;
}

// lanes (components) of a given type, as a compile-time bit mask
template <class Chain>
constexpr unsigned long long type_mask(unsigned t) {
  unsigned long long m = 0;
  for (int i = 0; i < Chain::N; ++i)
    if (Chain::comp[i].type == t) m |= 1ull << i;
  return m;
}

// ---------------------------------------------------------------------------
template <class Chain, bool DEC>
__device__ __forceinline__ void spec_kernel_body(const BlockJob* jobs, BlockResult* res, unsigned nblocks,
                                                 const DeviceTables* tb) {
  constexpr int N = Chain::N;
  static_assert(N >= 1 && N <= 64, "specialised kernel handles 1..64 components");
  constexpr int NMIX = Chain::NMIX > 0 ? Chain::NMIX : 1;
  constexpr int NSSE = Chain::NSSE > 0 ? Chain::NSSE : 1;
  constexpr int kSpecWaves = Chain::WAVES;                 // blocks per workgroup, chosen by the generator
  constexpr int kSpecWaveLds = spec_wave_lds_bytes(kSpecWaves);
  static_assert((int)sizeof(SpecTables) + kSpecWaves * kSpecWaveLds <= kSpecLdsBudget, "LDS budget");

  __shared__ SpecTables T;
  __shared__ __attribute__((aligned(16))) unsigned char wave_lds[kSpecWaves][kSpecWaveLds];
  for (unsigned i = threadIdx.x; i < 16384u; i += blockDim.x) T.stretch_hi[i] = tb->stretch[16384u + i];
  for (unsigned i = threadIdx.x; i < 1344u; i += blockDim.x) T.squash_mid[i] = tb->squash[1376u + i];
  for (unsigned i = threadIdx.x; i < 1024u; i += blockDim.x) { T.dt[i] = tb->dt[i]; T.ns[i] = tb->ns[i]; }
  for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) T.dt2k[i] = (uint16_t)tb->dt2k[i];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const unsigned b = blockIdx.x * kSpecWaves + wave;
  const bool live = b < nblocks;
  BlockJob job = jobs[live ? b : 0];
  g_u8* const arena = (g_u8*)sp_uni64((unsigned long long)job.arena);
  const g_u8* const in_ptr = (const g_u8*)sp_uni64((unsigned long long)job.in);
  g_u8* const out_ptr = (g_u8*)sp_uni64((unsigned long long)job.out);
  job.in_len = sp_uni(job.in_len);
  job.out_cap = sp_uni(job.out_cap);
  const unsigned rslot = sp_uni(job.res_slot);
  lds_u8* const wl = (lds_u8*)&wave_lds[wave][0];

  // ---- per-lane component constants (selected from the constexpr chain) ----
  // Tables are addressed as arena + 32-bit offset (the arena of a specialised plan is < 4 GiB),
  // so every global access is "uniform base + VGPR offset" and needs no 64-bit address math.
  // Lanes without a table of some kind point at a 64-byte dummy line instead: loads and stores can
  // then be issued by ALL lanes with no exec-mask juggling (divergent branches were a quarter of the
  // instruction stream), and land harmlessly.  The line is shared (stride 0): the idle lanes of a
  // memory instruction coalesce into one transaction; with a private line per lane every instruction
  // touched 64 lines, which cost 6 % with one wavefront per SIMD and all of the gain of two.
#ifndef ZPQ_DUMMY_STRIDE
#define ZPQ_DUMMY_STRIDE 0
#endif
  const unsigned dummy = (unsigned)Chain::OFF_RUN + (unsigned)lane * (unsigned)ZPQ_DUMMY_STRIDE;
  const unsigned dummy_lds = (unsigned)(kSpecWaveLds - 512) + (unsigned)lane * 8u;
  unsigned a2 = 0, a4 = 0, a5 = 0, limit = 0, mask0 = 0, mask1 = 63, sizebits = 0;
  unsigned off0 = dummy, off1 = dummy;
  int ldsoff = -1;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (lane == i) {
      const CompK c = Chain::comp[i];
      a2 = c.a2; a4 = c.a4; a5 = c.a5;
      limit = c.limit; mask0 = c.mask0; sizebits = c.a1 + 2;
      off0 = (unsigned)c.t0;
      if (c.type == C_ICM || c.type == C_ISSE || c.type == C_MATCH) { off1 = (unsigned)c.t1; mask1 = c.mask1; }
      ldsoff = c.lds;
    }
  }
  auto G32 = [&](unsigned off) __attribute__((always_inline)) -> g_u32& { return *(g_u32*)(arena + off); };
  auto G8 = [&](unsigned off) __attribute__((always_inline)) -> g_u8& { return *(g_u8*)(arena + off); };
  auto G128 = [&](unsigned off) __attribute__((always_inline)) -> g_u128& { return *(g_u128*)(arena + off); };
  auto L32 = [&](unsigned off) __attribute__((always_inline)) -> lds_u32& { return *(lds_u32*)(wl + off); };

  // side tables -> LDS (the arena copies were initialised by init_arena_kernel)
  if (live) {
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr CompK c = Chain::comp[decltype(ic)::value];
      if constexpr (c.lds >= 0 && (c.type == C_ICM || c.type == C_ISSE)) {
        constexpr int words = c.type == C_ICM ? 256 : 512;
        const g_u32* src = (const g_u32*)(arena + c.t0);
        lds_u32* dst = (lds_u32*)(wl + c.lds);
        for (int k = lane; k < words; k += 64) dst[k] = src[k];
      }
    });
    if constexpr (Chain::H_LDS >= 0)
      for (unsigned k = lane; k <= Chain::HMASK; k += 64) ((lds_u32*)(wl + Chain::H_LDS))[k] = 0;
    L32(dummy_lds) = 0;
    L32(dummy_lds + 4) = 0;
  }
  __syncthreads();
  if (!live) return;

  // HCOMP machine registers (A is the per-call input); M and R in the arena, H in LDS when it fits
  unsigned vm_b = 0, vm_c = 0, vm_d = 0, vm_f = 0;
  g_u8* const vm_M = arena + Chain::OFF_M;
  g_u32* const vm_R = (g_u32*)(arena + Chain::OFF_R);
  auto vm_H = [&]() {
    if constexpr (Chain::H_LDS >= 0) return (lds_u32*)(wl + Chain::H_LDS);
    else return (g_u32*)(arena + Chain::OFF_H);
  }();

  constexpr bool kIsseFast = isse_left_fed<Chain>();
  constexpr int kIsseDepth = isse_depth<Chain>();
  // ---- lane classes (compile-time masks over the chain) ----
  constexpr unsigned long long M_CM = type_mask<Chain>(C_CM), M_ICM = type_mask<Chain>(C_ICM),
                               M_ISSE = type_mask<Chain>(C_ISSE), M_MATCH = type_mask<Chain>(C_MATCH),
                               M_MIX2 = type_mask<Chain>(C_MIX2);
  const bool is_cm = (M_CM >> lane) & 1, is_icm = (M_ICM >> lane) & 1, is_isse = (M_ISSE >> lane) & 1;
  const bool is_match = (M_MATCH >> lane) & 1, is_mix2 = (M_MIX2 >> lane) & 1;
  const bool has_row = is_icm || is_isse;
  const bool is_ctx = is_cm || is_icm || is_match;   // prediction = one stretch lookup
  const unsigned ctx_shift = is_icm ? 8u : 17u;
  const bool gl = is_cm || is_mix2;           // lanes with one global table word per bit
  // may the next bit's word be fetched one bit early?  Only if consecutive bits can never
  // address the same element (else the early copy could miss this bit's update)
  const bool pf_lane = is_cm ? mask0 >= 511u : (is_mix2 && a5 == 255u && mask0 >= 255u);
  // a table with a single element never leaves its register (e.g. the final "mix2 0")
  const bool resident = gl && mask0 == 0u;
  const unsigned goff = gl ? off0 : dummy;                  // base of the per-bit global word
  const unsigned gmask = gl ? mask0 : 0u;                   // idle lanes always address element 0 of their dummy
  const unsigned rmask = has_row ? mask1 : 63u;             // idle lanes probe inside their 64-byte dummy
  const unsigned roff = has_row ? off1 : dummy;             // base of the bit-history hash table
  const unsigned ldsq = (has_row && ldsoff >= 0) ? (unsigned)ldsoff : dummy_lds;   // side table in LDS
  const bool side_global = has_row && ldsoff < 0;           // side table left in the arena (LDS full)
  const unsigned soff = side_global ? off0 : dummy;         // base of a side table that stayed in the arena
  // Lane-class selects in the per-bit code are bit blends with per-lane constant masks held in
  // VGPRs (one v_bfi_b32 / v_bitop3_b32 each), not v_cndmask on loop-invariant SGPR-pair predicates: the bit loop
  // had run out of SGPRs and was spilling those predicates to VGPR lanes.  The masks are opaque to
  // the optimiser, which otherwise rewrites the lane classes as range tests on the lane id.
  auto lane_mask = [&](bool b) __attribute__((always_inline)) -> unsigned {
    unsigned m = b ? 0xFFFFFFFFu : 0u;
    ZPQ_OPAQUE(m);
    return m;
  };
  const unsigned m_cm = lane_mask(is_cm), m_isse = lane_mask(is_isse), m_icm = lane_mask(is_icm);
  const unsigned m_match = lane_mask(is_match), m_row = lane_mask(has_row), m_ctx = lane_mask(is_ctx);
  const unsigned m_res = lane_mask(resident), m_pf = lane_mask(pf_lane);
  const unsigned m_lds2 = lane_mask(is_isse && !side_global);     // lanes with a second side-table word in LDS
  const unsigned bh_shift = is_isse ? 1u : 0u;                    // ISSE entries are two words wide
  const unsigned q1off = is_icm ? 0u : 4u;
  const unsigned n1base = (is_isse && !side_global) ? ldsq + 4u : dummy_lds + 4u;

  // ---- per-lane mutable state ----
  unsigned bh = 0;             // ICM/ISSE: bit history of this bit (Component::cxt)
  unsigned gidx = 0;           // CM/MIX2: element index of this bit
  unsigned h = 0;
  int p = 0;
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (Chain::comp[i].type == C_CONS) { if (lane == i) p = ((int)Chain::comp[i].a1 - 128) * 4; }
  });
  // table words read in predict and reused by update:
  //   CM: v0 = cm word | ICM: v0 = side-table word | ISSE: v0,v1 = weights | MIX2: v0 = weight
  unsigned v0 = 0, v1 = 0;
  // cached 16-byte bit-history row.  The tables start out all zero, so writing the initial
  // (zero) row back to offset 0 at the first probe changes nothing.
  unsigned row0 = 0, row1 = 0, row2 = 0, row3 = 0;
  unsigned rowoff = 0;
  unsigned touch_a = 0, touch_b = 0;   // keep-alive for the second-nibble line prefetch
  // MATCH, register resident (Component a=len, b=offset, limit=pos): the byte being built is c8
  // itself and the predicted byte is fetched once per byte
  unsigned ra = 0, rb = 0, rc = 0, rlimit = 0, mpred = 0, mdd = 0;
  // rows selected this bit (wave-uniform element indices) and their words, lane-parallel
  int mixw[NMIX];              // lane t holds weight t of each MIX row
  unsigned mixrow[NMIX];
  unsigned ssev[NSSE];         // lane t (<32) holds entry t of each SSE row
  unsigned ssecx[NSSE];        // element index of the trained SSE entry
  // one-bit-ahead candidates (next bit = 0 / next bit = 1)
  unsigned gwc0 = 0, gwc1 = 0;
  int mixc0[NMIX], mixc1[NMIX];
  unsigned ssec0[NSSE], ssec1[NSSE];
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { mixw[k] = 0; mixrow[k] = 0; mixc0[k] = 0; mixc1[k] = 0; }
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { ssev[k] = 0; ssecx[k] = 0; ssec0[k] = 0; ssec1[k] = 0; }
  // side tables in the arena: the entry of the NEXT bit is one of two known a bit ahead (both bit
  // histories sit in the cached row), so both are fetched early; the entry trained by the previous
  // update is forwarded from registers when the next bit selects it again.  Only with one wavefront
  // per SIMD: with two, the other wavefront already covers the latency and the extra loads cost
  // more than they save (2048 x 64 KiB -m5: 1536 ms with, 1420 ms without).
  constexpr bool kSidePf = Chain::WAVES <= 4;
  unsigned sca0 = 0, sca1 = 0, scb0 = 0, scb1 = 0;
  unsigned le0 = 0xFFFFFFFFu, ln0 = 0, ln1 = 0;
  // MIX / SSE rows: lane t reads word t of the selected row.  The lane's part of the address (table base +
  // 4 t) is loop invariant and kept in a register the compiler may not look through: otherwise it folds the
  // table base into 64-bit pointer arithmetic and spends four instructions per load instead of one add.
  unsigned mixbase[NMIX], ssebase[NSSE];
#pragma unroll
  for (int k = 0; k < NMIX; ++k) mixbase[k] = dummy;
#pragma unroll
  for (int k = 0; k < NSSE; ++k) ssebase[k] = dummy;
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr CompK c = Chain::comp[decltype(ic)::value];
    if constexpr (c.type == C_MIX) {
      mixbase[c.slot] = (unsigned)c.t0 + 4u * (unsigned)min(lane, (int)c.a3 - 1);
      ZPQ_OPAQUE(mixbase[c.slot]);
    } else if constexpr (c.type == C_SSE) {
      ssebase[c.slot] = (unsigned)c.t0 + 4u * (unsigned)(lane & 31);
      ZPQ_OPAQUE(ssebase[c.slot]);
    }
  });
  LaneK<N, NMIX, NSSE> lk;
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { lk.mixin[k] = 0; lk.mixst[k] = dummy; }
#pragma unroll
  for (int k = 0; k < NSSE; ++k) lk.ssest[k] = dummy;
  lk.lane0 = lane_mask(lane == 0);
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr CompK c = Chain::comp[i];
    lk.is[i] = 0;
    if constexpr (c.type == C_AVG || c.type == C_MIX2 || c.type == C_MIX || c.type == C_SSE || c.type == C_ISSE)
      lk.is[i] = lane_mask(lane == i);
    if constexpr (c.type == C_MIX) {
      lk.mixin[c.slot] = lane_mask(lane < (int)c.a3);
      lk.mixst[c.slot] = lane < (int)c.a3 ? (unsigned)c.t0 + 4u * (unsigned)lane : dummy;
      ZPQ_OPAQUE(lk.mixst[c.slot]);
    } else if constexpr (c.type == C_SSE) {
      lk.ssest[c.slot] = lane == 0 ? (unsigned)c.t0 : dummy;
      ZPQ_OPAQUE(lk.ssest[c.slot]);
    }
  });
  unsigned rw = G32(goff);     // the element of a resident (single-entry) table
  // LDS lookups that update needs but whose index is known during predict are issued there, where
  // their latency hides behind the dependent chain: both successors of the bit history, the
  // adaptation rate of the CM word, and squash(p) of every lane (lane N-1's is the coder's probability)
  unsigned nspair = 0, dtv = 0;
  int sq = 0;
  unsigned ssetr[NSSE], ssedt[NSSE];   // SSE: the entry that will be trained and its adaptation rate
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { ssetr[k] = 0; ssedt[k] = 0; }
  // wave-uniform copies of the contexts of the MIX / SSE components (they select rows; constant for a byte)
  unsigned hmix[NMIX], hsse[NSSE], hmix_n[NMIX], hsse_n[NSSE];
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { hmix[k] = 0; hmix_n[k] = 0; }
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { hsse[k] = 0; hsse_n[k] = 0; }
  int pdv[N];                  // MIX2: p[j] - p[k] of this bit (wave-uniform), read in predict, reused by update
#pragma unroll
  for (int k = 0; k < N; ++k) pdv[k] = 0;
  bool pf_valid = false;       // candidates fetched during the previous bit are usable (uniform)
  int ylast = 0;

  int c8 = 1, hmap4 = 1;
  unsigned low = 1, high = 0xFFFFFFFFu;
  unsigned steps = 0;
  int status = 0;

#ifdef ZPQ_PROF
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // predict, update, hcomp, total, rows, loads, chain
  const unsigned long long prof_t0 = __builtin_readcyclecounter();
#define SP_PROF_BEGIN const unsigned long long pt_ = __builtin_readcyclecounter();
#define SP_PROF_END(k) prof[k] += __builtin_readcyclecounter() - pt_;
#else
#define SP_PROF_BEGIN
#define SP_PROF_END(k)
#endif

  // element index of the per-bit global word (CM: h ^ hmap4, MIX2: h + (c8 & mask)); idle lanes get 0
  auto g_index = [&](int c8x, int hm4x) __attribute__((always_inline)) -> unsigned {
    return (is_cm ? (h ^ (unsigned)hm4x) : (h + (unsigned)(c8x & (int)a5))) & gmask;
  };

  // ---------------------------------------------------------------- predict
  // predict / update / after_bit take the bit's position in the byte (0 = first, most significant) as a
  // compile-time tag: the byte loop is unrolled, and everything that depends only on the position --
  // nibble starts, early-fetch validity, the hmap4 update rule, byte completion -- is decided at compile
  // time instead of by a ladder of scalar compares and branches per bit.  Tag -1 = position unknown
  // (decide at run time; ZPQ_UNROLL_BITS=0).
  auto predict = [&](auto bitc) __attribute__((always_inline)) -> unsigned {
    constexpr int B = decltype(bitc)::value;
    const bool nib = B >= 0 ? (B == 0 || B == 4) : ((c8 == 1) || ((c8 & 0xf0) == 16));
    const int slot = hmap4 & 15;
    const bool more = B >= 0 ? B < 7 : c8 < 128;               // another bit of this byte follows
    const bool pf_now = B >= 0 ? B > 0 : pf_valid;             // candidates were fetched during the previous bit
    const bool last_of_nibble = B >= 0 ? B == 3 : (c8 >= 8 && c8 < 16);
    const int c8a = c8 * 2, c8b = c8 * 2 + 1;
    const int hm4a = last_of_nibble ? ((hmap4 & 0xf) << 5 | 1) : ((hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2) & 0xf));
    const int hm4b = last_of_nibble ? ((hmap4 & 0xf) << 5 | 1 << 4 | 1)
                                    : ((hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + 1) & 0xf));
#ifdef ZPQ_PROF
    const unsigned long long pb0 = __builtin_readcyclecounter();
#endif
    // (B) ICM / ISSE: bit-history row in registers, probed once per nibble (all lanes take part;
    //     lanes without a row work on their dummy slot)
    if (nib) {
      ZPQ_KEEP2(touch_a, touch_b);
      const unsigned cx = h + 16u * (unsigned)c8;
      const unsigned chk = (cx >> sizebits) & 255u;
      const unsigned h0 = (cx * 16u) & (rmask - 15u);
      // The three candidate rows are fetched BEFORE the old row is written back, so that waiting
      // for them does not also wait for that store (vmcnt retires in order); if the old row is one
      // of the candidates, its register copy -- newer than memory -- is forwarded instead.
      uint4 r0 = G128(roff + h0);
      uint4 r1 = G128(roff + (h0 ^ 16u));
      uint4 r2 = G128(roff + (h0 ^ 32u));
      const uint4 oldrow = make_uint4(row0, row1, row2, row3);
      G128(roff + rowoff) = oldrow;
      if (rowoff == h0) r0 = oldrow;
      if (rowoff == (h0 ^ 16u)) r1 = oldrow;
      if (rowoff == (h0 ^ 32u)) r2 = oldrow;
      // Predictor::find (libzpaq.cpp:2072-2088)
      const bool m0 = (r0.x & 255u) == chk, m1 = (r1.x & 255u) == chk, m2 = (r2.x & 255u) == chk;
      const unsigned p0 = (r0.x >> 8) & 255u, p1 = (r1.x >> 8) & 255u, p2 = (r2.x >> 8) & 255u;
      const int victim = (p0 <= p1 && p0 <= p2) ? 0 : (p1 < p2 ? 1 : 2);
      const bool hit = m0 || m1 || m2;
      const int pick = m0 ? 0 : (m1 ? 1 : (m2 ? 2 : victim));
      rowoff = h0 ^ (unsigned)(pick << 4);
      row0 = hit ? (pick == 0 ? r0.x : (pick == 1 ? r1.x : r2.x)) : chk;
      row1 = hit ? (pick == 0 ? r0.y : (pick == 1 ? r1.y : r2.y)) : 0u;
      row2 = hit ? (pick == 0 ? r0.z : (pick == 1 ? r1.z : r2.z)) : 0u;
      row3 = hit ? (pick == 0 ? r0.w : (pick == 1 ? r1.w : r2.w)) : 0u;
    } else if (last_of_nibble) {
      // the second nibble's row will be one of two lines: pull both towards this XCD's L2 now
      const unsigned cxa = h + 16u * (unsigned)c8a, cxb = h + 16u * (unsigned)c8b;
      touch_a = G32(roff + ((cxa * 16u) & (rmask - 15u)));
      touch_b = G32(roff + ((cxb * 16u) & (rmask - 15u)));
    }
    bh = row_get_nb<(B >= 0 ? (B & 3) : -1)>(row0, row1, row2, row3, slot);   // bit history
    nspair = *(const unsigned short*)&T.ns[(bh & 255u) * 4u];            // ns[4*bh] | ns[4*bh+1] << 8
    // side table: ICM one word at [bh]; ISSE two words at [2*bh], [2*bh+1]; idle lanes: their dummy
    const unsigned e0 = (bh << bh_shift) & m_row;
    const unsigned el = side_global ? 0u : e0;             // LDS view: a lane whose table is global uses its dummy
    unsigned q0 = L32(ldsq + 4u * el);
    unsigned q1 = L32(ldsq + 4u * el + q1off);
    if constexpr (Chain::ANY_GLOBAL_SIDE) {
      const unsigned sidx = side_global ? e0 : 0u;
      if (nib || !kSidePf) {                                // new row: nothing was fetched ahead
        const unsigned g0 = G32(soff + 4u * sidx), g1 = G32(soff + 4u * sidx + 4u);
        q0 = side_global ? g0 : q0;
        q1 = side_global ? g1 : q1;
      } else {
        const bool fwd = sidx == le0;
        const unsigned g0 = fwd ? ln0 : (ylast ? scb0 : sca0), g1 = fwd ? ln1 : (ylast ? scb1 : sca1);
        q0 = side_global ? g0 : q0;
        q1 = side_global ? g1 : q1;
      }
      // candidates of the next bit (slots hm4a / hm4b of the same row; this bit's update only touches `slot`)
      if constexpr (kSidePf) {
      const unsigned bha = row_get(row0, row1, row2, row3, hm4a & 15), bhb = row_get(row0, row1, row2, row3, hm4b & 15);
      const unsigned ea = side_global ? (is_icm ? bha : 2u * bha) : 0u, eb = side_global ? (is_icm ? bhb : 2u * bhb) : 0u;
      sca0 = G32(soff + 4u * ea); sca1 = G32(soff + 4u * ea + 4u);
      scb0 = G32(soff + 4u * eb); scb1 = G32(soff + 4u * eb + 4u);
      }
    }
#ifdef ZPQ_PROF
    const unsigned long long pb1 = __builtin_readcyclecounter();
    prof[4] += pb1 - pb0;
#endif
    // (A) global words of this bit.  Addresses depend only on (h, c8, hmap4), so a word can be
    //     fetched one bit early for both values of the coming bit.  This bit's words come from the
    //     candidates fetched during the previous bit (registers, no wait on anything issued in
    //     this step) -- or are loaded now at the start of a byte / where early fetch is illegal.
    //     Then the next bit's candidates are issued, unconditionally and by every lane, so that
    //     the instruction stream -- and with it the compiler's vmcnt bookkeeping -- is the same
    //     on every path.
    unsigned gw;
    gidx = g_index(c8, hmap4);
    if (pf_now) {
      gw = ylast ? gwc1 : gwc0;
      static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
        constexpr CompK c = Chain::comp[decltype(ic)::value];
        if constexpr (c.type == C_MIX && mix_pf(c)) mixw[c.slot] = ylast ? mixc1[c.slot] : mixc0[c.slot];
        if constexpr (c.type == C_SSE && sse_pf(c)) ssev[c.slot] = ylast ? ssec1[c.slot] : ssec0[c.slot];
      });
      if constexpr (Chain::ANY_NONPF_GL) {
        if (gl && !pf_lane && !resident) gw = G32(goff + 4u * gidx);
      }
    } else {
      gw = G32(goff + 4u * gidx);
      static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr CompK c = Chain::comp[i];
        if constexpr (c.type == C_MIX && mix_pf(c)) {
          const unsigned r = ((hmix[c.slot] + (unsigned)(c8 & 255)) & c.mask0) * c.stride;
          mixw[c.slot] = (int)G32(mixbase[c.slot] + 4u * r);
        }
        if constexpr (c.type == C_SSE && sse_pf(c)) {
          const unsigned cx0 = ((hsse[c.slot] + (unsigned)c8) * 32u) & c.mask0;
          ssev[c.slot] = G32(ssebase[c.slot] + 4u * cx0);
        }
      });
    }
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr CompK c = Chain::comp[i];
      if constexpr (c.type == C_MIX) {
        const unsigned hi = hmix[c.slot];
        mixrow[c.slot] = ((hi + (unsigned)(c8 & (int)c.a5)) & c.mask0) * c.stride;
        if constexpr (mix_pf(c)) {
          mixc0[c.slot] = (int)G32(mixbase[c.slot] + 4u * (((hi + (unsigned)(c8a & 255)) & c.mask0) * c.stride));
          mixc1[c.slot] = (int)G32(mixbase[c.slot] + 4u * (((hi + (unsigned)(c8b & 255)) & c.mask0) * c.stride));
        } else {
          mixw[c.slot] = (int)G32(mixbase[c.slot] + 4u * mixrow[c.slot]);
        }
      } else if constexpr (c.type == C_SSE) {
        const unsigned hi = hsse[c.slot];
        ssecx[c.slot] = ((hi + (unsigned)c8) * 32u) & c.mask0;
        if constexpr (sse_pf(c)) {
          ssec0[c.slot] = G32(ssebase[c.slot] + 4u * (((hi + (unsigned)c8a) * 32u) & c.mask0));
          ssec1[c.slot] = G32(ssebase[c.slot] + 4u * (((hi + (unsigned)c8b) * 32u) & c.mask0));
        } else {
          ssev[c.slot] = G32(ssebase[c.slot] + 4u * ssecx[c.slot]);
        }
      }
    });
    {
      const unsigned ia = g_index(c8a, hm4a) & m_pf, ib = g_index(c8b, hm4b) & m_pf;
      gwc0 = G32(goff + 4u * ia);
      gwc1 = G32(goff + 4u * ib);
    }
    gw = sp_blend(m_res, rw, gw);
#ifdef ZPQ_PROF
    const unsigned long long pb2 = __builtin_readcyclecounter();
    prof[5] += pb2 - pb1;
#endif
    // (C) MATCH: pure register work (predicted byte and 2048/len were fetched at the byte boundary)
    const bool m_on = is_match && ra != 0;
    const int bitpos = B >= 0 ? B : 31 - __builtin_clz((unsigned)c8);   // bits of this byte already coded
    rc = m_on ? ((mpred >> (7 - bitpos)) & 1u) : rc;
    // no match: predict stretch(16384) = 0 (Predictor::predict0 case MATCH, "p[i]=0")
    const unsigned msx = m_on ? ((rc ? 0u - mdd : mdd) & 32767u) : 16384u;
    v0 = sp_blend(m_row, q0, gw);
    v1 = q1;
    // (D) one stretch lookup for every context-only component: ICM looks up cm >> 8, CM cm >> 17
    const unsigned sx = sp_blend(m_match, msx, v0 >> ctx_shift);
    const int st = sp_stretch(T, sx & 32767u);
    p = (int)sp_blend(m_ctx, (unsigned)st, (unsigned)p);
    dtv = (unsigned)T.dt[v0 & 0x3ffu];
    // (E) dependent components.  ISSE chains first, all at once, when every ISSE is fed by its left
    //     neighbour; then the rest in index order, unrolled with literal lanes; results are merged
    //     into p with the per-lane masks of `lk`.
    if constexpr (kIsseFast) {
      // every lane runs the same multiply-add; lanes that are not ISSE use weight 0 and addend p << 16,
      // which reproduces their p (all predictions are within +-2047, so the clamp is the identity)
      const int iw = (int)(v0 & m_isse);
      const int ia = (int)sp_blend(m_isse, v1 << 6, (unsigned)p << 16);
#pragma unroll
      for (int it = 0; it < kIsseDepth; ++it) p = sp_clamp2k(sp_mad24(iw, sp_shr1(p), ia) >> 16);
    }
    Dep<Chain, 0>::predict(T, lane, lk, c8, p, (int)v0, (int)v1, mixw, ssev, ssecx, ssetr, ssedt, pdv);
    pf_valid = more;
    sq = sp_squash(T, sp_clamp2k(p));
    const unsigned prr = sp_rlu((unsigned)sq, N - 1);     // p[N-1] is already within +-2047
#ifdef ZPQ_PROF
    prof[6] += __builtin_readcyclecounter() - pb2;
#endif
    return prr;
  };

  // ----------------------------------------------------------------- update
  auto update = [&](auto bitc, int y) __attribute__((always_inline)) {
    constexpr int B = decltype(bitc)::value;
    const bool byte_done = B >= 0 ? B == 7 : c8 >= 128;          // this bit completes the byte
    const int slot = hmap4 & 15;
    // inputs that live in other lanes: ISSE needs p[j]; MIX2 needs p[j] - p[k]
    int pj, pdiff = 0;
    if constexpr (kIsseFast) pj = sp_shr1(p);
    else pj = __shfl(p, (int)(a2 & 63));
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr CompK c = Chain::comp[i];
      if constexpr (c.type == C_MIX2) pdiff |= (int)((unsigned)pdv[i] & lk.is[i]);
    });
    // every lane's LDS lookups, issued together
    const unsigned nsv = y ? nspair >> 8 : nspair & 255u;
    const unsigned count = v0 & 0x3ffu;
    const int yq = y * 32767;
    const int err = yq - sq;
    // bit-history row and side table (ICM: one word; ISSE: two weights); idle lanes hit their dummies
    row_set_nb<(B >= 0 ? (B & 3) : -1)>(row0, row1, row2, row3, slot, nsv);
    const unsigned n0 = sp_blend(m_icm, v0 + (unsigned)((int)((unsigned)yq - (v0 >> 8)) >> 2),
                              (unsigned)sp_clamp512k((int)v0 + (sp_mad24(err, pj, 1 << 12) >> 13)));
    const unsigned n1 = (unsigned)sp_clamp512k((int)v1 + ((err + 16) >> 5));
    const unsigned e0 = (bh << bh_shift) & m_row;
    const unsigned el = side_global ? 0u : e0;
    L32(ldsq + 4u * el) = n0;
    L32(n1base + ((4u * el) & m_lds2)) = n1;
    if constexpr (Chain::ANY_GLOBAL_SIDE) {
      // idle lanes write their dummy; an ICM lane keeps word e0+1 (the next entry) unchanged
      const unsigned sidx = side_global ? e0 : 0u;
      G32(soff + 4u * sidx) = n0;
      G32((side_global && is_isse) ? soff + 4u * sidx + 4u : dummy + 4u) = n1;
      le0 = sidx; ln0 = n0; ln1 = is_isse ? n1 : v1;
    }
    // per-bit global word: CM (Predictor::train) or MIX2 weight; idle lanes write their dummy
    const int errcm = yq - (int)(v0 >> 17);
    const unsigned cm_new = v0 + ((unsigned)__mul24(errcm, (int)dtv) & 0xFFFFFC00u) + (count < limit ? 1u : 0u);
    const int err2 = __mul24(err, (int)a4) >> 5;
    const int w2 = min(max((int)v0 + (sp_mad24(err2, pdiff, 1 << 12) >> 13), 0), 65535);   // 19-bit x 13-bit
    const unsigned gnew = sp_blend(m_cm, cm_new, (unsigned)w2);
    G32(goff + 4u * gidx) = gnew;
    rw = gnew;
    // MATCH (Predictor::update0 case MATCH, libzpaq.cpp:1985-2008)
    ra = (is_match && (int)rc != y) ? 0u : ra;
    if (byte_done && is_match) {
      const unsigned mask = mask1;
      G8(off1 + (rlimit & mask)) = (unsigned char)(c8 * 2 + y);
      rlimit = (rlimit + 1) & mask;
      const unsigned eo = off0 + 4u * (h & mask0);
      if (ra == 0) {
        rb = rlimit - G32(eo);
        if (rb & mask)
          while (ra < 255 && G8(off1 + ((rlimit - ra - 1) & mask)) == G8(off1 + ((rlimit - ra - rb - 1) & mask))) ++ra;
      } else ra += ra < 255;
      G32(eo) = rlimit;
      if (ra != 0) { mpred = G8(off1 + ((rlimit - rb) & mask)); mdd = T.dt2k[ra]; }
    }
    Dep<Chain, 0>::update(T, arena, lane, lk, y, sq, p, mixw, mixrow, ssev, ssecx, ssetr, ssedt);
    ylast = y;
  };

  // The ENCODER knows the byte before it codes it, so it runs HCOMP one byte ahead: at the start of
  // byte k it already computes the contexts of byte k+1 and touches the lines byte k+1 will probe
  // first (hash rows of the first nibble, bit-0 MIX/SSE rows, CM line), which hides the only HBM
  // round trip left exposed per byte.  The decoder cannot (the byte is its output).
  // HCOMP is wave-uniform work: every lane runs the same program on the same machine state and the
  // same M/H/R (identical stores from all lanes coalesce).  The host-side emulator (tests/emu) runs
  // lanes one after the other, so there only lane 0 may apply the program's read-modify-writes.
  auto run_hcomp = [&](unsigned input) __attribute__((always_inline)) -> int {
#ifndef ZPQ_EMU
    return Chain::hcomp(input, vm_b, vm_c, vm_d, vm_f, vm_M, vm_H, vm_R);
#else
    int e = 0;
    if (lane == 0) e = Chain::hcomp(input, vm_b, vm_c, vm_d, vm_f, vm_M, vm_H, vm_R);
    return __builtin_amdgcn_readfirstlane(e);
#endif
  };
  unsigned h_next = 0, ka0 = 0, ka1 = 0, ka2 = 0;
  auto run_ahead = [&](int ch) __attribute__((always_inline)) -> int {
    ZPQ_KEEP3(ka0, ka1, ka2);
    SP_PROF_BEGIN
    const int e = run_hcomp((unsigned)ch);
    SP_PROF_END(2)
    if (e) return e;
    h_next = vm_H[(unsigned)lane & Chain::HMASK];
    const unsigned cx = h_next + 16u;
    ka0 = G32(roff + ((cx * 16u) & (rmask - 15u)));
    ka1 = G32(goff + 4u * ((is_cm ? (h_next ^ 1u) : (h_next + (1u & a5))) & gmask));
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr CompK c = Chain::comp[i];
      if constexpr (c.type == C_MIX) {
        hmix_n[c.slot] = sp_rlu(h_next, i);
        const unsigned r = ((hmix_n[c.slot] + (1u & c.a5)) & c.mask0) * c.stride;
        ka2 ^= G32(mixbase[c.slot] + 4u * r);
      } else if constexpr (c.type == C_SSE) {
        hsse_n[c.slot] = sp_rlu(h_next, i);
        const unsigned cx0 = ((hsse_n[c.slot] + 1u) * 32u) & c.mask0;
        ka2 ^= G32(ssebase[c.slot] + 4u * cx0);
      }
    });
    return 0;
  };

  auto after_bit = [&](auto bitc, int y) __attribute__((always_inline)) -> int {   // c8 / hmap4 bookkeeping (libzpaq.cpp:2055-2065)
    constexpr int B = decltype(bitc)::value;
    { SP_PROF_BEGIN update(bitc, y); SP_PROF_END(1) }
    c8 += c8 + y;
    if (B >= 0 ? B == 7 : c8 >= 256) {
      if constexpr (DEC) {
        SP_PROF_BEGIN
        const int e = run_hcomp((unsigned)(c8 - 256));
        SP_PROF_END(2)
        if (e) return e;
        h = vm_H[(unsigned)lane & Chain::HMASK];
        static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          constexpr CompK c = Chain::comp[i];
          if constexpr (c.type == C_MIX) hmix[c.slot] = sp_rlu(h, i);
          if constexpr (c.type == C_SSE) hsse[c.slot] = sp_rlu(h, i);
        });
      } else {
        h = h_next;            // computed by run_ahead() when this byte started
#pragma unroll
        for (int k = 0; k < NMIX; ++k) hmix[k] = hmix_n[k];
#pragma unroll
        for (int k = 0; k < NSSE; ++k) hsse[k] = hsse_n[k];
      }
      hmap4 = 1;
      c8 = 1;
    } else if (B >= 0 ? B == 3 : (c8 >= 16 && c8 < 32)) {
      hmap4 = (hmap4 & 0xf) << 5 | y << 4 | 1;
    } else {
      hmap4 = (hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + y) & 0xf);
    }
    return 0;
  };

  // the 8 bits of a byte: unrolled with the position as a compile-time tag, or a plain loop (tag -1)
#ifndef ZPQ_UNROLL_BITS
#define ZPQ_UNROLL_BITS 1
#endif
  auto for_bits = [&](auto&& f) __attribute__((always_inline)) {
    if constexpr (ZPQ_UNROLL_BITS) {
      static_for<0, 8>([&](auto ic) __attribute__((always_inline)) { f(ic, decltype(ic)::value); });
    } else {
      for (int pos = 0; pos < 8; ++pos) f(IC<-1>{}, pos);
    }
  };

  if (!DEC) {
    unsigned n = 0;
    auto encode = [&](int y, unsigned pr) __attribute__((always_inline)) {
      const unsigned mid = low + (unsigned)(((unsigned long long)(high - low) * pr) >> 16);
      if (y) high = mid; else low = mid + 1;
      while ((high ^ low) < 0x1000000u) {
        if (n < job.out_cap && lane == 0) out_ptr[n] = (unsigned char)(high >> 24);
        ++n;
        high = high << 8 | 255u;
        low = low << 8;
        low += (low == 0);
      }
    };
    for (unsigned k = 0; k < job.in_len && !status; ++k) {
      const int ch = (int)sp_uni(in_ptr[k]);
      status = run_ahead(ch);
      if (status) break;
      encode(0, 0);
      for_bits([&](auto bitc, int pos) __attribute__((always_inline)) {
        if (status) return;
        unsigned pr;
        { SP_PROF_BEGIN pr = predict(bitc); SP_PROF_END(0) }
        const int y = (ch >> (7 - pos)) & 1;
        encode(y, pr * 2 + 1);
        status = after_bit(bitc, y);
        ++steps;
      });
    }
    if (!status) encode(1, 0);
    if (!status && n > job.out_cap) status = 3;
    if (lane == 0) { res[rslot].out_len = n; res[rslot].consumed = job.in_len; }
  } else {
    unsigned rp = 0, n = 0, curr = 0;
    bool eos = false;
    // A block of several segments (job.nseg > 1): each segment has its own coded range and ends with its own
    // end-of-stream flag; the decoder re-reads 4 bytes at the start of each (Decoder::decompress, libzpaq.cpp:2129),
    // the model and the range state run on.
    typedef __attribute__((address_space(1))) SegRange g_seg;
    g_seg* const segs = (g_seg*)sp_uni64((unsigned long long)job.segs);
    const unsigned nseg = max(sp_uni(job.nseg), 1u);
    unsigned in_end = job.in_len;
    // Decoder::decode (libzpaq.cpp:2159-2181); sets `status` on a corrupt or truncated stream
    auto decode = [&](unsigned pr) __attribute__((always_inline)) -> int {
      if (curr < low || curr > high) { status = 2; return 0; }
      const unsigned mid = low + (unsigned)(((unsigned long long)(high - low) * pr) >> 16);
      int y;
      if (curr <= mid) { y = 1; high = mid; } else { y = 0; low = mid + 1; }
      while ((high ^ low) < 0x1000000u) {
        high = high << 8 | 255u;
        low = low << 8;
        low += (low == 0);
        if (rp >= in_end) { status = 6; break; }
        curr = curr << 8 | sp_uni(in_ptr[rp++]);
      }
      return y;
    };
    for (unsigned seg = 0; seg < nseg && !status; ++seg) {
      if (nseg > 1) { rp = sp_uni(segs[seg].in_begin); in_end = sp_uni(segs[seg].in_end); }
      eos = false;
      curr = 0;
      for (int i = 0; i < 4; ++i) {
        if (rp >= in_end) { status = 6; break; }
        curr = curr << 8 | sp_uni(in_ptr[rp++]);
      }
      while (!status && !eos && n < job.out_cap) {
        int ch = 1;
        const int flag = decode(0);                       // end-of-stream flag, coded with p = 0
        if (status) break;
        if (flag) { eos = true; if (curr != 0) status = 2; break; }
        for_bits([&](auto bitc, int) __attribute__((always_inline)) {
          if (status) return;
          const unsigned pr = predict(bitc) * 2 + 1;
          const int y = decode(pr);
          if (status) return;
          ch += ch + y;
          status = after_bit(bitc, y);
          ++steps;
        });
        if (status || eos) break;
        if (lane == 0) out_ptr[n] = (unsigned char)(ch - 256);
        ++n;
      }
      if (nseg > 1 && lane == 0) { segs[seg].out_end = n; segs[seg].status = eos ? 0u : 1u; }
      if (!eos) break;                                    // capacity reached (or an error) inside this segment
    }
    if (lane == 0) { res[rslot].out_len = n; res[rslot].consumed = eos ? rp : 0; }
  }
  if (lane == 0) { res[rslot].status = status; res[rslot].steps = steps; }
#ifdef ZPQ_PROF
  if (lane == 0 && b == 0) {
    prof[3] = __builtin_readcyclecounter() - prof_t0;
    printf("[zpq spec prof] block 0: steps=%u cycles/bit: predict=%.0f (rows=%.0f loads=%.0f chain=%.0f) update=%.0f hcomp=%.0f total=%.0f\n",
           steps, (double)prof[0] / steps, (double)prof[4] / steps, (double)prof[5] / steps, (double)prof[6] / steps,
           (double)prof[1] / steps, (double)prof[2] / steps, (double)prof[3] / steps);
  }
#endif
}

// ---- compile-time walk over the dependent components -------------------------
template <class Chain, int I>
struct Dep {
  template <int NM, int NS>
  static __device__ __forceinline__ void predict(const SpecTables& T, int lane, const LaneK<Chain::N, NM, NS>& lk, int c8,
                                                 int& p, int w0, int w1, int (&mixw)[NM], unsigned (&ssev)[NS],
                                                 unsigned (&ssecx)[NS], unsigned (&ssetr)[NS], unsigned (&ssedt)[NS],
                                                 int (&pdv)[Chain::N]) {
    if constexpr (I < Chain::N) {
      constexpr CompK c = Chain::comp[I];
      if constexpr (c.type == C_ISSE) {
        if constexpr (!isse_left_fed<Chain>()) {
          const int pj = sp_rl(p, (int)c.a2);
          const int val = sp_clamp2k(sp_mad24(w0, pj, w1 * 64) >> 16);
          p = (int)sp_blend(lk.is[I], (unsigned)val, (unsigned)p);
        }
      } else if constexpr (c.type == C_AVG) {
        const int pj = sp_rl(p, (int)c.a1), pk = sp_rl(p, (int)c.a2);
        const int val = (pj * (int)c.a3 + pk * (256 - (int)c.a3)) >> 8;
        p = (int)sp_blend(lk.is[I], (unsigned)val, (unsigned)p);
      } else if constexpr (c.type == C_MIX2) {
        const int pj = sp_rl(p, (int)c.a2), pk = sp_rl(p, (int)c.a3);
        pdv[I] = pj - pk;                                                   // update trains on this difference
        const int val = sp_mad24(w0, pj, __mul24(65536 - w0, pk)) >> 16;   // 17-bit x 12-bit products
        p = (int)sp_blend(lk.is[I], (unsigned)val, (unsigned)p);
      } else if constexpr (c.type == C_MIX) {
        // inputs p[j..j+m-1] sit in lanes j..j+m-1; weight t sits in lane t
        int pin = p;
        if constexpr (c.a2 != 0) pin = __shfl(p, (lane + (int)c.a2) & 63);
        const int x = (int)((unsigned)__mul24(mixw[c.slot] >> 8, pin) & lk.mixin[c.slot]);
        const int val = sp_clamp2k(sp_lanes_sum<(int)c.a3>(x) >> 8);
        p = (int)sp_blend(lk.is[I], (unsigned)val, (unsigned)p);
      } else if constexpr (c.type == C_SSE) {
        int pq = sp_rl(p, (int)c.a2) + 992;
        pq = min(max(pq, 0), 1983);
        const int wt = pq & 63;
        pq >>= 6;
        const unsigned e0 = sp_rlu(ssev[c.slot], pq), e1 = sp_rlu(ssev[c.slot], pq + 1);
        const int val = sp_stretch(T, ((e0 >> 10) * (unsigned)(64 - wt) + (e1 >> 10) * (unsigned)wt) >> 13);
        p = (int)sp_blend(lk.is[I], (unsigned)val, (unsigned)p);
        ssecx[c.slot] += (unsigned)(pq + (wt >> 5));        // element trained in update ...
        ssetr[c.slot] = (wt >> 5) ? e1 : e0;                // ... which is one of the two just read
        ssedt[c.slot] = (unsigned)T.dt[ssetr[c.slot] & 0x3ffu];
      }
      Dep<Chain, I + 1>::predict(T, lane, lk, c8, p, w0, w1, mixw, ssev, ssecx, ssetr, ssedt, pdv);
    }
  }

  template <int NM, int NS>
  static __device__ __forceinline__ void update(const SpecTables& T, g_u8* arena, int lane,
                                                const LaneK<Chain::N, NM, NS>& lk, int y, int sq, int p,
                                                int (&mixw)[NM], unsigned (&mixrow)[NM], unsigned (&ssev)[NS],
                                                unsigned (&ssecx)[NS], unsigned (&ssetr)[NS], unsigned (&ssedt)[NS]) {
    if constexpr (I < Chain::N) {
      constexpr CompK c = Chain::comp[I];
      if constexpr (c.type == C_MIX) {
        const int err = ((y * 32767 - sp_rl(sq, I)) * (int)c.a4) >> 4;   // sq = squash(p) of every lane, computed once
        int pin = p;
        if constexpr (c.a2 != 0) pin = __shfl(p, (lane + (int)c.a2) & 63);
        const int w = sp_clamp512k(mixw[c.slot] + (sp_mad24(err, pin, 1 << 12) >> 13));
        const unsigned wo = lk.mixst[c.slot] + ((4u * mixrow[c.slot]) & lk.mixin[c.slot]);
        *(g_i32*)(arena + wo) = w;
      } else if constexpr (c.type == C_SSE) {
        // Predictor::train on cm[cxt]; the word is still in lane (cxt & 31) of the row registers
        const unsigned e = ssecx[c.slot];
        const unsigned v = ssetr[c.slot];
        const unsigned count = v & 0x3ffu;
        const int err = y * 32767 - (int)(v >> 17);
        const unsigned prod = (unsigned)__mul24(err, (int)ssedt[c.slot]);    // low 32 bits, like Predictor::train
        const unsigned nv = v + (prod & 0xFFFFFC00u) + (count < c.limit ? 1u : 0u);
        *(g_u32*)(arena + lk.ssest[c.slot] + ((4u * (e & c.mask0)) & lk.lane0)) = nv;
      }
      Dep<Chain, I + 1>::update(T, arena, lane, lk, y, sq, p, mixw, mixrow, ssev, ssecx, ssetr, ssedt);
    }
  }
};

}  // namespace zpq
