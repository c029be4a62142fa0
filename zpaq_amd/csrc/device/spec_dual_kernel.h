// Decoder with TWO ZPAQ blocks per wavefront: component i of the first block on lane i, of the second on lane 32 + i
// (chains of up to 32 components), Chain::WAVES / 2 wavefronts per workgroup.
//
// Why: with two wavefronts per SIMD the one-block-per-wavefront decoder (spec_kernel.h) is bound by the SIMD's vector
// ALU -- every instruction occupies it for 4 cycles whatever the number of useful lanes, and a 23-component chain uses 23
// of 64.  Here one instruction stream serves two blocks.  The price: nothing is wave-uniform any more.  What the
// one-block kernel keeps on the scalar unit -- the byte being built (c8), the nibble map (hmap4), the arithmetic coder,
// the MIX / SSE row selection, HCOMP -- is per-HALF state held redundantly by the 32 lanes of a half (every lane of a
// half computes the same values), and a value handed from component to component is broadcast inside its half
// (two v_readlane + a select) instead of across the wavefront.
//
// The two halves run in lockstep: the flag bit and the 8 bits of a byte, position by position.  A half whose block has
// ended (end of stream, capacity, error) keeps stepping -- every cross-lane operation is executed by all 64 lanes -- but
// no longer reads its input, writes its output or runs HCOMP; what it still does to its own model state is of no
// consequence.  A half without a block (odd block count) is a half of idle lanes: bases on the first half's dummy
// line, masks zero.
//
// The per-lane model arithmetic is spec_kernel.h's, statement for statement (bit-exact with Predictor::predict0 /
// update0, libzpaq.cpp:1854-2066; Decoder::decompress / decode, 2104-2181); the generated `Chain` is the one of the
// 8-blocks-per-workgroup shape (LDS offsets inside a block's 15 KiB region), compiled with ZPQ_LANE_VM so that HCOMP's
// condition flag is per lane.  Blocks of several segments stay with the one-block kernel.
#pragma once
#ifndef ZPQ_LANE_VM
#define ZPQ_LANE_VM 1
#endif
#include "spec_kernel.h"

namespace zpq {

#ifdef ZPQ_EMU
__device__ __forceinline__ bool dual_any(bool x) { return emu::wave_any(x); }
#else
__device__ __forceinline__ bool dual_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }
#endif

// value of lane (half * 32 + j) for every lane of that half
__device__ __forceinline__ int dual_bc(int v, int j, bool upper) {
  const int a = sp_rl(v, j), b = sp_rl(v, 32 + j);
  return upper ? b : a;
}
__device__ __forceinline__ unsigned dual_bcu(unsigned v, int j, bool upper) { return (unsigned)dual_bc((int)v, j, upper); }

// sum over the lanes 0..LANES-1 of each half (the other lanes hold 0)
template <int LANES>
__device__ __forceinline__ int dual_half_sum(int x, bool upper) {
  int v = x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
  int a = sp_rl(v, 15), b = sp_rl(v, 47);
  if constexpr (LANES > 16) { a += sp_rl(v, 31); b += sp_rl(v, 63); }
  return upper ? b : a;
}

template <class Chain, int I>
struct DualDep;

template <class Chain>
__device__ __forceinline__ void spec_dual_decode_body(const BlockJob* jobs, BlockResult* res, unsigned nblocks,
                                                      const DeviceTables* tb) {
  constexpr int N = Chain::N;
  static_assert(N >= 1 && N <= 32, "two blocks per wavefront: chains of up to 32 components");
  static_assert(Chain::WAVES == 8, "the chain's LDS plan must be the one of the 8-blocks-per-workgroup shape");
  constexpr int NMIX = Chain::NMIX > 0 ? Chain::NMIX : 1;
  constexpr int NSSE = Chain::NSSE > 0 ? Chain::NSSE : 1;
  constexpr int kBlocks = 8;                                  // per workgroup: 4 wavefronts x 2
  constexpr int kRegion = spec_wave_lds_bytes(8);             // LDS of one block
  static_assert((int)sizeof(SpecTables) + kBlocks * kRegion <= kSpecLdsBudget, "LDS budget");

  __shared__ SpecTables T;
  __shared__ __attribute__((aligned(16))) unsigned char wave_lds[kBlocks][kRegion];
  for (unsigned i = threadIdx.x; i < 16384u; i += blockDim.x) T.stretch_hi[i] = tb->stretch[16384u + i];
  for (unsigned i = threadIdx.x; i < 1344u; i += blockDim.x) T.squash_mid[i] = tb->squash[1376u + i];
  for (unsigned i = threadIdx.x; i < 1024u; i += blockDim.x) { T.dt[i] = tb->dt[i]; T.ns[i] = tb->ns[i]; }
  for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) T.dt2k[i] = (uint16_t)tb->dt2k[i];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ci = lane & 31;                                   // component of this lane
  const bool upper = lane >= 32;                              // second block of the wavefront
  const unsigned b0 = (blockIdx.x * (unsigned)(kBlocks / 2) + (unsigned)wave) * 2u;
  const unsigned b = b0 + (upper ? 1u : 0u);
  const bool live = b < nblocks;                              // this half has a block
  const bool wave_live = b0 < nblocks;
  // the first block's arena is the (wave-uniform) base of every table address; the second block's tables are reached
  // through a 32-bit offset added to the per-lane table bases (the engine lays the arenas of a batch out back to back)
  const BlockJob job0 = jobs[wave_live ? b0 : 0];
  const BlockJob job = jobs[live ? b : (wave_live ? b0 : 0)];
  g_u8* const arena = (g_u8*)sp_uni64((unsigned long long)job0.arena);
  const unsigned long long delta64 = (unsigned long long)job.arena - (unsigned long long)job0.arena;
  const unsigned hoff = live ? (unsigned)delta64 : 0u;
  const g_u8* const in_ptr = (const g_u8*)job.in;             // per half
  g_u8* const out_ptr = (g_u8*)job.out;
  const unsigned in_len = job.in_len, out_cap = job.out_cap, rslot = job.res_slot;
  lds_u8* const wl = (lds_u8*)&wave_lds[wave * 2 + (upper ? 1 : 0)][0];

  // ---- per-lane component constants ----
  const unsigned dummy = (unsigned)Chain::OFF_RUN;            // the first block's dummy line: idle lanes of both halves
  const unsigned dummy_lds = (unsigned)(kRegion - 512) + (unsigned)ci * 8u;
  unsigned a2 = 0, a4 = 0, a5 = 0, limit = 0, mask0 = 0, mask1 = 63, sizebits = 0;
  unsigned off0 = dummy, off1 = dummy;
  int ldsoff = -1;
  unsigned ctype = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (ci == i && live) {
      const CompK c = Chain::comp[i];
      ctype = c.type;
      a2 = c.a2; a4 = c.a4; a5 = c.a5;
      limit = c.limit; mask0 = c.mask0; sizebits = c.a1 + 2;
      off0 = (unsigned)c.t0 + hoff;
      if (c.type == C_ICM || c.type == C_ISSE || c.type == C_MATCH) { off1 = (unsigned)c.t1 + hoff; mask1 = c.mask1; }
      ldsoff = c.lds;
    }
  }
  auto G32 = [&](unsigned off) __attribute__((always_inline)) -> g_u32& { return *(g_u32*)(arena + off); };
  auto G8 = [&](unsigned off) __attribute__((always_inline)) -> g_u8& { return *(g_u8*)(arena + off); };
  auto G128 = [&](unsigned off) __attribute__((always_inline)) -> g_u128& { return *(g_u128*)(arena + off); };
  auto L32 = [&](unsigned off) __attribute__((always_inline)) -> lds_u32& { return *(lds_u32*)(wl + off); };

  // side tables -> LDS, HCOMP's H array cleared: each half for its block
  if (live) {
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr CompK c = Chain::comp[decltype(ic)::value];
      if constexpr (c.lds >= 0 && (c.type == C_ICM || c.type == C_ISSE)) {
        constexpr int words = c.type == C_ICM ? 256 : 512;
        const g_u32* src = (const g_u32*)(arena + (unsigned)c.t0 + hoff);
        lds_u32* dst = (lds_u32*)(wl + c.lds);
        for (int k = ci; k < words; k += 32) dst[k] = src[k];
      }
    });
    if constexpr (Chain::H_LDS >= 0)
      for (unsigned k = ci; k <= Chain::HMASK; k += 32) ((lds_u32*)(wl + Chain::H_LDS))[k] = 0;
  }
  L32(dummy_lds) = 0;
  L32(dummy_lds + 4) = 0;
  __syncthreads();
  if (!wave_live) return;

  // HCOMP machine of this half (every lane of the half runs it: identical values, identical stores)
  unsigned vm_b = 0, vm_c = 0, vm_d = 0, vm_f = 0;
  g_u8* const vm_M = arena + (unsigned)Chain::OFF_M + hoff;
  g_u32* const vm_R = (g_u32*)(arena + (unsigned)Chain::OFF_R + hoff);
  auto vm_H = [&]() {
    if constexpr (Chain::H_LDS >= 0) return (lds_u32*)(wl + Chain::H_LDS);
    else return (g_u32*)(arena + (unsigned)Chain::OFF_H + hoff);
  }();

  constexpr bool kIsseFast = isse_left_fed<Chain>();
  constexpr int kIsseDepth = isse_depth<Chain>();
  const bool is_cm = ctype == C_CM, is_icm = ctype == C_ICM, is_isse = ctype == C_ISSE;
  const bool is_match = ctype == C_MATCH, is_mix2 = ctype == C_MIX2;
  const bool has_row = is_icm || is_isse;
  const bool is_ctx = is_cm || is_icm || is_match;
  const unsigned ctx_shift = is_icm ? 8u : 17u;
  const bool gl = is_cm || is_mix2;
  const bool pf_lane = is_cm ? mask0 >= 511u : (is_mix2 && a5 == 255u && mask0 >= 255u);
  const bool resident = gl && mask0 == 0u;
  const unsigned goff = gl ? off0 : dummy;
  const unsigned gmask = gl ? mask0 : 0u;
  const unsigned rmask = has_row ? mask1 : 63u;
  const unsigned roff = has_row ? off1 : dummy;
  const unsigned ldsq = (has_row && ldsoff >= 0) ? (unsigned)ldsoff : dummy_lds;
  const bool side_global = has_row && ldsoff < 0;
  const unsigned soff = side_global ? off0 : dummy;
  auto lane_mask = [&](bool x) __attribute__((always_inline)) -> unsigned {
    unsigned m = x ? 0xFFFFFFFFu : 0u;
    ZPQ_OPAQUE(m);
    return m;
  };
  const unsigned m_cm = lane_mask(is_cm), m_isse = lane_mask(is_isse), m_icm = lane_mask(is_icm);
  const unsigned m_match = lane_mask(is_match), m_row = lane_mask(has_row), m_ctx = lane_mask(is_ctx);
  const unsigned m_res = lane_mask(resident), m_pf = lane_mask(pf_lane);
  const unsigned m_lds2 = lane_mask(is_isse && !side_global);
  const unsigned bh_shift = is_isse ? 1u : 0u;
  const unsigned q1off = is_icm ? 0u : 4u;
  const unsigned n1base = (is_isse && !side_global) ? ldsq + 4u : dummy_lds + 4u;

  // ---- per-lane mutable state ----
  unsigned bh = 0, gidx = 0, h = 0;
  int p = 0;
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (Chain::comp[i].type == C_CONS) { if (ci == i) p = ((int)Chain::comp[i].a1 - 128) * 4; }
  });
  unsigned v0 = 0, v1 = 0;
  unsigned row0 = 0, row1 = 0, row2 = 0, row3 = 0, rowoff = 0;
  unsigned touch_a = 0, touch_b = 0;
  unsigned ra = 0, rb = 0, rc = 0, rlimit = 0, mpred = 0, mdd = 0;
  int mixw[NMIX];
  unsigned mixrow[NMIX];
  unsigned ssev[NSSE], ssecx[NSSE];
  unsigned gwc0 = 0, gwc1 = 0;
  int mixc0[NMIX], mixc1[NMIX];
  unsigned ssec0[NSSE], ssec1[NSSE];
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { mixw[k] = 0; mixrow[k] = 0; mixc0[k] = 0; mixc1[k] = 0; }
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { ssev[k] = 0; ssecx[k] = 0; ssec0[k] = 0; ssec1[k] = 0; }
  // MIX / SSE rows: lane t of a half reads word t of the half's selected row
  unsigned mixbase[NMIX], ssebase[NSSE], mixst[NMIX], mixin[NMIX], ssest[NSSE];
  int mixsrc[NMIX];                                           // lane that holds this lane's MIX input
  unsigned isl[N];                                            // all ones in the lanes (one per live half) of component i
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { mixbase[k] = dummy; mixst[k] = dummy; mixin[k] = 0; mixsrc[k] = lane; }
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { ssebase[k] = dummy; ssest[k] = dummy; }
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr CompK c = Chain::comp[i];
    isl[i] = 0;
    if constexpr (c.type == C_AVG || c.type == C_MIX2 || c.type == C_MIX || c.type == C_SSE || c.type == C_ISSE)
      isl[i] = lane_mask(ci == i && live);
    if constexpr (c.type == C_MIX) {
      static_assert(c.a2 + c.a3 <= 32, "MIX inputs must sit inside one half");
      mixbase[c.slot] = (unsigned)c.t0 + hoff + 4u * (unsigned)min(ci, (int)c.a3 - 1);
      ZPQ_OPAQUE(mixbase[c.slot]);
      mixin[c.slot] = lane_mask(ci < (int)c.a3 && live);
      mixst[c.slot] = (ci < (int)c.a3 && live) ? (unsigned)c.t0 + hoff + 4u * (unsigned)ci : dummy;
      ZPQ_OPAQUE(mixst[c.slot]);
      mixsrc[c.slot] = (lane & 32) | ((ci + (int)c.a2) & 31);
    } else if constexpr (c.type == C_SSE) {
      ssebase[c.slot] = (unsigned)c.t0 + hoff + 4u * (unsigned)ci;
      ZPQ_OPAQUE(ssebase[c.slot]);
      ssest[c.slot] = (ci == 0 && live) ? (unsigned)c.t0 + hoff : dummy;
      ZPQ_OPAQUE(ssest[c.slot]);
    }
  });
  const unsigned m_lane0 = lane_mask(ci == 0 && live);
  unsigned rw = G32(goff);
  unsigned nspair = 0, dtv = 0;
  int sq = 0;
  unsigned ssetr[NSSE], ssedt[NSSE];
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { ssetr[k] = 0; ssedt[k] = 0; }
  unsigned hmix[NMIX], hsse[NSSE];
#pragma unroll
  for (int k = 0; k < NMIX; ++k) hmix[k] = 0;
#pragma unroll
  for (int k = 0; k < NSSE; ++k) hsse[k] = 0;
  int pdiff = 0;                                              // MIX2 lane: p[j] - p[k] of this bit
  int ylast = 0;

  int c8 = 1, hmap4 = 1;                                      // per half (identical in its 32 lanes)
  unsigned low = 1, high = 0xFFFFFFFFu;
  unsigned steps = 0;
  int status = 0;

  auto g_index = [&](int c8x, int hm4x) __attribute__((always_inline)) -> unsigned {
    return (is_cm ? (h ^ (unsigned)hm4x) : (h + (unsigned)(c8x & (int)a5))) & gmask;
  };

  // ---------------------------------------------------------------- predict (bit position B of the byte, both halves)
  auto predict = [&](auto bitc) __attribute__((always_inline)) -> unsigned {
    constexpr int B = decltype(bitc)::value;
    constexpr bool nib = B == 0 || B == 4;
    constexpr bool pf_now = B > 0;
    constexpr bool last_of_nibble = B == 3;
    const int slot = hmap4 & 15;
    const int c8a = c8 * 2, c8b = c8 * 2 + 1;
    const int hm4a = last_of_nibble ? ((hmap4 & 0xf) << 5 | 1) : ((hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2) & 0xf));
    const int hm4b = last_of_nibble ? ((hmap4 & 0xf) << 5 | 1 << 4 | 1)
                                    : ((hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + 1) & 0xf));
    if constexpr (nib) {
      ZPQ_KEEP2(touch_a, touch_b);
      const unsigned cx = h + 16u * (unsigned)c8;
      const unsigned chk = (cx >> (sizebits & 31u)) & 255u;
      const unsigned h0 = (cx * 16u) & (rmask - 15u);
      uint4 r0 = G128(roff + h0);
      uint4 r1 = G128(roff + (h0 ^ 16u));
      uint4 r2 = G128(roff + (h0 ^ 32u));
      const uint4 oldrow = make_uint4(row0, row1, row2, row3);
      G128(roff + rowoff) = oldrow;
      if (rowoff == h0) r0 = oldrow;
      if (rowoff == (h0 ^ 16u)) r1 = oldrow;
      if (rowoff == (h0 ^ 32u)) r2 = oldrow;
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
    } else if constexpr (last_of_nibble) {
      const unsigned cxa = h + 16u * (unsigned)c8a, cxb = h + 16u * (unsigned)c8b;
      touch_a = G32(roff + ((cxa * 16u) & (rmask - 15u)));
      touch_b = G32(roff + ((cxb * 16u) & (rmask - 15u)));
    }
    bh = row_get_nb<(B & 3)>(row0, row1, row2, row3, slot);   // the position in the nibble is the same in both halves
    nspair = *(const unsigned short*)&T.ns[(bh & 255u) * 4u];
    const unsigned e0 = (bh << bh_shift) & m_row;
    const unsigned el = side_global ? 0u : e0;
    unsigned q0 = L32(ldsq + 4u * el);
    unsigned q1 = L32(ldsq + 4u * el + q1off);
    if constexpr (Chain::ANY_GLOBAL_SIDE) {
      const unsigned sidx = side_global ? e0 : 0u;
      const unsigned g0 = G32(soff + 4u * sidx), g1 = G32(soff + 4u * sidx + 4u);
      q0 = side_global ? g0 : q0;
      q1 = side_global ? g1 : q1;
    }
    unsigned gw;
    gidx = g_index(c8, hmap4);
    if constexpr (pf_now) {
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
        constexpr CompK c = Chain::comp[decltype(ic)::value];
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
      constexpr CompK c = Chain::comp[decltype(ic)::value];
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
    // MATCH
    const bool m_on = is_match && ra != 0;
    rc = m_on ? ((mpred >> (7 - B)) & 1u) : rc;
    const unsigned msx = m_on ? ((rc ? 0u - mdd : mdd) & 32767u) : 16384u;
    v0 = sp_blend(m_row, q0, gw);
    v1 = q1;
    const unsigned sx = sp_blend(m_match, msx, v0 >> ctx_shift);
    const int st = sp_stretch(T, sx & 32767u);
    p = (int)sp_blend(m_ctx, (unsigned)st, (unsigned)p);
    dtv = (unsigned)T.dt[v0 & 0x3ffu];
    if constexpr (kIsseFast) {
      const int iw = (int)(v0 & m_isse);
      const int ia = (int)sp_blend(m_isse, v1 << 6, (unsigned)p << 16);
#pragma unroll
      for (int it = 0; it < kIsseDepth; ++it) p = sp_clamp2k(sp_mad24(iw, sp_shr1(p), ia) >> 16);
    }
    DualDep<Chain, 0>::predict(T, upper, isl, mixin, mixsrc, p, (int)v0, (int)v1, mixw, ssev, ssecx, ssetr, ssedt, pdiff);
    sq = sp_squash(T, sp_clamp2k(p));
    return dual_bcu((unsigned)sq, N - 1, upper);
  };

  // ----------------------------------------------------------------- update
  auto update = [&](auto bitc, int y) __attribute__((always_inline)) {
    constexpr int B = decltype(bitc)::value;
    constexpr bool byte_done = B == 7;
    const int slot = hmap4 & 15;
    int pj;
    if constexpr (kIsseFast) pj = sp_shr1(p);
    else pj = __shfl(p, (lane & 32) | (int)(a2 & 31));
    const unsigned nsv = y ? nspair >> 8 : nspair & 255u;
    const unsigned count = v0 & 0x3ffu;
    const int yq = y * 32767;
    const int err = yq - sq;
    row_set_nb<(B & 3)>(row0, row1, row2, row3, slot, nsv);
    const unsigned n0 = sp_blend(m_icm, v0 + (unsigned)((int)((unsigned)yq - (v0 >> 8)) >> 2),
                              (unsigned)sp_clamp512k((int)v0 + (sp_mad24(err, pj, 1 << 12) >> 13)));
    const unsigned n1 = (unsigned)sp_clamp512k((int)v1 + ((err + 16) >> 5));
    const unsigned e0 = (bh << bh_shift) & m_row;
    const unsigned el = side_global ? 0u : e0;
    L32(ldsq + 4u * el) = n0;
    L32(n1base + ((4u * el) & m_lds2)) = n1;
    if constexpr (Chain::ANY_GLOBAL_SIDE) {
      const unsigned sidx = side_global ? e0 : 0u;
      G32(soff + 4u * sidx) = n0;
      G32((side_global && is_isse) ? soff + 4u * sidx + 4u : dummy + 4u) = n1;
    }
    const int errcm = yq - (int)(v0 >> 17);
    const unsigned cm_new = v0 + ((unsigned)__mul24(errcm, (int)dtv) & 0xFFFFFC00u) + (count < limit ? 1u : 0u);
    const int err2 = __mul24(err, (int)a4) >> 5;
    const int w2 = min(max((int)v0 + (sp_mad24(err2, pdiff, 1 << 12) >> 13), 0), 65535);
    const unsigned gnew = sp_blend(m_cm, cm_new, (unsigned)w2);
    G32(goff + 4u * gidx) = gnew;
    rw = gnew;
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
    DualDep<Chain, 0>::update(T, arena, upper, mixin, mixst, ssest, m_lane0, mixsrc, y, sq, p, mixw, mixrow, ssecx, ssetr, ssedt);
    ylast = y;
  };

  // ---- the coder of this half (Decoder::decode, libzpaq.cpp:2159-2181), on the vector unit ----
  unsigned rp = 0, nout = 0, curr = 0;
  bool run = live;                                            // this half is still decoding
  bool eos = false;
  if (live && (delta64 >> 31) != 0) { status = 8; run = false; }   // arenas not back to back within 2 GiB: not for this kernel
  auto decode = [&](unsigned pr) __attribute__((always_inline)) -> int {
    if (!run) return 0;
    if (curr < low || curr > high) { status = 2; run = false; return 0; }
    const unsigned mid = low + (unsigned)(((unsigned long long)(high - low) * pr) >> 16);
    int y;
    if (curr <= mid) { y = 1; high = mid; } else { y = 0; low = mid + 1; }
    while ((high ^ low) < 0x1000000u) {
      high = high << 8 | 255u;
      low = low << 8;
      low += (low == 0);
      if (rp >= in_len) { status = 6; run = false; break; }
      curr = curr << 8 | in_ptr[rp++];
    }
    return y;
  };
  auto run_hcomp = [&](unsigned input) __attribute__((always_inline)) -> int {
#ifndef ZPQ_EMU
    int e = 0;
    if (run) e = Chain::hcomp(input, vm_b, vm_c, vm_d, vm_f, vm_M, vm_H, vm_R);
    return e;
#else
    // the emulator runs the lanes one after the other: one lane per half applies the program's read-modify-writes
    int e = 0;
    if (run && ci == 0) e = Chain::hcomp(input, vm_b, vm_c, vm_d, vm_f, vm_M, vm_H, vm_R);
    return dual_bc(e, 0, upper);
#endif
  };

  for (int i = 0; i < 4; ++i) {
    if (!run) break;
    if (rp >= in_len) { status = 6; run = false; break; }
    curr = curr << 8 | in_ptr[rp++];
  }
  if (run && nout >= out_cap) run = false;
  while (dual_any(run)) {
    int ch = 1;
    const int flag = decode(0);                               // end-of-stream flag, coded with p = 0
    if (run && flag) { eos = true; if (curr != 0) status = 2; run = false; }
    static_for<0, 8>([&](auto bitc) __attribute__((always_inline)) {
      constexpr int B = decltype(bitc)::value;
      const unsigned pr = predict(bitc) * 2u + 1u;
      const int y = decode(pr);
      ch += ch + y;
      update(bitc, y);
      c8 += c8 + y;
      if constexpr (B == 7) {
        const int e = run_hcomp((unsigned)(c8 - 256));
        if (run && e) { status = e; run = false; }
        h = vm_H[(unsigned)ci & Chain::HMASK];
        static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          constexpr CompK c = Chain::comp[i];
          if constexpr (c.type == C_MIX) hmix[c.slot] = dual_bcu(h, i, upper);
          if constexpr (c.type == C_SSE) hsse[c.slot] = dual_bcu(h, i, upper);
        });
        hmap4 = 1;
        c8 = 1;
      } else if constexpr (B == 3) {
        hmap4 = (hmap4 & 0xf) << 5 | y << 4 | 1;
      } else {
        hmap4 = (hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + y) & 0xf);
      }
      if (run) ++steps;
    });
    if (run) {
      if (ci == 0) out_ptr[nout] = (unsigned char)(ch - 256);
      ++nout;
      if (nout >= out_cap) run = false;
    }
  }
  if (ci == 0 && live) {
    res[rslot].out_len = nout;
    res[rslot].consumed = eos ? rp : 0;
    res[rslot].status = status;
    res[rslot].steps = steps;
  }
}

// ---- the dependent components, broadcasts inside a half -------------------------
template <class Chain, int I>
struct DualDep {
  template <int NM, int NS>
  static __device__ __forceinline__ void predict(const SpecTables& T, bool upper, const unsigned (&isl)[Chain::N],
                                                 const unsigned (&mixin)[NM], const int (&mixsrc)[NM], int& p, int w0, int w1,
                                                 int (&mixw)[NM], unsigned (&ssev)[NS], unsigned (&ssecx)[NS],
                                                 unsigned (&ssetr)[NS], unsigned (&ssedt)[NS], int& pdiff) {
    if constexpr (I < Chain::N) {
      constexpr CompK c = Chain::comp[I];
      if constexpr (c.type == C_ISSE) {
        if constexpr (!isse_left_fed<Chain>()) {
          const int pj = dual_bc(p, (int)c.a2, upper);
          const int val = sp_clamp2k(sp_mad24(w0, pj, w1 * 64) >> 16);
          p = (int)sp_blend(isl[I], (unsigned)val, (unsigned)p);
        }
      } else if constexpr (c.type == C_AVG) {
        const int pj = dual_bc(p, (int)c.a1, upper), pk = dual_bc(p, (int)c.a2, upper);
        const int val = (pj * (int)c.a3 + pk * (256 - (int)c.a3)) >> 8;
        p = (int)sp_blend(isl[I], (unsigned)val, (unsigned)p);
      } else if constexpr (c.type == C_MIX2) {
        const int pj = dual_bc(p, (int)c.a2, upper), pk = dual_bc(p, (int)c.a3, upper);
        pdiff = (int)sp_blend(isl[I], (unsigned)(pj - pk), (unsigned)pdiff);
        const int val = sp_mad24(w0, pj, __mul24(65536 - w0, pk)) >> 16;
        p = (int)sp_blend(isl[I], (unsigned)val, (unsigned)p);
      } else if constexpr (c.type == C_MIX) {
        int pin = p;
        if constexpr (c.a2 != 0) pin = __shfl(p, mixsrc[c.slot]);
        const int x = (int)((unsigned)__mul24(mixw[c.slot] >> 8, pin) & mixin[c.slot]);
        const int val = sp_clamp2k(dual_half_sum<(int)c.a3>(x, upper) >> 8);
        p = (int)sp_blend(isl[I], (unsigned)val, (unsigned)p);
      } else if constexpr (c.type == C_SSE) {
        int pq = dual_bc(p, (int)c.a2, upper) + 992;
        pq = min(max(pq, 0), 1983);
        const int wt = pq & 63;
        pq >>= 6;
        const int base = upper ? 32 : 0;
        const unsigned e0 = __shfl(ssev[c.slot], base + pq), e1 = __shfl(ssev[c.slot], base + pq + 1);
        const int val = sp_stretch(T, ((e0 >> 10) * (unsigned)(64 - wt) + (e1 >> 10) * (unsigned)wt) >> 13);
        p = (int)sp_blend(isl[I], (unsigned)val, (unsigned)p);
        ssecx[c.slot] += (unsigned)(pq + (wt >> 5));
        ssetr[c.slot] = (wt >> 5) ? e1 : e0;
        ssedt[c.slot] = (unsigned)T.dt[ssetr[c.slot] & 0x3ffu];
      }
      DualDep<Chain, I + 1>::predict(T, upper, isl, mixin, mixsrc, p, w0, w1, mixw, ssev, ssecx, ssetr, ssedt, pdiff);
    }
  }

  template <int NM, int NS>
  static __device__ __forceinline__ void update(const SpecTables& T, g_u8* arena, bool upper, const unsigned (&mixin)[NM],
                                                const unsigned (&mixst)[NM], const unsigned (&ssest)[NS], unsigned m_lane0,
                                                const int (&mixsrc)[NM], int y, int sq, int p, int (&mixw)[NM],
                                                unsigned (&mixrow)[NM], unsigned (&ssecx)[NS], unsigned (&ssetr)[NS],
                                                unsigned (&ssedt)[NS]) {
    if constexpr (I < Chain::N) {
      constexpr CompK c = Chain::comp[I];
      if constexpr (c.type == C_MIX) {
        const int err = ((y * 32767 - dual_bc(sq, I, upper)) * (int)c.a4) >> 4;
        int pin = p;
        if constexpr (c.a2 != 0) pin = __shfl(p, mixsrc[c.slot]);
        const int w = sp_clamp512k(mixw[c.slot] + (sp_mad24(err, pin, 1 << 12) >> 13));
        const unsigned wo = mixst[c.slot] + ((4u * mixrow[c.slot]) & mixin[c.slot]);
        *(g_i32*)(arena + wo) = w;
      } else if constexpr (c.type == C_SSE) {
        const unsigned e = ssecx[c.slot];
        const unsigned v = ssetr[c.slot];
        const unsigned count = v & 0x3ffu;
        const int err = y * 32767 - (int)(v >> 17);
        const unsigned prod = (unsigned)__mul24(err, (int)ssedt[c.slot]);
        const unsigned nv = v + (prod & 0xFFFFFC00u) + (count < c.limit ? 1u : 0u);
        *(g_u32*)(arena + ssest[c.slot] + ((4u * (e & c.mask0)) & m_lane0)) = nv;
      }
      DualDep<Chain, I + 1>::update(T, arena, upper, mixin, mixst, ssest, m_lane0, mixsrc, y, sq, p, mixw, mixrow, ssecx, ssetr, ssedt);
    }
  }
};

}  // namespace zpq
