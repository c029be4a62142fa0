// EXPERIMENTAL -- two ZPAQ blocks per wavefront (lead 2).  Not selected by the engine;
// generated only on request (ZPAQ_AMD_SPEC_DUAL=1 for zpq_plan_spec_source) and exercised by the host-side
// wavefront emulator (tests/test_emu.py).  It has never run on a GPU.
//
// Same model code as spec_kernel.h, different mapping: lanes 0..31 carry the components of one block,
// lanes 32..63 those of another (chains of n <= 32 components).  Everything that was already
// lane-parallel now serves two blocks per instruction; what was wave-uniform (the arithmetic coder,
// c8 / hmap4, row indices, the SSE interpolation, HCOMP) becomes ordinary per-lane SIMT code whose
// value is uniform within a 32-lane group.  Cross-lane traffic stays inside a group:
//   * "p of component j"      -> ds_bpermute from lane (lane & 32) + j
//   * MIX dot product         -> DPP row scans + row_bcast:15, total in lane 31 / 63 of the group
//   * ISSE chains             -> wave_shr:1 as before (lane 32 is component 0 of its block, never an ISSE)
//   * SSE row                 -> 32 entries = exactly one group
// Both blocks advance in lock step (same bit position); a block that has finished keeps executing on dummy
// input with its stores disabled until its partner is done, so cross-lane operations stay wave-uniform.
#pragma once
#define ZPQ_DUAL 1
#include "spec_kernel.h"

namespace zpq {

template <class Chain, int I>
struct DepD;   // forward

// sum over lanes 0..LANES-1 of the caller's 32-lane group (the other lanes hold 0), broadcast to the group
template <int LANES>
__device__ __forceinline__ int sp_group_sum(int x, int gbase) {
  int v = x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
  if constexpr (LANES <= 16) return __shfl(v, gbase + 15);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // rows 1 and 3 += last lane of rows 0 and 2
  return __shfl(v, gbase + 31);
}

template <class Chain, bool DEC>
__device__ __forceinline__ void spec_kernel_body_dual(const BlockJob* jobs, BlockResult* res, unsigned nblocks,
                                                      const DeviceTables* tb) {
  constexpr int N = Chain::N;
  static_assert(N >= 1 && N <= 32, "two blocks per wavefront need chains of at most 32 components");
  constexpr int NMIX = Chain::NMIX > 0 ? Chain::NMIX : 1;
  constexpr int NSSE = Chain::NSSE > 0 ? Chain::NSSE : 1;
  constexpr int kWaves = Chain::WAVES;                          // wavefronts per workgroup; 2 blocks each
  constexpr int kGroupLds = spec_wave_lds_bytes(2 * kWaves);    // LDS of one block
  static_assert((int)sizeof(SpecTables) + 2 * kWaves * kGroupLds <= kSpecLdsBudget, "LDS budget");

  __shared__ SpecTables T;
  __shared__ __attribute__((aligned(16))) unsigned char group_lds[2 * kWaves][kGroupLds];
  for (unsigned i = threadIdx.x; i < 16384u; i += blockDim.x) T.stretch_hi[i] = tb->stretch[16384u + i];
  for (unsigned i = threadIdx.x; i < 1344u; i += blockDim.x) T.squash_mid[i] = tb->squash[1376u + i];
  for (unsigned i = threadIdx.x; i < 1024u; i += blockDim.x) { T.dt[i] = tb->dt[i]; T.ns[i] = tb->ns[i]; }
  for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) T.dt2k[i] = (uint16_t)tb->dt2k[i];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int gl = lane & 31;                 // component index inside the block
  const int gbase = lane & 32;              // first lane of my group
  const int grp = lane >> 5;
  const unsigned pair = blockIdx.x * kWaves + wave;
  const bool wave_live = 2u * pair < nblocks;
  // The job array is padded to an even number of entries by the launcher: the partner of a last odd
  // block is a job with in_len = 0, out_cap = 0 and an arena of its own.
  const BlockJob job = jobs[wave_live ? 2u * pair + (unsigned)grp : 0u];
  const unsigned long long arena0 = sp_uni64((unsigned long long)job.arena);      // group 0's arena: the wave's base
  g_u8* const arena = (g_u8*)arena0;
  const unsigned abase = (unsigned)((unsigned long long)job.arena - arena0);      // my block's arena, relative to it
  const g_u8* const in_ptr = (const g_u8*)job.in;                                  // per group
  g_u8* const out_ptr = (g_u8*)job.out;
  const unsigned in_len = job.in_len;
  unsigned cap = job.out_cap;               // 0 once the block has finished: disables its output stores
  const unsigned rslot = job.res_slot;
  lds_u8* const wl = (lds_u8*)&group_lds[wave * 2 + grp][0];

  auto grl = [&](int v, int j) __attribute__((always_inline)) -> int { return __shfl(v, gbase + j); };
  auto grlu = [&](unsigned v, int j) __attribute__((always_inline)) -> unsigned { return (unsigned)__shfl((int)v, gbase + j); };

  const unsigned dummy = abase + (unsigned)Chain::OFF_RUN;      // one shared line per block
  const unsigned dummy_lds = (unsigned)(kGroupLds - 512) + (unsigned)gl * 8u;
  unsigned a2 = 0, a4 = 0, a5 = 0, limit = 0, mask0 = 0, mask1 = 63, sizebits = 0;
  unsigned off0 = dummy, off1 = dummy;
  int ldsoff = -1;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (gl == i) {
      const CompK c = Chain::comp[i];
      a2 = c.a2; a4 = c.a4; a5 = c.a5;
      limit = c.limit; mask0 = c.mask0; sizebits = c.a1 + 2;
      off0 = abase + (unsigned)c.t0;
      if (c.type == C_ICM || c.type == C_ISSE || c.type == C_MATCH) { off1 = abase + (unsigned)c.t1; mask1 = c.mask1; }
      ldsoff = c.lds;
    }
  }
  auto G32 = [&](unsigned off) __attribute__((always_inline)) -> g_u32& { return *(g_u32*)(arena + off); };
  auto G8 = [&](unsigned off) __attribute__((always_inline)) -> g_u8& { return *(g_u8*)(arena + off); };
  auto G128 = [&](unsigned off) __attribute__((always_inline)) -> g_u128& { return *(g_u128*)(arena + off); };
  auto L32 = [&](unsigned off) __attribute__((always_inline)) -> lds_u32& { return *(lds_u32*)(wl + off); };

  if (wave_live) {
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr CompK c = Chain::comp[decltype(ic)::value];
      if constexpr (c.lds >= 0 && (c.type == C_ICM || c.type == C_ISSE)) {
        constexpr int words = c.type == C_ICM ? 256 : 512;
        const g_u32* src = (const g_u32*)(arena + abase + (unsigned)c.t0);
        lds_u32* dst = (lds_u32*)(wl + c.lds);
        for (int k = gl; k < words; k += 32) dst[k] = src[k];
      }
    });
    if constexpr (Chain::H_LDS >= 0)
      for (unsigned k = gl; k <= Chain::HMASK; k += 32) ((lds_u32*)(wl + Chain::H_LDS))[k] = 0;
    L32(dummy_lds) = 0;
    L32(dummy_lds + 4) = 0;
  }
  __syncthreads();
  if (!wave_live) return;

  unsigned vm_b = 0, vm_c = 0, vm_d = 0, vm_f = 0;
  g_u8* const vm_M = arena + abase + (unsigned)Chain::OFF_M;
  g_u32* const vm_R = (g_u32*)(arena + abase + (unsigned)Chain::OFF_R);
  auto vm_H = [&]() {
    if constexpr (Chain::H_LDS >= 0) return (lds_u32*)(wl + Chain::H_LDS);
    else return (g_u32*)(arena + abase + (unsigned)Chain::OFF_H);
  }();

  constexpr bool kIsseFast = isse_left_fed<Chain>();
  constexpr int kIsseDepth = isse_depth<Chain>();
  constexpr unsigned long long M_CM = type_mask<Chain>(C_CM), M_ICM = type_mask<Chain>(C_ICM),
                               M_ISSE = type_mask<Chain>(C_ISSE), M_MATCH = type_mask<Chain>(C_MATCH),
                               M_MIX2 = type_mask<Chain>(C_MIX2);
  const bool is_cm = (M_CM >> gl) & 1, is_icm = (M_ICM >> gl) & 1, is_isse = (M_ISSE >> gl) & 1;
  const bool is_match = (M_MATCH >> gl) & 1, is_mix2 = (M_MIX2 >> gl) & 1;
  const bool has_row = is_icm || is_isse;
  const bool is_ctx = is_cm || is_icm || is_match;
  const unsigned ctx_shift = is_icm ? 8u : 17u;
  const bool gword = is_cm || is_mix2;
  const bool pf_lane = is_cm ? mask0 >= 511u : (is_mix2 && a5 == 255u && mask0 >= 255u);
  const bool resident = gword && mask0 == 0u;
  const unsigned goff = gword ? off0 : dummy;
  const unsigned gmask = gword ? mask0 : 0u;
  const unsigned rmask = has_row ? mask1 : 63u;
  const unsigned roff = has_row ? off1 : dummy;
  const unsigned ldsq = (has_row && ldsoff >= 0) ? (unsigned)ldsoff : dummy_lds;
  const bool side_global = has_row && ldsoff < 0;
  const unsigned soff = side_global ? off0 : dummy;
  auto lane_mask = [&](bool b) __attribute__((always_inline)) -> unsigned {
    unsigned m = b ? 0xFFFFFFFFu : 0u;
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

  unsigned bh = 0, gidx = 0, h = 0;
  int p = 0;
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (Chain::comp[i].type == C_CONS) { if (gl == i) p = ((int)Chain::comp[i].a1 - 128) * 4; }
  });
  unsigned v0 = 0, v1 = 0;
  unsigned row0 = 0, row1 = 0, row2 = 0, row3 = 0;
  unsigned rowoff = 0;
  unsigned touch_a = 0, touch_b = 0;
  unsigned ra = 0, rb = 0, rc = 0, rlimit = 0, mpred = 0, mdd = 0;
  int mixw[NMIX];
  unsigned mixrow[NMIX];
  unsigned ssev[NSSE];
  unsigned ssecx[NSSE];
  unsigned gwc0 = 0, gwc1 = 0;
  int mixc0[NMIX], mixc1[NMIX];
  unsigned ssec0[NSSE], ssec1[NSSE];
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { mixw[k] = 0; mixrow[k] = 0; mixc0[k] = 0; mixc1[k] = 0; }
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { ssev[k] = 0; ssecx[k] = 0; ssec0[k] = 0; ssec1[k] = 0; }
  unsigned mixbase[NMIX], ssebase[NSSE];
#pragma unroll
  for (int k = 0; k < NMIX; ++k) mixbase[k] = dummy;
#pragma unroll
  for (int k = 0; k < NSSE; ++k) ssebase[k] = dummy;
  LaneK<N, NMIX, NSSE> lk;
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { lk.mixin[k] = 0; lk.mixst[k] = dummy; }
#pragma unroll
  for (int k = 0; k < NSSE; ++k) lk.ssest[k] = dummy;
  lk.lane0 = lane_mask(gl == 0);
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr CompK c = Chain::comp[i];
    lk.is[i] = 0;
    if constexpr (c.type == C_AVG || c.type == C_MIX2 || c.type == C_MIX || c.type == C_SSE || c.type == C_ISSE)
      lk.is[i] = lane_mask(gl == i);
    if constexpr (c.type == C_MIX) {
      static_assert(c.type != C_MIX || c.a2 + c.a3 <= 32, "MIX inputs must sit inside one 32-lane group");
      mixbase[c.slot] = abase + (unsigned)c.t0 + 4u * (unsigned)min(gl, (int)c.a3 - 1);
      ZPQ_OPAQUE(mixbase[c.slot]);
      lk.mixin[c.slot] = lane_mask(gl < (int)c.a3);
      lk.mixst[c.slot] = gl < (int)c.a3 ? abase + (unsigned)c.t0 + 4u * (unsigned)gl : dummy;
      ZPQ_OPAQUE(lk.mixst[c.slot]);
    } else if constexpr (c.type == C_SSE) {
      ssebase[c.slot] = abase + (unsigned)c.t0 + 4u * (unsigned)gl;
      ZPQ_OPAQUE(ssebase[c.slot]);
      lk.ssest[c.slot] = gl == 0 ? abase + (unsigned)c.t0 : dummy;
      ZPQ_OPAQUE(lk.ssest[c.slot]);
    }
  });
  unsigned rw = G32(goff);
  unsigned nspair = 0, dtv = 0;
  int sq = 0;
  unsigned ssetr[NSSE], ssedt[NSSE];
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { ssetr[k] = 0; ssedt[k] = 0; }
  unsigned hmix[NMIX], hsse[NSSE], hmix_n[NMIX], hsse_n[NSSE];     // uniform within a group
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { hmix[k] = 0; hmix_n[k] = 0; }
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { hsse[k] = 0; hsse_n[k] = 0; }
  int pdv[N];
#pragma unroll
  for (int k = 0; k < N; ++k) pdv[k] = 0;
  int ylast = 0;

  int c8 = 1, hmap4 = 1;                     // per block: uniform within a group, not across the wavefront
  unsigned low = 1, high = 0xFFFFFFFFu;
  unsigned steps = 0;
  int status = 0;

  auto g_index = [&](int c8x, int hm4x) __attribute__((always_inline)) -> unsigned {
    return (is_cm ? (h ^ (unsigned)hm4x) : (h + (unsigned)(c8x & (int)a5))) & gmask;
  };

  // ---------------------------------------------------------------- predict (B = bit position, compile time)
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
      const unsigned chk = (cx >> sizebits) & 255u;
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
    bh = row_get_nb<(B & 3)>(row0, row1, row2, row3, slot);
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
        if (gword && !pf_lane && !resident) gw = G32(goff + 4u * gidx);
      }
    } else {
      gw = G32(goff + 4u * gidx);
      static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
        constexpr CompK c = Chain::comp[decltype(ic)::value];
        if constexpr (c.type == C_MIX && mix_pf(c)) {
          const unsigned r = ((hmix[c.slot] + (unsigned)(c8 & 255)) & c.mask0) * c.a3;
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
        mixrow[c.slot] = ((hi + (unsigned)(c8 & (int)c.a5)) & c.mask0) * c.a3;
        if constexpr (mix_pf(c)) {
          mixc0[c.slot] = (int)G32(mixbase[c.slot] + 4u * (((hi + (unsigned)(c8a & 255)) & c.mask0) * c.a3));
          mixc1[c.slot] = (int)G32(mixbase[c.slot] + 4u * (((hi + (unsigned)(c8b & 255)) & c.mask0) * c.a3));
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
    DepD<Chain, 0>::predict(T, gl, gbase, lk, p, (int)v0, (int)v1, mixw, ssev, ssecx, ssetr, ssedt, pdv);
    sq = sp_squash(T, sp_clamp2k(p));
    return grlu((unsigned)sq, N - 1);
  };

  // ----------------------------------------------------------------- update
  auto update = [&](auto bitc, int y) __attribute__((always_inline)) {
    constexpr int B = decltype(bitc)::value;
    constexpr bool byte_done = B == 7;
    const int slot = hmap4 & 15;
    int pj, pdiff = 0;
    if constexpr (kIsseFast) pj = sp_shr1(p);
    else pj = __shfl(p, gbase + (int)(a2 & 31));
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr CompK c = Chain::comp[i];
      if constexpr (c.type == C_MIX2) pdiff |= (int)((unsigned)pdv[i] & lk.is[i]);
    });
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
    DepD<Chain, 0>::update(T, arena, gl, gbase, lk, y, sq, p, mixw, mixrow, ssev, ssecx, ssetr, ssedt);
    ylast = y;
  };

  // HCOMP of both blocks, SIMT: every lane of a group runs its block's program on its block's M/H/R
  // (identical accesses within the group coalesce); the two groups may take different branches.
  auto run_hcomp = [&](unsigned input) __attribute__((always_inline)) -> int {
#ifndef ZPQ_EMU
    return Chain::hcomp(input, vm_b, vm_c, vm_d, vm_f, vm_M, vm_H, vm_R);
#else
    int e = 0;
    if (gl == 0) e = Chain::hcomp(input, vm_b, vm_c, vm_d, vm_f, vm_M, vm_H, vm_R);
    return grl(e, 0);
#endif
  };
  auto refresh_contexts = [&](unsigned hv, unsigned (&hm)[NMIX], unsigned (&hs)[NSSE]) __attribute__((always_inline)) {
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr CompK c = Chain::comp[i];
      if constexpr (c.type == C_MIX) hm[c.slot] = grlu(hv, i);
      if constexpr (c.type == C_SSE) hs[c.slot] = grlu(hv, i);
    });
  };
  unsigned h_next = 0, ka0 = 0, ka1 = 0, ka2 = 0;
  auto run_ahead = [&](int ch) __attribute__((always_inline)) -> int {
    ZPQ_KEEP3(ka0, ka1, ka2);
    const int e = run_hcomp((unsigned)ch);
    h_next = vm_H[(unsigned)gl & Chain::HMASK];
    refresh_contexts(h_next, hmix_n, hsse_n);
    const unsigned cx = h_next + 16u;
    ka0 = G32(roff + ((cx * 16u) & (rmask - 15u)));
    ka1 = G32(goff + 4u * ((is_cm ? (h_next ^ 1u) : (h_next + (1u & a5))) & gmask));
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr CompK c = Chain::comp[decltype(ic)::value];
      if constexpr (c.type == C_MIX) {
        const unsigned r = ((hmix_n[c.slot] + (1u & c.a5)) & c.mask0) * c.a3;
        ka2 ^= G32(mixbase[c.slot] + 4u * r);
      } else if constexpr (c.type == C_SSE) {
        const unsigned cx0 = ((hsse_n[c.slot] + 1u) * 32u) & c.mask0;
        ka2 ^= G32(ssebase[c.slot] + 4u * cx0);
      }
    });
    return e;
  };

  // c8 / hmap4 bookkeeping (libzpaq.cpp:2055-2065); returns HCOMP's status at the end of a byte
  auto after_bit = [&](auto bitc, int y) __attribute__((always_inline)) -> int {
    constexpr int B = decltype(bitc)::value;
    update(bitc, y);
    c8 += c8 + y;
    int e = 0;
    if constexpr (B == 7) {
      if constexpr (DEC) {
        e = run_hcomp((unsigned)(c8 - 256));
        h = vm_H[(unsigned)gl & Chain::HMASK];
        refresh_contexts(h, hmix, hsse);
      } else {
        h = h_next;
#pragma unroll
        for (int k = 0; k < NMIX; ++k) hmix[k] = hmix_n[k];
#pragma unroll
        for (int k = 0; k < NSSE; ++k) hsse[k] = hsse_n[k];
      }
      hmap4 = 1;
      c8 = 1;
    } else if constexpr (B == 3) {
      hmap4 = (hmap4 & 0xf) << 5 | y << 4 | 1;
    } else {
      hmap4 = (hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + y) & 0xf);
    }
    return e;
  };

  bool fin = false;            // this block has delivered its result; it only keeps its partner company
  unsigned n = 0;
  if (!DEC) {
    auto encode = [&](int y, unsigned pr) __attribute__((always_inline)) {
      const unsigned mid = low + (unsigned)(((unsigned long long)(high - low) * pr) >> 16);
      if (y) high = mid; else low = mid + 1;
      while ((high ^ low) < 0x1000000u) {
        if (n < cap && gl == 0) out_ptr[n] = (unsigned char)(high >> 24);
        ++n;
        high = high << 8 | 255u;
        low = low << 8;
        low += (low == 0);
      }
    };
    auto finish = [&]() __attribute__((always_inline)) {
      if (!status) encode(1, 0);
      if (!status && n > cap) status = 3;
      if (gl == 0) { res[rslot].out_len = n; res[rslot].consumed = in_len; res[rslot].status = status; res[rslot].steps = steps; }
      fin = true;
      cap = 0;
    };
    const unsigned maxlen = max((unsigned)__shfl((int)in_len, 0), (unsigned)__shfl((int)in_len, 32));   // wave-uniform
    for (unsigned k = 0; k < maxlen; ++k) {
      if (!fin && (k >= in_len || status)) finish();
      const int ch = fin ? 0 : (int)in_ptr[k];
      const int e = run_ahead(ch);
      if (!fin && e) { status = e; finish(); }           // HCOMP failed: the block stops here, as the single-block kernel does
      encode(0, 0);
      static_for<0, 8>([&](auto bitc) __attribute__((always_inline)) {
        constexpr int B = decltype(bitc)::value;
        const unsigned pr = predict(bitc);
        const int y = (ch >> (7 - B)) & 1;
        encode(y, pr * 2 + 1);
        (void)after_bit(bitc, y);
        if (!fin && !status) ++steps;
      });
    }
    if (!fin) finish();
  } else {
    unsigned rp = 0, curr = 0;
    bool eos = false;
    for (int i = 0; i < 4; ++i) {
      if (rp >= in_len) { status = 6; break; }
      curr = curr << 8 | in_ptr[rp++];
    }
    auto decode = [&](unsigned pr) __attribute__((always_inline)) -> int {
      if (curr < low || curr > high) { status = 2; return 0; }
      const unsigned mid = low + (unsigned)(((unsigned long long)(high - low) * pr) >> 16);
      int y;
      if (curr <= mid) { y = 1; high = mid; } else { y = 0; low = mid + 1; }
      while ((high ^ low) < 0x1000000u) {
        high = high << 8 | 255u;
        low = low << 8;
        low += (low == 0);
        if (rp >= in_len) { status = 6; break; }
        curr = curr << 8 | in_ptr[rp++];
      }
      return y;
    };
    for (;;) {
      if (!fin && (status || eos || n >= cap)) {
        if (gl == 0) { res[rslot].out_len = n; res[rslot].consumed = eos ? rp : 0; res[rslot].status = status; res[rslot].steps = steps; }
        fin = true;
      }
      if ((__shfl((int)fin, 0) & __shfl((int)fin, 32)) != 0) break;   // both blocks done (wave-uniform)
      int ch = 1;
      bool idle = fin;                                                  // idle: keep in step, touch nothing of mine
      if (!idle) {
        const int flag = decode(0);
        if (status) idle = true;
        else if (flag) { eos = true; if (curr != 0) status = 2; idle = true; }
      }
      static_for<0, 8>([&](auto bitc) __attribute__((always_inline)) {
        const unsigned pr = predict(bitc) * 2 + 1;
        int y = 0;
        if (!idle) { y = decode(pr); if (status) idle = true; }
        if (!idle) { ch += ch + y; ++steps; }
        const int e = after_bit(bitc, y);
        if (!idle && e) { status = e; idle = true; }
      });
      if (!idle) {
        if (gl == 0) out_ptr[n] = (unsigned char)(ch - 256);
        ++n;
      }
    }
  }
}

// ---- compile-time walk over the dependent components, group-local cross-lane traffic -------------
template <class Chain, int I>
struct DepD {
  template <int NM, int NS>
  static __device__ __forceinline__ void predict(const SpecTables& T, int gl, int gbase,
                                                 const LaneK<Chain::N, NM, NS>& lk, int& p, int w0, int w1,
                                                 int (&mixw)[NM], unsigned (&ssev)[NS], unsigned (&ssecx)[NS],
                                                 unsigned (&ssetr)[NS], unsigned (&ssedt)[NS], int (&pdv)[Chain::N]) {
    if constexpr (I < Chain::N) {
      constexpr CompK c = Chain::comp[I];
      if constexpr (c.type == C_ISSE) {
        if constexpr (!isse_left_fed<Chain>()) {
          const int pj = __shfl(p, gbase + (int)c.a2);
          const int val = sp_clamp2k(sp_mad24(w0, pj, w1 * 64) >> 16);
          p = (int)sp_blend(lk.is[I], (unsigned)val, (unsigned)p);
        }
      } else if constexpr (c.type == C_AVG) {
        const int pj = __shfl(p, gbase + (int)c.a1), pk = __shfl(p, gbase + (int)c.a2);
        const int val = (pj * (int)c.a3 + pk * (256 - (int)c.a3)) >> 8;
        p = (int)sp_blend(lk.is[I], (unsigned)val, (unsigned)p);
      } else if constexpr (c.type == C_MIX2) {
        const int pj = __shfl(p, gbase + (int)c.a2), pk = __shfl(p, gbase + (int)c.a3);
        pdv[I] = pj - pk;
        const int val = sp_mad24(w0, pj, __mul24(65536 - w0, pk)) >> 16;
        p = (int)sp_blend(lk.is[I], (unsigned)val, (unsigned)p);
      } else if constexpr (c.type == C_MIX) {
        int pin = p;
        if constexpr (c.a2 != 0) pin = __shfl(p, gbase + ((gl + (int)c.a2) & 31));
        const int x = (int)((unsigned)__mul24(mixw[c.slot] >> 8, pin) & lk.mixin[c.slot]);
        const int val = sp_clamp2k(sp_group_sum<(int)c.a3>(x, gbase) >> 8);
        p = (int)sp_blend(lk.is[I], (unsigned)val, (unsigned)p);
      } else if constexpr (c.type == C_SSE) {
        int pq = __shfl(p, gbase + (int)c.a2) + 992;
        pq = min(max(pq, 0), 1983);
        const int wt = pq & 63;
        pq >>= 6;
        const unsigned e0 = (unsigned)__shfl((int)ssev[c.slot], gbase + pq);
        const unsigned e1 = (unsigned)__shfl((int)ssev[c.slot], gbase + pq + 1);
        const int val = sp_stretch(T, ((e0 >> 10) * (unsigned)(64 - wt) + (e1 >> 10) * (unsigned)wt) >> 13);
        p = (int)sp_blend(lk.is[I], (unsigned)val, (unsigned)p);
        ssecx[c.slot] += (unsigned)(pq + (wt >> 5));
        ssetr[c.slot] = (wt >> 5) ? e1 : e0;
        ssedt[c.slot] = (unsigned)T.dt[ssetr[c.slot] & 0x3ffu];
      }
      DepD<Chain, I + 1>::predict(T, gl, gbase, lk, p, w0, w1, mixw, ssev, ssecx, ssetr, ssedt, pdv);
    }
  }

  template <int NM, int NS>
  static __device__ __forceinline__ void update(const SpecTables& T, g_u8* arena, int gl, int gbase,
                                                const LaneK<Chain::N, NM, NS>& lk, int y, int sq, int p,
                                                int (&mixw)[NM], unsigned (&mixrow)[NM], unsigned (&ssev)[NS],
                                                unsigned (&ssecx)[NS], unsigned (&ssetr)[NS], unsigned (&ssedt)[NS]) {
    if constexpr (I < Chain::N) {
      constexpr CompK c = Chain::comp[I];
      if constexpr (c.type == C_MIX) {
        const int err = ((y * 32767 - __shfl(sq, gbase + I)) * (int)c.a4) >> 4;
        int pin = p;
        if constexpr (c.a2 != 0) pin = __shfl(p, gbase + ((gl + (int)c.a2) & 31));
        const int w = sp_clamp512k(mixw[c.slot] + (sp_mad24(err, pin, 1 << 12) >> 13));
        const unsigned wo = lk.mixst[c.slot] + ((4u * mixrow[c.slot]) & lk.mixin[c.slot]);
        *(g_i32*)(arena + wo) = w;
      } else if constexpr (c.type == C_SSE) {
        const unsigned e = ssecx[c.slot];
        const unsigned v = ssetr[c.slot];
        const unsigned count = v & 0x3ffu;
        const int err = y * 32767 - (int)(v >> 17);
        const unsigned prod = (unsigned)__mul24(err, (int)ssedt[c.slot]);
        const unsigned nv = v + (prod & 0xFFFFFC00u) + (count < c.limit ? 1u : 0u);
        *(g_u32*)(arena + lk.ssest[c.slot] + ((4u * (e & c.mask0)) & lk.lane0)) = nv;
      }
      DepD<Chain, I + 1>::update(T, arena, gl, gbase, lk, y, sq, p, mixw, mixrow, ssev, ssecx, ssetr, ssedt);
    }
  }
};

}  // namespace zpq
