// Decoder with the blocks of a workgroup in LOCKSTEP and the work of a bit split by KIND over its wavefronts:
//
//   ROW wavefronts    lane = (block, ICM / ISSE component): 16 lanes per block (4 blocks per wavefront; 32 lanes and
//                     2 blocks when a chain has more than 16 such components).  Predictor::find per nibble, the
//                     bit-history row in registers, the side table, stretch, the ISSE chains (DPP shift inside the
//                     block's lanes) -- what spec_kernel.h does on 16 of its 64 lanes, here on all 64.
//   MIXER wavefronts  two blocks per wavefront as in spec_dual_kernel.h, lane = (block, component): CM, MATCH, MIX2,
//                     MIX, SSE, AVG, the arithmetic coder and HCOMP; the predictions of the ICM / ISSE components
//                     arrive through LDS.
//
// Why.  One block per wavefront leaves 41 of 64 lanes idle and issues every instruction for all of them; two blocks
// per wavefront fix half of that but leave ONE wavefront per SIMD with nothing to hide its stalls behind
// (profiles/r03: 7.9 cycles per instruction, +8 %).  The decoder's true serial chain per bit -- ISSE chain, mixers,
// SSE, squash, coder -- is ~1 000 cycles; everything else (row probes, side-table and weight updates, candidate
// fetches) only has to be done by SOMEBODY before the next bit needs it.  Here 8 blocks share a workgroup of 6 (8)
// wavefronts, a bit is two phases separated by workgroup barriers
//
//     rows predict -> [A] -> mixers: chain, squash, decode y -> [B] -> rows update | mixers update    (+ [C] per byte: HCOMP)
//
// and while the mixers work the row wavefronts fetch what the next bit may need (both candidates), and vice versa.
// Per bit the CU issues ~2 x 200 + 4 x 250 instructions for 8 blocks instead of 8 x 415.
//
// Chains this kernel takes (the generator checks, the engine falls back to the other decoders otherwise): up to 32
// components, every ISSE fed by the ICM / ISSE before it (the last one of lower index: true of every chain
// compressBlock's methods and the legacy models produce), H in LDS (hh <= 10), MIX inputs inside one half, the 8
// arenas of a workgroup inside a 4 GiB window (the engine lays a batch's arenas out back to back), one segment per block.
//
// The per-lane model arithmetic is spec_kernel.h's, statement for statement (bit-exact with Predictor::predict0 /
// update0, libzpaq.cpp:1854-2066; Decoder::decompress / decode, 2104-2181).  LDS plan: the generated Chain of the
// 8-blocks-per-workgroup shape; the upper half of a block's 512-byte dummy area carries the exchange words.
#pragma once
#ifndef ZPQ_LANE_VM
#define ZPQ_LANE_VM 1
#endif
#include "spec_dual_kernel.h"

namespace zpq {

// Workgroup barrier that waits for this wavefront's LDS traffic only: __syncthreads() also drains vmcnt, i.e. every
// global fetch and store in flight -- exactly what the phases issue early in order NOT to wait for.
#ifdef ZPQ_EMU
#define ZPQ_TEAM_BARRIER() emu::block_barrier()
#else
// (the scheduling fences keep the compiler from moving a phase's arithmetic ahead of the barrier that hands the
// other wavefronts their input: asm volatile orders memory operations only)
#define ZPQ_TEAM_BARRIER()                                              \
  do {                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                  \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                                  \
  } while (0)
#endif

// The workgroup's shared tables: stretch in the encoder's compact form (groups of 8 as start value + seven 1-bit increments
// for x in [16384, 32512), the steep top end direct, the lower half by stretch(x) = -stretch(32767 - x): exact, checked
// exhaustively on the host), the non-trivial part of squash, dt, dt2k, the state table.
typedef __attribute__((address_space(3))) unsigned short lds_u16;

struct TeamTables {
  unsigned stretch_cb[2016];
  short stretch_top[256];
  uint16_t squash_mid[1344];
  int32_t dt[1024];
  uint16_t dt2k[256];
  uint8_t ns[1024];
};
static_assert(sizeof(TeamTables) == kTeamTablesBytes, "host codegen and device disagree on the lockstep decoder's LDS tables");
__device__ __forceinline__ int sp_stretch(const TeamTables& T, unsigned x) {   // x in 0..32767
  const bool lo = x < 16384u;
  const unsigned y = lo ? 32767u - x : x;
  const unsigned e = T.stretch_cb[min((y - 16384u) >> 3, 2015u)];
  const int mid = (int)(short)(unsigned short)e + __builtin_popcount((e >> 16) & ((1u << (y & 7u)) - 1u));
  const int hi = T.stretch_top[y >= 32512u ? y - 32512u : 0u];
  const int v = y >= 32512u ? hi : mid;
  return lo ? -v : v;
}
__device__ __forceinline__ int sp_squash(const TeamTables& T, int p) {         // p in -2048..2047
  const int i = p + 2048 - 1376;
  const int v = T.squash_mid[min(max(i, 0), 1343)];
  return i < 0 ? 0 : (i > 1343 ? 32767 : v);
}

// -DZPQ_PROF: wavefront 0 of the row kind and of the mixer kind of workgroup 0 count the cycles of their phases
// (s_memtime) and print them per coded bit at the end; compiled out otherwise.
// ZPQ_TEAM_EARLY2: the second nibble's three candidate rows are LOADED for both values of bit 3 while that bit is decoded
// (24 registers per lane) instead of only pulled towards the L2
#ifndef ZPQ_TEAM_EARLY2
#define ZPQ_TEAM_EARLY2 1
#endif
// ZPQ_TEAM_LATE_UPDATE7: the mixers train a byte's last bit after HCOMP and [C] instead of before
#ifndef ZPQ_TEAM_LATE_UPDATE7
#define ZPQ_TEAM_LATE_UPDATE7 1
#endif
// ZPQ_TEAM_LATE_STORES: the row wavefronts' global stores (the row that is left, the entry trained in a side table that
// stayed in the arena) are issued behind [A] instead of in front of the next prediction's dependent loads -- vmcnt counts
// in order, so a load issued after a store is not there before the store has been acknowledged
#ifndef ZPQ_TEAM_LATE_STORES
#define ZPQ_TEAM_LATE_STORES 1
#endif
#if defined(ZPQ_PROF) && !defined(ZPQ_EMU)
#define TEAM_PROF_DECL unsigned long long tp_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tp_t_ = __builtin_readcyclecounter(), tp_n_ = 0;
#define TEAM_PROF(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); tp_[k] += n_ - tp_t_; tp_t_ = n_; } while (0)
#define TEAM_PROF_BIT() (++tp_n_)
#ifdef ZPQ_PROF2       // (profile build only) every memory operation of the wavefront has landed; then the phase counter
#define TEAM_PROF_VM(k) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TEAM_PROF(k); } while (0)
#else
#define TEAM_PROF_VM(k) do {} while (0)
#endif
#else
#define TEAM_PROF_DECL
#define TEAM_PROF(k) do {} while (0)
#define TEAM_PROF_BIT() do {} while (0)
#define TEAM_PROF_VM(k) do {} while (0)
#endif

// Rows a bit's tables select are fetched one bit ahead (both candidates).  -DZPQ_TOUCH2=1 (off by default: measured -2 % on the MI355X, profiles/r04): the four rows the
// bit after that may select are touched as well -- loads whose data nobody reads -- so that the candidate fetch a bit later
// finds its line in the cache instead of paying the full trip to HBM (page walk included) inside one bit's time.
#ifndef ZPQ_TOUCH2
#define ZPQ_TOUCH2 0
#endif
constexpr bool team_far_table(unsigned long long bytes) { return bytes > (256u << 10); }   // tables the caches do not hold

// Where a bit-history row lives in THIS decoder's hash tables.  Predictor::find's three candidates of a context are the rows
// h0, h0 ^ 16, h0 ^ 32 of one 64-byte line (libzpaq.cpp:2072-2088); which line is the decoder's own business as long as it is
// a bijection on lines (the tables start zeroed and nobody else reads them: the predictions depend on the rows' contents
// only).  The second nibble of a byte is looked up under c8 = 16 + high nibble, and the lockstep decoder fetches the lines
// of BOTH values of the nibble's last bit while that bit is decoded: 256 bytes apart in the reference's layout -- two
// random lines.  With address bits 6 and 8 exchanged they are the two halves of ONE aligned 128-byte line: one DRAM row
// activation instead of two for 18 of the ~100 lines a decoded byte asks for (profiles/r05: FETCH_SIZE of the decoder).
__device__ __forceinline__ unsigned team_row_line(unsigned h0, unsigned rmask) {
  const unsigned t = ((h0 >> 6) ^ (h0 >> 8)) & 1u;
  return rmask >= 511u ? h0 ^ (t << 6 | t << 8) : h0;
}

template <int N>
struct TeamMap {
  int nrows;
  int rc[32];        // chain index of the k-th ICM / ISSE component
  int depth;         // longest run of ISSEs among them: rounds of (shift, multiply-add, clamp) that resolve every chain
  bool ok;
};

template <class Chain>
constexpr TeamMap<Chain::N> team_map() {
  TeamMap<Chain::N> m{};
  m.nrows = 0;
  m.depth = 0;
  m.ok = Chain::N <= 32 && Chain::H_LDS >= 0;
  for (int i = 0; i < 32; ++i) m.rc[i] = 0;
  int run = 0;
  for (int i = 0; i < Chain::N; ++i) {
    const unsigned t = Chain::comp[i].type;
    if (t == C_ICM || t == C_ISSE) {
      if (m.nrows < 32) m.rc[m.nrows] = i;
      ++m.nrows;
      run = t == C_ISSE ? run + 1 : 0;
      if (run > m.depth) m.depth = run;
    }
    // an ISSE takes its input from the row component before it (its left neighbour among the row lanes)
    if (t == C_ISSE && (m.nrows < 2 || m.nrows > 32 || Chain::comp[i].a2 != (unsigned)m.rc[m.nrows - 2])) m.ok = false;
  }
  return m;
}

constexpr int kTeamBlocks = 8;                               // blocks per workgroup
// exchange words in the upper half of a block's dummy area (offsets from the start of the block's LDS region)
constexpr int kTeamX = team_block_lds_bytes() - 256;         // X[32]: stretch-domain predictions of the row components, by chain index
constexpr int kTeamY = kTeamX + 128;                         // the decoded bit
constexpr int kTeamRun = kTeamX + 132;                       // block still decoding (written once per byte)

// lanes per block in a row wavefront: 16 (4 blocks per wavefront) when the chain's ICM / ISSE components fit, else 32.
// (Measured and dropped, profiles/r04: 32 lanes for every chain -- eight wavefronts, a row and a mixer wavefront on every
// SIMD -- is 2.5 % slower; a higher priority for the mixer wavefronts changes nothing.)
template <class Chain>
constexpr int team_row_lanes() { return team_map<Chain>().nrows <= 16 ? 16 : 32; }
template <class Chain>
constexpr bool team_tail_ok();            // (below: chains whose scalar work moves to ONE tail wavefront)
template <class Chain>
constexpr int team_threads() { return 64 * (kTeamBlocks / (64 / team_row_lanes<Chain>()) + kTeamBlocks / 2 + (team_tail_ok<Chain>() ? 1 : 0)); }

// =====================================================================================================================
// ROW wavefront
template <class Chain, class TT>
__device__ __forceinline__ void team_rows(const TT& T, lds_u8* const lds0, const BlockJob* jobs, unsigned nblocks, int wave, int lane) {
  constexpr auto TM = team_map<Chain>();
  constexpr int R = TM.nrows;
  constexpr int RL = team_row_lanes<Chain>();
  constexpr int BPR = 64 / RL;
  constexpr int kRegion = team_block_lds_bytes();
  const int k = lane % RL, q = lane / RL;
  const unsigned bw = (unsigned)(wave * BPR + q);            // block of this lane inside the workgroup
  const unsigned wg0 = blockIdx.x * (unsigned)kTeamBlocks;
  const unsigned b = wg0 + bw;
  const bool live = b < nblocks && k < R;
  const BlockJob job0 = jobs[wg0];
  const BlockJob job = jobs[b < nblocks ? b : wg0];
  g_u8* const arena = (g_u8*)sp_uni64((unsigned long long)job0.arena);
  const unsigned long long delta64 = (unsigned long long)job.arena - (unsigned long long)job0.arena;
  const unsigned hoff = live ? (unsigned)delta64 : 0u;
  lds_u8* const wl = lds0 + bw * (unsigned)kRegion;

  const unsigned dummy = (unsigned)Chain::OFF_RUN;
  const unsigned dummy_lds = (unsigned)(kRegion - 512) + (unsigned)(k & 31) * 8u;
  unsigned mask1 = 63, sizebits = 0, off0 = dummy, off1 = dummy, ctype = 0;
  int ldsoff = -1, cidx = 0;
  static_for<0, (R < 32 ? R : 32)>([&](auto kc) __attribute__((always_inline)) {
    constexpr int kk = decltype(kc)::value;
    if (k == kk && live) {
      constexpr CompK c = Chain::comp[TM.rc[kk]];
      ctype = c.type;
      sizebits = c.a1 + 2;
      off0 = (unsigned)c.t0 + hoff;
      off1 = (unsigned)c.t1 + hoff;
      mask1 = c.mask1;
      ldsoff = c.lds;
      cidx = TM.rc[kk];
    }
  });
  auto G32 = [&](unsigned off) __attribute__((always_inline)) -> g_u32& { return *(g_u32*)(arena + off); };
  auto G128 = [&](unsigned off) __attribute__((always_inline)) -> g_u128& { return *(g_u128*)(arena + off); };
  auto L32 = [&](unsigned off) __attribute__((always_inline)) -> lds_u32& { return *(lds_u32*)(wl + off); };

  const bool is_icm = ctype == C_ICM, is_isse = ctype == C_ISSE;
  const bool has_row = is_icm || is_isse;
  const unsigned rmask = has_row ? mask1 : 63u;
  const unsigned roff = has_row ? off1 : dummy;
  const unsigned ldsq = (has_row && ldsoff >= 0) ? (unsigned)ldsoff : dummy_lds;
  const bool side_global = has_row && ldsoff < 0;
  const unsigned soff = side_global ? off0 : dummy;
  // the ISSE's second word: one store at a per-lane offset (written as a choice between two addresses it compiles to a
  // ladder of branches on the path every bit takes)
  unsigned s1base = (side_global && is_isse) ? soff + 4u : dummy + 4u, s1mask = (side_global && is_isse) ? ~0u : 0u;
  ZPQ_OPAQUE(s1base);
  ZPQ_OPAQUE(s1mask);
  auto lane_mask = [&](bool x) __attribute__((always_inline)) -> unsigned {
    unsigned m = x ? 0xFFFFFFFFu : 0u;
    ZPQ_OPAQUE(m);
    return m;
  };
  const unsigned m_isse = lane_mask(is_isse), m_icm = lane_mask(is_icm), m_row = lane_mask(has_row);
  const unsigned bh_shift = is_isse ? 1u : 0u;
  // packed side table in LDS (layout.h kTeamIcmLds / kTeamIsseLds): 16-bit array(s) at the table's start -- ICM: cm's low half,
  // 2 bytes per entry; ISSE: the two weights' low halves, 4 bytes per entry --, then one byte per entry with the bits above.
  // Lanes without such a table (idle, or the table stayed in the arena) work on their 8-byte dummy slot.
  const bool in_lds = has_row && ldsoff >= 0;
  const unsigned pk_sh = is_isse ? 2u : 1u;
  const unsigned pk_hi = in_lds ? ldsq + (is_isse ? 1024u : 512u) : dummy_lds + 4u;
  const unsigned m_inlds = lane_mask(in_lds), m_lds2 = lane_mask(is_isse && in_lds);
  auto side_lds_get = [&](unsigned bhv, unsigned& q0, unsigned& q1) __attribute__((always_inline)) {
    const unsigned i = bhv & m_inlds;
    const unsigned a1 = ldsq + (i << pk_sh);
    const unsigned lo = *(const lds_u16*)(wl + a1), hi16 = *(const lds_u16*)(wl + a1 + 2u), h8 = *(const lds_u8*)(wl + pk_hi + i);
    const unsigned cm = lo | h8 << 16;
    const unsigned w0 = (unsigned)((int)((lo | (h8 & 15u) << 16) << 12) >> 12);
    const unsigned w1 = (unsigned)((int)((hi16 | (h8 >> 4) << 16) << 12) >> 12);
    q0 = sp_blend(m_icm, cm, w0);
    q1 = w1;
  };
  auto side_lds_put = [&](unsigned bhv, unsigned n0, unsigned n1) __attribute__((always_inline)) {
    const unsigned i = bhv & m_inlds;
    const unsigned a1 = ldsq + (i << pk_sh);
    const unsigned a2 = sp_blend(m_lds2, a1 + 2u, dummy_lds + 2u);
    *(lds_u16*)(wl + a1) = (unsigned short)n0;
    *(lds_u16*)(wl + a2) = (unsigned short)n1;
    *(lds_u8*)(wl + pk_hi + i) = (unsigned char)sp_blend(m_icm, n0 >> 16, ((n0 >> 16) & 15u) | ((n1 >> 16) & 15u) << 4);
  };
  const unsigned xoff = live ? (unsigned)kTeamX + 4u * (unsigned)cidx : dummy_lds;      // where this lane publishes p
  constexpr int kIsseDepth = TM.depth;

  unsigned bh = 0, h = 0, v0 = 0, v1 = 0, nspair = 0;
  int p = 0, sq = 0;
  unsigned row0 = 0, row1 = 0, row2 = 0, row3 = 0, rowoff = 0;
  unsigned touch_a = 0, touch_b = 0;
  uint4 wb = make_uint4(0, 0, 0, 0);
  unsigned wboff = 0;
#if ZPQ_TEAM_EARLY2
  uint4 ea0 = make_uint4(0, 0, 0, 0), ea1 = ea0, ea2 = ea0, eb0 = ea0, eb1 = ea0, eb2 = ea0;
#endif
  // side tables that stayed in the arena: both candidates of the next bit are fetched while the mixers work
  unsigned sca0 = 0, sca1 = 0, scb0 = 0, scb1 = 0;
  unsigned le0 = 0xFFFFFFFFu, ln0 = 0, ln1 = 0;
  int c8 = 1, hmap4 = 1, ylast = 0;
  TEAM_PROF_DECL

  bool any = true;
  {
    ZPQ_TEAM_BARRIER();                                      // [S] the mixers have published who runs
    unsigned r = 0;
    for (int i = 0; i < kTeamBlocks; ++i) r |= *(lds_u32*)(lds0 + (unsigned)(i * kRegion + kTeamRun));
    any = r != 0;
  }
  while (any) {
    static_for<0, 8>([&](auto bitc) __attribute__((always_inline)) {
      constexpr int B = decltype(bitc)::value;
      constexpr bool nib = B == 0 || B == 4;
      constexpr bool last_of_nibble = B == 3;
      const int slot = hmap4 & 15;
      const int c8a = c8 * 2, c8b = c8 * 2 + 1;
      const int hm4a = last_of_nibble ? ((hmap4 & 0xf) << 5 | 1) : ((hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2) & 0xf));
      const int hm4b = last_of_nibble ? ((hmap4 & 0xf) << 5 | 1 << 4 | 1)
                                      : ((hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + 1) & 0xf));
      // ---- predict: Predictor::find per nibble, bit history, side table, ISSE chains
      if constexpr (nib) {
        ZPQ_KEEP2(touch_a, touch_b);
        TEAM_PROF_VM((B == 0 ? 8 : 11));                      // what is still in flight from before (B = 4: the early rows)
        const unsigned cx = h + 16u * (unsigned)c8;
        const unsigned chk = (cx >> (sizebits & 31u)) & 255u;
        const unsigned h0 = team_row_line((cx * 16u) & (rmask - 15u), rmask);
#if ZPQ_TEAM_EARLY2
        uint4 r0, r1, r2;
        if constexpr (B == 4) {                               // both candidate lines have been on their way since bit 3's [A]
          r0 = ylast ? eb0 : ea0;
          r1 = ylast ? eb1 : ea1;
          r2 = ylast ? eb2 : ea2;
        } else {
          r0 = G128(roff + h0);
          r1 = G128(roff + (h0 ^ 16u));
          r2 = G128(roff + (h0 ^ 32u));
        }
#else
        uint4 r0 = G128(roff + h0);
        uint4 r1 = G128(roff + (h0 ^ 16u));
        uint4 r2 = G128(roff + (h0 ^ 32u));
#endif
        TEAM_PROF_VM((B == 0 ? 9 : 12));                      // (B = 0: the rows)
        const uint4 oldrow = make_uint4(row0, row1, row2, row3);
#if ZPQ_TEAM_LATE_STORES
        wb = oldrow;
        wboff = rowoff;
#else
        G128(roff + rowoff) = oldrow;
#endif
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
      }
      bh = row_get_nb<(B & 3)>(row0, row1, row2, row3, slot);
      nspair = *(const unsigned short*)&T.ns[(bh & 255u) * 4u];
      const unsigned e0 = (bh << bh_shift) & m_row;
      unsigned q0, q1;
      side_lds_get(bh, q0, q1);
      if constexpr (Chain::ANY_GLOBAL_SIDE) {
        const unsigned sidx = side_global ? e0 : 0u;
        unsigned g0, g1;
        if constexpr (nib) {                                  // new row: nothing was fetched ahead
          g0 = G32(soff + 4u * sidx);
          g1 = G32(soff + 4u * sidx + 4u);
#if ZPQ_TEAM_LATE_STORES
          const bool fwd = sidx == le0;                       // (the entry trained a bit ago is not in memory yet)
          g0 = fwd ? ln0 : g0;
          g1 = fwd ? ln1 : g1;
#endif
        } else {
          const bool fwd = sidx == le0;
          g0 = fwd ? ln0 : (ylast ? scb0 : sca0);
          g1 = fwd ? ln1 : (ylast ? scb1 : sca1);
        }
        q0 = side_global ? g0 : q0;
        q1 = side_global ? g1 : q1;
        if constexpr (nib) TEAM_PROF_VM((B == 0 ? 10 : 13));  // the entries of the side tables that stayed in the arena
      }
      v0 = q0;
      v1 = q1;
      {
        const int st = sp_stretch(T, (v0 >> 8) & 32767u);
        p = (int)((unsigned)st & m_icm);
        const int iw = (int)(v0 & m_isse);
        const int ia = (int)sp_blend(m_isse, v1 << 6, (unsigned)p << 16);
#pragma unroll
        for (int it = 0; it < kIsseDepth; ++it) p = sp_clamp2k(sp_mad24(iw, sp_shr1(p), ia) >> 16);
      }
      L32(xoff) = (unsigned)p;
      TEAM_PROF((B == 0 ? 5 : (B == 4 ? 6 : 0)));
      ZPQ_TEAM_BARRIER();                                    // [A] the mixers take over
      TEAM_PROF(1);
      // ---- while the mixers work: what the update and the next bit will need
      sq = sp_squash(T, sp_clamp2k(p));
      const int pj = sp_shr1(p);
#if ZPQ_TEAM_LATE_STORES
      if constexpr (nib) G128(roff + wboff) = wb;
      if constexpr (Chain::ANY_GLOBAL_SIDE) {
        const bool pend = le0 != 0xFFFFFFFFu;
        unsigned so0 = pend ? soff + 4u * le0 : dummy, so1 = pend ? s1base + 4u * (le0 & s1mask) : dummy + 4u;
        ZPQ_OPAQUE(so0);
        ZPQ_OPAQUE(so1);
        G32(so0) = ln0;
        G32(so1) = ln1;
      }
#endif
      if constexpr (last_of_nibble) {
        // the second nibble's row will be one of two lines: pull both towards this XCD's L2 now
        const unsigned cxa = h + 16u * (unsigned)c8a, cxb = h + 16u * (unsigned)c8b;
#if ZPQ_TEAM_EARLY2
        const unsigned ha = team_row_line((cxa * 16u) & (rmask - 15u), rmask), hb = team_row_line((cxb * 16u) & (rmask - 15u), rmask);
        ea0 = G128(roff + ha); ea1 = G128(roff + (ha ^ 16u)); ea2 = G128(roff + (ha ^ 32u));
        eb0 = G128(roff + hb); eb1 = G128(roff + (hb ^ 16u)); eb2 = G128(roff + (hb ^ 32u));
#else
        touch_a = G32(roff + team_row_line((cxa * 16u) & (rmask - 15u), rmask));
        touch_b = G32(roff + team_row_line((cxb * 16u) & (rmask - 15u), rmask));
#endif
      }
      if constexpr (Chain::ANY_GLOBAL_SIDE && B != 3 && B != 7) {
        const unsigned bha = row_get(row0, row1, row2, row3, hm4a & 15), bhb = row_get(row0, row1, row2, row3, hm4b & 15);
        const unsigned ea = side_global ? (bha << bh_shift) : 0u, eb = side_global ? (bhb << bh_shift) : 0u;
        sca0 = G32(soff + 4u * ea); sca1 = G32(soff + 4u * ea + 4u);
        scb0 = G32(soff + 4u * eb); scb1 = G32(soff + 4u * eb + 4u);
      }
      TEAM_PROF(2);
      if constexpr (team_tail_ok<Chain>()) ZPQ_TEAM_BARRIER();   // [A2] the mix wavefronts hand the MIX outputs to the tail (nothing to do here)
      ZPQ_TEAM_BARRIER();                                    // [B] the bit is known
      TEAM_PROF(3);
      TEAM_PROF_BIT();
      const int y = (int)L32((unsigned)kTeamY);
      // ---- update (Predictor::update0 cases ICM, ISSE)
      {
        const unsigned nsv = y ? nspair >> 8 : nspair & 255u;
        const int yq = y * 32767;
        const int err = yq - sq;
        row_set_nb<(B & 3)>(row0, row1, row2, row3, slot, nsv);
        const unsigned n0 = sp_blend(m_icm, v0 + (unsigned)((int)((unsigned)yq - (v0 >> 8)) >> 2),
                                  (unsigned)sp_clamp512k((int)v0 + (sp_mad24(err, pj, 1 << 12) >> 13)));
        const unsigned n1 = (unsigned)sp_clamp512k((int)v1 + ((err + 16) >> 5));
        side_lds_put(bh, n0, n1);
        if constexpr (Chain::ANY_GLOBAL_SIDE) {
          const unsigned sidx = side_global ? e0 : 0u;
#if !ZPQ_TEAM_LATE_STORES
          G32(soff + 4u * sidx) = n0;
          G32(s1base + 4u * (sidx & s1mask)) = n1;
#endif
          le0 = sidx; ln0 = n0; ln1 = is_isse ? n1 : v1;
        }
        ylast = y;
      }
      c8 += c8 + y;
      if constexpr (B == 7) {
        TEAM_PROF(0);
        ZPQ_TEAM_BARRIER();                                  // [C] HCOMP has run: contexts of the next byte, who still runs
        TEAM_PROF(4);
        h = ((lds_u32*)(wl + Chain::H_LDS))[(unsigned)cidx & Chain::HMASK];
        unsigned r = 0;
        for (int i = 0; i < kTeamBlocks; ++i) r |= *(lds_u32*)(lds0 + (unsigned)(i * kRegion + kTeamRun));
        any = r != 0;
        hmap4 = 1;
        c8 = 1;
      } else if constexpr (B == 3) {
        hmap4 = (hmap4 & 0xf) << 5 | y << 4 | 1;
      } else {
        hmap4 = (hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + y) & 0xf);
      }
    });
  }
  // (the last nibble's row is never written back: the block's model state is of no use after its last byte)
#if defined(ZPQ_PROF) && !defined(ZPQ_EMU)
  if (blockIdx.x == 0 && wave == 0 && lane == 0 && tp_n_)
    printf("[zpq team prof] rows   : bits=%llu cycles: update+predict per bit inside a nibble=%.0f, per first bit of a byte=%.0f, per first bit of the "
           "second nibble=%.0f; per bit: wait[A]=%.0f idle-window work=%.0f wait[B]=%.0f wait[C]/8=%.0f\n",
           tp_n_, (double)tp_[0] / (tp_n_ * 0.75), (double)tp_[5] / (tp_n_ * 0.125), (double)tp_[6] / (tp_n_ * 0.125), (double)tp_[1] / tp_n_,
           (double)tp_[2] / tp_n_, (double)tp_[3] / tp_n_, (double)tp_[4] / tp_n_);
#ifdef ZPQ_PROF2
  if (blockIdx.x == 0 && wave == 0 && lane == 0 && tp_n_)
    printf("[zpq team prof] rows, a byte's first bit: in flight from before=%.0f rows=%.0f side entries in the arena=%.0f rest=%.0f; second nibble's "
           "first bit: update + early rows=%.0f (rows)=%.0f side entries=%.0f rest=%.0f\n",
           (double)tp_[8] / (tp_n_ * 0.125), (double)tp_[9] / (tp_n_ * 0.125), (double)tp_[10] / (tp_n_ * 0.125), (double)tp_[5] / (tp_n_ * 0.125),
           (double)tp_[11] / (tp_n_ * 0.125), (double)tp_[12] / (tp_n_ * 0.125), (double)tp_[13] / (tp_n_ * 0.125), (double)tp_[6] / (tp_n_ * 0.125));
#endif
#endif
}

// =====================================================================================================================
// MIXER wavefront: two blocks per wavefront, lane = (block, component) as in spec_dual_kernel.h, without the ICM / ISSE
// work -- and with every DEPENDENT component (AVG, MIX2, MIX, SSE) computed redundantly by all 32 lanes of its half:
// a MIX's dot product ends in a half-wide sum anyway, and what follows it (MIX2, SSE, the final MIX2, squash, the coder)
// then runs on values every lane of the half already has, with no broadcast (two v_readlane + moves + select each) between
// its stages.  MIX2 weights are fetched by all lanes of the half from one address (one transaction), both candidates of the
// next bit like the MIX / SSE rows.
constexpr bool team_dep_type(unsigned t) { return t == C_AVG || t == C_MIX2 || t == C_MIX || t == C_SSE; }
template <class Chain>
constexpr int team_mix2_slot(int i) {
  int s = 0;
  for (int k = 0; k < i; ++k) s += Chain::comp[k].type == C_MIX2;
  return s;
}
constexpr bool mix2_pf(const CompK& c) { return c.a5 == 255u && c.mask0 >= 255u; }

typedef unsigned long long __attribute__((aligned(1))) team_u64u;
typedef __attribute__((address_space(1))) const team_u64u g_u64u;

template <class Chain, class TT>
__device__ __forceinline__ void team_mixers(const TT& T, lds_u8* const lds0, const BlockJob* jobs, BlockResult* res, unsigned nblocks,
                                            int tw, int lane) {
  constexpr int N = Chain::N;
  constexpr int NMIX = Chain::NMIX > 0 ? Chain::NMIX : 1;
  constexpr int NSSE = Chain::NSSE > 0 ? Chain::NSSE : 1;
  constexpr int NMIX2 = team_mix2_slot<Chain>(N) > 0 ? team_mix2_slot<Chain>(N) : 1;
  constexpr int kRegion = team_block_lds_bytes();
  const int ci = lane & 31;
  const bool upper = lane >= 32;
  const unsigned wg0 = blockIdx.x * (unsigned)kTeamBlocks;
  const unsigned bw = (unsigned)tw * 2u + (upper ? 1u : 0u);
  const unsigned b = wg0 + bw;
  const bool live = b < nblocks;
  const BlockJob job0 = jobs[wg0];
  const BlockJob job = jobs[live ? b : wg0];
  g_u8* const arena = (g_u8*)sp_uni64((unsigned long long)job0.arena);
  const unsigned long long delta64 = (unsigned long long)job.arena - (unsigned long long)job0.arena;
  const unsigned hoff = live ? (unsigned)delta64 : 0u;
  const g_u8* const in_ptr = (const g_u8*)job.in;
  g_u8* const out_ptr = (g_u8*)job.out;
  const unsigned in_len = job.in_len, out_cap = job.out_cap, rslot = job.res_slot;
  lds_u8* const wl = lds0 + bw * (unsigned)kRegion;

  const unsigned dummy = (unsigned)Chain::OFF_RUN;
  unsigned limit = 0, mask0 = 0, mask1 = 63;
  unsigned off0 = dummy, off1 = dummy;
  unsigned ctype = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (ci == i && live) {
      const CompK c = Chain::comp[i];
      ctype = c.type;
      limit = c.limit; mask0 = c.mask0;
      if (c.type == C_CM || c.type == C_MATCH) {
        off0 = (unsigned)c.t0 + hoff;
        if (c.type == C_MATCH) { off1 = (unsigned)c.t1 + hoff; mask1 = c.mask1; }
      }
    }
  }
  auto G32 = [&](unsigned off) __attribute__((always_inline)) -> g_u32& { return *(g_u32*)(arena + off); };
  auto G8 = [&](unsigned off) __attribute__((always_inline)) -> g_u8& { return *(g_u8*)(arena + off); };
  auto L32 = [&](unsigned off) __attribute__((always_inline)) -> lds_u32& { return *(lds_u32*)(wl + off); };

  // HCOMP machine of this half (every lane of the half runs it: identical values, identical stores)
  unsigned vm_b = 0, vm_c = 0, vm_d = 0, vm_f = 0;
  g_u8* const vm_M = arena + (unsigned)Chain::OFF_M + hoff;
  g_u32* const vm_R = (g_u32*)(arena + (unsigned)Chain::OFF_R + hoff);
  lds_u32* const vm_H = (lds_u32*)(wl + Chain::H_LDS);

  const bool is_cm = ctype == C_CM, is_match = ctype == C_MATCH;
  const bool is_rowc = ctype == C_ICM || ctype == C_ISSE;     // predicted by the row wavefronts
  const bool is_ctx = is_cm || is_match;
  const bool pf_lane = is_cm && mask0 >= 511u;
  const unsigned goff = is_cm ? off0 : dummy;
  const unsigned gmask = is_cm ? mask0 : 0u;
  auto lane_mask = [&](bool x) __attribute__((always_inline)) -> unsigned {
    unsigned m = x ? 0xFFFFFFFFu : 0u;
    ZPQ_OPAQUE(m);
    return m;
  };
  const unsigned m_match = lane_mask(is_match), m_ctx = lane_mask(is_ctx);
  const unsigned m_pf = lane_mask(pf_lane), m_rowc = lane_mask(is_rowc);
  const unsigned xoff = (unsigned)kTeamX + 4u * (unsigned)ci;

  unsigned gidx = 0, h = 0;
  int p = 0;
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (Chain::comp[i].type == C_CONS) { if (ci == i) p = ((int)Chain::comp[i].a1 - 128) * 4; }
  });
  unsigned v0 = 0;
  unsigned ra = 0, rb = 0, rc = 0, rlimit = 0, mpred = 0, mdd = 0;
  // MATCH: what the end of a byte reads -- the index entry of the byte's context, the history behind the candidate it
  // names, the byte a match predicts next -- has addresses known when the byte starts (pipe_kernel.h::pipe_match): fetched
  // then (the entry, the continuing match's next byte) and half way (the 8 bytes behind the candidate, the byte at it)
  unsigned mcmv = 0, mcont = 0, mcand_at = 0;
  unsigned long long mcand = 0, mhist = 0;
  int mixw[NMIX], mixp[NMIX];                                // lane t: weight t of the selected row, the input it multiplies
  unsigned mixrow[NMIX];
  unsigned ssev[NSSE], ssecx[NSSE];
  unsigned gwc0 = 0, gwc1 = 0;
  int mixc0[NMIX], mixc1[NMIX];
  unsigned ssec0[NSSE], ssec1[NSSE];
  int m2w[NMIX2], m2c0[NMIX2], m2c1[NMIX2], m2d[NMIX2];     // MIX2: weight (every lane of the half), candidates, p[j] - p[k]
  unsigned m2idx[NMIX2];
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { mixw[k] = 0; mixp[k] = 0; mixrow[k] = 0; mixc0[k] = 0; mixc1[k] = 0; }
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { ssev[k] = 0; ssecx[k] = 0; ssec0[k] = 0; ssec1[k] = 0; }
#pragma unroll
  for (int k = 0; k < NMIX2; ++k) { m2w[k] = 0; m2c0[k] = 0; m2c1[k] = 0; m2d[k] = 0; m2idx[k] = 0; }
  unsigned mixbase[NMIX], ssebase[NSSE], mixst[NMIX], mixin[NMIX], ssest[NSSE], m2base[NMIX2], m2st[NMIX2];
  int mixsrc[NMIX];
  unsigned isl[N];
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { mixbase[k] = dummy; mixst[k] = dummy; mixin[k] = 0; mixsrc[k] = lane; }
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { ssebase[k] = dummy; ssest[k] = dummy; }
#pragma unroll
  for (int k = 0; k < NMIX2; ++k) { m2base[k] = dummy; m2st[k] = dummy; }
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr CompK c = Chain::comp[i];
    isl[i] = 0;
    if constexpr (team_dep_type(c.type)) isl[i] = lane_mask(ci == i && live);
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
    } else if constexpr (c.type == C_MIX2) {
      constexpr int k2 = team_mix2_slot<Chain>(i);
      m2base[k2] = live ? (unsigned)c.t0 + hoff : dummy;
      ZPQ_OPAQUE(m2base[k2]);
      m2st[k2] = (ci == 0 && live) ? (unsigned)c.t0 + hoff : dummy;
      ZPQ_OPAQUE(m2st[k2]);
      if constexpr (c.mask0 == 0u) m2w[k2] = (int)G32(m2base[k2]);       // a table of one weight never leaves its register
    }
  });
  const unsigned m_lane0 = lane_mask(ci == 0 && live);
  unsigned dtv = 0;
  unsigned ssetr[NSSE], ssedt[NSSE];
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { ssetr[k] = 0; ssedt[k] = 0; }
  unsigned hmix[NMIX], hsse[NSSE], hm2[NMIX2];
#pragma unroll
  for (int k = 0; k < NMIX; ++k) hmix[k] = 0;
#pragma unroll
  for (int k = 0; k < NSSE; ++k) hsse[k] = 0;
#pragma unroll
  for (int k = 0; k < NMIX2; ++k) hm2[k] = 0;
  int rep[N], rsq[N];                                         // dependent components: prediction and its squash, in every lane of the half
#pragma unroll
  for (int k = 0; k < N; ++k) { rep[k] = 0; rsq[k] = 0; }
  int ylast = 0;
  unsigned tch = 0;                                           // what the touches loaded (kept alive, never used)
  TEAM_PROF_DECL

  int c8 = 1, hmap4 = 1;
  unsigned low = 1, high = 0xFFFFFFFFu;
  unsigned steps = 0;
  int status = 0;

  // ---- before [A]: everything of this bit that does not need the row components' predictions
  auto pre = [&](auto bitc) __attribute__((always_inline)) {
    constexpr int B = decltype(bitc)::value;
    constexpr bool pf_now = B > 0;
    constexpr bool last_of_nibble = B == 3;
    const int c8a = c8 * 2, c8b = c8 * 2 + 1;
    const int hm4a = last_of_nibble ? ((hmap4 & 0xf) << 5 | 1) : ((hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2) & 0xf));
    const int hm4b = last_of_nibble ? ((hmap4 & 0xf) << 5 | 1 << 4 | 1)
                                    : ((hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + 1) & 0xf));
    unsigned gw;
    gidx = (h ^ (unsigned)hmap4) & gmask;
    if constexpr (pf_now) {
      gw = ylast ? gwc1 : gwc0;
      static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr CompK c = Chain::comp[i];
        if constexpr (c.type == C_MIX && mix_pf(c)) mixw[c.slot] = ylast ? mixc1[c.slot] : mixc0[c.slot];
        if constexpr (c.type == C_SSE && sse_pf(c)) ssev[c.slot] = ylast ? ssec1[c.slot] : ssec0[c.slot];
        if constexpr (c.type == C_MIX2 && mix2_pf(c)) {
          constexpr int k2 = team_mix2_slot<Chain>(i);
          m2w[k2] = ylast ? m2c1[k2] : m2c0[k2];
        }
      });
      if constexpr (Chain::ANY_NONPF_GL) {
        if (is_cm && !pf_lane) gw = G32(goff + 4u * gidx);
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
        if constexpr (c.type == C_MIX2 && mix2_pf(c)) {
          constexpr int k2 = team_mix2_slot<Chain>(i);
          m2w[k2] = (int)G32(m2base[k2] + 4u * ((hm2[k2] + (unsigned)(c8 & 255)) & c.mask0));
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
      } else if constexpr (c.type == C_MIX2 && c.mask0 != 0u) {
        constexpr int k2 = team_mix2_slot<Chain>(i);
        const unsigned hi = hm2[k2];
        m2idx[k2] = (hi + (unsigned)(c8 & (int)c.a5)) & c.mask0;
        if constexpr (mix2_pf(c)) {
          m2c0[k2] = (int)G32(m2base[k2] + 4u * ((hi + (unsigned)(c8a & 255)) & c.mask0));
          m2c1[k2] = (int)G32(m2base[k2] + 4u * ((hi + (unsigned)(c8b & 255)) & c.mask0));
        } else {
          m2w[k2] = (int)G32(m2base[k2] + 4u * m2idx[k2]);
        }
      }
    });
    {
      const unsigned ia = ((h ^ (unsigned)hm4a) & gmask) & m_pf, ib = ((h ^ (unsigned)hm4b) & gmask) & m_pf;
      gwc0 = G32(goff + 4u * ia);
      gwc1 = G32(goff + 4u * ib);
    }
    if constexpr (ZPQ_TOUCH2 != 0 && B <= 5) {
      // the rows of the bit after next: four per table that is too large for the caches
      ZPQ_OPAQUE(tch);
      unsigned t = 0;
      static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr CompK c = Chain::comp[i];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const unsigned c8f = (unsigned)(c8 * 4 + f);
          if constexpr (c.type == C_MIX && mix_pf(c) && team_far_table(4ull * c.stride * (c.mask0 + 1ull)))
            t ^= G32(mixbase[c.slot] + 4u * (((hmix[c.slot] + (c8f & 255u)) & c.mask0) * c.stride));
          if constexpr (c.type == C_SSE && sse_pf(c) && team_far_table(4ull * (c.mask0 + 1ull)))
            t ^= G32(ssebase[c.slot] + 4u * (((hsse[c.slot] + c8f) * 32u) & c.mask0));
          if constexpr (c.type == C_MIX2 && mix2_pf(c) && team_far_table(4ull * (c.mask0 + 1ull)))
            t ^= G32(m2base[team_mix2_slot<Chain>(i)] + 4u * ((hm2[team_mix2_slot<Chain>(i)] + (c8f & 255u)) & c.mask0));
        }
      });
      tch = t;
    }
    // MATCH
    if constexpr (B == 0) {
      mcmv = G32(is_match ? off0 + 4u * (h & mask0) : dummy);
      mcont = G8(is_match ? off1 + ((rlimit + 1u - rb) & mask1) : dummy);
    } else if constexpr (B == 4) {
      const unsigned cpos = (mcmv - 8u) & mask1;
      const bool wraps = cpos + 8u > mask1 + 1u || mask1 < 15u;
      mcand = *(g_u64u*)(arena + (is_match && !wraps ? off1 + cpos : dummy));
      mcand_at = G8(is_match ? off1 + (mcmv & mask1) : dummy);
    }
    const bool m_on = is_match && ra != 0;
    rc = m_on ? ((mpred >> (7 - B)) & 1u) : rc;
    const unsigned msx = m_on ? ((rc ? 0u - mdd : mdd) & 32767u) : 16384u;
    v0 = gw;
    const unsigned sx = sp_blend(m_match, msx, v0 >> 17);
    const int st = sp_stretch(T, sx & 32767u);
    p = (int)sp_blend(m_ctx, (unsigned)st, (unsigned)p);
    dtv = (unsigned)T.dt[v0 & 0x3ffu];
  };

  // prediction of component j for every lane of the half: a dependent component's is there already
  auto pred_of = [&](auto jc) __attribute__((always_inline)) -> int {
    constexpr int j = decltype(jc)::value;
    if constexpr (team_dep_type(Chain::comp[j].type)) return rep[j];
    else return dual_bc(p, j, upper);
  };

  // ---- after [A]: the dependent components, the final probability
  auto chain = [&]() __attribute__((always_inline)) -> unsigned {
    const unsigned px = L32(xoff);
    p = (int)sp_blend(m_rowc, px, (unsigned)p);
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr CompK c = Chain::comp[i];
      if constexpr (c.type == C_AVG) {
        const int pj = pred_of(IC<(int)c.a1>{}), pk = pred_of(IC<(int)c.a2>{});
        rep[i] = (pj * (int)c.a3 + pk * (256 - (int)c.a3)) >> 8;
        p = (int)sp_blend(isl[i], (unsigned)rep[i], (unsigned)p);
      } else if constexpr (c.type == C_MIX2) {
        constexpr int k2 = team_mix2_slot<Chain>(i);
        const int pj = pred_of(IC<(int)c.a2>{}), pk = pred_of(IC<(int)c.a3>{});
        m2d[k2] = pj - pk;
        const int w = m2w[k2];
        rep[i] = sp_mad24(w, pj, __mul24(65536 - w, pk)) >> 16;              // 17-bit x 12-bit products
        rsq[i] = sp_squash(T, sp_clamp2k(rep[i]));
        p = (int)sp_blend(isl[i], (unsigned)rep[i], (unsigned)p);
      } else if constexpr (c.type == C_MIX) {
        int pin = p;
        if constexpr (c.a2 != 0) pin = __shfl(p, mixsrc[c.slot]);
        mixp[c.slot] = pin;
        const int x = (int)((unsigned)__mul24(mixw[c.slot] >> 8, pin) & mixin[c.slot]);
        rep[i] = sp_clamp2k(dual_half_sum<(int)c.a3>(x, upper) >> 8);
        rsq[i] = sp_squash(T, rep[i]);
        p = (int)sp_blend(isl[i], (unsigned)rep[i], (unsigned)p);
      } else if constexpr (c.type == C_SSE) {
        int pq = pred_of(IC<(int)c.a2>{}) + 992;
        pq = min(max(pq, 0), 1983);
        const int wt = pq & 63;
        pq >>= 6;
        const int base = upper ? 32 : 0;
        const unsigned e0 = __shfl(ssev[c.slot], base + pq), e1 = __shfl(ssev[c.slot], base + pq + 1);
        rep[i] = sp_stretch(T, ((e0 >> 10) * (unsigned)(64 - wt) + (e1 >> 10) * (unsigned)wt) >> 13);
        p = (int)sp_blend(isl[i], (unsigned)rep[i], (unsigned)p);
        ssecx[c.slot] += (unsigned)(pq + (wt >> 5));
        ssetr[c.slot] = (wt >> 5) ? e1 : e0;
        ssedt[c.slot] = (unsigned)T.dt[ssetr[c.slot] & 0x3ffu];
      }
    });
    constexpr unsigned tlast = Chain::comp[N - 1].type;
    if constexpr (tlast == C_MIX2 || tlast == C_MIX) return (unsigned)rsq[N - 1];
    else return (unsigned)sp_squash(T, sp_clamp2k(pred_of(IC<N - 1>{})));
  };

  // ---- after [B]: update of this wavefront's components (Predictor::update0 cases CM, MATCH, MIX2, MIX, SSE)
  auto update = [&](auto bitc, int y) __attribute__((always_inline)) {
    constexpr int B = decltype(bitc)::value;
    constexpr bool byte_done = B == 7;
    const unsigned count = v0 & 0x3ffu;
    const int yq = y * 32767;
    const int errcm = yq - (int)(v0 >> 17);
    const unsigned cm_new = v0 + ((unsigned)__mul24(errcm, (int)dtv) & 0xFFFFFC00u) + (count < limit ? 1u : 0u);
    G32(goff + 4u * gidx) = cm_new;                           // (lanes that are no CM: their dummy)
    ra = (is_match && (int)rc != y) ? 0u : ra;
    if (byte_done && is_match) {
      // Predictor::update0 case MATCH at the end of a byte (libzpaq.cpp:1992-2006), from registers
      const unsigned mask = mask1;
      const unsigned byte = (unsigned)(c8 * 2 + y) & 255u;
      const unsigned wpos = rlimit & mask;                     // where this byte goes
      G8(off1 + wpos) = (unsigned char)byte;
      mhist = mhist << 8 | byte;
      rlimit = (rlimit + 1) & mask;
      const unsigned eo = off0 + 4u * (h & mask0);
      bool fresh = false;
      if (ra == 0) {
        rb = rlimit - mcmv;
        if (rb & mask) {
          // 8 bytes behind the candidate against the last 8 bytes coded: legal when they do not wrap and were not touched
          // by this byte's store (before 8 bytes have been coded `mhist` holds zeros where the reference reads the
          // never-written end of the buffer); longer matches and the special cases take the reference's byte loop
          const unsigned cpos = (mcmv - 8u) & mask;
          const bool wraps = cpos + 8u > mask + 1u || mask < 15u;
          const bool overlap = ((mcmv - 1u - wpos) & mask) < 8u;
          unsigned m = 0;
          if (!wraps && !overlap) {
            const unsigned long long diff = __builtin_bswap64(mcand) ^ mhist;
            m = diff ? (unsigned)(__builtin_ctzll(diff) >> 3) : 8u;
          }
          ra = m;
          if (wraps || overlap || m == 8u)
            while (ra < 255 && G8(off1 + ((rlimit - ra - 1) & mask)) == G8(off1 + ((rlimit - ra - rb - 1) & mask))) ++ra;
        }
        fresh = true;
      } else ra += ra < 255;
      G32(eo) = rlimit;
      if (ra != 0) {
        const unsigned ppos = (rlimit - rb) & mask;            // = the candidate for a fresh match, the continuing position otherwise
        const unsigned early = fresh ? mcand_at : mcont;
        mpred = ppos == wpos ? byte : early;
        mdd = T.dt2k[ra];
      }
    }
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr CompK c = Chain::comp[i];
      if constexpr (c.type == C_MIX) {
        const int err = ((yq - rsq[i]) * (int)c.a4) >> 4;
        const int w = sp_clamp512k(mixw[c.slot] + (sp_mad24(err, mixp[c.slot], 1 << 12) >> 13));
        const unsigned wo = mixst[c.slot] + ((4u * mixrow[c.slot]) & mixin[c.slot]);
        *(g_i32*)(arena + wo) = w;
      } else if constexpr (c.type == C_MIX2) {
        constexpr int k2 = team_mix2_slot<Chain>(i);
        const int err2 = __mul24(yq - rsq[i], (int)c.a4) >> 5;
        const int w2 = min(max(m2w[k2] + (sp_mad24(err2, m2d[k2], 1 << 12) >> 13), 0), 65535);   // 19-bit x 13-bit
        if constexpr (c.mask0 == 0u) m2w[k2] = w2;
        else *(g_i32*)(arena + m2st[k2] + ((4u * m2idx[k2]) & m_lane0)) = w2;
      } else if constexpr (c.type == C_SSE) {
        const unsigned e = ssecx[c.slot];
        const unsigned v = ssetr[c.slot];
        const unsigned cnt = v & 0x3ffu;
        const int err = yq - (int)(v >> 17);
        const unsigned prod = (unsigned)__mul24(err, (int)ssedt[c.slot]);
        const unsigned nv = v + (prod & 0xFFFFFC00u) + (cnt < c.limit ? 1u : 0u);
        *(g_u32*)(arena + ssest[c.slot] + ((4u * (e & c.mask0)) & m_lane0)) = nv;
      }
    });
    ylast = y;
  };

  // ---- the coder of this half (Decoder::decode, libzpaq.cpp:2159-2181), on the vector unit
  unsigned rp = 0, nout = 0, curr = 0;
  bool run = live;
  bool eos = false;
  if (live && (delta64 >> 32) != 0) { status = 8; run = false; }   // arenas of the workgroup not inside a 4 GiB window
  // Coded bytes come from a window of 8 in registers, rebuilt once per input byte from a 16-byte fetch issued a byte
  // earlier: a byte shifted into the coder is then no memory operation.  (Fetched where it is needed, it waits -- vmcnt
  // counts in order -- for every candidate row requested just before it: a full trip to HBM, and with 8 blocks in lockstep
  // some block needs a byte at almost every bit.)  Reads stay inside the input rounded up to 64 bytes, which the engine
  // guarantees to be readable.
  const unsigned in_lim = (in_len + 63u) & ~63u;
  unsigned long long iwin = 0, if0 = 0, if1 = 0;
  unsigned iavail = 0, ifpos = 0;
  auto in_fetch = [&]() __attribute__((always_inline)) {      // the 16 bytes at the read position (or the last 16 readable ones)
    ifpos = in_lim >= 16u ? min(rp, in_lim - 16u) : 0u;
    if0 = *(g_u64u*)(in_ptr + ifpos);
    if1 = *(g_u64u*)(in_ptr + ifpos + 8u);
  };
  auto in_window = [&]() __attribute__((always_inline)) {     // window <- what the last fetch holds from the read position on
    const unsigned o = rp - ifpos;
    const unsigned long long b0 = __builtin_bswap64(if0), b1 = __builtin_bswap64(if1);
    const unsigned sh = (o & 7u) * 8u;
    const unsigned long long lo = sh ? (b0 << sh) | (b1 >> (64u - sh)) : b0;
    const unsigned long long hi = b1 << sh;
    iwin = o < 8u ? lo : hi;
    iavail = in_lim < 16u ? 0u : (o < 8u ? 8u : (o < 16u ? 16u - o : 0u));
  };
  auto in_byte = [&]() __attribute__((always_inline)) -> unsigned {     // the coded byte at rp (rp < in_len)
    unsigned v;
    if (iavail) {
      v = (unsigned)(iwin >> 56);
      iwin <<= 8;
      --iavail;
    } else {
      v = in_ptr[rp];
      ZPQ_OPAQUE(v);            // (waited for inside this rarely taken branch, not where the paths join)
    }
    ++rp;
    return v;
  };
  // in two halves: the bit first -- the row wavefronts wait for it --, the range update and the bytes shifted in behind [B]
  unsigned dmid = 0;
  auto decode_bit = [&](unsigned pr) __attribute__((always_inline)) -> int {     // (selects, no branches: every bit passes here)
    const bool bad = run && (curr < low || curr > high);
    status = bad ? 2 : status;
    run = run && !bad;
    dmid = low + (unsigned)(((unsigned long long)(high - low) * pr) >> 16);
    return (run && curr <= dmid) ? 1 : 0;
  };
  auto decode_shift = [&](int y) __attribute__((always_inline)) {
    if (!run) return;
    if (y) high = dmid; else low = dmid + 1;
    while ((high ^ low) < 0x1000000u) {
      high = high << 8 | 255u;
      low = low << 8;
      low += (low == 0);
      if (rp >= in_len) { status = 6; run = false; break; }
      curr = curr << 8 | in_byte();
    }
  };
  auto decode = [&](unsigned pr) __attribute__((always_inline)) -> int {
    const int y = decode_bit(pr);
    decode_shift(y);
    return y;
  };
  auto run_hcomp = [&](unsigned input) __attribute__((always_inline)) -> int {
#ifndef ZPQ_EMU
    int e = 0;
    if (run) e = Chain::hcomp(input, vm_b, vm_c, vm_d, vm_f, vm_M, vm_H, vm_R);
    return e;
#else
    int e = 0;
    if (run && ci == 0) e = Chain::hcomp(input, vm_b, vm_c, vm_d, vm_f, vm_M, vm_H, vm_R);
    return dual_bc(e, 0, upper);
#endif
  };
  auto any_running = [&]() __attribute__((always_inline)) -> bool {
    unsigned r = 0;
    for (int i = 0; i < kTeamBlocks; ++i) r |= *(lds_u32*)(lds0 + (unsigned)(i * kRegion + kTeamRun));
    return r != 0;
  };

  if (run && in_len) { in_fetch(); in_window(); }
  for (int i = 0; i < 4; ++i) {
    if (!run) break;
    if (rp >= in_len) { status = 6; run = false; break; }
    curr = curr << 8 | in_byte();
  }
  if (run && nout >= out_cap) run = false;
  if (ci == 0) L32((unsigned)kTeamRun) = run ? 1u : 0u;
  ZPQ_TEAM_BARRIER();                                        // [S]
  bool any = any_running();
  if (run) in_fetch();
  while (any) {
    int ch = 1;
    // the window for this byte's 9 decoding steps from the fetch of a byte ago, then the fetch for the next byte
    if (run) { in_window(); in_fetch(); }
    const int flag = decode(0);                               // end-of-stream flag, coded with p = 0
    if (run && flag) { eos = true; if (curr != 0) status = 2; run = false; }
    static_for<0, 8>([&](auto bitc) __attribute__((always_inline)) {
      constexpr int B = decltype(bitc)::value;
      pre(bitc);
      TEAM_PROF((B == 0 ? 7 : 0));
      ZPQ_TEAM_BARRIER();                                    // [A] the row components' predictions are in LDS
      TEAM_PROF(1);
      const unsigned pr = chain() * 2u + 1u;
      const int y = decode_bit(pr);
      if (ci == 0) L32((unsigned)kTeamY) = (unsigned)y;
      TEAM_PROF(2);
      ZPQ_TEAM_BARRIER();                                    // [B]
      TEAM_PROF(3);
      TEAM_PROF_BIT();
      decode_shift(y);
      ch += ch + y;
      if constexpr (B != 7 || !ZPQ_TEAM_LATE_UPDATE7) {
        update(bitc, y);
        c8 += c8 + y;
        TEAM_PROF(4);
      }
      if constexpr (B == 7) {
        // HCOMP needs the byte, not the trained components: it runs first, and the last bit's update behind [C], while the
        // row wavefronts are out for the next byte's rows (it reads h, c8 and the contexts of the byte that ends: they
        // change only below)
        const int e = run_hcomp((unsigned)((ZPQ_TEAM_LATE_UPDATE7 ? c8 + c8 + y : c8) - 256));
        if (run && e) { status = e; run = false; }
        if (run) {
          if (ci == 0) out_ptr[nout] = (unsigned char)(ch - 256);
          ++nout;
          ++steps;
          if (nout >= out_cap) run = false;
        }
        if (ci == 0) L32((unsigned)kTeamRun) = run ? 1u : 0u;
        TEAM_PROF(5);
        ZPQ_TEAM_BARRIER();                                  // [C]
        TEAM_PROF(6);
        if constexpr (ZPQ_TEAM_LATE_UPDATE7) {
          update(bitc, y);
          TEAM_PROF(4);
        }
        any = any_running();
        h = vm_H[(unsigned)ci & Chain::HMASK];
        static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          constexpr CompK c = Chain::comp[i];
          if constexpr (c.type == C_MIX) hmix[c.slot] = dual_bcu(h, i, upper);
          if constexpr (c.type == C_SSE) hsse[c.slot] = dual_bcu(h, i, upper);
          if constexpr (c.type == C_MIX2 && c.mask0 != 0u) hm2[team_mix2_slot<Chain>(i)] = dual_bcu(h, i, upper);
        });
        hmap4 = 1;
        c8 = 1;
      } else if constexpr (B == 3) {
        hmap4 = (hmap4 & 0xf) << 5 | y << 4 | 1;
        if (run) ++steps;
      } else {
        hmap4 = (hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + y) & 0xf);
        if (run) ++steps;
      }
    });
  }
  ZPQ_KEEP2(tch, tch);
  if (ci == 0 && live) {
    res[rslot].out_len = nout;
    res[rslot].consumed = eos ? rp : 0;
    res[rslot].status = status;
    res[rslot].steps = steps;
  }
#if defined(ZPQ_PROF) && !defined(ZPQ_EMU)
  if (blockIdx.x == 0 && tw == 0 && lane == 0 && tp_n_)
    printf("[zpq team prof] mixers : bits=%llu cycles: pre per bit but a byte's first=%.0f, per first bit of a byte=%.0f; per bit: wait[A]=%.0f chain+decode=%.0f "
           "wait[B]=%.0f update=%.0f hcomp/8=%.0f wait[C]/8=%.0f\n",
           tp_n_, (double)tp_[0] / (tp_n_ * 0.875), (double)tp_[7] / (tp_n_ * 0.125), (double)tp_[1] / tp_n_, (double)tp_[2] / tp_n_, (double)tp_[3] / tp_n_,
           (double)tp_[4] / tp_n_, (double)tp_[5] / tp_n_, (double)tp_[6] / tp_n_);
#endif
}

// =====================================================================================================================
// Round 6: the TAIL wavefront.  A mixer wavefront above serves two blocks and runs the WHOLE stream of everything that is
// not an ICM / ISSE -- CM, MATCH, the MIX dot products, MIX2, SSE, squash, the arithmetic coder, HCOMP -- so the compute unit
// issues that stream four times per bit for its 8 blocks, and the profile of round 4 says the bit is bound by issue slots
// (per SIMD a row and a mixer wavefront: ~150 + ~440 instructions, ~2 400 of a bit's ~2 650 cycles).  What needs 32 lanes per
// block is only the MIX dot product (lane t = weight t).  Everything else is a handful of scalars per block.  So, for chains
// with at least one MIX and at most 8 CM / MATCH components (TeamTailMap):
//
//   MIX wavefronts (4)  two blocks each, lane = weight index: the candidate rows of the next bit, the dot products (inputs from
//                       the exchange words X in LDS, outputs back into X), the weights' training.  ~110 instructions per bit.
//   TAIL wavefront (1)  lane = (block, role), 8 lanes per block: role r owns the r-th CM / MATCH of the chain (its table word,
//                       candidates, training); AVG / MIX2 / SSE, squash, the coder and HCOMP are computed by all 8 lanes of a
//                       block alike (values every lane has: nothing to broadcast; an SSE row's 32 entries sit 4 per lane and
//                       the two the prediction needs come through ds_bpermute).  ONE stream for the 8 blocks.
//
//     rows predict, tail: CM / MATCH predict -> [A] -> mix: dot products -> [A2] -> tail: MIX2, SSE, squash, decode -> [B] -> update
//
// One more barrier per bit, ~970 instead of ~1 840 wave-instructions per bit and compute unit.  The model arithmetic is the
// mixers' above, statement for statement.
// Measured (profiles/r06_results.md): SLOWER than the form above, 139.6 against 159.5 MB/s on the mixed corpus at the decoder's
// operating point.  The bit is a chain of latencies -- rows -> [A] -> dot products -> MIX2 / SSE -> coder -> [B] -> updates -- and
// the split puts a third barrier and an LDS hand-over into it; the issue slots it frees were not what the bit waited for.
// Kept behind ZPAQ_AMD_TEAM_TAIL=1 (host/codegen.cpp writes the define), bit-exact in the emulator and on the GPU.
#ifndef ZPQ_TEAM_TAIL
#define ZPQ_TEAM_TAIL 0
#endif

template <class Chain>
struct TeamTailMap {
  int nrole;              // CM / MATCH components
  int role_comp[8];       // chain index of role r's component (-1: none)
  bool ok;
};
template <class Chain>
constexpr TeamTailMap<Chain> team_tail_map() {
  TeamTailMap<Chain> m{};
  m.nrole = 0;
  m.ok = ZPQ_TEAM_TAIL != 0 && Chain::NMIX > 0;
  for (int i = 0; i < 8; ++i) m.role_comp[i] = -1;
  for (int i = 0; i < Chain::N; ++i) {
    const unsigned t = Chain::comp[i].type;
    if (t == C_CM || t == C_MATCH) {
      if (m.nrole < 8) m.role_comp[m.nrole] = i;
      ++m.nrole;
    }
  }
  if (m.nrole > 8) m.ok = false;
  // the mix wavefronts run BEFORE the tail's dependent components: a MIX may take row components, CM, MATCH, CONS and earlier
  // MIXes as inputs (every chain compressBlock's methods and the legacy models make), not an AVG / MIX2 / SSE
  for (int i = 0; i < Chain::N; ++i) {
    if (Chain::comp[i].type != C_MIX) continue;
    for (unsigned t = 0; t < Chain::comp[i].a3; ++t) {
      const unsigned ty = Chain::comp[Chain::comp[i].a2 + t].type;
      if (ty == C_AVG || ty == C_MIX2 || ty == C_SSE) m.ok = false;
    }
  }
  return m;
}
template <class Chain>
constexpr bool team_tail_ok() { return team_tail_map<Chain>().ok; }

// ---- MIX wavefront: two blocks, lane = (block, weight index) ----------------------------------------------------------
template <class Chain, class TT>
__device__ __forceinline__ void team_mix(const TT& T, lds_u8* const lds0, const BlockJob* jobs, unsigned nblocks, int tw, int lane) {
  constexpr int N = Chain::N;
  constexpr int NMIX = Chain::NMIX > 0 ? Chain::NMIX : 1;
  constexpr int kRegion = team_block_lds_bytes();
  const int ci = lane & 31;
  const bool upper = lane >= 32;
  const unsigned wg0 = blockIdx.x * (unsigned)kTeamBlocks;
  const unsigned bw = (unsigned)tw * 2u + (upper ? 1u : 0u);
  const unsigned b = wg0 + bw;
  const bool live = b < nblocks;
  const BlockJob job0 = jobs[wg0];
  const BlockJob job = jobs[live ? b : wg0];
  g_u8* const arena = (g_u8*)sp_uni64((unsigned long long)job0.arena);
  const unsigned long long delta64 = (unsigned long long)job.arena - (unsigned long long)job0.arena;
  const unsigned hoff = live ? (unsigned)delta64 : 0u;
  lds_u8* const wl = lds0 + bw * (unsigned)kRegion;
  const unsigned dummy = (unsigned)Chain::OFF_RUN;
  auto G32 = [&](unsigned off) __attribute__((always_inline)) -> g_u32& { return *(g_u32*)(arena + off); };
  auto L32 = [&](unsigned off) __attribute__((always_inline)) -> lds_u32& { return *(lds_u32*)(wl + off); };
  lds_u32* const vm_H = (lds_u32*)(wl + Chain::H_LDS);
  auto lane_mask = [&](bool x) __attribute__((always_inline)) -> unsigned {
    unsigned m = x ? 0xFFFFFFFFu : 0u;
    ZPQ_OPAQUE(m);
    return m;
  };
  int mixw[NMIX], mixp[NMIX], mixc0[NMIX], mixc1[NMIX], rsq[NMIX];
  unsigned mixrow[NMIX], mixbase[NMIX], mixst[NMIX], mixin[NMIX], mixx[NMIX], hmix[NMIX];
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { mixw[k] = 0; mixp[k] = 0; mixc0[k] = 0; mixc1[k] = 0; rsq[k] = 0; mixrow[k] = 0; mixbase[k] = dummy; mixst[k] = dummy; mixin[k] = 0; mixx[k] = (unsigned)kTeamX; hmix[k] = 0; }
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr CompK c = Chain::comp[i];
    if constexpr (c.type == C_MIX) {
      static_assert(c.a2 + c.a3 <= 32, "MIX inputs must sit inside one half");
      mixbase[c.slot] = (unsigned)c.t0 + hoff + 4u * (unsigned)min(ci, (int)c.a3 - 1);
      ZPQ_OPAQUE(mixbase[c.slot]);
      mixin[c.slot] = lane_mask(ci < (int)c.a3 && live);
      mixst[c.slot] = (ci < (int)c.a3 && live) ? (unsigned)c.t0 + hoff + 4u * (unsigned)ci : dummy;
      ZPQ_OPAQUE(mixst[c.slot]);
      mixx[c.slot] = (unsigned)kTeamX + 4u * (unsigned)(((int)c.a2 + min(ci, (int)c.a3 - 1)) & 31);     // the input this lane multiplies
    }
  });
  int c8 = 1, ylast = 0;
  auto any_running = [&]() __attribute__((always_inline)) -> bool {
    unsigned r = 0;
    for (int i = 0; i < kTeamBlocks; ++i) r |= *(lds_u32*)(lds0 + (unsigned)(i * kRegion + kTeamRun));
    return r != 0;
  };
  ZPQ_TEAM_BARRIER();                                        // [S]
  bool any = any_running();
  while (any) {
    static_for<0, 8>([&](auto bitc) __attribute__((always_inline)) {
      constexpr int B = decltype(bitc)::value;
      const int c8a = c8 * 2, c8b = c8 * 2 + 1;
      // ---- before [A]: this bit's rows (fetched a bit ago as candidates), the next bit's candidates
      static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr CompK c = Chain::comp[i];
        if constexpr (c.type == C_MIX) {
          const unsigned hi = hmix[c.slot];
          mixrow[c.slot] = ((hi + (unsigned)(c8 & (int)c.a5)) & c.mask0) * c.stride;
          if constexpr (mix_pf(c)) {
            if constexpr (B > 0) mixw[c.slot] = ylast ? mixc1[c.slot] : mixc0[c.slot];
            else mixw[c.slot] = (int)G32(mixbase[c.slot] + 4u * mixrow[c.slot]);
            mixc0[c.slot] = (int)G32(mixbase[c.slot] + 4u * (((hi + (unsigned)(c8a & 255)) & c.mask0) * c.stride));
            mixc1[c.slot] = (int)G32(mixbase[c.slot] + 4u * (((hi + (unsigned)(c8b & 255)) & c.mask0) * c.stride));
          } else {
            mixw[c.slot] = (int)G32(mixbase[c.slot] + 4u * mixrow[c.slot]);
          }
        }
      });
      ZPQ_TEAM_BARRIER();                                    // [A] the rows' and the tail's predictions are in X
      static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr CompK c = Chain::comp[i];
        if constexpr (c.type == C_MIX) {
          const int pin = (int)L32(mixx[c.slot]);             // (an earlier MIX of this wavefront wrote its output just above: LDS operations of a wavefront are in order)
          mixp[c.slot] = pin;
          const int x = (int)((unsigned)__mul24(mixw[c.slot] >> 8, pin) & mixin[c.slot]);
          const int rp = sp_clamp2k(dual_half_sum<(int)c.a3>(x, upper) >> 8);
          rsq[c.slot] = sp_squash(T, rp);
          if (ci == 0) L32((unsigned)kTeamX + 4u * (unsigned)i) = (unsigned)rp;     // the MIX's prediction: for a later MIX of this wavefront and for the tail
        }
      });
      ZPQ_TEAM_BARRIER();                                    // [A2] the tail takes over
      ZPQ_TEAM_BARRIER();                                    // [B] the bit is known
      const int y = (int)L32((unsigned)kTeamY);
      const int yq = y * 32767;
      static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr CompK c = Chain::comp[i];
        if constexpr (c.type == C_MIX) {
          const int err = ((yq - rsq[c.slot]) * (int)c.a4) >> 4;
          const int w = sp_clamp512k(mixw[c.slot] + (sp_mad24(err, mixp[c.slot], 1 << 12) >> 13));
          const unsigned wo = mixst[c.slot] + ((4u * mixrow[c.slot]) & mixin[c.slot]);
          *(g_i32*)(arena + wo) = w;
        }
      });
      ylast = y;
      c8 += c8 + y;
      if constexpr (B == 7) {
        ZPQ_TEAM_BARRIER();                                  // [C] HCOMP has run
        any = any_running();
        static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          constexpr CompK c = Chain::comp[i];
          if constexpr (c.type == C_MIX) hmix[c.slot] = vm_H[(unsigned)i & Chain::HMASK];
        });
        c8 = 1;
      }
    });
  }
}

// ---- TAIL wavefront: lane = (block, role), 8 lanes per block ------------------------------------------------------------
template <class Chain, class TT>
__device__ __forceinline__ void team_tail(const TT& T, lds_u8* const lds0, const BlockJob* jobs, BlockResult* res, unsigned nblocks, int lane) {
  constexpr int N = Chain::N;
  constexpr auto TM = team_tail_map<Chain>();
  constexpr int NSSE = Chain::NSSE > 0 ? Chain::NSSE : 1;
  constexpr int NMIX2 = team_mix2_slot<Chain>(N) > 0 ? team_mix2_slot<Chain>(N) : 1;
  constexpr int kRegion = team_block_lds_bytes();
  const int r = lane & 7;                                     // role
  const unsigned bw = (unsigned)lane >> 3;                    // block of this lane inside the workgroup
  const unsigned wg0 = blockIdx.x * (unsigned)kTeamBlocks;
  const unsigned b = wg0 + bw;
  const bool live = b < nblocks;
  const BlockJob job0 = jobs[wg0];
  const BlockJob job = jobs[live ? b : wg0];
  g_u8* const arena = (g_u8*)sp_uni64((unsigned long long)job0.arena);
  const unsigned long long delta64 = (unsigned long long)job.arena - (unsigned long long)job0.arena;
  const unsigned hoff = live ? (unsigned)delta64 : 0u;
  const g_u8* const in_ptr = (const g_u8*)job.in;
  g_u8* const out_ptr = (g_u8*)job.out;
  const unsigned in_len = job.in_len, out_cap = job.out_cap, rslot = job.res_slot;
  lds_u8* const wl = lds0 + bw * (unsigned)kRegion;
  const unsigned dummy = (unsigned)Chain::OFF_RUN;
  const unsigned dummy_lds = (unsigned)(kRegion - 512) + (unsigned)r * 8u;

  // this lane's own component: the r-th CM / MATCH of the chain
  unsigned limit = 0, mask0 = 0, mask1 = 63, off0 = dummy, off1 = dummy, ctype = 0;
  int cidx = 0;
  static_for<0, 8>([&](auto rc) __attribute__((always_inline)) {
    constexpr int rr = decltype(rc)::value;
    if constexpr (TM.role_comp[rr] >= 0) {
      if (r == rr && live) {
        constexpr CompK c = Chain::comp[TM.role_comp[rr]];
        ctype = c.type; limit = c.limit; mask0 = c.mask0;
        off0 = (unsigned)c.t0 + hoff;
        if (c.type == C_MATCH) { off1 = (unsigned)c.t1 + hoff; mask1 = c.mask1; }
        cidx = TM.role_comp[rr];
      }
    }
  });
  auto G32 = [&](unsigned off) __attribute__((always_inline)) -> g_u32& { return *(g_u32*)(arena + off); };
  auto G8 = [&](unsigned off) __attribute__((always_inline)) -> g_u8& { return *(g_u8*)(arena + off); };
  auto G128 = [&](unsigned off) __attribute__((always_inline)) -> g_u128& { return *(g_u128*)(arena + off); };
  auto L32 = [&](unsigned off) __attribute__((always_inline)) -> lds_u32& { return *(lds_u32*)(wl + off); };

  // HCOMP machine of this block (every lane of the block runs it: identical values, identical stores)
  unsigned vm_b = 0, vm_c = 0, vm_d = 0, vm_f = 0;
  g_u8* const vm_M = arena + (unsigned)Chain::OFF_M + hoff;
  g_u32* const vm_R = (g_u32*)(arena + (unsigned)Chain::OFF_R + hoff);
  lds_u32* const vm_H = (lds_u32*)(wl + Chain::H_LDS);

  const bool is_cm = ctype == C_CM, is_match = ctype == C_MATCH;
  const bool is_ctx = is_cm || is_match;
  const bool pf_lane = is_cm && mask0 >= 511u;
  const unsigned goff = is_cm ? off0 : dummy;
  const unsigned gmask = is_cm ? mask0 : 0u;
  auto lane_mask = [&](bool x) __attribute__((always_inline)) -> unsigned {
    unsigned m = x ? 0xFFFFFFFFu : 0u;
    ZPQ_OPAQUE(m);
    return m;
  };
  const unsigned m_match = lane_mask(is_match), m_pf = lane_mask(pf_lane);
  const unsigned m_lane0 = lane_mask(r == 0 && live);
  const unsigned xoff = is_ctx ? (unsigned)kTeamX + 4u * (unsigned)cidx : dummy_lds;      // where this lane publishes its prediction

  unsigned gidx = 0, h = 0, v0 = 0, dtv = 0;
  unsigned ra = 0, rb = 0, rc = 0, rlimit = 0, mpred = 0, mdd = 0;
  unsigned mcmv = 0, mcont = 0, mcand_at = 0;
  unsigned long long mcand = 0, mhist = 0;
  unsigned gwc0 = 0, gwc1 = 0;
  uint4 ssev[NSSE], ssec0[NSSE], ssec1[NSSE];
  unsigned ssecx[NSSE], ssetr[NSSE], ssedt[NSSE], ssebase[NSSE], ssest[NSSE], hsse[NSSE];
  int m2w[NMIX2], m2c0[NMIX2], m2c1[NMIX2], m2d[NMIX2];
  unsigned m2idx[NMIX2], m2base[NMIX2], m2st[NMIX2], hm2[NMIX2];
#pragma unroll
  for (int k = 0; k < NSSE; ++k) {
    ssev[k] = make_uint4(0, 0, 0, 0); ssec0[k] = ssev[k]; ssec1[k] = ssev[k];
    ssecx[k] = 0; ssetr[k] = 0; ssedt[k] = 0; ssebase[k] = dummy; ssest[k] = dummy; hsse[k] = 0;
  }
#pragma unroll
  for (int k = 0; k < NMIX2; ++k) { m2w[k] = 0; m2c0[k] = 0; m2c1[k] = 0; m2d[k] = 0; m2idx[k] = 0; m2base[k] = dummy; m2st[k] = dummy; hm2[k] = 0; }
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr CompK c = Chain::comp[i];
    if constexpr (c.type == C_SSE) {
      ssebase[c.slot] = (unsigned)c.t0 + hoff + 16u * (unsigned)r;       // this lane's 4 of the row's 32 entries (a lane without a block reads the first block's table, never writes)
      ZPQ_OPAQUE(ssebase[c.slot]);
      ssest[c.slot] = (r == 0 && live) ? (unsigned)c.t0 + hoff : dummy;
      ZPQ_OPAQUE(ssest[c.slot]);
    } else if constexpr (c.type == C_MIX2) {
      constexpr int k2 = team_mix2_slot<Chain>(i);
      m2base[k2] = live ? (unsigned)c.t0 + hoff : dummy;
      ZPQ_OPAQUE(m2base[k2]);
      m2st[k2] = (r == 0 && live) ? (unsigned)c.t0 + hoff : dummy;
      ZPQ_OPAQUE(m2st[k2]);
      if constexpr (c.mask0 == 0u) m2w[k2] = (int)G32(m2base[k2]);       // a table of one weight never leaves its register
    }
  });
  // rows of an SSE table are 128 bytes: with ssebase a lane's 16, candidate rows are whole-line fetches of the 8 lanes
  auto sse_row = [&](int k, unsigned cx) __attribute__((always_inline)) -> uint4 { return G128(ssebase[k] + 4u * cx); };
  int rep[N], rsq[N];                                         // dependent components: prediction and its squash, in every lane
#pragma unroll
  for (int k = 0; k < N; ++k) { rep[k] = 0; rsq[k] = 0; }
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (Chain::comp[i].type == C_CONS) {
      rep[i] = ((int)Chain::comp[i].a1 - 128) * 4;
      if (r == 0) L32((unsigned)kTeamX + 4u * (unsigned)i) = (unsigned)rep[i];       // a constant: published once
    }
  });
  int ylast = 0;
  int c8 = 1, hmap4 = 1;
  unsigned low = 1, high = 0xFFFFFFFFu;
  unsigned steps = 0;
  int status = 0;
  int p = 0;

  // ---- before [A]: this bit's table words (fetched a bit ago as candidates), the next bit's candidates, CM / MATCH predict
  auto pre = [&](auto bitc) __attribute__((always_inline)) {
    constexpr int B = decltype(bitc)::value;
    constexpr bool pf_now = B > 0;
    constexpr bool last_of_nibble = B == 3;
    const int c8a = c8 * 2, c8b = c8 * 2 + 1;
    const int hm4a = last_of_nibble ? ((hmap4 & 0xf) << 5 | 1) : ((hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2) & 0xf));
    const int hm4b = last_of_nibble ? ((hmap4 & 0xf) << 5 | 1 << 4 | 1)
                                    : ((hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + 1) & 0xf));
    unsigned gw;
    gidx = (h ^ (unsigned)hmap4) & gmask;
    if constexpr (pf_now) {
      gw = ylast ? gwc1 : gwc0;
      if constexpr (Chain::ANY_NONPF_GL) {
        if (is_cm && !pf_lane) gw = G32(goff + 4u * gidx);
      }
    } else {
      gw = G32(goff + 4u * gidx);
    }
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr CompK c = Chain::comp[i];
      if constexpr (c.type == C_SSE) {
        const unsigned hi = hsse[c.slot];
        ssecx[c.slot] = ((hi + (unsigned)c8) * 32u) & c.mask0;
        if constexpr (sse_pf(c)) {
          if constexpr (pf_now) {
            ssev[c.slot].x = ylast ? ssec1[c.slot].x : ssec0[c.slot].x; ssev[c.slot].y = ylast ? ssec1[c.slot].y : ssec0[c.slot].y;
            ssev[c.slot].z = ylast ? ssec1[c.slot].z : ssec0[c.slot].z; ssev[c.slot].w = ylast ? ssec1[c.slot].w : ssec0[c.slot].w;
          } else ssev[c.slot] = sse_row(c.slot, ssecx[c.slot]);
          ssec0[c.slot] = sse_row(c.slot, ((hi + (unsigned)c8a) * 32u) & c.mask0);
          ssec1[c.slot] = sse_row(c.slot, ((hi + (unsigned)c8b) * 32u) & c.mask0);
        } else {
          ssev[c.slot] = sse_row(c.slot, ssecx[c.slot]);
        }
      } else if constexpr (c.type == C_MIX2 && c.mask0 != 0u) {
        constexpr int k2 = team_mix2_slot<Chain>(i);
        const unsigned hi = hm2[k2];
        m2idx[k2] = (hi + (unsigned)(c8 & (int)c.a5)) & c.mask0;
        if constexpr (mix2_pf(c)) {
          if constexpr (pf_now) m2w[k2] = ylast ? m2c1[k2] : m2c0[k2];
          else m2w[k2] = (int)G32(m2base[k2] + 4u * m2idx[k2]);
          m2c0[k2] = (int)G32(m2base[k2] + 4u * ((hi + (unsigned)(c8a & 255)) & c.mask0));
          m2c1[k2] = (int)G32(m2base[k2] + 4u * ((hi + (unsigned)(c8b & 255)) & c.mask0));
        } else {
          m2w[k2] = (int)G32(m2base[k2] + 4u * m2idx[k2]);
        }
      }
    });
    {
      const unsigned ia = ((h ^ (unsigned)hm4a) & gmask) & m_pf, ib = ((h ^ (unsigned)hm4b) & gmask) & m_pf;
      gwc0 = G32(goff + 4u * ia);
      gwc1 = G32(goff + 4u * ib);
    }
    // MATCH
    if constexpr (B == 0) {
      mcmv = G32(is_match ? off0 + 4u * (h & mask0) : dummy);
      mcont = G8(is_match ? off1 + ((rlimit + 1u - rb) & mask1) : dummy);
    } else if constexpr (B == 4) {
      const unsigned cpos = (mcmv - 8u) & mask1;
      const bool wraps = cpos + 8u > mask1 + 1u || mask1 < 15u;
      mcand = *(g_u64u*)(arena + (is_match && !wraps ? off1 + cpos : dummy));
      mcand_at = G8(is_match ? off1 + (mcmv & mask1) : dummy);
    }
    const bool m_on = is_match && ra != 0;
    rc = m_on ? ((mpred >> (7 - B)) & 1u) : rc;
    const unsigned msx = m_on ? ((rc ? 0u - mdd : mdd) & 32767u) : 16384u;
    v0 = gw;
    const unsigned sx = sp_blend(m_match, msx, v0 >> 17);
    p = sp_stretch(T, sx & 32767u);
    dtv = (unsigned)T.dt[v0 & 0x3ffu];
    L32(xoff) = (unsigned)p;                                  // CM / MATCH lanes: their prediction for the mix wavefronts (the others: a dummy word)
  };

  // prediction of component j, in every lane: a dependent component's is there already, everything else is in X
  auto pred_of = [&](auto jc) __attribute__((always_inline)) -> int {
    constexpr int j = decltype(jc)::value;
    constexpr unsigned t = Chain::comp[j].type;
    if constexpr (t == C_AVG || t == C_MIX2 || t == C_SSE || t == C_CONS) return rep[j];
    else return (int)L32((unsigned)kTeamX + 4u * (unsigned)j);
  };

  // ---- after [A2]: the dependent components behind the MIX dot products, the final probability
  auto chain = [&]() __attribute__((always_inline)) -> unsigned {
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr CompK c = Chain::comp[i];
      if constexpr (c.type == C_AVG) {
        const int pj = pred_of(IC<(int)c.a1>{}), pk = pred_of(IC<(int)c.a2>{});
        rep[i] = (pj * (int)c.a3 + pk * (256 - (int)c.a3)) >> 8;
      } else if constexpr (c.type == C_MIX2) {
        constexpr int k2 = team_mix2_slot<Chain>(i);
        const int pj = pred_of(IC<(int)c.a2>{}), pk = pred_of(IC<(int)c.a3>{});
        m2d[k2] = pj - pk;
        const int w = m2w[k2];
        rep[i] = sp_mad24(w, pj, __mul24(65536 - w, pk)) >> 16;              // 17-bit x 12-bit products
        rsq[i] = sp_squash(T, sp_clamp2k(rep[i]));
      } else if constexpr (c.type == C_SSE) {
        int pq = pred_of(IC<(int)c.a2>{}) + 992;
        pq = min(max(pq, 0), 1983);
        const int wt = pq & 63;
        pq >>= 6;
        // entries pq and pq + 1 of the row: 4 per lane, lane (pq >> 2) of this block holds entry pq as component pq & 3
        const int q0 = pq & 3, q1 = (pq + 1) & 3;
        const unsigned s0 = q0 == 0 ? ssev[c.slot].x : (q0 == 1 ? ssev[c.slot].y : (q0 == 2 ? ssev[c.slot].z : ssev[c.slot].w));
        const unsigned s1 = q1 == 0 ? ssev[c.slot].x : (q1 == 1 ? ssev[c.slot].y : (q1 == 2 ? ssev[c.slot].z : ssev[c.slot].w));
        const int base = (int)(bw * 8u);
        const unsigned e0 = __shfl(s0, base + (pq >> 2)), e1 = __shfl(s1, base + ((pq + 1) >> 2));
        rep[i] = sp_stretch(T, ((e0 >> 10) * (unsigned)(64 - wt) + (e1 >> 10) * (unsigned)wt) >> 13);
        ssecx[c.slot] += (unsigned)(pq + (wt >> 5));
        ssetr[c.slot] = (wt >> 5) ? e1 : e0;
        ssedt[c.slot] = (unsigned)T.dt[ssetr[c.slot] & 0x3ffu];
      }
    });
    constexpr unsigned tlast = Chain::comp[N - 1].type;
    if constexpr (tlast == C_MIX2) return (unsigned)rsq[N - 1];
    else return (unsigned)sp_squash(T, sp_clamp2k(pred_of(IC<N - 1>{})));
  };

  // ---- after [B]: update of the tail's components (Predictor::update0 cases CM, MATCH, MIX2, SSE)
  auto update = [&](auto bitc, int y) __attribute__((always_inline)) {
    constexpr int B = decltype(bitc)::value;
    constexpr bool byte_done = B == 7;
    const unsigned count = v0 & 0x3ffu;
    const int yq = y * 32767;
    const int errcm = yq - (int)(v0 >> 17);
    const unsigned cm_new = v0 + ((unsigned)__mul24(errcm, (int)dtv) & 0xFFFFFC00u) + (count < limit ? 1u : 0u);
    G32(goff + 4u * gidx) = cm_new;                           // (lanes that are no CM: their dummy)
    ra = (is_match && (int)rc != y) ? 0u : ra;
    if (byte_done && is_match) {
      // Predictor::update0 case MATCH at the end of a byte (libzpaq.cpp:1992-2006), from registers
      const unsigned mask = mask1;
      const unsigned byte = (unsigned)(c8 * 2 + y) & 255u;
      const unsigned wpos = rlimit & mask;                     // where this byte goes
      G8(off1 + wpos) = (unsigned char)byte;
      mhist = mhist << 8 | byte;
      rlimit = (rlimit + 1) & mask;
      const unsigned eo = off0 + 4u * (h & mask0);
      bool fresh = false;
      if (ra == 0) {
        rb = rlimit - mcmv;
        if (rb & mask) {
          const unsigned cpos = (mcmv - 8u) & mask;
          const bool wraps = cpos + 8u > mask + 1u || mask < 15u;
          const bool overlap = ((mcmv - 1u - wpos) & mask) < 8u;
          unsigned m = 0;
          if (!wraps && !overlap) {
            const unsigned long long diff = __builtin_bswap64(mcand) ^ mhist;
            m = diff ? (unsigned)(__builtin_ctzll(diff) >> 3) : 8u;
          }
          ra = m;
          if (wraps || overlap || m == 8u)
            while (ra < 255 && G8(off1 + ((rlimit - ra - 1) & mask)) == G8(off1 + ((rlimit - ra - rb - 1) & mask))) ++ra;
        }
        fresh = true;
      } else ra += ra < 255;
      G32(eo) = rlimit;
      if (ra != 0) {
        const unsigned ppos = (rlimit - rb) & mask;
        const unsigned early = fresh ? mcand_at : mcont;
        mpred = ppos == wpos ? byte : early;
        mdd = T.dt2k[ra];
      }
    }
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr CompK c = Chain::comp[i];
      if constexpr (c.type == C_MIX2) {
        constexpr int k2 = team_mix2_slot<Chain>(i);
        const int err2 = __mul24(yq - rsq[i], (int)c.a4) >> 5;
        const int w2 = min(max(m2w[k2] + (sp_mad24(err2, m2d[k2], 1 << 12) >> 13), 0), 65535);   // 19-bit x 13-bit
        if constexpr (c.mask0 == 0u) m2w[k2] = w2;
        else *(g_i32*)(arena + m2st[k2] + ((4u * m2idx[k2]) & m_lane0)) = w2;
      } else if constexpr (c.type == C_SSE) {
        const unsigned e = ssecx[c.slot];
        const unsigned v = ssetr[c.slot];
        const unsigned cnt = v & 0x3ffu;
        const int err = yq - (int)(v >> 17);
        const unsigned prod = (unsigned)__mul24(err, (int)ssedt[c.slot]);
        const unsigned nv = v + (prod & 0xFFFFFC00u) + (cnt < c.limit ? 1u : 0u);
        *(g_u32*)(arena + ssest[c.slot] + ((4u * (e & c.mask0)) & m_lane0)) = nv;
      }
    });
    ylast = y;
  };

  // ---- the coder of this block (Decoder::decode, libzpaq.cpp:2159-2181): every lane of the block alike
  unsigned rp = 0, nout = 0, curr = 0;
  bool run = live;
  bool eos = false;
  if (live && (delta64 >> 32) != 0) { status = 8; run = false; }   // arenas of the workgroup not inside a 4 GiB window
  const unsigned in_lim = (in_len + 63u) & ~63u;
  unsigned long long iwin = 0, if0 = 0, if1 = 0;
  unsigned iavail = 0, ifpos = 0;
  auto in_fetch = [&]() __attribute__((always_inline)) {
    ifpos = in_lim >= 16u ? min(rp, in_lim - 16u) : 0u;
    if0 = *(g_u64u*)(in_ptr + ifpos);
    if1 = *(g_u64u*)(in_ptr + ifpos + 8u);
  };
  auto in_window = [&]() __attribute__((always_inline)) {
    const unsigned o = rp - ifpos;
    const unsigned long long b0 = __builtin_bswap64(if0), b1 = __builtin_bswap64(if1);
    const unsigned sh = (o & 7u) * 8u;
    const unsigned long long lo = sh ? (b0 << sh) | (b1 >> (64u - sh)) : b0;
    const unsigned long long hi = b1 << sh;
    iwin = o < 8u ? lo : hi;
    iavail = in_lim < 16u ? 0u : (o < 8u ? 8u : (o < 16u ? 16u - o : 0u));
  };
  auto in_byte = [&]() __attribute__((always_inline)) -> unsigned {
    unsigned v;
    if (iavail) {
      v = (unsigned)(iwin >> 56);
      iwin <<= 8;
      --iavail;
    } else {
      v = in_ptr[rp];
      ZPQ_OPAQUE(v);
    }
    ++rp;
    return v;
  };
  unsigned dmid = 0;
  auto decode_bit = [&](unsigned pr) __attribute__((always_inline)) -> int {
    const bool bad = run && (curr < low || curr > high);
    status = bad ? 2 : status;
    run = run && !bad;
    dmid = low + (unsigned)(((unsigned long long)(high - low) * pr) >> 16);
    return (run && curr <= dmid) ? 1 : 0;
  };
  auto decode_shift = [&](int y) __attribute__((always_inline)) {
    if (!run) return;
    if (y) high = dmid; else low = dmid + 1;
    while ((high ^ low) < 0x1000000u) {
      high = high << 8 | 255u;
      low = low << 8;
      low += (low == 0);
      if (rp >= in_len) { status = 6; run = false; break; }
      curr = curr << 8 | in_byte();
    }
  };
  auto decode = [&](unsigned pr) __attribute__((always_inline)) -> int {
    const int y = decode_bit(pr);
    decode_shift(y);
    return y;
  };
  auto run_hcomp = [&](unsigned input) __attribute__((always_inline)) -> int {
#ifndef ZPQ_EMU
    int e = 0;
    if (run) e = Chain::hcomp(input, vm_b, vm_c, vm_d, vm_f, vm_M, vm_H, vm_R);
    return e;
#else
    int e = 0;
    if (run && r == 0) e = Chain::hcomp(input, vm_b, vm_c, vm_d, vm_f, vm_M, vm_H, vm_R);
    return __shfl(e, (int)(bw * 8u));
#endif
  };
  auto any_running = [&]() __attribute__((always_inline)) -> bool {
    unsigned rr = 0;
    for (int i = 0; i < kTeamBlocks; ++i) rr |= *(lds_u32*)(lds0 + (unsigned)(i * kRegion + kTeamRun));
    return rr != 0;
  };

  if (run && in_len) { in_fetch(); in_window(); }
  for (int i = 0; i < 4; ++i) {
    if (!run) break;
    if (rp >= in_len) { status = 6; run = false; break; }
    curr = curr << 8 | in_byte();
  }
  if (run && nout >= out_cap) run = false;
  if (r == 0) L32((unsigned)kTeamRun) = run ? 1u : 0u;
  ZPQ_TEAM_BARRIER();                                        // [S]
  bool any = any_running();
  if (run) in_fetch();
  while (any) {
    int ch = 1;
    if (run) { in_window(); in_fetch(); }
    const int flag = decode(0);                               // end-of-stream flag, coded with p = 0
    if (run && flag) { eos = true; if (curr != 0) status = 2; run = false; }
    static_for<0, 8>([&](auto bitc) __attribute__((always_inline)) {
      constexpr int B = decltype(bitc)::value;
      pre(bitc);
      ZPQ_TEAM_BARRIER();                                    // [A] CM / MATCH and the row components are in X: the mix wavefronts work
      ZPQ_TEAM_BARRIER();                                    // [A2] the MIX outputs are in X
      const unsigned pr = chain() * 2u + 1u;
      const int y = decode_bit(pr);
      if (r == 0) L32((unsigned)kTeamY) = (unsigned)y;
      ZPQ_TEAM_BARRIER();                                    // [B]
      decode_shift(y);
      ch += ch + y;
      if constexpr (B != 7) {
        update(bitc, y);
        c8 += c8 + y;
        if (run) ++steps;
        if constexpr (B == 3) hmap4 = (hmap4 & 0xf) << 5 | y << 4 | 1;
        else hmap4 = (hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + y) & 0xf);
      } else {
        // HCOMP needs the byte, not the trained components: it runs first, and the last bit's update behind [C], while the
        // row wavefronts are out for the next byte's rows
        const int e = run_hcomp((unsigned)(c8 + c8 + y - 256));
        if (run && e) { status = e; run = false; }
        if (run) {
          if (r == 0) out_ptr[nout] = (unsigned char)(ch - 256);
          ++nout;
          ++steps;
          if (nout >= out_cap) run = false;
        }
        if (r == 0) L32((unsigned)kTeamRun) = run ? 1u : 0u;
        ZPQ_TEAM_BARRIER();                                  // [C]
        update(bitc, y);
        any = any_running();
        h = vm_H[(unsigned)cidx & Chain::HMASK];
        static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          constexpr CompK c = Chain::comp[i];
          if constexpr (c.type == C_SSE) hsse[c.slot] = vm_H[(unsigned)i & Chain::HMASK];
          if constexpr (c.type == C_MIX2 && c.mask0 != 0u) hm2[team_mix2_slot<Chain>(i)] = vm_H[(unsigned)i & Chain::HMASK];
        });
        hmap4 = 1;
        c8 = 1;
      }
    });
  }
  if (r == 0 && live) {
    res[rslot].out_len = nout;
    res[rslot].consumed = eos ? rp : 0;
    res[rslot].status = status;
    res[rslot].steps = steps;
  }
}

// =====================================================================================================================
template <class Chain>
__device__ __forceinline__ void spec_team_decode_body(const BlockJob* jobs, BlockResult* res, unsigned nblocks,
                                                      const DeviceTables* tb) {
  constexpr int N = Chain::N;
  static_assert(team_map<Chain>().ok, "chain not for the lockstep decoder");
  static_assert(Chain::WAVES == 8, "the chain's LDS plan must be the one of the 8-blocks-per-workgroup shape");
  constexpr int kRegion = team_block_lds_bytes();
  constexpr int RL = team_row_lanes<Chain>();
  constexpr int NRW = kTeamBlocks / (64 / RL);                // row wavefronts
  static_assert((int)sizeof(TeamTables) + kTeamBlocks * kRegion <= kSpecLdsBudget, "LDS budget");

  __shared__ TeamTables T;
  __shared__ __attribute__((aligned(16))) unsigned char wave_lds[kTeamBlocks][kRegion];
  for (unsigned i = threadIdx.x; i < 2016u; i += blockDim.x) T.stretch_cb[i] = tb->stretch_cb[i];
  for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) T.stretch_top[i] = tb->stretch_top[i];
  for (unsigned i = threadIdx.x; i < 1344u; i += blockDim.x) T.squash_mid[i] = tb->squash[1376u + i];
  for (unsigned i = threadIdx.x; i < 1024u; i += blockDim.x) { T.dt[i] = tb->dt[i]; T.ns[i] = tb->ns[i]; }
  for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) T.dt2k[i] = (uint16_t)tb->dt2k[i];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  lds_u8* const lds0 = (lds_u8*)&wave_lds[0][0];
  const unsigned wg0 = blockIdx.x * (unsigned)kTeamBlocks;
  // side tables -> LDS, H cleared, dummy and exchange words zeroed: the threads of the workgroup, block by block
  for (int bw = 0; bw < kTeamBlocks; ++bw) {
    lds_u8* const wl = lds0 + (unsigned)(bw * kRegion);
    const unsigned b = wg0 + (unsigned)bw;
    if (b < nblocks) {
      const g_u8* const arena_b = (const g_u8*)jobs[b].arena;
      static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
        constexpr CompK c = Chain::comp[decltype(ic)::value];
        if constexpr (c.lds >= 0 && (c.type == C_ICM || c.type == C_ISSE)) {
          // packed: ICM  u16 lo[256] | u8 hi[256];  ISSE  (u16 w0 lo, u16 w1 lo)[256] | u8 (w0 hi nibble | w1 hi nibble << 4)[256]
          const g_u32* src = (const g_u32*)(arena_b + (unsigned)c.t0);
          lds_u8* const tab = wl + c.lds;
          for (unsigned k = threadIdx.x; k < 256u; k += blockDim.x) {
            if constexpr (c.type == C_ICM) {
              const unsigned v = src[k];
              *(lds_u16*)(tab + 2u * k) = (unsigned short)v;
              tab[512u + k] = (unsigned char)(v >> 16);
            } else {
              const unsigned w0 = src[2u * k], w1 = src[2u * k + 1u];
              *(lds_u32*)(tab + 4u * k) = (w0 & 0xFFFFu) | (w1 << 16);
              tab[1024u + k] = (unsigned char)(((w0 >> 16) & 15u) | (((w1 >> 16) & 15u) << 4));
            }
          }
        }
      });
      for (unsigned k = threadIdx.x; k <= Chain::HMASK; k += blockDim.x) ((lds_u32*)(wl + Chain::H_LDS))[k] = 0;
    }
    for (unsigned k = threadIdx.x; k < 128u; k += blockDim.x) ((lds_u32*)(wl + (kRegion - 512)))[k] = 0;
  }
  __syncthreads();
  if constexpr (team_tail_ok<Chain>()) {
    if (wave < NRW) team_rows<Chain>(T, lds0, jobs, nblocks, wave, lane);
    else if (wave < NRW + kTeamBlocks / 2) team_mix<Chain>(T, lds0, jobs, nblocks, wave - NRW, lane);
    else team_tail<Chain>(T, lds0, jobs, res, nblocks, lane);
  } else {
    if (wave < NRW) team_rows<Chain>(T, lds0, jobs, nblocks, wave, lane);
    else team_mixers<Chain>(T, lds0, jobs, res, nblocks, wave - NRW, lane);
  }
}

}  // namespace zpq
