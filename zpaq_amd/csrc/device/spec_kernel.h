// Specialised wave-parallel coder: the kernel template that a GENERATED
// translation unit instantiates for ONE block header (the GPU analogue of the
// reference's per-header x86 JIT, libzpaq.cpp:3824-4583 / 3231-3811).
//
// The generator (host/codegen.cpp) emits a `Chain` type holding the COMP list as
// constexpr data plus the HCOMP program translated to straight-line HIP C++;
// everything below is ordinary hand-written HIP for gfx950 that the compiler
// folds against those constants.
//
// Mapping: one ZPAQ block per wavefront, component i on lane i (n <= 64), four
// blocks per workgroup.  What changed against the generic kernel (model_wave.h):
//   * the small hot tables live in LDS: half of stretch (mirrored), the
//     non-trivial part of squash, dt, the state table, and -- per block -- the
//     ICM/ISSE bit-history->probability side tables and HCOMP's H array;
//   * the 16-byte bit-history row of every ICM/ISSE is cached in 4 VGPRs for the
//     4 bits of a nibble: one probe (3 x 16 B in one 64-B line) + one 16-B
//     write-back per nibble instead of a byte load/store per bit;
//   * all global loads of a bit (CM word, MIX2 weight, MIX rows, SSE row) have
//     addresses that depend only on (h, c8, hmap4): they are issued together at
//     the top of predict and overlap the LDS work; MIX weights and the SSE row
//     stay in registers for update (no re-load);
//   * the dependent chain (ISSE/AVG/MIX2/MIX/SSE) is unrolled at compile time
//     with literal lane numbers (v_readlane / DPP), no descriptor fetches;
//   * HCOMP runs as compiled code, not through an interpreter.
// Integer arithmetic is bit-exact with Predictor::predict0/update0.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "layout.h"

namespace zpq {

// ---- compile-time description of one component (emitted by the generator) ----
struct CompK {
  unsigned type, a1, a2, a3, a4, a5;
  unsigned limit, mask0, mask1;
  unsigned long long t0, t1;   // arena byte offsets
  int lds;                     // byte offset of the side table in the wave's LDS region, or -1
  int slot;                    // ordinal among MIX (resp. SSE) components, else -1
};

constexpr int kSpecWaves = 4;                  // blocks per workgroup
constexpr int kLdsBudget = 163840 - 256;       // gfx950: 160 KiB per workgroup, a little slack

struct SpecTables {                            // shared by the 4 waves of a workgroup
  int16_t stretch_hi[16384];                   // stretch(x) for x >= 16384; mirrored below
  uint16_t squash_mid[1344];                   // squash index 1376..2719
  int32_t dt[1024];
  uint16_t dt2k[256];
  uint8_t ns[1024];
};
static_assert(sizeof(SpecTables) == kSpecTablesBytes, "host codegen and device disagree on LDS tables");
constexpr int kSpecWaveLds = kSpecWaveLdsBytes;
static_assert((int)sizeof(SpecTables) + kSpecWaves * kSpecWaveLds <= kLdsBudget, "LDS budget");

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
__device__ __forceinline__ unsigned long long sp_uni64(unsigned long long v) {
  return (unsigned long long)sp_uni((unsigned)(v >> 32)) << 32 | sp_uni((unsigned)v);
}

// 64-lane integer sum (DPP row scans + row broadcasts); result wave-uniform.
__device__ __forceinline__ int sp_wave_sum(int x) {
  int v = x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
  return sp_rl(v, 63);
}

// LDS-qualified views: loads/stores through these are ds_read/ds_write, never flat.
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) int2 lds_i2;

// Explicit global-memory views of the arena: pointers loaded from the job
// descriptor are generic to the compiler, and flat_* accesses would tie vmcnt
// and lgkmcnt together (every LDS wait would also wait for HBM).
typedef __attribute__((address_space(1))) unsigned char g_u8;
typedef __attribute__((address_space(1))) unsigned short g_u16;
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

template <class Chain, int I>
struct Dep;   // forward

// compile-time loop: f(IC<B>{}), f(IC<B+1>{}), ...
template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}

// ---------------------------------------------------------------------------
template <class Chain, bool DEC>
__device__ __forceinline__ void spec_kernel_body(const BlockJob* jobs, BlockResult* res, unsigned nblocks,
                                                 const DeviceTables* tb) {
  constexpr int N = Chain::N;
  static_assert(N >= 1 && N <= 64, "specialised kernel handles 1..64 components");
  constexpr int NMIX = Chain::NMIX > 0 ? Chain::NMIX : 1;
  constexpr int NSSE = Chain::NSSE > 0 ? Chain::NSSE : 1;

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
  lds_u8* const wl = (lds_u8*)&wave_lds[wave][0];

  // ---- per-lane component constants (selected from the constexpr chain) ----
  unsigned type = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, limit = 0, mask0 = 0, mask1 = 0, sizebits = 0;
  g_u8* t0 = nullptr;
  g_u8* t1 = nullptr;
  int ldsoff = -1;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (lane == i) {
      const CompK c = Chain::comp[i];
      type = c.type; a2 = c.a2; a3 = c.a3; a4 = c.a4; a5 = c.a5;
      limit = c.limit; mask0 = c.mask0; mask1 = c.mask1; sizebits = c.a1 + 2;
      t0 = arena + c.t0;
      t1 = arena + c.t1;
      ldsoff = c.lds;
    }
  }
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

  // ---- per-lane mutable state ----
  unsigned cxt = 0, ra = 0, rb = 0, rc = 0, rlimit = 0;   // Component::cxt,a,b,c,limit
  unsigned h = 0;
  int p = 0;
  static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (Chain::comp[i].type == C_CONS) { if (lane == i) p = ((int)Chain::comp[i].a1 - 128) * 4; }
  });
  // table words read in predict and reused by update:
  //   CM: v0 = cm word | ICM: v0 = side-table word | ISSE: v0,v1 = weights | MIX2: v0 = weight
  unsigned v0 = 0, v1 = 0;
  unsigned row0 = 0, row1 = 0, row2 = 0, row3 = 0;   // cached 16-byte bit-history row
  unsigned rowoff = 0xFFFFFFFFu;   // offset of the cached row in ht, or none
  int mixw[NMIX];              // lane t holds weight t of each MIX row
  unsigned mixrow[NMIX];       // element index of the selected row (uniform)
  unsigned ssev[NSSE];         // lane t (<32) holds entry t of each SSE row
  unsigned ssecx[NSSE];        // element index of the trained SSE entry (uniform)
#pragma unroll
  for (int k = 0; k < NMIX; ++k) { mixw[k] = 0; mixrow[k] = 0; }
#pragma unroll
  for (int k = 0; k < NSSE; ++k) { ssev[k] = 0; ssecx[k] = 0; }

  int c8 = 1, hmap4 = 1;
  unsigned low = 1, high = 0xFFFFFFFFu;
  unsigned steps = 0;
  int status = 0;

  const bool is_icm = type == C_ICM, is_isse = type == C_ISSE;
  const bool has_row = is_icm || is_isse;

  // ---------------------------------------------------------------- predict
  auto predict = [&]() __attribute__((always_inline)) -> unsigned {
    const bool nib = (c8 == 1) || ((c8 & 0xf0) == 16);
    const int slot = hmap4 & 15;
    // (A) issue this bit's global loads: addresses depend only on (h, c8, hmap4)
    if (type == C_CM) {
      cxt = (h ^ (unsigned)hmap4) & mask0;
      v0 = ((const g_u32*)t0)[cxt];
    } else if (type == C_MIX2) {
      cxt = (h + (unsigned)(c8 & (int)a5)) & mask0;
      v0 = ((const g_u16*)t0)[cxt];
    }
    static_for<0, N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr CompK c = Chain::comp[i];
      if constexpr (c.type == C_MIX) {
        const unsigned hi = sp_rlu(h, i);
        const unsigned r = ((hi + (unsigned)(c8 & (int)c.a5)) & c.mask0) * c.a3;
        mixrow[c.slot] = r;
        if (lane < (int)c.a3) mixw[c.slot] = ((const g_i32*)(arena + c.t0))[r + lane];
      } else if constexpr (c.type == C_SSE) {
        const unsigned hi = sp_rlu(h, i);
        const unsigned cx0 = ((hi + (unsigned)c8) * 32u) & c.mask0;
        ssecx[c.slot] = cx0;
        if (lane < 32) ssev[c.slot] = ((const g_u32*)(arena + c.t0))[cx0 + lane];
      }
    });
    // (B) ICM / ISSE: bit-history row in registers, side table in LDS
    if (has_row) {
      if (nib) {
        if (rowoff != 0xFFFFFFFFu) *(g_u128*)(t1 + rowoff) = make_uint4(row0, row1, row2, row3);   // write back
        const unsigned cx = h + 16u * (unsigned)c8;
        const unsigned chk = (cx >> sizebits) & 255u;
        const unsigned h0 = (cx * 16u) & (mask1 - 15u);
        const uint4 r0 = *(const g_u128*)(t1 + h0);
        const uint4 r1 = *(const g_u128*)(t1 + (h0 ^ 16u));
        const uint4 r2 = *(const g_u128*)(t1 + (h0 ^ 32u));
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
      }
      cxt = row_get(row0, row1, row2, row3, slot);                         // bit history
      // ICM: one word at [bh]; ISSE: two words at [2*bh], [2*bh+1]
      const unsigned e0 = is_icm ? cxt : 2u * cxt, e1 = is_icm ? cxt : 2u * cxt + 1u;
      if (ldsoff >= 0) {
        const lds_u32* q = (const lds_u32*)(wl + ldsoff);
        v0 = q[e0];
        v1 = q[e1];
      } else {
        const g_u32* q = (const g_u32*)t0;
        v0 = q[e0];
        v1 = q[e1];
      }
      if (is_icm) p = sp_stretch(T, v0 >> 8);
    } else if (type == C_MATCH) {
      if (ra == 0) p = 0;
      else {
        rc = (t1[(rlimit - rb) & mask1] >> (7 - cxt)) & 1u;
        const int dd = T.dt2k[ra];
        p = sp_stretch(T, (unsigned)((rc ? -dd : dd) & 32767));
      }
    } else if (type == C_CM) {
      p = sp_stretch(T, v0 >> 17);
    }
    // (C) dependent components, in index order, unrolled with literal lanes
    Dep<Chain, 0>::predict(T, lane, c8, p, (int)v0, (int)v1, mixw, ssev, ssecx, cxt);
    return sp_uni((unsigned)sp_squash(T, sp_rl(p, N - 1)));
  };

  // ----------------------------------------------------------------- update
  auto update = [&](int y) __attribute__((always_inline)) {
    const int slot = hmap4 & 15;
    const int pj = __shfl(p, (int)(a2 & 63));     // ISSE j / MIX2 j
    const int pk = __shfl(p, (int)(a3 & 63));     // MIX2 k
    if (type == C_CM) {
      const unsigned count = v0 & 0x3ffu;
      const int err = y * 32767 - (int)(v0 >> 17);
      const unsigned prod = (unsigned)err * (unsigned)T.dt[count];
      ((g_u32*)t0)[cxt] = v0 + (prod & 0xFFFFFC00u) + (count < limit ? 1u : 0u);
    } else if (is_icm) {
      row_set(row0, row1, row2, row3, slot, T.ns[cxt * 4 + y]);
      const unsigned nv = v0 + (unsigned)((int)((unsigned)(y * 32767) - (v0 >> 8)) >> 2);
      if (ldsoff >= 0) ((lds_u32*)(wl + ldsoff))[cxt] = nv; else ((g_u32*)t0)[cxt] = nv;
    } else if (is_isse) {
      const int err = y * 32767 - sp_squash(T, p);
      const unsigned nw0 = (unsigned)sp_clamp512k((int)v0 + ((err * pj + (1 << 12)) >> 13));
      const unsigned nw1 = (unsigned)sp_clamp512k((int)v1 + ((err + 16) >> 5));
      if (ldsoff >= 0) { lds_u32* q = (lds_u32*)(wl + ldsoff) + 2 * cxt; q[0] = nw0; q[1] = nw1; }
      else { g_u32* q = (g_u32*)t0 + 2 * cxt; q[0] = nw0; q[1] = nw1; }
      row_set(row0, row1, row2, row3, slot, T.ns[cxt * 4 + y]);
    } else if (type == C_MATCH) {
      g_u8* buf = t1;
      const unsigned mask = mask1;
      if ((int)rc != y) ra = 0;
      buf[rlimit & mask] = (unsigned char)(buf[rlimit & mask] * 2 + y);
      if (++cxt == 8) {
        cxt = 0;
        rlimit = (rlimit + 1) & mask;
        g_u32* e = (g_u32*)t0 + (h & mask0);
        if (ra == 0) {
          rb = rlimit - *e;
          if (rb & mask)
            while (ra < 255 && buf[(rlimit - ra - 1) & mask] == buf[(rlimit - ra - rb - 1) & mask]) ++ra;
        } else ra += ra < 255;
        *e = rlimit;
      }
    } else if (type == C_MIX2) {
      const int err = ((y * 32767 - sp_squash(T, p)) * (int)a4) >> 5;
      int w = (int)v0 + ((err * (pj - pk) + (1 << 12)) >> 13);
      w = min(max(w, 0), 65535);
      ((g_u16*)t0)[cxt] = (unsigned short)w;
    }
    Dep<Chain, 0>::update(T, arena, lane, y, p, mixw, mixrow, ssev, ssecx);
  };

  auto after_bit = [&](int y) __attribute__((always_inline)) -> int {   // c8 / hmap4 bookkeeping (libzpaq.cpp:2055-2065)
    update(y);
    c8 += c8 + y;
    if (c8 >= 256) {
      const int e = Chain::hcomp((unsigned)(c8 - 256), vm_b, vm_c, vm_d, vm_f, vm_M, vm_H, vm_R);
      if (e) return e;
      hmap4 = 1;
      c8 = 1;
      h = vm_H[(unsigned)lane & Chain::HMASK];
    } else if (c8 >= 16 && c8 < 32) {
      hmap4 = (hmap4 & 0xf) << 5 | y << 4 | 1;
    } else {
      hmap4 = (hmap4 & 0x1f0) | (((hmap4 & 0xf) * 2 + y) & 0xf);
    }
    return 0;
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
      encode(0, 0);
      for (int i = 7; i >= 0; --i) {
        const unsigned pr = predict();
        const int y = (ch >> i) & 1;
        encode(y, pr * 2 + 1);
        status = after_bit(y);
        ++steps;
        if (status) break;
      }
    }
    if (!status) encode(1, 0);
    if (!status && n > job.out_cap) status = 3;
    if (lane == 0) { res[b].out_len = n; res[b].consumed = job.in_len; }
  } else {
    unsigned rp = 0, n = 0, curr = 0;
    bool eos = false;
    for (int i = 0; i < 4; ++i) {
      if (rp >= job.in_len) { status = 6; break; }
      curr = curr << 8 | sp_uni(in_ptr[rp++]);
    }
    while (!status && !eos && n < job.out_cap) {
      int ch = 1;
      for (int bit = -1; bit < 8; ++bit) {
        unsigned pr = 0;
        if (bit >= 0) pr = predict() * 2 + 1;
        if (curr < low || curr > high) { status = 2; break; }
        const unsigned mid = low + (unsigned)(((unsigned long long)(high - low) * pr) >> 16);
        int y;
        if (curr <= mid) { y = 1; high = mid; } else { y = 0; low = mid + 1; }
        while ((high ^ low) < 0x1000000u) {
          high = high << 8 | 255u;
          low = low << 8;
          low += (low == 0);
          if (rp >= job.in_len) { status = 6; break; }
          curr = curr << 8 | sp_uni(in_ptr[rp++]);
        }
        if (status) break;
        if (bit < 0) {
          if (y) { eos = true; if (curr != 0) status = 2; break; }
        } else {
          ch += ch + y;
          status = after_bit(y);
          ++steps;
          if (status) break;
        }
      }
      if (status || eos) break;
      if (lane == 0) out_ptr[n] = (unsigned char)(ch - 256);
      ++n;
    }
    if (lane == 0) { res[b].out_len = n; res[b].consumed = eos ? rp : 0; }
  }
  if (lane == 0) { res[b].status = status; res[b].steps = steps; }
}

// ---- compile-time walk over the dependent components -------------------------
template <class Chain, int I>
struct Dep {
  template <int NM, int NS>
  static __device__ __forceinline__ void predict(const SpecTables& T, int lane, int c8, int& p, int w0, int w1,
                                                 int (&mixw)[NM], unsigned (&ssev)[NS], unsigned (&ssecx)[NS],
                                                 unsigned& cxt) {
    if constexpr (I < Chain::N) {
      constexpr CompK c = Chain::comp[I];
      if constexpr (c.type == C_ISSE) {
        const int pj = sp_rl(p, (int)c.a2);
        const int val = sp_clamp2k((w0 * pj + w1 * 64) >> 16);
        p = lane == I ? val : p;
      } else if constexpr (c.type == C_AVG) {
        const int pj = sp_rl(p, (int)c.a1), pk = sp_rl(p, (int)c.a2);
        const int val = (pj * (int)c.a3 + pk * (256 - (int)c.a3)) >> 8;
        p = lane == I ? val : p;
      } else if constexpr (c.type == C_MIX2) {
        const int pj = sp_rl(p, (int)c.a2), pk = sp_rl(p, (int)c.a3);
        const int val = (w0 * pj + (65536 - w0) * pk) >> 16;
        p = lane == I ? val : p;
      } else if constexpr (c.type == C_MIX) {
        // inputs p[j..j+m-1] sit in lanes j..j+m-1; weight t sits in lane t
        int pin = p;
        if constexpr (c.a2 != 0) pin = __shfl(p, (lane + (int)c.a2) & 63);
        const int x = lane < (int)c.a3 ? (mixw[c.slot] >> 8) * pin : 0;
        const int val = sp_clamp2k(sp_wave_sum(x) >> 8);
        p = lane == I ? val : p;
      } else if constexpr (c.type == C_SSE) {
        int pq = sp_rl(p, (int)c.a2) + 992;
        pq = min(max(pq, 0), 1983);
        const int wt = pq & 63;
        pq >>= 6;
        const unsigned e0 = sp_rlu(ssev[c.slot], pq), e1 = sp_rlu(ssev[c.slot], pq + 1);
        const int val = sp_stretch(T, ((e0 >> 10) * (unsigned)(64 - wt) + (e1 >> 10) * (unsigned)wt) >> 13);
        p = lane == I ? val : p;
        ssecx[c.slot] += (unsigned)(pq + (wt >> 5));        // element trained in update
      }
      Dep<Chain, I + 1>::predict(T, lane, c8, p, w0, w1, mixw, ssev, ssecx, cxt);
    }
  }

  template <int NM, int NS>
  static __device__ __forceinline__ void update(const SpecTables& T, g_u8* arena, int lane, int y, int p,
                                                int (&mixw)[NM], unsigned (&mixrow)[NM], unsigned (&ssev)[NS],
                                                unsigned (&ssecx)[NS]) {
    if constexpr (I < Chain::N) {
      constexpr CompK c = Chain::comp[I];
      if constexpr (c.type == C_MIX) {
        const int err = ((y * 32767 - sp_squash(T, sp_rl(p, I))) * (int)c.a4) >> 4;
        int pin = p;
        if constexpr (c.a2 != 0) pin = __shfl(p, (lane + (int)c.a2) & 63);
        if (lane < (int)c.a3) {
          const int w = sp_clamp512k(mixw[c.slot] + ((err * pin + (1 << 12)) >> 13));
          ((g_i32*)(arena + c.t0))[mixrow[c.slot] + lane] = w;
        }
      } else if constexpr (c.type == C_SSE) {
        // Predictor::train on cm[cxt]; the word is still in lane (cxt & 31) of the row registers
        const unsigned e = ssecx[c.slot];
        const unsigned v = sp_rlu(ssev[c.slot], (int)(e & 31u));
        const unsigned count = v & 0x3ffu;
        const int err = y * 32767 - (int)(v >> 17);
        const unsigned prod = (unsigned)err * (unsigned)T.dt[count];
        const unsigned nv = v + (prod & 0xFFFFFC00u) + (count < c.limit ? 1u : 0u);
        if (lane == 0) ((g_u32*)(arena + c.t0))[e & c.mask0] = nv;
      }
      Dep<Chain, I + 1>::update(T, arena, lane, y, p, mixw, mixrow, ssev, ssecx);
    }
  }
};

}  // namespace zpq
