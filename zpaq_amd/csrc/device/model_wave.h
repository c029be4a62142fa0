// Wave-parallel coder for gfx950: ONE ZPAQ BLOCK PER WAVEFRONT, one model
// component per lane (n <= 64), four blocks per workgroup sharing the
// predictor's constant tables in LDS.
//
// Per coded bit:
//   phase 1  every lane evaluates its own context-only component (CM, ICM,
//            MATCH) or fetches what its dependent component will need (ISSE
//            weights, MIX2 weight, MIX row index) -- all lanes in parallel;
//   phase 2  components whose input is an earlier p[j] (ISSE, AVG, MIX2, MIX,
//            SSE) are resolved in index order with wave-uniform control flow;
//            values travel between lanes with v_readlane / DPP, MIX is a
//            lane-parallel dot product (coalesced weight-row load + DPP
//            reduction);
//   coding   the 32-bit range coder runs wave-uniformly;
//   update   every lane trains its own component in parallel; MIX rows are
//            updated lane-parallel (coalesced read-modify-write);
//   per byte the HCOMP program runs wave-uniformly and lanes reload h[lane].
//
// The integer arithmetic is bit-exact with Predictor::predict0/update0
// (libzpaq.cpp:1854-2066).  No MFMA: there is no dense contraction on this path.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"
#include "model_serial.h"

namespace zpq {

constexpr int kWavesPerGroup = 4;

// Constant tables staged in LDS once per workgroup (78 KiB).
struct LdsTables {
  int16_t stretch[32768];
  uint16_t squash[4096];
  int32_t dt[1024];
  int32_t dt2k[256];
  uint8_t ns[1024];
};

__device__ inline int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ inline uint32_t rlu(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline int unii(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline uint64_t uni64(uint64_t v) {
  return (uint64_t)uni((uint32_t)(v >> 32)) << 32 | uni((uint32_t)v);
}
__device__ inline uint64_t rl64(uint64_t v, int lane) {
  uint32_t lo = rlu((uint32_t)v, lane), hi = rlu((uint32_t)(v >> 32), lane);
  return (uint64_t)hi << 32 | lo;
}

// Sum of x over the 64 lanes, returned wave-uniformly.  DPP row shifts build a
// per-row (16-lane) inclusive scan, row_bcast:15/31 fold the four rows; the
// total lands in lane 63.
__device__ inline int wave_sum(int x) {
  int v = x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2,3
  return rl(v, 63);
}

// Wave-uniform HCOMP interpreter: every lane executes the same instruction on
// identical register values (kept scalar through readfirstlane); M/H/R live in
// the block's arena.
struct WaveVm {
  const uint8_t* prog;
  uint32_t prog_len, hmask, mmask;
  uint32_t* H;
  uint8_t* M;
  uint32_t* R;
  uint32_t A, B, C, D;
  int F;
};

__device__ inline uint32_t wvm_src(WaveVm& s, int k, uint32_t& pc) {
  switch (k) {
    case 0: return s.A;
    case 1: return s.B;
    case 2: return s.C;
    case 3: return s.D;
    case 4: return uni(s.M[s.B & s.mmask]);
    case 5: return uni(s.M[s.C & s.mmask]);
    case 6: return uni(s.H[s.D & s.hmask]);
    default: return uni(s.prog[pc++]);
  }
}
__device__ inline void wvm_dst(WaveVm& s, int g, uint32_t v) {
  switch (g) {
    case 0: s.A = v; break;
    case 1: s.B = v; break;
    case 2: s.C = v; break;
    case 3: s.D = v; break;
    case 4: s.M[s.B & s.mmask] = (uint8_t)v; break;
    case 5: s.M[s.C & s.mmask] = (uint8_t)v; break;
    case 6: s.H[s.D & s.hmask] = v; break;
  }
}

__device__ inline int wvm_run(WaveVm& s, uint32_t input) {
  const uint32_t len = s.prog_len;
  uint32_t pc = 0;
  s.A = input;
  for (uint32_t steps = 0; steps < kMaxVmSteps; ++steps) {
    if (pc >= len) return 5;
    const int op = (int)uni(s.prog[pc++]);
    const int g = op >> 3, k = op & 7;
    if (op < 64) {
      if (g == 7) {
        if (op == 56) return 0;
        else if (op == 57) { }
        else if (op == 59) s.A = (s.A + uni(s.M[s.B & s.mmask]) + 512u) * 773u;
        else if (op == 60) { uint32_t* d = &s.H[s.D & s.hmask]; *d = (uni(*d) + s.A + 512u) * 773u; }
        else if (op == 63) pc += 1 + (int)(int8_t)uni(s.prog[pc]);
        else return 5;
      } else if (k == 7) {
        if (pc >= len) return 5;
        const uint32_t N = uni(s.prog[pc]);
        if (g < 4) { wvm_dst(s, g, uni(s.R[N])); ++pc; }
        else if (g == 4) { if (s.F) pc += 1 + (int)(int8_t)N; else ++pc; }
        else if (g == 5) { if (!s.F) pc += 1 + (int)(int8_t)N; else ++pc; }
        else { s.R[N] = s.A; ++pc; }
      } else {
        if (op == 0 || k > 4) return 5;
        const uint32_t x = wvm_src(s, g, pc);
        if (k == 0) {
          const uint32_t a = s.A;
          if (g == 4 || g == 5) { wvm_dst(s, g, a & 255u); s.A = (a & 0xFFFFFF00u) | x; }
          else { wvm_dst(s, g, a); s.A = x; }
        } else if (k == 1) wvm_dst(s, g, x + 1);
        else if (k == 2) wvm_dst(s, g, x - 1);
        else if (k == 3) wvm_dst(s, g, ~x);
        else wvm_dst(s, g, 0);
      }
    } else if (op < 120) {
      const uint32_t v = wvm_src(s, k, pc);
      wvm_dst(s, g - 8, v);
    } else if (op < 128) {
      return 5;
    } else if (op < 240) {
      const uint32_t v = wvm_src(s, k, pc);
      switch (g - 16) {
        case 0: s.A += v; break;
        case 1: s.A -= v; break;
        case 2: s.A *= v; break;
        case 3: s.A = v ? s.A / v : 0; break;
        case 4: s.A = v ? s.A % v : 0; break;
        case 5: s.A &= v; break;
        case 6: s.A &= ~v; break;
        case 7: s.A |= v; break;
        case 8: s.A ^= v; break;
        case 9: s.A <<= (v & 31); break;
        case 10: s.A >>= (v & 31); break;
        case 11: s.F = (s.A == v); break;
        case 12: s.F = (s.A < v); break;
        default: s.F = (s.A > v); break;
      }
    } else if (op == 255) {
      if (pc + 1 >= len) return 5;
      const uint32_t t = uni(s.prog[pc]) + 256u * uni(s.prog[pc + 1]);
      if (t >= len) return 5;
      pc = t;
    } else return 5;
  }
  return 5;
}

// Per-lane component state (registers).
struct Lane {
  uint32_t type, a1, a2, a3, a4, a5, limit, mask0, mask1, stride;
  uint8_t* t0;
  uint8_t* t1;
  uint32_t cxt, ra, rb, rc, rlimit;   // Component::cxt, a, b, c, limit
  uint32_t h;                         // this component's context hash
  int p;                              // stretch-domain prediction p[i]
  int w0, w1;                         // ISSE weights / MIX2 weight fetched in phase 1
  uint32_t v;                         // CM/ICM/SSE table word read in predict, reused by update
};

struct WaveModel {
  const LdsTables* L;
  uint64_t dep_mask, mix_mask;
  int n, lane;
  int c8, hmap4;
};

__device__ inline void wave_predict_phase1(const WaveModel& m, Lane& s) {
  const LdsTables& L = *m.L;
  const int c8 = m.c8, hmap4 = m.hmap4;
  const bool nib = (c8 == 1) || ((c8 & 0xf0) == 16);
  switch (s.type) {
    case C_CM: {
      s.cxt = (s.h ^ (uint32_t)hmap4) & s.mask0;
      s.v = ((const uint32_t*)s.t0)[s.cxt];
      s.p = L.stretch[s.v >> 17];
      break;
    }
    case C_ICM: {
      if (nib) s.rc = d_find(s.t1, s.mask1, (int)s.a1 + 2, s.h + 16u * (uint32_t)c8);
      s.cxt = s.t1[s.rc + (hmap4 & 15)];
      s.v = ((const uint32_t*)s.t0)[s.cxt];
      s.p = L.stretch[s.v >> 8];
      break;
    }
    case C_ISSE: {
      if (nib) s.rc = d_find(s.t1, s.mask1, (int)s.a1 + 2, s.h + 16u * (uint32_t)c8);
      s.cxt = s.t1[s.rc + (hmap4 & 15)];
      const int2 w = ((const int2*)s.t0)[s.cxt];
      s.w0 = w.x;
      s.w1 = w.y;
      break;
    }
    case C_MATCH: {
      if (s.ra == 0) s.p = 0;
      else {
        s.rc = (s.t1[(s.rlimit - s.rb) & s.mask1] >> (7 - s.cxt)) & 1u;
        const int d = L.dt2k[s.ra];
        s.p = L.stretch[(s.rc ? -d : d) & 32767];
      }
      break;
    }
    case C_MIX2: {
      s.cxt = (s.h + (uint32_t)(c8 & (int)s.a5)) & s.mask0;
      s.w0 = (int)((const uint32_t*)s.t0)[s.cxt];
      break;
    }
    case C_MIX: {
      s.cxt = ((s.h + (uint32_t)(c8 & (int)s.a5)) & s.mask0) * s.stride;
      break;
    }
    default: break;   // CONS fixed; AVG, SSE entirely in phase 2
  }
}

// Resolve dependent components in index order (wave-uniform control flow).
__device__ inline void wave_predict_phase2(const WaveModel& m, Lane& s) {
  const LdsTables& L = *m.L;
  uint64_t deps = m.dep_mask;
  while (deps) {
    const int i = unii(__builtin_ctzll(deps));
    deps &= deps - 1;
    const uint32_t ty = rlu(s.type, i);
    int val = 0;
    uint32_t newcxt = 0;
    if (ty == C_ISSE) {
      const int pj = rl(s.p, (int)rlu(s.a2, i));
      val = d_clamp2k((s.w0 * pj + s.w1 * 64) >> 16);          // valid in lane i (own w0,w1)
    } else if (ty == C_MIX) {
      const int j = (int)rlu(s.a2, i), mm = (int)rlu(s.a3, i);
      const int32_t* row = (const int32_t*)rl64((uint64_t)s.t0, i) + rlu(s.cxt, i);
      const int pin = __shfl(s.p, (j + m.lane) & 63);
      int x = 0;
      if (m.lane < mm) x = (row[m.lane] >> 8) * pin;
      val = d_clamp2k(wave_sum(x) >> 8);
    } else if (ty == C_MIX2) {
      const int pj = rl(s.p, (int)rlu(s.a2, i)), pk = rl(s.p, (int)rlu(s.a3, i));
      val = (s.w0 * pj + (65536 - s.w0) * pk) >> 16;            // valid in lane i (own w0)
    } else if (ty == C_SSE) {
      const uint32_t* cm = (const uint32_t*)rl64((uint64_t)s.t0, i);
      const uint32_t mask0 = rlu(s.mask0, i);
      uint32_t cx = (rlu(s.h, i) + (uint32_t)m.c8) * 32u;
      int pq = rl(s.p, (int)rlu(s.a2, i)) + 992;
      pq = pq < 0 ? 0 : (pq > 1983 ? 1983 : pq);
      const int wt = pq & 63;
      pq >>= 6;
      cx += (uint32_t)pq;
      const uint32_t e0 = uni(cm[cx & mask0]), e1 = uni(cm[(cx + 1) & mask0]);
      val = L.stretch[((e0 >> 10) * (uint32_t)(64 - wt) + (e1 >> 10) * (uint32_t)wt) >> 13];
      newcxt = (cx + (uint32_t)(wt >> 5)) & mask0;
    } else {  // C_AVG
      const int pj = rl(s.p, (int)rlu(s.a1, i)), pk = rl(s.p, (int)rlu(s.a2, i));
      const int wt = (int)rlu(s.a3, i);
      val = (pj * wt + pk * (256 - wt)) >> 8;
    }
    if (m.lane == i) {
      s.p = val;
      if (ty == C_SSE) s.cxt = newcxt;
    }
  }
}

__device__ inline void wave_update(const WaveModel& m, Lane& s, int y) {
  const LdsTables& L = *m.L;
  const int hmap4 = m.hmap4;
  // inputs of ISSE (a2) and MIX2 (a2, a3) live in other lanes
  const int pj = __shfl(s.p, (int)(s.a2 & 63));
  const int pk = __shfl(s.p, (int)(s.a3 & 63));
  switch (s.type) {
    case C_CM:
    case C_SSE: {
      uint32_t* pn = (uint32_t*)s.t0 + s.cxt;
      const uint32_t v = *pn;
      const uint32_t count = v & 0x3ffu;
      const int32_t err = y * 32767 - (int32_t)(v >> 17);
      const uint32_t prod = (uint32_t)err * (uint32_t)L.dt[count];
      *pn = v + (prod & 0xFFFFFC00u) + (count < s.limit ? 1u : 0u);
      break;
    }
    case C_ICM: {
      s.t1[s.rc + (hmap4 & 15)] = L.ns[s.cxt * 4 + y];
      const uint32_t v = s.v;
      ((uint32_t*)s.t0)[s.cxt] = v + (uint32_t)((int32_t)((uint32_t)(y * 32767) - (v >> 8)) >> 2);
      break;
    }
    case C_ISSE: {
      const int err = y * 32767 - (int)L.squash[s.p + 2048];
      int2 w;
      w.x = d_clamp512k(s.w0 + ((err * pj + (1 << 12)) >> 13));
      w.y = d_clamp512k(s.w1 + ((err + 16) >> 5));
      ((int2*)s.t0)[s.cxt] = w;
      s.t1[s.rc + (hmap4 & 15)] = L.ns[s.cxt * 4 + y];
      break;
    }
    case C_MATCH: {
      uint8_t* buf = s.t1;
      const uint32_t mask = s.mask1;
      if ((int)s.rc != y) s.ra = 0;
      buf[s.rlimit & mask] = (uint8_t)(buf[s.rlimit & mask] * 2 + y);
      if (++s.cxt == 8) {
        s.cxt = 0;
        s.rlimit = (s.rlimit + 1) & mask;
        uint32_t* e = (uint32_t*)s.t0 + (s.h & s.mask0);
        if (s.ra == 0) {
          s.rb = s.rlimit - *e;
          if (s.rb & mask)
            while (s.ra < 255 && buf[(s.rlimit - s.ra - 1) & mask] == buf[(s.rlimit - s.ra - s.rb - 1) & mask]) ++s.ra;
        } else s.ra += s.ra < 255;
        *e = s.rlimit;
      }
      break;
    }
    case C_MIX2: {
      const int err = ((y * 32767 - (int)L.squash[s.p + 2048]) * (int)s.a4) >> 5;
      int w = s.w0;
      w += (err * (pj - pk) + (1 << 12)) >> 13;
      w = w < 0 ? 0 : (w > 65535 ? 65535 : w);
      ((uint32_t*)s.t0)[s.cxt] = (uint32_t)w;
      break;
    }
    default: break;
  }
  // MIX rows: lane-parallel read-modify-write, one mixer at a time
  uint64_t mixes = m.mix_mask;
  while (mixes) {
    const int i = unii(__builtin_ctzll(mixes));
    mixes &= mixes - 1;
    const int j = (int)rlu(s.a2, i), mm = (int)rlu(s.a3, i);
    const int err = ((y * 32767 - (int)L.squash[rl(s.p, i) + 2048]) * (int)rlu(s.a4, i)) >> 4;
    int32_t* row = (int32_t*)rl64((uint64_t)s.t0, i) + rlu(s.cxt, i);
    const int pin = __shfl(s.p, (j + m.lane) & 63);
    if (m.lane < mm) row[m.lane] = d_clamp512k(row[m.lane] + ((err * pin + (1 << 12)) >> 13));
  }
}

template <bool DEC>
__global__ __launch_bounds__(64 * kWavesPerGroup) void code_wave_kernel(const BlockJob* jobs, BlockResult* res,
                                                                         uint32_t nblocks, const DeviceTables* tb) {
  __shared__ LdsTables L;
  {
    // stage the constant tables: 78 KiB as 16-B vectors
    const uint4* src = (const uint4*)tb;      // stretch|squash|dt|dt2k|ns are contiguous in DeviceTables
    uint4* dst = (uint4*)&L;
    for (uint32_t i = threadIdx.x; i < sizeof(LdsTables) / 16; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t b = blockIdx.x * kWavesPerGroup + (threadIdx.x >> 6);
  if (b >= nblocks) return;
  BlockJob job = jobs[b];
  // everything in the job / plan header is wave-uniform: keep it scalar
  job.plan = (const uint8_t*)uni64((uint64_t)job.plan);
  job.arena = (uint8_t*)uni64((uint64_t)job.arena);
  job.in = (const uint8_t*)uni64((uint64_t)job.in);
  job.out = (uint8_t*)uni64((uint64_t)job.out);
  job.in_len = uni(job.in_len);
  job.out_cap = uni(job.out_cap);
  const uint32_t rslot = uni(job.res_slot);
  const PlanHeader* ph = (const PlanHeader*)job.plan;
  const CompDesc* comp = (const CompDesc*)(job.plan + uni(ph->off_comp));

  WaveModel m;
  m.L = &L;
  m.dep_mask = uni64(ph->dep_mask);
  m.mix_mask = uni64(ph->mix_mask);
  m.n = unii((int)ph->n);
  m.lane = lane;
  m.c8 = 1;
  m.hmap4 = 1;

  WaveVm vm;
  vm.prog = job.plan + uni(ph->off_prog);
  vm.prog_len = uni(ph->prog_len);
  vm.hmask = uni(ph->hmask);
  vm.mmask = uni(ph->mmask);
  vm.H = (uint32_t*)(job.arena + uni64(ph->off_H));
  vm.M = job.arena + uni64(ph->off_M);
  vm.R = (uint32_t*)(job.arena + uni64(ph->off_R));
  vm.A = vm.B = vm.C = vm.D = 0;
  vm.F = 0;

  Lane s;
  {
    CompDesc c;
    if (lane < m.n) c = comp[lane];
    else { c.type = C_NONE; c.a1 = c.a2 = c.a3 = c.a4 = c.a5 = 0; c.limit = c.mask0 = c.mask1 = 0; c.stride = 0; c.t0 = c.t1 = 0; }
    s.type = c.type; s.a1 = c.a1; s.a2 = c.a2; s.a3 = c.a3; s.a4 = c.a4; s.a5 = c.a5; s.stride = c.stride;
    s.limit = c.limit; s.mask0 = c.mask0; s.mask1 = c.mask1;
    s.t0 = job.arena + c.t0;
    s.t1 = job.arena + c.t1;
    s.cxt = s.ra = s.rb = s.rc = s.rlimit = 0;
    s.h = 0;
    s.p = (c.type == C_CONS) ? ((int)c.a1 - 128) * 4 : 0;
    s.w0 = s.w1 = 0;
    s.v = 0;
  }

  uint32_t low = 1, high = 0xFFFFFFFFu;
  uint32_t steps = 0;
  int status = 0;

#ifdef ZPQ_PROF
  unsigned long long prof[5] = {0, 0, 0, 0, 0};   // phase1, phase2, update, vm, total
  const unsigned long long prof_t0 = __builtin_readcyclecounter();
#define PROF_BEGIN unsigned long long pt_ = __builtin_readcyclecounter();
#define PROF_END(k) prof[k] += __builtin_readcyclecounter() - pt_;
#else
#define PROF_BEGIN
#define PROF_END(k)
#endif
  auto bit_step_post = [&](int y) -> int {   // update + c8/hmap4 bookkeeping (libzpaq.cpp:2055-2065)
    { PROF_BEGIN wave_update(m, s, y); PROF_END(2) }
    m.c8 += m.c8 + y;
    if (m.c8 >= 256) {
      PROF_BEGIN
      const int e = wvm_run(vm, (uint32_t)(m.c8 - 256));
      PROF_END(3)
      if (e) return e;
      m.hmap4 = 1;
      m.c8 = 1;
      s.h = vm.H[(uint32_t)lane & vm.hmask];
    } else if (m.c8 >= 16 && m.c8 < 32) {
      m.hmap4 = (m.hmap4 & 0xf) << 5 | y << 4 | 1;
    } else {
      m.hmap4 = (m.hmap4 & 0x1f0) | (((m.hmap4 & 0xf) * 2 + y) & 0xf);
    }
    return 0;
  };
  auto predict = [&]() -> uint32_t {
    { PROF_BEGIN wave_predict_phase1(m, s); PROF_END(0) }
    { PROF_BEGIN wave_predict_phase2(m, s); PROF_END(1) }
    return uni((uint32_t)L.squash[rl(s.p, m.n - 1) + 2048]);
  };

  if (!DEC) {
    uint32_t n = 0;
    auto encode = [&](int y, uint32_t p) {
      const uint32_t mid = low + (uint32_t)(((uint64_t)(high - low) * p) >> 16);
      if (y) high = mid; else low = mid + 1;
      while ((high ^ low) < 0x1000000u) {
        if (n < job.out_cap && lane == 0) job.out[n] = (uint8_t)(high >> 24);
        ++n;
        high = high << 8 | 255u;
        low = low << 8;
        low += (low == 0);
      }
    };
    for (uint32_t k = 0; k < job.in_len && !status; ++k) {
      const int c = (int)uni(job.in[k]);
      encode(0, 0);
      for (int i = 7; i >= 0; --i) {
        const uint32_t pr = predict();
        const int y = (c >> i) & 1;
        encode(y, pr * 2 + 1);
        status = bit_step_post(y);
        ++steps;
        if (status) break;
      }
    }
    if (!status) encode(1, 0);
    if (!status && n > job.out_cap) status = 3;
    if (lane == 0) { res[rslot].out_len = n; res[rslot].consumed = job.in_len; }
  } else {
    uint32_t rp = 0, n = 0, curr = 0;
    bool eos = false;
    for (int i = 0; i < 4; ++i) {
      if (rp >= job.in_len) { status = 6; break; }
      curr = curr << 8 | uni(job.in[rp++]);
    }
    while (!status && !eos && n < job.out_cap) {
      int c = 1;
      for (int bit = -1; bit < 8; ++bit) {
        uint32_t p = 0;
        if (bit >= 0) p = predict() * 2 + 1;
        if (curr < low || curr > high) { status = 2; break; }
        const uint32_t mid = low + (uint32_t)(((uint64_t)(high - low) * p) >> 16);
        int y;
        if (curr <= mid) { y = 1; high = mid; } else { y = 0; low = mid + 1; }
        while ((high ^ low) < 0x1000000u) {
          high = high << 8 | 255u;
          low = low << 8;
          low += (low == 0);
          if (rp >= job.in_len) { status = 6; break; }
          curr = curr << 8 | uni(job.in[rp++]);
        }
        if (status) break;
        if (bit < 0) {
          if (y) { eos = true; if (curr != 0) status = 2; break; }
        } else {
          c += c + y;
          status = bit_step_post(y);
          ++steps;
          if (status) break;
        }
      }
      if (status || eos) break;
      if (lane == 0) job.out[n] = (uint8_t)(c - 256);
      ++n;
    }
    if (lane == 0) { res[rslot].out_len = n; res[rslot].consumed = eos ? rp : 0; }
  }
  if (lane == 0) { res[rslot].status = status; res[rslot].steps = steps; }
#ifdef ZPQ_PROF
  if (lane == 0 && b == 0) {
    prof[4] = __builtin_readcyclecounter() - prof_t0;
    printf("[zpq prof] block 0: steps=%u cycles/bit: phase1=%.0f phase2=%.0f update=%.0f vm=%.0f total=%.0f\n", steps,
           (double)prof[0] / steps, (double)prof[1] / steps, (double)prof[2] / steps, (double)prof[3] / steps,
           (double)prof[4] / steps);
  }
#endif
}

}  // namespace zpq
