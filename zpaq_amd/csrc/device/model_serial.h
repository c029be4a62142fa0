// Generic one-lane predictor: any valid header (n <= 255 components), all state
// in HBM.  This is the correctness baseline on the device and the fallback for
// chains the wave-parallel kernel does not cover (n > 64).  Same arithmetic as
// SURVEY App. A (Predictor::predict0/update0, libzpaq.cpp:1854-2066; find 2072;
// ZPAQL::execute 1041-1262), expressed over the arena layout of layout.h.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

namespace zpq {

struct RunVars {            // per-component mutable scalars (Component::cxt,a,b,c,limit libzpaq.h:1085)
  uint32_t cxt, a, b, c, limit;
};

struct SerialCtx {
  const PlanHeader* ph;
  const CompDesc* comp;
  const uint8_t* prog;
  const DeviceTables* tb;
  uint8_t* arena;
  uint32_t* H;
  uint8_t* M;
  uint32_t* R;
  uint32_t* h;      // [256] context hashes as seen by the predictor
  int32_t* p;       // [256] stretch-domain predictions
  RunVars* rv;      // [256]
  uint32_t A, B, C, D;
  int F;
  int c8, hmap4;
  int n;
};

__device__ inline int d_clamp2k(int x) { return x < -2048 ? -2048 : (x > 2047 ? 2047 : x); }
__device__ inline int d_clamp512k(int x) {
  return x < -(1 << 19) ? -(1 << 19) : (x >= (1 << 19) ? (1 << 19) - 1 : x);
}

// ---- HCOMP VM (one input byte) -------------------------------------------
__device__ inline uint32_t vm_src(SerialCtx& s, int k, uint32_t& pc) {
  switch (k) {
    case 0: return s.A;
    case 1: return s.B;
    case 2: return s.C;
    case 3: return s.D;
    case 4: return s.M[s.B & s.ph->mmask];
    case 5: return s.M[s.C & s.ph->mmask];
    case 6: return s.H[s.D & s.ph->hmask];
    default: return s.prog[pc++];
  }
}
__device__ inline void vm_dst(SerialCtx& s, int g, uint32_t v) {
  switch (g) {
    case 0: s.A = v; break;
    case 1: s.B = v; break;
    case 2: s.C = v; break;
    case 3: s.D = v; break;
    case 4: s.M[s.B & s.ph->mmask] = (uint8_t)v; break;
    case 5: s.M[s.C & s.ph->mmask] = (uint8_t)v; break;
    case 6: s.H[s.D & s.ph->hmask] = v; break;
  }
}

// Returns 0 or ZPQ_E_VM-style nonzero.
__device__ inline int vm_run(SerialCtx& s, uint32_t input) {
  const uint32_t len = s.ph->prog_len;
  uint32_t pc = 0;
  s.A = input;
  for (uint32_t steps = 0; steps < kMaxVmSteps; ++steps) {
    if (pc >= len) return 5;
    const int op = s.prog[pc++];
    const int g = op >> 3, k = op & 7;
    if (op < 64) {
      if (g == 7) {
        if (op == 56) return 0;
        else if (op == 57) { /* OUT: HCOMP has no output sink */ }
        else if (op == 59) s.A = (s.A + s.M[s.B & s.ph->mmask] + 512u) * 773u;
        else if (op == 60) { uint32_t* d = &s.H[s.D & s.ph->hmask]; *d = (*d + s.A + 512u) * 773u; }
        else if (op == 63) pc += 1 + (int)(int8_t)s.prog[pc];
        else return 5;
      } else if (k == 7) {
        if (pc >= len) return 5;
        if (g < 4) vm_dst(s, g, s.R[s.prog[pc++]]);
        else if (g == 4) { if (s.F) pc += 1 + (int)(int8_t)s.prog[pc]; else ++pc; }
        else if (g == 5) { if (!s.F) pc += 1 + (int)(int8_t)s.prog[pc]; else ++pc; }
        else s.R[s.prog[pc++]] = s.A;
      } else {
        if (op == 0 || k > 4) return 5;
        uint32_t x = vm_src(s, g, pc);
        if (k == 0) {
          uint32_t a = s.A;
          if (g == 4 || g == 5) { vm_dst(s, g, a & 255u); s.A = (a & 0xFFFFFF00u) | x; }
          else { vm_dst(s, g, a); s.A = x; }
        } else if (k == 1) vm_dst(s, g, x + 1);
        else if (k == 2) vm_dst(s, g, x - 1);
        else if (k == 3) vm_dst(s, g, ~x);
        else vm_dst(s, g, 0);
      }
    } else if (op < 120) {
      uint32_t v = vm_src(s, k, pc);
      vm_dst(s, g - 8, v);
    } else if (op < 128) {
      return 5;
    } else if (op < 240) {
      uint32_t v = vm_src(s, k, pc);
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
      uint32_t t = s.prog[pc] + 256u * s.prog[pc + 1];
      if (t >= len) return 5;
      pc = t;
    } else return 5;
  }
  return 5;
}

// ---- hashed bit-history row lookup (Predictor::find 2072-2088) ------------
__device__ inline uint32_t d_find(uint8_t* ht, uint32_t mask1, int sizebits, uint32_t cxt) {
  const uint32_t chk = (cxt >> sizebits) & 255u;
  const uint32_t h0 = (cxt * 16u) & (mask1 - 15u);   // & (ht_n - 16)
  if (ht[h0] == chk) return h0;
  const uint32_t h1 = h0 ^ 16u;
  if (ht[h1] == chk) return h1;
  const uint32_t h2 = h0 ^ 32u;
  if (ht[h2] == chk) return h2;
  const uint32_t p0 = ht[h0 + 1], p1 = ht[h1 + 1], p2 = ht[h2 + 1];
  uint32_t v;
  if (p0 <= p1 && p0 <= p2) v = h0;
  else if (p1 < p2) v = h1;
  else v = h2;
  uint4* row = (uint4*)(ht + v);
  *row = make_uint4(chk, 0, 0, 0);
  return v;
}

__device__ inline void d_train(uint32_t* pn, uint32_t limit, int y, const DeviceTables* tb) {
  uint32_t v = *pn;
  uint32_t count = v & 0x3ffu;
  int32_t err = y * 32767 - (int32_t)(v >> 17);
  uint32_t prod = (uint32_t)err * (uint32_t)tb->dt[count];
  *pn = v + (prod & 0xFFFFFC00u) + (count < limit ? 1u : 0u);
}

__device__ inline int serial_predict(SerialCtx& s) {
  const DeviceTables* tb = s.tb;
  const int c8 = s.c8, hmap4 = s.hmap4;
  int32_t* p = s.p;
  for (int i = 0; i < s.n; ++i) {
    const CompDesc& c = s.comp[i];
    RunVars& r = s.rv[i];
    switch (c.type) {
      case C_CM: {
        uint32_t* cm = (uint32_t*)(s.arena + c.t0);
        r.cxt = (s.h[i] ^ (uint32_t)hmap4) & c.mask0;
        p[i] = tb->stretch[cm[r.cxt] >> 17];
        break;
      }
      case C_ICM: {
        uint8_t* ht = s.arena + c.t1;
        if (c8 == 1 || (c8 & 0xf0) == 16) r.c = d_find(ht, c.mask1, c.a1 + 2, s.h[i] + 16u * (uint32_t)c8);
        r.cxt = ht[r.c + (hmap4 & 15)];
        const uint32_t* cm = (const uint32_t*)(s.arena + c.t0);
        p[i] = tb->stretch[cm[r.cxt] >> 8];
        break;
      }
      case C_MATCH: {
        if (r.a == 0) p[i] = 0;
        else {
          const uint8_t* buf = s.arena + c.t1;
          r.c = (buf[(r.limit - r.b) & c.mask1] >> (7 - r.cxt)) & 1u;
          int d = tb->dt2k[r.a];
          p[i] = tb->stretch[(r.c ? -d : d) & 32767];
        }
        break;
      }
      case C_AVG:
        p[i] = (p[c.a1] * (int)c.a3 + p[c.a2] * (256 - (int)c.a3)) >> 8;
        break;
      case C_MIX2: {
        const uint32_t* a16 = (const uint32_t*)(s.arena + c.t0);
        r.cxt = (s.h[i] + (uint32_t)(c8 & (int)c.a5)) & c.mask0;
        int w = a16[r.cxt];
        p[i] = (w * p[c.a2] + (65536 - w) * p[c.a3]) >> 16;
        break;
      }
      case C_MIX: {
        const int m = (int)c.a3;
        r.cxt = ((s.h[i] + (uint32_t)(c8 & (int)c.a5)) & c.mask0) * c.stride;
        const int32_t* wt = (const int32_t*)(s.arena + c.t0) + r.cxt;
        int sum = 0;
        for (int j = 0; j < m; ++j) sum += (wt[j] >> 8) * p[c.a2 + j];
        p[i] = d_clamp2k(sum >> 8);
        break;
      }
      case C_ISSE: {
        uint8_t* ht = s.arena + c.t1;
        if (c8 == 1 || (c8 & 0xf0) == 16) r.c = d_find(ht, c.mask1, c.a1 + 2, s.h[i] + 16u * (uint32_t)c8);
        r.cxt = ht[r.c + (hmap4 & 15)];
        const int32_t* wt = (const int32_t*)(s.arena + c.t0) + r.cxt * 2;
        p[i] = d_clamp2k((wt[0] * p[c.a2] + wt[1] * 64) >> 16);
        break;
      }
      case C_SSE: {
        const uint32_t* cm = (const uint32_t*)(s.arena + c.t0);
        uint32_t cx = (s.h[i] + (uint32_t)c8) * 32u;
        int pq = p[c.a2] + 992;
        pq = pq < 0 ? 0 : (pq > 1983 ? 1983 : pq);
        const int wt = pq & 63;
        pq >>= 6;
        cx += (uint32_t)pq;
        p[i] = tb->stretch[((cm[cx & c.mask0] >> 10) * (uint32_t)(64 - wt) +
                            (cm[(cx + 1) & c.mask0] >> 10) * (uint32_t)wt) >> 13];
        r.cxt = (cx + (uint32_t)(wt >> 5)) & c.mask0;
        break;
      }
      default: break;   // CONS: p[i] fixed at init
    }
  }
  return tb->squash[p[s.n - 1] + 2048];
}

// Returns 0 or a nonzero status (VM error).
__device__ inline int serial_update(SerialCtx& s, int y) {
  const DeviceTables* tb = s.tb;
  const int hmap4 = s.hmap4;
  int32_t* p = s.p;
  for (int i = 0; i < s.n; ++i) {
    const CompDesc& c = s.comp[i];
    RunVars& r = s.rv[i];
    switch (c.type) {
      case C_CM:
      case C_SSE:
        d_train((uint32_t*)(s.arena + c.t0) + r.cxt, c.limit, y, tb);
        break;
      case C_ICM: {
        uint8_t* slot = s.arena + c.t1 + r.c + (hmap4 & 15);
        *slot = tb->ns[*slot * 4 + y];
        uint32_t* pn = (uint32_t*)(s.arena + c.t0) + r.cxt;
        uint32_t v = *pn;
        *pn = v + (uint32_t)((int32_t)((uint32_t)(y * 32767) - (v >> 8)) >> 2);
        break;
      }
      case C_MATCH: {
        uint8_t* buf = s.arena + c.t1;
        uint32_t* idx = (uint32_t*)(s.arena + c.t0);
        const uint32_t mask = c.mask1;
        if ((int)r.c != y) r.a = 0;
        buf[r.limit & mask] = (uint8_t)(buf[r.limit & mask] * 2 + y);
        if (++r.cxt == 8) {
          r.cxt = 0;
          r.limit = (r.limit + 1) & mask;
          uint32_t* e = &idx[s.h[i] & c.mask0];
          if (r.a == 0) {
            r.b = r.limit - *e;
            if (r.b & mask)
              while (r.a < 255 && buf[(r.limit - r.a - 1) & mask] == buf[(r.limit - r.a - r.b - 1) & mask]) ++r.a;
          } else r.a += r.a < 255;
          *e = r.limit;
        }
        break;
      }
      case C_MIX2: {
        uint32_t* a16 = (uint32_t*)(s.arena + c.t0);
        int err = ((y * 32767 - (int)tb->squash[p[i] + 2048]) * (int)c.a4) >> 5;
        int w = a16[r.cxt];
        w += (err * (p[c.a2] - p[c.a3]) + (1 << 12)) >> 13;
        w = w < 0 ? 0 : (w > 65535 ? 65535 : w);
        a16[r.cxt] = (uint32_t)w;
        break;
      }
      case C_MIX: {
        const int m = (int)c.a3;
        int err = ((y * 32767 - (int)tb->squash[p[i] + 2048]) * (int)c.a4) >> 4;
        int32_t* wt = (int32_t*)(s.arena + c.t0) + r.cxt;
        for (int j = 0; j < m; ++j) wt[j] = d_clamp512k(wt[j] + ((err * p[c.a2 + j] + (1 << 12)) >> 13));
        break;
      }
      case C_ISSE: {
        int err = y * 32767 - (int)tb->squash[p[i] + 2048];
        int32_t* wt = (int32_t*)(s.arena + c.t0) + r.cxt * 2;
        wt[0] = d_clamp512k(wt[0] + ((err * p[c.a2] + (1 << 12)) >> 13));
        wt[1] = d_clamp512k(wt[1] + ((err + 16) >> 5));
        s.arena[c.t1 + r.c + (hmap4 & 15)] = tb->ns[r.cxt * 4 + y];
        break;
      }
      default: break;
    }
  }
  s.c8 += s.c8 + y;
  if (s.c8 >= 256) {
    int e = vm_run(s, (uint32_t)(s.c8 - 256));
    if (e) return e;
    s.hmap4 = 1;
    s.c8 = 1;
    for (int i = 0; i < s.n; ++i) s.h[i] = s.H[(uint32_t)i & s.ph->hmask];
  } else if (s.c8 >= 16 && s.c8 < 32) {
    s.hmap4 = (s.hmap4 & 0xf) << 5 | y << 4 | 1;
  } else {
    s.hmap4 = (s.hmap4 & 0x1f0) | (((s.hmap4 & 0xf) * 2 + y) & 0xf);
  }
  return 0;
}

__device__ inline void serial_open(SerialCtx& s, const BlockJob& job, const DeviceTables* tb) {
  s.ph = (const PlanHeader*)job.plan;
  s.comp = (const CompDesc*)(job.plan + s.ph->off_comp);
  s.prog = job.plan + s.ph->off_prog;
  s.tb = tb;
  s.arena = job.arena;
  s.H = (uint32_t*)(job.arena + s.ph->off_H);
  s.M = job.arena + s.ph->off_M;
  s.R = (uint32_t*)(job.arena + s.ph->off_R);
  uint8_t* run = job.arena + s.ph->off_run;
  s.h = (uint32_t*)run;
  s.p = (int32_t*)(run + 1024);
  s.rv = (RunVars*)(run + 2048);
  s.A = s.B = s.C = s.D = 0;
  s.F = 0;
  s.c8 = 1;
  s.hmap4 = 1;
  s.n = (int)s.ph->n;
  for (int i = 0; i < s.n; ++i)
    if (s.comp[i].type == C_CONS) s.p[i] = ((int)s.comp[i].a1 - 128) * 4;
}

}  // namespace zpq
