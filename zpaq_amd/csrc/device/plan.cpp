// Header -> plan: replaces ZPAQL::read (libzpaq.cpp:887-931) and the sizing /
// validation half of Predictor::init (1776-1846), and lays the per-block arena
// out for HBM (device/layout.h).
#include "plan.hpp"

#include <cmath>
#include <cstring>

namespace zpq {

static thread_local int tl_plan_device = 0;
int plan_device_index() { return tl_plan_device; }
void set_plan_device_index(int dev) { tl_plan_device = (dev >= 0 && dev < zpq_plan::kMaxDevices) ? dev : 0; }

static const int kCompLen[10] = {0, 2, 3, 2, 3, 4, 6, 6, 3, 5};   // libzpaq.cpp:714

static uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// list_only: the caller only locates, lists or skips the block (Decompresser::findBlock): the header is checked with the
// REFERENCE's limits (sizes up to 32, ZPAQL::read, libzpaq.cpp:887-1006) and only `memory` of the result means anything;
// the limits of this build -- a block lane's tables live in HBM -- apply where a plan is needed to code or decode.
zpq_plan* plan_from_header(const U8* h, size_t hlen, bool list_only) {
  if (!h || hlen < 8) fail(ZPQ_E_HEADER, "header too short");
  size_t hsize = h[0] + 256u * h[1];
  if (hsize + 2 != hlen) fail(ZPQ_E_HEADER, "header size field does not match");
  const int hh = h[2], hm = h[3], n = h[6];
  if (hh > (list_only ? 32 : 24)) fail(ZPQ_E_HEADER, "H too big");     // reference: >32 (libzpaq.cpp:1018)
  if (hm > (list_only ? 32 : 28)) fail(ZPQ_E_HEADER, "M too big");
  auto too_big = [&](int v, int build_limit) { return v > (list_only ? 32 : build_limit); };

  std::vector<CompDesc> comps(n);
  std::vector<Segment> segs;
  uint64_t off = 0;
  double mem = std::ldexp(1.0, hh + 2) + std::ldexp(1.0, hm) + std::ldexp(1.0, h[4] + 2) +
               std::ldexp(1.0, h[5]) + (double)(hsize + 300);
  double algo = 0;
  bool wave_ok = n >= 1 && n <= 64;
  uint64_t dep_mask = 0, mix_mask = 0;

  auto seg = [&](uint64_t bytes, uint32_t kind, uint32_t value) {
    uint64_t o = off;
    uint64_t padded = align_up(bytes, 256);
    // init_arena_kernel fills in 16-byte stores: a table of 4 or 8 bytes (H with hh < 2, a CM / MATCH index of one or two
    // entries) was left as the previous batch had it and its padding was written through a misaligned pointer (round 6: found
    // by the GPU test of the small chains -- a one-ICM chain with hh = 0 behind another chain's batch).  The fill covers whole
    // 16-byte units; what it writes past the table lies in the table's own padding.
    const uint64_t fill = align_up(bytes, 16);
    segs.push_back(Segment{o, fill, kind, value});
    if (padded > fill) segs.push_back(Segment{o + fill, padded - fill, F_ZERO, 0});
    off += padded;
    return o;
  };

  size_t pos = 7;
  for (int i = 0; i < n; ++i) {
    if (pos >= hlen) fail(ZPQ_E_HEADER, "COMP overflows header");
    int type = h[pos];
    if (type < 1 || type > 9) fail(ZPQ_E_HEADER, "Invalid component type");
    if (pos + kCompLen[type] > hlen) fail(ZPQ_E_HEADER, "COMP overflows header");
    const U8* cp = h + pos;
    CompDesc& c = comps[i];
    memset(&c, 0, sizeof(c));
    c.type = type;
    c.a1 = cp[1];
    if (kCompLen[type] > 2) c.a2 = cp[2];
    if (kCompLen[type] > 3) c.a3 = cp[3];
    if (kCompLen[type] > 4) c.a4 = cp[4];
    if (kCompLen[type] > 5) c.a5 = cp[5];
    double size = std::ldexp(1.0, cp[1]);
    switch (type) {
      case C_CONS: break;
      case C_CM:
        if (too_big(cp[1], 28)) fail(ZPQ_E_HEADER, list_only ? "max size for CM is 32" : "max size for CM is 28 in this build");
        c.mask0 = (uint32_t)((1ull << cp[1]) - 1);
        c.limit = cp[2] * 4u;
        c.t0 = seg(4ull << cp[1], F_U32, 0x80000000u);
        mem += 4 * size; algo += 64;
        break;
      case C_ICM:
        if (cp[1] > (list_only ? 26 : 24)) fail(ZPQ_E_HEADER, list_only ? "max size for ICM is 26" : "max size for ICM is 24 in this build");
        c.limit = 1023;
        c.t0 = seg(1024, F_ICM, 0);
        c.mask1 = (uint32_t)((64ull << cp[1]) - 1);
        c.t1 = seg(64ull << cp[1], F_ZERO, 0);
        mem += 64 * size + 1024; algo += 64;
        break;
      case C_MATCH:
        if (too_big(cp[1], 28) || too_big(cp[2], 30)) fail(ZPQ_E_HEADER, list_only ? "max size for MATCH is 32 32" : "max size for MATCH is 28 30 in this build");
        c.mask0 = (uint32_t)((1ull << cp[1]) - 1);
        c.mask1 = (uint32_t)((1ull << cp[2]) - 1);
        c.t0 = seg(4ull << cp[1], F_ZERO, 0);
        c.t1 = seg(align_up(1ull << cp[2], 16), F_MATCHBUF, 0);
        mem += 4 * size + std::ldexp(1.0, cp[2]); algo += 10;
        break;
      case C_AVG:
        if (cp[1] >= i) fail(ZPQ_E_HEADER, "AVG j >= i");
        if (cp[2] >= i) fail(ZPQ_E_HEADER, "AVG k >= i");
        dep_mask |= 1ull << (i & 63);
        break;
      case C_MIX2:
        if (too_big(cp[1], 29)) fail(ZPQ_E_HEADER, list_only ? "max size for MIX2 is 32" : "max size for MIX2 is 29 in this build");
        if (cp[3] >= i) fail(ZPQ_E_HEADER, "MIX2 k >= i");
        if (cp[2] >= i) fail(ZPQ_E_HEADER, "MIX2 j >= i");
        c.mask0 = (uint32_t)((1ull << cp[1]) - 1);
        // device layout: one dword per weight (the reference packs U16) so that CM and MIX2
        // words are fetched, prefetched and written back by the same dword instructions
        c.t0 = seg(align_up(4ull << cp[1], 16), F_U32, 32768u);
        mem += 2 * size; algo += 32;
        dep_mask |= 1ull << (i & 63);
        break;
      case C_MIX:
        if (too_big(cp[1], 24)) fail(ZPQ_E_HEADER, list_only ? "max size for MIX is 32" : "max size for MIX is 24 in this build");
        if (cp[2] >= i) fail(ZPQ_E_HEADER, "MIX j >= i");
        if (cp[3] < 1 || cp[3] > i - cp[2]) fail(ZPQ_E_HEADER, "MIX m not in 1..i-j");
        c.mask0 = (uint32_t)((1ull << cp[1]) - 1);
        c.stride = mix_row_stride(cp[3]);              // padded rows: one 128-byte line per row (layout.h)
        c.t0 = seg(align_up((4ull * c.stride) << cp[1], 16), F_U32, (uint32_t)(65536 / cp[3]));
        mem += 4 * size * cp[3]; algo += 64.0 * cp[3];
        dep_mask |= 1ull << (i & 63);
        mix_mask |= 1ull << (i & 63);
        break;
      case C_ISSE:
        if (too_big(cp[1], 24)) fail(ZPQ_E_HEADER, list_only ? "max size for ISSE is 32" : "max size for ISSE is 24 in this build");
        if (cp[2] >= i) fail(ZPQ_E_HEADER, "ISSE j >= i");
        c.t0 = seg(2048, F_ISSE, 0);
        c.mask1 = (uint32_t)((64ull << cp[1]) - 1);
        c.t1 = seg(64ull << cp[1], F_ZERO, 0);
        mem += 64 * size + 2048; algo += 64;
        dep_mask |= 1ull << (i & 63);
        break;
      case C_SSE:
        if (too_big(cp[1], 24)) fail(ZPQ_E_HEADER, list_only ? "max size for SSE is 32" : "max size for SSE is 24 in this build");
        if (cp[2] >= i) fail(ZPQ_E_HEADER, "SSE j >= i");
        if (cp[3] > cp[4] * 4) fail(ZPQ_E_HEADER, "SSE start > limit*4");
        c.mask0 = (uint32_t)((32ull << cp[1]) - 1);
        c.limit = cp[4] * 4u;
        c.t0 = seg(128ull << cp[1], F_SSE, cp[3]);
        mem += 128 * size; algo += 96;
        dep_mask |= 1ull << (i & 63);
        break;
    }
    pos += kCompLen[type];
  }
  if (pos >= hlen || h[pos] != 0) fail(ZPQ_E_HEADER, "missing COMP END");
  ++pos;
  if (pos >= hlen) fail(ZPQ_E_HEADER, "missing HCOMP");
  if (h[hlen - 1] != 0) fail(ZPQ_E_HEADER, "missing HCOMP END");
  const U8* prog = h + pos;
  const uint32_t prog_len = (uint32_t)(hlen - pos);

  PlanHeader ph;
  memset(&ph, 0, sizeof(ph));
  ph.n = n;
  ph.hmask = (uint32_t)((1ull << hh) - 1);        // (hh / hm = 32 is legal when only memory() is asked for: no shift by the type's width)
  ph.mmask = (uint32_t)((1ull << hm) - 1);
  ph.prog_len = prog_len;
  ph.off_H = seg(4ull << hh, F_ZERO, 0);
  ph.off_M = seg(align_up(1ull << hm, 16), F_ZERO, 0);
  ph.off_R = seg(1024, F_ZERO, 0);
  ph.off_run = seg(16384, F_ZERO, 0);
  off = align_up(off, 4096);
  ph.arena_bytes = off;
  ph.nseg = (uint32_t)segs.size();
  ph.wave_ok = wave_ok;
  ph.dep_mask = dep_mask;
  ph.mix_mask = mix_mask;
  ph.off_comp = (uint32_t)align_up(sizeof(PlanHeader), 64);
  ph.off_seg = (uint32_t)align_up(ph.off_comp + sizeof(CompDesc) * (size_t)n, 64);
  ph.off_prog = (uint32_t)align_up(ph.off_seg + sizeof(Segment) * segs.size(), 64);
  ph.total_bytes = (uint32_t)align_up(ph.off_prog + prog_len + 8, 64);

  zpq_plan* p = new zpq_plan;
  p->header.assign(h, h + hlen);
  p->blob.assign(ph.total_bytes, 0);
  memcpy(p->blob.data(), &ph, sizeof(ph));
  if (n) memcpy(p->blob.data() + ph.off_comp, comps.data(), sizeof(CompDesc) * (size_t)n);
  memcpy(p->blob.data() + ph.off_seg, segs.data(), sizeof(Segment) * segs.size());
  memcpy(p->blob.data() + ph.off_prog, prog, prog_len);
  p->memory = mem;
  p->algo_bytes = algo;
  return p;
}

}  // namespace zpq
