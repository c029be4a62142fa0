// Constant tables of the predictor, regenerated from their closed forms
//  and gated by the reference's own known-answer checksums
// (libzpaq.cpp:1752-1761).  The device never computes exp/log: it receives
// these exact arrays.
#include <cmath>
#include <cstring>
#include <mutex>

#include "common.hpp"

namespace zpq {

namespace {

// Bit-history state machine of the public ZPAQ specification: a state is a
// (n0,n1[,last bit]) triple; states are numbered by increasing n0+n1 then n1.
// Reproduces sns[] (libzpaq.cpp:726-855) exactly (tests/test_host.py).
struct StateBuilder {
  static int count(int n0, int n1) {
    static const int bound[6] = {20, 48, 15, 8, 6, 5};
    if (n0 < n1) std::swap(n0, n1);
    if (n0 < 0 || n1 < 0 || n1 >= 6 || n0 > bound[n1]) return 0;
    return 1 + (n1 > 0 && n0 + n1 <= 17);
  }
  static int discount(int n) {
    return (n >= 1) + (n >= 2) + (n >= 3) + (n >= 4) + (n >= 5) + (n >= 7) + (n >= 8);
  }
  static void step(int& n0, int& n1, int y) {
    if (n0 < n1) { step(n1, n0, 1 - y); return; }
    if (y) { ++n1; n0 = discount(n0); } else { ++n0; n1 = discount(n1); }
    while (!count(n0, n1)) {
      if (n1 < 2) --n0;
      else { n0 = (n0 * (n1 - 1) + n1 / 2) / n1; --n1; }
    }
  }
  static void build(U8 ns[1024]) {
    const int N = 50;
    static U8 id[N][N][2];
    memset(id, 0, sizeof(id));
    int next = 0;
    for (int s = 0; s < N; ++s)
      for (int n1 = 0; n1 <= s; ++n1) {
        int n0 = s - n1, k = count(n0, n1);
        if (k) { id[n0][n1][0] = (U8)next; id[n0][n1][1] = (U8)(next + k - 1); next += k; }
      }
    memset(ns, 0, 1024);
    for (int n0 = 0; n0 < N; ++n0)
      for (int n1 = 0; n1 < N; ++n1)
        for (int y = 0; y < count(n0, n1); ++y) {
          int s = id[n0][n1][y];
          int a0 = n0, a1 = n1, b0 = n0, b1 = n1;
          step(a0, a1, 0);
          step(b0, b1, 1);
          ns[s * 4] = id[a0][a1][0];
          ns[s * 4 + 1] = id[b0][b1][1];
          ns[s * 4 + 2] = (U8)n0;
          ns[s * 4 + 3] = (U8)n1;
        }
  }
};

inline int clamp512k(int x) { return x < -(1 << 19) ? -(1 << 19) : x >= (1 << 19) ? (1 << 19) - 1 : x; }

Tables* build() {
  Tables* t = new Tables;
  t->dt2k[0] = 0;
  for (int i = 1; i < 256; ++i) t->dt2k[i] = 2048 / i;
  for (int i = 0; i < 1024; ++i) t->dt[i] = (1 << 17) / (i * 2 + 3) * 2;
  for (int i = 0; i < 4096; ++i)
    t->squash[i] = i < 1376 ? 0 : i >= 2720 ? 32767
                 : (U16)(int)(32768.0 / (1 + std::exp((i - 2048) * (-1.0 / 64))));
  for (int i = 16384; i < 32768; ++i)
    t->stretch[i] = (int16_t)((int)(std::log((i + 0.5) / (32767.5 - i)) * 64 + 0.5 + 100000) - 100000);
  for (int i = 0; i < 16384; ++i) t->stretch[i] = (int16_t)-t->stretch[32767 - i];
  StateBuilder::build(t->ns);
  U32 sq = 0, st = 0;
  for (int i = 32767; i >= 0; --i) st = st * 3 + (U32)(int32_t)t->stretch[i];
  for (int i = 4095; i >= 0; --i) sq = sq * 3 + t->squash[i];
  if (st != 3887533746u || sq != 2278286169u)
    fail(ZPQ_E_DEVICE, "squash/stretch tables fail the libzpaq checksums on this host");
  for (int s = 0; s < 256; ++s) {
    int cminit = ((t->ns[s * 4 + 3] * 2 + 1) << 22) / (t->ns[s * 4 + 2] + t->ns[s * 4 + 3] + 1);
    t->icm_init[s] = (U32)cminit;
    t->isse_init[s * 2] = 1 << 15;
    t->isse_init[s * 2 + 1] = (U32)clamp512k(t->stretch[cminit >> 8] * 1024);
  }
  for (int e = 0; e < 2016; ++e) {
    const int base = 16384 + 8 * e;
    U32 bm = 0;
    for (int i = 1; i < 8; ++i) {
      const int d = t->stretch[base + i] - t->stretch[base + i - 1];
      if (d < 0 || d > 1) fail(ZPQ_E_DEVICE, "stretch table is steeper than the compact form assumes");
      bm |= (U32)d << (i - 1);
    }
    t->stretch_cb[e] = ((U32)(uint16_t)t->stretch[base]) | bm << 16;
  }
  for (int i = 0; i < 256; ++i) t->stretch_top[i] = t->stretch[32512 + i];
  for (int j = 0; j < 32; ++j) t->sse_row[j] = (U32)t->squash[j * 64 - 992 + 2048] << 17;
  return t;
}

}  // namespace

const Tables& tables() {
  static std::once_flag once;
  static Tables* t = nullptr;
  static std::string err;
  std::call_once(once, [] {
    try { t = build(); } catch (Failure& f) { err = f.what(); }
  });
  if (!t) fail(ZPQ_E_DEVICE, err);
  return *t;
}

}  // namespace zpq
