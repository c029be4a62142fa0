// TEST INFRASTRUCTURE.  The arithmetic coder's byte-shifting loop (Encoder::encode, libzpaq.cpp:2411-2415) against the closed
// form the pipelined encoder's CODER unit uses (zpaq_amd/csrc/device/pipe_kernel.h pipe_coder::encode): same number of bytes,
// same bytes, same high and low, over random and edge-case states (equal bounds, low = 0, bounds that share 0 .. 4 leading
// bytes, low with only high-order bits), and the fast form of the latency shape's coder (pipe_coder_fast) against the whole of
// Encoder::encode.  tests/test_emu.py runs it; exit code 0 = no mismatch.
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
static uint64_t rng = 88172645463325252ull;
static uint32_t r32(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 16); }
int main(void) {
  long bad = 0, tot = 0;
  for (long it = 0; it < 20000000L; ++it) {
    uint32_t low, high;
    uint32_t a = r32(), b = r32();
    int mode = it % 5;
    if (mode == 0) { low = a; high = b; }
    else if (mode == 1) { low = a; high = a + (b & 0xFFFF); }
    else if (mode == 2) { low = a & 0xFF0000FF; high = low | (b & 0xFF); }
    else if (mode == 3) { low = a << (8 * (b & 3)); high = low + (r32() & 0xFFFFFF); }
    else { low = a; high = a; }
    if (low == 0 && (it & 64)) low = 1;
    if (high < low) { uint32_t t = high; high = low; low = t; }
    // reference loop (libzpaq.cpp:2411-2415)
    uint32_t l = low, h = high; uint64_t ob = 0; int n = 0;
    while ((h ^ l) < 0x1000000u) { ob = ob << 8 | (h >> 24); ++n; h = h << 8 | 255u; l = l << 8; l += (l == 0); }
    // closed form
    uint32_t x = high ^ low;
    unsigned k = x == 0 ? 4 : (unsigned)__builtin_clz(x) >> 3;
    unsigned sh = (8 * k) & 31;
    uint32_t h2 = k == 4 ? 0xFFFFFFFFu : ((high << sh) | ((1u << sh) - 1u));
    unsigned j = (39 - (unsigned)__builtin_ctz(low | 0x80000000u)) >> 3;
    uint32_t l2 = j <= k ? 1u << ((8 * (k - j)) & 31) : low << sh;
    uint64_t by = ((uint64_t)high << (8 * k)) >> 32;
    ++tot;
    if (n != (int)k || h != h2 || l != l2 || ob != by) { if (bad < 10) printf("low %08x high %08x: loop n=%d h=%08x l=%08x ob=%llx | k=%u h=%08x l=%08x by=%llx j=%u\n", low, high, n, h, l, (unsigned long long)ob, k, h2, l2, (unsigned long long)by, j); ++bad; }
  }
  // The latency shape's coder (pipe_coder_fast): the multiply as one mul_hi of a pre-shifted probability, y by mask, and the
  // shift-out for the states that pass its test -- high ^ low != 0 and the low 16 bits of low not all zero: k <= 3, high <<
  // 8k | ones, max(low << 8k, 1), the 4 top bytes of high stored blind with the first k of them final -- against the whole of
  // Encoder::encode (libzpaq.cpp:2402-2416) on normalised states (what encode leaves behind: top bytes differ, 1 <= low < high).
  long fast = 0, careful = 0;
  for (long it = 0; it < 20000000L; ++it) {
    uint32_t low, high;
    do {
      low = r32(); high = r32();
      if (it & 1) low = high - (r32() & 0xFFFFFFu);
      if (it % 7 == 0) low &= 0xFFFF0000u;
      if (it % 11 == 0) low &= 0xFFFFFF00u;
      if (it % 13 == 0) { const uint32_t edge = r32() << 24; low = edge - 1u - (r32() & 0xFFu); high = edge + (r32() & 0xFFu); }      // a tiny range across a top-byte boundary
    } while (!(low >= 1 && low < high && ((low ^ high) >> 24)));
    const uint32_t p16 = it % 5 == 0 ? 1u : (it % 17 == 0 ? 65535u : ((r32() & 0x7FFFu) * 2u + 1u));
    const int y = (int)(r32() >> 31), marker = it % 9 == 0;           // marker: encode(0, 0) in front of every byte
    uint32_t l = low, h = high; uint8_t ob[8]; int n = 0;
    const uint32_t mid = marker ? l : l + (uint32_t)(((uint64_t)(h - l) * p16) >> 16);
    if (!marker && y) h = mid; else l = mid + 1u;
    while (((h ^ l) & 0xFF000000u) == 0) { ob[n++] = (uint8_t)(h >> 24); h = h << 8 | 255u; l <<= 8; l += (l == 0); }
    const uint32_t P = p16 << 16, ym = 0u - (uint32_t)y;
    uint32_t high1, low1;
    if (marker) { high1 = high; low1 = low + 1u; }
    else { const uint32_t m2 = low + (uint32_t)(((uint64_t)(high - low) * P) >> 32); high1 = (m2 & ym) | (high & ~ym); low1 = (low & ym) | ((m2 + 1u) & ~ym); }
    const uint32_t x = high1 ^ low1;
    if (x == 0 || (low1 & 0xFFFFu) == 0) { ++careful; continue; }       // (the lane takes the byte again with the loop)
    const uint32_t sh = (uint32_t)__builtin_clz(x | 1u) & 24u;
    const uint32_t nh = (high1 << sh) | ((1u << sh) - 1u);
    uint32_t nl = low1 << sh; if (nl < 1u) nl = 1u;
    const int k = (int)(sh >> 3);
    ++fast; ++tot;
    int wrong = k != n || nh != h || nl != l;
    for (int i = 0; i < n && i < 4; ++i) wrong |= (uint8_t)(high1 >> (24 - 8 * i)) != ob[i];
    if (wrong) { if (bad < 10) printf("fast form: low %08x high %08x p %u y %d marker %d: loop n=%d h=%08x l=%08x | k=%d h=%08x l=%08x\n", low, high, p16, y, marker, n, h, l, k, nh, nl); ++bad; }
  }
  printf("fast form: %ld states, %ld sent to the loop\n", fast, careful);
  printf("%ld cases, %ld mismatches\n", tot, bad);
  return bad != 0 || fast < 1000000 || careful < 1000;
}
