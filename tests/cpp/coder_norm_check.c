// TEST INFRASTRUCTURE.  The arithmetic coder's byte-shifting loop (Encoder::encode, libzpaq.cpp:2411-2415) against the closed
// form the pipelined encoder's CODER unit uses (zpaq_amd/csrc/device/pipe_kernel.h pipe_coder::encode): same number of bytes,
// same bytes, same high and low, over random and edge-case states (equal bounds, low = 0, bounds that share 0 .. 4 leading
// bytes, low with only high-order bits).  tests/test_emu.py runs it; exit code 0 = no mismatch.
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
  printf("%ld cases, %ld mismatches\n", tot, bad);
  return bad != 0;
}
