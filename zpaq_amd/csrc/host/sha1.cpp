// SHA-1 (FIPS 180-1), written from the standard.  Replaces libzpaq::SHA1
// (libzpaq.h:934-954, libzpaq.cpp:106-177): the digest of every segment's
// uncompressed data goes into the archive trailer (253 + 20 bytes).
#include <cstdlib>
#include <cstring>
#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
#endif

#include "common.hpp"

namespace zpq {

static inline U32 rol(U32 x, int n) { return (x << n) | (x >> (32 - n)); }

void Sha1::reset() {
  h_[0] = 0x67452301u; h_[1] = 0xEFCDAB89u; h_[2] = 0x98BADCFEu; h_[3] = 0x10325476u; h_[4] = 0xC3D2E1F0u;
  len_ = 0;
}

// Portable compression function: 16-word circular schedule, the four round types unrolled.
static void sha1_compress_scalar(U32 h_[5], const U8* p) {
  U32 w[16];
  for (int i = 0; i < 16; ++i)
    w[i] = (U32)p[4 * i] << 24 | (U32)p[4 * i + 1] << 16 | (U32)p[4 * i + 2] << 8 | p[4 * i + 3];
  U32 a = h_[0], b = h_[1], c = h_[2], d = h_[3], e = h_[4];
#define ZPQ_W(i) (w[(i) & 15] = rol(w[((i) + 13) & 15] ^ w[((i) + 8) & 15] ^ w[((i) + 2) & 15] ^ w[(i) & 15], 1))
#define ZPQ_R(f, k, x) { const U32 t = rol(a, 5) + (f) + e + (k) + (x); e = d; d = c; c = rol(b, 30); b = a; a = t; }
  for (int i = 0; i < 16; ++i) ZPQ_R((b & c) | (~b & d), 0x5A827999u, w[i])
  for (int i = 16; i < 20; ++i) ZPQ_R((b & c) | (~b & d), 0x5A827999u, ZPQ_W(i))
  for (int i = 20; i < 40; ++i) ZPQ_R(b ^ c ^ d, 0x6ED9EBA1u, ZPQ_W(i))
  for (int i = 40; i < 60; ++i) ZPQ_R((b & c) | (b & d) | (c & d), 0x8F1BBCDCu, ZPQ_W(i))
  for (int i = 60; i < 80; ++i) ZPQ_R(b ^ c ^ d, 0xCA62C1D6u, ZPQ_W(i))
#undef ZPQ_R
#undef ZPQ_W
  h_[0] += a; h_[1] += b; h_[2] += c; h_[3] += d; h_[4] += e;
}

#if defined(__x86_64__)
// x86 SHA extensions (sha1rnds4 does four rounds; sha1msg1/msg2 the message schedule), selected at run time.
// W[g] = the four schedule words of rounds 4g..4g+3; E of a group = rol30(A) of four rounds earlier (sha1nexte).
__attribute__((target("sha,sse4.1,ssse3"))) static void sha1_compress_shani(U32 h_[5], const U8* p) {
  const __m128i bswap = _mm_set_epi64x(0x0001020304050607ll, 0x08090a0b0c0d0e0fll);
  __m128i abcd = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i*)h_), 0x1B);
  __m128i e = _mm_set_epi32((int)h_[4], 0, 0, 0);
  const __m128i abcd_save = abcd, e_save = e;
  __m128i w[4], prev = abcd;   // prev = ABCD before the previous group
#define ZPQ_GROUP(g, imm)                                                                                   \
  {                                                                                                          \
    __m128i m;                                                                                               \
    if ((g) < 4) m = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * (g))), bswap);               \
    else m = _mm_sha1msg2_epu32(_mm_xor_si128(_mm_sha1msg1_epu32(w[(g) & 3], w[((g) + 1) & 3]), w[((g) + 2) & 3]), \
                                w[((g) + 3) & 3]);                                                           \
    w[(g) & 3] = m;                                                                                          \
    const __m128i x = (g) == 0 ? _mm_add_epi32(e, m) : _mm_sha1nexte_epu32(prev, m);                         \
    prev = abcd;                                                                                             \
    abcd = _mm_sha1rnds4_epu32(abcd, x, imm);                                                                \
  }
  ZPQ_GROUP(0, 0) ZPQ_GROUP(1, 0) ZPQ_GROUP(2, 0) ZPQ_GROUP(3, 0) ZPQ_GROUP(4, 0)
  ZPQ_GROUP(5, 1) ZPQ_GROUP(6, 1) ZPQ_GROUP(7, 1) ZPQ_GROUP(8, 1) ZPQ_GROUP(9, 1)
  ZPQ_GROUP(10, 2) ZPQ_GROUP(11, 2) ZPQ_GROUP(12, 2) ZPQ_GROUP(13, 2) ZPQ_GROUP(14, 2)
  ZPQ_GROUP(15, 3) ZPQ_GROUP(16, 3) ZPQ_GROUP(17, 3) ZPQ_GROUP(18, 3) ZPQ_GROUP(19, 3)
#undef ZPQ_GROUP
  e = _mm_sha1nexte_epu32(prev, e_save);
  abcd = _mm_add_epi32(abcd, abcd_save);
  _mm_storeu_si128((__m128i*)h_, _mm_shuffle_epi32(abcd, 0x1B));
  h_[4] = (U32)_mm_extract_epi32(e, 3);
}

static bool cpu_has_sha() {
  unsigned a, b, c, d;
  if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
  if (!((b >> 29) & 1)) return false;                       // CPUID.7.0:EBX.SHA
  if (!__get_cpuid(1, &a, &b, &c, &d)) return false;
  return ((c >> 19) & 1) && ((c >> 9) & 1);                  // SSE4.1, SSSE3
}
#endif

static bool g_sha1_portable = false;          // tests: zpq_sha1_force_portable()
void sha1_force_portable(bool yes) { g_sha1_portable = yes; }

void sha1_compress(U32 h_[5], const U8* p) {
#if defined(__x86_64__)
  static const bool shani = cpu_has_sha();
  if (shani && !g_sha1_portable) { sha1_compress_shani(h_, p); return; }
#endif
  sha1_compress_scalar(h_, p);
}

void Sha1::block(const U8* p) { sha1_compress(h_, p); }

void Sha1::update(const void* data, size_t n) {
  const U8* p = (const U8*)data;
  size_t fill = (size_t)(len_ & 63);
  len_ += n;
  if (fill) {
    size_t k = 64 - fill;
    if (k > n) k = n;
    memcpy(buf_ + fill, p, k);
    p += k; n -= k; fill += k;
    if (fill < 64) return;
    block(buf_);
  }
  while (n >= 64) { block(p); p += 64; n -= 64; }
  if (n) memcpy(buf_, p, n);
}

const U8* Sha1::result() {
  U64 bits = len_ * 8;
  U8 pad[72];
  size_t fill = (size_t)(len_ & 63);
  size_t padlen = (fill < 56 ? 56 : 120) - fill;
  memset(pad, 0, sizeof(pad));
  pad[0] = 0x80;
  for (int i = 0; i < 8; ++i) pad[padlen + i] = (U8)(bits >> (56 - 8 * i));
  update(pad, padlen + 8);
  for (int i = 0; i < 5; ++i) {
    out_[4 * i] = (U8)(h_[i] >> 24); out_[4 * i + 1] = (U8)(h_[i] >> 16);
    out_[4 * i + 2] = (U8)(h_[i] >> 8); out_[4 * i + 3] = (U8)h_[i];
  }
  reset();
  return out_;
}

}  // namespace zpq
