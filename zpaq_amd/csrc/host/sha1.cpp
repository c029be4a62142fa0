// SHA-1 (FIPS 180-1), written from the standard.  Replaces libzpaq::SHA1
// (libzpaq.h:934-954, libzpaq.cpp:106-177): the digest of every segment's
// uncompressed data goes into the archive trailer (253 + 20 bytes).
#include <cstring>

#include "common.hpp"

namespace zpq {

static inline U32 rol(U32 x, int n) { return (x << n) | (x >> (32 - n)); }

void Sha1::reset() {
  h_[0] = 0x67452301u; h_[1] = 0xEFCDAB89u; h_[2] = 0x98BADCFEu; h_[3] = 0x10325476u; h_[4] = 0xC3D2E1F0u;
  len_ = 0;
}

void sha1_compress(U32 h_[5], const U8* p) {
  U32 w[80];
  for (int i = 0; i < 16; ++i)
    w[i] = (U32)p[4 * i] << 24 | (U32)p[4 * i + 1] << 16 | (U32)p[4 * i + 2] << 8 | p[4 * i + 3];
  for (int i = 16; i < 80; ++i) w[i] = rol(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
  U32 a = h_[0], b = h_[1], c = h_[2], d = h_[3], e = h_[4];
  for (int i = 0; i < 80; ++i) {
    U32 f, k;
    if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
    else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
    else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
    else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
    U32 t = rol(a, 5) + f + e + k + w[i];
    e = d; d = c; c = rol(b, 30); b = a; a = t;
  }
  h_[0] += a; h_[1] += b; h_[2] += c; h_[3] += d; h_[4] += e;
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
