// ZPAQ level-1/2 container: writing the bytes around the coded payload and
// locating blocks/segments when reading.  Replaces the format
// half of Compressor (libzpaq.cpp:2776-3004) and Decompresser (2247-2374).
#include <cstring>

#include "container.hpp"

namespace zpq {

const U8 kBlockTag[13] = {0x37, 0x6B, 0x53, 0x74, 0xA0, 0x31, 0x83, 0xD3, 0x8C, 0xB2, 0x28, 0xB0, 0xD3};

void write_block_prologue(std::vector<U8>& out, const std::vector<U8>& header, const char* filename,
                          const std::string& comment) {
  // Compressor::writeTag 2776, startBlock 2856-2868, startSegment 2871-2883
  out.insert(out.end(), kBlockTag, kBlockTag + 13);
  out.push_back('z'); out.push_back('P'); out.push_back('Q');
  out.push_back((U8)(1 + (header[6] == 0)));
  out.push_back(1);
  out.insert(out.end(), header.begin(), header.end());
  out.push_back(1);
  if (filename) out.insert(out.end(), (const U8*)filename, (const U8*)filename + strlen(filename));
  out.push_back(0);
  out.insert(out.end(), comment.begin(), comment.end());
  out.push_back(0);
  out.push_back(0);
}

void write_block_epilogue(std::vector<U8>& out, const U8* sha1) {
  // Compressor::endSegment 2947-2968, endBlock 3000-3004
  for (int i = 0; i < 4; ++i) out.push_back(0);
  if (sha1) { out.push_back(253); out.insert(out.end(), sha1, sha1 + 20); }
  else out.push_back(254);
  out.push_back(255);
}

void write_stored_payload(std::vector<U8>& out, const U8* pp, size_t npp, const U8* data, size_t n) {
  // Encoder::compress, n==0 branch (libzpaq.cpp:2436-2446): chunks of <= 64 KiB
  // over the concatenation pp|data, each prefixed by its 32-bit big-endian length.
  const size_t total = npp + n;
  size_t pos = 0;
  while (pos < total) {
    size_t k = total - pos;
    if (k > 65536) k = 65536;
    out.push_back((U8)(k >> 24)); out.push_back((U8)(k >> 16)); out.push_back((U8)(k >> 8)); out.push_back((U8)k);
    for (size_t i = 0; i < k; ++i) {
      const size_t q = pos + i;
      out.push_back(q < npp ? pp[q] : data[q - npp]);
    }
    pos += k;
  }
}

// ------------------------------------------------------------------ reading
namespace {
struct Cursor {
  const U8* a; size_t n, pos;
  int get() { return pos < n ? a[pos++] : -1; }
};
}  // namespace

bool find_block(const U8* a, size_t n, size_t& pos, FoundBlock& blk) {
  // Decompresser::findBlock (2247-2274): four rolling hashes pre-seeded with the
  // 13 tag bytes, so "zPQ" right at `pos` also matches.
  Cursor c{a, n, pos};
  U32 h1 = 0x3D49B113u, h2 = 0x29EB7F93u, h3 = 0x2614BE13u, h4 = 0x3828EB13u;
  int ch;
  while ((ch = c.get()) != -1) {
    h1 = h1 * 12 + (U32)ch;
    h2 = h2 * 20 + (U32)ch;
    h3 = h3 * 28 + (U32)ch;
    h4 = h4 * 44 + (U32)ch;
    if (h1 == 0xB16B88F1u && h2 == 0xFF5376F1u && h3 == 0x72AC5BF1u && h4 == 0x2F909AF1u) break;
  }
  if (ch == -1) { pos = c.pos; return false; }
  const int level = c.get();
  if (level != 1 && level != 2) fail(ZPQ_E_HEADER, "unsupported ZPAQ level");
  if (c.get() != 1) fail(ZPQ_E_HEADER, "unsupported ZPAQL type");
  if (c.pos + 2 > n) fail(ZPQ_E_EOF, "unexpected end of file");
  const size_t hsize = a[c.pos] + 256u * a[c.pos + 1];
  if (c.pos + 2 + hsize > n) fail(ZPQ_E_EOF, "unexpected end of file");
  blk.level = level;
  blk.header.assign(a + c.pos, a + c.pos + 2 + hsize);
  if (hsize < 6) fail(ZPQ_E_HEADER, "header too short");
  if (level == 1 && blk.header[6] == 0) fail(ZPQ_E_HEADER, "ZPAQ level 1 requires at least 1 component");
  pos = c.pos + 2 + hsize;
  return true;
}

bool find_segment(const U8* a, size_t n, size_t& pos, FoundSegment& seg) {
  // Decompresser::findFilename / readComment (2278-2311)
  Cursor c{a, n, pos};
  const int t = c.get();
  if (t == 255) { pos = c.pos; return false; }
  if (t != 1) fail(ZPQ_E_HEADER, "missing segment or end of block");
  seg.filename.clear(); seg.comment.clear();
  int ch;
  while ((ch = c.get()) > 0) seg.filename.push_back((char)ch);
  if (ch < 0) fail(ZPQ_E_EOF, "unexpected EOF");
  while ((ch = c.get()) > 0) seg.comment.push_back((char)ch);
  if (ch < 0) fail(ZPQ_E_EOF, "unexpected EOF");
  if (c.get() != 0) fail(ZPQ_E_HEADER, "missing reserved byte");
  seg.payload_begin = c.pos;
  pos = c.pos;
  return true;
}

size_t skip_payload(const U8* a, size_t n, size_t pos, bool modeled) {
  // Decoder::skip (2158-2181): returns the offset of the byte AFTER the
  // terminator (where 253/254 sits).
  Cursor c{a, n, pos};
  if (modeled) {
    U32 curr = 0;
    int ch = 0;
    while (curr == 0) { ch = c.get(); if (ch < 0) fail(ZPQ_E_EOF, "skipped to EOF"); curr = (U32)ch; }
    while (curr && (ch = c.get()) >= 0) curr = curr << 8 | (U32)ch;
    if (curr) fail(ZPQ_E_EOF, "skipped to EOF");
    while (c.pos < n && a[c.pos] == 0) ++c.pos;
    return c.pos;
  }
  for (;;) {
    if (c.pos + 4 > n) fail(ZPQ_E_EOF, "skipped to EOF");
    const U32 len = (U32)a[c.pos] << 24 | (U32)a[c.pos + 1] << 16 | (U32)a[c.pos + 2] << 8 | a[c.pos + 3];
    c.pos += 4;
    if (len == 0) break;
    if (c.pos + len > n) fail(ZPQ_E_EOF, "skipped to EOF");
    c.pos += len;
  }
  return c.pos;
}

void read_segment_end(const U8* a, size_t n, size_t& pos, FoundSegment& seg) {
  // Decompresser::readSegmentEnd (2348-2374)
  if (pos >= n) fail(ZPQ_E_EOF, "unexpected EOF");
  const int t = a[pos++];
  if (t == 254) seg.has_sha1 = false;
  else if (t == 253) {
    if (pos + 20 > n) fail(ZPQ_E_EOF, "unexpected EOF");
    seg.has_sha1 = true;
    memcpy(seg.sha1, a + pos, 20);
    pos += 20;
  } else fail(ZPQ_E_HEADER, "missing end of segment marker");
}

}  // namespace zpq
