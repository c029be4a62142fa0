#pragma once
#include <string>
#include <vector>

#include "common.hpp"

namespace zpq {

extern const U8 kBlockTag[13];

// ---- writing (one block, one segment) ----
void write_block_prologue(std::vector<U8>& out, const std::vector<U8>& header, const char* filename,
                          const std::string& comment);
void write_stored_payload(std::vector<U8>& out, const U8* pp, size_t npp, const U8* data, size_t n);
// 00 00 00 00, then 253+sha1[20] or 254, then 255
void write_block_epilogue(std::vector<U8>& out, const U8* sha1_or_null);

// ---- reading ----
struct FoundBlock {
  int level = 0;
  std::vector<U8> header;
};
struct FoundSegment {
  std::string filename, comment;
  size_t payload_begin = 0;
  bool has_sha1 = false;
  U8 sha1[20];
};
// Scans from `pos` for the next block; on success leaves `pos` just after the header.
bool find_block(const U8* a, size_t n, size_t& pos, FoundBlock& blk);
// At `pos`: 1 = segment follows (fills seg, pos -> first payload byte), 255 = end of block (false).
bool find_segment(const U8* a, size_t n, size_t& pos, FoundSegment& seg);
// Offset just past the payload and its zero terminator (Decoder::skip).
size_t skip_payload(const U8* a, size_t n, size_t pos, bool modeled);
// Reads 253+sha1 / 254 at pos.
void read_segment_end(const U8* a, size_t n, size_t& pos, FoundSegment& seg);

}  // namespace zpq
