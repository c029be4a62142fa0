// Block-level operations shared by the C ABI (capi.cpp) and the
// libzpaq-compatible C++ layer (libzpaq_compat.cpp).
#pragma once
#include <functional>
#include <vector>

#include "common.hpp"

namespace zpq {

struct BlockInput {
  U8* data;               // may be modified in place where the reference would (E8E9)
  U32 n;
  const char* filename;   // may be null
  const char* comment;    // may be null; appended to the decimal size
};

// Batched libzpaq::compressBlock (libzpaq.cpp:7543-7731): one archive (tag ..
// 255) per input, all modelled payloads coded on the device in one batch.
void compress_blocks(const char* method, const std::vector<BlockInput>& in, bool dosha1,
                     std::vector<std::vector<U8>>& archives);

// Codes pp|data then EOS with the model of `header` (n>0) on the device and
// returns the coded bytes (Encoder::compress loop, 2419-2447).
std::vector<U8> encode_payload(const std::vector<U8>& header, const U8* pp, size_t npp, const U8* data, size_t n);

// Decodes one modelled payload (coded bytes + zero terminator) to EOS on the
// device; returns the decoded bytes, PP header included.  `hint` = expected
// decoded size or 0.
std::vector<U8> decode_payload(const std::vector<U8>& header, const U8* payload, size_t len, U64 hint);

// libzpaq::decompress (2378-2389) over a whole in-memory archive; `sink`
// receives each segment's post-processed data in order.
void decode_archive(const U8* a, size_t n, const std::function<void(const U8*, size_t)>& sink);

// PostProcessor::write (2195-2241): turns a decoded segment (PP header + payload) into the
// segment's data -- either passing it through or running the PCOMP program it carries.
void post_process(const std::vector<U8>& header, const std::vector<U8>& decoded, std::vector<U8>& data);

}  // namespace zpq
