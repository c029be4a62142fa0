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
  const char* method = nullptr;   // this block's own method (null: the call's)
};

// Wall-clock phases of the last compress_blocks call of this process (the end-to-end figure SURVEY section 8(d)
// asks for: method expansion + SHA-1, H2D + kernels + D2H, framing), for bench.py.
struct ApiTiming {
  double front_ms = 0;        // SHA-1, method expansion, ZPAQL assembly, block prologue (host threads)
  double device_ms = 0;       // staging copy + H2D + Predictor init + coding kernels + D2H
  double stitch_ms = 0;       // coded bytes + segment trailer into the archives
  double total_ms = 0;
  double kernel_init_ms = 0, kernel_code_ms = 0;   // inside device_ms: the kernels alone (hipEvents)
  size_t blocks = 0;
  U32 sa_device_blocks = 0;   // blocks whose suffix array was built on the device (LZ77 / BWT pre-processors)
};
ApiTiming last_api_timing();

// Batched libzpaq::compressBlock (libzpaq.cpp:7543-7731): one archive (tag ..
// 255) per input, all modelled payloads coded on the device in one batch.
void compress_blocks(const char* method, const std::vector<BlockInput>& in, bool dosha1,
                     std::vector<std::vector<U8>>& archives);

// Codes pp|data then EOS with the model of `header` (n>0) on the device and
// returns the coded bytes (Encoder::compress loop, 2419-2447).
std::vector<U8> encode_payload(const std::vector<U8>& header, const U8* pp, size_t npp, const U8* data, size_t n);

// The same for a block of several segments (libzpaq.cpp:2889-2891: the encoder and the model are initialised once per
// block): segments[0] starts with the PP header; returns each segment's coded bytes (its end-of-segment code included).
std::vector<std::vector<U8>> encode_payload_segments(const std::vector<U8>& header, const std::vector<std::vector<U8>>& segments);
// ... and back: payloads[s] = coded bytes of segment s incl. terminator -> decoded bytes per segment (PP header in the first)
std::vector<U8> decode_payload_prefix(const std::vector<U8>& header, const std::vector<U8>& payload, U64 limit, bool* complete);
std::vector<std::vector<U8>> decode_payload_segments(const std::vector<U8>& header, const std::vector<std::vector<U8>>& payloads, U64 hint);

// Decodes one modelled payload (coded bytes + zero terminator) to EOS on the
// device; returns the decoded bytes, PP header included.  `hint` = expected
// decoded size or 0.
std::vector<U8> decode_payload(const std::vector<U8>& header, const U8* payload, size_t len, U64 hint);

// libzpaq::decompress (2378-2389) over a whole in-memory archive; `sink`
// receives each segment's post-processed data in order.
void decode_archive(const U8* a, size_t n, const std::function<void(const U8*, size_t)>& sink);

// PostProcessor (libzpaq.cpp:2183-2241) of one block: the first segment's decoded bytes start with the PP header
// (0 = PASS, or 1 len16 PCOMP program); every later segment of the block continues in the same mode -- PASS copies,
// PROG feeds the same ZPAQL machine -- and each segment ends with the machine's EOS call.
class PostProcessor {
 public:
  PostProcessor(int ph, int pm);
  ~PostProcessor();
  PostProcessor(const PostProcessor&) = delete;
  PostProcessor& operator=(const PostProcessor&) = delete;
  void segment(const U8* decoded, size_t n, std::vector<U8>& data);   // appends the segment's data to `data`
  bool loaded() const;                          // the PP header has been read
  const std::vector<U8>& program() const;       // PCOMP code (empty: PASS)
 private:
  struct Impl;
  Impl* impl_;
};

// whether the host runs this PCOMP program (code without its 2 length bytes) as C++ translated at build time
bool pcomp_is_translated(const U8* code, size_t len, int ph, int pm);

// The same for a block of ONE segment: turns a decoded segment (PP header + payload) into the
// segment's data -- either passing it through or running the PCOMP program it carries.
void post_process(const std::vector<U8>& header, const std::vector<U8>& decoded, std::vector<U8>& data);

}  // namespace zpq
