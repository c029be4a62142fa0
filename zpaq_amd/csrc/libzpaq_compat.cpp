// libzpaq-compatible C++ layer (include/libzpaq.h) over the device engine.
// Each function names the reference code it stands in for; the hot path
// (predict/update/encode/decode) is never executed on the host here.
#include "../../include/libzpaq.h"

#include <cstdio>
#include <stdexcept>

#include "device/plan.hpp"
#include "host/blocks.hpp"
#include "host/common.hpp"
#include "host/container.hpp"

namespace libzpaq {

// Weak default; an application definition of libzpaq::error overrides it
// (the reference leaves it undefined, libzpaq.h:858).
__attribute__((weak)) void error(const char* msg) { throw std::runtime_error(msg ? msg : "libzpaq error"); }

namespace {
// Internal failures surface through error() with the reference's wording where
// one exists ("Out of memory" matters: zpaq.cpp:123-126 maps it to bad_alloc).
[[noreturn]] void raise(const zpq::Failure& f) {
  if (f.code == ZPQ_E_NOMEM) error("Out of memory");
  error(f.what());
  throw std::logic_error("libzpaq::error returned");
}
template <typename F>
void guarded(F&& fn) {
  try { fn(); }
  catch (const zpq::Failure& f) { raise(f); }
  catch (const std::bad_alloc&) { error("Out of memory"); throw; }
}

struct VecWriter : public Writer {
  std::vector<U8>& v;
  explicit VecWriter(std::vector<U8>& v_) : v(v_) {}
  void put(int c) { v.push_back((U8)c); }
  void write(const char* b, int n) { if (n > 0) v.insert(v.end(), (const U8*)b, (const U8*)b + n); }
};
}  // namespace

int toU16(const char* p) { return (p[0] & 255) + 256 * (p[1] & 255); }

int Reader::read(char* buf, int n) {
  int i = 0, c;
  while (i < n && (c = get()) >= 0) buf[i++] = (char)c;
  return i;
}
void Writer::write(const char* buf, int n) {
  for (int i = 0; i < n; ++i) put((U8)buf[i]);
}

// ---------------------------------------------------------------- SHA1
// (reference libzpaq.cpp:106-177)
static void sha1_reset(U32 h[5]) {
  h[0] = 0x67452301u; h[1] = 0xEFCDAB89u; h[2] = 0x98BADCFEu; h[3] = 0x10325476u; h[4] = 0xC3D2E1F0u;
}
SHA1::SHA1() : len_(0) { sha1_reset(h_); }
void SHA1::put(int c) {
  buf_[len_ & 63] = (U8)c;
  if ((++len_ & 63) == 0) zpq::sha1_compress(h_, buf_);
}
void SHA1::write(const char* b, int64_t n) {
  int64_t i = 0;
  while (i < n && (len_ & 63)) put((U8)b[i++]);
  for (; i + 64 <= n; i += 64) { zpq::sha1_compress(h_, (const U8*)b + i); len_ += 64; }
  while (i < n) put((U8)b[i++]);
}
const char* SHA1::result() {
  const U64 bits = len_ * 8;
  put(0x80);
  while ((len_ & 63) != 56) put(0);
  for (int i = 7; i >= 0; --i) put((int)(bits >> (8 * i)) & 255);
  for (int i = 0; i < 5; ++i) {
    out_[4 * i] = (char)(h_[i] >> 24); out_[4 * i + 1] = (char)(h_[i] >> 16);
    out_[4 * i + 2] = (char)(h_[i] >> 8); out_[4 * i + 3] = (char)h_[i];
  }
  sha1_reset(h_);
  len_ = 0;
  return out_;
}

// ------------------------------------------------------ one-call functions
// compressBlock (reference libzpaq.cpp:7543-7731)
void compressBlock(StringBuffer* in, Writer* out, const char* method, const char* filename,
                   const char* comment, bool dosha1) {
  StringBuffer* ins[1] = {in};
  Writer* outs[1] = {out};
  const char* fn[1] = {filename};
  const char* cm[1] = {comment};
  compressBlocks(ins, outs, 1, method, fn, cm, dosha1);
}

void compressBlocks(StringBuffer* const* in, Writer* const* out, int n, const char* method,
                    const char* const* filename, const char* const* comment, bool dosha1) {
  if (n <= 0) return;
  std::vector<zpq::BlockInput> inputs((size_t)n);
  for (int i = 0; i < n; ++i)
    inputs[i] = zpq::BlockInput{in[i]->data(), (U32)in[i]->size(), filename ? filename[i] : 0, comment ? comment[i] : 0};
  std::vector<std::vector<U8>> archives;
  guarded([&] { zpq::compress_blocks(method, inputs, dosha1, archives); });
  for (int i = 0; i < n; ++i) {
    size_t pos = 0;
    while (pos < archives[i].size()) {   // Writer::write takes an int length
      const size_t k = std::min<size_t>(archives[i].size() - pos, 1u << 30);
      out[i]->write((const char*)archives[i].data() + pos, (int)k);
      pos += k;
    }
  }
}

// ... with a method per block (what an archiver's block queue holds: zpaq.cpp gives every block its own redundancy / type hints)
void compressBlocks(StringBuffer* const* in, Writer* const* out, int n, const char* const* methods,
                    const char* const* filename, const char* const* comment, bool dosha1) {
  if (n <= 0) return;
  if (!methods) error("compressBlocks: no methods");
  std::vector<zpq::BlockInput> inputs((size_t)n);
  for (int i = 0; i < n; ++i)
    inputs[i] = zpq::BlockInput{in[i]->data(), (U32)in[i]->size(), filename ? filename[i] : 0, comment ? comment[i] : 0, methods[i]};
  std::vector<std::vector<U8>> archives;
  guarded([&] { zpq::compress_blocks(nullptr, inputs, dosha1, archives); });
  for (int i = 0; i < n; ++i) {
    size_t pos = 0;
    while (pos < archives[i].size()) {
      const size_t k = std::min<size_t>(archives[i].size() - pos, 1u << 30);
      out[i]->write((const char*)archives[i].data() + pos, (int)k);
      pos += k;
    }
  }
}

// compress (reference libzpaq.cpp:3008-3031): the stream is cut into blocks of
// 2^(20+B)-4096 bytes; here up to kBatch blocks are gathered per device batch.
void compress(Reader* in, Writer* out, const char* method, const char* filename, const char* comment,
              bool dosha1) {
  int bs = 4;
  if (method && method[0] && method[1] >= '0' && method[1] <= '9') {
    bs = method[1] - '0';
    if (method[2] >= '0' && method[2] <= '9') bs = bs * 10 + method[2] - '0';
    if (bs > 11) bs = 11;
  }
  const size_t block = ((size_t)0x100000 << bs) - 4096;
  const size_t kBatchBytes = (size_t)1 << 31;   // host staging bound per batch
  const size_t kBatch = std::max<size_t>(1, std::min<size_t>(2048, kBatchBytes / block));   // 2048 = 8 blocks per CU: two wavefronts per SIMD
  bool first = true, eof = !in;
  while (!eof) {
    std::vector<StringBuffer*> bufs;
    while (bufs.size() < kBatch) {
      StringBuffer* sb = new StringBuffer(block);
      sb->write(0, (int)block);
      const int got = in->read((char*)sb->data(), (int)block);
      if (got <= 0) { delete sb; eof = true; break; }
      sb->resize((size_t)got);
      bufs.push_back(sb);
      if ((size_t)got < block) { /* short read: keep going, the next read decides EOF */ }
    }
    if (!bufs.empty()) {
      std::vector<Writer*> outs(bufs.size(), out);
      std::vector<const char*> fn(bufs.size(), (const char*)0), cm(bufs.size(), (const char*)0);
      if (first) { fn[0] = filename; cm[0] = comment; first = false; }
      try { compressBlocks(bufs.data(), outs.data(), (int)bufs.size(), method, fn.data(), cm.data(), dosha1); }
      catch (...) { for (StringBuffer* b : bufs) delete b; throw; }
    }
    for (StringBuffer* b : bufs) delete b;
  }
}

// decompress (reference libzpaq.cpp:2378-2389)
void decompress(Reader* in, Writer* out) {
  std::vector<U8> all;
  if (in) {
    char tmp[1 << 16];
    int got;
    while ((got = in->read(tmp, sizeof(tmp))) > 0) all.insert(all.end(), tmp, tmp + got);
  }
  guarded([&] {
    zpq::decode_archive(all.data(), all.size(), [&](const U8* p, size_t len) {
      size_t pos = 0;
      while (out && pos < len) {
        const size_t k = std::min<size_t>(len - pos, 1u << 30);
        out->write((const char*)p + pos, (int)k);
        pos += k;
      }
    });
  });
}

// ---------------------------------------------------- archiver-only services
// SHA-256 from FIPS 180-2.
static const U32 kSha256K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
void SHA256::init() {
  static const U32 h0[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(s_, h0, sizeof(s_));
  len_ = 0;
}
void SHA256::block() {
  U32 w[64];
  for (int i = 0; i < 16; ++i) w[i] = (U32)buf_[4 * i] << 24 | (U32)buf_[4 * i + 1] << 16 | (U32)buf_[4 * i + 2] << 8 | buf_[4 * i + 3];
  auto ror = [](U32 x, int n) { return x >> n | x << (32 - n); };
  for (int i = 16; i < 64; ++i)
    w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
           (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
  U32 a = s_[0], b = s_[1], c = s_[2], d = s_[3], e = s_[4], f = s_[5], g = s_[6], h = s_[7];
  for (int i = 0; i < 64; ++i) {
    const U32 t1 = h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + kSha256K[i] + w[i];
    const U32 t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s_[0] += a; s_[1] += b; s_[2] += c; s_[3] += d; s_[4] += e; s_[5] += f; s_[6] += g; s_[7] += h;
}
void SHA256::put(int c) {
  buf_[len_ & 63] = (U8)c;
  if ((++len_ & 63) == 0) block();
}
const char* SHA256::result() {
  const U64 bits = len_ * 8;
  put(0x80);
  while ((len_ & 63) != 56) put(0);
  for (int i = 7; i >= 0; --i) put((int)(bits >> (8 * i)) & 255);
  for (int i = 0; i < 8; ++i) {
    out_[4 * i] = (char)(s_[i] >> 24); out_[4 * i + 1] = (char)(s_[i] >> 16);
    out_[4 * i + 2] = (char)(s_[i] >> 8); out_[4 * i + 3] = (char)s_[i];
  }
  init();
  return out_;
}

AES_CTR::AES_CTR(const char*, int, const char*) { error("encrypted archives are outside this build's scope"); }
void AES_CTR::encrypt(U32, U32, U32, U32, unsigned char*) { error("encrypted archives are outside this build's scope"); }
void AES_CTR::encrypt(char*, int, U64) { error("encrypted archives are outside this build's scope"); }
void stretchKey(char*, const char*, const char*) { error("encrypted archives are outside this build's scope"); }

void random(char* buf, int n) {
  FILE* f = fopen("/dev/urandom", "rb");
  if (!f || (int)fread(buf, 1, (size_t)n, f) != n) { if (f) fclose(f); error("key generation failed"); }
  fclose(f);
  if (n >= 1 && (buf[0] == '7' || buf[0] == 'z')) buf[0] ^= 0x80;
}

// ------------------------------------------------------------ Decompresser
// (reference libzpaq.cpp:2247-2374).  A segment's payload is located with the
// same scan Decoder::skip uses, decoded completely on the device the first time
// decompress() is called for it, and then served in the requested pieces.
Decompresser::Decompresser()
    : in_(0), out_(0), sha1_(0), rpos_(0), plan_(0), dpos_(0), payload_end_(0), seg_decoded_(false),
      segs_in_block_(0), pp_(0), skipped_in_block_(false), prefix_tried_(false), prefix_active_(false), state_(BLOCK) {}
Decompresser::~Decompresser() { delete (zpq::PostProcessor*)pp_; }

int Decompresser::getc() {
  if (rpos_ == buf_.size()) {
    // drop consumed bytes, refill (the read-ahead the reference's Decoder::get does)
    buf_.clear();
    rpos_ = 0;
    if (!in_) return -1;
    buf_.resize(1 << 16);
    const int got = in_->read((char*)buf_.data(), (int)buf_.size());
    buf_.resize(got > 0 ? (size_t)got : 0);
    if (buf_.empty()) return -1;
  }
  return buf_[rpos_++];
}

// byte at rpos_ + off, reading ahead as far as needed without consuming anything (buffered() stays exact)
int Decompresser::peek(size_t off) {
  while (rpos_ + off >= buf_.size()) {
    if (!in_) return -1;
    const size_t old = buf_.size();
    buf_.resize(old + (1 << 16));
    const int got = in_->read((char*)buf_.data() + old, 1 << 16);
    buf_.resize(old + (got > 0 ? (size_t)got : 0));
    if (got <= 0) return -1;
  }
  return buf_[rpos_ + off];
}

bool Decompresser::findBlock(double* memptr) {
  U32 h1 = 0x3D49B113u, h2 = 0x29EB7F93u, h3 = 0x2614BE13u, h4 = 0x3828EB13u;
  int c;
  while ((c = getc()) != -1) {
    h1 = h1 * 12 + (U32)c; h2 = h2 * 20 + (U32)c; h3 = h3 * 28 + (U32)c; h4 = h4 * 44 + (U32)c;
    if (h1 == 0xB16B88F1u && h2 == 0xFF5376F1u && h3 == 0x72AC5BF1u && h4 == 0x2F909AF1u) break;
  }
  if (c == -1) return false;
  const int level = getc();
  if (level != 1 && level != 2) error("unsupported ZPAQ level");
  if (getc() != 1) error("unsupported ZPAQL type");
  header_.clear();
  const int lo = getc(), hi = getc();
  if (lo < 0 || hi < 0) error("unexpected end of file");
  header_.push_back((U8)lo); header_.push_back((U8)hi);
  const int hsize = lo + 256 * hi;
  for (int i = 0; i < hsize; ++i) {
    const int b = getc();
    if (b < 0) error("unexpected end of file");
    header_.push_back((U8)b);
  }
  if (hsize < 6) error("header too short");
  if (level == 1 && header_[6] == 0) error("ZPAQ level 1 requires at least 1 component");
  // ZPAQL::read's checks and ZPAQL::memory() for every block, modelled or not (libzpaq.cpp:1145-1216, 1228-1270) -- with the
  // reference's limits: locating, listing or skipping a block needs no plan; this build's size limits apply when it is decoded
  guarded([&] {
    zpq_plan* p = zpq::plan_from_header(header_.data(), header_.size(), /*list_only=*/true);
    if (memptr) *memptr = p->memory;
    delete p;
  });
  segs_in_block_ = 0;
  block_cache_.clear();
  skipped_in_block_ = false;
  delete (zpq::PostProcessor*)pp_;
  pp_ = new zpq::PostProcessor(header_[4], header_[5]);
  state_ = FILENAME;
  return true;
}

void Decompresser::hcomp(Writer* out2) {
  for (size_t i = 0; i < header_.size(); ++i) out2->put(header_[i]);
}
// The PCOMP program of the block as the reference returns it (ZPAQL::write(out2, true), libzpaq.cpp:866-884):
// len16 + code, false when the block is not post-processed.  Known once the first segment has been decoded.
bool Decompresser::pcomp(Writer* out2) {
  zpq::PostProcessor* pp = (zpq::PostProcessor*)pp_;
  if (!pp || !pp->loaded() || pp->program().empty()) return false;
  const std::vector<U8>& code = pp->program();
  if (out2) {
    out2->put((int)(code.size() & 255));
    out2->put((int)(code.size() >> 8));
    for (size_t i = 0; i < code.size(); ++i) out2->put(code[i]);
  }
  return true;
}

bool Decompresser::findFilename(Writer* filename) {
  const int c = getc();
  if (c == 1) {
    for (;;) {
      const int b = getc();
      if (b == -1) error("unexpected EOF");
      if (b == 0) { state_ = COMMENT; return true; }
      if (filename) filename->put(b);
    }
  } else if (c == 255) { state_ = BLOCK; return false; }
  error("missing segment or end of block");
  return false;
}

void Decompresser::readComment(Writer* comment) {
  state_ = DATA;
  for (;;) {
    const int b = getc();
    if (b == -1) error("unexpected EOF");
    if (b == 0) break;
    if (comment) comment->put(b);
  }
  if (getc() != 0) error("missing reserved byte");
  seg_decoded_ = false;
  decoded_.clear();
  dpos_ = 0;
  prefix_tried_ = prefix_active_ = false;
  prefix_.clear();
}

// The coded payloads of the block's remaining segments, read AHEAD of the input position without consuming anything: this
// payload, its trailer and any further segments up to the end of the block.
void Decompresser::peek_block_payloads(std::vector<std::vector<U8> >& payloads) {
  size_t off = 0;
  auto need = [&](size_t o) -> int { const int c = peek(o); if (c < 0) error("unexpected end of file"); return c; };
  for (;;) {
    std::vector<U8> pl;
    U32 curr = 0;
    int c = 0;
    while (curr == 0) { c = need(off++); pl.push_back((U8)c); curr = (U32)c; }
    while (curr) { c = need(off++); pl.push_back((U8)c); curr = curr << 8 | (U32)c; }
    while ((c = peek(off)) == 0) { pl.push_back(0); ++off; }     // a flush byte of 00 puts a fifth zero in front
    payloads.push_back(pl);
    c = need(off++);                                             // trailer
    if (c == 253) off += 20;
    else if (c != 254) error("missing end of segment marker");
    c = need(off++);
    if (c != 1) break;                                           // 255: end of block
    while (need(off++) != 0) {}                                  // filename
    while (need(off++) != 0) {}                                  // comment
    if (need(off++) != 0) error("missing reserved byte");
  }
}

// Decompresser::decompress(n) with a small n (libzpaq.cpp:2315-2343; zpaq.cpp:2859-2866 stops as soon as it has the
// fragments it wants): the device decodes a block from its first bit, and its time is the number of bits it decodes, so
// the first call decodes a PREFIX -- kPrefixBytes of output, ~6 % of a 1 MiB block's decode time -- and hands out of that;
// only a caller that asks for more pays for the whole block (decode_segment).  Taken for the first segment of a modelled
// block that has no other segment and no PCOMP program (the prefix of a post-processed stream is not the prefix of its
// output); nothing is consumed from the input and no state but prefix_ changes, so every other path goes on as if this
// had not happened.
static const size_t kPrefixBytes = 65536;
void Decompresser::try_prefix() {
  prefix_tried_ = true;
  if (header_[6] == 0 || skipped_in_block_ || segs_in_block_ != 0 || pp_ == 0) return;
  std::vector<std::vector<U8> > payloads;
  peek_block_payloads(payloads);
  if (payloads.size() != 1 || payloads[0].size() < 4096) return;          // (a short payload: the whole block costs little)
  bool complete = false;
  std::vector<U8> got;
  guarded([&] { got = zpq::decode_payload_prefix(header_, payloads[0], kPrefixBytes + 1, &complete); });
  if (complete || got.empty() || got[0] != 0) return;                     // ended inside the prefix / PCOMP follows: the ordinary way
  prefix_.assign(got.begin() + 1, got.end());
  prefix_active_ = true;
}

// Gathers this segment's payload from the input (up to and including the zero
// terminator) and decodes it on the device.
void Decompresser::decode_segment() {
  std::vector<U8> payload;
  const bool modeled = header_[6] != 0;
  if (modeled) {
    if (skipped_in_block_) error("decompression after skipped segment");
    const int my_index = segs_in_block_++;
    if (my_index == 0) {
      // The model and the coder run on from segment to segment, so the block's segments are decoded together
      std::vector<std::vector<U8> > payloads;
      peek_block_payloads(payloads);
      guarded([&] { block_cache_ = zpq::decode_payload_segments(header_, payloads, 0); });
    }
    if ((size_t)my_index >= block_cache_.size()) error("segment not found in its block");
    // consume this segment's payload
    {
      U32 curr = 0;
      int c = 0;
      while (curr == 0) { c = getc(); if (c < 0) error("unexpected end of file"); curr = (U32)c; }
      while (curr) { c = getc(); if (c < 0) error("unexpected end of file"); curr = curr << 8 | (U32)c; }
      while ((c = getc()) == 0) {}
      if (c >= 0) --rpos_;
    }
    decoded_.swap(block_cache_[(size_t)my_index]);
  } else {
    for (;;) {
      U32 len = 0;
      for (int i = 0; i < 4; ++i) { const int c = getc(); if (c < 0) error("unexpected end of file"); len = len << 8 | (U32)c; }
      if (!len) break;
      for (U32 i = 0; i < len; ++i) { const int c = getc(); if (c < 0) error("unexpected end of file"); decoded_.push_back((U8)c); }
    }
  }
  // PostProcessor of the block: pass through, or run the PCOMP program its first segment carried
  guarded([&] {
    std::vector<U8> data;
    ((zpq::PostProcessor*)pp_)->segment(decoded_.data(), decoded_.size(), data);
    decoded_.swap(data);
  });
  dpos_ = 0;
  seg_decoded_ = true;
}

bool Decompresser::decompress(int n) {
  if (state_ != DATA) error("decompression after skipped segment");
  // a segment of this block was left before its end (readSegmentEnd in the DATA state: Decoder::skip, decode_state = SKIP,
  // libzpaq.cpp:2346-2352): the reference refuses every later segment of the block, with or without a model (2300)
  if (skipped_in_block_) error("decompression after skipped segment");
  if (!seg_decoded_) {
    if (!prefix_tried_ && n >= 0 && (size_t)n < kPrefixBytes) try_prefix();
    if (prefix_active_) {
      if ((size_t)(n < 0 ? 0x7FFFFFFF : n) <= prefix_.size() - dpos_) {      // the prefix holds what is asked for
        if (n) {
          if (out_) out_->write((const char*)prefix_.data() + dpos_, n);
          if (sha1_) sha1_->write((const char*)prefix_.data() + dpos_, (int64_t)n);
          dpos_ += (size_t)n;
        }
        return true;
      }
      const size_t handed_out = dpos_;          // more than that: the whole segment, going on behind what was handed out
      prefix_active_ = false;
      prefix_.clear();
      decode_segment();
      dpos_ = handed_out;
    } else decode_segment();
  }
  size_t avail = decoded_.size() - dpos_;
  size_t take = (n < 0 || (size_t)n > avail) ? avail : (size_t)n;
  if (take) {
    if (out_) out_->write((const char*)decoded_.data() + dpos_, (int)take);
    if (sha1_) sha1_->write((const char*)decoded_.data() + dpos_, (int64_t)take);
    dpos_ += take;
  }
  // the reference returns false exactly when the EOS symbol is decoded (2333-2340):
  // i.e. when more bytes were asked for than the segment still held
  if (n < 0 || (size_t)n > take) { state_ = SEGEND; return false; }
  return true;
}

void Decompresser::readSegmentEnd(char* sha1string) {
  int c = 0;
  if (state_ == DATA) {
    if (!seg_decoded_) {
      // skip without decoding (Decoder::skip 2158-2181)
      if (header_[6]) {
        U32 curr = 0;
        while (curr == 0) { c = getc(); if (c < 0) error("unexpected end of file"); curr = (U32)c; }
        while (curr && (c = getc()) >= 0) curr = curr << 8 | (U32)c;
        ++segs_in_block_;
      } else {
        for (;;) {
          U32 len = 0;
          for (int i = 0; i < 4; ++i) { c = getc(); if (c < 0) error("skipped to EOF"); len = len << 8 | (U32)c; }
          if (!len) break;
          for (U32 i = 0; i < len; ++i) if (getc() < 0) error("skipped to EOF");
        }
      }
    }
    while ((c = getc()) == 0) {}
    skipped_in_block_ = true;      // left before its end, read or not: no later segment of this block decodes
  } else {
    while ((c = getc()) == 0) {}   // a flush byte of 00 leaves extra zeros before the marker
  }
  state_ = FILENAME;
  if (c == 254) { if (sha1string) sha1string[0] = 0; }
  else if (c == 253) {
    if (sha1string) sha1string[0] = 1;
    for (int i = 1; i <= 20; ++i) { const int b = getc(); if (sha1string) sha1string[i] = (char)b; }
  } else error("missing end of segment marker");
}

// -------------------------------------------------------------- Compressor
// (reference libzpaq.cpp:2776-3004).  Bytes are gathered per segment and coded
// on the device when the segment ends.
Compressor::Compressor() : out_(0), in_(0), verify_(false), pp_(0), segs_(0), state_(INIT) { memset(sha1result_, 0, 20); }
Compressor::~Compressor() { delete (zpq::PostProcessor*)pp_; }

// Header bytes of the three built-in models (Compressor::startBlock(int), libzpaq.cpp:2796-2822: min.cfg, mid.cfg,
// max.cfg).  They are format constants: an archive made with level 1..3 stores exactly these bytes.
static const char* const kBuiltinModels[3] = {
    "1a00010200000203100813000060041c3b0a3b70190a3b0a3b703800",
    "450003030000080305080d000811010812020812030813040416180710000718ff0011684a045f013b700a193b700a193b700a193b700a19"
    "3b700a193b0a3b701945cf08703800",
    "c400050900001601a00305080d010810020812030813040813050814060416180311081309030d030d030d030e0710000f18ff0708001"
    "00aff06000f10180009081120ff0608111210ff09101320ff0600131410000011684a045f023b700a193b700a193b700a193b700a193b700a"
    "193b0a3b700a193b700a1945b720ef402f0ee75b2f0a193c1a30869714703f0946df00270319701a3419194a0a043b70190a043b70190a04"
    "3b7019418fd448043b70088fd80844af3c3c1945cf09701919191919703800"};

void Compressor::writeTag() {
  for (int i = 0; i < 13; ++i) out_->put(zpq::kBlockTag[i]);
}

}  // namespace libzpaq
namespace zpq {
// the stored header of built-in model `level` (1 .. 3); empty for any other level  (C ABI: zpq_builtin_model_header)
std::vector<U8> builtin_model(int level) {
  std::vector<U8> bytes;
  if (level < 1 || level > 3) return bytes;
  const char* hex = libzpaq::kBuiltinModels[level - 1];
  for (size_t i = 0; hex[i] && hex[i + 1]; i += 2) {
    auto nib = [](char ch) { return ch <= '9' ? ch - '0' : ch - 'a' + 10; };
    bytes.push_back((U8)(nib(hex[i]) * 16 + nib(hex[i + 1])));
  }
  return bytes;
}
}  // namespace zpq
namespace libzpaq {

void Compressor::startBlock(int level) {
  if (level < 1) error("compression level must be at least 1");
  if (level > 3) error("compression level too high");
  const std::vector<zpq::U8> bytes = zpq::builtin_model(level);
  startBlock((const char*)bytes.data());
}

void Compressor::startBlock(const char* hcomp) {
  const size_t hsize = (size_t)toU16(hcomp);
  header_.assign((const U8*)hcomp, (const U8*)hcomp + hsize + 2);
  pcomp_.clear();
  guarded([&] { delete zpq::plan_from_header(header_.data(), header_.size()); });   // validate
  out_->put('z'); out_->put('P'); out_->put('Q');
  out_->put(1 + (header_[6] == 0));
  out_->put(1);
  for (size_t i = 0; i < header_.size(); ++i) out_->put(header_[i]);
  segs_ = 0;
  segq_.clear();
  delete (zpq::PostProcessor*)pp_;
  pp_ = 0;
  state_ = BLOCK1;
}

void Compressor::startBlock(const char* config, int* args, Writer* pcomp_cmd) {
  zpq::Assembled as;
  guarded([&] { as = zpq::assemble(config, args); });
  if (pcomp_cmd) for (char ch : as.pcomp_cmd) pcomp_cmd->put((U8)ch);
  header_ = as.hcomp;
  pcomp_ = as.pcomp;
  out_->put('z'); out_->put('P'); out_->put('Q');
  out_->put(1 + (header_[6] == 0));
  out_->put(1);
  for (size_t i = 0; i < header_.size(); ++i) out_->put(header_[i]);
  segs_ = 0;
  segq_.clear();
  delete (zpq::PostProcessor*)pp_;
  pp_ = 0;
  state_ = BLOCK1;
}

void Compressor::hcomp(Writer* out2) { for (size_t i = 0; i < header_.size(); ++i) out2->put(header_[i]); }
bool Compressor::pcomp(Writer* out2) {
  if (pcomp_.empty()) return false;
  for (size_t i = 0; i < pcomp_.size(); ++i) out2->put(pcomp_[i]);
  return true;
}

void Compressor::startSegment(const char* filename, const char* comment) {
  if (header_.size() < 7) error("no block started");     // the reference asserts (and reads a null header under NDEBUG)
  std::vector<U8> head;
  head.push_back(1);
  while (filename && *filename) head.push_back((U8)*filename++);
  head.push_back(0);
  while (comment && *comment) head.push_back((U8)*comment++);
  head.push_back(0);
  head.push_back(0);
  if (header_[6] == 0) out_->write((const char*)head.data(), (int)head.size());
  else { segq_.push_back(SegBuf()); segq_.back().head.swap(head); }
  if (state_ == BLOCK1) state_ = SEG1;
  if (state_ == BLOCK2) state_ = SEG2;
  pending_.clear();
}

void Compressor::postProcess(const char* pcomp, int len) {
  if (state_ == SEG2) return;
  std::vector<U8> code;
  if (!pcomp) { if (pcomp_.size() > 2) code.assign(pcomp_.begin() + 2, pcomp_.end()); }
  else {
    if (len == 0) { len = toU16(pcomp); pcomp += 2; }
    code.assign((const U8*)pcomp, (const U8*)pcomp + len);
  }
  if (!code.empty()) {
    pending_.push_back(1);
    pending_.push_back((U8)(code.size() & 255));
    pending_.push_back((U8)(code.size() >> 8));
    pending_.insert(pending_.end(), code.begin(), code.end());
  } else pending_.push_back(0);
  state_ = SEG2;
}

bool Compressor::compress(int n) {
  if (!in_) error("no input");
  if (state_ == SEG1) postProcess();
  char buf[1 << 14];
  while (n) {
    int want = (int)sizeof(buf);
    if (n >= 0 && n < want) want = n;
    const int got = in_->read(buf, want);
    if (got < 0 || got > want) error("invalid read size");
    if (got <= 0) return false;
    if (n >= 0) n -= got;
    pending_.insert(pending_.end(), (const U8*)buf, (const U8*)buf + got);
  }
  return true;
}

void Compressor::flush_segment() {
  if (state_ == SEG1) postProcess();
  if (verify_) {
    // what the decompresser's PostProcessor will produce from this segment: its SHA-1 and size are what
    // endSegmentChecksum() reports (libzpaq.cpp:2935-2939, 2976-2991)
    guarded([&] {
      if (!pp_) pp_ = new zpq::PostProcessor(header_[4], header_[5]);
      std::vector<U8> data;
      ((zpq::PostProcessor*)pp_)->segment(pending_.data(), pending_.size(), data);
      seg_sha1_.write((const char*)data.data(), (int64_t)data.size());
    });
  }
  if (header_[6] == 0) {
    std::vector<U8> framed;
    zpq::write_stored_payload(framed, 0, 0, pending_.data(), pending_.size());
    if (!framed.empty()) out_->write((const char*)framed.data(), (int)framed.size());
    for (int i = 0; i < 4; ++i) out_->put(0);
  } else {
    segq_.back().data.swap(pending_);        // coded at endBlock(), together with the block's other segments
  }
  pending_.clear();
}

static void put_trailer(std::vector<U8>& t, const char* sha1) {
  if (sha1) { t.push_back(253); t.insert(t.end(), (const U8*)sha1, (const U8*)sha1 + 20); }
  else t.push_back(254);
}

void Compressor::endSegment(const char* sha1string) {
  if (header_.size() < 7) error("no block started");     // the reference asserts (and reads a null header under NDEBUG)
  flush_segment();
  std::vector<U8> t;
  put_trailer(t, sha1string);
  if (header_[6] == 0) out_->write((const char*)t.data(), (int)t.size());
  else segq_.back().tail.swap(t);
  state_ = BLOCK2;
}

char* Compressor::endSegmentChecksum(int64_t* size, bool dosha1) {
  if (header_.size() < 7) error("no block started");     // the reference asserts (and reads a null header under NDEBUG)
  flush_segment();
  if (verify_) {
    if (size) *size = (int64_t)seg_sha1_.usize();
    memcpy(sha1result_, seg_sha1_.result(), 20);       // result() also resets the hash for the next segment
  }
  std::vector<U8> t;
  put_trailer(t, verify_ && dosha1 ? sha1result_ : 0);
  if (header_[6] == 0) out_->write((const char*)t.data(), (int)t.size());
  else segq_.back().tail.swap(t);
  state_ = BLOCK2;
  return verify_ ? sha1result_ : 0;
}

void Compressor::endBlock() {
  if (header_.size() < 7) error("no block started");     // the reference asserts (and reads a null header under NDEBUG)
  if (header_[6] != 0 && !segq_.empty()) {
    // the block's segments through the model in one device job (Predictor and Encoder are initialised once per block,
    // libzpaq.cpp:2889-2891), then the bytes in archive order
    std::vector<std::vector<U8> > inputs, coded;
    for (size_t i = 0; i < segq_.size(); ++i) inputs.push_back(segq_[i].data);
    guarded([&] { coded = zpq::encode_payload_segments(header_, inputs); });
    for (size_t i = 0; i < segq_.size(); ++i) {
      out_->write((const char*)segq_[i].head.data(), (int)segq_[i].head.size());
      size_t pos = 0;
      while (pos < coded[i].size()) {
        const size_t k = std::min<size_t>(coded[i].size() - pos, 1u << 30);
        out_->write((const char*)coded[i].data() + pos, (int)k);
        pos += k;
      }
      for (int z = 0; z < 4; ++z) out_->put(0);
      out_->write((const char*)segq_[i].tail.data(), (int)segq_[i].tail.size());
    }
    segq_.clear();
  }
  out_->put(255);
  state_ = INIT;
}

}  // namespace libzpaq
