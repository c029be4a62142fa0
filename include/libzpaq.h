// include/libzpaq.h -- libzpaq-compatible C++ surface of zpaq_amd.
//
// A caller written against zpaq 7.15's libzpaq (reference libzpaq.h:858-876,
// 934-954, 1243-1268, 1340-1506) can include this header instead and link
// libzpaq_amd.so: same namespace, class names, method names, argument meaning
// and error behaviour for the hot path (context-mixing model + arithmetic
// coder), which runs on an MI355X through the C ABI of zpaq_amd.h.
//
// What is here:   Reader, Writer, error(), toU16, Array<T>, SHA1, StringBuffer,
//                 compress(), compressBlock(), decompress(), Compressor,
//                 Decompresser, plus the batched extension compressBlocks().
//                 Every method compressBlock() accepts is served: levels 0-5,
//                 explicit x/s methods, the LZ77 / BWT / E8E9 pre-processors and
//                 the PCOMP post-processors they put into the archive.
// What is not:    encryption (AES_CTR / stretchKey are declared so that zpaq.cpp
//                 links, and call error()).
//
// Threading: like the reference, every function is re-entrant; calls from many
// threads are serialised on the device queue.  error() must not return (it may
// throw or exit the thread), exactly as the reference requires (libzpaq.h:858).
#ifndef ZPAQ_AMD_LIBZPAQ_H
#define ZPAQ_AMD_LIBZPAQ_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace libzpaq {

typedef uint8_t U8;
typedef uint16_t U16;
typedef uint32_t U32;
typedef uint64_t U64;

// Application-supplied fatal-error hook.  libzpaq_amd.so carries a weak default
// that throws std::runtime_error; a definition in the application overrides it.
extern void error(const char* msg);

// Byte source / sink (reference libzpaq.h:864-876).
class Reader {
 public:
  virtual int get() = 0;                 // next byte 0..255, or -1 at end of input
  virtual int read(char* buf, int n);    // bulk read, default loops over get()
  virtual ~Reader() {}
};

class Writer {
 public:
  virtual void put(int c) = 0;                   // write the low 8 bits of c
  virtual void write(const char* buf, int n);    // bulk write, default loops over put()
  virtual ~Writer() {}
};

int toU16(const char* p);   // little-endian 16-bit value at p

// Zero-initialised, 64-byte aligned array without constructors
// (reference libzpaq.h:887-929): a[i] unchecked, a(i) index mod size (size 2^k).
template <typename T>
class Array {
 public:
  explicit Array(size_t sz = 0, int ex = 0) : base_(0), data_(0), n_(0) { resize(sz, ex); }
  ~Array() { resize(0); }
  void resize(size_t sz, int ex = 0) {
    while (ex-- > 0) {
      if (sz > sz * 2) error("Array too big");
      sz *= 2;
    }
    if (base_) ::free(base_);
    base_ = 0; data_ = 0; n_ = 0;
    if (!sz) return;
    const size_t bytes = sz * sizeof(T);
    if (bytes / sizeof(T) != sz || bytes + 128 < bytes) error("Array too big");
    base_ = ::calloc(bytes + 128, 1);
    if (!base_) error("Out of memory");
    data_ = (T*)(((uintptr_t)base_ + 64) & ~(uintptr_t)63);
    n_ = sz;
  }
  size_t size() const { return n_; }
  int isize() const { return (int)n_; }
  T& operator[](size_t i) { return data_[i]; }
  T& operator()(size_t i) { return data_[i & (n_ - 1)]; }
 private:
  Array(const Array&);
  void operator=(const Array&);
  void* base_;
  T* data_;
  size_t n_;
};

// SHA-1 (reference libzpaq.h:934-954).
class SHA1 {
 public:
  SHA1();
  void put(int c);
  void write(const char* buf, int64_t n);
  double size() const { return (double)len_; }
  uint64_t usize() const { return len_; }
  const char* result();    // 20-byte digest; resets the object
 private:
  void block(const U8* p);
  U32 h_[5];
  U64 len_;
  U8 buf_[64];
  char out_[20];
};

// ---- archiver-only services of the reference library (libzpaq.h:956-1015) ----
// zpaq.cpp needs these names to link.  SHA-256 and random() are real; encryption (AES in CTR mode keyed through
// scrypt) is outside this library's scope: constructing an AES_CTR or calling stretchKey() calls error().
class SHA256 {
 public:
  SHA256() { init(); }
  void put(int c);
  double size() const { return (double)len_; }
  uint64_t usize() const { return len_; }
  const char* result();    // 32-byte digest; resets the object
 private:
  void init();
  void block();
  U32 s_[8];
  U64 len_;
  U8 buf_[64];
  char out_[32];
};

class AES_CTR {
 public:
  AES_CTR(const char* key, int keylen, const char* iv = 0);
  void encrypt(U32 s0, U32 s1, U32 s2, U32 s3, unsigned char* ct);
  void encrypt(char* buf, int n, U64 offset);
};

void stretchKey(char* out, const char* key, const char* salt);
void random(char* buf, int n);    // n random bytes, the first never '7' or 'z' (a ZPAQ archive cannot start with them)

// In-memory Reader+Writer (reference libzpaq.h:1377-1494).
class StringBuffer : public Reader, public Writer {
 public:
  explicit StringBuffer(size_t n = 0) : p_(0), al_(0), wpos_(0), rpos_(0), limit_((size_t)-1), init_(n > 128 ? n : 128) {}
  ~StringBuffer() { if (p_) free(p_); }
  void setLimit(size_t n) { limit_ = n; }
  unsigned char* data() { return p_; }
  const char* c_str() const { return (const char*)p_; }
  size_t size() const { return wpos_; }
  size_t remaining() const { return wpos_ - rpos_; }
  void reset() { if (p_) free(p_); p_ = 0; al_ = rpos_ = wpos_ = 0; }
  void put(int c) { grow(1); p_[wpos_++] = (unsigned char)c; }
  void write(const char* buf, int n) {
    if (n < 1) return;
    grow((size_t)n);
    if (buf) memcpy(p_ + wpos_, buf, (size_t)n);
    wpos_ += (size_t)n;
  }
  int get() { return rpos_ < wpos_ ? p_[rpos_++] : -1; }
  int read(char* buf, int n) {
    if (rpos_ + (size_t)n > wpos_) n = (int)(wpos_ - rpos_);
    if (n > 0 && buf) memcpy(buf, p_ + rpos_, (size_t)n);
    rpos_ += (size_t)n;
    return n;
  }
  void resize(size_t i) { wpos_ = i; if (rpos_ > wpos_) rpos_ = wpos_; }
  void swap(StringBuffer& s) {
    std::swap(p_, s.p_); std::swap(al_, s.al_); std::swap(wpos_, s.wpos_);
    std::swap(rpos_, s.rpos_); std::swap(limit_, s.limit_);
  }
 private:
  StringBuffer(const StringBuffer&);
  void operator=(const StringBuffer&);
  void grow(size_t n) {
    if (wpos_ + n > limit_ || wpos_ + n < wpos_) error("StringBuffer overflow");
    if (wpos_ + n <= al_) return;
    size_t a = al_;
    while (wpos_ + n >= a) a = a * 2 + init_;
    unsigned char* q = (unsigned char*)(p_ ? realloc(p_, a) : malloc(a));
    if (!q) error("Out of memory");
    p_ = q; al_ = a;
  }
  unsigned char* p_;
  size_t al_, wpos_, rpos_, limit_;
  const size_t init_;
};

// ---- one-call API (reference libzpaq.h:1268, 1501-1506) ----
// Stream -> blocks of (2^(20+B) - 4096) bytes, B from the method's second
// digit (default 4); blocks are gathered and coded on the GPU in batches.
void compress(Reader* in, Writer* out, const char* method, const char* filename = 0,
              const char* comment = 0, bool dosha1 = true);
// One buffer -> one block with one segment.  May modify *in (E8E9 methods).
void compressBlock(StringBuffer* in, Writer* out, const char* method, const char* filename = 0,
                   const char* comment = 0, bool dosha1 = true);
// Extension: n independent buffers -> n blocks in ONE device batch; out[i]
// receives block i.  This is the call a multi-threaded archiver should use
// instead of n concurrent compressBlock() calls.
void compressBlocks(StringBuffer* const* in, Writer* const* out, int n, const char* method,
                    const char* const* filename = 0, const char* const* comment = 0, bool dosha1 = true);
// ... the same with a method per block: what an archiver's queue of blocks holds (zpaq.cpp gives every block its own
// redundancy / type hints: the method string is built per block, zpaq.cpp:2501-2504, from the statistics of 2399-2471); patches/zpaq_batch.patch hands CompressJob's whole queue to it.
void compressBlocks(StringBuffer* const* in, Writer* const* out, int n, const char* const* methods,
                    const char* const* filename = 0, const char* const* comment = 0, bool dosha1 = true);
// Every block and segment of `in`, concatenated, to `out`.
void decompress(Reader* in, Writer* out);

// ---- streaming decoder (reference libzpaq.h:1243-1264) ----
class Decompresser {
 public:
  Decompresser();
  ~Decompresser();
  void setInput(Reader* in) { in_ = in; }
  bool findBlock(double* memptr = 0);        // locate next block; *memptr = model memory
  void hcomp(Writer* out2);                  // stored COMP+HCOMP header
  bool findFilename(Writer* filename = 0);   // true: a segment follows
  void readComment(Writer* comment = 0);
  void setOutput(Writer* out) { out_ = out; }
  void setSHA1(SHA1* s) { sha1_ = s; }
  bool decompress(int n = -1);               // n more OUTPUT bytes (-1: to end of segment); false at end.  A first call with a
                                             // small n decodes only the segment's first 64 KiB on the device.  (The reference
                                             // counts n bytes into the post-processor and flushes its output every 64 KiB:
                                             // same totals, different pieces for LZ77 / BWT blocks.)
  bool pcomp(Writer* out2);
  void readSegmentEnd(char* sha1string = 0); // 21 bytes: flag + digest
  int buffered() { return (int)(buf_.size() - rpos_); }
 private:
  Decompresser(const Decompresser&);
  void operator=(const Decompresser&);
  int getc();
  void decode_segment();
  void try_prefix();
  void peek_block_payloads(std::vector<std::vector<U8> >& payloads);
  Reader* in_;
  Writer* out_;
  SHA1* sha1_;
  std::vector<U8> buf_;      // read-ahead window over the input
  size_t rpos_;
  std::vector<U8> header_;
  void* plan_;
  std::vector<U8> decoded_;  // current segment, fully decoded on the device
  size_t dpos_;
  size_t payload_end_;
  bool seg_decoded_;
  int segs_in_block_;
  void* pp_;                 // PostProcessor of the current block (first segment carries the PP header)
  std::vector<std::vector<U8> > block_cache_;   // modelled block of several segments: all of them, decoded together
  bool skipped_in_block_;
  std::vector<U8> prefix_;   // decompress(n) with a small n: the first bytes of the segment, decoded without the rest
  bool prefix_tried_, prefix_active_;
  int peek(size_t off);      // byte at rpos_ + off without consuming it (-1 at EOF)
  enum { BLOCK, FILENAME, COMMENT, DATA, SEGEND } state_;
};

// ---- streaming encoder (reference libzpaq.h:1340-1371) ----
class Compressor {
 public:
  Compressor();
  ~Compressor();
  void setOutput(Writer* out) { out_ = out; }
  void writeTag();
  void startBlock(int level);                 // built-in models 1..3 (min/mid/max)
  void startBlock(const char* hcomp);         // stored header bytes
  void startBlock(const char* config, int* args, Writer* pcomp_cmd = 0);   // ZPAQL source
  void setVerify(bool v) { verify_ = v; }     // endSegmentChecksum() then hashes the post-processed input
  void hcomp(Writer* out2);
  bool pcomp(Writer* out2);
  void startSegment(const char* filename = 0, const char* comment = 0);
  void setInput(Reader* i) { in_ = i; }
  void postProcess(const char* pcomp = 0, int len = 0);
  bool compress(int n = -1);                  // gather n bytes (-1: to EOF); coded at endSegment
  void endSegment(const char* sha1string = 0);
  char* endSegmentChecksum(int64_t* size = 0, bool dosha1 = true);
  int64_t getSize() { return (int64_t)seg_sha1_.usize(); }
  const char* getChecksum() { return seg_sha1_.result(); }
  void endBlock();
 private:
  Compressor(const Compressor&);
  void operator=(const Compressor&);
  void flush_segment();
  Writer* out_;
  Reader* in_;
  std::vector<U8> header_, pcomp_;
  std::vector<U8> pending_;   // PP header + segment bytes awaiting the device
  // a modelled block is coded when it ends (all its segments in ONE device job: the model runs on across them):
  // per segment its header bytes, its data, its trailer
  struct SegBuf { std::vector<U8> head, data, tail; };
  std::vector<SegBuf> segq_;
  SHA1 seg_sha1_;
  char sha1result_[20];
  bool verify_;
  void* pp_;                  // verify mode: PostProcessor fed with what the decompresser will see
  int segs_;
  enum { INIT, BLOCK1, SEG1, BLOCK2, SEG2 } state_;
};

}  // namespace libzpaq

#endif  // ZPAQ_AMD_LIBZPAQ_H
