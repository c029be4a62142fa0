// Shared host-side declarations for the zpaq_amd library (not part of the ABI).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/zpaq_amd.h"

namespace zpq {

typedef uint8_t U8;
typedef uint16_t U16;
typedef uint32_t U32;
typedef uint64_t U64;

// Internal failure carrying a ZPQ_E_* code; converted to a return code (C ABI)
// or to libzpaq::error() (C++ API) at the boundary.
struct Failure : public std::runtime_error {
  int code;
  Failure(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
[[noreturn]] inline void fail(int code, const std::string& msg) { throw Failure(code, msg); }

void set_last_error(const std::string& s);

// ---- sha1.cpp ----
void sha1_compress(U32 h[5], const U8* block64);   // one 64-byte block
void postproc_set_step_limit(U64 steps);             // ZPAQL steps one call of a custom PCOMP program may take (0: 2^34)
unsigned usable_cpus();                             // host CPUs this process may really use: affinity mask and cgroup CPU quota
void sha1_force_portable(bool yes);                 // tests: the portable compression function instead of the SHA extensions
class Sha1 {
 public:
  Sha1() { reset(); }
  void reset();
  void update(const void* data, size_t n);
  void put(int c) { U8 b = (U8)c; update(&b, 1); }
  // Finishes, returns 20 bytes and resets (like libzpaq::SHA1::result()).
  const U8* result();
  U64 size() const { return len_; }
 private:
  void block(const U8* p);
  U32 h_[5];
  U64 len_;
  U8 buf_[64];
  U8 out_[20];
};

// ---- tables.cpp ----
struct Tables {
  U16 squash[4096];
  int16_t stretch[32768];
  int32_t dt[1024];
  int32_t dt2k[256];
  U8 ns[1024];
  // ICM / ISSE initial side tables (Predictor::init libzpaq.cpp:1795, 1834-1837)
  U32 icm_init[256];
  U32 isse_init[512];
  // SSE row pattern without the `start` count (1844): squash((j&31)*64-992)<<17
  U32 sse_row[32];
  // compact stretch for LDS (device/pipe_kernel.h): x in [16384, 32512) in groups of 8 as value-at-group-start |
  // 7 one-bit increments << 16; x >= 32512 direct; x < 16384 by the mirror rule stretch(x) = -stretch(32767 - x)
  U32 stretch_cb[2016];
  int16_t stretch_top[256];
};
// Built once from closed forms, verified against the reference's checksums
// (libzpaq.cpp:1759-1760); throws Failure(ZPQ_E_DEVICE) if they do not hold.
const Tables& tables();

// ---- zpaql_asm.cpp ----
struct Assembled {
  std::vector<U8> hcomp;  // block header as stored: hsize16 hh hm ph pm n COMP 0 HCOMP 0
  std::vector<U8> pcomp;  // len16 + PCOMP code (incl. trailing 0), empty if none
  std::string pcomp_cmd;  // text between "pcomp" and ";"
};
// Clean-room ZPAQL assembler (replaces libzpaq::Compiler, libzpaq.cpp:2494-2770).
Assembled assemble(const char* source, const int* args9);

// ---- method.cpp ----
// compressBlock's level expansion (libzpaq.cpp:7551-7691).
std::string expand_method(const std::string& method, const U8* data, U32 n);
// makeConfig (libzpaq.cpp:6887-7535) restated: "x.." -> ZPAQL source; fills args[9].
std::string make_config(const std::string& xmethod, int args[9]);

// ---- preproc.cpp: compression-side pre-processors (libzpaq.cpp:6450-6883) ----
void e8e9_forward(U8* buf, U32 n);
void e8e9_inverse(U8* buf, U32 n);
std::vector<U32> suffix_array(const U8* in, U32 n);
// false: the (possibly E8E9-filtered, in place) input itself is coded; true: `out` holds the LZ77 / BWT stream
// sa = the suffix array of the (E8E9-filtered) block when the caller already has it (built on the device for a whole
// batch: device/sa_kernels.hip), e8e9_done = the caller applied e8e9_forward itself (it must, before sorting)
bool preprocess_block(U8* data, U32 n, const int args[9], std::vector<U8>& out, const U32* sa = nullptr, bool e8e9_done = false);
// does this method's pre-processor sort the block's suffixes (BWT, or LZ77 searching through a suffix array)?
bool preprocess_needs_suffix_array(const int args[9]);
// The LZ77 parse through a suffix array as a list of matches, in order: the search stood at `i`, `blit` literals in front of
// the match belong to it (a look-ahead match), then `len` bytes from `off` back.  Every position the list does not cover is
// a literal.  lz77_host_tokens: the host's parse; lz77_serialize: the coded stream (LZBuffer's byte-aligned or bit-packed
// codes, libzpaq.cpp:6647-6883) from a list -- the host's or the one device/lz77_kernel.h made for a whole batch.
struct LzToken { U32 i, off, len, blit; };
void lz77_host_tokens(const U8* data, U32 n, const int args[9], const U32* sa, std::vector<LzToken>& toks);
void lz77_serialize(const U8* data, U32 n, const int args[9], const LzToken* toks, size_t ntok, std::vector<U8>& out);

}  // namespace zpq
