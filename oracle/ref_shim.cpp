// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A thin extern "C" driver around the UNMODIFIED reference library, which is
// compiled where it lies (/root/reference/libzpaq.cpp) by oracle/Makefile into
// oracle/_ref/libzpaq_ref.so.  No reference source is copied into this repo:
// this file only *calls* the public libzpaq API (libzpaq.h:858-876, 1243-1268,
// 1340-1371, 1501-1506) plus the linkable-but-undeclared helper
// libzpaq::makeConfig (libzpaq.cpp:6887).
//
// Used by: tests/ (golden generation + differential parity), bench.py's
// cpu_baseline leg ("kind":"reference").
#include "libzpaq.h"   // found with -I/root/reference at oracle build time
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>
#include <stdexcept>
#include <cstring>

namespace libzpaq {
// The reference requires the application to define error(); it must not return.
void error(const char* msg) { throw std::runtime_error(msg ? msg : "libzpaq error"); }
// Not declared in libzpaq.h but has external linkage (libzpaq.cpp:6887).
std::string makeConfig(const char* method, int args[]);
}

namespace {
thread_local std::string g_err;

struct CapWriter : public libzpaq::Writer {
  unsigned char* p; size_t cap, n;
  CapWriter(unsigned char* p_, size_t cap_) : p(p_), cap(cap_), n(0) {}
  void put(int c) override { if (n < cap) p[n] = (unsigned char)c; ++n; }
  void write(const char* buf, int len) override {
    for (int i = 0; i < len; ++i) put((unsigned char)buf[i]);
  }
};
struct MemReader : public libzpaq::Reader {
  const unsigned char* p; size_t n, pos;
  MemReader(const unsigned char* p_, size_t n_) : p(p_), n(n_), pos(0) {}
  int get() override { return pos < n ? p[pos++] : -1; }
  int read(char* buf, int len) override {
    size_t k = n - pos; if ((size_t)len < k) k = len;
    memcpy(buf, p + pos, k); pos += k; return (int)k;
  }
};
}  // namespace

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }

// libzpaq::compressBlock (libzpaq.cpp:7543) on one buffer.  Returns the archive
// size (may exceed cap: then output is truncated) or -1 on error().
long long ref_compress_block(const unsigned char* in, size_t n, const char* method,
                             const char* filename, const char* comment, int dosha1,
                             unsigned char* out, size_t cap) {
  try {
    libzpaq::StringBuffer sb(n + 8);
    if (n) sb.write((const char*)in, (int)n);
    CapWriter w(out, cap);
    libzpaq::compressBlock(&sb, &w, method, filename, comment, dosha1 != 0);
    return (long long)w.n;
  } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// libzpaq::compress (libzpaq.cpp:3008): stream -> blocks.
long long ref_compress(const unsigned char* in, size_t n, const char* method,
                       const char* filename, const char* comment, int dosha1,
                       unsigned char* out, size_t cap) {
  try {
    MemReader r(in, n); CapWriter w(out, cap);
    libzpaq::compress(&r, &w, method, filename, comment, dosha1 != 0);
    return (long long)w.n;
  } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// libzpaq::decompress (libzpaq.cpp:2378).
long long ref_decompress(const unsigned char* in, size_t n, unsigned char* out, size_t cap) {
  try {
    MemReader r(in, n); CapWriter w(out, cap);
    libzpaq::decompress(&r, &w);
    return (long long)w.n;
  } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// Legacy built-in models: Compressor::startBlock(int level) (libzpaq.cpp:2793),
// one block, one segment, optional SHA-1 trailer.
long long ref_compress_level(const unsigned char* in, size_t n, int level,
                             const char* filename, const char* comment, int dosha1,
                             unsigned char* out, size_t cap) {
  try {
    MemReader r(in, n); CapWriter w(out, cap);
    libzpaq::Compressor co;
    co.setOutput(&w);
    co.setInput(&r);
    co.writeTag();
    co.startBlock(level);
    co.startSegment(filename, comment);
    co.compress(-1);
    if (dosha1) {
      libzpaq::SHA1 s; s.write((const char*)in, (int64_t)n);
      co.endSegment(s.result());
    } else co.endSegment(0);
    co.endBlock();
    return (long long)w.n;
  } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// One block of SEVERAL segments through a built-in model (libzpaq.cpp:2889-2891: the model and the coder are
// initialised once per block and run on across its segments).  Segment i is named "s<i>", has a SHA-1 trailer.
long long ref_compress_level_segments(const unsigned char* in, const size_t* lens, int nseg, int level,
                                      unsigned char* out, size_t cap) {
  try {
    CapWriter w(out, cap);
    libzpaq::Compressor co;
    co.setOutput(&w);
    co.writeTag();
    co.startBlock(level);
    size_t pos = 0;
    for (int i = 0; i < nseg; ++i) {
      MemReader r(in + pos, lens[i]);
      co.setInput(&r);
      std::string name = "s" + std::to_string(i);
      co.startSegment(name.c_str(), 0);
      co.compress(-1);
      libzpaq::SHA1 s; s.write((const char*)in + pos, (int64_t)lens[i]);
      co.endSegment(s.result());
      pos += lens[i];
    }
    co.endBlock();
    return (long long)w.n;
  } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// Arbitrary ZPAQL source config through Compressor::startBlock(config,args)
// (libzpaq.cpp:2856) -- used to exercise all nine component types.
long long ref_compress_config(const unsigned char* in, size_t n, const char* config,
                              const int* args9, const char* filename, const char* comment,
                              int dosha1, unsigned char* out, size_t cap) {
  try {
    int args[9] = {0};
    if (args9) memcpy(args, args9, sizeof(args));
    MemReader r(in, n); CapWriter w(out, cap);
    libzpaq::Compressor co;
    co.setOutput(&w);
    co.setInput(&r);
    co.writeTag();
    co.startBlock(config, args);
    co.startSegment(filename, comment);
    co.compress(-1);
    if (dosha1) {
      libzpaq::SHA1 s; s.write((const char*)in, (int64_t)n);
      co.endSegment(s.result());
    } else co.endSegment(0);
    co.endBlock();
    return (long long)w.n;
  } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// makeConfig (libzpaq.cpp:6887): method string "x..." -> ZPAQL source + args[9].
long long ref_make_config(const char* method, int* args9, char* out, size_t cap) {
  try {
    int args[9] = {0};
    std::string s = libzpaq::makeConfig(method, args);
    if (args9) memcpy(args9, args, sizeof(args));
    size_t k = s.size() < cap ? s.size() : cap;
    memcpy(out, s.data(), k);
    return (long long)s.size();
  } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// Compiler (libzpaq.cpp:2698): ZPAQL source -> HCOMP header bytes (as written in
// the archive) and PCOMP bytes (as coded through the model: len16 + code).
long long ref_compile(const char* config, const int* args9, unsigned char* hcomp, size_t hcap,
                      unsigned char* pcomp, size_t pcap, long long* pcomp_len) {
  try {
    int args[9] = {0};
    if (args9) memcpy(args, args9, sizeof(args));
    CapWriter w(0, 0);
    libzpaq::Compressor co;
    co.setOutput(&w);
    co.startBlock(config, args);
    CapWriter h(hcomp, hcap), p(pcomp, pcap);
    co.hcomp(&h);
    bool has = co.pcomp(&p);
    if (pcomp_len) *pcomp_len = has ? (long long)p.n : 0;
    return (long long)h.n;
  } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// SHA1 (libzpaq.cpp:106-177).
void ref_sha1(const unsigned char* in, size_t n, unsigned char out20[20]) {
  libzpaq::SHA1 s; s.write((const char*)in, (int64_t)n);
  memcpy(out20, s.result(), 20);
}

// StateTable::ns (libzpaq.cpp:726-860).
void ref_state_table(unsigned char out1024[1024]) {
  libzpaq::StateTable st; memcpy(out1024, st.ns, 1024);
}

// Model memory as reported by Decompresser::findBlock(&mem) (libzpaq.cpp:2270).
double ref_block_memory(const unsigned char* archive, size_t n) {
  try {
    MemReader r(archive, n);
    libzpaq::Decompresser d; d.setInput(&r);
    double mem = 0;
    if (!d.findBlock(&mem)) return -1;
    return mem;
  } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// CPU baseline: compressBlock over nblocks equal-sized inputs from a pthread
// work queue, like zpaq.cpp:1918-1965 does.  Returns wall seconds; out_len[b]
// gets each archive's size.  Outputs themselves are discarded unless out!=0
// (then block b is written at out + b*out_stride).
// deadline_s > 0: no new block is started after that many seconds (blocks not
// started keep out_len[b] = -2), so the call is bounded by deadline + one block.
double ref_compress_blocks_mt(const unsigned char* in, size_t block_bytes, int nblocks,
                              const char* method, int nthreads, long long* out_len,
                              unsigned char* out, size_t out_stride, double deadline_s) {
  std::atomic<int> next(0);
  std::atomic<int> failed(0);
  auto t0 = std::chrono::steady_clock::now();
  if (out_len) for (int b = 0; b < nblocks; ++b) out_len[b] = -2;
  auto work = [&]() {
    std::vector<unsigned char> tmp;
    for (;;) {
      if (deadline_s > 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > deadline_s) break;
      int b = next.fetch_add(1);
      if (b >= nblocks) break;
      unsigned char* dst; size_t cap;
      if (out) { dst = out + (size_t)b * out_stride; cap = out_stride; }
      else { tmp.resize(block_bytes + block_bytes / 2 + 4096); dst = tmp.data(); cap = tmp.size(); }
      // method "L1" / "L2" / "L3": the built-in models through Compressor::startBlock(int level) (min / mid / max.cfg)
      long long r = (method && method[0] == 'L')
          ? ref_compress_level(in + (size_t)b * block_bytes, block_bytes, method[1] - '0', 0, 0, 1, dst, cap)
          : ref_compress_block(in + (size_t)b * block_bytes, block_bytes, method, 0, 0, 1, dst, cap);
      if (r < 0) failed = 1;
      if (out_len) out_len[b] = r;
    }
  };
  std::vector<std::thread> th;
  for (int i = 0; i < nthreads; ++i) th.emplace_back(work);
  for (auto& t : th) t.join();
  double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return failed ? -1.0 : s;
}

// Same for decompression (libzpaq::decompress per archive).
double ref_decompress_blocks_mt(const unsigned char* const* archives, const size_t* lens,
                                int nblocks, size_t block_bytes, int nthreads) {
  std::atomic<int> next(0);
  std::atomic<int> failed(0);
  auto t0 = std::chrono::steady_clock::now();
  auto work = [&]() {
    std::vector<unsigned char> tmp(block_bytes + 4096);
    for (;;) {
      int b = next.fetch_add(1);
      if (b >= nblocks) break;
      if (ref_decompress(archives[b], lens[b], tmp.data(), tmp.size()) < 0) failed = 1;
    }
  };
  std::vector<std::thread> th;
  for (int i = 0; i < nthreads; ++i) th.emplace_back(work);
  for (auto& t : th) t.join();
  double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return failed ? -1.0 : s;
}

}  // extern "C"
