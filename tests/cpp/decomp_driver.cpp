// Drives libzpaq::Decompresser over one archive the way zpaq.cpp's extract / list / test do (libzpaq.h:1280-1337:
// findBlock, hcomp, findFilename, readComment, decompress(n) in pieces or to the end, pcomp, readSegmentEnd, skipping
// segments) and prints what it saw, one line per event.  The SAME source compiles against include/libzpaq.h +
// libzpaq_amd.so and against the reference's libzpaq.h + libzpaq.cpp; tests/test_cpp_api.py compares the two outputs on
// valid and damaged archives (the drop-in claim of the C++ API, host-decoded blocks: no GPU needed).
//
//   decomp_driver <archive> <piece> <mode>
//     piece: decompress(piece) until false; -1 = decompress() once
//     mode:  0 read everything, 1 skip every second segment (no decompress call), 2 stop reading a segment half way,
//            3 stop after the FIRST decompress(piece) call of every segment and say how long it took (stderr: not compared)
#include <libzpaq.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

void libzpaq::error(const char* msg) { throw std::runtime_error(msg); }

// present in libzpaq_amd.so only: a damaged PCOMP program that loops is given up quickly (the reference never gives up;
// the test's timeout covers that side)
extern "C" void zpq_set_pcomp_step_limit(uint64_t) __attribute__((weak));

namespace {

struct In : libzpaq::Reader {
  std::string s;
  size_t p = 0;
  int get() override { return p < s.size() ? (unsigned char)s[p++] : -1; }
  int read(char* buf, int n) override {
    size_t k = s.size() - p < (size_t)n ? s.size() - p : (size_t)n;
    for (size_t i = 0; i < k; ++i) buf[i] = s[p + i];
    p += k;
    return (int)k;
  }
};

struct Out : libzpaq::Writer {
  std::string s;
  void put(int c) override { s.push_back((char)c); }
  void write(const char* buf, int n) override { s.append(buf, (size_t)n); }
};

std::string hex(const std::string& b, size_t limit = 1u << 30) {
  static const char* d = "0123456789abcdef";
  std::string r;
  for (size_t i = 0; i < b.size() && i < limit; ++i) { r.push_back(d[(unsigned char)b[i] >> 4]); r.push_back(d[b[i] & 15]); }
  return r;
}

unsigned long long fnv(const std::string& b) {
  unsigned long long h = 1469598103934665603ull;
  for (unsigned char c : b) h = (h ^ c) * 1099511628211ull;
  return h;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  if (zpq_set_pcomp_step_limit) zpq_set_pcomp_step_limit(1u << 22);
  In in;
  {
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) in.s.append(buf, n);
    fclose(f);
  }
  const int piece = atoi(argv[2]), mode = atoi(argv[3]);
  try {
    libzpaq::Decompresser d;
    d.setInput(&in);
    double mem = 0;
    int nseg = 0;
    while (d.findBlock(&mem)) {
      Out hdr;
      d.hcomp(&hdr);
      printf("block mem=%.0f header=%s\n", mem, hex(hdr.s).c_str());
      Out name;
      while (d.findFilename(&name)) {
        Out comment;
        d.readComment(&comment);
        printf("segment name=%s comment=%s\n", hex(name.s).c_str(), hex(comment.s).c_str());
        name.s.clear();
        const bool skip = mode == 1 && (nseg & 1);
        Out out;
        libzpaq::SHA1 sha;
        if (!skip) {
          d.setOutput(&out);
          d.setSHA1(&sha);
          if (piece < 0) {
            d.decompress();
          } else {
            int calls = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (d.decompress(piece)) {
              if (mode == 3) {
                fprintf(stderr, "first_call_ms=%.1f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                break;
              }
              if (mode == 2 && ++calls == 2) break;
            }
          }
          if (nseg == 0) {
            Out pc;
            const bool has = d.pcomp(&pc);
            printf("pcomp %d %s\n", (int)has, hex(pc.s, 16).c_str());
          }
          const unsigned long long len = (unsigned long long)sha.usize();
          printf("data n=%zu fnv=%016llx sha_n=%llu sha=%s\n", out.s.size(), fnv(out.s), len, hex(std::string(sha.result(), 20)).c_str());
        }
        char tail[21] = {0};
        d.readSegmentEnd(tail);
        printf("end flag=%d sha=%s\n", tail[0], tail[0] ? hex(std::string(tail + 1, 20)).c_str() : "");
        ++nseg;
      }
    }
    printf("done consumed=%zu of %zu\n", in.p - (size_t)d.buffered(), in.s.size());
  } catch (std::exception& e) {
    printf("error\n");
    fprintf(stderr, "%s\n", e.what());
  }
  return 0;
}
