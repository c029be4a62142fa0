// Exercises the libzpaq-compatible C++ API (include/libzpaq.h) the way a libzpaq caller would.
//   compat_test host   : only paths that need no GPU (stored blocks, container logic, SHA1, errors)
//   compat_test gpu    : everything, including the modelled methods
// Prints "COMPAT_OK <n checks>" on success.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "libzpaq.h"

// the application supplies error(), exactly as with the reference (libzpaq.h:858)
void libzpaq::error(const char* msg) { throw std::runtime_error(msg); }

static int checks = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } ++checks; } while (0)

static std::string text(size_t n, unsigned seed) {
  static const char* words[] = {"alpha", "beta", "gamma", "delta", "the", "of", "zpaq", "context", "mixing", "wave"};
  std::string s;
  unsigned x = seed;
  while (s.size() < n) {
    x = x * 1664525u + 1013904223u;
    s += words[(x >> 24) % 10];
    s += (x >> 20) % 16 ? ' ' : '\n';
  }
  s.resize(n);
  return s;
}

static std::string roundtrip_stream(const std::string& data, const char* method) {
  libzpaq::StringBuffer in, arc, out;
  in.write(data.data(), (int)data.size());
  libzpaq::compress(&in, &arc, method, "file.txt", "note");
  libzpaq::decompress(&arc, &out);
  return std::string(out.c_str() ? out.c_str() : "", out.size());
}

int main(int argc, char** argv) {
  const bool gpu = argc > 1 && std::string(argv[1]) == "gpu";
  try {
    // SHA1
    libzpaq::SHA1 sha;
    sha.write("abc", 3);
    const unsigned char want[20] = {0xa9, 0x99, 0x3e, 0x36, 0x47, 0x06, 0x81, 0x6a, 0xba, 0x3e,
                                    0x25, 0x71, 0x78, 0x50, 0xc2, 0x6c, 0x9c, 0xd0, 0xd8, 0x9d};
    CHECK(memcmp(sha.result(), want, 20) == 0);
    CHECK(libzpaq::toU16("\x34\x12") == 0x1234);
    libzpaq::Array<libzpaq::U32> arr(3, 2);
    CHECK(arr.size() == 12 && arr[5] == 0 && ((uintptr_t)&arr[0] & 63) == 0);

    // stored method: host only
    const std::string d0 = text(70000, 1);
    CHECK(roundtrip_stream(d0, "0") == d0);
    CHECK(roundtrip_stream("", "0").empty());

    // Compressor / Decompresser streaming interface on a stored block, two segments
    {
      libzpaq::StringBuffer arc, s1, s2;
      s1.write(d0.data(), 1000);
      s2.write(d0.data() + 1000, 2000);
      libzpaq::Compressor co;
      co.setOutput(&arc);
      co.writeTag();
      co.startBlock("comp 0 0 0 0 0 hcomp end\n", 0);
      co.startSegment("a", "1000");
      co.setInput(&s1);
      while (co.compress(300)) {}
      co.endSegment(0);
      co.startSegment("b", "2000");
      co.setInput(&s2);
      co.compress(-1);
      libzpaq::SHA1 h2;
      h2.write(d0.data() + 1000, 2000);
      co.endSegment(h2.result());
      co.endBlock();

      libzpaq::Decompresser de;
      de.setInput(&arc);
      double mem = -1;
      CHECK(de.findBlock(&mem));
      libzpaq::StringBuffer fn, cm, out;
      CHECK(de.findFilename(&fn));
      de.readComment(&cm);
      CHECK(std::string(fn.c_str(), fn.size()) == "a" && std::string(cm.c_str(), cm.size()) == "1000");
      de.setOutput(&out);
      CHECK(de.decompress(400));             // first 400 bytes only
      CHECK(out.size() == 400);
      while (de.decompress(250)) {}
      CHECK(out.size() == 1000 && memcmp(out.c_str(), d0.data(), 1000) == 0);
      char sh[21];
      de.readSegmentEnd(sh);
      CHECK(sh[0] == 0);
      fn.reset(); CHECK(de.findFilename(&fn));
      de.readComment();
      de.readSegmentEnd(sh);                 // skip the second segment without decoding it
      CHECK(sh[0] == 1);
      libzpaq::SHA1 h3; h3.write(d0.data() + 1000, 2000);
      CHECK(memcmp(sh + 1, h3.result(), 20) == 0);
      CHECK(!de.findFilename());
      CHECK(!de.findBlock());
    }

    // error() contract: bad method / bad config must call error(), which throws here
    bool threw = false;
    try { libzpaq::StringBuffer a, b; a.write("x", 1); libzpaq::compressBlock(&a, &b, "x0,9"); } catch (std::runtime_error&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { libzpaq::Compressor c; libzpaq::StringBuffer b; c.setOutput(&b); c.startBlock("comp 0 0 0 0 1 0 nope hcomp end", 0); }
    catch (std::runtime_error&) { threw = true; }
    CHECK(threw);

    if (gpu) {
      const std::string d1 = text(60000, 7);
      CHECK(roundtrip_stream(d1, "5") == d1);
      CHECK(roundtrip_stream(d1, "4") == d1);
      // batched extension: 5 buffers -> 5 blocks in one device batch, then one decompress() over all
      std::vector<libzpaq::StringBuffer*> ins;
      std::vector<libzpaq::Writer*> outs;
      libzpaq::StringBuffer arc;
      std::string all;
      for (int i = 0; i < 5; ++i) {
        const std::string d = text(20000 + 7000 * i, 100 + i);
        libzpaq::StringBuffer* sb = new libzpaq::StringBuffer;
        sb->write(d.data(), (int)d.size());
        ins.push_back(sb);
        outs.push_back(&arc);
        all += d;
      }
      libzpaq::compressBlocks(ins.data(), outs.data(), 5, "5");
      libzpaq::StringBuffer out;
      libzpaq::decompress(&arc, &out);
      CHECK(std::string(out.c_str(), out.size()) == all);
      for (auto* p : ins) delete p;

      // Compressor with an explicit ZPAQL config (max-style chain fragment) + streaming Decompresser
      const char* cfg = "comp 2 8 0 0 3 0 icm 12 1 isse 14 0 2 mix 8 0 2 24 255 hcomp c++ *c=a b=c a=0 d= 1 hash *d=a halt end\n";
      libzpaq::StringBuffer src, a2, o2;
      src.write(d1.data(), 30000);
      libzpaq::Compressor co;
      co.setOutput(&a2);
      co.setInput(&src);
      co.writeTag();
      co.startBlock(cfg, 0);
      co.startSegment("seg", 0);
      co.compress(-1);
      co.endSegment(0);
      co.endBlock();
      libzpaq::Decompresser de;
      de.setInput(&a2);
      CHECK(de.findBlock());
      CHECK(de.findFilename());
      de.readComment();
      de.setOutput(&o2);
      CHECK(de.decompress(12345));
      CHECK(o2.size() == 12345);
      de.decompress(-1);
      de.readSegmentEnd();
      CHECK(o2.size() == 30000 && memcmp(o2.c_str(), d1.data(), 30000) == 0);
    }
  } catch (std::exception& e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 2;
  }
  printf("COMPAT_OK %d\n", checks);
  return 0;
}
