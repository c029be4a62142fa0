// Drives libzpaq::Compressor through call sequences of its public interface (libzpaq.h:1340-1371) on blocks WITHOUT a
// model (n = 0: coded on the host, no GPU needed) and prints the archive and what the calls returned.  The SAME source
// compiles against include/libzpaq.h + libzpaq_amd.so and against the reference's libzpaq.h + libzpaq.cpp;
// tests/test_cpp_api.py compares the two outputs line for line.
//
//   comp_driver <scenario> <seed>
#include <libzpaq.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

void libzpaq::error(const char* msg) { throw std::runtime_error(msg); }

namespace {

struct In : libzpaq::Reader {
  std::string s;
  size_t p = 0;
  int get() override { return p < s.size() ? (unsigned char)s[p++] : -1; }
};

struct Out : libzpaq::Writer {
  std::string s;
  void put(int c) override { s.push_back((char)c); }
};

std::string hex(const std::string& b) {
  static const char* d = "0123456789abcdef";
  std::string r;
  for (unsigned char c : b) { r.push_back(d[c >> 4]); r.push_back(d[c & 15]); }
  return r;
}

unsigned rnd_state = 1;
unsigned rnd() { rnd_state = rnd_state * 1103515245u + 12345u; return rnd_state >> 16 & 0x7fff; }

std::string data(size_t n, int kind) {
  std::string s;
  for (size_t i = 0; i < n; ++i) {
    if (kind == 0) s.push_back((char)("the quick brown fox "[i % 20]));
    else if (kind == 1) s.push_back((char)(rnd() & 255));
    else s.push_back((char)(i % 7 == 0 ? 0xE8 : (rnd() & 3)));
  }
  return s;
}

// identity post-processor, and one that adds 1 to every byte (the caller subtracts 1 first): exercised with setVerify
const char* kPass = "comp 0 0 0 0 0 hcomp end\n";
const char* kIdentity = "comp 0 0 0 0 0 hcomp pcomp cat ; a> 255 ifnot out endif halt end\n";
const char* kPlusOne = "comp 1 2 3 4 0 hcomp pcomp $1 plus$2 ; a> 255 ifnot a++ out endif halt end\n";

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int scenario = atoi(argv[1]);
  rnd_state = (unsigned)atoi(argv[2]) * 2654435761u + 1;
  Out arc;
  try {
    libzpaq::Compressor co;
    co.setOutput(&arc);
    if (rnd() & 1) co.writeTag();
    int args[9] = {3, 7, 0, 0, 0, 0, 0, 0, 0};
    Out cmd;
    const int nseg = 1 + rnd() % 3;
    if (scenario == 0) {
      co.startBlock(kPass, args, &cmd);
    } else if (scenario == 1) {
      co.setVerify(true);
      co.startBlock(kIdentity, args, &cmd);
    } else if (scenario == 2) {
      co.setVerify((rnd() & 1) != 0);
      co.startBlock(kPlusOne, args, &cmd);
    } else if (scenario == 3) {
      // stored header bytes of a block without components (what method 0 writes), handed over as bytes
      static const char hdr[] = {7, 0, 0, 0, 0, 0, 0, 0, 0};
      co.startBlock(hdr);
    } else if (scenario == 4) {
      co.startBlock(kIdentity, 0, 0);      // no arguments, no command writer
    } else if (scenario == 5) {
      co.startBlock(kPass, args, &cmd);
      co.startSegment("a");
      co.startSegment("b");                // error: a segment inside a segment
    } else {
      co.startBlock("comp 0 0 0 0 1 hcomp end\n", args, &cmd);   // error in the config (component missing)
    }
    printf("cmd=%s\n", hex(cmd.s).c_str());
    Out h, p;
    co.hcomp(&h);
    const bool hasp = co.pcomp(&p);
    printf("hcomp=%s pcomp=%d %s\n", hex(h.s).c_str(), (int)hasp, hex(p.s).c_str());
    for (int s = 0; s < nseg; ++s) {
      In in;
      in.s = data(rnd() % 3000, (int)(rnd() % 3));
      if (scenario == 2) for (char& c : in.s) c = (char)(c - 1);
      const std::string name = s == 0 || (rnd() & 1) ? "file" + std::to_string(s) : "";
      const std::string comment = rnd() & 1 ? std::to_string(in.s.size()) + " jDC\x01" : "";
      co.startSegment(name.empty() ? 0 : name.c_str(), comment.empty() ? 0 : comment.c_str());
      if (s == 0 || (rnd() & 1)) {
        if (scenario == 3 && (rnd() & 1)) {
          const char prog[] = {57, 56, 0};                    // out halt: identity for bytes, ignores EOF
          co.postProcess(prog, 3);
        } else co.postProcess();
      }
      co.setInput(&in);
      const int piece = (rnd() & 1) ? -1 : 1 + (int)(rnd() % 700);
      int calls = 0;
      while (co.compress(piece)) ++calls;
      printf("seg %d n=%zu piece=%d calls=%d\n", s, in.s.size(), piece, calls);
      const unsigned how = rnd() % 3;
      if (how == 0) {
        co.endSegment();
      } else if (how == 1) {
        libzpaq::SHA1 sh;
        sh.write(in.s.data(), (int64_t)in.s.size());
        co.endSegment(sh.result());
      } else {
        int64_t size = -1;
        const bool dosha = (rnd() & 1) != 0;
        const char* r = co.endSegmentChecksum(&size, dosha);
        printf("checksum size=%lld %s\n", (long long)size, r ? hex(std::string(r, 20)).c_str() : "null");
        printf("getSize=%lld getChecksum=%s\n", (long long)co.getSize(), hex(std::string(co.getChecksum(), 20)).c_str());
      }
    }
    co.endBlock();
  } catch (std::exception& e) {
    printf("error %s\n", e.what());
  }
  printf("archive=%s\n", hex(arc.s).c_str());
  return 0;
}
