// Host environment of the PCOMP programs translated at build time (tools/gen_pcomp_std.cpp -> pcomp_std_gen.cpp): the
// translator (host/codegen.cpp) writes for the device; these few definitions let the same text compile for the host.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#ifndef __device__
#define __device__
#endif
#ifndef __forceinline__
#define __forceinline__ inline
#endif

namespace zpq {

inline unsigned vm_flag(bool c) { return c ? 1u : 0u; }

// registers and arrays of one PostProcessor's ZPAQL machine (H: 2^ph words, M: 2^pm bytes, both zero at the start)
struct PcompHostState {
  unsigned b = 0, c = 0, d = 0, f = 0;
  uint8_t* M = nullptr;
  uint32_t* H = nullptr;
  uint32_t R[256] = {};
};

struct PcompHostOut {
  std::vector<uint8_t>* v;
  void operator()(unsigned a) { v->push_back((uint8_t)a); }
};

// feeds n bytes (and the end-of-segment call when eos) to the program; 0, or the translated program's error status
typedef int (*PcompHostRun)(PcompHostState& s, const uint8_t* in, size_t n, bool eos, std::vector<uint8_t>& out);

template <class Post>
int pcomp_host_run(PcompHostState& s, const uint8_t* in, size_t n, bool eos, std::vector<uint8_t>& out) {
  PcompHostOut o{&out};
  for (size_t k = 0; k < n; ++k) {
    const int st = Post::pcomp(in[k], s.b, s.c, s.d, s.f, s.M, s.H, s.R, o);
    if (st) return st;
  }
  return eos ? Post::pcomp(0xFFFFFFFFu, s.b, s.c, s.d, s.f, s.M, s.H, s.R, o) : 0;
}

// one translated program: its bytes as they travel in the archive (without the 2 length bytes), ph, pm, entry point
struct PcompStd {
  const unsigned char* code;
  unsigned len;
  int ph, pm;
  PcompHostRun run;
};
extern const PcompStd kPcompStd[];      // terminated by an entry with code == nullptr

}  // namespace zpq
