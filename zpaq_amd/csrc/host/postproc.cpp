// PCOMP post-processing on the host (PostProcessor::write libzpaq.cpp:2195-2241 driving
// ZPAQL::run with OUT, 1027-1262).  The decoded stream of a segment starts with 0 (PASS: the rest
// is the data) or 1 len16 program[len] (PROG: the rest is fed byte by byte, then EOF, to that ZPAQL
// program, whose OUT instructions produce the data).  The program travels inside the archive, so
// any archive written by the reference's LZ77 / BWT / E8E9 methods decodes here without this
// library knowing those programs.
#include <cstring>

#include "blocks.hpp"

namespace zpq {

namespace {

class PcompVm {
 public:
  PcompVm(const U8* prog, size_t len, int ph, int pm, std::vector<U8>& out)
      : prog_(prog), len_((U32)len), out_(out) {
    if (ph > 28 || pm > 30) fail(ZPQ_E_NOMEM, "Out of memory");     // reference: > 32 is an error
    H_.assign((size_t)1 << ph, 0);
    M_.assign((size_t)1 << pm, 0);
    hmask_ = (U32)H_.size() - 1;
    mmask_ = (U32)M_.size() - 1;
    memset(R_, 0, sizeof(R_));
  }

  void run(U32 input) {
    U32 pc = 0;
    a_ = input;
    for (U64 steps = 0;; ++steps) {
      if (steps > (1ull << 34) || pc >= len_) bad();
      const int op = prog_[pc++];
      const int g = op >> 3, k = op & 7;
      if (op < 64) {
        if (g == 7) {
          if (op == 56) return;
          else if (op == 57) out_.push_back((U8)a_);
          else if (op == 59) a_ = (a_ + M_[b_ & mmask_] + 512u) * 773u;
          else if (op == 60) { U32& d = H_[d_ & hmask_]; d = (d + a_ + 512u) * 773u; }
          else if (op == 63) pc += 1 + (int)(int8_t)at(pc);
          else bad();
        } else if (k == 7) {
          const U32 n = at(pc);
          if (g < 4) { set(g, R_[n]); ++pc; }
          else if (g == 4) { if (f_) pc += 1 + (int)(int8_t)n; else ++pc; }
          else if (g == 5) { if (!f_) pc += 1 + (int)(int8_t)n; else ++pc; }
          else { R_[n] = a_; ++pc; }
        } else {
          if (op == 0 || k > 4) bad();
          const U32 x = get(g, pc);
          if (k == 0) {
            const U32 a = a_;
            if (g == 4 || g == 5) { set(g, a & 255u); a_ = (a & 0xFFFFFF00u) | x; }
            else { set(g, a); a_ = x; }
          } else if (k == 1) set(g, x + 1);
          else if (k == 2) set(g, x - 1);
          else if (k == 3) set(g, ~x);
          else set(g, 0);
        }
      } else if (op < 120) {
        const U32 v = get(k, pc);
        set(g - 8, v);
      } else if (op < 128) {
        bad();
      } else if (op < 240) {
        const U32 v = get(k, pc);
        switch (g - 16) {
          case 0: a_ += v; break;
          case 1: a_ -= v; break;
          case 2: a_ *= v; break;
          case 3: a_ = v ? a_ / v : 0; break;
          case 4: a_ = v ? a_ % v : 0; break;
          case 5: a_ &= v; break;
          case 6: a_ &= ~v; break;
          case 7: a_ |= v; break;
          case 8: a_ ^= v; break;
          case 9: a_ <<= (v & 31); break;
          case 10: a_ >>= (v & 31); break;
          case 11: f_ = a_ == v; break;
          case 12: f_ = a_ < v; break;
          default: f_ = a_ > v; break;
        }
      } else if (op == 255) {
        const U32 t = at(pc) + 256u * at(pc + 1);
        if (t >= len_) bad();
        pc = t;
      } else bad();
    }
  }

 private:
  [[noreturn]] static void bad() { fail(ZPQ_E_VM, "ZPAQL execution error"); }
  U32 at(U32 pc) const { if (pc >= len_) bad(); return prog_[pc]; }
  U32 get(int k, U32& pc) {
    switch (k) {
      case 0: return a_;
      case 1: return b_;
      case 2: return c_;
      case 3: return d_;
      case 4: return M_[b_ & mmask_];
      case 5: return M_[c_ & mmask_];
      case 6: return H_[d_ & hmask_];
      default: return at(pc++);
    }
  }
  void set(int g, U32 v) {
    switch (g) {
      case 0: a_ = v; break;
      case 1: b_ = v; break;
      case 2: c_ = v; break;
      case 3: d_ = v; break;
      case 4: M_[b_ & mmask_] = (U8)v; break;
      case 5: M_[c_ & mmask_] = (U8)v; break;
      default: H_[d_ & hmask_] = v; break;
    }
  }
  const U8* prog_;
  U32 len_;
  std::vector<U8>& out_;
  std::vector<U32> H_;
  std::vector<U8> M_;
  U32 R_[256];
  U32 hmask_ = 0, mmask_ = 0;
  U32 a_ = 0, b_ = 0, c_ = 0, d_ = 0;
  bool f_ = false;
};

}  // namespace

void post_process(const std::vector<U8>& header, const std::vector<U8>& decoded, std::vector<U8>& data) {
  data.clear();
  if (decoded.empty()) fail(ZPQ_E_CORRUPT, "Unexpected EOS");
  if (decoded[0] == 0) {                       // PASS
    data.assign(decoded.begin() + 1, decoded.end());
    return;
  }
  if (decoded[0] != 1) fail(ZPQ_E_CORRUPT, "unknown post processing type");
  if (decoded.size() < 3) fail(ZPQ_E_CORRUPT, "Unexpected EOS");
  const size_t len = decoded[1] + 256u * decoded[2];
  if (len < 1) fail(ZPQ_E_CORRUPT, "Empty PCOMP");
  if (decoded.size() < 3 + len) fail(ZPQ_E_CORRUPT, "Unexpected EOS");
  PcompVm vm(decoded.data() + 3, len, header[4], header[5], data);
  for (size_t i = 3 + len; i < decoded.size(); ++i) vm.run(decoded[i]);
  vm.run(0xFFFFFFFFu);                         // EOS: ZPAQL::run(-1) (libzpaq.cpp:2236-2237)
}

}  // namespace zpq
