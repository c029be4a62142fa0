// PCOMP post-processing on the host (PostProcessor::write libzpaq.cpp:2195-2241 driving
// ZPAQL::run with OUT, 1027-1262).  The decoded stream of a segment starts with 0 (PASS: the rest
// is the data) or 1 len16 program[len] (PROG: the rest is fed byte by byte, then EOF, to that ZPAQL
// program, whose OUT instructions produce the data).  The program travels inside the archive, so
// any archive written by the reference's LZ77 / BWT / E8E9 methods decodes here without this
// library knowing those programs.
#include <cstring>
#include <memory>

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

// PostProcessor of ONE block (the reference initialises it once per block, libzpaq.cpp:2320-2330: only the first
// segment carries the PP header; later segments continue in the same mode and, for PROG, with the same machine).
struct PostProcessor::Impl {
  int ph, pm;
  int state = 0;                 // 0 initial, 1 PASS, 2..4 loading PROG, 5 PROG loaded (PostProcessor::write)
  size_t want = 0;
  std::vector<U8> prog;          // PCOMP code as carried by the first segment
  std::vector<U8> sink;
  std::unique_ptr<PcompVm> vm;
};

PostProcessor::PostProcessor(int ph, int pm) : impl_(new Impl) { impl_->ph = ph; impl_->pm = pm; }
PostProcessor::~PostProcessor() { delete impl_; }

void PostProcessor::segment(const U8* p, size_t n, std::vector<U8>& data) {
  Impl& m = *impl_;
  size_t i = 0;
  while (i < n && m.state != 1 && m.state != 5) {
    const U8 c = p[i++];
    switch (m.state) {
      case 0:
        if (c > 1) fail(ZPQ_E_CORRUPT, "unknown post processing type");
        m.state = c + 1;
        break;
      case 2: m.want = c; m.state = 3; break;
      case 3:
        m.want += 256u * c;
        if (m.want < 1) fail(ZPQ_E_CORRUPT, "Empty PCOMP");
        m.prog.clear();
        m.state = 4;
        break;
      case 4:
        m.prog.push_back(c);
        if (m.prog.size() == m.want) {
          m.vm.reset(new PcompVm(m.prog.data(), m.prog.size(), m.ph, m.pm, m.sink));
          m.state = 5;
        }
        break;
    }
  }
  if (m.state != 1 && m.state != 5) fail(ZPQ_E_CORRUPT, "Unexpected EOS");      // the segment ended inside the PP header
  if (m.state == 1) {
    data.insert(data.end(), p + i, p + n);
    return;
  }
  m.sink.clear();
  for (; i < n; ++i) m.vm->run(p[i]);
  m.vm->run(0xFFFFFFFFu);                      // EOS: ZPAQL::run(-1) (libzpaq.cpp:2236-2237)
  data.insert(data.end(), m.sink.begin(), m.sink.end());
  m.sink.clear();
}

bool PostProcessor::loaded() const { return impl_->state == 1 || impl_->state == 5; }
const std::vector<U8>& PostProcessor::program() const { return impl_->prog; }

void post_process(const std::vector<U8>& header, const std::vector<U8>& decoded, std::vector<U8>& data) {
  data.clear();
  PostProcessor pp(header[4], header[5]);
  pp.segment(decoded.data(), decoded.size(), data);
}

}  // namespace zpq
