// PCOMP post-processing on the host (PostProcessor::write libzpaq.cpp:2195-2241 driving
// ZPAQL::run with OUT, 1027-1262).  The decoded stream of a segment starts with 0 (PASS: the rest
// is the data) or 1 len16 program[len] (PROG: the rest is fed byte by byte, then EOF, to that ZPAQL
// program, whose OUT instructions produce the data).  The program travels inside the archive, so
// any archive written by the reference's LZ77 / BWT / E8E9 methods decodes here without this
// library knowing those programs.  The programs compressBlock's own methods generate are, in addition, translated
// to C++ at build time (tools/gen_pcomp_std.cpp, by the translator that serves the device): a block whose program is
// byte for byte one of those runs natively -- the reference gets the same effect from its x86 JIT
// (libzpaq.cpp:3231-3811) -- and anything else is interpreted.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "blocks.hpp"
#include "pcomp_host.h"

namespace zpq {

namespace {

std::atomic<U64> g_pcomp_steps{(U64)1 << 34};   // steps one call of a custom post-processor may take (postproc_set_step_limit)

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
    // The reference puts no limit on a program's run time (a damaged BWT stream makes its inverse loop for good).  The
    // standard programs need a few dozen steps per array element at most (inverse BWT at the end of a segment): 64 per
    // element of H and M, and never less than 2^28, ends a hostile stream in about a second instead of a minute.  That
    // bound is only safe for programs whose cost is known, so it applies to the standard ones (set_standard(true)); a
    // custom program keeps 2^34 steps per call (about a minute; zpq_set_pcomp_step_limit changes it) -- the one deviation from the reference here, which
    // would never give up.
    max_steps_ = g_pcomp_steps.load(std::memory_order_relaxed);
    tight_steps_ = std::max<U64>((U64)1 << 28, 64ull * ((U64)H_.size() + (U64)M_.size()));
  }

  // a program whose cost per input is known (one of the standard LZ77 / BWT / E8E9 inverses): the tight bound applies
  void set_standard(bool yes) { if (yes) max_steps_ = std::min(max_steps_, tight_steps_); }

  void run(U32 input) {
    U32 pc = 0;
    a_ = input;
    for (U64 steps = 0;; ++steps) {
      if (steps > max_steps_ || pc >= len_) bad();
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
  U64 max_steps_ = 0, tight_steps_ = 0;
  U32 a_ = 0, b_ = 0, c_ = 0, d_ = 0;
  bool f_ = false;
};

}  // namespace

void postproc_set_step_limit(U64 steps) { g_pcomp_steps.store(steps ? steps : (U64)1 << 34, std::memory_order_relaxed); }

// PostProcessor of ONE block (the reference initialises it once per block, libzpaq.cpp:2320-2330: only the first
// segment carries the PP header; later segments continue in the same mode and, for PROG, with the same machine).
struct PostProcessor::Impl {
  int ph, pm;
  int state = 0;                 // 0 initial, 1 PASS, 2..4 loading PROG, 5 PROG loaded (PostProcessor::write)
  size_t want = 0;
  std::vector<U8> prog;          // PCOMP code as carried by the first segment
  std::vector<U8> sink;
  std::unique_ptr<PcompVm> vm;   // the interpreter's machine, made when first needed
  // The block's first segment through a translated program: the interpreter's machine has not seen it.  Blocks of several
  // segments share one machine, so the segment is kept and replayed into the interpreter if a second one arrives.
  size_t segments = 0;
  bool native_first = false;
  std::vector<U8> replay;
  PcompVm& machine();
};

namespace {
const PcompStd* standard_program(const std::vector<U8>& prog, int ph, int pm) {
  for (const PcompStd* e = kPcompStd; e->code; ++e)
    if (e->len == prog.size() && e->ph == ph && e->pm == pm && memcmp(e->code, prog.data(), prog.size()) == 0) return e;
  return nullptr;
}
const PcompStd* translated(const std::vector<U8>& prog, int ph, int pm) {
  if (const char* m = getenv("ZPAQ_AMD_PCOMP")) { if (!strcmp(m, "interpret")) return nullptr; }     // tests: the interpreter instead of the translated program
  return standard_program(prog, ph, pm);
}

// one segment through a translated program on a fresh machine; false = it reported an error (the caller interprets)
bool run_translated(const PcompStd& e, const U8* p, size_t n, std::vector<U8>& out) {
  if (e.ph > 28 || e.pm > 30) return false;
  std::vector<U32> H((size_t)1 << e.ph, 0);
  std::vector<U8> M((size_t)1 << e.pm, 0);
  PcompHostState st;
  st.H = H.data();
  st.M = M.data();
  out.reserve(out.size() + n * 3);
  return e.run(st, p, n, true, out) == 0;
}
}  // namespace

PcompVm& PostProcessor::Impl::machine() {
  if (!vm) {
    vm.reset(new PcompVm(prog.data(), prog.size(), ph, pm, sink));
    vm->set_standard(standard_program(prog, ph, pm) != nullptr);
  }
  return *vm;
}

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
        if (m.prog.size() == m.want) m.state = 5;
        break;
    }
  }
  if (m.state != 1 && m.state != 5) fail(ZPQ_E_CORRUPT, "Unexpected EOS");      // the segment ended inside the PP header
  if (m.state == 1) {
    data.insert(data.end(), p + i, p + n);
    return;
  }
  const size_t seg = m.segments++;
  if (seg == 0) {
    if (const PcompStd* e = translated(m.prog, m.ph, m.pm)) {
      const size_t before = data.size();
      if (run_translated(*e, p + i, n - i, data)) {
        m.native_first = true;
        m.replay.assign(p + i, p + n);
        return;
      }
      data.resize(before);                     // the translated program gave up (step budget): interpret from the start
    }
  }
  PcompVm& vm = m.machine();
  if (m.native_first) {                        // a second segment: bring the interpreter's machine to where the first one left it
    m.sink.clear();
    for (const U8 c : m.replay) vm.run(c);
    vm.run(0xFFFFFFFFu);
    m.native_first = false;
    std::vector<U8>().swap(m.replay);
  }
  m.sink.clear();
  for (; i < n; ++i) vm.run(p[i]);
  vm.run(0xFFFFFFFFu);                         // EOS: ZPAQL::run(-1) (libzpaq.cpp:2236-2237)
  data.insert(data.end(), m.sink.begin(), m.sink.end());
  m.sink.clear();
}

bool PostProcessor::loaded() const { return impl_->state == 1 || impl_->state == 5; }
const std::vector<U8>& PostProcessor::program() const { return impl_->prog; }

bool pcomp_is_translated(const U8* code, size_t len, int ph, int pm) {
  return translated(std::vector<U8>(code, code + len), ph, pm) != nullptr;
}

void post_process(const std::vector<U8>& header, const std::vector<U8>& decoded, std::vector<U8>& data) {
  data.clear();
  PostProcessor pp(header[4], header[5]);
  pp.segment(decoded.data(), decoded.size(), data);
}

}  // namespace zpq
