// Method layer (host, per block): what libzpaq::compressBlock does before it
// touches the coder.
//
//   expand_method : "0".."9" level strings (+ optional ",R,t" hints) -> the
//                   explicit "x.." / "0.." method (libzpaq.cpp:7551-7691),
//                   including the level-5 scan of the data for byte periods.
//   make_config   : "x.." method -> ZPAQL source (libzpaq.cpp:6887-7535).  Only
//                   the context-model half (COMP list + HCOMP program) is
//                   restated; methods that need the LZ77/BWT/E8E9
//                   pre-processors (their PCOMP programs + encoders are
//                   SURVEY §8(f) row 1, not built yet) fail with
//                   ZPQ_E_UNSUPPORTED rather than produce a different archive.
//
// The generated text only has to ASSEMBLE to the same bytes as the
// reference's; tests/test_host.py checks that against the compiled reference for every
// level x block size x hint combination in scope.
#include <cctype>
#include <cstring>

#include "common.hpp"

namespace zpq {

namespace {

std::string num(long v) { return std::to_string(v); }

// floor(log2(x))+1, 0 for 0  (libzpaq.cpp:6566)
int bitlen(unsigned x) { int r = 0; while (x) { ++r; x >>= 1; } return r; }
int popcount(unsigned x) { int r = 0; for (; x; x >>= 1) r += x & 1; return r; }

// Splits "c0,2,0,255" style commands: letter followed by a comma/dot separated
// list of decimal numbers.
struct Command { char letter; std::vector<int> v; };

std::vector<Command> parse_commands(const char* s) {
  std::vector<Command> out;
  while (*s) {
    Command c;
    c.letter = *s++;
    if (isdigit((unsigned char)*s)) {
      c.v.push_back(0);
      while (isdigit((unsigned char)*s) || *s == ',' || *s == '.') {
        if (isdigit((unsigned char)*s)) c.v.back() = c.v.back() * 10 + (*s - '0');
        else c.v.push_back(0);
        ++s;
      }
    }
    out.push_back(c);
  }
  return out;
}

// Accumulates the COMP list and the HCOMP program text.
class ModelWriter {
 public:
  ModelWriter(int membits) : membits_(membits), ncomp_(0), sb_(5) {
    hcomp_ = "hcomp\nc-- *c=a a+= 255 d=a *d=c\n";
  }
  int ncomp() const { return ncomp_; }
  const std::string& comp() const { return comp_; }
  const std::string& hcomp() const { return hcomp_; }
  void raw_hcomp(const std::string& s) { hcomp_ += s; }

  // c: CM/ICM with order/mask/periodic/distance contexts (libzpaq.cpp:7375-7437)
  void context_model(std::vector<int> v) {
    while (v.size() < 2) v.push_back(0);   // v[0]=N1, v[1]=N2, v[2..]=N3..
    const int n1 = v[0], n2 = v[1];
    sb_ = 11;
    sb_ += (n2 < 256) ? bitlen((unsigned)n2) : 6;
    for (size_t i = 2; i < v.size(); ++i)
      if (v[i] < 512) sb_ += popcount((unsigned)v[i]) * 3 / 4;
    if (sb_ > membits_) sb_ = membits_;
    comp_ += num(ncomp_) + " ";
    if (n1 % 1000 == 0) comp_ += "icm " + num(sb_ - 6 - n1 / 1000) + "\n";
    else comp_ += "cm " + num(sb_ - 2 - n1 / 1000) + " " + num(n1 % 1000 - 1) + "\n";

    hcomp_ += "d= " + num(ncomp_) + " *d=0\n";
    if (n2 > 1 && n2 <= 255) {                       // position mod n2
      if (bitlen((unsigned)n2) != bitlen((unsigned)n2 - 1)) hcomp_ += "a=c a&= " + num(n2 - 1) + " hashd\n";
      else hcomp_ += "a=c a%= " + num(n2) + " hashd\n";
    } else if (n2 >= 1000 && n2 <= 1255) {           // distance to last occurrence of byte n2-1000
      hcomp_ += "a= 255 a+= " + num(n2 - 1000) + " d=a a=*d a-=c a> 255 if a= 255 endif d= " +
                num(ncomp_) + " hashd\n";
    }
    for (size_t i = 2; i < v.size(); ++i) {
      const int x = v[i];
      if (i == 2) hcomp_ += "b=c ";
      if (x == 255) hcomp_ += "a=*b hashd\n";
      else if (x > 0 && x < 255) hcomp_ += "a=*b a&= " + num(x) + " hashd\n";
      else if (x >= 256 && x < 512) {                 // LZ77 parse-state context
        hcomp_ += "a=r 1 a> 1 if\n  a=r 2 a< 64 if\n    a=*b ";
        if (x < 511) hcomp_ += "a&= " + num(x - 256);
        hcomp_ += " hashd\n  else\n    a>>= 6 hashd a=r 1 hashd\n  endif\nelse\n  a= 255 hashd a=r 2 hashd\nendif\n";
      } else if (x >= 1256) {
        hcomp_ += "a= " + num(((x - 1000) >> 8) & 255) + " a<<= 8 a+= " + num((x - 1000) & 255) + " a+=b b=a\n";
      } else if (x > 1000) {
        hcomp_ += "a= " + num(x - 1000) + " a+=b b=a\n";
      }
      if (x < 512 && i + 1 < v.size()) hcomp_ += "b++ ";
    }
    ++ncomp_;
  }

  // m / t / s: MIX, MIX2, SSE over everything so far (libzpaq.cpp:7442-7470)
  void mixer(char kind, std::vector<int> v) {
    if (ncomp_ <= (kind == 't' ? 1 : 0)) return;
    if (v.size() < 1) v.push_back(8);
    if (v.size() < 2) v.push_back(24 + 8 * (kind == 's'));
    if (kind == 's' && v.size() < 3) v.push_back(255);
    int bits = v[0];
    comp_ += num(ncomp_);
    sb_ = 5 + bits * 3 / 4;
    if (kind == 'm')
      comp_ += " mix " + num(bits) + " 0 " + num(ncomp_) + " " + num(v[1]) + " 255\n";
    else if (kind == 't')
      comp_ += " mix2 " + num(bits) + " " + num(ncomp_ - 1) + " " + num(ncomp_ - 2) + " " + num(v[1]) + " 255\n";
    else
      comp_ += " sse " + num(bits) + " " + num(ncomp_ - 1) + " " + num(v[1]) + " " + num(v[2]) + "\n";
    if (bits > 8) {                                   // order-1/2 selector context
      hcomp_ += "d= " + num(ncomp_) + " *d=0 b=c a=0\n";
      for (; bits >= 16; bits -= 8) {
        hcomp_ += "a<<= 8 a+=*b";
        if (bits > 16) hcomp_ += " b++";
        hcomp_ += "\n";
      }
      if (bits > 8) hcomp_ += "a<<= 8 a+=*b a>>= " + num(16 - bits) + "\n";
      hcomp_ += "a<<= 8 *d=a\n";
    }
    ++ncomp_;
  }

  // i: ISSE chain extending the previous component's context (7473-7490)
  void isse_chain(const std::vector<int>& v) {
    if (ncomp_ <= 0) return;
    hcomp_ += "d= " + num(ncomp_ - 1) + " b=c a=*d d++\n";
    for (size_t i = 0; i < v.size() && ncomp_ < 254; ++i) {
      for (int j = 0; j < v[i] % 10; ++j) {
        hcomp_ += "hash ";
        if (i + 1 < v.size() || j < v[i] % 10 - 1) hcomp_ += "b++ ";
        sb_ += 6;
      }
      hcomp_ += "*d=a";
      if (i + 1 < v.size()) hcomp_ += " d++";
      hcomp_ += "\n";
      if (sb_ > membits_) sb_ = membits_;
      comp_ += num(ncomp_) + " isse " + num(sb_ - 6 - v[i] / 10) + " " + num(ncomp_ - 1) + "\n";
      ++ncomp_;
    }
  }

  // a: MATCH (7493-7503)
  void match_model(std::vector<int> v) {
    if (v.size() < 1) v.push_back(24);
    while (v.size() < 3) v.push_back(0);
    comp_ += num(ncomp_) + " match " + num(membits_ - v[2] - 2) + " " + num(membits_ - v[1]) + "\n";
    hcomp_ += "d= " + num(ncomp_) + " a=*d a*= " + num(v[0]) + " a+=*c a++ *d=a\n";
    sb_ = 5 + (membits_ - v[1]) * 3 / 4;
    ++ncomp_;
  }

  // w: word-model ICM-ISSE chain (7509-7535)
  void word_model(std::vector<int> v) {
    static const int dflt[6] = {1, 65, 26, 223, 20, 0};
    for (size_t i = v.size(); i < 6; ++i) v.push_back(dflt[i]);
    const int len = v[0], lo = v[1], span = v[2], mask = v[3], mul = v[4], shrink = v[5];
    comp_ += num(ncomp_) + " icm " + num(membits_ - 6 - shrink) + "\n";
    for (int i = 1; i < len; ++i)
      comp_ += num(ncomp_ + i) + " isse " + num(membits_ - 6 - shrink) + " " + num(ncomp_ + i - 1) + "\n";
    hcomp_ += "a=*c a&= " + num(mask) + " a-= " + num(lo) + " a&= 255 a< " + num(span) + " if\n";
    for (int i = 0; i < len; ++i) {
      hcomp_ += (i == 0) ? "  d= " + num(ncomp_) : std::string("  d++");
      hcomp_ += " a=*d a*= " + num(mul) + " a+=*c a++ *d=a\n";
    }
    hcomp_ += "else\n";
    for (int i = len - 1; i > 0; --i) hcomp_ += "  d= " + num(ncomp_ + i - 1) + " a=*d d++ *d=a\n";
    hcomp_ += "  d= " + num(ncomp_) + " *d=0\nendif\n";
    ncomp_ += len - 1;
    sb_ = membits_ - shrink;
    ++ncomp_;
  }

 private:
  int membits_, ncomp_, sb_;
  std::string comp_, hcomp_;
};

}  // namespace

std::string make_config(const std::string& xmethod, int args[9]) {
  if (xmethod.empty()) fail(ZPQ_E_ARG, "empty method");
  const char type = xmethod[0];
  if (type != 'x' && type != 's' && type != '0' && type != 'i') fail(ZPQ_E_ARG, "method must start with x, s, i or 0");
  for (int i = 0; i < 9; ++i) args[i] = 0;
  // "{x|s|i|0}N1,N2,...,N9" prefix -> args (libzpaq.cpp:6903-6910)
  const char* p = xmethod.c_str() + 1;
  for (int i = 0; i < 9 && (isdigit((unsigned char)*p) || *p == ',' || *p == '.');) {
    if (isdigit((unsigned char)*p)) args[i] = args[i] * 10 + (*p - '0');
    else if (++i < 9) args[i] = 0;
    ++p;
  }
  if (type == '0') return "comp 0 0 0 0 0 hcomp end\n";

  const int level = args[1] & 3;
  const bool doe8 = args[1] >= 4 && args[1] <= 7;
  if (level != 0 || doe8)
    fail(ZPQ_E_UNSUPPORTED,
         "method needs the LZ77/BWT/E8E9 pre/post-processors, which are outside this build's hot-path scope");
  if (type == 'i') fail(ZPQ_E_UNSUPPORTED, "index-block methods are outside this build's scope");

  ModelWriter w(args[0] + 20);
  for (const Command& c : parse_commands(p)) {
    if (w.ncomp() >= 254) break;
    switch (c.letter) {
      case 'c': w.context_model(c.v); break;
      case 'm': case 't': case 's': w.mixer(c.letter, c.v); break;
      case 'i': w.isse_chain(c.v); break;
      case 'a': w.match_model(c.v); break;
      case 'w': w.word_model(c.v); break;
      default: break;   // unknown letters are skipped, like the reference's loop
    }
  }
  return "comp 9 16 0 0 " + num(w.ncomp()) + "\n" + w.comp() + w.hcomp() + "halt\nend\n";
}

std::string expand_method(const std::string& method_in, const U8* data, U32 n) {
  if (method_in.empty()) fail(ZPQ_E_ARG, "empty method");
  if (!isdigit((unsigned char)method_in[0])) return method_in;
  const int arg0 = std::max(bitlen(n + 4095) - 20, 0);
  // "LB,R,t": R = redundancy 0..255, t = 0 binary, 1 text, 2 exe, 3 both (7556-7565)
  unsigned type = 0;
  {
    int commas = 0, arg[4] = {0, 0, 0, 0};
    for (size_t i = 1; i < method_in.size() && commas < 4; ++i) {
      const char ch = method_in[i];
      if (ch == ',' || ch == '.') ++commas;
      else if (isdigit((unsigned char)ch)) arg[commas] = arg[commas] * 10 + (ch - '0');
    }
    type = commas == 0 ? 512u : (unsigned)(arg[1] * 4 + arg[2]);
  }
  const int level = method_in[0] - '0';
  const int doe8 = (type & 2) * 2;
  std::string m = "x" + num(arg0);
  const std::string htsz = "," + num(19 + arg0 + (arg0 <= 6));
  const std::string sasz = "," + num(21 + arg0);
  if (level == 0) return "0" + num(arg0) + ",0";
  if (level == 1) {
    if (type < 40) return m + ",0";
    m += "," + num(1 + doe8) + ",";
    if (type < 80) m += "4,0,1,15";
    else if (type < 128) m += "4,0,2,16";
    else if (type < 256) m += "4,0,2" + htsz;
    else if (type < 960) m += "5,0,3" + htsz;
    else m += "6,0,3" + htsz;
    return m;
  }
  if (level == 2) {
    if (type < 32) return m + ",0";
    m += "," + num(1 + doe8) + ",";
    if (type < 64) m += "4,0,3" + htsz;
    else m += "4,0,7" + sasz + ",1";
    return m;
  }
  if (level == 3) {
    if (type < 20) m += ",0";
    else if (type < 48) m += "," + num(1 + doe8) + ",4,0,3" + htsz;
    else if (type >= 640 || (type & 1)) m += "," + num(3 + doe8) + "ci1";
    else m += "," + num(2 + doe8) + ",12,0,7" + sasz + ",1c0,0,511i2";
    return m;
  }
  if (level == 4) {
    if (type < 12) m += ",0";
    else if (type < 24) m += "," + num(1 + doe8) + ",4,0,3" + htsz;
    else if (type < 48) m += "," + num(2 + doe8) + ",5,0,7" + sasz + "1c0,0,511";
    else if (type < 900) {
      m += "," + num(doe8) + "ci1,1,1,1,2a";
      if (type & 1) m += "w";
      m += "m";
    } else m += "," + num(3 + doe8) + "ci1";
    return m;
  }
  // levels 5..9: the big CM, with periodic models where the data shows a period
  m += "," + num(doe8);
  m += (type & 1) ? "w2c0,1010,255i1" : "w1i1";
  m += "c256ci1,1,1,1,1,1,2a";
  {
    const int NR = 1 << 12;
    std::vector<int> gap(NR, 0);
    int last[256];
    for (int i = 0; i < 256; ++i) last[i] = 0;
    for (U32 i = 0; i < n; ++i) {
      const int k = (int)i - last[data[i]];
      if (k > 0 && k < NR) ++gap[k];
      last[data[i]] = (int)i;
    }
    int n1 = (int)n - gap[1] - gap[2] - gap[3];
    for (int round = 0; round < 2; ++round) {
      int period = 0;
      double score = 0;
      int t = 0;
      for (int j = 5; j < NR && t < n1; ++j) {
        const double s = gap[j] / (256.0 + n1 - t);
        if (s > score) { score = s; period = j; }
        t += gap[j];
      }
      if (period > 4 && score > 0.1) {
        m += "c0,0," + num(999 + period) + ",255i1";
        if (period <= 255) m += "c0," + num(period) + "i1";
        n1 -= gap[period];
        gap[period] = 0;
      } else break;
    }
  }
  m += "c0,2,0,255i1c0,3,0,0,255i1c0,4,0,0,0,255i1mm16ts19t0";
  return m;
}

}  // namespace zpq
