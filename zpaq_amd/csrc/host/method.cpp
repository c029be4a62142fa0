// Method layer (host, per block): what libzpaq::compressBlock does before it
// touches the coder.
//
//   expand_method : "0".."9" level strings (+ optional ",R,t" hints) -> the
//                   explicit "x.." / "0.." method (libzpaq.cpp:7551-7691),
//                   including the level-5 scan of the data for byte periods.
//   make_config   : "x.." method -> ZPAQL source (libzpaq.cpp:6887-7535): the
//                   context model (COMP list + HCOMP program) and, for methods
//                   with LZ77 / BWT / E8E9 pre-processing, the PCOMP program
//                   that inverts it.  PCOMP travels inside the archive, so its
//                   assembled bytes are format data: the token sequences below
//                   are the reference's (comments and layout are not); the
//                   encoders themselves live in host/preproc.cpp.
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

// ---- PCOMP programs: the inverse transforms the decompresser runs (libzpaq.cpp:6921-7324) ----

// inverse E8E9 over M[b..d) with output; enters with the loop's `do` already open
const char* const kE8Loop =
    "a=b a==d ifnot a+= 4 a<d if a=*b a&= 254 a== 232 if c=b b++ b++ b++ b++ a=*b a++ a&= 254 a== 0 if "
    "b-- a=*b b-- a<<= 8 a+=*b b-- a<<= 8 a+=*b a-=b a++ *b=a a>>= 8 b++ *b=a a>>= 8 b++ *b=a b++ "
    "endif b=c endif endif a=*b out b++ forever endif\n";

// bit-packed LZ77 (codes of host/preproc.cpp level 1); r1 state, r2 length, r3 offset bits, r4 write pointer,
// r5 low offset bits, c / d the bit buffer and its fill
std::string pcomp_lz77_bits(int arg0, bool doe8) {
  const int rb = arg0 > 4 ? arg0 - 4 : 0;
  std::string s = "pcomp lazy2 3 ;\na> 255 if\n";
  if (doe8) s += std::string("b=0 d=r 4 do ") + kE8Loop;
  s += "a=0 b=0 c=0 d=0 r=a 1 r=a 2 r=a 3 r=a 4 halt endif\n"
       "a<<=d a+=c c=a a= 8 a+=d d=a\n"
       "a=r 1 a== 0 if a= 1 r=a 2 a=c a&= 3 a> 0 if\n"
       "a-- a<<= 3 r=a 3 a=c a>>= 2 c=a b=r 3 a&= 7 a+=b r=a 3 a=c a>>= 3 c=a a=d a-= 5 d=a a= 1 r=a 1\n"
       "else a=c a>>= 2 c=a d-- d-- a= 3 r=a 1 endif endif\n"
       "do a=r 1 a== 1 if a=d a> 2 if a=c a&= 1 a== 1 if\n"
       "a=c a>>= 1 c=a b=r 2 a=c a&= 1 a+=b a+=b r=a 2 a=c a>>= 1 c=a d-- d--\n"
       "else a=c a>>= 1 c=a a=r 2 a<<= 2 b=a a=c a&= 3 a+=b r=a 2 a=c a>>= 2 c=a d-- d-- d--\n";
  s += rb ? "a= 5 r=a 1\n" : "a= 2 r=a 1\n";
  s += "endif forever endif endif\n";
  if (rb)
    s += "a=r 1 a== 5 if a=d a> " + num(rb - 1) + " if a=c a&= " + num((1 << rb) - 1) + " r=a 5 a=c a>>= " + num(rb) +
         " c=a a=d a-= " + num(rb) + " d=a a= 2 r=a 1 endif endif\n";
  s += "a=r 1 a== 2 if a=r 3 a>d ifnot a=c r=a 6 a=d r=a 7 b=r 3 a= 1 a<<=b d=a a-- a&=c a+=d\n";
  if (rb) s += "a<<= " + num(rb) + " d=r 5 a+=d a-= " + num((1 << rb) - 1) + "\n";
  s += "d=a b=r 4 a=b a-=d c=a d=r 2 do a=d a> 0 if d-- a=*c *b=a c++ b++\n";
  if (!doe8) s += "out\n";
  s += "forever endif a=b r=a 4 a=r 6 b=r 3 a>>=b c=a a=r 7 a-=b d=a a=0 r=a 1 endif endif\n"
       "do a=r 1 a== 3 if a=d a> 1 if a=c a&= 1 a== 1 if\n"
       "a=c a>>= 1 c=a b=r 2 a&= 1 a+=b a+=b r=a 2 a=c a>>= 1 c=a d-- d--\n"
       "else a=c a>>= 1 c=a d-- a= 4 r=a 1 endif forever endif endif\n"
       "a=r 1 a== 4 if a=d a> 7 if b=r 4 a=c *b=a\n";
  if (!doe8) s += "out\n";
  s += "b++ a=b r=a 4 a=c a>>= 8 c=a a=d a-= 8 d=a a=r 2 a-- r=a 2 a== 0 if a=0 r=a 1 endif endif endif\n"
       "halt\nend\n";
  return s;
}

// byte-aligned LZ77 (level 2): d state, r1 length, r2 offset so far, b write pointer
std::string pcomp_lz77_bytes(bool doe8) {
  std::string s = "pcomp lzpre c ;\na> 255 if\n";
  if (doe8) s += std::string("d=b b=0 do ") + kE8Loop;
  s += "b=0 c=0 d=0 a=0 r=a 1 r=a 2 halt endif\n"
       "c=a a=d a== 0 if a=c a>>= 6 a++ d=a a== 1 if a+=c r=a 1 a=0 r=a 2\n"
       "else d++ a=c a&= 63 a+= $3 r=a 1 a=0 r=a 2 endif\n"
       "else a== 1 if a=c *b=a b++\n";
  if (!doe8) s += "out\n";
  s += "a=r 1 a-- a== 0 if d=0 endif r=a 1\n"
       "else a> 2 if a=r 2 a<<= 8 a|=c r=a 2 d--\n"
       "else a=r 2 a<<= 8 a|=c c=a a=b a-=c a-- c=a d=r 1 do a=*c *b=a c++ b++\n";
  if (!doe8) s += "out\n";
  s += "d-- a=d a> 0 while endif endif endif\nhalt\nend\n";
  return s;
}

// inverse BWT: counts and the linked list in H, the text in M
std::string pcomp_bwt(int arg0, bool doe8) {
  std::string s =
      "pcomp bwtrle c ;\na> 255 ifnot *b=a b++ elsel\n"
      "b-- a=*b b-- a<<= 8 a+=*b b-- a<<= 8 a+=*b b-- a<<= 8 a+=*b c=a r=a 1 a=b r=a 2\n"
      "do a=b a> 0 if b-- a=*b a++ a&= 255 d=a d! *d++ forever endif\n"
      "d=0 d! *d= 1 a=0 do a+=*d *d=a d-- d<>a a! a> 255 a! d<>a until\n"
      "b=0 do a=c a>b if d=*b d! *d++ d=*d d-- *d=b b++ forever endif\n"
      "b=c b++ c=r 2 do a=c a>b if d=*b d! *d++ d=*d d-- *d=b b++ forever endif\n";
  if (arg0 <= 4) {          // blocks up to 16 MiB: the byte rides in the low 8 bits of the list entry
    s += "b=0 do a=c a>b if d=b a=*d a<<= 8 a+=*b *d=a b++ forever endif\n"
         "d=r 1 b=0 do a=d a== 0 ifnot a=*d a>>= 8 d=a\n";
    s += doe8 ? "*b=*d b++\n" : "a=*d out\n";
    s += "forever endif\n";
    if (doe8) s += std::string("d=b b=0 do ") + kE8Loop;
    s += "endif\nhalt\nend\n";
  } else if (doe8) {        // any block size, E8E9 undone on the fly through a 5-byte window in r4:r5
    s += "a=r 2 a-- r=a 2 c=0 d=r 1 do a=d a== 0 ifnot d=*d b=d a=*b a<<= 24 b=a a=r 4 r=a 5 a>>= 8 a|=b r=a 4\n"
         "a=c a> 3 if a=r 5 a&= 254 a== 232 if a=r 4 a>>= 24 b=a a++ a&= 254 a< 2 if\n"
         "a=r 4 a-=c a+= 4 a<<= 8 a>>= 8 b<>a a<<= 24 a+=b r=a 4 endif endif endif\n"
         "a=c a> 3 if a=r 5 out endif c++ forever endif\n"
         "b=r 4 a=c a> 3 a=b if out endif a>>= 8 b=a a=c a> 2 a=b if out endif a>>= 8 b=a\n"
         "a=c a> 1 a=b if out endif a>>= 8 b=a a=c a> 0 a=b if out endif\n"
         "endif\nhalt\nend\n";
  } else {
    s += "d=r 1 do a=d a== 0 ifnot d=*d b=d a=*b out forever endif\nendif\nhalt\nend\n";
  }
  return s;
}

// inverse E8E9 alone, through a 5-byte window in b (4 bytes) and *b
const char* const kPcompE8 =
    "pcomp e8e9 d ;\na> 255 if a=c a> 4 if c= 4 else a! a+= 5 a<<= 3 d=a a=b a>>=d b=a endif\n"
    "do a=c a> 0 if a=b out a>>= 8 b=a c-- forever endif\n"
    "else *b=b a<<= 24 d=a a=b a>>= 8 a+=d b=a c++ a=c a> 4 if a=*b out a&= 254 a== 232 if\n"
    "a=b a>>= 24 a++ a&= 254 a== 0 if a=b a>>= 24 a<<= 24 d=a a=b a-=c a+= 5 a<<= 8 a>>= 8 a|=d b=a\n"
    "endif endif endif endif\nhalt\nend\n";

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
  if (args[1] > 7) fail(ZPQ_E_ARG, "Unsupported method");
  if (type == 'i') fail(ZPQ_E_UNSUPPORTED, "index-block methods are outside this build's scope");
  std::string hdr, pcomp;
  if (level == 1) { hdr = "comp 9 16 0 $1+20 "; pcomp = pcomp_lz77_bits(args[0], doe8); }
  else if (level == 2) { hdr = "comp 9 16 0 $1+20 "; pcomp = pcomp_lz77_bytes(doe8); }
  else if (level == 3) { hdr = "comp 9 16 $1+20 $1+20 "; pcomp = pcomp_bwt(args[0], doe8); }
  else { hdr = "comp 9 16 0 0 "; pcomp = doe8 ? kPcompE8 : "end\n"; }

  ModelWriter w(args[0] + 20);
  if (level == 2)     // the model follows the byte-aligned LZ77 parse: r1 = 1 + bytes until the next code, r2 = the code
    w.raw_hcomp("a=r 1 a== 0 if a= " + num(111 + 57 * doe8) + " else a== 1 if a=*c r=a 2 a> 63 if a>>= 6 a++ a++ "
                "else a++ a++ endif else a-- endif endif r=a 1\n");
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
  return hdr + num(w.ncomp()) + "\n" + w.comp() + w.hcomp() + "halt\n" + pcomp;
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
