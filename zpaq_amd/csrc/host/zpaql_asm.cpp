// Clean-room ZPAQL assembler: ZPAQL source text -> block header bytes (COMP +
// HCOMP) and PCOMP bytes.  It replaces libzpaq::Compiler (libzpaq.cpp:
// 2494-2770) and must reproduce its output byte for byte, because the header is
// stored in the archive.
//
// Design: a tokenizer (whitespace-separated, nested "( )" comments, case
// insensitive), a mnemonic table GENERATED from the ISA's regular structure
// (operand codes a b c d *b *c *d / immediate) instead of a
// literal opcode list, and a small structured-control-flow emitter.
#include <cctype>
#include <cstring>
#include <map>
#include <mutex>

#include "common.hpp"

namespace zpq {

namespace {

enum Pseudo {
  P_POST = 256, P_PCOMP, P_END, P_IF, P_IFNOT, P_ELSE, P_ENDIF, P_DO, P_WHILE, P_UNTIL,
  P_FOREVER, P_IFL, P_IFNOTL, P_ELSEL, P_SEMI
};
enum { OP_JT = 39, OP_JF = 47, OP_JMP = 63, OP_LJ = 255 };

const std::map<std::string, int>& mnemonics() {
  static std::map<std::string, int> m;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* reg[7] = {"a", "b", "c", "d", "*b", "*c", "*d"};
    // 0..63: unary ops on each operand, then the irregular k==7 / g==7 slots
    const char* unary[5] = {"<>a", "++", "--", "!", "=0"};
    for (int g = 0; g < 7; ++g)
      for (int k = 0; k < 5; ++k)
        if (g || k) m[std::string(reg[g]) + unary[k]] = g * 8 + k;
    m["error"] = 0;
    for (int g = 0; g < 4; ++g) m[std::string(reg[g]) + "=r"] = g * 8 + 7;
    m["jt"] = OP_JT; m["jf"] = OP_JF; m["r=a"] = 55;
    m["halt"] = 56; m["out"] = 57; m["hash"] = 59; m["hashd"] = 60; m["jmp"] = OP_JMP;
    // 64..119: dst = src
    for (int g = 0; g < 7; ++g) {
      for (int k = 0; k < 7; ++k) m[std::string(reg[g]) + "=" + reg[k]] = 64 + g * 8 + k;
      m[std::string(reg[g]) + "="] = 64 + g * 8 + 7;
    }
    // 128..239: a OP= src
    const char* bin[14] = {"+=", "-=", "*=", "/=", "%=", "&=", "&~", "|=", "^=", "<<=", ">>=", "==", "<", ">"};
    for (int o = 0; o < 14; ++o) {
      for (int k = 0; k < 7; ++k) m[std::string("a") + bin[o] + reg[k]] = 128 + o * 8 + k;
      m[std::string("a") + bin[o]] = 128 + o * 8 + 7;
    }
    m["lj"] = OP_LJ;
    const char* pseudo[15] = {"post", "pcomp", "end", "if", "ifnot", "else", "endif", "do", "while",
                              "until", "forever", "ifl", "ifnotl", "elsel", ";"};
    for (int i = 0; i < 15; ++i) m[pseudo[i]] = 256 + i;
  });
  return m;
}

const char* const kCompName[10] = {"", "const", "cm", "icm", "match", "avg", "mix2", "mix", "isse", "sse"};
const int kCompLen[10] = {0, 2, 3, 2, 3, 4, 6, 6, 3, 5};

class Lexer {
 public:
  Lexer(const char* src, const int* args) : p_(src), args_(args), line_(1) {}

  // Next token (lower-cased); fails at end of input.
  std::string word() {
    skip();
    if (!*p_) err("unexpected end of config");
    const char* b = p_;
    while ((unsigned char)*p_ > ' ' && *p_ != '(') ++p_;
    std::string t(b, p_);
    for (auto& ch : t) ch = (char)tolower((unsigned char)ch);
    last_ = t;
    return t;
  }

  // Number token: decimal, optionally negative, or $N / $N+M (libzpaq.cpp:2559-2571).
  int number(int lo, int hi) {
    skip();
    if (!*p_) err("unexpected end of config");
    const char* b = p_;
    while ((unsigned char)*p_ > ' ' && *p_ != '(') ++p_;
    last_.assign(b, p_);
    long r = 0;
    if (b[0] == '$' && b[1] >= '1' && b[1] <= '9') {
      if (b[2] == '+') r = atoi_like(b + 3);
      if (args_) r += args_[b[1] - '1'];
    } else if (b[0] == '-' || (b[0] >= '0' && b[0] <= '9')) r = atoi_like(b);
    else err("expected a number");
    if (r < lo) err("number too low");
    if (r > hi) err("number too high");
    return (int)r;
  }

  // Raw text up to ';' (the "pcomp <cmd> ;" clause, case preserved).
  std::string until_semicolon() {
    skip();
    std::string s;
    while (*p_ && *p_ != ';') s.push_back(*p_++);
    if (*p_) ++p_;
    return s;
  }

  [[noreturn]] void err(const std::string& msg) {
    fail(ZPQ_E_HEADER, "Config line " + std::to_string(line_) + " at " + last_ + ": " + msg);
  }

 private:
  static long atoi_like(const char* s) {
    long sign = 1, v = 0;
    if (*s == '-') { sign = -1; ++s; } else if (*s == '+') ++s;
    while (*s >= '0' && *s <= '9') { v = v * 10 + (*s - '0'); if (v > 100000000) break; ++s; }
    return sign * v;
  }
  void skip() {
    int depth = 0;
    for (; *p_; ++p_) {
      if (*p_ == '\n') ++line_;
      if (*p_ == '(') ++depth;
      else if (depth > 0) { if (*p_ == ')') --depth; }
      else if ((unsigned char)*p_ > ' ') break;
    }
  }
  const char* p_;
  const int* args_;
  int line_;
  std::string last_;
};

// Emits one HCOMP/PCOMP program; returns the terminating pseudo-op.
int assemble_code(Lexer& lx, std::vector<U8>& code, size_t header_overhead) {
  const auto& mn = mnemonics();
  std::vector<int> if_stack, do_stack;   // positions in `code`
  auto pop = [&](std::vector<int>& st, const char* what) {
    if (st.empty()) lx.err(what);
    int v = st.back(); st.pop_back(); return v;
  };
  for (;;) {
    const std::string tok = lx.word();
    auto it = mn.find(tok);
    if (it == mn.end()) lx.err("unexpected");
    int op = it->second;
    if (op == P_POST || op == P_PCOMP || op == P_END) { code.push_back(0); return op; }
    int operand = -1, operand2 = -1;
    const int here = (int)code.size();
    if (op == P_IF || op == P_IFNOT) {
      op = (op == P_IF) ? OP_JF : OP_JT;
      operand = 0;
      if_stack.push_back(here + 1);
    } else if (op == P_IFL || op == P_IFNOTL) {
      code.push_back(op == P_IFL ? OP_JT : OP_JF);
      code.push_back(3);
      op = OP_LJ;
      operand = operand2 = 0;
      if_stack.push_back((int)code.size() + 1);
    } else if (op == P_ELSE || op == P_ELSEL) {
      const bool longj = (op == P_ELSEL);
      op = longj ? OP_LJ : OP_JMP;
      operand = 0;
      if (longj) operand2 = 0;
      const int a = pop(if_stack, "unmatched IF or DO");
      if (code[a - 1] != OP_LJ) {
        const int j = here - a + 1 + (longj ? 1 : 0);
        if (j > 127) lx.err("IF too big, try IFL, IFNOTL");
        code[a] = (U8)j;
      } else {
        const int j = here + 2 + (longj ? 1 : 0);
        code[a] = (U8)(j & 255);
        code[a + 1] = (U8)(j >> 8);
      }
      if_stack.push_back(here + 1);
    } else if (op == P_ENDIF) {
      const int a = pop(if_stack, "unmatched IF or DO");
      if (code[a - 1] != OP_LJ) {
        const int j = here - a - 1;
        if (j > 127) lx.err("IF too big, try IFL, IFNOTL, ELSEL");
        code[a] = (U8)j;
      } else {
        code[a] = (U8)(here & 255);
        code[a + 1] = (U8)(here >> 8);
      }
      continue;
    } else if (op == P_DO) {
      do_stack.push_back(here);
      continue;
    } else if (op == P_WHILE || op == P_UNTIL || op == P_FOREVER) {
      const int a = pop(do_stack, "unmatched IF or DO");
      const int j = a - here - 2;
      if (j >= -127) {
        operand = j & 255;
        op = (op == P_WHILE) ? OP_JT : (op == P_UNTIL) ? OP_JF : OP_JMP;
      } else {
        if (op == P_WHILE) { code.push_back(OP_JF); code.push_back(3); }
        if (op == P_UNTIL) { code.push_back(OP_JT); code.push_back(3); }
        op = OP_LJ;
        operand = a & 255;
        operand2 = a >> 8;
      }
    } else if (op == P_SEMI) {
      continue;             // a stray ";" inside a program emits nothing (compile_comp stores only op <= 255, libzpaq.cpp:2686)
    } else if ((op & 7) == 7) {
      if (op == OP_LJ) {
        const int v = lx.number(0, 65535);
        operand = v & 255;
        operand2 = v >> 8;
      } else if (op == OP_JT || op == OP_JF || op == OP_JMP) operand = lx.number(-128, 127) & 255;
      else operand = lx.number(0, 255);
    }
    code.push_back((U8)op);
    if (operand >= 0) code.push_back((U8)operand);
    if (operand2 >= 0) code.push_back((U8)operand2);
    if (code.size() + header_overhead > 65535) lx.err("program too big");
  }
}

}  // namespace

Assembled assemble(const char* source, const int* args9) {
  if (!source) fail(ZPQ_E_ARG, "null config");
  Lexer lx(source, args9);
  Assembled out;
  auto expect = [&](const char* w) { if (lx.word() != w) lx.err(std::string("expected ") + w); };
  expect("comp");
  std::vector<U8> head(7, 0);
  head[2] = (U8)lx.number(0, 255);  // hh
  head[3] = (U8)lx.number(0, 255);  // hm
  head[4] = (U8)lx.number(0, 255);  // ph
  head[5] = (U8)lx.number(0, 255);  // pm
  const int n = lx.number(0, 255);
  head[6] = (U8)n;
  for (int i = 0; i < n; ++i) {
    lx.number(i, i);
    const std::string name = lx.word();
    int type = 0;
    for (int t = 1; t < 10; ++t) if (name == kCompName[t]) type = t;
    if (!type) lx.err("unexpected");
    head.push_back((U8)type);
    for (int j = 1; j < kCompLen[type]; ++j) head.push_back((U8)lx.number(0, 255));
  }
  head.push_back(0);  // COMP END
  expect("hcomp");
  std::vector<U8> code;
  const int endtok = assemble_code(lx, code, head.size() - 2);
  const size_t hsize = head.size() - 2 + code.size();
  head[0] = (U8)(hsize & 255);
  head[1] = (U8)(hsize >> 8);
  out.hcomp = head;
  out.hcomp.insert(out.hcomp.end(), code.begin(), code.end());
  if (endtok == P_POST) {
    lx.number(0, 0);
    expect("end");
  } else if (endtok == P_PCOMP) {
    out.pcomp_cmd = lx.until_semicolon();
    std::vector<U8> pcode;
    if (assemble_code(lx, pcode, 6) != P_END) lx.err("expected END");
    out.pcomp.push_back((U8)(pcode.size() & 255));
    out.pcomp.push_back((U8)(pcode.size() >> 8));
    out.pcomp.insert(out.pcomp.end(), pcode.begin(), pcode.end());
  }
  return out;
}

}  // namespace zpq
