// Per-header kernel specialisation: header -> HIP source for gfx950.
//
// The reference speeds its CPU path up by JIT-compiling predict/update and the
// HCOMP program to x86 per block header (libzpaq.cpp:3824-4583, 3231-3811).  The
// MI355X analogue: emit a tiny translation unit that (a) pins the COMP list as
// constexpr data for the hand-written kernel template in device/spec_kernel.h
// and (b) translates the HCOMP bytecode to straight-line C++
// with gotos.  The text is compiled ahead of time for the standard method
// chains (zpaq_amd/prebuild.py -> zpaq_amd/spec_cache/*.hsaco) or at run time
// through hipRTC (device/spec_loader.cpp).
#include "codegen.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <algorithm>

namespace zpq {

namespace {

struct Insn { int pc, len, op; };

// operand expression (read)
std::string src_expr(int k, int imm) {
  switch (k) {
    case 0: return "a";
    case 1: return "b";
    case 2: return "c";
    case 3: return "d";
    case 4: return "(unsigned)M[b & MMASK]";
    case 5: return "(unsigned)M[c & MMASK]";
    case 6: return "H[d & HMASK]";
    default: return std::to_string(imm) + "u";
  }
}
// assignment to operand g of expression e
std::string dst_stmt(int g, const std::string& e) {
  switch (g) {
    case 0: return "a = " + e + ";";
    case 1: return "b = " + e + ";";
    case 2: return "c = " + e + ";";
    case 3: return "d = " + e + ";";
    case 4: return "M[b & MMASK] = (unsigned char)(" + e + ");";
    case 5: return "M[c & MMASK] = (unsigned char)(" + e + ");";
    default: return "H[d & HMASK] = " + e + ";";
  }
}

// Translates the HCOMP program; returns false if it cannot be compiled
// statically (then the generic interpreter kernel is used instead).
bool translate_hcomp(const U8* prog, int len, std::ostringstream& out, bool with_out = false) {
  std::map<int, Insn> insns;
  std::vector<int> work(1, 0);
  std::set<int> bad;    // pcs where execution is an error
  auto decode_len = [&](int pc) -> int {
    const int op = prog[pc];
    if (op == 255) return 3;
    return (op & 7) == 7 ? 2 : 1;
  };
  while (!work.empty()) {
    const int pc = work.back();
    work.pop_back();
    if (insns.count(pc) || bad.count(pc)) continue;
    if (pc < 0 || pc >= len) { bad.insert(pc); continue; }
    const int op = prog[pc];
    const int l = decode_len(pc);
    if (pc + l > len) { bad.insert(pc); continue; }
    insns[pc] = Insn{pc, l, op};
    if (insns.size() > 20000) return false;
    const int imm = l >= 2 ? prog[pc + 1] : 0;
    if (op == 56) continue;                                            // HALT
    if (op == 63) { work.push_back(pc + 2 + (int)(int8_t)imm); continue; }   // JMP
    if (op == 39 || op == 47) work.push_back(pc + 2 + (int)(int8_t)imm);      // JT / JF
    if (op == 255) { work.push_back(prog[pc + 1] + 256 * prog[pc + 2]); continue; }
    work.push_back(pc + l);
  }
  // R[n] is only ever addressed by an immediate: the slots a program uses become locals, loaded on entry and written
  // back at HALT, so no R access sits in the machine's dependent chain as a memory round trip
  std::set<int> r_used, r_written;
  for (auto& kv : insns) {
    const Insn& in = kv.second;
    if (in.op < 64 && (in.op & 7) == 7 && in.len >= 2) {
      const int g = in.op >> 3;
      if (g < 4) r_used.insert(prog[in.pc + 1]);
      else if (g == 6) { r_used.insert(prog[in.pc + 1]); r_written.insert(prog[in.pc + 1]); }
    }
  }
  if (with_out)     // PCOMP: the same machine plus OUT (libzpaq.cpp:1177), and loops as long as the block
    out << "  template <class MP, class HP, class RP, class OP>\n"
           "  static __device__ __forceinline__ int pcomp(unsigned input, unsigned& rb, unsigned& rc, unsigned& rd,\n"
           "                                              unsigned& rf, MP M, HP H, RP R, OP& out) {\n"
           "    unsigned a = input, b = rb, c = rc, d = rd, f = rf;\n"
           "    unsigned budget = " << (1u << 30) << "u;\n";
  else
  out << "  template <class MP, class HP, class RP>\n"
         "  static __device__ __forceinline__ int hcomp(unsigned input, unsigned& rb, unsigned& rc, unsigned& rd,\n"
         "                                              unsigned& rf, MP M, HP H, RP R) {\n"
         "    unsigned a = input, b = rb, c = rc, d = rd, f = rf;\n"
         "    unsigned budget = " << kMaxVmSteps << "u;\n";
  out <<
         "    (void)R; (void)budget;\n";
  for (int r : r_used) out << "    unsigned r_" << r << " = R[" << r << "];\n";
  out << "    goto L0;\n";
  auto label = [&](int pc) -> std::string {
    if (insns.count(pc)) return "L" + std::to_string(pc);
    return "Lerr";
  };
  for (auto it = insns.begin(); it != insns.end(); ++it) {
    const Insn& in = it->second;
    const int pc = in.pc, op = in.op, g = op >> 3, k = op & 7;
    const int imm = in.len >= 2 ? prog[pc + 1] : 0;
    const int next = pc + in.len;
    std::string st;
    bool falls = true;
    auto jump = [&](int target, const std::string& cond) {
      std::string s = cond.empty() ? "" : "if (" + cond + ") ";
      if (target <= pc) s += "{ if (--budget == 0u) goto Lerr; goto " + label(target) + "; }";
      else s += "goto " + label(target) + ";";
      return s;
    };
    if (op < 64) {
      if (g == 7) {
        if (op == 56) { st = "goto Lhalt;"; falls = false; }
        else if (op == 57) st = with_out ? "out(a);" : ";";
        else if (op == 59) st = "a = (a + (unsigned)M[b & MMASK] + 512u) * 773u;";
        else if (op == 60) st = "H[d & HMASK] = (H[d & HMASK] + a + 512u) * 773u;";
        else if (op == 63) { st = jump(pc + 2 + (int)(int8_t)imm, ""); falls = false; }
        else { st = "goto Lerr;"; falls = false; }
      } else if (k == 7) {
        if (g < 4) st = dst_stmt(g, "r_" + std::to_string(imm));
        else if (g == 4) st = jump(pc + 2 + (int)(int8_t)imm, "f");
        else if (g == 5) st = jump(pc + 2 + (int)(int8_t)imm, "!f");
        else st = "r_" + std::to_string(imm) + " = a;";
      } else if (op == 0 || k > 4) { st = "goto Lerr;"; falls = false; }
      else if (k == 0) {
        if (g == 4 || g == 5) {
          const std::string m = g == 4 ? "M[b & MMASK]" : "M[c & MMASK]";
          st = "{ const unsigned x = " + m + "; " + m + " = (unsigned char)a; a = (a & 0xFFFFFF00u) | x; }";
        } else st = "{ const unsigned x = " + src_expr(g, 0) + "; " + dst_stmt(g, "a") + " a = x; }";
      } else if (k == 1) st = dst_stmt(g, src_expr(g, 0) + " + 1u");
      else if (k == 2) st = dst_stmt(g, src_expr(g, 0) + " - 1u");
      else if (k == 3) st = dst_stmt(g, "~" + src_expr(g, 0));
      else st = dst_stmt(g, "0u");
    } else if (op < 120) {
      st = dst_stmt(g - 8, src_expr(k, imm));
    } else if (op < 128) { st = "goto Lerr;"; falls = false; }
    else if (op < 240) {
      const std::string v = src_expr(k, imm);
      switch (g - 16) {
        case 0: st = "a += " + v + ";"; break;
        case 1: st = "a -= " + v + ";"; break;
        case 2: st = "a *= " + v + ";"; break;
        case 3: st = "{ const unsigned x = " + v + "; a = x ? a / x : 0u; }"; break;
        case 4: st = "{ const unsigned x = " + v + "; a = x ? a % x : 0u; }"; break;
        case 5: st = "a &= " + v + ";"; break;
        case 6: st = "a &= ~(" + v + ");"; break;
        case 7: st = "a |= " + v + ";"; break;
        case 8: st = "a ^= " + v + ";"; break;
        case 9: st = "a <<= ((" + v + ") & 31u);"; break;
        case 10: st = "a >>= ((" + v + ") & 31u);"; break;
        case 11: st = "f = zpq::vm_flag(a == (" + v + "));"; break;
        case 12: st = "f = zpq::vm_flag(a < (" + v + "));"; break;
        default: st = "f = zpq::vm_flag(a > (" + v + "));"; break;
      }
    } else if (op == 255) { st = jump(prog[pc + 1] + 256 * prog[pc + 2], ""); falls = false; }
    else { st = "goto Lerr;"; falls = false; }
    out << "    L" << pc << ": " << st;
    if (falls) {
      auto nx = std::next(it);
      if (nx == insns.end() || nx->first != next) out << " goto " << label(next) << ";";
    }
    out << "\n";
  }
  out << "    Lhalt:";
  for (int r : r_written) out << " R[" << r << "] = r_" << r << ";";
  out << " rb = b; rc = c; rd = d; rf = f; return 0;\n"
         "    Lerr: return 5;\n"
         "  }\n";
  return true;
}

}  // namespace

// the lockstep decoder's TAIL wavefront (device/spec_team_kernel.h team_tail_map: the same rule): a chain with a MIX, at most 8
// CM / MATCH components, and no MIX fed by an AVG / MIX2 / SSE.  OFF by default: measured on the MI355X (profiles/r06, call 2:
// 2 048 x 1 MiB of the mixed corpus, every byte verified) the form with the tail wavefront decodes at 139.6 MB/s against 159.5
// without it -- the bit is a latency chain (rows -> MIX -> MIX2 / SSE -> coder), not an issue-slot budget, and the tail adds a
// barrier and an LDS hand-over to that chain.  ZPAQ_AMD_TEAM_TAIL=1 builds it (tests keep it bit-exact).
bool team_tail_wanted() {
  static const bool on = [] { const char* v = getenv("ZPAQ_AMD_TEAM_TAIL"); return v && v[0] == '1'; }();
  return on;
}
static bool team_tail_ok(const zpq_plan& plan) {
  if (!team_tail_wanted()) return false;
  const CompDesc* comp = plan.comps();
  int nmix = 0, nrole = 0;
  for (uint32_t i = 0; i < plan.hdr().n; ++i) {
    nmix += comp[i].type == C_MIX;
    nrole += comp[i].type == C_CM || comp[i].type == C_MATCH;
    if (comp[i].type == C_MIX)
      for (uint32_t t = 0; t < comp[i].a3; ++t) {
        const uint32_t ty = comp[comp[i].a2 + t].type;
        if (ty == C_AVG || ty == C_MIX2 || ty == C_SSE) return false;
      }
  }
  return nmix > 0 && nrole <= 8;
}
int team_threads(const zpq_plan& plan) {
  int rows = 0;
  for (uint32_t i = 0; i < plan.hdr().n; ++i) rows += plan.comps()[i].type == C_ICM || plan.comps()[i].type == C_ISSE;
  return (rows <= 16 ? 384 : 512) + (team_tail_ok(plan) ? 64 : 0);
}

bool generate_spec_source(const zpq_plan& plan, int waves, std::string& source, std::string& why_not, int shape) {
  const bool team = shape == 2;
  const bool dual = shape == 1 || team;          // (the lockstep decoder's mixer wavefronts are the dual kernel's: same limits)
  const PlanHeader& ph = plan.hdr();
  const int n = (int)ph.n;
  if (n < 1 || n > 64) { why_not = "more than 64 components"; return false; }
  if (ph.arena_bytes >= (1ull << 32)) { why_not = "model state of 4 GiB or more per block"; return false; }
  const CompDesc* comp = plan.comps();
  // LDS plan for the wave's region: H first, then ICM/ISSE side tables while they fit
  if (waves != 4 && waves != 8) { why_not = "unsupported workgroup shape"; return false; }
  if (dual) {
    if (waves != 8) { why_not = "two blocks per wavefront use the LDS plan of the 8-block shape"; return false; }
    if (n > 32) { why_not = "more than 32 components"; return false; }
    if (ph.arena_bytes >= (1ull << 31)) { why_not = "model state of 2 GiB or more per block"; return false; }
  }
  if (team) {
    if (ph.arena_bytes * 8 >= (1ull << 32)) { why_not = "eight blocks' model state does not fit a 4 GiB window"; return false; }
    if (4u * (ph.hmask + 1) > 4096u) { why_not = "H does not fit the LDS"; return false; }
    int prev_row = -1;
    for (int i = 0; i < n; ++i) {
      if (comp[i].type == C_ISSE && (prev_row < 0 || comp[i].a2 != (uint32_t)prev_row)) {
        why_not = "an ISSE that is not fed by the ICM / ISSE before it";
        return false;
      }
      if (comp[i].type == C_ICM || comp[i].type == C_ISSE) prev_row = i;
    }
  }
  const int wave_lds = team ? team_block_lds_bytes() : spec_wave_lds_bytes(waves);          // LDS of ONE block
  int lds_used = 0, h_lds = -1;
  const int h_bytes = (int)(4u * (ph.hmask + 1));
  if (h_bytes <= 4096) { h_lds = 0; lds_used = (h_bytes + 15) & ~15; }
  std::ostringstream o;
  o << "// generated by zpaq_amd codegen v" << kCodegenVersion << " -- do not edit\n"
    << (team && team_tail_wanted() ? "#define ZPQ_TEAM_TAIL 1\n" : "")
    << "#include \"" << (team ? "spec_team_kernel.h" : (dual ? "spec_dual_kernel.h" : "spec_kernel.h")) << "\"\n"
       "namespace zpq_gen {\n"
       "struct Chain {\n";
  int nmix = 0, nsse = 0;
  bool any_global_side = false, any_nonpf_gl = false;
  std::ostringstream comps;
  for (int i = 0; i < n; ++i) {
    const CompDesc& c = comp[i];
    int lds = -1, slot = -1;
    if (c.type == C_ICM || c.type == C_ISSE) {
      const int bytes = team ? (c.type == C_ICM ? kTeamIcmLds : kTeamIsseLds) : (c.type == C_ICM ? 1024 : 2048);   // (the lockstep decoder packs its entries)
      if (lds_used + bytes <= wave_lds - 512) { lds = lds_used; lds_used += bytes; }   // last 512 B: dummy slots
      else any_global_side = true;
    }
    if (c.type == C_CM && c.mask0 < 511u && c.mask0 != 0u) any_nonpf_gl = true;
    if (c.type == C_MIX2 && !(c.a5 == 255u && c.mask0 >= 255u) && c.mask0 != 0u) any_nonpf_gl = true;
    if (c.type == C_MIX) slot = nmix++;
    if (c.type == C_SSE) slot = nsse++;
    if (c.type == C_MIX && (c.a3 > 64 || c.a2 + c.a3 > 64)) { why_not = "MIX wider than a wavefront"; return false; }
    if (dual && c.type == C_MIX && c.a2 + c.a3 > 32) { why_not = "MIX wider than half a wavefront"; return false; }
    comps << "    {" << c.type << "u," << c.a1 << "u," << c.a2 << "u," << c.a3 << "u," << c.a4 << "u," << c.a5 << "u, "
          << c.limit << "u," << c.mask0 << "u," << c.mask1 << "u, " << c.t0 << "ull," << c.t1 << "ull, " << lds << ","
          << slot << "," << c.stride << "u},\n";
  }
  o << "  static constexpr int N = " << n << ", NMIX = " << nmix << ", NSSE = " << nsse << ", WAVES = " << waves << ";\n"
    << "  static constexpr unsigned HMASK = " << ph.hmask << "u, MMASK = " << ph.mmask << "u;\n"
    << "  static constexpr int H_LDS = " << h_lds << ";\n"
    << "  static constexpr bool ANY_GLOBAL_SIDE = " << (any_global_side ? "true" : "false")
    << ", ANY_NONPF_GL = " << (any_nonpf_gl ? "true" : "false") << ";\n"
    << "  static constexpr unsigned long long OFF_RUN = " << ph.off_run << "ull;\n"
    << "  static constexpr unsigned long long OFF_H = " << ph.off_H << "ull, OFF_M = " << ph.off_M
    << "ull, OFF_R = " << ph.off_R << "ull;\n"
    << "  static constexpr zpq::CompK comp[N] = {\n" << comps.str() << "  };\n";
  const U8* prog = plan.blob.data() + ph.off_prog;
  if (!translate_hcomp(prog, (int)ph.prog_len, o)) { why_not = "HCOMP program too irregular to translate"; return false; }
  const char* body = "zpq::spec_kernel_body";
  if (team) {
    o << "};\n"
         "}  // namespace zpq_gen\n"
         "extern \"C\" __global__ __launch_bounds__(" << team_threads(plan) << ") void zpq_spec_decode3(const zpq::BlockJob* jobs, "
         "zpq::BlockResult* res, unsigned nblocks, const zpq::DeviceTables* tb) {\n"
         "  zpq::spec_team_decode_body<zpq_gen::Chain>(jobs, res, nblocks, tb);\n}\n";
    source = o.str();
    return true;
  }
  if (dual) {
    o << "};\n"
         "}  // namespace zpq_gen\n"
         "extern \"C\" __global__ __launch_bounds__(256) void zpq_spec_decode2(const zpq::BlockJob* jobs, "
         "zpq::BlockResult* res, unsigned nblocks, const zpq::DeviceTables* tb) {\n"
         "  zpq::spec_dual_decode_body<zpq_gen::Chain>(jobs, res, nblocks, tb);\n}\n";
    source = o.str();
    return true;
  }
  o << "};\n"
       "}  // namespace zpq_gen\n"
       "extern \"C\" __global__ __launch_bounds__(64 * zpq_gen::Chain::WAVES) void zpq_spec_encode(const zpq::BlockJob* jobs, "
       "zpq::BlockResult* res, unsigned nblocks, const zpq::DeviceTables* tb) {\n"
       "  " << body << "<zpq_gen::Chain, false>(jobs, res, nblocks, tb);\n}\n"
       "extern \"C\" __global__ __launch_bounds__(64 * zpq_gen::Chain::WAVES) void zpq_spec_decode(const zpq::BlockJob* jobs, "
       "zpq::BlockResult* res, unsigned nblocks, const zpq::DeviceTables* tb) {\n"
       "  " << body << "<zpq_gen::Chain, true>(jobs, res, nblocks, tb);\n}\n";
  source = o.str();
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// Pipelined encoder: dataflow levels + buffer layout (see device/pipe_kernel.h for the design)
PipeOptions pipe_options(int variant) {
  PipeOptions o;
  o.mode = variant ? 1 : 0;
  o.chunk = variant == 2 ? 2048 : 0;
  o.wide = variant == 3;
  return o;
}

// The persistent launch: units, their dependencies, and a packing of a group's unit wavefronts into workgroups whose LDS
// fits a CU (device/pipe_persist.h).  Leaves L.persist_ok false (with a reason) when the chain cannot be packed; the six
// kernels then code it step by step.
static const int kPersistRoBytes = 4096 + 512 + 2688 + (2016 * 4 + 512) + 1024;      // = sizeof(zpq::PipeRO) on the device
static const int kPersistLdsCap = 163840 - 512;                                       // gfx950: 160 KiB per workgroup
static uint32_t mix_pstride(uint32_t m) {              // = device pipe_mix_pstride
  const uint32_t need = 12u * ((m + 3u) / 4u);
  uint32_t p = 16u;
  while (p < need) p *= 2u;
  return p;
}
// lds_rows_wish: rows of a SMALL packed MIX table (up to 256 rows: the mixer selected by the partial byte alone) kept in the LDS
static void plan_persistent(const zpq_plan& plan, PipeLayout& L, int lds_rows_wish, bool halves) {
  const CompDesc* comp = plan.comps();
  const PlanHeader& ph = plan.hdr();
  const int n = L.n, G = L.G;
  enum { K_CONS = 2, K_CM, K_MATCH, K_AVG, K_MIX2, K_SSE, K_CODER, K_CM_BITS, K_MIX2_BITS, K_SSE_BITS };
  L.persist_ok = false;
  L.ps_slots.clear(); L.ps_deps.clear();
  // units: 0 = HCOMP, then per component its ROW unit (ICM / ISSE) and the unit that writes its p stream, then the coder
  int nunit = 1;
  std::vector<int> row_unit(n, -1), p_unit(n, -1);
  for (int i = 0; i < n; ++i) {
    if (L.row[i] >= 0) row_unit[i] = nunit++;
    p_unit[i] = nunit++;
  }
  const int coder_unit = nunit++;
  L.ps_nunit = nunit;
  std::vector<int> unit_waves(nunit, 0);
  std::vector<PipeLayout::Slot> slots;
  // `lines` of a unit wavefront: the distinct 128-byte memory lines it asks for per input byte -- rows of the model's tables AND the
  // elements of the streams it reads and writes (a ctx element is 4 bytes per block, bh 8, p 16: G blocks side by side).  Two
  // weights exist for A/Bs -- lines of a table small enough to stay on the die (up to 256 KiB per block: 1024 blocks = the
  // Infinity Cache), stream lines -- and both are 1: measured (profiles/r06 calls 5-7, six packings) counting every line alike
  // packs as well as any weighting tried (0.7 / 0.25: the same; 0.5 / 0: -5 %; 0.3 / 0.25: -10 %).
  static const float ondie_w = [] { const char* v = getenv("ZPAQ_AMD_PACK_ONDIE_WEIGHT"); return v ? (float)atof(v) : 1.0f; }();
  static const float stream_w = [] { const char* v = getenv("ZPAQ_AMD_PACK_STREAM_WEIGHT"); return v ? (float)atof(v) : 1.0f; }();
  // MATCH: the index entry, the history behind the candidate, the byte a match predicts -- three lines per block and byte
  // (measured against 2 and 4, profiles/r06 call 12: text 389.7 -> 395.4 MB/s, 4 MiB blocks of zeros -- MATCH's worst case -- 399 -> 431)
  static const float match_lines = [] { const char* v = getenv("ZPAQ_AMD_PACK_MATCH_LINES"); return v ? (float)atof(v) : 3.0f; }();
  const float ctx_l = G * 4 / 128.f, bh_l = G * 8 / 128.f, p_l = G * 16 / 128.f;
  auto add = [&](int kind, int role, int sub, int unit, int lds, float cost, float lines = 0.f, float stream_lines = 0.f) {
    PipeLayout::Slot s; s.kind = kind; s.role = role; s.sub = sub; s.unit = unit; s.lds = (lds + 255) & ~255; s.cost = cost;
    s.lines = lines + stream_w * stream_lines;
    slots.push_back(s); ++unit_waves[unit];
  };
  auto line_weight = [](uint64_t table_bytes) { return table_bytes <= (256u << 10) ? ondie_w : 1.0f; };
  // Latency shape (round 6, profiles/r06_results.md section 5): a launch that leaves the machine empty goes at the pace of its
  // longest per-bit instruction stream.  Its coder stores once per bit instead of keeping a window (pipe_coder_fast), and a
  // chain of a few units (configs[1]'s n = 2: six wavefronts per group) gets a SIMD per wavefront -- two workgroups of 4 instead
  // of one of 6, where the ROW units shared their SIMDs with the ICM map and HCOMP -- and, with LDS to spare, unpacked ISSE
  // pairs and whole squash / stretch tables, ROW units with a lane per nibble, streams read four bytes ahead.  "A few": up to 32
  // unit wavefronts per group -- mid.cfg (24) went from 139 to 205 MB/s on 256 blocks when it was let in (profiles/r06 call 19);
  // the -m5 chains (110 and more) keep the general latency shape.
  static const bool fast_off = [] { const char* v = getenv("ZPAQ_AMD_CODER_FAST"); return v && v[0] == '0'; }();
  static const bool small_off = [] { const char* v = getenv("ZPAQ_AMD_SMALL_CHAIN"); return v && v[0] == '0'; }();
  static const int small_max = [] { const char* v = getenv("ZPAQ_AMD_SMALL_CHAIN_WAVES"); return v ? atoi(v) : 32; }();
  L.ps_coder_fast = L.mode == 1 && !fast_off;
  int small_waves = 0;
  {
    int waves = std::max(1, G / std::min(L.hcomp_lanes, G)) + (int)L.rows.size() + (int)L.light.size() + (int)L.icm.size() + (int)L.isse.size();
    for (size_t r = 0; r < L.mix.size(); ++r) waves += L.mix_bits ? L.mix_waves_of(L.mix_ql[r]) : std::max(1, G * L.mix_ql[r] / 64);
    L.ps_small = L.mode == 1 && !small_off && waves <= small_max;
    small_waves = waves;
    // Variant 3 of a larger chain (a wavefront per SIMD: 28 workgroups per group for -m5, LDS to spare in every one) takes the
    // small chains' LDS-rich ICM / ISSE maps as well -- the packed ISSE maps set its pace (profiles/r06 call 29) -- but keeps the
    // ROW units with a lane per block: with a lane per nibble some lane of the 64 re-fetches a clashing row at nearly every
    // byte of text (call 30).  ZPAQ_AMD_WIDE_RICH=0: without.
    static const bool wide_rich_off = [] { const char* v = getenv("ZPAQ_AMD_WIDE_RICH"); return v && v[0] == '0'; }();
    L.ps_row_halves = L.ps_small;
    if (L.ps_wide && L.mode == 1 && !small_off && !wide_rich_off) L.ps_small = true;
    // ... and so does variant 1 of a larger chain (workgroups of 8; -m5: 16 per group instead of 14, so up to 16 groups): call 34,
    // 320 / 384 / 512 blocks 167.1 / 201.5 / 254.8 -> 199.9 / 238.4 / 301.0 MB/s.  ZPAQ_AMD_LATENCY_RICH=0: without.
    static const bool latency_rich_off = [] { const char* v = getenv("ZPAQ_AMD_LATENCY_RICH"); return v && v[0] == '0'; }();
    L.ps_icm_full = L.ps_small;
    // (in workgroups of 8 an ICM map keeps the compact stretch table: with 64 KiB more per ICM map every one of the -m5 chain's 16
    //  maps needs a workgroup of its own, 16 per group -- and two chains of 8 groups each, the archiver's batch, then need every
    //  compute unit of the device at once: call 36, `add` 3.1 -> 4.3 s; without, 14 per group as before)
    if (!latency_rich_off && L.mode == 1 && !small_off) L.ps_small = true;
    // (ZPAQ_AMD_LATENCY_ICM_FULL=1: call 34's form -- whole stretch tables there too, 16 workgroups per group: 320 / 384 / 512 blocks
    //  200 / 238 / 301 MB/s against this form's 168 / 199 / 248 (call 37), at the price named above)
    static const bool latency_icm_full = [] { const char* v = getenv("ZPAQ_AMD_LATENCY_ICM_FULL"); return v && v[0] == '1'; }();
    if (latency_icm_full && L.ps_small) L.ps_icm_full = true;
  }
  // (a small chain's units read their streams four bytes ahead: device pipe_icm_unit)
  static const int ahead = [] { const char* v = getenv("ZPAQ_AMD_STREAM_AHEAD"); return v ? (atoi(v) >= 3 ? 3 : (atoi(v) >= 1 ? 1 : 0)) : 3; }();        // (rings of 1, 2 or 4 slots: a chunk's length is a multiple)
  // (the other chains: measured behind knobs -- ZPAQ_AMD_STREAM_AHEAD_BIG for the ICM / ISSE maps, ZPAQ_AMD_ROW_RING for the ROW units)
  static const int ahead_big = [] { const char* v = getenv("ZPAQ_AMD_STREAM_AHEAD_BIG"); return v ? (atoi(v) >= 3 ? 3 : (atoi(v) >= 1 ? 1 : 0)) : 0; }();
  static const bool row_ring = [] { const char* v = getenv("ZPAQ_AMD_ROW_RING"); return v && v[0] == '1'; }();
  L.ps_ahead = L.ps_small ? ahead : ahead_big;
  L.ps_row_ring = row_ring;
  const bool small = L.ps_small;
  const int hl = std::min(L.hcomp_lanes, G);
  // HCOMP's M array of up to 256 bytes beside H (round 6: the legacy models' programs write the byte into M and read the last
  // seven back -- loads behind a store, a memory round trip per byte: mid.cfg's HCOMP unit set the pace of its launch)
  L.hcomp_m_lds = ph.mmask + 1u <= 256u;
  const int hbytes = (L.hcomp_h_lds ? (int)(4u * (ph.hmask + 1u)) * L.hcomp_lanes : 16) + (L.hcomp_m_lds ? (int)(ph.mmask + 1u) * L.hcomp_lanes : 0);
  // relative time per chunk of a unit wavefront inside a full launch (-m5, 1024 blocks, profiles/r05/call5: ms per 2049 chunks / 1000)
  for (int sub = 0; sub < G / hl; ++sub) add(0, 0, sub, 0, hbytes, 0.94f, 0.f, (float)L.nctx * ctx_l * hl / G);
  for (size_t r = 0; r < L.rows.size(); ++r)      // two finds per byte and block: two lines
    add(1, (int)r, 0, row_unit[L.rows[r]], 0, 1.8f, 2.f * G * line_weight((uint64_t)comp[L.rows[r]].mask1 + 1u), ctx_l + bh_l);
  for (size_t r = 0; r < L.light.size(); ++r) {
    const int k = L.light[r].first, i = L.light[r].second;
    static const float lc[12] = {0, 0, 0.1f, 2.5f, 3.1f, 0.3f, 1.5f, 2.5f, 1.43f, 0.45f, 0.45f, 0.78f};
    // (the workgroups of a light unit with a lane per bit position share their code: role = the first of them, sub = which eighth of the group)
    int code_role = (int)r;
    for (size_t r2 = 0; r2 < r; ++r2) if (L.light[r2] == L.light[r]) { code_role = (int)r2; break; }
    // a CM / MIX2 with a table of up to 512 words keeps it in the LDS (device/pipe_persist.h pipe_light_lds_words)
    int lds = 0;
    if ((k == K_CM || (k == K_MIX2 && comp[i].mask0 != 0u)) && comp[i].mask0 + 1u <= 512u && comp[i].mask0 >= 3u) lds = (int)(comp[i].mask0 + 1u) * G * 4;
    // lines per byte of the unit's wavefront: CM 2 per block (a nibble's four words share a line), MATCH ~2, MIX2 / SSE one per bit;
    // with a lane per bit position a wavefront holds 8 blocks
    float lines = 0.f;
    if (!lds) {
      const float wcm = line_weight(4ull * (comp[i].mask0 + 1ull));
      if (k == K_CM) lines = 2.f * G * wcm;
      else if (k == K_CM_BITS) lines = 2.f * 8 * wcm;
      else if (k == K_MATCH) lines = match_lines * G;
      else if (k == K_MIX2 && comp[i].mask0 != 0u) lines = 8.f * G * wcm;
      else if (k == K_MIX2_BITS) lines = 8.f * 8 * wcm;
      else if (k == K_SSE) lines = 8.f * G * wcm;
      else if (k == K_SSE_BITS) lines = 8.f * 8 * wcm;
    }
    const bool bits = k >= K_CM_BITS;                                   // a quarter of the group's blocks per wavefront
    float sl = (k == K_CODER ? p_l : p_l + ctx_l) + (k == K_MIX2 || k == K_MIX2_BITS || k == K_AVG ? 2.f * p_l : (k == K_SSE || k == K_SSE_BITS ? p_l : 0.f));
    if (bits) sl *= 8.f * 8.f / (float)(G * 8);
    if (k == K_CODER && L.ps_coder_fast) lds = 4096 * 4;                 // (its table of probabilities: pipe_coder_fast)
    add(2, code_role, L.light_sub[r], k == K_CODER ? coder_unit : p_unit[i], lds, lds && k != K_CODER ? 0.9f : lc[k < 12 ? k : 0], lines, sl);
  }
  // (a small chain: the whole stretch table behind the ICM's side table, device PipeStretchFull)
  for (size_t r = 0; r < L.icm.size(); ++r) add(3, (int)r, 0, p_unit[L.icm[r]], 256 * G * 4 + (small && L.ps_icm_full ? 65536 : 0), 1.1f, 0.f, bh_l + p_l);
  // (packed pairs: pipe_isse_packed_unit; a small chain: two words per pair and the whole squash table, pipe_isse_unit)
  for (size_t r = 0; r < L.isse.size(); ++r) add(4, (int)r, 0, p_unit[L.isse[r]], small ? 512 * G * 4 + 8192 : 256 * G * 4 + 64 * G * 4, 1.5f, 0.f, bh_l + 2.f * p_l);
  // (a wavefront of the persistent launch is 64 lanes wide whatever the group size: a MIX unit's lane groups fill it --
  //  64 / QL blocks per wavefront, not the G / QL of the step kernels' G-thread workgroups)
  // throughput shape: a MIX whose 8 rows of a byte are distinct splits the byte over two lane groups (device: pipe_mix_unit NH = 2)
  L.ps_mix_nh = 1;
  // (round 5, call 12: no faster than one lane group per block.  Round 6, after the other units had caught up: the two MIX units
  //  were the longest streams of the launch -- 2.5-2.9 s busy of 3.0 -- and in halves the -m5 headline goes from 349.7 to 369.2 MB/s
  //  (profiles/r06 call 3; 8 more wavefronts per group: 62 of the 64 slots of its 8 workgroups).  ZPAQ_AMD_MIX_HALVES=0: one group.)
  if (halves && !L.mix_bits && !L.mix.empty()) {
    bool ok = true;
    for (size_t r = 0; r < L.mix.size(); ++r) {
      const CompDesc& c = comp[L.mix[r]];
      ok = ok && c.a5 == 255u && c.mask0 >= 255u && L.mix_ql[r] * 2 <= 64;
    }
    if (ok) L.ps_mix_nh = 2;
  }
  L.mix_lds_rows.assign(L.mix.size(), 0);
  for (size_t r = 0; r < L.mix.size(); ++r) {
    const int nw = L.mix_bits ? L.mix_waves_of(L.mix_ql[r]) : std::max(1, G * L.mix_ql[r] * L.ps_mix_nh / 64);
    // packed rows of a small table: the first rows in the LDS of every wavefront of the unit (3 dwords per weight quad, row and
    // block of the wavefront: pipe_mix_packed_unit)
    int lds = 0;
    const CompDesc& c = comp[L.mix[r]];
    if (!L.mix_bits && L.mix_packed[r] && c.mask0 + 1u <= 256u && lds_rows_wish > 0) {
      const int rows = std::min<int>(lds_rows_wish, (int)(c.mask0 + 1u));
      const int nq = ((int)c.a3 + 3) / 4, bpw = std::max(1, std::min(G, 64 / (L.mix_ql[r] * L.ps_mix_nh)));
      L.mix_lds_rows[r] = rows;
      lds = 3 * rows * nq * bpw * 4;
    }
    // a lane group touches one row per bit it codes: (lane groups per wavefront) x (bits per lane group) lines per byte
    const float groups_per_wave = 64.f / (float)L.mix_ql[r];
    float lines = groups_per_wave * (L.mix_bits ? 1.f : 8.f / (float)L.ps_mix_nh) * line_weight(4ull * c.stride * (c.mask0 + 1ull));
    if (L.mix_lds_rows[r]) lines *= 1.f / 8.f;
    for (int sub = 0; sub < nw; ++sub)
      // (every input stream: one line per wavefront and byte, whatever part of it the wavefront's blocks are)
      add(5, (int)r, sub, p_unit[L.mix[r]], lds, L.mix_bits ? 0.55f : (L.ps_mix_nh == 2 ? 1.7f : (lds ? 1.6f : 2.6f)), lines, (float)c.a3 + ctx_l);
  }
  // who reads whose streams
  std::vector<std::vector<int>> producers(nunit);
  auto reads = [&](int u, int v) { if (u != v && std::find(producers[u].begin(), producers[u].end(), v) == producers[u].end()) producers[u].push_back(v); };
  for (int i = 0; i < n; ++i) {
    const CompDesc& c = comp[i];
    const int u = p_unit[i];
    if (L.row[i] >= 0) { reads(row_unit[i], 0); reads(u, row_unit[i]); }       // ROW unit: contexts; the map: bit histories
    else if (L.ctx[i] >= 0) reads(u, 0);
    switch (c.type) {
      case C_ISSE: reads(u, p_unit[c.a2]); break;
      case C_AVG: reads(u, p_unit[c.a1]); reads(u, p_unit[c.a2]); break;
      case C_MIX2: reads(u, p_unit[c.a2]); reads(u, p_unit[c.a3]); break;
      case C_SSE: reads(u, p_unit[c.a2]); break;
      case C_MIX: for (unsigned t = 0; t < c.a3; ++t) reads(u, p_unit[c.a2 + t]); break;
      default: break;
    }
  }
  reads(coder_unit, p_unit[n - 1]);
  reads(coder_unit, 0);                                  // HCOMP's status word
  std::vector<std::vector<PipeLayout::Dep>> unit_deps(nunit);
  for (int u = 0; u < nunit; ++u)
    for (int v : producers[u]) {
      unit_deps[u].push_back({v, 0, unit_waves[v]});        // v has finished the chunk
      unit_deps[v].push_back({u, L.S, unit_waves[u]});      // u is done with the ring slot v is about to overwrite
    }
  // A unit of several wavefronts shares ONE counter: its wavefronts start chunk c together (counter >= waves * c), so that a
  // sum of waves * (c + 1) can only be made of everybody's chunk c (a wavefront that ran ahead would otherwise stand in for a late one)
  for (int u = 0; u < nunit; ++u)
    if (unit_waves[u] > 1) unit_deps[u].push_back({u, 1, unit_waves[u]});
  // pack: wavefronts per workgroup W (<= 8: the MIX units' ~180 VGPRs allow two wavefronts per SIMD), LDS per workgroup
  const int total = (int)slots.size();
  const int cap = kPersistLdsCap - kPersistRoBytes;
  for (const auto& s : slots) if (s.lds > cap) { L.persist_why = "a unit's tables do not fit a workgroup's LDS"; return; }
  // Workgroups per group.  A chain of the standard size (-m5: 54 unit wavefronts) is packed into 8 even where 7 would do: 32
  // groups x 8 = the MI355X's 256 compute units, one workgroup each (measured, profiles/r05 call14: 346 MB/s against 326 with
  // 7 x 32 = 224 workgroups); otherwise the fewest that hold the units.  The tables go first (largest first, each into the
  // bin it fits best -- tried emptiest-first as well), then the units without tables to the bin with the least work.
  static const int wpg_floor = [] { const char* v = getenv("ZPAQ_AMD_PERSIST_WPG_MIN"); return v ? atoi(v) : -1; }();      // (experiments)
  // (a wavefront per SIMD while that takes at most 8 workgroups per group; the larger chains of the latency shape keep 8 per workgroup)
  static const int small_w4 = [] { const char* v = getenv("ZPAQ_AMD_SMALL_CHAIN_W4_WAVES"); return v ? atoi(v) : 32; }();
  // (experiment of call 29: a wavefront per SIMD for EVERY latency-shape chain, without the small chains' LDS-rich units)
  static const bool latency_w4 = [] { const char* v = getenv("ZPAQ_AMD_LATENCY_W4"); return v && v[0] == '1'; }();
  // Variant 3 (round 6, profiles/r06 call 29): the latency shape of the LARGER chains with a wavefront per SIMD as well -- -m5 on
  // 64 / 128 / 256 blocks 40.2 / 78.8 / 141.0 -> 47.1 / 94.2 / 177.6 MB/s -- at twice the workgroups per group (28 for -m5), so the
  // engine takes it while those fit the device (up to 9 groups) and variant 1 beyond (up to 18).
  const int Wmax = (small && small_waves <= small_w4) || ((latency_w4 || L.ps_wide) && L.mode == 1) ? 4 : 8;
  const int wpg_min = (total + Wmax - 1) / Wmax;
  std::vector<int> tries;
  if (wpg_floor >= 0) { for (int w = std::max(wpg_min, std::min(wpg_floor, total)); w <= total; ++w) tries.push_back(w); }
  else {
    if (total >= 40 && wpg_min < 8) tries.push_back(8);
    for (int w = wpg_min; w <= total; ++w) tries.push_back(w);
  }
  // Round 6: a workgroup is a compute unit, and what paces a compute unit's wavefronts is how many memory lines they have in
  // flight together (per-unit profile, profiles/r06 call 2: the workgroup of a group with 288 table lines per byte had every
  // wavefront at 2.5-2.8 s, those with 160 at 1.6-1.9 s, whatever the units were; the decoder's compute units and the encoder's
  // both run at ~90 lines per microsecond, which x 256 is the machine's random-access rate of profiles/r03/gups.hip).  So the
  // packing balances LINES: every unit, heaviest first, goes to the workgroup with the fewest lines so far that still has a
  // wavefront slot and the LDS for its tables (units without any line -- none once stream lines count -- would follow by LDS fit
  // and cost).  -m5 headline: 355-367 -> 401-412 MB/s on one box
  // (calls 6, 8).  A least-squares fit of per-unit loads from nine profiled launches with a randomised min-max search
  // (profiles/r06/fit_packing.py; call 8) balanced its own model to 2 % and ran 5 % slower than this: what it cannot see is
  // who shares a SIMD with whom.  Tried for the workgroup count the table-first packing below needs; when it does not fit,
  // that packing stands.
  static const bool balance_lines = [] { const char* v = getenv("ZPAQ_AMD_PACK_LINES"); return !(v && v[0] == '0'); }();
  auto finish = [&](int wpg, int W, const std::vector<std::vector<int>>& bin_items) {
    L.ps_wpg = wpg; L.ps_waves = W;
    int max_lds = 0;
    L.ps_slots.assign((size_t)wpg * W, PipeLayout::Slot());
    L.ps_deps.assign((size_t)wpg * W, {});
    for (int b = 0; b < wpg; ++b) {
      // wavefront w runs on SIMD w % 4: the four heaviest units first, then the next four against them
      std::vector<int> v = bin_items[b];
      std::stable_sort(v.begin(), v.end(), [&](int x, int y) { return slots[x].cost > slots[y].cost; });
      std::vector<int> at(W, -1);
      if (W == 8) {
        // two wavefronts per SIMD (w and w + 4): the units sorted by cost, the k-th heaviest shares its SIMD with the k-th lightest
        // (an idle slot counts as the lightest)
        std::vector<int> padded = v;
        padded.resize(8, -1);
        // (measured against "k-th with (k + 4)-th", profiles/r06 call 10, three alternating pairs of runs: 407 against 403 MB/s -- no difference)
        for (int k = 0; k < 4; ++k) { at[k] = padded[k]; at[4 + k] = padded[7 - k]; }
      } else {
        for (int k = 0; k < (int)v.size(); ++k) at[k] = v[k];
      }
      // (a wavefront without a unit exits at once: the unit that would have shared its SIMD has it to itself)
      int off = (kPersistRoBytes + 255) & ~255;
      for (int w = 0; w < W; ++w) {
        if (at[w] < 0) continue;
        PipeLayout::Slot s = slots[at[w]];
        s.lds_off = off; off += s.lds;
        L.ps_slots[(size_t)b * W + w] = s;
        L.ps_deps[(size_t)b * W + w] = unit_deps[s.unit];
      }
      max_lds = std::max(max_lds, off);
    }
    L.ps_lds_bytes = max_lds;
    L.persist_ok = true;
  };
  auto pack_by_lines = [&](int wpg, int W, std::vector<std::vector<int>>& out) -> bool {
    struct Bin { std::vector<int> s; int lds = 0; float cost = 0, lines = 0; };
    std::vector<Bin> bins(wpg);
    std::vector<int> order;
    for (int i = 0; i < total; ++i) if (slots[i].lines > 0.f) order.push_back(i);
    const int nline = (int)order.size(), nrest = total - nline;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return slots[x].lines > slots[y].lines; });
    // every workgroup keeps room for its share of the units without lines (they carry the LDS tables)
    const int keep = std::min(W - 1, (nrest + wpg - 1) / wpg);
    for (int idx : order) {
      int best = -1;
      for (int b = 0; b < wpg; ++b) {
        if ((int)bins[b].s.size() >= W - keep || bins[b].lds + slots[idx].lds > cap) continue;
        if (best < 0 || bins[b].lines < bins[best].lines || (bins[b].lines == bins[best].lines && bins[b].cost < bins[best].cost)) best = b;
      }
      if (best < 0) {       // no room under the reserve: any workgroup with a free wavefront
        for (int b = 0; b < wpg; ++b)
          if ((int)bins[b].s.size() < W && bins[b].lds + slots[idx].lds <= cap && (best < 0 || bins[b].lines < bins[best].lines)) best = b;
      }
      if (best < 0) return false;
      bins[best].s.push_back(idx); bins[best].lds += slots[idx].lds; bins[best].cost += slots[idx].cost; bins[best].lines += slots[idx].lines;
    }
    std::vector<int> rest;
    for (int i = 0; i < total; ++i) if (!(slots[i].lines > 0.f)) rest.push_back(i);
    std::stable_sort(rest.begin(), rest.end(), [&](int x, int y) {
      if (slots[x].lds != slots[y].lds) return slots[x].lds > slots[y].lds;
      return slots[x].cost > slots[y].cost;
    });
    for (int idx : rest) {
      const auto& s = slots[idx];
      int best = -1;
      for (int b = 0; b < wpg; ++b) {
        if ((int)bins[b].s.size() >= W || bins[b].lds + s.lds > cap) continue;
        if (best < 0) { best = b; continue; }
        if (s.lds) {
          // a table: the workgroup with the most free wavefront slots, then the emptiest LDS (the tables have to spread: a workgroup
          // with few free wavefronts cannot take many)
          const int fb = W - (int)bins[b].s.size(), fbest = W - (int)bins[best].s.size();
          if (fb > fbest || (fb == fbest && bins[b].lds < bins[best].lds)) best = b;
        } else if (bins[b].cost < bins[best].cost) best = b;
      }
      if (best < 0) return false;
      bins[best].s.push_back(idx); bins[best].lds += s.lds; bins[best].cost += s.cost;
    }
    out.clear();
    for (auto& b : bins) out.push_back(b.s);
    return true;
  };
  for (size_t ti = 0; ti < tries.size() * 2; ++ti) {
    const int wpg = tries[ti / 2];
    const bool best_fit = (ti & 1) == 1;          // (spread the tables when that works: the units that own them are LDS-bound together)
    const int W = std::min(Wmax, total);
    struct Bin { std::vector<int> s; int lds = 0; float cost = 0; };
    std::vector<Bin> bins(wpg);
    std::vector<int> order(total);
    for (int i = 0; i < total; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
      if (slots[x].lds != slots[y].lds) return slots[x].lds > slots[y].lds;
      return slots[x].cost > slots[y].cost;
    });
    int with_tables = 0;
    for (const auto& sl : slots) with_tables += sl.lds != 0;
    bool fits = true;
    for (int idx : order) {
      const auto& s = slots[idx];
      int best = -1;
      for (int b = 0; b < wpg; ++b) {
        if ((int)bins[b].s.size() >= W || bins[b].lds + s.lds > cap) continue;
        if (best < 0) { best = b; continue; }
        if (s.lds) {
          // a table: the fullest bin it still fits (best fit) while every bin can still get its share of wavefronts, or the emptiest
          const bool fuller = bins[b].lds > bins[best].lds;
          if (best_fit ? (fuller && (int)bins[b].s.size() < (with_tables + wpg - 1) / wpg + 1) : !fuller && bins[b].lds != bins[best].lds) best = b;
        } else if (bins[b].cost < bins[best].cost) best = b;
      }
      if (best < 0) { fits = false; break; }
      bins[best].s.push_back(idx); bins[best].lds += s.lds; bins[best].cost += s.cost;
    }
    if (!fits) continue;
    std::vector<std::vector<int>> items;
    if (!(balance_lines && W == 8 && pack_by_lines(wpg, W, items))) {
      items.clear();
      for (auto& b : bins) items.push_back(b.s);
    }
    finish(wpg, W, items);
    return;
  }
  L.persist_why = "the units of a group cannot be packed into workgroups";
}

static bool pipe_layout_compute(const zpq_plan& plan, const PipeOptions& opt, PipeLayout& L, std::string& why_not);

// The layout of a (header, options) pair is asked for many times per batch (mode choice, buffer sizes, launch geometry):
// computed once per process and remembered.
bool pipe_layout(const zpq_plan& plan, const PipeOptions& opt, PipeLayout& L, std::string& why_not) {
  struct Memo { bool ok; PipeLayout L; std::string why; };
  static std::mutex mu;
  static std::map<std::string, Memo> memo;
  std::string key((const char*)plan.header.data(), plan.header.size());
  key += '|'; key += std::to_string(opt.mode); key += ','; key += std::to_string(opt.chunk); key += ','; key += std::to_string(opt.group);
  key += ','; key += opt.persist ? '1' : '0';
  key += opt.wide ? 'w' : 'n';
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = memo.find(key);
    if (it != memo.end()) { L = it->second.L; why_not = it->second.why; return it->second.ok; }
  }
  Memo m;
  m.ok = pipe_layout_compute(plan, opt, m.L, m.why);
  L = m.L; why_not = m.why;
  std::lock_guard<std::mutex> g(mu);
  if (memo.size() > 4096) memo.clear();
  memo[key] = m;
  return m.ok;
}

static bool pipe_layout_compute(const zpq_plan& plan, const PipeOptions& opt, PipeLayout& L, std::string& why_not) {
  const PlanHeader& ph = plan.hdr();
  const int n = (int)ph.n;
  if (n < 1 || n > 64) { why_not = "more than 64 components"; return false; }
  if (ph.arena_bytes >= (1ull << 32)) { why_not = "model state of 4 GiB or more per block"; return false; }
  const CompDesc* comp = plan.comps();
  L = PipeLayout();
  L.n = n;
  L.mode = opt.mode ? 1 : 0;
  L.ps_wide = opt.wide && L.mode == 1;
  if (opt.chunk >= 64 && opt.chunk <= 8192 && (opt.chunk & (opt.chunk - 1)) == 0) L.C = opt.chunk;
  if (opt.group == 8 || opt.group == 16 || opt.group == 32 || opt.group == 64) L.G = opt.group;
  enum { K_ROW = 1, K_CONS, K_CM, K_MATCH, K_AVG, K_MIX2, K_SSE, K_CODER };     // = device PipeKind
  bool mix_bits_ok = true;
  L.light_bits = L.mode ? 7 : 4;           // 1 CM | 2 MIX2 | 4 SSE
  enum { K_CM_BITS = 9, K_MIX2_BITS, K_SSE_BITS };
  // a light unit: one workgroup per group, or -- a lane per bit position -- G / 8 of them (64 lanes = 8 blocks x 8 positions)
  auto light_unit = [&](int kind, int bits_kind, bool bits_ok, int i) {
    if (bits_ok && (L.light_bits >> (bits_kind - K_CM_BITS) & 1)) { for (int sub = 0; sub < L.G * 8 / L.light_threads(); ++sub) { L.light.push_back({bits_kind, i}); L.light_sub.push_back(sub); } }
    else { L.light.push_back({kind, i}); L.light_sub.push_back(0); }
  };
  for (int i = 0; i < n; ++i) {
    const CompDesc& c = comp[i];
    L.ctx[i] = L.row[i] = L.state[i] = -1;
    int lv = 1;
    switch (c.type) {
      case C_CONS: light_unit(K_CONS, K_CM_BITS, false, i); break;
      case C_CM: L.ctx[i] = L.nctx++; light_unit(K_CM, K_CM_BITS, c.mask0 >= 511u, i); break;
      case C_MATCH: L.ctx[i] = L.nctx++; L.state[i] = L.nstate; L.nstate += 12; light_unit(K_MATCH, K_CM_BITS, false, i); break;
      case C_ICM: L.ctx[i] = L.nctx++; L.row[i] = L.nrow++; lv = 2; L.icm.push_back(i); break;
      case C_ISSE: L.ctx[i] = L.nctx++; L.row[i] = L.nrow++; lv = std::max(2, L.level[c.a2] + 1); L.isse.push_back(i); break;
      case C_AVG: lv = std::max(L.level[c.a1], L.level[c.a2]) + 1; light_unit(K_AVG, K_CM_BITS, false, i); break;
      case C_MIX2: L.ctx[i] = L.nctx++; lv = std::max(L.level[c.a2], L.level[c.a3]) + 1;
        light_unit(K_MIX2, K_MIX2_BITS, c.a5 == 255u && c.mask0 >= 255u, i); break;
      case C_SSE: L.ctx[i] = L.nctx++; lv = L.level[c.a2] + 1; light_unit(K_SSE, K_SSE_BITS, c.mask0 >= 32u * 256u - 1u, i); break;
      case C_MIX: {
        if (c.a3 > 64) { why_not = "MIX with more than 64 inputs"; return false; }
        L.ctx[i] = L.nctx++;
        for (unsigned t = 0; t < c.a3; ++t) lv = std::max(lv, L.level[c.a2 + t] + 1);
        const int m = (int)c.a3, nq = (m + 3) / 4;     // lanes that hold weights: 4 per lane, one 16-byte access
        int ql = 1;
        while (ql < nq) ql *= 2;
        if (ql > L.G) { why_not = "MIX lane group wider than the block group"; return false; }
        L.mix.push_back(i);
        L.mix_ql.push_back(ql);
        // a lane per bit position: the 8 rows of a byte must be distinct and a block's 8 x ql lanes fit one wavefront
        mix_bits_ok = mix_bits_ok && c.a5 == 255u && c.mask0 >= 255u && ql <= 8 && L.G * ql % 8 == 0;
        break;
      }
      default: why_not = "unknown component type"; return false;
    }
    L.level[i] = lv;
  }
  if (L.mode && mix_bits_ok && !L.mix.empty()) L.mix_bits = 1;
  // ROW units (level 1) have a kernel of their own; the light kernel: the components in COMP order, then the coder
  for (int i = 0; i < n; ++i) if (L.row[i] >= 0) L.rows.push_back(i);
  light_unit(K_CODER, K_CM_BITS, false, n - 1);
  L.coder_level = L.level[n - 1] + 1;
  L.S = L.coder_level + 1 + L.slack;
  {
    // which kernel produces the p stream of component i, and who reads what
    auto kernel_of = [&](int i) { const unsigned t = comp[i].type; return t == C_ICM ? 3 : (t == C_ISSE ? 4 : (t == C_MIX ? 5 : 2)); };
    auto reads = [&](int consumer_kernel, int j) { const int pk = kernel_of(j); if (pk != consumer_kernel) L.consumes[consumer_kernel][pk] = true; };
    for (int i = 0; i < n; ++i) {
      const CompDesc& c = comp[i];
      const int k = kernel_of(i);
      if (L.ctx[i] >= 0 && c.type != C_ICM && c.type != C_ISSE) L.consumes[k][0] = true;     // contexts from HCOMP
      if (L.row[i] >= 0) { L.consumes[1][0] = true; L.consumes[k][1] = true; }                 // ROW unit reads ctx, the map reads bh
      switch (c.type) {
        case C_ISSE: reads(k, (int)c.a2); break;
        case C_AVG: reads(k, (int)c.a1); reads(k, (int)c.a2); break;
        case C_MIX2: reads(k, (int)c.a2); reads(k, (int)c.a3); break;
        case C_SSE: reads(k, (int)c.a2); break;
        case C_MIX: for (unsigned t = 0; t < c.a3; ++t) reads(k, (int)(c.a2 + t)); break;
        default: break;
      }
    }
    reads(2, n - 1);                       // the coder (light kernel) reads the last component
    L.consumes[2][0] = true;               // ... and HCOMP's status word
  }
  L.hcomp_state = L.nstate; L.nstate += 8;
  L.coder_state = L.nstate; L.nstate += 4;
  const uint64_t hbytes = 4ull * (ph.hmask + 1);
  L.hcomp_h_lds = hbytes <= 8192;
  L.hcomp_lanes = 64;
  if (L.hcomp_h_lds) while (L.hcomp_lanes > 8 && hbytes * L.hcomp_lanes > 65536) L.hcomp_lanes /= 2;
  const uint64_t C = (uint64_t)L.C, S = (uint64_t)L.S, G = (uint64_t)L.G;
  uint64_t off = 0;
  L.off_ctx = off;   off += S * (uint64_t)std::max(L.nctx, 1) * C * G * 4;
  L.off_bh = off;    off += S * (uint64_t)std::max(L.nrow, 1) * C * G * 8;
  L.off_p = off;     off += S * (uint64_t)n * C * G * 16;
  L.off_state = off; off += (uint64_t)L.nstate * G * 4;
  L.group_bytes = (off + 4095) & ~4095ull;
  if (L.group_bytes >= (1ull << 32)) { why_not = "stream buffer of a block group exceeds 4 GiB"; return false; }
  // packed MIX rows (device/pipe_kernel.h): a mixer whose rows are distinct per bit, with 3 .. 64 inputs, whose packed row is
  // smaller than its padded one or whose table is small enough to be worth keeping in the LDS; not with a lane per bit position
  L.mix_packed.assign(L.mix.size(), 0);
  // OFF by default -- built, bit-exact, measured, slower (profiles/r06 calls 1 and 3, -m5 headline): 286 MB/s against 356 with one lane
  // group per block, 317 against 369 in halves, and the same 317 whether 0, 64 or 128 rows of m8 live in the LDS.  The units' own
  // instruction streams grew (619 -> 811 / 1209 instructions per byte) and became the longest of the launch, while the requests
  // they saved were the cheap ones (profiles/r06/gups2.hip: a row touch that stays on the die costs about half of one that
  // goes to HBM, and a launch at 0.8 of the machine's request rate is not paced by requests alone).  ZPAQ_AMD_MIX_PACKED=1 builds it.
  static const bool packed_off = [] { const char* v = getenv("ZPAQ_AMD_MIX_PACKED"); return !(v && v[0] == '1'); }();
  for (size_t r = 0; r < L.mix.size(); ++r) {
    const CompDesc& c = comp[L.mix[r]];
    const uint32_t ps = mix_pstride(c.a3);
    if (!packed_off && !L.mix_bits && c.a5 == 255u && c.mask0 >= 255u && c.a3 >= 3u && ps <= 4u * c.stride && (ps < 4u * c.stride || c.mask0 + 1u <= 256u))
      L.mix_packed[r] = 1;
  }
  // the persistent launch: with as many of a small packed table's rows in the LDS as still pack into the same number of
  // workgroups per group as none would (a ninth workgroup per group would cost the headline its one-round residency)
  static const int rows_forced = [] { const char* v = getenv("ZPAQ_AMD_MIX_LDS_ROWS"); return v ? atoi(v) : -1; }();
  // MIX units in halves (two lane groups per block: twice the wavefronts, half the stream each) when that costs no extra
  // workgroup per group -- a ninth would cost a 1024-block batch its one-round residency
  static const bool halves_wanted = [] { const char* v = getenv("ZPAQ_AMD_MIX_HALVES"); return !(v && v[0] == '0'); }();
  plan_persistent(plan, L, 0, false);
  bool halves = false;
  if (L.persist_ok && halves_wanted) {
    PipeLayout T = L;
    plan_persistent(plan, T, 0, true);
    if (T.persist_ok && T.ps_mix_nh == 2 && T.ps_wpg <= L.ps_wpg) { L = T; halves = true; }
  }
  if (L.persist_ok) {
    const int wpg0 = L.ps_wpg;
    for (int wish : {128, 64, 32}) {
      if (rows_forced >= 0 && wish != rows_forced) continue;
      PipeLayout T = L;
      plan_persistent(plan, T, wish, halves);
      bool any = false;
      for (int v : T.mix_lds_rows) any = any || v != 0;
      if (!any) break;
      if (T.persist_ok && T.ps_wpg <= wpg0) { L = T; break; }
    }
  }
  return true;
}

bool generate_pipe_source(const zpq_plan& plan, const PipeOptions& opt, std::string& source, std::string& why_not) {
  PipeLayout L;
  if (!pipe_layout(plan, opt, L, why_not)) return false;
  const PlanHeader& ph = plan.hdr();
  const int n = L.n;
  const CompDesc* comp = plan.comps();
  std::ostringstream o;
  o << "// generated by zpaq_amd codegen v" << kCodegenVersion << " (pipelined encoder) -- do not edit\n"
       "#define ZPQ_PERSIST_LDS_BYTES " << (L.persist_ok && opt.persist && opt.chunk != 2048 ? L.ps_lds_bytes : 16) << "\n"
       "#include \"pipe_persist.h\"\n"
       "namespace zpq_gen {\n"
       "struct Chain {\n";
  int nmix = 0, nsse = 0;
  std::ostringstream comps;
  for (int i = 0; i < n; ++i) {
    const CompDesc& c = comp[i];
    int slot = -1;
    if (c.type == C_MIX) slot = nmix++;
    if (c.type == C_SSE) slot = nsse++;
    comps << "    {" << c.type << "u," << c.a1 << "u," << c.a2 << "u," << c.a3 << "u," << c.a4 << "u," << c.a5 << "u, "
          << c.limit << "u," << c.mask0 << "u," << c.mask1 << "u, " << c.t0 << "ull," << c.t1 << "ull, -1," << slot << "," << c.stride << "u},\n";
  }
  auto arr = [&](const char* name, const int* v, int cnt) {
    o << "  static constexpr int " << name << "[" << std::max(cnt, 1) << "] = {";
    for (int i = 0; i < std::max(cnt, 1); ++i) o << (i ? "," : "") << (i < cnt ? v[i] : 0);
    o << "};\n";
  };
  o << "  static constexpr int N = " << n << ", NMIX = " << nmix << ", NSSE = " << nsse << ";\n"
    << "  static constexpr unsigned HMASK = " << ph.hmask << "u, MMASK = " << ph.mmask << "u;\n"
    << "  static constexpr unsigned long long OFF_RUN = " << ph.off_run << "ull;\n"
    << "  static constexpr unsigned long long OFF_H = " << ph.off_H << "ull, OFF_M = " << ph.off_M
    << "ull, OFF_R = " << ph.off_R << "ull;\n"
    << "  static constexpr zpq::CompK comp[N] = {\n" << comps.str() << "  };\n"
    << "  static constexpr unsigned PIPE_G = " << L.G << "u;\n"
    << "  static constexpr int PIPE_C = " << L.C << ", PIPE_S = " << L.S << ", PIPE_NCTX = " << std::max(L.nctx, 1)
    << ", PIPE_NROW = " << std::max(L.nrow, 1) << ";\n"
    << "  static constexpr unsigned long long PIPE_OFF_CTX = " << L.off_ctx << "ull, PIPE_OFF_BH = " << L.off_bh
    << "ull, PIPE_OFF_P = " << L.off_p << "ull, PIPE_OFF_STATE = " << L.off_state << "ull, PIPE_GROUP_BYTES = "
    << L.group_bytes << "ull;\n"
    << "  static constexpr int CODER_LEVEL = " << L.coder_level << ", CODER_STATE = " << L.coder_state
    << ", HCOMP_STATE = " << L.hcomp_state << ", HCOMP_LANES = " << L.hcomp_lanes << ";\n"
    << "  static constexpr bool HCOMP_H_LDS = " << (L.hcomp_h_lds ? "true" : "false") << ";\n";
  arr("P_LEVEL", L.level, n);
  arr("P_CTX", L.ctx, n);
  arr("P_ROW", L.row, n);
  arr("P_STATE", L.state, n);
  std::vector<int> lk, lc, mf;
  for (auto& r : L.light) { lk.push_back(r.first); lc.push_back(r.second); }
  int first = 0;
  // MIX_FIRST: lane groups of earlier MIX roles (x MIX_SPLIT = wavefronts per group); with bit lanes: their wavefronts per group
  for (int q : L.mix_ql) { mf.push_back(first); first += L.mix_waves_of(q); }
  o << "  static constexpr int PIPE_MODE = " << L.mode << ", MIX_BITS = " << L.mix_bits << ", MIX_DEPTH = " << L.depth << ";\n";
  o << "  static constexpr int NROWU = " << L.rows.size() << ", NLIGHT = " << L.light.size() << ", NICM = " << L.icm.size() << ", NISSE = " << L.isse.size()
    << ", NMIXR = " << L.mix.size() << ";\n";
  arr("LIGHT_KIND", lk.data(), (int)lk.size());
  arr("LIGHT_COMP", lc.data(), (int)lc.size());
  arr("LIGHT_SUB", L.light_sub.data(), (int)L.light_sub.size());
  o << "  static constexpr int LIGHT_THREADS = " << L.light_threads() << ", LIGHT_DEPTH = " << L.depth << ";\n";
  arr("ROW_COMP", L.rows.data(), (int)L.rows.size());
  arr("ICM_COMP", L.icm.data(), (int)L.icm.size());
  arr("ISSE_COMP", L.isse.data(), (int)L.isse.size());
  arr("MIX_COMP", L.mix.data(), (int)L.mix.size());
  arr("MIX_QL", L.mix_ql.data(), (int)L.mix_ql.size());
  arr("MIX_FIRST", mf.data(), (int)mf.size());
  arr("MIX_PACKED", L.mix_packed.data(), (int)L.mix_packed.size());
  const U8* prog = plan.blob.data() + ph.off_prog;
  if (!translate_hcomp(prog, (int)ph.prog_len, o)) { why_not = "HCOMP program too irregular to translate"; return false; }
  o << "};\n";
  if (L.persist_ok && opt.persist && opt.chunk != 2048) {
    // the persistent launch: the same chain, its units packed into workgroups (device/pipe_persist.h)
    std::vector<int> kind, role, sub, unit, ldso, dep0, ndep, du, dl, dm;
    for (size_t i = 0; i < L.ps_slots.size(); ++i) {
      const PipeLayout::Slot& sl = L.ps_slots[i];
      kind.push_back(sl.kind); role.push_back(sl.role); sub.push_back(sl.sub); unit.push_back(sl.unit); ldso.push_back(sl.lds_off);
      dep0.push_back((int)du.size()); ndep.push_back(sl.kind < 0 ? 0 : (int)L.ps_deps[i].size());
      if (sl.kind >= 0) for (const PipeLayout::Dep& d : L.ps_deps[i]) { du.push_back(d.unit); dl.push_back(d.lag); dm.push_back(d.mult); }
    }
    if (du.empty()) { du.push_back(0); dl.push_back(0); dm.push_back(0); }
    o << "struct ChainP : Chain {\n"
         "  static constexpr bool PIPE_PERSIST = true;\n"
         "  static constexpr int PS_MIX_NH = " << L.ps_mix_nh << ";\n"
         "  static constexpr bool HCOMP_M_LDS = " << (L.hcomp_m_lds ? "true" : "false") << ";\n"
         "  static constexpr int PS_AHEAD = " << L.ps_ahead << ";\n"
         "  static constexpr bool PS_ROW_RING = " << (L.ps_row_ring ? "true" : "false") << ";\n"
         "  static constexpr bool PS_CODER_FAST = " << (L.ps_coder_fast ? "true" : "false") << ", PS_SMALL = " << (L.ps_small ? "true" : "false") << ";\n"
      << (L.ps_small && !L.ps_row_halves ? "  static constexpr bool PS_ROW_HALVES = false;\n" : "")
      << (L.ps_small && !L.ps_icm_full ? "  static constexpr bool PS_ICM_FULL = false;\n" : "")
      << "";
    o <<
         "  static constexpr int PS_WAVES = " << L.ps_waves << ", PS_WPG = " << L.ps_wpg << ", PS_NSLOT = " << L.ps_slots.size()
      << ", PS_NUNIT = " << L.ps_nunit << ", PS_LDS_BYTES = " << L.ps_lds_bytes << ";\n";
    arr("PS_KIND", kind.data(), (int)kind.size());
    arr("PS_ROLE", role.data(), (int)role.size());
    arr("PS_SUB", sub.data(), (int)sub.size());
    arr("PS_UNIT", unit.data(), (int)unit.size());
    arr("PS_LDS", ldso.data(), (int)ldso.size());
    arr("PS_DEP0", dep0.data(), (int)dep0.size());
    arr("PS_NDEP", ndep.data(), (int)ndep.size());
    arr("PS_DEP_UNIT", du.data(), (int)du.size());
    arr("PS_DEP_LAG", dl.data(), (int)dl.size());
    arr("PS_DEP_MULT", dm.data(), (int)dm.size());
    arr("MIX_LDS_ROWS", L.mix_lds_rows.data(), (int)L.mix_lds_rows.size());
    o << "};\n";
  }
  o << "}  // namespace zpq_gen\n";
  if (L.persist_ok && opt.persist && opt.chunk != 2048)
    o << "extern \"C\" __global__ __launch_bounds__(" << 64 * L.ps_waves << ") void zpq_pipe_persist(zpq::PipeArgs a) {\n"
         "  zpq::pipe_persist_body<zpq_gen::ChainP>(a);\n}\n";
  else
    o << "#ifdef ZPQ_EMU\nextern \"C\" void zpq_pipe_persist(zpq::PipeArgs) {}\n#endif\n";     // (the emulator's driver links against the name)
  o << "extern \"C\" __global__ __launch_bounds__(256) void zpq_pipe_repack(zpq::PipeArgs a) {\n"      // one workgroup per block, before the first launch
       "  zpq::pipe_repack_body<zpq_gen::Chain>(a);\n}\n";
  const char* names[6] = {"hcomp", "rows", "light", "icm", "isse", "mix"};
  for (int k = 0; k < 6; ++k)
    o << "extern \"C\" __global__ __launch_bounds__(64) void zpq_pipe_" << names[k] << "(zpq::PipeArgs a) {\n"   // launched with PIPE_G threads (hcomp, light, bit-lane mix: 64)
         "  ZPQ_PIPE_TRACE(a, " << k << ");\n"
         "  zpq::pipe_" << names[k] << "_body<zpq_gen::Chain>(a);\n}\n";
  source = o.str();
  return true;
}

// PCOMP on the device: the post-processing program a block carries (PostProcessor, libzpaq.cpp:2183-2241), translated
// like HCOMP and run one lane per segment (device/pcomp_kernel.h).  `code` = the PCOMP bytes without the length.
bool generate_pcomp_source(const U8* code, size_t len, int ph, int pm, std::string& source, std::string& why_not) {
  if (len < 1 || len > 65535) { why_not = "empty or oversized PCOMP"; return false; }
  if (ph > 28 || pm > 30) { why_not = "PCOMP arrays too large"; return false; }
  std::ostringstream o;
  o << "// generated by zpaq_amd codegen v" << kCodegenVersion << " (PCOMP post-processor) -- do not edit\n"
       "#include \"pcomp_kernel.h\"\n"
       "namespace zpq_gen {\n"
       "struct Post {\n"
       "  static constexpr unsigned HMASK = " << ((1u << ph) - 1u) << "u, MMASK = " << ((1u << pm) - 1u) << "u;\n";
  if (!translate_hcomp(code, (int)len, o, true)) { why_not = "PCOMP program too irregular to translate"; return false; }
  o << "};\n"
       "}  // namespace zpq_gen\n"
       "extern \"C\" __global__ __launch_bounds__(64) void zpq_pcomp_run(const zpq::PcompJob* jobs, unsigned n) {\n"
       "  zpq::pcomp_body<zpq_gen::Post>(jobs, n);\n}\n";
  source = o.str();
  return true;
}

std::string spec_cache_key(const std::string& source) {
  // key = SHA-1 of the generated text and of the kernel template it instantiates
  Sha1 s;
  s.update(source.data(), source.size());
  const U8* d = s.result();
  char hex[41];
  for (int i = 0; i < 20; ++i) snprintf(hex + 2 * i, 3, "%02x", d[i]);
  return std::string(hex, 40);
}

}  // namespace zpq
