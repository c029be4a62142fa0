// C ABI (include/zpaq_amd.h): converts internal Failure exceptions into status
// codes and implements the batched block-level drop-ins on top of the engine.
#include <cstring>
#include <map>
#include <memory>

#include "device/engine.hpp"
#include "device/plan.hpp"
#include "device/spec_loader.hpp"
#include "host/codegen.hpp"
#include "host/common.hpp"
#include "host/blocks.hpp"
#include "host/container.hpp"

namespace zpq { std::vector<U8> builtin_model(int level); }      // libzpaq_compat.cpp: the stored headers of min / mid / max.cfg

namespace zpq {
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
}  // namespace zpq

using namespace zpq;

#define ZPQ_TRY try {
#define ZPQ_CATCH                                                        \
  }                                                                      \
  catch (const Failure& f) { set_last_error(f.what()); return f.code; } \
  catch (const std::bad_alloc&) { set_last_error("Out of memory"); return ZPQ_E_NOMEM; } \
  catch (const std::exception& ex) { set_last_error(ex.what()); return ZPQ_E_DEVICE; }

extern "C" {

const char* zpq_last_error(void) { return g_last_error.c_str(); }
const char* zpq_version(void) { return "zpaq_amd 0.1 (ZPAQ level 2, libzpaq 7.15 compatible, gfx950)"; }

int zpq_init(int device) { ZPQ_TRY engine_init(device); return ZPQ_OK; ZPQ_CATCH }
int zpq_device_count(void) { return engine_device_count(); }
void zpq_shard_range(uint64_t n, uint32_t parts, uint32_t k, uint64_t* lo, uint64_t* hi) { engine_shard_range(n, parts ? parts : 1, k, lo, hi); }
void zpq_shutdown(void) { try { engine_shutdown(); } catch (...) {} }
int zpq_set_state_budget(uint64_t bytes) { engine_set_budget(bytes); return ZPQ_OK; }
int zpq_set_kernel(int which) { if (which < 0 || which > 6) return ZPQ_E_ARG; engine_set_kernel(which); return ZPQ_OK; }

int zpq_plan_create(const uint8_t* header, size_t hlen, zpq_plan** out) {
  ZPQ_TRY
  if (!out) fail(ZPQ_E_ARG, "null out");
  *out = plan_from_header(header, hlen);
  return ZPQ_OK;
  ZPQ_CATCH
}
void zpq_plan_destroy(zpq_plan* p) { if (p) { engine_plan_release(p); delete p; } }
int zpq_plan_ncomp(const zpq_plan* p) { return p ? (int)p->hdr().n : 0; }
double zpq_plan_memory(const zpq_plan* p) { return p ? p->memory : 0; }
uint64_t zpq_plan_state_bytes(const zpq_plan* p) { return p ? p->hdr().arena_bytes : 0; }
double zpq_plan_algo_bytes_per_byte(const zpq_plan* p) { return p ? p->algo_bytes : 0; }

int zpq_plan_spec_source(const zpq_plan* p, char* src, size_t cap, size_t* len, char key41[41]) {
  ZPQ_TRY
  if (!p) fail(ZPQ_E_ARG, "null plan");
  std::string source, key, why;
  const int variant = spec_variant_forced() > 0 ? spec_variant_forced() : 0;   // ZPAQ_AMD_SPEC_WAVES selects the shape
  if (!spec_source_and_key(*p, variant, source, key, why)) fail(ZPQ_E_UNSUPPORTED, why);
  if (len) *len = source.size();
  if (key41) { memcpy(key41, key.c_str(), 40); key41[40] = 0; }
  if (source.size() + 1 > cap) fail(ZPQ_E_OVERFLOW, "source buffer too small");
  memcpy(src, source.c_str(), source.size() + 1);
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_plan_spec_dual_source(const zpq_plan* p, char* src, size_t cap, size_t* len, char key41[41]) {
  ZPQ_TRY
  if (!p) fail(ZPQ_E_ARG, "null plan");
  std::string source, key, why;
  if (!spec_source_and_key(*p, 2, source, key, why)) fail(ZPQ_E_UNSUPPORTED, why);
  if (len) *len = source.size();
  if (key41) { memcpy(key41, key.c_str(), 40); key41[40] = 0; }
  if (source.size() + 1 > cap) fail(ZPQ_E_OVERFLOW, "source buffer too small");
  memcpy(src, source.c_str(), source.size() + 1);
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_plan_spec_team_source(const zpq_plan* p, char* src, size_t cap, size_t* len, char key41[41]) {
  ZPQ_TRY
  if (!p) fail(ZPQ_E_ARG, "null plan");
  std::string source, key, why;
  if (!spec_source_and_key(*p, 3, source, key, why)) fail(ZPQ_E_UNSUPPORTED, why);
  if (len) *len = source.size();
  if (key41) { memcpy(key41, key.c_str(), 40); key41[40] = 0; }
  if (source.size() + 1 > cap) fail(ZPQ_E_OVERFLOW, "source buffer too small");
  memcpy(src, source.c_str(), source.size() + 1);
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_plan_kernel_kind(zpq_plan* p, char* note, size_t cap) { return zpq_plan_kernel_kind2(p, 0, note, cap); }

int zpq_plan_kernel_kind2(zpq_plan* p, int decode, char* note, size_t cap) { return zpq_plan_kernel_kind4(p, decode, 0, 0, note, cap); }
int zpq_plan_kernel_kind3(zpq_plan* p, int decode, uint32_t nblocks, char* note, size_t cap) { return zpq_plan_kernel_kind4(p, decode, nblocks, 0, note, cap); }

int zpq_plan_kernel_kind4(zpq_plan* p, int decode, uint32_t nblocks, uint32_t block_bytes, char* note, size_t cap) {
  try {
    std::string n;
    const int k = engine_plan_kernel_kind(p, n, decode != 0, nblocks, block_bytes);
    if (note && cap) { strncpy(note, n.c_str(), cap - 1); note[cap - 1] = 0; }
    return k;
  } catch (const Failure& f) { set_last_error(f.what()); return -f.code; }
  catch (const std::exception& ex) { set_last_error(ex.what()); return -ZPQ_E_DEVICE; }
}

size_t zpq_plan_spec_jit(const zpq_plan* p, char* log, size_t cap) {
  try {
    std::string l;
    const size_t n = p ? spec_jit_compile_only(*p, spec_variant_forced() > 0 ? spec_variant_forced() : 0, l) : 0;
    if (log && cap) { strncpy(log, l.c_str(), cap - 1); log[cap - 1] = 0; }
    return n;
  } catch (const std::exception& ex) {
    if (log && cap) { strncpy(log, ex.what(), cap - 1); log[cap - 1] = 0; }
    return 0;
  }
}

size_t zpq_plan_spec_dual_jit(const zpq_plan* p, char* log, size_t cap) {
  try {
    std::string l;
    const size_t n = p ? spec_jit_compile_only(*p, 2, l) : 0;
    if (log && cap) { strncpy(log, l.c_str(), cap - 1); log[cap - 1] = 0; }
    return n;
  } catch (const std::exception& ex) {
    if (log && cap) { strncpy(log, ex.what(), cap - 1); log[cap - 1] = 0; }
    return 0;
  }
}

size_t zpq_plan_spec_team_jit(const zpq_plan* p, char* log, size_t cap) {
  try {
    std::string l;
    const size_t n = p ? spec_jit_compile_only(*p, 3, l) : 0;
    if (log && cap) { strncpy(log, l.c_str(), cap - 1); log[cap - 1] = 0; }
    return n;
  } catch (const std::exception& ex) {
    if (log && cap) { strncpy(log, ex.what(), cap - 1); log[cap - 1] = 0; }
    return 0;
  }
}

int zpq_pcomp_is_translated(const uint8_t* code, size_t codelen, int ph, int pm) {
  return code && pcomp_is_translated(code, codelen, ph, pm) ? 1 : 0;
}

int zpq_precompile(const zpq_plan* const* plans, size_t n, int decode, int threads) {
  try {
    if (!plans && n) fail(ZPQ_E_ARG, "null plans");
    std::vector<const zpq_plan*> v(plans, plans + n);
    const int forced = spec_variant_forced();
    std::string log;
    const int done = spec_precompile(v, decode == 0, forced > 0 ? forced : 0, (int)std::min<size_t>(n, 1u << 20),
                                     threads > 0 ? threads : engine_jit_threads(), &log);
    if (!log.empty()) set_last_error(log);
    return done;
  } catch (const Failure& f) { set_last_error(f.what()); return -f.code; }
  catch (const std::exception& ex) { set_last_error(ex.what()); return -ZPQ_E_DEVICE; }
}

const uint8_t* zpq_plan_blob(const zpq_plan* p, size_t* len) {
  if (len) *len = p ? p->blob.size() : 0;
  return p ? p->blob.data() : nullptr;
}

int zpq_plan_pipe_source(const zpq_plan* p, char* src, size_t cap, size_t* len, char key41[41]) {
  return zpq_plan_pipe_source_opts(p, 0, 0, 0, src, cap, len, key41);
}

static PipeOptions opts_of(int mode, int chunk, int group) {     // mode: a variant number of pipe_options (0, 1, 2)
  PipeOptions o = pipe_options(mode);
  if (chunk) o.chunk = chunk;
  o.group = group;
  return o;
}

int zpq_plan_pipe_source_opts(const zpq_plan* p, int mode, int chunk, int group, char* src, size_t cap, size_t* len, char key41[41]) {
  ZPQ_TRY
  if (!p) fail(ZPQ_E_ARG, "null plan");
  std::string source, key, why;
  if (!pipe_source_and_key(*p, opts_of(mode, chunk, group), source, key, why)) fail(ZPQ_E_UNSUPPORTED, why);
  if (len) *len = source.size();
  if (key41) { memcpy(key41, key.c_str(), 40); key41[40] = 0; }
  if (source.size() + 1 > cap) fail(ZPQ_E_OVERFLOW, "source buffer too small");
  memcpy(src, source.c_str(), source.size() + 1);
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_pcomp_source(const uint8_t* code, size_t codelen, int ph, int pm, char* src, size_t cap, size_t* len, char key41[41]) {
  ZPQ_TRY
  std::string source, key, why;
  if (!pcomp_source_and_key(code, codelen, ph, pm, source, key, why)) fail(ZPQ_E_UNSUPPORTED, why);
  if (len) *len = source.size();
  if (key41) { memcpy(key41, key.c_str(), 40); key41[40] = 0; }
  if (source.size() + 1 > cap) fail(ZPQ_E_OVERFLOW, "source buffer too small");
  memcpy(src, source.c_str(), source.size() + 1);
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_plan_pipe_layout(const zpq_plan* p, uint64_t out[16]) { return zpq_plan_pipe_layout_opts(p, 0, 0, 0, out); }

int zpq_plan_pipe_layout_opts(const zpq_plan* p, int mode, int chunk, int group, uint64_t out[16]) {
  ZPQ_TRY
  if (!p || !out) fail(ZPQ_E_ARG, "null argument");
  PipeLayout L;
  std::string why;
  if (!pipe_layout(*p, opts_of(mode, chunk, group), L, why)) fail(ZPQ_E_UNSUPPORTED, why);
  memset(out, 0, 16 * sizeof(uint64_t));
  out[0] = L.group_bytes; out[1] = (uint64_t)L.S; out[2] = (uint64_t)L.C; out[3] = L.light.size();
  out[4] = L.icm.size(); out[5] = L.isse.size(); out[6] = (uint64_t)L.mix_waves_per_group();
  out[7] = (uint64_t)L.hcomp_lanes; out[8] = (uint64_t)L.coder_level; out[9] = (uint64_t)L.G; out[10] = L.rows.size();
  out[11] = (uint64_t)L.mix_threads(); out[12] = (uint64_t)L.G; out[13] = (uint64_t)L.light_threads();
  // the persistent launch (0: the chain cannot be packed): workgroups per group, wavefronts per workgroup | progress counters per group << 16
  out[14] = L.persist_ok ? (uint64_t)L.ps_wpg : 0; out[15] = (uint64_t)L.ps_waves | (uint64_t)L.ps_nunit << 16;
  return ZPQ_OK;
  ZPQ_CATCH
}

const char* zpq_spec_cache_dir(void) { static std::string s; s = spec_cache_dir(); return s.c_str(); }
const char* zpq_spec_include_dir(void) { static std::string s; s = spec_include_dir(); return s.c_str(); }

int zpq_encode_batch(const zpq_plan* const* plans, const uint8_t* const* in, const uint32_t* in_len,
                     uint32_t nblocks, uint8_t* const* out, const uint32_t* out_cap, uint32_t* out_len,
                     int32_t* status) {
  ZPQ_TRY
  std::vector<HostBlock> hb(nblocks);
  for (uint32_t b = 0; b < nblocks; ++b) {
    if (!plans[b] || plans[b]->hdr().n == 0) fail(ZPQ_E_ARG, "encode_batch needs a modelled plan (n > 0)");
    hb[b] = HostBlock{plans[b], nullptr, 0, in[b], in_len[b], out[b], out_cap[b]};
  }
  std::vector<BlockResult> res;
  engine_code_host(false, hb, res);
  int worst = ZPQ_OK;
  for (uint32_t b = 0; b < nblocks; ++b) {
    if (out_len) out_len[b] = res[b].out_len;
    if (status) status[b] = res[b].status;
    if (res[b].status && !worst) worst = res[b].status;
  }
  if (worst) set_last_error("one or more blocks failed; see status[]");
  return worst;
  ZPQ_CATCH
}

int zpq_decode_batch(const zpq_plan* const* plans, const uint8_t* const* in, const uint32_t* in_len,
                     uint32_t nblocks, uint8_t* const* out, const uint32_t* max_out, uint32_t* out_len,
                     uint32_t* consumed, int32_t* status) {
  ZPQ_TRY
  std::vector<HostBlock> hb(nblocks);
  for (uint32_t b = 0; b < nblocks; ++b) {
    if (!plans[b] || plans[b]->hdr().n == 0) fail(ZPQ_E_ARG, "decode_batch needs a modelled plan (n > 0)");
    hb[b] = HostBlock{plans[b], nullptr, 0, in[b], in_len[b], out[b], max_out[b]};
  }
  std::vector<BlockResult> res;
  engine_code_host(true, hb, res);
  int worst = ZPQ_OK;
  for (uint32_t b = 0; b < nblocks; ++b) {
    if (out_len) out_len[b] = res[b].out_len;
    if (consumed) consumed[b] = res[b].consumed;
    if (status) status[b] = res[b].status;
    if (res[b].status && !worst) worst = res[b].status;
  }
  if (worst) set_last_error("one or more blocks failed; see status[]");
  return worst;
  ZPQ_CATCH
}

int zpq_encode_device(const zpq_plan* plan, const void* d_in, const uint64_t* in_off, const uint32_t* in_len,
                      uint32_t nblocks, void* d_out, const uint64_t* out_off, const uint32_t* out_cap,
                      zpq_block_result* d_res, void* stream, int timed) {
  ZPQ_TRY
  if (!plan || plan->hdr().n == 0) fail(ZPQ_E_ARG, "needs a modelled plan");
  engine_code_device(false, &plan, true, d_in, in_off, in_len, nblocks, d_out, out_off, out_cap, (BlockResult*)d_res,
                     stream, timed != 0);
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_decode_device(const zpq_plan* plan, const void* d_in, const uint64_t* in_off, const uint32_t* in_len,
                      uint32_t nblocks, void* d_out, const uint64_t* out_off, const uint32_t* max_out,
                      zpq_block_result* d_res, void* stream, int timed) {
  ZPQ_TRY
  if (!plan || plan->hdr().n == 0) fail(ZPQ_E_ARG, "needs a modelled plan");
  engine_code_device(true, &plan, true, d_in, in_off, in_len, nblocks, d_out, out_off, max_out, (BlockResult*)d_res,
                     stream, timed != 0);
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_code_device_multi(int decode, const zpq_plan* const* plans, const void* d_in, const uint64_t* in_off,
                          const uint32_t* in_len, uint32_t nblocks, void* d_out, const uint64_t* out_off,
                          const uint32_t* cap, zpq_block_result* d_res, void* stream, int timed) {
  ZPQ_TRY
  for (uint32_t b = 0; b < nblocks; ++b)
    if (!plans[b] || plans[b]->hdr().n == 0) fail(ZPQ_E_ARG, "needs modelled plans");
  engine_code_device(decode != 0, plans, false, d_in, in_off, in_len, nblocks, d_out, out_off, cap, (BlockResult*)d_res,
                     stream, timed != 0);
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_engine_count(void) { return engine_count(); }
int zpq_last_persistent(void) { return engine_last_persistent() ? 1 : 0; }
double zpq_last_persist_abort_ms(void) { return engine_last_persist_abort_ms(); }

int zpq_last_timing(float* init_ms, float* code_ms, uint32_t* blocks) {
  Timing t = engine_last_timing();
  if (init_ms) *init_ms = t.init_ms;
  if (code_ms) *code_ms = t.code_ms;
  if (blocks) *blocks = t.blocks;
  return ZPQ_OK;
}

int zpq_selftest(int32_t out8[8]) { ZPQ_TRY return engine_selftest(out8); ZPQ_CATCH }

// ------------------------------------------------------------ block drop-ins
int zpq_compress_blocks(const char* method, uint8_t* const* in, const uint32_t* in_len, uint32_t nblocks,
                        const char* const* filename, const char* const* comment, int dosha1,
                        uint8_t* const* out, const uint64_t* out_cap, uint64_t* out_len) {
  ZPQ_TRY
  std::vector<BlockInput> inputs(nblocks);
  for (uint32_t b = 0; b < nblocks; ++b)
    inputs[b] = BlockInput{in[b], in_len[b], filename ? filename[b] : nullptr, comment ? comment[b] : nullptr};
  std::vector<std::vector<U8>> archives;
  compress_blocks(method, inputs, dosha1 != 0, archives);
  int rc = ZPQ_OK;
  for (uint32_t b = 0; b < nblocks; ++b) {
    if (out_len) out_len[b] = archives[b].size();
    if (out && out[b] && out_cap && out_cap[b] >= archives[b].size())
      memcpy(out[b], archives[b].data(), archives[b].size());
    else { rc = ZPQ_E_OVERFLOW; set_last_error("output buffer too small"); }
  }
  return rc;
  ZPQ_CATCH
}

int zpq_decompress(const uint8_t* archive, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_len) {
  ZPQ_TRY
  U64 total = 0;
  bool overflow = false;
  decode_archive(archive, (size_t)n, [&](const U8* p, size_t len) {
    if (out && total + len <= cap) { if (len) memcpy(out + total, p, len); }      // (an empty segment's data pointer may be null)
    else overflow = true;
    total += len;
  });
  if (out_len) *out_len = total;
  if (overflow) { set_last_error("output buffer too small"); return ZPQ_E_OVERFLOW; }
  return ZPQ_OK;
  ZPQ_CATCH
}

// ------------------------------------------------------------- host pieces
void zpq_sha1(const uint8_t* in, uint64_t n, uint8_t out20[20]) {
  Sha1 s; s.update(in, (size_t)n); memcpy(out20, s.result(), 20);
}
void zpq_sha1_force_portable(int yes) { sha1_force_portable(yes != 0); }
void zpq_set_pcomp_step_limit(uint64_t steps) { postproc_set_step_limit(steps); }

void zpq_e8e9(uint8_t* data, uint32_t n) { e8e9_forward(data, n); }

// the host's suffix sorter (SA-IS, host/preproc.cpp): what the library uses when the device is not asked
int zpq_suffix_array_host(const uint8_t* in, uint32_t n, uint32_t* out) {
  ZPQ_TRY
  const std::vector<U32> sa = suffix_array(in, n);
  if (n) memcpy(out, sa.data(), 4ull * n);
  return ZPQ_OK;
  ZPQ_CATCH
}

// Suffix arrays of n host buffers, built on the device in one call (device/sa_kernels.hip); out[i] receives len[i] entries.
int zpq_suffix_arrays_device(const uint8_t* const* in, const uint32_t* len, uint32_t n, uint32_t* const* out) {
  ZPQ_TRY
  std::vector<std::pair<const U8*, U32>> blk;
  for (uint32_t i = 0; i < n; ++i) blk.push_back({in[i], len[i]});
  std::vector<std::vector<U32>> sa;
  std::string note;
  if (!engine_suffix_arrays(blk, sa, note)) fail(ZPQ_E_UNSUPPORTED, "suffix arrays on the device unavailable: " + note);
  for (uint32_t i = 0; i < n; ++i) if (len[i]) memcpy(out[i], sa[i].data(), 4ull * len[i]);
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_expand_method(const char* method, const uint8_t* data, uint32_t n, char* out, size_t cap) {
  ZPQ_TRY
  const std::string m = expand_method(method ? method : "", data, n);
  if (m.size() + 1 > cap) fail(ZPQ_E_OVERFLOW, "method buffer too small");
  memcpy(out, m.c_str(), m.size() + 1);
  return ZPQ_OK;
  ZPQ_CATCH
}

static int copy_assembled(const Assembled& as, uint8_t* hcomp, size_t hcap, size_t* hlen, uint8_t* pcomp,
                          size_t pcap, size_t* plen) {
  if (hlen) *hlen = as.hcomp.size();
  if (plen) *plen = as.pcomp.size();
  if (as.hcomp.size() > hcap || as.pcomp.size() > pcap) fail(ZPQ_E_OVERFLOW, "header buffer too small");
  if (hcomp) memcpy(hcomp, as.hcomp.data(), as.hcomp.size());
  if (pcomp && !as.pcomp.empty()) memcpy(pcomp, as.pcomp.data(), as.pcomp.size());
  return ZPQ_OK;
}

int zpq_method_to_header(const char* xmethod, int* args9, uint8_t* hcomp, size_t hcap, size_t* hlen,
                         uint8_t* pcomp, size_t pcap, size_t* plen) {
  ZPQ_TRY
  int args[9];
  const std::string cfg = make_config(xmethod ? xmethod : "", args);
  if (args9) memcpy(args9, args, sizeof(args));
  return copy_assembled(assemble(cfg.c_str(), args), hcomp, hcap, hlen, pcomp, pcap, plen);
  ZPQ_CATCH
}

// the stored header of one of the reference's three built-in models (Compressor::startBlock(int level), libzpaq.cpp:2793-2839:
// min.cfg, mid.cfg, max.cfg) -- what a block coded at that level carries; the batch entry points take it like any other header
int zpq_builtin_model_header(int level, uint8_t* hcomp, size_t hcap, size_t* hlen) {
  ZPQ_TRY
  const std::vector<U8> h = builtin_model(level);
  if (h.empty()) fail(ZPQ_E_ARG, "built-in models are levels 1, 2 and 3");
  if (hlen) *hlen = h.size();
  if (h.size() > hcap || !hcomp) fail(ZPQ_E_OVERFLOW, "header buffer too small");
  memcpy(hcomp, h.data(), h.size());
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_assemble(const char* config, const int* args9, uint8_t* hcomp, size_t hcap, size_t* hlen,
                 uint8_t* pcomp, size_t pcap, size_t* plen) {
  ZPQ_TRY
  return copy_assembled(assemble(config, args9), hcomp, hcap, hlen, pcomp, pcap, plen);
  ZPQ_CATCH
}

int zpq_preprocess_block(const char* xmethod, uint8_t* data, uint32_t n, uint8_t* out, size_t cap, size_t* len) {
  return zpq_preprocess_block_sa(xmethod, data, n, nullptr, out, cap, len);
}

// ... with the block's suffix array supplied by the caller (what zpq_compress_blocks does with the arrays the device built
// for a whole batch): for a method with E8E9 the caller applies zpq_e8e9 first -- the sort is over the filtered bytes --
// and says so with sa != null (then the call does not filter again)
int zpq_preprocess_block_sa(const char* xmethod, uint8_t* data, uint32_t n, const uint32_t* sa, uint8_t* out, size_t cap, size_t* len) {
  ZPQ_TRY
  if (!xmethod || (!data && n) || !len) fail(ZPQ_E_ARG, "null argument");
  int args[9];
  (void)make_config(xmethod, args);
  std::vector<U8> pre;
  const bool made = preprocess_block(data, n, args, pre, sa, sa != nullptr);
  if (!made) pre.assign(data, data + n);
  *len = pre.size();
  if (pre.size() > cap) fail(ZPQ_E_OVERFLOW, "output buffer too small");
  if (!pre.empty()) memcpy(out, pre.data(), pre.size());
  return ZPQ_OK;
  ZPQ_CATCH
}

// ---- the LZ77 parse through a suffix array as a token list (host/common.hpp LzToken: i, off, len, blit) ----
static void lz_args(const char* xmethod, int args[9]) {
  if (!xmethod) fail(ZPQ_E_ARG, "null argument");
  (void)make_config(xmethod, args);
  const int level = args[1] & 3;
  if (args[1] < 1 || args[1] > 7 || level < 1 || level > 2 || args[5] - args[0] < 21) fail(ZPQ_E_ARG, "not an LZ77 method that searches a suffix array");
}

// The host's parse of one block (data is E8E9-filtered in place first when the method says so).
int zpq_lz77_tokens_host(const char* xmethod, uint8_t* data, uint32_t n, uint32_t* tokens4, size_t cap, size_t* count) {
  ZPQ_TRY
  if ((!data && n) || !count || (!tokens4 && cap)) fail(ZPQ_E_ARG, "null argument");
  int args[9];
  lz_args(xmethod, args);
  if (args[1] > 4) e8e9_forward(data, n);
  std::vector<LzToken> toks;
  lz77_host_tokens(data, n, args, nullptr, toks);
  *count = toks.size();
  if (toks.size() > cap) fail(ZPQ_E_OVERFLOW, "token buffer too small");
  if (!toks.empty()) memcpy(tokens4, toks.data(), toks.size() * sizeof(LzToken));
  return ZPQ_OK;
  ZPQ_CATCH
}

// The coded stream of a token list (what zpq_preprocess_block returns for the same block and the host's list); `data` as the
// parse saw it (E8E9 already applied).
int zpq_lz77_serialize(const char* xmethod, const uint8_t* data, uint32_t n, const uint32_t* tokens4, size_t ntok, uint8_t* out, size_t cap, size_t* len) {
  ZPQ_TRY
  if ((!data && n) || (!tokens4 && ntok) || !len) fail(ZPQ_E_ARG, "null argument");
  int args[9];
  lz_args(xmethod, args);
  std::vector<U8> pre;
  lz77_serialize(data, n, args, (const LzToken*)tokens4, ntok, pre);
  *len = pre.size();
  if (pre.size() > cap) fail(ZPQ_E_OVERFLOW, "output buffer too small");
  if (!pre.empty()) memcpy(out, pre.data(), pre.size());
  return ZPQ_OK;
  ZPQ_CATCH
}

// What zpq_preprocess_block makes, for n host buffers in one call with the sort, the LZ77 parse and the BWT on the device
// (device/sa_kernels.hip, device/lz77_kernel.h); only for methods whose pre-processor sorts.  E8E9 is applied in place.
int zpq_preprocess_blocks_device(const char* xmethod, uint8_t* const* data, const uint32_t* len, uint32_t n, uint8_t* const* out, const size_t* cap,
                                 size_t* outlen) {
  ZPQ_TRY
  if (!xmethod || (n && (!data || !len || !out || !cap || !outlen))) fail(ZPQ_E_ARG, "null argument");
  int args[9];
  (void)make_config(xmethod, args);
  if (!preprocess_needs_suffix_array(args)) fail(ZPQ_E_ARG, "the method's pre-processor does not sort");
  // E8E9 rewrites the caller's buffers in place (as the reference rewrites its input).  They keep the filter only when the call
  // succeeds: on EVERY failure exit -- the device declines, something throws between the first filtered buffer and the last
  // copied output, an output buffer is too small -- the buffers filtered so far go back as they came, so that a caller
  // that falls back to zpq_preprocess_block (or calls again with larger buffers) never filters a block twice.
  struct Unfilter {
    uint8_t* const* data; const uint32_t* len; uint32_t done = 0; bool keep = false;
    ~Unfilter() { if (!keep) for (uint32_t i = 0; i < done; ++i) e8e9_inverse(data[i], len[i]); }
  } guard{data, len};
  std::vector<SortJob> jobs;
  for (uint32_t i = 0; i < n; ++i) {
    if (args[1] > 4) { e8e9_forward(data[i], len[i]); guard.done = i + 1; }
    jobs.push_back(sort_job(data[i], len[i], args));
  }
  std::vector<SortOut> outs;
  std::string note;
  if (!engine_sort_preprocess(jobs, outs, note)) fail(ZPQ_E_UNSUPPORTED, "pre-processing on the device unavailable: " + note);
  std::vector<std::vector<U8>> pres(n);
  bool fits = true;
  for (uint32_t i = 0; i < n; ++i) {
    std::vector<U8>& pre = pres[i];
    if (len[i] == 0) (void)preprocess_block(data[i], 0, args, pre, nullptr, true);
    else if (jobs[i].kind == 3) pre.swap(outs[i].bwt);
    else lz77_serialize(data[i], len[i], args, outs[i].toks.data(), outs[i].toks.size(), pre);
    outlen[i] = pre.size();                               // (every size is reported, also when some buffer is too small)
    fits = fits && pre.size() <= cap[i];
  }
  if (!fits) fail(ZPQ_E_OVERFLOW, "output buffer too small");
  for (uint32_t i = 0; i < n; ++i)
    if (!pres[i].empty()) memcpy(out[i], pres[i].data(), pres[i].size());
  guard.keep = true;
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_sha1_batch_device(const uint8_t* const* in, const uint32_t* len, uint32_t n, uint8_t* out20n) {
  ZPQ_TRY
  if (n && (!in || !len || !out20n)) fail(ZPQ_E_ARG, "null argument");
  engine_sha1_host(in, len, n, out20n);
  return ZPQ_OK;
  ZPQ_CATCH
}

int zpq_last_api_timing(double out[8]) {
  if (!out) return ZPQ_E_ARG;
  const ApiTiming t = last_api_timing();
  out[0] = t.total_ms; out[1] = t.front_ms; out[2] = t.device_ms; out[3] = t.stitch_ms;
  out[4] = t.kernel_init_ms; out[5] = t.kernel_code_ms; out[6] = (double)t.blocks; out[7] = (double)t.sa_device_blocks;
  return ZPQ_OK;
}

size_t zpq_table(int which, void* out, size_t cap) {
  try {
    const Tables& t = tables();
    const void* src; size_t n;
    switch (which) {
      case 0: src = t.squash; n = sizeof(t.squash); break;
      case 1: src = t.stretch; n = sizeof(t.stretch); break;
      case 2: src = t.dt; n = sizeof(t.dt); break;
      case 3: src = t.dt2k; n = sizeof(t.dt2k); break;
      case 4: src = t.ns; n = sizeof(t.ns); break;
      case 5: src = t.icm_init; n = sizeof(t.icm_init); break;
      case 6: src = t.isse_init; n = sizeof(t.isse_init); break;
      case 7: src = t.sse_row; n = sizeof(t.sse_row); break;
      case 8: src = t.stretch_cb; n = sizeof(t.stretch_cb); break;
      case 9: src = t.stretch_top; n = sizeof(t.stretch_top); break;
      default: return 0;
    }
    if (n > cap) return 0;
    memcpy(out, src, n);
    return n;
  } catch (...) { return 0; }
}

}  // extern "C"
