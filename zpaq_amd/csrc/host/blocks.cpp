// Block-level drop-ins: the host work libzpaq::compressBlock / decompress do
// around the coder (method expansion, header, container bytes, SHA-1), with the
// coder itself replaced by one device batch.
#include "blocks.hpp"

#include <sched.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <thread>

#include "../device/engine.hpp"
#include "../device/plan.hpp"
#include "container.hpp"

namespace zpq {

namespace {

struct PlanDeleter {
  void operator()(zpq_plan* p) const { if (p) { engine_plan_release(p); delete p; } }
};
typedef std::shared_ptr<zpq_plan> PlanPtr;

// Plans are shared by every call of the process (keyed by the header bytes): a plan owns device copies and loaded
// code objects, and rebuilding them per call would re-read, re-hash and re-load the per-header kernels every time.
// A call keeps the plans it uses alive through its PlanCache (shared_ptr), so trimming the global map is safe.
struct GlobalPlans {
  std::mutex mu;
  std::map<std::vector<U8>, PlanPtr> map;
};
GlobalPlans& global_plans() { static GlobalPlans g; return g; }

typedef std::map<std::vector<U8>, PlanPtr> PlanCache;     // the plans one call uses

zpq_plan* plan_for(PlanCache& cache, const std::vector<U8>& header) {
  auto it = cache.find(header);
  if (it != cache.end()) return it->second.get();
  GlobalPlans& g = global_plans();
  PlanPtr p;
  {
    std::lock_guard<std::mutex> lk(g.mu);
    auto gi = g.map.find(header);
    if (gi != g.map.end()) p = gi->second;
  }
  if (!p) {
    p = PlanPtr(plan_from_header(header.data(), header.size()), PlanDeleter());
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.map.size() >= 256) g.map.clear();           // data-dependent chains: bound the cache
    auto ins = g.map.emplace(header, p);
    p = ins.first->second;                            // another thread may have been faster
  }
  cache.emplace(header, p);
  return p.get();
}

std::mutex g_api_mu;
ApiTiming g_api_last;
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Device-side capacity for the coded form of n input bytes.  Incompressible
// data costs about n*(1+2^-12)+4; a mispredicted bit can cost up to 16 bits, so
// adversarial inputs could exceed any linear bound: they come back as
// ZPQ_E_OVERFLOW and are retried with the worst case (17n+64).
U32 coded_cap(U64 n, bool worst) {
  U64 c = worst ? 17 * n + 64 : n + n / 4 + 4096;
  if (c > 0xFFFFFFF0ull) fail(ZPQ_E_NOMEM, "block too large");
  return (U32)c;
}

// Runs the encoder for a list of (plan, pp, data) jobs, retrying overflowed ones.
struct EncJob { zpq_plan* plan; const U8* pp; U32 npp; const U8* data; U32 n; std::vector<U8>* coded; };

void encode_jobs(std::vector<EncJob>& jobs, bool announced = false) {
  std::vector<size_t> todo(jobs.size());
  for (size_t i = 0; i < jobs.size(); ++i) todo[i] = i;
  bool worst = false;
  while (!todo.empty()) {
    std::vector<HostBlock> hb;
    for (size_t i : todo) {
      EncJob& j = jobs[i];
      j.coded->resize(coded_cap((U64)j.n + j.npp, worst));
      hb.push_back(HostBlock{j.plan, j.pp, j.npp, j.data, j.n, j.coded->data(), (U32)j.coded->size()});
    }
    std::vector<BlockResult> res;
    engine_code_host(false, hb, res, announced);
    announced = false;
    std::vector<size_t> again;
    for (size_t k = 0; k < todo.size(); ++k) {
      EncJob& j = jobs[todo[k]];
      if (res[k].status == ZPQ_E_OVERFLOW && !worst) { again.push_back(todo[k]); continue; }
      if (res[k].status)
        fail(res[k].status, res[k].status == ZPQ_E_VM ? "ZPAQL execution error"
                                                       : "device coder failed (status " + std::to_string(res[k].status) + ")");
      j.coded->resize(res[k].out_len);
    }
    todo.swap(again);
    worst = true;
  }
}

// Per-block host work (SHA-1, period scan, header assembly, archive stitching) is independent per
// block: spread it over the host cores the process may use, like zpaq.cpp's compressThread pool.
template <class F>
void parallel_blocks(size_t n, F&& fn) {
  unsigned hw = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) hw = std::min<unsigned>(hw ? hw : 1, (unsigned)CPU_COUNT(&set));
  const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 1, 32), n / 4));
  if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  std::atomic<size_t> next(0);
  std::exception_ptr err;
  std::atomic<bool> failed(false);
  std::vector<std::thread> pool;
  for (size_t t = 0; t < nt; ++t)
    pool.emplace_back([&] {
      try {
        for (size_t i; (i = next.fetch_add(1)) < n && !failed;) fn(i);
      } catch (...) {
        if (!failed.exchange(true)) err = std::current_exception();
      }
    });
  for (auto& th : pool) th.join();
  if (failed) std::rethrow_exception(err);
}

}  // namespace

void compress_blocks(const char* method, const std::vector<BlockInput>& in, bool dosha1,
                     std::vector<std::vector<U8>>& archives) {
  if (!method || !method[0]) fail(ZPQ_E_ARG, "empty method");
  const size_t nb = in.size();
  struct Work {
    std::vector<U8> pp, coded, header;
    std::vector<U8> pre;        // LZ77 / BWT stream when the method pre-processes (else the input itself is coded)
    bool use_pre = false;
    U8 sha1[20];
  };
  std::vector<Work> work(nb);
  archives.assign(nb, std::vector<U8>());
  ApiTiming tm;
  const double t0 = now_ms();
  // other threads doing the same thing should end up in the same device batch: tell the queue we are coming
  struct Announce {
    bool on = true;
    Announce() { engine_caller_enter(); }
    void off() { if (on) { engine_caller_leave(); on = false; } }
    ~Announce() { off(); }
  } announce;
  // 1. host front half, parallel over blocks
  parallel_blocks(nb, [&](size_t b) {
    Work& w = work[b];
    const U32 n = in[b].n;
    if ((U64)n > 0x7FFFF000ull) fail(ZPQ_E_ARG, "block too large");
    if (dosha1) { Sha1 s; s.update(in[b].data, n); memcpy(w.sha1, s.result(), 20); }
    const std::string xm = expand_method(method, in[b].data, n);
    int args[9];
    const std::string cfg = make_config(xm, args);
    const Assembled as = assemble(cfg.c_str(), args);
    if ((U64)n + 4096 > (0x100000ull << args[0])) fail(ZPQ_E_ARG, "block larger than the method's block size");
    // LZ77 / BWT / E8E9 (libzpaq.cpp:7709-7716); E8E9 rewrites the caller's buffer in place, as the reference does
    w.use_pre = preprocess_block(in[b].data, n, args, w.pre);
    std::string cs = std::to_string(n);
    if (in[b].comment) cs += std::string(" ") + in[b].comment;
    write_block_prologue(archives[b], as.hcomp, in[b].filename, cs);
    // PP header: 0 = pass, or 1 len16 pcomp  (Compressor::postProcess 2888-2917)
    if (as.pcomp.empty()) w.pp.push_back(0);
    else { w.pp.push_back(1); w.pp.insert(w.pp.end(), as.pcomp.begin(), as.pcomp.end()); }
    const U8* src = w.use_pre ? w.pre.data() : in[b].data;
    const size_t srcn = w.use_pre ? w.pre.size() : n;
    if (as.hcomp[6] == 0) write_stored_payload(archives[b], w.pp.data(), w.pp.size(), src, srcn);
    else w.header = as.hcomp;
  });
  // 2. one device batch for every modelled block (blocks with identical headers share a plan)
  PlanCache plans;
  std::vector<EncJob> jobs;
  for (size_t b = 0; b < nb; ++b)
    if (!work[b].header.empty())
      jobs.push_back(EncJob{plan_for(plans, work[b].header), work[b].pp.data(), (U32)work[b].pp.size(),
                            work[b].use_pre ? work[b].pre.data() : in[b].data,
                            work[b].use_pre ? (U32)work[b].pre.size() : in[b].n, &work[b].coded});
  const double t1 = now_ms();
  if (jobs.empty()) announce.off();
  else { announce.on = false; encode_jobs(jobs, true); }    // the queue withdraws the announcement when our blocks are in it
  const double t2 = now_ms();
  const Timing dev = engine_last_timing();
  // 3. stitch the archives
  parallel_blocks(nb, [&](size_t b) {
    archives[b].insert(archives[b].end(), work[b].coded.begin(), work[b].coded.end());
    write_block_epilogue(archives[b], dosha1 ? work[b].sha1 : nullptr);
  });
  const double t3 = now_ms();
  tm.front_ms = t1 - t0; tm.device_ms = t2 - t1; tm.stitch_ms = t3 - t2; tm.total_ms = t3 - t0;
  tm.kernel_init_ms = dev.init_ms; tm.kernel_code_ms = dev.code_ms;
  tm.blocks = nb;
  std::lock_guard<std::mutex> lk(g_api_mu);
  g_api_last = tm;
}

ApiTiming last_api_timing() {
  std::lock_guard<std::mutex> lk(g_api_mu);
  return g_api_last;
}

std::vector<U8> encode_payload(const std::vector<U8>& header, const U8* pp, size_t npp, const U8* data, size_t n) {
  PlanCache plans;
  std::vector<U8> coded;
  std::vector<EncJob> jobs(1, EncJob{plan_for(plans, header), pp, (U32)npp, data, (U32)n, &coded});
  if ((U64)n + npp > 0x7FFFF000ull) fail(ZPQ_E_ARG, "segment too large");
  encode_jobs(jobs);
  return coded;
}

namespace {

struct DecJob { zpq_plan* plan; const U8* payload; U32 len; U64 hint; std::vector<U8>* decoded; };

void decode_jobs(std::vector<DecJob>& jobs) {
  std::vector<size_t> todo(jobs.size());
  std::vector<U64> cap(jobs.size());
  for (size_t i = 0; i < jobs.size(); ++i) {
    todo[i] = i;
    // the hint is the uncompressed size; an LZ77/BWT coded stream can exceed it by a few percent
    cap[i] = (jobs[i].hint ? jobs[i].hint + jobs[i].hint / 16 : 4 * (U64)jobs[i].len) + 65536 + 8;
  }
  while (!todo.empty()) {
    std::vector<HostBlock> hb;
    for (size_t i : todo) {
      if (cap[i] > 0xFFFFFFF0ull) fail(ZPQ_E_NOMEM, "segment too large");
      jobs[i].decoded->resize(cap[i]);
      hb.push_back(HostBlock{jobs[i].plan, nullptr, 0, jobs[i].payload, jobs[i].len, jobs[i].decoded->data(), (U32)cap[i]});
    }
    std::vector<BlockResult> res;
    engine_code_host(true, hb, res);
    std::vector<size_t> again;
    for (size_t k = 0; k < todo.size(); ++k) {
      const size_t i = todo[k];
      if (res[k].status == ZPQ_E_CORRUPT) fail(ZPQ_E_CORRUPT, "archive corrupted");
      if (res[k].status == ZPQ_E_EOF) fail(ZPQ_E_EOF, "unexpected end of file");
      if (res[k].status == ZPQ_E_VM) fail(ZPQ_E_VM, "ZPAQL execution error");
      if (res[k].status) fail(res[k].status, "device decoder failed");
      if (res[k].consumed == 0) { cap[i] *= 4; again.push_back(i); }   // reached max_out before EOS
      else jobs[i].decoded->resize(res[k].out_len);
    }
    todo.swap(again);
  }
}

}  // namespace

std::vector<U8> decode_payload(const std::vector<U8>& header, const U8* payload, size_t len, U64 hint) {
  PlanCache plans;
  std::vector<U8> decoded;
  if (len > 0xFFFFFFF0ull) fail(ZPQ_E_NOMEM, "segment too large");
  std::vector<DecJob> jobs(1, DecJob{plan_for(plans, header), payload, (U32)len, hint, &decoded});
  decode_jobs(jobs);
  return decoded;
}

void decode_archive(const U8* a, size_t n, const std::function<void(const U8*, size_t)>& sink) {
  struct Seg {
    FoundSegment fs;
    zpq_plan* plan = nullptr;     // null: stored block
    std::vector<U8> header;
    size_t payload_end = 0;
    std::vector<U8> decoded;      // PP byte(s) + data
    U64 hint = 0;
    size_t block = 0;             // index of the block the segment belongs to
  };
  std::vector<std::unique_ptr<Seg>> segs;
  PlanCache plans;
  size_t pos = 0, nblocks = 0;
  FoundBlock blk;
  while (find_block(a, n, pos, blk)) {
    const size_t this_block = nblocks++;
    const bool modeled = blk.header[6] != 0;
    zpq_plan* plan = modeled ? plan_for(plans, blk.header) : nullptr;
    int nseg = 0;
    for (;;) {
      std::unique_ptr<Seg> s(new Seg);
      if (!find_segment(a, n, pos, s->fs)) break;
      if (modeled && ++nseg > 1)
        fail(ZPQ_E_UNSUPPORTED, "multi-segment modelled blocks are outside this build's scope");
      s->plan = plan;
      s->header = blk.header;
      s->block = this_block;
      s->payload_end = skip_payload(a, n, pos, modeled);
      pos = s->payload_end;
      read_segment_end(a, n, pos, s->fs);
      // compressBlock writes the uncompressed size as the leading decimal of the comment (7706)
      for (char ch : s->fs.comment) {
        if (ch < '0' || ch > '9') break;
        s->hint = s->hint * 10 + (U64)(ch - '0');
        if (s->hint > (1ull << 40)) { s->hint = 0; break; }
      }
      segs.push_back(std::move(s));
    }
  }
  std::vector<DecJob> jobs;
  for (auto& s : segs)
    if (s->plan)
      jobs.push_back(DecJob{s->plan, a + s->fs.payload_begin, (U32)(s->payload_end - s->fs.payload_begin),
                            s->hint ? s->hint + 1 : 0, &s->decoded});
  decode_jobs(jobs);
  std::unique_ptr<PostProcessor> pp;
  size_t pp_block = (size_t)-1;
  for (auto& s : segs) {
    if (!s->plan) {   // stored: Decoder::decompress n==0 branch (2146-2154)
      size_t p = s->fs.payload_begin;
      for (;;) {
        const U32 l = (U32)a[p] << 24 | (U32)a[p + 1] << 16 | (U32)a[p + 2] << 8 | a[p + 3];
        p += 4;
        if (!l) break;
        s->decoded.insert(s->decoded.end(), a + p, a + p + l);
        p += l;
      }
    }
    // one PostProcessor per block: only its first segment carries the PP header (libzpaq.cpp:2320-2330)
    if (s->block != pp_block) { pp.reset(new PostProcessor(s->header[4], s->header[5])); pp_block = s->block; }
    std::vector<U8> data;
    pp->segment(s->decoded.data(), s->decoded.size(), data);
    std::vector<U8>().swap(s->decoded);
    if (s->fs.has_sha1) {
      Sha1 h; h.update(data.data(), data.size());
      if (memcmp(h.result(), s->fs.sha1, 20) != 0) fail(ZPQ_E_CORRUPT, "segment checksum mismatch");
    }
    sink(data.data(), data.size());
  }
}

}  // namespace zpq
