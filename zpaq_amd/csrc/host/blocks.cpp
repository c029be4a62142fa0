// Block-level drop-ins: the host work libzpaq::compressBlock / decompress do
// around the coder (method expansion, header, container bytes, SHA-1), with the
// coder itself replaced by one device batch.
#include "blocks.hpp"

#include <sched.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <thread>

#include "../device/engine.hpp"
#include "../device/plan.hpp"
#include "container.hpp"

namespace zpq {

// nproc says what the machine has, not what this process gets: the affinity mask, and the CPU quota of the cgroup
// (cgroup v2 cpu.max "quota period", v1 cpu.cfs_quota_us / cpu.cfs_period_us) -- a container with 16 CPUs of quota on a
// 256-thread host runs 16 threads at full speed and is throttled with more.
unsigned usable_cpus() {
  static const unsigned cached = [] {
    unsigned hw = std::thread::hardware_concurrency();
    if (!hw) hw = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) hw = std::min<unsigned>(hw, (unsigned)std::max(1, CPU_COUNT(&set)));
    auto read_two = [](const char* path, double& a, double& b) -> int {
      FILE* f = fopen(path, "r");
      if (!f) return 0;
      char x[64] = {0}, y[64] = {0};
      const int got = fscanf(f, "%63s %63s", x, y);
      fclose(f);
      if (got >= 1 && strcmp(x, "max") == 0) return -1;          // no quota
      a = atof(x);
      b = got >= 2 ? atof(y) : 0;
      return got;
    };
    double q = 0, p = 0;
    int r = read_two("/sys/fs/cgroup/cpu.max", q, p);
    if (r == 0) {
      double q1 = 0, p1 = 0, unused = 0;
      if (read_two("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q1, unused) >= 1 && read_two("/sys/fs/cgroup/cpu/cpu.cfs_period_us", p1, unused) >= 1) {
        q = q1; p = p1; r = 2;
      }
    }
    if (r >= 2 && q > 0 && p > 0) hw = std::min<unsigned>(hw, (unsigned)std::max(1.0, q / p + 0.5));
    return hw;
  }();
  return cached;
}

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
    // data-dependent chains: bound the cache.  The evicted plans are destroyed AFTER the lock is dropped: releasing a
    // plan frees device memory and unloads code objects (device-synchronising calls), which must not stall every other
    // caller of plan_for.
    std::map<std::vector<U8>, PlanPtr> evicted;
    {
      std::lock_guard<std::mutex> lk(g.mu);
      if (g.map.size() >= 256) evicted.swap(g.map);
      auto ins = g.map.emplace(header, p);
      p = ins.first->second;                          // another thread may have been faster
    }
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
// Bytes the device writes before anybody reads them: resize() must not fill them first.  (A batch of 1024 x 1 MiB blocks
// asks for 1.3 GB of output capacity; value-initialising it, block after block on the calling thread, was 200 of the 250 ms
// the API call spent beside its kernels.)
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = DefaultInitAlloc<U>; };
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
    else ::new ((void*)p) U(std::forward<A>(a)...);
  }
};
typedef std::vector<U8, DefaultInitAlloc<U8>> RawBytes;

struct EncJob {
  zpq_plan* plan; const U8* pp; U32 npp; const U8* data; U32 n; RawBytes* coded; U8* sha1_out = nullptr;
  // several segments in one block: lengths of the segments' shares of `data`, and where each one's code ends
  U32 nseg = 0; const U32* seg_len = nullptr; U32* seg_out_end = nullptr;
};

void encode_jobs(std::vector<EncJob>& jobs, bool* announced = nullptr) {
  std::vector<size_t> todo(jobs.size());
  for (size_t i = 0; i < jobs.size(); ++i) todo[i] = i;
  bool worst = false;
  while (!todo.empty()) {
    std::vector<HostBlock> hb;
    for (size_t i : todo) {
      EncJob& j = jobs[i];
      j.coded->resize(coded_cap((U64)j.n + j.npp, worst));
      HostBlock h{j.plan, j.pp, j.npp, j.data, j.n, j.coded->data(), (U32)j.coded->size()};
      h.sha1_out = j.sha1_out;
      h.nseg = j.nseg; h.seg_len = j.seg_len; h.seg_out_end = j.seg_out_end;
      hb.push_back(h);
    }
    std::vector<BlockResult> res;
    engine_code_host(false, hb, res, announced);
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
  const unsigned hw = usable_cpus();
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
  const size_t nb = in.size();
  for (size_t b = 0; b < nb; ++b) {
    const char* m = in[b].method ? in[b].method : method;
    if (!m || !m[0]) fail(ZPQ_E_ARG, "empty method");
  }
  struct Work {
    std::vector<U8> pp, header;
    RawBytes coded;
    std::vector<U8> pre;        // LZ77 / BWT stream when the method pre-processes (else the input itself is coded)
    bool use_pre = false;
    bool sha1_on_device = false;
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
  // 1. host front half, parallel over blocks: method -> chain; then the pre-processor; then the archive's front
  struct Front { int args[9]; Assembled as; bool sorts = false; };
  std::vector<Front> front(nb);
  parallel_blocks(nb, [&](size_t b) {
    Work& w = work[b];
    Front& f = front[b];
    const U32 n = in[b].n;
    if ((U64)n > 0x7FFFF000ull) fail(ZPQ_E_ARG, "block too large");
    const std::string xm = expand_method(in[b].method ? in[b].method : method, in[b].data, n);
    const std::string cfg = make_config(xm, f.args);
    f.as = assemble(cfg.c_str(), f.args);
    // The segment trailer carries the SHA-1 of the ORIGINAL block.  A modelled block that is coded as it is goes to the
    // device unchanged, so it is hashed there (sha1_blocks_kernel, beside the coder); everything else here.
    w.sha1_on_device = dosha1 && f.args[1] == 0 && f.as.hcomp[6] != 0;
    if (dosha1 && !w.sha1_on_device) { Sha1 s; s.update(in[b].data, n); memcpy(w.sha1, s.result(), 20); }
    if ((U64)n + 4096 > (0x100000ull << f.args[0])) fail(ZPQ_E_ARG, "block larger than the method's block size");
    f.sorts = n > 0 && preprocess_needs_suffix_array(f.args);
  });
  // Blocks whose pre-processor sorts suffixes (byte-aligned LZ77 with a suffix array: level 3; BWT): one suffix sort for
  // all of them on the device when there are enough to fill it (device/sa_kernels.hip), the host's SA-IS per block otherwise
  // or when the device declines (no GPU, blocks of 16 MiB and more, not enough memory).  The array is canonical: either
  // source gives the reference's parse.  E8E9 comes first where the method has it (it changes the bytes that are sorted).
  // ... and behind the sort, on the device as well (device/lz77_kernel.h): the LZ77 parse comes back as a list of matches the
  // host only has to write LZBuffer's codes for, the BWT as its bytes -- 16 bytes per match or n + 5 bytes over PCIe instead
  // of the 4 n of the array.  ZPAQ_AMD_DEVICE_PARSE=0: only the sort there, the host parses (the previous behaviour).
  std::vector<std::vector<U32>> dev_sa(nb);
  std::vector<SortOut> dev_pre(nb);
  std::vector<char> have_pre(nb, 0);
  {
    std::vector<size_t> sorting;
    U64 sort_bytes = 0;
    for (size_t b = 0; b < nb; ++b) if (front[b].sorts) { sorting.push_back(b); sort_bytes += in[b].n; }
    // the device sorter's range is known up front (blocks below 16 MiB, 2 GiB per batch, 65535 blocks): a batch outside it
    // is left to the host sorter before anything is touched
    bool in_range = sorting.size() <= 65535 && sort_bytes < (1ull << 31);
    for (size_t b : sorting) in_range = in_range && in[b].n < (1u << 24);
    if (in_range && sorting.size() >= 4 && sort_bytes >= (1u << 20) && engine_device_count() > 0) {
      parallel_blocks(sorting.size(), [&](size_t k) {        // E8E9 first where the method has it: it changes the bytes that are sorted
        const size_t b = sorting[k];
        if (front[b].args[1] > 4) e8e9_forward(in[b].data, in[b].n);
      });
      std::string note;
      bool got = false;
      const char* knob = getenv("ZPAQ_AMD_DEVICE_PARSE");
      if (!knob || knob[0] != '0') {
        std::vector<SortJob> sj;
        for (size_t b : sorting) sj.push_back(sort_job(in[b].data, in[b].n, front[b].args));
        std::vector<SortOut> so;
        try { got = engine_sort_preprocess(sj, so, note); } catch (const Failure&) { got = false; }   // (any device trouble: the next path)
        if (got)
          for (size_t k = 0; k < sorting.size(); ++k) { dev_pre[sorting[k]] = std::move(so[k]); have_pre[sorting[k]] = 1; }
      }
      if (!got) {
        std::vector<std::pair<const U8*, U32>> blk;
        for (size_t b : sorting) blk.push_back({in[b].data, in[b].n});
        std::vector<std::vector<U32>> sa;
        try { got = engine_suffix_arrays(blk, sa, note); } catch (const Failure&) { got = false; }      // (any device trouble: the host sorts)
        for (size_t k = 0; k < sorting.size(); ++k) if (got) dev_sa[sorting[k]].swap(sa[k]);
      }
      for (size_t b : sorting) front[b].sorts = true;                 // (E8E9 is done either way)
      tm.sa_device_blocks = got ? (U32)sorting.size() : 0;
    } else {
      for (size_t b : sorting) front[b].sorts = false;   // nothing was done up front: preprocess_block does it all
    }
  }
  parallel_blocks(nb, [&](size_t b) {
    Work& w = work[b];
    const Front& f = front[b];
    const int* args = f.args;
    const Assembled& as = f.as;
    const U32 n = in[b].n;
    // LZ77 / BWT / E8E9 (libzpaq.cpp:7709-7716); E8E9 rewrites the caller's buffer in place, as the reference does
    if (have_pre[b] && n > 0) {                        // parsed / transformed on the device: the coder, or nothing left to do
      if ((args[1] & 3) == 3) w.pre.swap(dev_pre[b].bwt);
      else lz77_serialize(in[b].data, n, args, dev_pre[b].toks.data(), dev_pre[b].toks.size(), w.pre);
      w.use_pre = true;
      SortOut().toks.swap(dev_pre[b].toks);
    } else {
      w.use_pre = preprocess_block(in[b].data, n, args, w.pre, dev_sa[b].empty() ? nullptr : dev_sa[b].data(), f.sorts);
    }
    std::vector<U32>().swap(dev_sa[b]);
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
                            work[b].use_pre ? (U32)work[b].pre.size() : in[b].n, &work[b].coded,
                            work[b].sha1_on_device ? work[b].sha1 : nullptr, 0, nullptr, nullptr});
  const double t1 = now_ms();
  if (jobs.empty()) announce.off();
  else encode_jobs(jobs, &announce.on);       // the queue withdraws the announcement when our blocks are in it
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
  RawBytes coded;
  std::vector<EncJob> jobs(1, EncJob{plan_for(plans, header), pp, (U32)npp, data, (U32)n, &coded, nullptr, 0, nullptr, nullptr});
  if ((U64)n + npp > 0x7FFFF000ull) fail(ZPQ_E_ARG, "segment too large");
  encode_jobs(jobs);
  return std::vector<U8>(coded.begin(), coded.end());
}

// A modelled block of several segments: one device job, the end-of-segment code between them.
std::vector<std::vector<U8>> encode_payload_segments(const std::vector<U8>& header, const std::vector<std::vector<U8>>& segments) {
  PlanCache plans;
  std::vector<U8> all;
  RawBytes coded;
  std::vector<U32> lens, ends(segments.size(), 0);
  for (const auto& sg : segments) {
    if ((U64)all.size() + sg.size() > 0x7FFFF000ull) fail(ZPQ_E_ARG, "block too large");
    all.insert(all.end(), sg.begin(), sg.end());
    lens.push_back((U32)sg.size());
  }
  std::vector<EncJob> jobs(1, EncJob{plan_for(plans, header), nullptr, 0, all.data(), (U32)all.size(), &coded, nullptr,
                                     (U32)segments.size(), lens.data(), ends.data()});
  if (segments.size() <= 1) jobs[0].nseg = 0;
  encode_jobs(jobs);
  std::vector<std::vector<U8>> out;
  if (segments.size() <= 1) { out.emplace_back(coded.begin(), coded.end()); return out; }
  U32 at = 0;
  for (size_t i = 0; i < segments.size(); ++i) {
    if (ends[i] < at || ends[i] > coded.size()) fail(ZPQ_E_DEVICE, "device coder returned inconsistent segment ends");
    out.emplace_back(coded.begin() + at, coded.begin() + ends[i]);
    at = ends[i];
  }
  return out;
}

namespace {

struct DecJob {
  zpq_plan* plan; const U8* payload; U32 len; U64 hint; RawBytes* decoded;
  U32 nseg = 0; const U32* seg_len = nullptr; U32* seg_out_end = nullptr;     // several segments: coded lengths / decoded ends
};

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
      HostBlock h{jobs[i].plan, nullptr, 0, jobs[i].payload, jobs[i].len, jobs[i].decoded->data(), (U32)cap[i]};
      h.nseg = jobs[i].nseg; h.seg_len = jobs[i].seg_len; h.seg_out_end = jobs[i].seg_out_end;
      hb.push_back(h);
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
  RawBytes decoded;
  if (len > 0xFFFFFFF0ull) fail(ZPQ_E_NOMEM, "segment too large");
  std::vector<DecJob> jobs(1, DecJob{plan_for(plans, header), payload, (U32)len, hint, &decoded, 0, nullptr, nullptr});
  decode_jobs(jobs);
  return std::vector<U8>(decoded.begin(), decoded.end());
}

// The first `limit` bytes of ONE segment's coded stream (Decompresser::decompress(n), libzpaq.cpp:2315-2343: a caller may
// stop there).  The device decodes from the block's first bit and stops at `limit` bytes or at the end of the segment,
// whichever comes first: *complete says which.
std::vector<U8> decode_payload_prefix(const std::vector<U8>& header, const std::vector<U8>& payload, U64 limit, bool* complete) {
  PlanCache plans;
  RawBytes decoded;
  if (payload.size() > 0xFFFFFFF0ull || limit > 0xFFFFFFF0ull) fail(ZPQ_E_NOMEM, "segment too large");
  decoded.resize(limit);
  std::vector<HostBlock> hb(1, HostBlock{plan_for(plans, header), nullptr, 0, payload.data(), (U32)payload.size(), decoded.data(), (U32)limit});
  std::vector<BlockResult> res;
  engine_code_host(true, hb, res);
  if (res[0].status == ZPQ_E_CORRUPT) fail(ZPQ_E_CORRUPT, "archive corrupted");
  if (res[0].status == ZPQ_E_EOF) fail(ZPQ_E_EOF, "unexpected end of file");
  if (res[0].status == ZPQ_E_VM) fail(ZPQ_E_VM, "ZPAQL execution error");
  if (res[0].status) fail(res[0].status, "device decoder failed");
  *complete = res[0].consumed != 0;                       // (0: the limit was reached before the end-of-segment code)
  decoded.resize(std::min<U64>(res[0].out_len, limit));
  return std::vector<U8>(decoded.begin(), decoded.end());
}

// The segments of ONE modelled block decoded together (model and coder state run on from segment to segment):
// payloads[s] = coded bytes of segment s incl. its terminator; returns the decoded bytes per segment.
std::vector<std::vector<U8>> decode_payload_segments(const std::vector<U8>& header, const std::vector<std::vector<U8>>& payloads, U64 hint) {
  PlanCache plans;
  std::vector<U8> all;
  RawBytes decoded;
  std::vector<U32> lens, ends(payloads.size(), 0);
  for (const auto& p : payloads) {
    if ((U64)all.size() + p.size() > 0xFFFFFFF0ull) fail(ZPQ_E_NOMEM, "block too large");
    all.insert(all.end(), p.begin(), p.end());
    lens.push_back((U32)p.size());
  }
  std::vector<DecJob> jobs(1, DecJob{plan_for(plans, header), all.data(), (U32)all.size(), hint, &decoded,
                                     (U32)payloads.size(), lens.data(), ends.data()});
  if (payloads.size() <= 1) jobs[0].nseg = 0;
  decode_jobs(jobs);
  std::vector<std::vector<U8>> out;
  if (payloads.size() <= 1) { out.emplace_back(decoded.begin(), decoded.end()); return out; }
  U32 at = 0;
  for (size_t i = 0; i < payloads.size(); ++i) {
    if (ends[i] < at || ends[i] > decoded.size()) fail(ZPQ_E_DEVICE, "device decoder returned inconsistent segment ends");
    out.emplace_back(decoded.begin() + at, decoded.begin() + ends[i]);
    at = ends[i];
  }
  return out;
}

void decode_archive(const U8* a, size_t n, const std::function<void(const U8*, size_t)>& sink) {
  struct Seg {
    FoundSegment fs;
    zpq_plan* plan = nullptr;     // null: stored block
    std::vector<U8> header;
    size_t payload_end = 0;
    RawBytes decoded;             // PP byte(s) + data (written by the device: not filled first)
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
    // every block header goes through the parser (ZPAQL::read's checks, libzpaq.cpp:1145-1216: sizes, COMP END, HCOMP
    // END), also the ones without a model, whose segments never see a plan
    zpq_plan* parsed = plan_for(plans, blk.header);
    zpq_plan* plan = modeled ? parsed : nullptr;
    int nseg = 0;
    (void)nseg;
    for (;;) {
      std::unique_ptr<Seg> s(new Seg);
      if (!find_segment(a, n, pos, s->fs)) break;
      ++nseg;
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
  // one device job per modelled block; a block of several segments is ONE job (its model and coder state run on
  // from segment to segment), fed with the segments' payloads back to back
  struct Multi { std::vector<U8> all; RawBytes decoded; std::vector<U32> lens, ends; std::vector<size_t> members; };
  std::vector<std::unique_ptr<Multi>> multis;
  std::vector<DecJob> jobs;
  {
    std::vector<size_t> count(nblocks, 0);
    for (auto& s : segs) if (s->plan) ++count[s->block];
    std::map<size_t, Multi*> open;
    for (size_t i = 0; i < segs.size(); ++i) {
      Seg& s = *segs[i];
      if (!s.plan) continue;
      const U8* payload = a + s.fs.payload_begin;
      const size_t plen = s.payload_end - s.fs.payload_begin;
      if (count[s.block] == 1) {
        jobs.push_back(DecJob{s.plan, payload, (U32)plen, s.hint ? s.hint + 1 : 0, &s.decoded, 0, nullptr, nullptr});
        continue;
      }
      Multi*& m = open[s.block];
      if (!m) { multis.emplace_back(new Multi); m = multis.back().get(); }
      if ((U64)m->all.size() + plen > 0xFFFFFFF0ull) fail(ZPQ_E_NOMEM, "block too large");
      m->all.insert(m->all.end(), payload, payload + plen);
      m->lens.push_back((U32)plen);
      m->members.push_back(i);
    }
    for (auto& m : multis) {
      m->ends.assign(m->lens.size(), 0);
      U64 hint = 0;
      for (size_t i : m->members) hint = (hint || segs[i]->hint) ? hint + segs[i]->hint + 1 : 0;
      jobs.push_back(DecJob{segs[m->members[0]]->plan, m->all.data(), (U32)m->all.size(), hint, &m->decoded,
                            (U32)m->lens.size(), m->lens.data(), m->ends.data()});
    }
  }
  decode_jobs(jobs);
  for (auto& m : multis) {
    U32 at = 0;
    for (size_t k = 0; k < m->members.size(); ++k) {
      if (m->ends[k] < at || m->ends[k] > m->decoded.size()) fail(ZPQ_E_DEVICE, "device decoder returned inconsistent segment ends");
      segs[m->members[k]]->decoded.assign(m->decoded.begin() + at, m->decoded.begin() + m->ends[k]);
      at = m->ends[k];
    }
  }
  // stored segments: Decoder::decompress n == 0 branch (2146-2154)
  for (auto& s : segs) {
    if (s->plan) continue;
    size_t p = s->fs.payload_begin;
    for (;;) {
      const U32 l = (U32)a[p] << 24 | (U32)a[p + 1] << 16 | (U32)a[p + 2] << 8 | a[p + 3];
      p += 4;
      if (!l) break;
      s->decoded.insert(s->decoded.end(), a + p, a + p + l);
      p += l;
    }
  }
  // Post-processing.  Segments whose block carries a PCOMP program (LZ77 / BWT / E8E9 methods) go through it ON THE
  // DEVICE when there is enough of them to fill lanes: one lane per segment, the program translated like HCOMP
  // (device/pcomp_kernel.h).  Blocks of several segments share one machine across segments and small jobs are not
  // worth a launch: those run through the host interpreter (host/postproc.cpp), like stored blocks on a box without
  // a GPU.  ZPAQ_AMD_PCOMP=device|host forces one (interpret: host, and the interpreter instead of the translated programs).
  std::vector<size_t> per_block(nblocks, 0);
  for (auto& s : segs) ++per_block[s->block];
  std::vector<std::vector<U8>> done(segs.size());
  std::vector<char> on_device(segs.size(), 0);
  {
    const char* mode = getenv("ZPAQ_AMD_PCOMP");
    const bool force_dev = mode && !strcmp(mode, "device"), force_host = mode && (!strcmp(mode, "host") || !strcmp(mode, "interpret"));
    std::map<std::vector<U8>, std::vector<size_t>> by_prog;     // key: ph pm code
    U64 prog_bytes = 0;
    for (size_t i = 0; i < segs.size() && !force_host; ++i) {
      const Seg& s = *segs[i];
      const RawBytes& d = s.decoded;
      if (per_block[s.block] != 1 || d.size() < 4 || d[0] != 1) continue;
      const size_t len = d[1] + 256u * d[2];
      if (len < 1 || d.size() < 3 + len) continue;
      std::vector<U8> key;
      key.push_back(s.header[4]); key.push_back(s.header[5]);
      key.insert(key.end(), d.begin() + 3, d.begin() + 3 + (long)len);
      by_prog[key].push_back(i);
      prog_bytes += d.size();
    }
    size_t nprog = 0;
    for (auto& kv : by_prog) nprog += kv.second.size();
    if (nprog && (force_dev || nprog >= 4 || prog_bytes >= (256u << 10)) && engine_device_count() > 0) {
      for (auto& kv : by_prog) {
        const std::vector<U8>& key = kv.first;
        std::vector<PcompSeg> ps;
        for (size_t i : kv.second) {
          const Seg& s = *segs[i];
          const size_t skip = 3 + (key.size() - 2);
          ps.push_back(PcompSeg{s.decoded.data() + skip, (U32)(s.decoded.size() - skip), s.hint, &done[i]});
        }
        std::string note;
        if (engine_pcomp(key.data() + 2, key.size() - 2, key[0], key[1], ps, note))
          for (size_t i : kv.second) on_device[i] = 1;
        else if (force_dev) fail(ZPQ_E_UNSUPPORTED, "PCOMP on the device unavailable: " + note);
      }
    }
  }
  // Host post-processing and the checksum are per block (its segments in order: one PostProcessor per block, only the
  // first segment carries the PP header, libzpaq.cpp:2320-2330) and blocks are independent: a window of blocks at a time
  // on the host cores, then the window's data to the sink in archive order.
  std::vector<std::pair<size_t, size_t>> spans;                 // [first, last) segment of each block, in archive order
  for (size_t si = 0; si < segs.size();) {
    size_t e = si + 1;
    while (e < segs.size() && segs[e]->block == segs[si]->block) ++e;
    spans.push_back({si, e});
    si = e;
  }
  const size_t kWindow = 64;
  for (size_t w0 = 0; w0 < spans.size(); w0 += kWindow) {
    const size_t w1 = std::min(spans.size(), w0 + kWindow);
    parallel_blocks(w1 - w0, [&](size_t k) {
      const auto span = spans[w0 + k];
      std::unique_ptr<PostProcessor> pp;
      for (size_t si = span.first; si < span.second; ++si) {
        auto& s = segs[si];
        if (!on_device[si]) {
          if (!pp) pp.reset(new PostProcessor(s->header[4], s->header[5]));
          done[si].clear();
          pp->segment(s->decoded.data(), s->decoded.size(), done[si]);
        }
        RawBytes().swap(s->decoded);
        if (s->fs.has_sha1) {
          Sha1 h; h.update(done[si].data(), done[si].size());
          if (memcmp(h.result(), s->fs.sha1, 20) != 0) fail(ZPQ_E_CORRUPT, "segment checksum mismatch");
        }
      }
    });
    for (size_t b = w0; b < w1; ++b)
      for (size_t si = spans[b].first; si < spans[b].second; ++si) {
        sink(done[si].data(), done[si].size());
        std::vector<U8>().swap(done[si]);
      }
  }
}

}  // namespace zpq
