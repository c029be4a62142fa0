// Batch engine: owns the device, the HBM arena pool and the launch sequence
// (init_arena -> code_*).  One process drives one GPU; callers are serialised.
#include "engine.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "kernels.h"
#include "spec_loader.hpp"

namespace zpq {

#define HIP_CHECK(expr)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      int code_ = (e_ == hipErrorOutOfMemory) ? ZPQ_E_NOMEM : ZPQ_E_DEVICE;                   \
      fail(code_, std::string(#expr) + ": " + hipGetErrorString(e_));                         \
    }                                                                                         \
  } while (0)

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  void ensure(size_t n) {
    if (n <= cap) return;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = (n + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1);
    HIP_CHECK(hipMalloc(&p, want));
    cap = want;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Engine {
  std::mutex mu;
  bool ready = false;
  int device = -1;
  hipStream_t stream = nullptr;
  DeviceTables* d_tables = nullptr;
  uint64_t budget = 0;
  int kernel_choice = 0;
  DevBuf arena, io_in, io_out, jobs, results;
  std::vector<hipStream_t> side;       // extra streams: independent launch groups run concurrently
  Timing last{};
  int last_kind = 0;
  int jit_left = 0;                    // hipRTC compilations still allowed in the current call
  int cus = 256;                       // compute units of the device
};

Engine& eng() {
  static Engine e;
  return e;
}

int jit_budget() {
  if (const char* v = getenv("ZPAQ_AMD_MAX_JIT")) return atoi(v);
  return 16;
}

void require_ready(Engine& e) {
  if (!e.ready) {
    // lazy default init: LOCAL_RANK selects the device (one process per GPU)
    int dev = 0;
    if (const char* lr = getenv("LOCAL_RANK")) dev = atoi(lr);
    engine_init_locked(dev);
  }
}

}  // namespace

void engine_init_locked(int device) {
  Engine& e = eng();
  if (e.ready && e.device == device) return;
  int count = 0;
  hipError_t err = hipGetDeviceCount(&count);
  if (err != hipSuccess || count <= 0)
    fail(ZPQ_E_DEVICE, "no HIP device available (the modelled path has no CPU fallback)");
  if (device < 0 || device >= count) device = device % count;
  HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    fail(ZPQ_E_DEVICE, std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 only");
  e.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (!e.stream) HIP_CHECK(hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking));
  // constant tables
  const Tables& t = tables();
  static DeviceTables host_tb;
  memcpy(host_tb.stretch, t.stretch, sizeof(host_tb.stretch));
  memcpy(host_tb.squash, t.squash, sizeof(host_tb.squash));
  memcpy(host_tb.dt, t.dt, sizeof(host_tb.dt));
  memcpy(host_tb.dt2k, t.dt2k, sizeof(host_tb.dt2k));
  memcpy(host_tb.ns, t.ns, sizeof(host_tb.ns));
  memcpy(host_tb.icm_init, t.icm_init, sizeof(host_tb.icm_init));
  memcpy(host_tb.isse_init, t.isse_init, sizeof(host_tb.isse_init));
  memcpy(host_tb.sse_row, t.sse_row, sizeof(host_tb.sse_row));
  if (!e.d_tables) HIP_CHECK(hipMalloc((void**)&e.d_tables, sizeof(DeviceTables)));
  HIP_CHECK(hipMemcpy(e.d_tables, &host_tb, sizeof(DeviceTables), hipMemcpyHostToDevice));
  size_t free_b = 0, total_b = 0;
  HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
  if (!e.budget) e.budget = (uint64_t)(free_b * 0.85);
  e.device = device;
  e.ready = true;
}

void engine_init(int device) {
  std::lock_guard<std::mutex> g(eng().mu);
  engine_init_locked(device);
}

int engine_device_count() {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

void engine_shutdown() {
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  if (!e.ready) return;
  (void)hipSetDevice(e.device);
  (void)hipStreamSynchronize(e.stream);
  e.arena.release(); e.io_in.release(); e.io_out.release(); e.jobs.release(); e.results.release();
  if (e.d_tables) (void)hipFree(e.d_tables);
  e.d_tables = nullptr;
  if (e.stream) (void)hipStreamDestroy(e.stream);
  e.stream = nullptr;
  e.ready = false;
}

void engine_set_budget(uint64_t bytes) { std::lock_guard<std::mutex> g(eng().mu); eng().budget = bytes; }
void engine_set_kernel(int which) { std::lock_guard<std::mutex> g(eng().mu); eng().kernel_choice = which; }
Timing engine_last_timing() { std::lock_guard<std::mutex> g(eng().mu); return eng().last; }

static const uint8_t* plan_on_device(Engine& e, const zpq_plan* plan) {
  zpq_plan* p = const_cast<zpq_plan*>(plan);
  if (p->d_blob && p->d_device == e.device) return (const uint8_t*)p->d_blob;
  void* d = nullptr;
  HIP_CHECK(hipMalloc(&d, p->blob.size()));
  HIP_CHECK(hipMemcpy(d, p->blob.data(), p->blob.size(), hipMemcpyHostToDevice));
  p->d_blob = d;
  p->d_device = e.device;
  return (const uint8_t*)d;
}

void engine_plan_release(zpq_plan* p) {
  if (p && p->d_blob) { (void)hipFree(p->d_blob); p->d_blob = nullptr; }
  spec_kernel_release(p);
}

// Which kernel codes a plan: 3 = per-header specialised kernel, 2 = generic
// wave kernel, 1 = generic one-lane kernel.  kernel_choice 0 picks the best
// available; 1/2/3 force one (3 fails loudly if specialisation is unavailable).
// `dense` = the launch holds more blocks than one wavefront per SIMD can take (4 x CUs): then the
// 8-blocks-per-workgroup shape (two wavefronts per SIMD, half the side tables in LDS) has the higher
// throughput; below that the 4-block shape (everything in LDS, one workgroup per CU) is faster.
static int kernel_kind(Engine& e, const zpq_plan* plan, bool dense, SpecKernel** spec_out = nullptr) {
  zpq_plan* p = const_cast<zpq_plan*>(plan);
  if (spec_out) *spec_out = nullptr;
  const int want = e.kernel_choice;
  if (want == 1) return 1;
  if (!plan->hdr().wave_ok) {
    if (want >= 2) fail(ZPQ_E_UNSUPPORTED, "wave kernels need n <= 64 components");
    return 1;
  }
  if (want == 2) return 2;
  // Each unseen header costs a ~2-4 s hipRTC compile.  A batch whose blocks all carry different
  // (data-dependent) chains must not spend minutes compiling: a few per call, the rest run on the
  // generic wave kernel this time and are picked up by later calls.
  const int forced = spec_variant_forced();
  const int first = forced >= 0 ? forced : (dense ? 1 : 0);
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (attempt == 1 && forced >= 0) break;                                     // a forced shape has no fallback
    const int variant = attempt == 0 ? first : 1 - first;
    if (attempt == 1 && p->spec_state[variant] <= 0) break;                     // fall back only to a shape already loaded
    bool did = false;
    SpecKernel* k = spec_kernel_for(p, variant, want == 3 || e.jit_left > 0, nullptr, &did);
    if (did && e.jit_left > 0) --e.jit_left;
    if (k) { if (spec_out) *spec_out = k; return 3; }
  }
  if (want == 3) fail(ZPQ_E_UNSUPPORTED, "specialised kernel unavailable: " + p->spec_note);
  return 2;
}

int engine_plan_kernel_kind(zpq_plan* p, std::string& note) {
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e);
  HIP_CHECK(hipSetDevice(e.device));
  e.jit_left = jit_budget();
  const int k = kernel_kind(e, p, false);
  note = p->spec_note;
  return k;
}

struct LaunchGroup { int kind; SpecKernel* spec; uint32_t first, count; };

static hipError_t launch_spec(SpecKernel* k, bool decode, const BlockJob* d_jobs, BlockResult* d_res, uint32_t n,
                              const DeviceTables* d_tb, hipStream_t st) {
  void* args[4] = {(void*)&d_jobs, (void*)&d_res, (void*)&n, (void*)&d_tb};
  const uint32_t w = (uint32_t)k->waves;
  return hipModuleLaunchKernel(decode ? k->decode : k->encode, (n + w - 1) / w, 1, 1, 64 * w, 1, 1, 0, st, args, nullptr);
}

// Launch init + coding kernels for jobs already resident on the device, grouped
// so that each group is one launch.
static void launch_all(Engine& e, bool decode, const BlockJob* d_jobs, BlockResult* d_res,
                       const std::vector<LaunchGroup>& groups, uint32_t nb, uint64_t max_arena, hipStream_t st,
                       bool timed) {
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  if (timed) for (auto& x : ev) HIP_CHECK(hipEventCreate(&x));
  // enough 256-thread groups per block to stream the arena at HBM rate
  uint64_t per = max_arena / (256 * 16 * 8) + 1;
  uint32_t chunks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(per, 1), 64);
  if ((uint64_t)chunks * nb > 16384) chunks = (uint32_t)std::max<uint64_t>(1, 16384 / nb);
  if (timed) HIP_CHECK(hipEventRecord(ev[0], st));
  HIP_CHECK(launch_init_arena(d_jobs, nb, e.d_tables, chunks, st));
  if (timed) HIP_CHECK(hipEventRecord(ev[1], st));
  if (timed) HIP_CHECK(hipEventRecord(ev[2], st));
  // Groups (one per kernel kind / plan) are independent: fan them out over side streams so a
  // batch mixing several chains does not serialise one launch after the other.
  const size_t nside = groups.size() > 1 ? std::min<size_t>(groups.size() - 1, 7) : 0;
  while (e.side.size() < nside) {
    hipStream_t s2;
    HIP_CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    e.side.push_back(s2);
  }
  hipEvent_t fork = nullptr;
  if (nside) {
    HIP_CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(fork, st));
  }
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    const LaunchGroup& g = groups[gi];
    hipStream_t gs = (nside && gi % (nside + 1)) ? e.side[gi % (nside + 1) - 1] : st;
    if (gs != st && gi <= nside) HIP_CHECK(hipStreamWaitEvent(gs, fork, 0));
    // every kernel writes res[job.res_slot] relative to the SAME results base
    if (g.kind == 3) HIP_CHECK(launch_spec(g.spec, decode, d_jobs + g.first, d_res, g.count, e.d_tables, gs));
    else if (g.kind == 2) HIP_CHECK(launch_code_wave(decode, d_jobs + g.first, d_res, g.count, e.d_tables, gs));
    else HIP_CHECK(launch_code_serial(decode, d_jobs + g.first, d_res, g.count, e.d_tables, gs));
  }
  for (size_t k = 0; k < nside; ++k) {          // join the side streams back into `st`
    hipEvent_t done;
    HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(done, e.side[k]));
    HIP_CHECK(hipStreamWaitEvent(st, done, 0));
    HIP_CHECK(hipEventDestroy(done));
  }
  if (fork) HIP_CHECK(hipEventDestroy(fork));
  if (timed) {
    HIP_CHECK(hipEventRecord(ev[3], st));
    HIP_CHECK(hipEventSynchronize(ev[3]));
    float a = 0, b = 0;
    HIP_CHECK(hipEventElapsedTime(&a, ev[0], ev[1]));
    HIP_CHECK(hipEventElapsedTime(&b, ev[2], ev[3]));
    e.last.init_ms = a;
    e.last.code_ms = b;
    e.last.blocks = nb;
    for (auto& x : ev) (void)hipEventDestroy(x);
  }
}

void engine_code_host(bool decode, const std::vector<HostBlock>& blocks, std::vector<BlockResult>& results) {
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e);
  HIP_CHECK(hipSetDevice(e.device));
  const size_t nb = blocks.size();
  e.jit_left = jit_budget();
  results.assign(nb, BlockResult{0, 0, 0, 0});
  size_t pos = 0;
  e.last = Timing{};
  while (pos < nb) {
    // one residency wave: as many blocks as fit the state budget
    uint64_t need = 0, in_bytes = 0, out_bytes = 0, max_arena = 0;
    size_t end = pos;
    while (end < nb) {
      const HostBlock& hb = blocks[end];
      uint64_t a = hb.plan->hdr().arena_bytes;
      if (end > pos && need + a > e.budget) break;
      need += a;
      max_arena = std::max(max_arena, a);
      in_bytes += ((uint64_t)hb.in_len + hb.prefix_len + 63) & ~63ull;
      out_bytes += ((uint64_t)hb.out_cap + 63) & ~63ull;
      ++end;
    }
    if (need > e.budget) fail(ZPQ_E_NOMEM, "Out of memory: one block's model state exceeds the device budget");
    const size_t cnt = end - pos;
    e.arena.ensure(need);
    e.io_in.ensure(in_bytes + 64);
    e.io_out.ensure(out_bytes + 64);
    e.jobs.ensure(cnt * sizeof(BlockJob));
    e.results.ensure(cnt * sizeof(BlockResult));
    // order jobs so that every (kernel kind, plan) group is one contiguous launch
    std::vector<size_t> order;
    order.reserve(cnt);
    for (size_t i = pos; i < end; ++i) order.push_back(i);
    std::vector<int> kind_of(nb, 0);
    std::vector<SpecKernel*> spec_of(nb, nullptr);
    const bool dense = cnt > (size_t)4 * e.cus;
    for (size_t i = pos; i < end; ++i) kind_of[i] = kernel_kind(e, blocks[i].plan, dense, &spec_of[i]);
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
      if (kind_of[x] != kind_of[y]) return kind_of[x] > kind_of[y];
      if (kind_of[x] == 3) return blocks[x].plan < blocks[y].plan;
      return false;
    });
    std::vector<LaunchGroup> groups;
    for (size_t k = 0; k < cnt; ++k) {
      const zpq_plan* pl = blocks[order[k]].plan;
      const int kd = kind_of[order[k]];
      SpecKernel* sk = spec_of[order[k]];
      (void)pl;
      if (!groups.empty() && groups.back().kind == kd && groups.back().spec == sk) ++groups.back().count;
      else groups.push_back(LaunchGroup{kd, sk, (uint32_t)k, 1});
    }
    std::vector<BlockJob> jobs(cnt);
    std::vector<uint8_t> stage(in_bytes + 64);
    std::vector<uint64_t> out_off(cnt);
    uint64_t a_off = 0, i_off = 0, o_off = 0;
    for (size_t k = 0; k < cnt; ++k) {
      const HostBlock& hb = blocks[order[k]];
      BlockJob& j = jobs[k];
      memset(&j, 0, sizeof(j));
      j.plan = plan_on_device(e, hb.plan);
      j.arena = (uint8_t*)e.arena.p + a_off;
      j.in = (const uint8_t*)e.io_in.p + i_off;
      j.out = (uint8_t*)e.io_out.p + o_off;
      j.in_len = hb.in_len + hb.prefix_len;
      j.out_cap = hb.out_cap;
      j.res_slot = (uint32_t)k;
      if (hb.prefix_len) memcpy(stage.data() + i_off, hb.prefix, hb.prefix_len);
      if (hb.in_len) memcpy(stage.data() + i_off + hb.prefix_len, hb.in, hb.in_len);
      out_off[k] = o_off;
      a_off += hb.plan->hdr().arena_bytes;
      i_off += ((uint64_t)hb.in_len + hb.prefix_len + 63) & ~63ull;
      o_off += ((uint64_t)hb.out_cap + 63) & ~63ull;
    }
    HIP_CHECK(hipMemcpyAsync(e.io_in.p, stage.data(), in_bytes, hipMemcpyHostToDevice, e.stream));
    HIP_CHECK(hipMemcpyAsync(e.jobs.p, jobs.data(), cnt * sizeof(BlockJob), hipMemcpyHostToDevice, e.stream));
    Timing before = e.last;
    launch_all(e, decode, (const BlockJob*)e.jobs.p, (BlockResult*)e.results.p, groups, (uint32_t)cnt, max_arena,
               e.stream, true);
    e.last.init_ms += before.init_ms;
    e.last.code_ms += before.code_ms;
    e.last.blocks += before.blocks;
    std::vector<BlockResult> res(cnt);
    HIP_CHECK(hipMemcpyAsync(res.data(), e.results.p, cnt * sizeof(BlockResult), hipMemcpyDeviceToHost, e.stream));
    HIP_CHECK(hipStreamSynchronize(e.stream));
    for (size_t k = 0; k < cnt; ++k) {
      const HostBlock& hb = blocks[order[k]];
      results[order[k]] = res[k];
      uint32_t got = std::min(res[k].out_len, hb.out_cap);
      if (got && hb.out)
        HIP_CHECK(hipMemcpyAsync(hb.out, (const uint8_t*)e.io_out.p + out_off[k], got, hipMemcpyDeviceToHost, e.stream));
    }
    HIP_CHECK(hipStreamSynchronize(e.stream));
    pos = end;
  }
}

void engine_code_device(bool decode, const zpq_plan* const* plans, bool one_plan, const void* d_in,
                        const uint64_t* in_off, const uint32_t* in_len, uint32_t nblocks, void* d_out,
                        const uint64_t* out_off, const uint32_t* out_cap, BlockResult* d_res, void* stream, bool timed) {
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e);
  HIP_CHECK(hipSetDevice(e.device));
  hipStream_t st = stream ? (hipStream_t)stream : e.stream;
  e.jit_left = jit_budget();
  auto plan_of = [&](uint32_t b) { return one_plan ? plans[0] : plans[b]; };
  uint64_t need = 0, max_arena = 0;
  for (uint32_t b = 0; b < nblocks; ++b) {
    const uint64_t a = plan_of(b)->hdr().arena_bytes;
    need += a;
    max_arena = std::max(max_arena, a);
  }
  if (need > e.budget) fail(ZPQ_E_NOMEM, "Out of memory: batch state exceeds the device budget (split the batch)");
  e.arena.ensure(need);
  e.jobs.ensure((size_t)nblocks * sizeof(BlockJob));
  // group blocks by (kernel kind, plan); results keep the caller's block order through res_slot
  std::vector<uint32_t> order(nblocks);
  std::vector<int> kind_of(nblocks);
  std::vector<SpecKernel*> spec_of(nblocks, nullptr);
  const bool dense = nblocks > 4u * (uint32_t)e.cus;
  for (uint32_t b = 0; b < nblocks; ++b) { order[b] = b; kind_of[b] = kernel_kind(e, plan_of(b), dense, &spec_of[b]); }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    if (kind_of[x] != kind_of[y]) return kind_of[x] > kind_of[y];
    if (kind_of[x] == 3) return plan_of(x) < plan_of(y);
    return false;
  });
  std::vector<BlockJob> jobs(nblocks);
  std::vector<LaunchGroup> groups;
  uint64_t a_off = 0;
  for (uint32_t k = 0; k < nblocks; ++k) {
    const uint32_t b = order[k];
    const zpq_plan* pl = plan_of(b);
    BlockJob& j = jobs[k];
    memset(&j, 0, sizeof(j));
    j.plan = plan_on_device(e, pl);
    j.arena = (uint8_t*)e.arena.p + a_off;
    j.in = (const uint8_t*)d_in + in_off[b];
    j.out = (uint8_t*)d_out + out_off[b];
    j.in_len = in_len[b];
    j.out_cap = out_cap[b];
    j.res_slot = b;
    a_off += pl->hdr().arena_bytes;
    SpecKernel* sk = spec_of[b];
    if (!groups.empty() && groups.back().kind == kind_of[b] && groups.back().spec == sk) ++groups.back().count;
    else groups.push_back(LaunchGroup{kind_of[b], sk, k, 1});
  }
  HIP_CHECK(hipMemcpyAsync(e.jobs.p, jobs.data(), (size_t)nblocks * sizeof(BlockJob), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipStreamSynchronize(st));   // jobs vector goes out of scope below
  e.last = Timing{};
  e.last_kind = groups.empty() ? 0 : groups[0].kind;
  launch_all(e, decode, (const BlockJob*)e.jobs.p, d_res, groups, nblocks, max_arena, st, timed);
}

int engine_selftest(int32_t out[8]) {
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e);
  HIP_CHECK(hipSetDevice(e.device));
  int32_t* d = nullptr;
  HIP_CHECK(hipMalloc((void**)&d, 8 * sizeof(int32_t)));
  HIP_CHECK(hipMemsetAsync(d, 0, 8 * sizeof(int32_t), e.stream));
  HIP_CHECK(launch_selftest(d, e.stream));
  HIP_CHECK(hipMemcpyAsync(out, d, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, e.stream));
  HIP_CHECK(hipStreamSynchronize(e.stream));
  (void)hipFree(d);
  return 0;
}

}  // namespace zpq
