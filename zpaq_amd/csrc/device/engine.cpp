// Batch engine: owns the device, the HBM arena pool and the launch sequence
// (init_arena -> code_*).  One process drives one GPU; callers are serialised.
#include "engine.hpp"

#include <hip/hip_runtime_api.h>
#include <sched.h>

#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <functional>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <thread>

#include "../host/codegen.hpp"
#include "kernels.h"
#include "sa_kernels.h"
#include "spec_loader.hpp"

// The pipelined encoder keeps seven HIP streams busy (one per kernel plus the caller's).  The runtime maps streams onto
// GPU_MAX_HW_QUEUES hardware queues (default 4) and a queue runs its packets in order, so with the default two of the six
// kernels of a step wait for each other although nothing orders them: measured on the MI355X, -m3 on 256 blocks 766 ->
// 493 ms, -m5 on 64 blocks 3.0 -> 2.1 s with 8 queues, no change for batches that fill the machine (profiles/r03).  The
// variable is read when the HIP runtime initialises, so it is set when this library is loaded -- before the process's
// first HIP call in a C++ host such as zpaq; a host that initialised HIP earlier (python with torch) sets it itself
// (bench.py, tests/conftest.py do).  A value the user exported is left alone.
__attribute__((constructor)) static void zpq_more_hardware_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

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

// Page-locked host memory (ZPAQ_AMD_PINNED_STAGE): the staging buffer of host-buffer batches, DMA-able at link speed
struct HostPinned {
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t n) {                 // false: not available (the caller stages through pageable memory)
    if (n <= cap) return true;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    size_t want = (n + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1);
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
    cap = want;
    return true;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// hipEvent_t that cannot leak when a HIP_CHECK throws between create and destroy
struct Event {
  hipEvent_t ev = nullptr;
  explicit Event(bool timing = false) {
    HIP_CHECK(hipEventCreateWithFlags(&ev, timing ? hipEventDefault : hipEventDisableTiming));
  }
  ~Event() { if (ev) (void)hipEventDestroy(ev); }
  Event(const Event&) = delete;
  Event& operator=(const Event&) = delete;
  operator hipEvent_t() const { return ev; }
};

// compute units of the device the batches run on (set when an engine is created; read without the engine's lock by the mode
// choice and the submission queue: atomic.  Engines on devices of different sizes -- partition modes -- share the figure of
// the one created last; a residency estimate that is off is caught by the arrival handshake of the persistent launch)
static std::atomic<int> g_cus_hint{256};

struct Engine {
  std::mutex mu;
  bool ready = false;
  int device = -1;                     // HIP device this engine drives
  int slot = 0;                        // its index in g_engines = the slot of every plan's per-engine state (zpq_plan::dev[])
  hipStream_t stream = nullptr;
  DeviceTables* d_tables = nullptr;
  uint64_t budget = 0;
  unsigned sharers = 1;          // engines configured on this engine's physical device (ZPAQ_AMD_DEVICES may name one twice)
  int kernel_choice = 0;
  DevBuf arena, io_in, io_out, jobs, results;
  DevBuf segs;                         // segment tables of multi-segment blocks
  DevBuf sha_jobs, sha_out;            // SHA-1 of the staged inputs (sha1_blocks_kernel)
  DevBuf pipe;                         // stream buffers of the pipelined encoder (device/pipe_kernel.h)
  DevBuf pipe_ctl;                     // the persistent launch's progress counters, chunk counts and abort words (device/pipe_persist.h)
  HostPinned pin_in;                   // page-locked staging of host inputs, kept between calls (ZPAQ_AMD_PINNED_STAGE=0: pageable)
  HostPinned pin_out;                  // ... and of the outputs of a large batch
  hipStream_t pstream[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // one per pipe kernel
  std::vector<hipStream_t> side;       // extra streams: independent launch groups run concurrently
  Timing last{};
  int last_kind = 0;
  bool last_persist = false;           // the last batch's pipelined groups ran as persistent launches
  double last_persist_abort_ms = 0.0;  // ... or: how long it took until a persistent launch of the last batch was given up (0: none was)
  int jit_left = 0;                    // hipRTC compilations still allowed in the current call
  int cus = 256;                       // compute units of the device
  hipEvent_t busy = nullptr;           // recorded after the last launch of a call that returned with work in flight
};

// One engine per configured GPU, each in its own SLOT of g_engines.  `primary` serves the device-resident entry points
// and every batch when only one engine is configured; host-buffer batches are sharded over `ids` (the slots in use:
// contiguous block ranges, one thread and one engine each, no collective: blocks are independent, libzpaq.h:57-59).
// Default: one engine for the device LOCAL_RANK names (one process per GPU under torch.distributed / zpaq_amd.dist), in
// the slot of that number; zpq_init(-1) or ZPAQ_AMD_DEVICES=all|0,1,.. configures more -- slot i drives the i-th device
// named, and a device may be named twice (two engines, two streams sets, two arenas on one GPU: how the sharding path is
// exercised on a one-GPU box).
struct DeviceSet {
  std::mutex mu;
  std::vector<int> ids;                // slots in use
  int dev_of[zpq_plan::kMaxDevices];   // slot -> HIP device (-1: the slot's own number)
  int primary = -1;
  DeviceSet() { for (int& d : dev_of) d = -1; }
  uint64_t budget_override = 0;
  int kernel_choice = 0;
};
DeviceSet& devset() { static DeviceSet d; return d; }
Engine g_engines[zpq_plan::kMaxDevices];

int default_device() {
  if (const char* lr = getenv("LOCAL_RANK")) return atoi(lr);
  return 0;
}

int primary_device() {
  DeviceSet& d = devset();
  std::lock_guard<std::mutex> g(d.mu);
  if (d.primary < 0) {
    d.primary = default_device();
    if (d.primary < 0 || d.primary >= zpq_plan::kMaxDevices) d.primary = 0;
  }
  return d.primary;
}

Engine& eng(int slot = -1) {
  if (slot < 0) slot = primary_device();
  return g_engines[slot];
}

int device_of_slot(int slot) {
  DeviceSet& d = devset();
  std::lock_guard<std::mutex> g(d.mu);
  return d.dev_of[slot] >= 0 ? d.dev_of[slot] : slot;
}

// binds the calling thread to an engine: HIP's current device and the plan slot the loaders address
void bind_slot(int device, int slot) {
  HIP_CHECK(hipSetDevice(device));
  set_plan_device_index(slot);
}
void bind_device(Engine& e) { bind_slot(e.device, e.slot); }

// hipRTC compilations one call may spend on headers nobody compiled before (the rest of that call's unseen headers
// run on the generic kernel and are picked up by later calls), and the host threads that compile side by side
int jit_budget() {
  if (const char* v = getenv("ZPAQ_AMD_MAX_JIT")) return atoi(v);
  return 64;
}
int jit_threads() {
  return (int)std::max(1u, std::min(usable_cpus(), 16u));
}

void engine_init_device(Engine& e, int slot, int device);
void require_ready(Engine& e, int slot = -1) {
  if (slot < 0) slot = primary_device();
  if (!e.ready) engine_init_device(e, slot, device_of_slot(slot));          // lazy default init
}

}  // namespace

namespace {
void engine_init_device(Engine& e, int slot, int device) {
  if (e.ready && e.device == device && e.slot == slot) return;
  int count = 0;
  hipError_t err = hipGetDeviceCount(&count);
  if (err != hipSuccess || count <= 0)
    fail(ZPQ_E_DEVICE, "no HIP device available (the modelled path has no CPU fallback)");
  if (device < 0 || device >= count) fail(ZPQ_E_DEVICE, "no such HIP device: " + std::to_string(device));
  bind_slot(device, slot);
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    fail(ZPQ_E_DEVICE, std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 only");
  e.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  g_cus_hint = e.cus;
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
  memcpy(host_tb.stretch_cb, t.stretch_cb, sizeof(host_tb.stretch_cb));
  memcpy(host_tb.stretch_top, t.stretch_top, sizeof(host_tb.stretch_top));
  if (!e.d_tables) HIP_CHECK(hipMalloc((void**)&e.d_tables, sizeof(DeviceTables)));
  HIP_CHECK(hipMemcpy(e.d_tables, &host_tb, sizeof(DeviceTables), hipMemcpyHostToDevice));
  size_t free_b = 0, total_b = 0;
  HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
  // engines that share one physical device (ZPAQ_AMD_DEVICES=0,0,..) share its memory: each budgets its part of what is free
  unsigned sharers = 0;
  { DeviceSet& d = devset(); std::lock_guard<std::mutex> g(d.mu); e.budget = d.budget_override; e.kernel_choice = d.kernel_choice;
    for (int x : d.dev_of) sharers += x == device; }
  e.sharers = sharers ? sharers : 1u;
  if (!e.budget) e.budget = (uint64_t)(free_b * 0.85 / e.sharers);
  e.device = device;
  e.slot = slot;
  e.ready = true;
}
}  // namespace

// Contiguous share of `n` blocks for shard `k` of `parts` (same rule as zpaq_amd.dist.shard_range).
void engine_shard_range(uint64_t n, uint32_t parts, uint32_t k, uint64_t* lo, uint64_t* hi) {
  *lo = n * k / parts;
  *hi = n * (k + 1) / parts;
}

// device >= 0: drive that one GPU.  device == -1: every visible GPU, or the list in ZPAQ_AMD_DEVICES ("all" | "0,2,3").
void engine_init(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    fail(ZPQ_E_DEVICE, "no HIP device available (the modelled path has no CPU fallback)");
  std::vector<int> ids, devs;            // slots and the device each drives
  const char* env = getenv("ZPAQ_AMD_DEVICES");
  if (device >= 0 && !(env && env[0])) { ids.push_back(device % count); devs.push_back(device % count); }
  else if (!env || !env[0] || !strcmp(env, "all")) { for (int i = 0; i < count && i < zpq_plan::kMaxDevices; ++i) { ids.push_back(i); devs.push_back(i); } }
  else {
    // an explicit list: slot i drives the i-th device named (a device may be named more than once)
    for (const char* p = env; *p;) {
      if (*p >= '0' && *p <= '9') {
        const int v = atoi(p);
        if (v < count && (int)ids.size() < zpq_plan::kMaxDevices) { ids.push_back((int)ids.size()); devs.push_back(v); }
        while (*p >= '0' && *p <= '9') ++p;
      } else ++p;
    }
    if (ids.empty()) fail(ZPQ_E_ARG, "ZPAQ_AMD_DEVICES names no usable device");
  }
  {
    DeviceSet& d = devset();
    std::lock_guard<std::mutex> g(d.mu);
    d.ids = ids;
    for (int& x : d.dev_of) x = -1;
    for (size_t i = 0; i < ids.size(); ++i) d.dev_of[ids[i]] = devs[i];
    d.primary = ids[0];
    for (size_t i = 0; i < ids.size(); ++i)
      if (device >= 0 && devs[i] == device % count) { d.primary = ids[i]; break; }
  }
  for (size_t i = 0; i < ids.size(); ++i) {
    Engine& e = g_engines[ids[i]];
    std::lock_guard<std::mutex> g(e.mu);
    if (e.ready && (e.device != devs[i] || e.slot != ids[i])) fail(ZPQ_E_ARG, "zpq_init: the engine of this slot already drives another device (zpq_shutdown first)");
    engine_init_device(e, ids[i], devs[i]);
  }
  { Engine& e = eng(); bind_device(e); }
}

int engine_count() {
  DeviceSet& ds = devset();
  std::lock_guard<std::mutex> g(ds.mu);
  return (int)ds.ids.size();
}

int engine_device_count() {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

void engine_shutdown() {
  for (int id = 0; id < zpq_plan::kMaxDevices; ++id) {
    Engine& e = g_engines[id];
    std::lock_guard<std::mutex> g(e.mu);
    if (!e.ready) continue;
    (void)hipSetDevice(e.device);
    (void)hipStreamSynchronize(e.stream);
    e.arena.release(); e.io_in.release(); e.io_out.release(); e.jobs.release(); e.results.release(); e.pipe.release(); e.sha_jobs.release(); e.sha_out.release(); e.segs.release(); e.pin_in.release(); e.pin_out.release(); e.pipe_ctl.release();
    for (auto& ps : e.pstream) { if (ps) (void)hipStreamDestroy(ps); ps = nullptr; }
    for (auto& ss : e.side) (void)hipStreamDestroy(ss);
    e.side.clear();
    if (e.busy) { (void)hipEventDestroy(e.busy); e.busy = nullptr; }
    if (e.d_tables) (void)hipFree(e.d_tables);
    e.d_tables = nullptr;
    if (e.stream) (void)hipStreamDestroy(e.stream);
    e.stream = nullptr;
    e.ready = false;
  }
}

// bytes of model state (+ stream buffers) one residency wave may hold; 0 = back to the default (85 % of what is free now)
void engine_set_budget(uint64_t bytes) {
  { DeviceSet& d = devset(); std::lock_guard<std::mutex> g(d.mu); d.budget_override = bytes; }
  for (Engine& e : g_engines) {
    std::lock_guard<std::mutex> g(e.mu);
    if (!e.ready) continue;
    if (bytes) { e.budget = bytes; continue; }
    size_t free_b = 0, total_b = 0;
    if (hipSetDevice(e.device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      e.budget = (uint64_t)((free_b / e.sharers + e.arena.cap + e.pipe.cap + e.io_in.cap + e.io_out.cap) * 0.85);     // (what the engine holds itself counts as free)
  }
}
void engine_set_kernel(int which) {
  { DeviceSet& d = devset(); std::lock_guard<std::mutex> g(d.mu); d.kernel_choice = which; }
  for (Engine& e : g_engines) { std::lock_guard<std::mutex> g(e.mu); e.kernel_choice = which; }
}
Timing engine_last_timing() { Engine& e = eng(); std::lock_guard<std::mutex> g(e.mu); return e.last; }
double engine_last_persist_abort_ms() { Engine& e = eng(); std::lock_guard<std::mutex> g(e.mu); return e.last_persist_abort_ms; }
bool engine_last_persistent() { Engine& e = eng(); std::lock_guard<std::mutex> g(e.mu); return e.last_persist; }

static const uint8_t* plan_on_device(Engine& e, const zpq_plan* plan) {
  zpq_plan* p = const_cast<zpq_plan*>(plan);
  zpq_plan::OnDevice& od = p->dev[e.slot];
  if (od.d_blob) return (const uint8_t*)od.d_blob;
  void* d = nullptr;
  HIP_CHECK(hipMalloc(&d, p->blob.size()));
  HIP_CHECK(hipMemcpy(d, p->blob.data(), p->blob.size(), hipMemcpyHostToDevice));
  od.d_blob = d;
  return (const uint8_t*)d;
}

void engine_plan_release(zpq_plan* p) {
  if (!p) return;
  int before = -1;
  (void)hipGetDevice(&before);
  for (int id = 0; id < zpq_plan::kMaxDevices; ++id) {
    zpq_plan::OnDevice& od = p->dev[id];
    if (!od.d_blob && !od.pipe[0] && !od.pipe[1] && !od.pipe[2] && !od.pipe[3] && !od.spec[0] && !od.spec[1] && !od.spec[2]) continue;
    if (hipSetDevice(g_engines[id].device >= 0 ? g_engines[id].device : id) != hipSuccess) continue;
    set_plan_device_index(id);
    if (od.d_blob) { (void)hipFree(od.d_blob); od.d_blob = nullptr; }
    spec_kernel_release(p);
  }
  if (before >= 0) (void)hipSetDevice(before);
  set_plan_device_index(primary_device());
}

// Which kernel codes a plan: 4 = pipelined encoder (compression only, device/pipe_kernel.h), 3 = per-header
// specialised wavefront kernel, 2 = generic wave kernel, 1 = generic one-lane kernel.  kernel_choice 0 picks the
// best available; 1..4 force one (3 / 4 fail loudly if the kernel is unavailable; 4 still decodes with 3).
// `dense` = the launch holds more blocks than one wavefront per SIMD can take (4 x CUs): then the
// 8-blocks-per-workgroup shape (two wavefronts per SIMD, half the side tables in LDS) has the higher
// throughput; below that the 4-block shape (everything in LDS, one workgroup per CU) is faster.
struct KernelPick { int kind = 0; SpecKernel* spec = nullptr; PipeKernel* pipe = nullptr; int mode = 0; };

// The pipelined encoder has two shapes per chain (host/codegen.hpp PipeOptions): a chain with few blocks in the batch is
// latency bound -- a step costs one wavefront's serial chain however empty the machine is -- and runs the units with a
// lane per bit position; a chain that fills the machine is bound by HBM transactions and runs the lane-per-block units,
// which issue fewer requests.  Measured crossover on the MI355X, -m5 / 1 MiB blocks, 8 hardware queues: latency mode is
// 1.40 x faster at 64 blocks, 1.07 x at 512, 0.95 x at 768, 0.86 x at 1024 (profiles/r03/call6_summary.txt).
// ZPAQ_AMD_PIPE_MODE=latency|throughput forces one (A/B, tests).
static const uint32_t kLatencyModeBlocks = 640;
static const uint32_t kLongStepBytes = 128u << 10;       // latency shape: blocks this long may take 2048-byte steps (codegen.hpp)
// ... when what the units of one step pass each other stays small: a step's streams (blocks x 2048 bytes x the chain's
// ctx / bh / p bytes per input byte) are written and read once within a few steps, and up to ~100 MB of them live in the
// 256 MB Infinity Cache instead of HBM.  Measured (profiles/r03/call10_summary.txt): -m3's n = 2 chain on 256 blocks
// (29 MB per step) 483 -> 440 ms; -m5 (588 B per byte) +4 % at 64 blocks (77 MB), +6 % at 512 (616 MB), -20 % at 640.
static const uint64_t kLongStepStreamBytes = 96ull << 20;
// Workgroups of a persistent launch of `groups` groups x `wpg` workgroups that one XCD gets (the dispatcher deals a launch's
// workgroups round-robin over the 8 XCDs; pipe_persist.h maps whole sets of 8 groups one group per XCD, the rest in launch order)
static uint32_t persist_xcd_share(uint64_t groups, uint64_t wpg) {
  if (groups >= 8) return (uint32_t)((groups / 8) * wpg + ((groups % 8) * wpg + 7) / 8);
  return (uint32_t)((groups * wpg + 7) / 8);
}
// persist_expected: the call will take the persistent launch if the chain can (it waits for its results: zpq_*_device with timed = 0
// returns with the work in flight and runs the step kernels) -- the shape is chosen for the launch form that will really run
static int pipe_mode_for(uint32_t blocks_of_plan, uint32_t longest_block, uint32_t stream_bytes_per_byte, const zpq_plan* plan = nullptr,
                         bool persist_expected = true) {
  bool latency = blocks_of_plan <= kLatencyModeBlocks;
  const char* pp_env = getenv("ZPAQ_AMD_PIPE_PERSIST");
  bool persist_off = (pp_env && !strcmp(pp_env, "0")) || !persist_expected;
  // With the persistent launch the shapes differ in how many workgroups a group of blocks needs (-m5: 14 against 8): the
  // latency shape is the faster one exactly while ALL its workgroups are resident together (measured, profiles/r05
  // call13: 512 blocks 268 MB/s against 187; beyond that -- 640 blocks: 280 workgroups -- it would need a second round,
  // which costs a whole block's serial time, and the throughput shape in one round wins: 768 blocks 264 MB/s, 1024: 350)
  {
    if (plan && !persist_off) {
      PipeLayout L1;
      std::string why;
      if (pipe_layout(*plan, pipe_options(1), L1, why) && L1.persist_ok) {
        const uint64_t groups = (blocks_of_plan + (uint32_t)L1.G - 1) / (uint32_t)L1.G;
        latency = groups * (uint64_t)L1.ps_wpg <= (uint64_t)g_cus_hint.load();
      } else persist_off = true;       // (a chain that cannot be packed: the step kernels, by round 4's rule)
    }
  }
  if (const char* m = getenv("ZPAQ_AMD_PIPE_MODE")) {
    if (!strcmp(m, "latency")) latency = true;
    if (!strcmp(m, "throughput")) latency = false;
  }
  if (!latency) return 0;
  // (long steps exist to spread the per-step launch cost; the persistent launch has none and takes the 512-byte shape)
  const bool long_steps = persist_off && longest_block >= kLongStepBytes && stream_bytes_per_byte &&
                          (uint64_t)blocks_of_plan * 2048u * stream_bytes_per_byte <= kLongStepStreamBytes;
  if (long_steps) return 2;
  // Variant 3: the latency shape with a wavefront per SIMD (twice the workgroups per group) while THOSE all fit the device
  // (round 6, call 29: -m5 on 64 / 128 / 256 blocks +17 / +20 / +26 %); a small chain's variant 1 is that shape already.
  static const bool wide_off = [] { const char* v = getenv("ZPAQ_AMD_PIPE_WIDE"); return v && v[0] == '0'; }();
  if (plan && !persist_off && !wide_off && !getenv("ZPAQ_AMD_PIPE_MODE")) {
    PipeLayout L1, L3;
    std::string why;
    if (pipe_layout(*plan, pipe_options(1), L1, why) && L1.persist_ok && pipe_layout(*plan, pipe_options(3), L3, why) && L3.persist_ok &&
        L3.ps_waves < L1.ps_waves) {
      const uint64_t groups = (blocks_of_plan + (uint32_t)L3.G - 1) / (uint32_t)L3.G;
      if (groups * (uint64_t)L3.ps_wpg <= (uint64_t)g_cus_hint.load()) return 3;
    }
  }
  return 1;
}
// bytes the units of a chain pass each other per input byte (ctx 4, bh 8, p 16 per stream); 0: no pipelined encoder
static uint32_t pipe_stream_bytes_per_byte(const zpq_plan* plan) {
  PipeLayout L;
  std::string why;
  if (!pipe_layout(*plan, pipe_options(1), L, why)) return 0;
  return (uint32_t)(L.nctx * 4 + L.nrow * 8 + L.n * 16);
}

// must_specialise: the plan has blocks of several segments in this batch; only the per-header kernels carry coder and model
// state across segments, so such a plan compiles whatever the batch's JIT budget says.
static const bool kAutoTeam = true;     // kernel choice 0 prefers the lockstep decoder for launches that fill the machine
static KernelPick kernel_kind(Engine& e, const zpq_plan* plan, bool dense, bool decode, int mode, bool must_specialise = false) {
  zpq_plan* p = const_cast<zpq_plan*>(plan);
  KernelPick r;
  const int want = e.kernel_choice;
  if (want == 1) { r.kind = 1; return r; }
  if (!plan->hdr().wave_ok) {
    if (want >= 2) fail(ZPQ_E_UNSUPPORTED, "wave kernels need n <= 64 components");
    r.kind = 1;
    return r;
  }
  if (want == 2) { r.kind = 2; return r; }
  // Each unseen header costs a hipRTC compile of several seconds.  A batch whose blocks all carry different
  // (data-dependent) chains must not spend minutes compiling: a few per call, the rest run on the
  // generic wave kernel this time and are picked up by later calls.
  if (!decode && (want == 0 || want == 4 || want == 5 || want == 6)) {      // (5 and 6 choose among the decoders only)
    bool did = false;
    PipeKernel* k = pipe_kernel_for(p, mode, want == 4 || must_specialise || e.jit_left > 0, &did);
    if (did && e.jit_left > 0) --e.jit_left;
    if (k) { r.kind = 4; r.pipe = k; r.mode = mode; return r; }
    if (want == 4) fail(ZPQ_E_UNSUPPORTED, "pipelined encoder unavailable: " + p->cur().pipe_note);
  }
  const int forced = spec_variant_forced();
  // Decoding a launch that fills the machine: two blocks per wavefront (device/spec_dual_kernel.h) where the chain allows
  // it (up to 32 components, no block of several segments).  zpq_set_kernel(5) forces it, 3 keeps one block per wavefront.
  // The lockstep decoder (device/spec_team_kernel.h) first: zpq_set_kernel(6) forces it.
  if (decode && !must_specialise && forced < 0 && (want == 6 || (want == 0 && dense && kAutoTeam)) && p->cur().spec_state[3] >= 0) {
    bool did = false;
    SpecKernel* k = spec_kernel_for(p, 3, want == 6 || e.jit_left > 0, nullptr, &did);
    if (did && e.jit_left > 0) --e.jit_left;
    if (k) { r.kind = 3; r.spec = k; return r; }
    if (want == 6) fail(ZPQ_E_UNSUPPORTED, "lockstep decoder unavailable: " + p->cur().spec_note);
  }
  if (decode && !must_specialise && forced < 0 && (want == 5 || (want == 0 && dense)) && p->cur().spec_state[2] >= 0) {
    bool did = false;
    SpecKernel* k = spec_kernel_for(p, 2, want == 5 || e.jit_left > 0, nullptr, &did);
    if (did && e.jit_left > 0) --e.jit_left;
    if (k) { r.kind = 3; r.spec = k; return r; }
    if (want == 5) fail(ZPQ_E_UNSUPPORTED, "decoder with two blocks per wavefront unavailable: " + p->cur().spec_note);
  }
  const int first = forced >= 0 ? forced : (dense ? 1 : 0);
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (attempt == 1 && forced >= 0) break;                                     // a forced shape has no fallback
    const int variant = attempt == 0 ? first : 1 - first;
    if (attempt == 1 && p->cur().spec_state[variant] <= 0) break;                     // fall back only to a shape already loaded
    bool did = false;
    SpecKernel* k = spec_kernel_for(p, variant, want >= 3 || must_specialise || e.jit_left > 0, nullptr, &did);
    if (did && e.jit_left > 0) --e.jit_left;
    if (k) { r.kind = 3; r.spec = k; return r; }
  }
  if (want >= 3) fail(ZPQ_E_UNSUPPORTED, "specialised kernel unavailable: " + p->cur().spec_note);
  r.kind = 2;
  return r;
}

int engine_plan_kernel_kind(zpq_plan* p, std::string& note, bool decode, uint32_t nblocks, uint32_t block_bytes) {
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e);
  bind_device(e);
  e.jit_left = jit_budget();
  const KernelPick k = kernel_kind(e, p, nblocks > (uint32_t)4 * e.cus, decode, pipe_mode_for(nblocks ? nblocks : 0xFFFFFFFFu, block_bytes, pipe_stream_bytes_per_byte(p), p));
  // (the note of the kernel that was PICKED: the plan's spec_note is the note of whichever shape was loaded last)
  if (k.kind == 3 && k.spec)
    note = k.spec->origin + (k.spec->encode ? "" : (k.spec->threads > 256 ? " (zpq_spec_decode3: row / mixer wavefronts in lockstep)"
                                                                          : " (zpq_spec_decode2: two blocks per wavefront)"));
  else note = k.kind == 4 ? p->cur().pipe_note : p->cur().spec_note;
  return k.kind;
}

// One launch group = consecutive jobs coded by the same kernel (and, for the per-header kernels, the same plan).
struct LaunchGroup {
  KernelPick pick;
  const zpq_plan* plan;
  uint32_t first, count;
  uint32_t max_len;          // longest input of the group (the pipelined encoder's step count)
};

static bool same_group(const LaunchGroup& g, const KernelPick& k, const zpq_plan* plan) {
  if (g.pick.kind != k.kind) return false;
  if (k.kind == 4) return g.pick.pipe == k.pipe && g.plan == plan;
  if (k.kind == 3) return g.pick.spec == k.spec;
  return true;
}

// bytes of stream buffer the pipelined encoder needs for `count` blocks of `plan`
static uint64_t pipe_bytes(const zpq_plan* plan, uint32_t count, int mode) {
  PipeLayout L;
  std::string why;
  if (!pipe_layout(*plan, pipe_options(mode), L, why)) return 0;
  return (uint64_t)((count + (uint32_t)L.G - 1) / (uint32_t)L.G) * L.group_bytes;
}

static hipError_t launch_spec(SpecKernel* k, bool decode, const BlockJob* d_jobs, BlockResult* d_res, uint32_t n,
                              const DeviceTables* d_tb, hipStream_t st) {
  void* args[4] = {(void*)&d_jobs, (void*)&d_res, (void*)&n, (void*)&d_tb};
  const uint32_t w = (uint32_t)k->waves;
  if (!decode && !k->encode) return hipErrorInvalidDeviceFunction;
  return hipModuleLaunchKernel(decode ? k->decode : k->encode, (n + w - 1) / w, 1, 1, (uint32_t)k->threads, 1, 1, 0, st, args, nullptr);
}

// The pipelined encoder: every step launches the six kernels of every pipe group on six streams (they are
// independent inside a step; they are separate kernels only because their static LDS needs differ), then joins the streams -- the step boundary is the only producer/consumer barrier.
struct PipeRun {
  PipeKernel* k;
  PipeArgs args;
  uint32_t grid[6];
  uint32_t nsteps;
  uint32_t ngroups;
  uint32_t threads;          // lanes per workgroup of every kernel but hcomp and mix (= blocks per group)
  uint32_t mix_threads;      // ... of the mix kernel (= threads unless its lanes are per bit position)
  uint32_t light_threads;    // ... of the light kernel (= threads unless some of its units have a lane per bit position)
  bool consumes[6][6];
  int slack;
  uint32_t chunk;            // input bytes per step
  // the persistent launch (device/pipe_persist.h)
  uint32_t group_blocks = 0; // blocks per group
  uint32_t ps_wpg = 0, ps_waves = 0, ps_nunit = 0;     // workgroups per group (0: the chain has none), wavefronts per workgroup, counters per group
  uint32_t ps_nslot = 0;
  std::vector<uint32_t> group_chunks;                  // per group: chunks of its longest block
};

// ZPAQ_AMD_PIPE_PROFILE=1: run every unit type of every step alone on one stream between two events and print the
// average time of each (stderr).  A measuring aid: the step's real duration is the slowest unit, not the sum.
static void launch_pipe_profiled(Engine& e, std::vector<PipeRun>& runs, hipStream_t st) {
  struct Rec { int kernel; uint32_t role; hipEvent_t a, b; };
  std::vector<Rec> recs;
  for (auto& r : runs) {
    for (uint32_t step = 0; step < r.nsteps; ++step) {
      r.args.step = (int32_t)step;
      for (int k = 0; k < 6; ++k) {
        if (!r.grid[k]) continue;
        const uint32_t per = k == 0 ? r.grid[0] : r.ngroups;      // workgroups of one unit type
        for (uint32_t w0 = 0; w0 < r.grid[k]; w0 += per) {
          r.args.wg0 = w0;
          void* args[1] = {(void*)&r.args};
          Rec rec{k, w0 / per, nullptr, nullptr};
          HIP_CHECK(hipEventCreate(&rec.a));
          HIP_CHECK(hipEventCreate(&rec.b));
          HIP_CHECK(hipEventRecord(rec.a, st));
          HIP_CHECK(hipModuleLaunchKernel(r.k->fn[k], std::min(per, r.grid[k] - w0), 1, 1, k == 0 ? 64u : (k == 5 ? r.mix_threads : (k == 2 ? r.light_threads : r.threads)), 1, 1, 0, st, args, nullptr));
          HIP_CHECK(hipEventRecord(rec.b, st));
          recs.push_back(rec);
        }
      }
      r.args.wg0 = 0;
    }
  }
  HIP_CHECK(hipStreamSynchronize(st));
  static const char* names[6] = {"hcomp", "rows", "light", "icm", "isse", "mix"};
  std::map<std::pair<int, uint32_t>, std::pair<double, int>> acc;
  for (auto& rec : recs) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, rec.a, rec.b);
    auto& x = acc[{rec.kernel, rec.role}];
    x.first += ms; x.second += 1;
    (void)hipEventDestroy(rec.a); (void)hipEventDestroy(rec.b);
  }
  for (auto& kv : acc)
    fprintf(stderr, "[zpq pipe profile] %-5s unit %2u: %8.3f ms per step (%d launches)\n", names[kv.first.first], kv.first.second,
            kv.second.first / kv.second.second, kv.second.second);
}

// The tail of a batch's input may still be on its way when the first steps are launched (engine_code_host_on: the units
// of step s read input bytes below (s + 1) x chunk only): `late` is called once, before the launches of the first step
// that may read bytes from `from_byte` on; it enqueues the rest of the copy and returns the event the kernels have to wait for.
struct LateInput {
  uint32_t from_byte = 0;                       // 0: nothing is late
  std::function<hipEvent_t()> arrive;
};

// ---- the persistent launch ------------------------------------------------------------------------------------
// ONE launch per run for the whole sequence: every unit of every group is a wavefront that walks its chunks and waits for
// the units it depends on through progress counters in HBM (device/pipe_persist.h).  All workgroups must be resident at
// once, so the engine launches at most what the device holds (occupancy API x compute units): a run with more groups goes
// in rounds, several runs go side by side when they fit together -- otherwise the six kernels code the batch.
// Returns false when the persistent path does not apply (nothing was launched).  A launch whose watchdog fired (a
// workgroup did not get a compute unit: something else held the GPU) sets *aborted: the caller re-initialises the arenas
// and runs the six kernels.
static uint32_t persist_timeout_ticks() {
  uint32_t ms = 3000;
  if (const char* t = getenv("ZPAQ_AMD_PERSIST_TIMEOUT_MS")) ms = (uint32_t)std::max(1, atoi(t));
  return (uint32_t)std::min<uint64_t>((uint64_t)ms * 100000ull, 0xFFFFFFF0ull);      // s_memrealtime: 100 MHz
}

// ... and how long the workgroups of a launch may wait for each other to become resident (pipe_persist.h pipe_arrived) before the
// launch is given up untouched: 20 ms without a new arrival (a full grid arrives within microseconds of its first workgroup;
// measured with 200 of 256 compute units held by another process, profiles/r06: the engine knows ~2 x this after the launch --
// the workgroups that were left in the queue still have to be dispatched, see the flag and leave)
static uint32_t persist_arrive_ticks() {
  uint32_t ms = 20;
  if (const char* t = getenv("ZPAQ_AMD_PERSIST_ARRIVE_MS")) ms = (uint32_t)std::max(1, atoi(t));
  return (uint32_t)std::min<uint64_t>((uint64_t)ms * 100000ull, 0xFFFFFFF0ull);
}

static bool persist_wanted(const std::vector<PipeRun>& runs, bool single_launch_batch) {
  const char* v = getenv("ZPAQ_AMD_PIPE_PERSIST");
  if (v && !strcmp(v, "0")) return false;
  if (getenv("ZPAQ_AMD_PIPE_PROFILE") || getenv("ZPAQ_AMD_PIPE_TRACE")) return false;
  if (!single_launch_batch) return false;
  for (const PipeRun& r : runs) if (!r.k->persist || !r.ps_wpg) return false;
  return !runs.empty();
}

// workgroups of a run's persistent kernel the device holds at once (0: unknown / none)
static uint32_t persist_capacity(Engine& e, PipeRun& r) {
  if (!r.k->persist_wg_per_cu) {
    int nb = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, r.k->persist, (int)(64 * r.ps_waves), 0) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
    r.k->persist_wg_per_cu = nb > 0 ? nb : -1;
  }
  return r.k->persist_wg_per_cu > 0 ? (uint32_t)r.k->persist_wg_per_cu * (uint32_t)e.cus : 0u;
}

// Engines that drive the SAME physical device (ZPAQ_AMD_DEVICES=0,0,..: how the sharding path runs on a one-GPU box) must not
// have persistent launches in flight together: each is sized to the whole device, and two half-resident launches would wait
// for each other's compute units until the watchdog ends both.
static std::mutex& persist_device_mutex(int device) {
  static std::mutex mu[64];
  return mu[(unsigned)device % 64u];
}

static bool launch_pipe_persist(Engine& e, std::vector<PipeRun>& runs, hipStream_t st, bool* aborted, std::string* what, bool* untouched) {
  *aborted = false;
  *untouched = false;
  std::lock_guard<std::mutex> device_turn(persist_device_mutex(e.device >= 0 ? e.device : e.slot));
  // what the device holds
  std::vector<uint32_t> cap(runs.size());
  for (size_t i = 0; i < runs.size(); ++i) {
    PipeRun& r = runs[i];
    cap[i] = persist_capacity(e, r);
    if (cap[i] < r.ps_wpg) return false;
  }
  // Several runs: only side by side (a second round costs a whole block's serial time whatever it holds).  The dispatcher deals
  // the workgroups of a launch round-robin over the 8 XCDs and does not look for room elsewhere, so what has to fit is every
  // XCD's share of every run.  (Measured with the archiver's batch of 14 + 2 groups, calls 24-27: sized against the device as a
  // whole -- 238 of 256 compute units -- the short run found 2 free compute units per XCD on six XCDs where it needed 3-4,
  // sat half resident until the long one ended, and the batch took the sum of both: 3.2 s instead of 1.7.)
  if (runs.size() > 1) {
    uint32_t share = 0, room = 0xFFFFFFFFu;
    for (size_t i = 0; i < runs.size(); ++i) {
      share += persist_xcd_share(runs[i].ngroups, runs[i].ps_wpg);
      room = std::min(room, cap[i] / 8);
    }
    if (share > room) return false;
  }
  // control block per run: [ctl 4 words][group_chunks ngroups][prog ngroups * nunit]
  uint64_t words = 0;
  std::vector<uint64_t> base(runs.size());
  for (size_t i = 0; i < runs.size(); ++i) {
    base[i] = words;
    words += 4 + runs[i].ngroups + (uint64_t)runs[i].ngroups * runs[i].ps_nunit;
    words = (words + 63) & ~63ull;
  }
  e.pipe_ctl.ensure(words * 4);
  std::vector<uint32_t> host(words, 0);
  for (size_t i = 0; i < runs.size(); ++i)
    for (uint32_t g = 0; g < runs[i].ngroups; ++g) host[base[i] + 4 + g] = runs[i].group_chunks[g];
  HIP_CHECK(hipMemcpyAsync(e.pipe_ctl.p, host.data(), words * 4, hipMemcpyHostToDevice, st));
  HIP_CHECK(hipStreamSynchronize(st));            // (host vector; and nothing else of this call may still occupy compute units)
  const uint32_t timeout = persist_timeout_ticks();
  // ZPAQ_AMD_PERSIST_PROF=<file>: where every unit wavefront's time went (waiting / working, 100 MHz ticks), first run only
  const char* prof_path = getenv("ZPAQ_AMD_PERSIST_PROF");
  unsigned long long* d_prof = nullptr;
  uint64_t prof_words = 0;
  uint32_t prof_nslot = 0;
  size_t prof_run = 0;                               // ZPAQ_AMD_PERSIST_PROF_RUN: which run of the batch (default the first)
  if (const char* v = getenv("ZPAQ_AMD_PERSIST_PROF_RUN")) prof_run = std::min<size_t>((size_t)atoi(v), runs.empty() ? 0 : runs.size() - 1);
  if (prof_path && prof_path[0] && !runs.empty()) {
    prof_nslot = runs[prof_run].ps_nslot;
    prof_words = 4ull * runs[prof_run].ngroups * prof_nslot;
    HIP_CHECK(hipMalloc((void**)&d_prof, prof_words * 8));
    HIP_CHECK(hipMemset(d_prof, 0, prof_words * 8));
  }
  while (e.side.size() + 1 < runs.size()) {
    hipStream_t s2;
    HIP_CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    e.side.push_back(s2);
  }
  Event fork;
  if (runs.size() > 1) HIP_CHECK(hipEventRecord(fork, st));
  std::vector<std::unique_ptr<Event>> joins;
  uint32_t rounds_max = 0;
  for (size_t i = 0; i < runs.size(); ++i) {
    PipeRun& r = runs[i];
    hipStream_t rs = i == 0 ? st : e.side[i - 1];
    if (i) HIP_CHECK(hipStreamWaitEvent(rs, fork, 0));
    uint32_t* ctl = (uint32_t*)e.pipe_ctl.p + base[i];
    const uint32_t most = std::max<uint32_t>(1u, cap[i] / r.ps_wpg);                  // groups that are resident together
    const uint32_t rounds = (r.ngroups + most - 1) / most;
    const uint32_t per_round = (r.ngroups + rounds - 1) / rounds;                     // (rounds of equal size)
    rounds_max = std::max(rounds_max, rounds);
    uint32_t arrived_before = 0;                                                      // workgroups of the run's earlier rounds
    for (uint32_t g0 = 0; g0 < r.ngroups; g0 += per_round) {
      PipeArgs a = r.args;
      a.step = 0; a.wg0 = 0;
      a.ctl = ctl;
      a.group_chunks = ctl + 4;
      a.prog = ctl + 4 + r.ngroups;
      a.group0 = g0;
      a.ngroups_here = std::min(per_round, r.ngroups - g0);
      a.timeout_ticks = timeout;
      a.spread = a.ngroups_here >= 8 ? 8u : 1u;
      if (const char* v = getenv("ZPAQ_AMD_PERSIST_SPREAD")) a.spread = atoi(v) == 8 ? 8u : 1u;      // (experiments)
      a.trace = i == prof_run ? d_prof : nullptr;
      const uint32_t grid = a.ngroups_here * r.ps_wpg;
      arrived_before += grid;
      a.arrive_need = arrived_before;
      a.arrive_ticks = persist_arrive_ticks();
      void* args[1] = {(void*)&a};
      HIP_CHECK(hipModuleLaunchKernel(r.k->persist, grid, 1, 1, 64u * r.ps_waves, 1, 1, 0, rs, args, nullptr));
    }
    if (i) {
      joins.emplace_back(new Event());
      HIP_CHECK(hipEventRecord(*joins.back(), rs));
      HIP_CHECK(hipStreamWaitEvent(st, *joins.back(), 0));
    }
  }
  HIP_CHECK(hipStreamSynchronize(st));
  if (d_prof) {
    std::vector<unsigned long long> hp(prof_words);
    HIP_CHECK(hipMemcpy(hp.data(), d_prof, prof_words * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_prof);
    if (FILE* f = fopen(prof_path, "wb")) {
      const unsigned long long hdr[4] = {runs[prof_run].ngroups, prof_nslot, runs[prof_run].ps_waves, runs[prof_run].ps_wpg};
      fwrite(hdr, 8, 4, f);
      fwrite(hp.data(), 8, prof_words, f);
      fclose(f);
    }
  }
  size_t n_untouched = 0;
  for (size_t i = 0; i < runs.size(); ++i) {
    uint32_t c[4] = {0, 0, 0, 0};
    HIP_CHECK(hipMemcpy(c, (uint32_t*)e.pipe_ctl.p + base[i], sizeof c, hipMemcpyDeviceToHost));
    if (c[0]) {
      *aborted = true;
      n_untouched += c[0] == 2u;
      if (what) *what = "run " + std::to_string(i) + (c[0] == 2u
          ? ": only " + std::to_string(c[3]) + " of its workgroups became resident together (something else holds compute units)"
          : ": the unit in slot " + std::to_string(c[1]) + " gave up waiting at chunk " + std::to_string(c[2]));
    }
  }
  // every run stopped at the arrival handshake of its FIRST round: no arena, stream or result has been written
  *untouched = n_untouched == runs.size() && rounds_max == 1;
  return true;
}

static void launch_pipe(Engine& e, std::vector<PipeRun>& runs, hipStream_t st, LateInput* late = nullptr) {
  if (runs.empty()) { if (late && late->from_byte) (void)late->arrive(); return; }
  if (getenv("ZPAQ_AMD_PIPE_PROFILE")) {
    if (late && late->from_byte) HIP_CHECK(hipStreamWaitEvent(st, late->arrive(), 0));
    launch_pipe_profiled(e, runs, st);
    return;
  }
  for (auto& ps : e.pstream)
    if (!ps) HIP_CHECK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  uint32_t nsteps = 0;
  for (auto& r : runs) nsteps = std::max(nsteps, r.nsteps);
  // Each kernel has its own in-order stream.  A launch of kernel k for step s waits only for what it really depends on:
  //   * the producers of the streams it reads, at step s-1 (every value it reads was written at an earlier step);
  //   * the consumers of the streams it writes, at step s-1-slack (the ring slot it is about to overwrite held a chunk
  //     whose last reader ran at the latest then): with `slack` extra ring slots a fast producer runs that many steps
  //     ahead of a slow consumer, so the run proceeds at the pace of the slowest KERNEL, not the sum of each step's slowest.
  bool cons[6][6] = {};
  int slack = 1 << 30;
  for (auto& r : runs) {
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) cons[a][b] = cons[a][b] || r.consumes[a][b];
    slack = std::min(slack, r.slack);
  }
  bool used[6] = {};
  for (auto& r : runs) for (int k = 0; k < 6; ++k) used[k] = used[k] || r.grid[k] != 0;
  // ZPAQ_AMD_PIPE_TRACE=<file> (with kernels built -DZPQ_TRACE, e.g. ZPAQ_AMD_SPEC_DEFS=-DZPQ_TRACE): one record of four
  // 64-bit words per workgroup and launch -- [kernel << 56 | step << 32 | workgroup], [XCC_ID << 32 | HW_ID], entry and exit
  // on the 100 MHz reference clock -- written to the file when the sequence has finished (profiles/pipe_trace.py)
  const char* trace_path = getenv("ZPAQ_AMD_PIPE_TRACE");
  unsigned long long* d_trace = nullptr;
  uint64_t trace_records = 0;
  std::vector<uint64_t> trace_run_base;                  // first record of each run
  if (trace_path && trace_path[0]) {
    for (auto& r : runs) {
      trace_run_base.push_back(trace_records);
      uint64_t per_step = 0;
      for (int k = 0; k < 6; ++k) per_step += r.grid[k];
      trace_records += per_step * r.nsteps;
    }
    if (trace_records >= (1ull << 32)) fail(ZPQ_E_ARG, "ZPAQ_AMD_PIPE_TRACE: too many launches for one trace");
    HIP_CHECK(hipMalloc((void**)&d_trace, trace_records * 32));
    HIP_CHECK(hipMemset(d_trace, 0, trace_records * 32));
  }
  const int R = slack + 2;                               // events kept per kernel
  std::vector<std::unique_ptr<Event>> ev[6];
  for (int k = 0; k < 6; ++k) for (int i = 0; i < R; ++i) ev[k].emplace_back(new Event());
  Event fork;
  HIP_CHECK(hipEventRecord(fork, st));
  for (int k = 0; k < 6; ++k) HIP_CHECK(hipStreamWaitEvent(e.pstream[k], fork, 0));
  uint32_t late_step = 0xFFFFFFFFu;
  if (late && late->from_byte) {
    uint32_t maxc = 1;
    for (auto& r : runs) maxc = std::max(maxc, r.chunk);
    late_step = late->from_byte / maxc;          // the first step whose units may read a byte at or beyond from_byte
  }
  for (uint32_t step = 0; step < nsteps; ++step) {
    if (step == late_step) {
      hipEvent_t arrived = late->arrive();
      for (int k = 0; k < 6; ++k) if (used[k]) HIP_CHECK(hipStreamWaitEvent(e.pstream[k], arrived, 0));
      late_step = 0xFFFFFFFFu;
    }
    for (int k = 0; k < 6; ++k) {
      if (!used[k]) continue;
      for (int p = 0; p < 6; ++p) {
        if (p == k || !used[p]) continue;
        if (cons[k][p] && step >= 1) HIP_CHECK(hipStreamWaitEvent(e.pstream[k], *ev[p][(step - 1) % R], 0));
        if (cons[p][k] && step >= 1u + (uint32_t)slack) HIP_CHECK(hipStreamWaitEvent(e.pstream[k], *ev[p][(step - 1 - slack) % R], 0));
      }
      for (auto& r : runs) {
        if (step >= r.nsteps || !r.grid[k]) continue;
        r.args.step = (int32_t)step;
        if (d_trace) {
          uint64_t per_step = 0, before = 0;
          for (int q = 0; q < 6; ++q) { if (q < k) before += r.grid[q]; per_step += r.grid[q]; }
          r.args.trace = d_trace;
          r.args.trace_base = (uint32_t)(trace_run_base[(size_t)(&r - &runs[0])] + per_step * step + before);
        }
        void* args[1] = {(void*)&r.args};
        HIP_CHECK(hipModuleLaunchKernel(r.k->fn[k], r.grid[k], 1, 1, k == 0 ? 64u : (k == 5 ? r.mix_threads : (k == 2 ? r.light_threads : r.threads)), 1, 1, 0, e.pstream[k], args, nullptr));
      }
      HIP_CHECK(hipEventRecord(*ev[k][step % R], e.pstream[k]));
    }
  }
  if (late_step != 0xFFFFFFFFu) HIP_CHECK(hipStreamWaitEvent(st, late->arrive(), 0));     // (fewer steps than expected)
  for (int k = 0; k < 6; ++k)
    if (used[k] && nsteps) HIP_CHECK(hipStreamWaitEvent(st, *ev[k][(nsteps - 1) % R], 0));
  if (d_trace) {
    HIP_CHECK(hipStreamSynchronize(st));
    std::vector<unsigned long long> host(trace_records * 4);
    HIP_CHECK(hipMemcpy(host.data(), d_trace, trace_records * 32, hipMemcpyDeviceToHost));
    (void)hipFree(d_trace);
    if (FILE* f = fopen(trace_path, "wb")) {
      fwrite(host.data(), 32, trace_records, f);
      fclose(f);
      fprintf(stderr, "[zpq pipe trace] %llu records -> %s\n", (unsigned long long)trace_records, trace_path);
    }
  }
}

// Launch init + coding kernels for jobs already resident on the device, grouped
// so that each group is one launch (or one launch sequence).
static void launch_all(Engine& e, bool decode, const BlockJob* d_jobs, BlockResult* d_res,
                       const std::vector<LaunchGroup>& groups, uint32_t nb, uint64_t max_arena, hipStream_t st,
                       bool timed, LateInput* late = nullptr, const BlockJob* h_jobs = nullptr,
                       const std::function<void()>* before_persist = nullptr) {
  Event ev0(true), ev1(true), ev2(true), ev3(true);
  // enough 256-thread groups per block to stream the arena at HBM rate
  uint64_t per = max_arena / (256 * 16 * 8) + 1;
  uint32_t chunks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(per, 1), 64);
  if ((uint64_t)chunks * nb > 16384) chunks = (uint32_t)std::max<uint64_t>(1, 16384 / nb);
  if (timed) HIP_CHECK(hipEventRecord(ev0, st));
  for (uint32_t b0 = 0; b0 < nb; b0 += 65535u)        // gridDim.y carries the block index
    HIP_CHECK(launch_init_arena(d_jobs + b0, std::min(nb - b0, 65535u), e.d_tables, chunks, st));
  if (timed) HIP_CHECK(hipEventRecord(ev1, st));
  if (timed) HIP_CHECK(hipEventRecord(ev2, st));
  // stream buffers of the pipe groups
  std::vector<PipeRun> runs;
  uint64_t pipe_need = 0;
  for (const LaunchGroup& g : groups)
    if (g.pick.kind == 4) pipe_need += pipe_bytes(g.plan, g.count, g.pick.mode);
  if (pipe_need) e.pipe.ensure(pipe_need);
  uint64_t pipe_off = 0;
  for (const LaunchGroup& g : groups) {
    if (g.pick.kind != 4) continue;
    PipeLayout L;
    std::string why;
    if (!pipe_layout(*g.plan, pipe_options(g.pick.mode), L, why)) fail(ZPQ_E_DEVICE, "pipe layout vanished: " + why);
    PipeRun r;
    r.k = g.pick.pipe;
    r.args = PipeArgs{d_jobs + g.first, d_res, g.count, e.d_tables, (uint8_t*)e.pipe.p + pipe_off, 0, 0u};
    const uint32_t ng = (g.count + (uint32_t)L.G - 1) / (uint32_t)L.G;
    r.ngroups = ng;
    r.threads = (uint32_t)L.G;
    r.mix_threads = (uint32_t)L.mix_threads();
    r.light_threads = (uint32_t)L.light_threads();
    memcpy(r.consumes, L.consumes, sizeof(r.consumes));
    r.slack = L.slack;
    r.chunk = (uint32_t)L.C;
    r.grid[0] = (g.count + (uint32_t)L.hcomp_lanes - 1) / (uint32_t)L.hcomp_lanes;
    r.grid[1] = (uint32_t)L.rows.size() * ng;
    r.grid[2] = (uint32_t)L.light.size() * ng;
    r.grid[3] = (uint32_t)L.icm.size() * ng;
    r.grid[4] = (uint32_t)L.isse.size() * ng;
    r.grid[5] = (uint32_t)L.mix_waves_per_group() * ng;
    const uint32_t nchunks = g.max_len ? (g.max_len + (uint32_t)L.C - 1) / (uint32_t)L.C : 1u;
    r.nsteps = nchunks + (uint32_t)L.coder_level;
    r.group_blocks = (uint32_t)L.G;
    if (L.persist_ok && h_jobs) {
      r.ps_wpg = (uint32_t)L.ps_wpg; r.ps_waves = (uint32_t)L.ps_waves; r.ps_nunit = (uint32_t)L.ps_nunit;
      r.ps_nslot = (uint32_t)L.ps_slots.size();
      r.group_chunks.assign(ng, 1u);
      for (uint32_t b = 0; b < g.count; ++b) {
        const uint32_t len = h_jobs[g.first + b].in_len;
        uint32_t& gc = r.group_chunks[b / (uint32_t)L.G];
        gc = std::max(gc, len ? (len + (uint32_t)L.C - 1) / (uint32_t)L.C : 1u);
      }
    }
    pipe_off += (uint64_t)ng * L.group_bytes;
    runs.push_back(r);
  }
  // packed MIX rows: the tables of the runs that keep them so are rewritten from Predictor::init's pattern (pipe_repack_body)
  auto repack_runs = [&]() {
    for (PipeRun& r : runs) {
      if (!r.k->any_packed) continue;
      PipeArgs a = r.args;
      void* args[1] = {(void*)&a};
      HIP_CHECK(hipModuleLaunchKernel(r.k->repack, r.args.nblocks, 1, 1, 256, 1, 1, 0, st, args, nullptr));
    }
  };
  repack_runs();
  // Groups (one per kernel kind / plan) are independent: fan the single-launch ones out over side streams so a
  // batch mixing several chains does not serialise one launch after the other.
  size_t nsingle = 0;
  for (const LaunchGroup& g : groups) nsingle += g.pick.kind != 4;
  const size_t nside = nsingle > 1 ? std::min<size_t>(nsingle - 1, 7) : 0;
  while (e.side.size() < nside) {
    hipStream_t s2;
    HIP_CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    e.side.push_back(s2);
  }
  Event fork;
  if (nside) HIP_CHECK(hipEventRecord(fork, st));
  size_t gi = 0;
  for (const LaunchGroup& g : groups) {
    if (g.pick.kind == 4) continue;
    hipStream_t gs = (nside && gi % (nside + 1)) ? e.side[gi % (nside + 1) - 1] : st;
    if (gs != st && gi <= nside) HIP_CHECK(hipStreamWaitEvent(gs, fork, 0));
    // every kernel writes res[job.res_slot] relative to the SAME results base
    if (g.pick.kind == 3) HIP_CHECK(launch_spec(g.pick.spec, decode, d_jobs + g.first, d_res, g.count, e.d_tables, gs));
    else if (g.pick.kind == 2) HIP_CHECK(launch_code_wave(decode, d_jobs + g.first, d_res, g.count, e.d_tables, gs));
    else HIP_CHECK(launch_code_serial(decode, d_jobs + g.first, d_res, g.count, e.d_tables, gs));
    ++gi;
  }
  // the persistent launch when every group of the batch is a pipelined one, the call waits for its results anyway, and the
  // groups are resident together (one round, or rounds that are nearly full: a round costs a block's serial time whatever it holds)
  bool coded = false;
  e.last_persist_abort_ms = 0.0;
  if (timed && nsingle == 0 && persist_wanted(runs, true)) {
    bool fits = true;
    if (runs.size() == 1 && persist_capacity(e, runs[0]) >= runs[0].ps_wpg) {
      const uint32_t per_round = std::max<uint32_t>(1u, persist_capacity(e, runs[0]) / runs[0].ps_wpg);
      const uint32_t rounds = (runs[0].ngroups + per_round - 1) / per_round;
      fits = rounds == 1 || (double)runs[0].ngroups / ((double)rounds * per_round) >= 0.85 || getenv("ZPAQ_AMD_PIPE_PERSIST");
    }
    if (fits) {
      if (late && late->from_byte) HIP_CHECK(hipStreamWaitEvent(st, late->arrive(), 0));
      // nothing of this call may hold compute units when the grid arrives (the hashing kernel beside the coder, 46 ms for 1024
      // blocks, kept a few workgroups out for longer than the arrival handshake waits: the launch stepped aside for the
      // library's own kernel -- round 6, call 9)
      if (before_persist) (*before_persist)();
      bool aborted = false, untouched = false;
      std::string what;
      const auto t_launch = std::chrono::steady_clock::now();
      if (launch_pipe_persist(e, runs, st, &aborted, &what, &untouched)) {
        coded = !aborted;
        if (aborted) {
          e.last_persist_abort_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_launch).count();
          fprintf(stderr, "[zpaq_amd] persistent encoder launch gave up after %.1f ms (%s): coding the batch with the step kernels\n",
                  e.last_persist_abort_ms, what.c_str());
          if (!untouched) {
            for (uint32_t b0 = 0; b0 < nb; b0 += 65535u)
              HIP_CHECK(launch_init_arena(d_jobs + b0, std::min(nb - b0, 65535u), e.d_tables, chunks, st));
            repack_runs();
          }
        }
      }
    }
  }
  e.last_persist = coded;
  if (getenv("ZPAQ_AMD_LOG") && !runs.empty()) {
    std::string d;
    for (PipeRun& r : runs)
    {
      uint32_t longest = 0;
      for (uint32_t c : r.group_chunks) longest = std::max(longest, c);
      d += " [" + std::to_string(r.ngroups) + " groups x " + std::to_string(r.ps_wpg) + " workgroups, device holds " +
           std::to_string(r.k->persist ? persist_capacity(e, r) : 0u) + ", longest group " + std::to_string(longest) + " chunks]";
    }
    fprintf(stderr, "[zpaq_amd] pipelined encoder: %zu chain(s)%s -> %s\n", runs.size(), d.c_str(),
            coded ? "one persistent launch each, side by side" : "step kernels");
  }
  if (!coded) launch_pipe(e, runs, st, late);
  for (size_t k = 0; k < nside; ++k) {          // join the side streams back into `st`
    Event done;
    HIP_CHECK(hipEventRecord(done, e.side[k]));
    HIP_CHECK(hipStreamWaitEvent(st, done, 0));
  }
  if (timed) {
    HIP_CHECK(hipEventRecord(ev3, st));
    HIP_CHECK(hipEventSynchronize(ev3));
    float a = 0, b = 0;
    HIP_CHECK(hipEventElapsedTime(&a, ev0, ev1));
    HIP_CHECK(hipEventElapsedTime(&b, ev2, ev3));
    e.last.init_ms = a;
    e.last.code_ms = b;
    e.last.blocks = nb;
  }
}

// The engine's arena / job / stream buffers are shared by all calls: a call that returned with work still in flight
// on the caller's stream (zpq_*_device with timed = 0) must have drained before they are touched again.
static void wait_in_flight(Engine& e) {
  if (e.busy) {
    HIP_CHECK(hipEventSynchronize(e.busy));
  }
}
static void mark_in_flight(Engine& e, hipStream_t st) {
  if (!e.busy) HIP_CHECK(hipEventCreateWithFlags(&e.busy, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(e.busy, st));
}

// kernel kind of sorted job k
static int kind_of_sorted(const std::vector<LaunchGroup>& groups, size_t k) {
  for (const LaunchGroup& g : groups)
    if (k >= g.first && k < (size_t)g.first + g.count) return g.pick.kind;
  return 0;
}

// Sort `order` so that every (kernel, plan) group is contiguous, and cut it into launch groups.
// A batch with several headers nobody has compiled yet (level-5 chains with data-dependent periodic models): their
// kernels are compiled side by side on the host cores before the kernels are picked, so that such a batch pays about one
// compilation time, not one per header.  The code objects land in the cache directory / the loader's in-process store;
// kernel_kind() then finds them there.
// blocks of the batch per plan (and the longest of them) -> the variant of its pipelined encoder
template <class PlanOf, class LenOf>
static std::map<const zpq_plan*, int> pipe_modes(const std::vector<uint32_t>& order, PlanOf plan_of, LenOf len_of, bool persist_expected = true) {
  std::map<const zpq_plan*, std::pair<uint32_t, uint32_t>> cnt;
  for (uint32_t b : order) { auto& c = cnt[plan_of(b)]; ++c.first; c.second = std::max(c.second, (uint32_t)len_of(b)); }
  std::map<const zpq_plan*, int> mode;
  for (auto& kv : cnt) mode[kv.first] = pipe_mode_for(kv.second.first, kv.second.second, pipe_stream_bytes_per_byte(kv.first), kv.first, persist_expected);
  // several chains in one batch share the device's workgroup slots: the persistent launches run side by side only when they
  // are resident TOGETHER, so chains go from the latency shape to the throughput shape (fewer workgroups per group), the one
  // that frees the most first, until the batch fits
  const char* pp = getenv("ZPAQ_AMD_PIPE_PERSIST");
  // (several chains in one batch: variant 3's workgroups are not part of the arithmetic below -- variant 1 there)
  if (cnt.size() > 1) for (auto& kv : mode) if (kv.second == 3) kv.second = 1;
  if (cnt.size() > 1 && !(pp && !strcmp(pp, "0")) && persist_expected && !getenv("ZPAQ_AMD_PIPE_MODE")) {
    struct Need { const zpq_plan* p; uint64_t lat, thr; };      // what an XCD has to hold of the chain in either shape
    std::vector<Need> need;
    bool all = true;
    for (auto& kv : cnt) {
      PipeLayout L0, L1;
      std::string why;
      if (!pipe_layout(*kv.first, pipe_options(0), L0, why) || !L0.persist_ok || !pipe_layout(*kv.first, pipe_options(1), L1, why) || !L1.persist_ok) { all = false; break; }
      const uint64_t groups = (kv.second.first + (uint32_t)L0.G - 1) / (uint32_t)L0.G;
      need.push_back(Need{kv.first, persist_xcd_share(groups, (uint64_t)L1.ps_wpg), persist_xcd_share(groups, (uint64_t)L0.ps_wpg)});
    }
    if (all) {
      for (;;) {
        uint64_t total = 0;
        for (const Need& n : need) total += mode[n.p] == 0 ? n.thr : n.lat;
        if (total <= (uint64_t)g_cus_hint.load() / 8) break;             // (launch_pipe_persist's rule: every XCD's share of every run fits)
        const Need* best = nullptr;
        for (const Need& n : need)
          if (mode[n.p] != 0 && n.lat > n.thr && (!best || n.lat - n.thr > best->lat - best->thr)) best = &n;
        if (!best) break;
        mode[best->p] = 0;
      }
    }
  }
  return mode;
}

template <class PlanOf>
static void precompile_unseen(Engine& e, bool decode, bool dense, const std::vector<uint32_t>& order, PlanOf plan_of,
                              const std::map<const zpq_plan*, int>& mode_of) {
  const int want = e.kernel_choice;
  if (want == 1 || want == 2 || e.jit_left <= 1) return;
  const bool pipe = !decode && (want == 0 || want == 4 || want == 5 || want == 6);
  const int forced = spec_variant_forced();
  const bool team = decode && forced < 0 && (want == 6 || (want == 0 && dense && kAutoTeam));   // as kernel_kind() will choose
  const bool dual = decode && forced < 0 && (want == 5 || (want == 0 && dense));
  const int variant = forced >= 0 ? forced : (team ? 3 : (dual ? 2 : (dense ? 1 : 0)));
  std::vector<const zpq_plan*> unseen;
  std::vector<int> modes;
  const zpq_plan* last = nullptr;
  for (uint32_t b : order) {
    const zpq_plan* p = plan_of(b);
    if (p == last) continue;
    last = p;
    if (!p->hdr().wave_ok) continue;
    const auto& d = p->cur();
    const int mode = mode_of.at(p);
    if (pipe ? d.pipe_state[mode] != 0 : d.spec_state[variant] != 0) continue;        // loaded, or known not to work
    if (std::find(unseen.begin(), unseen.end(), p) == unseen.end()) { unseen.push_back(p); modes.push_back(mode); }
  }
  if (unseen.size() < 2) return;            // a single header is compiled where it is loaded
  e.jit_left -= spec_precompile(unseen, pipe, variant, e.jit_left, jit_threads(), nullptr, &modes);
  if (e.jit_left < 1) e.jit_left = 1;       // (what was compiled is found in the cache; the budget only counts compilations)
}

template <class PlanOf, class LenOf>
static std::vector<LaunchGroup> make_groups(Engine& e, bool decode, std::vector<uint32_t>& order, PlanOf plan_of, LenOf len_of,
                                            const std::set<const zpq_plan*>* multi_segment = nullptr, bool persist_expected = true) {
  const size_t cnt = order.size();
  const bool dense = cnt > (size_t)4 * e.cus;
  const std::map<const zpq_plan*, int> mode_of = pipe_modes(order, plan_of, len_of, persist_expected);
  precompile_unseen(e, decode, dense, order, plan_of, mode_of);
  std::vector<KernelPick> pick(cnt);
  for (size_t k = 0; k < cnt; ++k) {
    const zpq_plan* pl = plan_of(order[k]);
    pick[k] = kernel_kind(e, pl, dense, decode, mode_of.at(pl), multi_segment && multi_segment->count(pl));
  }
  std::vector<uint32_t> idx(cnt);
  for (size_t k = 0; k < cnt; ++k) idx[k] = (uint32_t)k;
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) {
    if (pick[x].kind != pick[y].kind) return pick[x].kind > pick[y].kind;
    if (pick[x].kind >= 3) return plan_of(order[x]) < plan_of(order[y]);
    return false;
  });
  std::vector<uint32_t> sorted(cnt);
  std::vector<LaunchGroup> groups;
  for (size_t k = 0; k < cnt; ++k) {
    const uint32_t b = order[idx[k]];
    sorted[k] = b;
    const KernelPick& pk = pick[idx[k]];
    if (!groups.empty() && same_group(groups.back(), pk, plan_of(b))) {
      ++groups.back().count;
      groups.back().max_len = std::max(groups.back().max_len, len_of(b));
    } else groups.push_back(LaunchGroup{pk, plan_of(b), (uint32_t)k, 1, len_of(b)});
  }
  order.swap(sorted);
  return groups;
}

// ---- submission queue ------------------------------------------------------------------------------------------
// libzpaq callers are threads that each code ONE block (zpaq.cpp:1918-1965 compressThread, 2848-2867
// decompressThread).  One block is one lane of one wavefront per unit: launched alone it leaves the GPU empty, so
// concurrent callers are coalesced: the first one to arrive becomes the leader, lingers while other callers are
// known to be on their way (they announce themselves when they enter the host front half), takes everything that
// queued up, runs it as ONE batch and hands the results back.  A lone caller is never delayed.
namespace {
struct Ticket {
  const std::vector<HostBlock>* blocks;
  std::vector<BlockResult>* results;
  bool done = false;
  std::exception_ptr err;
};
struct Batcher {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Ticket*> queue[2];      // [decode]
  bool leader[2] = {false, false};
  int approaching = 0;                // callers inside the host front half that will submit a block shortly
  std::chrono::steady_clock::time_point last_arrival[2];   // when the youngest ticket of a direction was queued
};
Batcher& batcher() { static Batcher b; return b; }
void engine_code_host_now(bool decode, const std::vector<HostBlock>& blocks, std::vector<BlockResult>& results);
}  // namespace

void engine_caller_enter() { Batcher& b = batcher(); std::lock_guard<std::mutex> g(b.mu); ++b.approaching; }
void engine_caller_leave() {
  Batcher& b = batcher();
  std::lock_guard<std::mutex> g(b.mu);
  if (b.approaching > 0) --b.approaching;
  b.cv.notify_all();
}

// `announced`: the caller told the queue it was coming (engine_caller_enter) and still owns that announcement; it is
// withdrawn here, under the queue's lock, at the moment the blocks are queued -- and stays the caller's to withdraw if
// anything throws before that.
void engine_code_host(bool decode, const std::vector<HostBlock>& blocks, std::vector<BlockResult>& results, bool* announced) {
  Batcher& b = batcher();
  const int d = decode ? 1 : 0;
  Ticket me{&blocks, &results};
  std::unique_lock<std::mutex> lk(b.mu);
  b.queue[d].push_back(&me);
  b.last_arrival[d] = std::chrono::steady_clock::now();
  if (announced && *announced) {
    if (b.approaching > 0) --b.approaching;
    *announced = false;
  }
  b.cv.notify_all();
  for (;;) {
    b.cv.wait(lk, [&] { return me.done || !b.leader[d]; });
    if (me.done) break;
    // become the leader
    b.leader[d] = true;
    // Linger while callers keep coming.  A pool of threads around compressBlock / Decompresser (zpaq.cpp:1918-1965,
    // 2848-2867) is fed by ONE thread that cuts the input into blocks, so its callers arrive milliseconds apart, each
    // after the previous one has already left the host front half: "nobody has announced himself right now" does not mean
    // nobody is coming (measured with the unmodified archiver, profiles/r04: 96 blocks from 16 threads reached the device
    // as 24 batches of 1 - 15, each costing the ~1.7 s of a 1 MiB block's bit chain).  So the leader waits until no
    // caller has arrived for `gap` and none is announced -- `gap` and the cap on the whole wait are a small share of what
    // the queued blocks will cost on the device anyway (1.5 % and 15 %; a block's serial chain takes ~1.7 us per input
    // byte to code and ~15 us per byte to decode), at most 25 ms and 250 ms.
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      uint64_t longest = 0;
      for (const Ticket* t : b.queue[d])
        for (const HostBlock& hb : *t->blocks) longest = std::max<uint64_t>(longest, decode ? hb.out_cap : hb.in_len);
      const double est_ms = (double)longest * (decode ? 15e-3 : 1.7e-3);
      const auto gap = std::chrono::microseconds((long)(1000.0 * std::min(25.0, std::max(0.2, est_ms * 0.015))));
      const auto cap = std::chrono::microseconds((long)(1000.0 * std::min(250.0, std::max(1.0, est_ms * 0.15))));
      const auto now = std::chrono::steady_clock::now();
      if (now - t0 >= cap) break;
      if (b.approaching == 0 && now - b.last_arrival[d] >= gap) break;
      // a batch that fills the machine already (a bulk zpq_compress_blocks / decode_archive call, or a pool that has queued
      // that much) gains nothing from company: go at once unless somebody is announced
      {
        uint64_t queued = 0;
        for (const Ticket* t : b.queue[d]) queued += t->blocks->size();
        if (b.approaching == 0 && queued >= (uint64_t)4 * (uint64_t)g_cus_hint.load()) break;
      }
      b.cv.wait_for(lk, std::chrono::microseconds(500));
    }
    std::vector<Ticket*> batch;
    batch.swap(b.queue[d]);
    lk.unlock();
    // Everything the leader does for the others sits inside one try block: whatever throws (the merge's allocations
    // included), every queued ticket is marked done with the error and the leader flag is cleared below.
    std::exception_ptr err;
    try {
      if (batch.size() == 1) {
        engine_code_host_now(decode, *batch[0]->blocks, *batch[0]->results);
      } else {
        std::vector<HostBlock> all;
        for (Ticket* t : batch) all.insert(all.end(), t->blocks->begin(), t->blocks->end());
        std::vector<BlockResult> res;
        engine_code_host_now(decode, all, res);
        size_t pos = 0;
        for (Ticket* t : batch) {
          t->results->assign(res.begin() + (long)pos, res.begin() + (long)(pos + t->blocks->size()));
          pos += t->blocks->size();
        }
      }
    } catch (...) { err = std::current_exception(); }
    lk.lock();
    for (Ticket* t : batch) { t->err = err; t->done = true; }
    b.leader[d] = false;
    b.cv.notify_all();
    if (me.done) break;       // (always: the leader's own ticket was in the batch)
  }
  lk.unlock();
  if (me.err) std::rethrow_exception(me.err);
}

namespace {
void engine_code_host_on(int dev, bool decode, const std::vector<HostBlock>& blocks, std::vector<BlockResult>& results);

// One batch of host blocks: on the one configured device, or sharded over all of them (block b of B goes to shard
// b * G / B -- contiguous ranges, output order preserved), one thread per device.
void engine_code_host_now(bool decode, const std::vector<HostBlock>& blocks, std::vector<BlockResult>& results) {
  std::vector<int> ids;
  { DeviceSet& d = devset(); std::lock_guard<std::mutex> g(d.mu); ids = d.ids; }
  if (ids.empty()) ids.push_back(primary_device());
  const size_t parts = std::min(ids.size(), blocks.size());
  if (parts <= 1) { engine_code_host_on(ids[0], decode, blocks, results); return; }
  results.assign(blocks.size(), BlockResult{0, 0, 0, 0});
  std::vector<std::exception_ptr> errs(parts);
  std::vector<std::thread> pool;
  for (size_t k = 0; k < parts; ++k)
    pool.emplace_back([&, k] {
      try {
        uint64_t lo, hi;
        engine_shard_range(blocks.size(), (uint32_t)parts, (uint32_t)k, &lo, &hi);
        std::vector<HostBlock> mine(blocks.begin() + (long)lo, blocks.begin() + (long)hi);
        std::vector<BlockResult> res;
        engine_code_host_on(ids[k], decode, mine, res);
        std::copy(res.begin(), res.end(), results.begin() + (long)lo);
      } catch (...) { errs[k] = std::current_exception(); }
    });
  for (auto& t : pool) t.join();
  for (auto& e : errs) if (e) std::rethrow_exception(e);
}

void engine_code_host_on(int dev, bool decode, const std::vector<HostBlock>& blocks, std::vector<BlockResult>& results) {
  Engine& e = eng(dev);
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e, dev);
  bind_device(e);
  wait_in_flight(e);
  const size_t nb = blocks.size();
  e.jit_left = jit_budget();
  results.assign(nb, BlockResult{0, 0, 0, 0});
  size_t pos = 0;
  e.last = Timing{};
  while (pos < nb) {
    // one residency wave: as many blocks as fit the state budget (model state + the encoder's stream buffers)
    uint64_t need = 0, in_bytes = 0, out_bytes = 0, max_arena = 0;
    size_t end = pos;
    while (end < nb) {
      const HostBlock& hb = blocks[end];
      uint64_t a = hb.plan->hdr().arena_bytes;
      // share of a full group's stream buffer, in the largest of the encoder's shapes (which one the wave gets is decided
      // from its block count further down: the 2048-byte steps of the latency shape need about four times the throughput shape's)
      if (!decode && e.kernel_choice != 1 && e.kernel_choice != 2 && e.kernel_choice != 3) {
        // (a wave that outgrows the latency shapes' block count runs the throughput shape: from there on its share is charged)
        uint64_t share = pipe_bytes(hb.plan, 64, 0) / 64;
        if (end - pos < kLatencyModeBlocks)
          for (int m = 1; m < kPipeVariants; ++m) share = std::max(share, pipe_bytes(hb.plan, 64, m) / 64);
        a += share;
      }
      if (end > pos && need + a > e.budget) break;
      need += a;
      max_arena = std::max(max_arena, hb.plan->hdr().arena_bytes);
      in_bytes += ((uint64_t)hb.in_len + hb.prefix_len + 63) & ~63ull;
      out_bytes += ((uint64_t)hb.out_cap + 63) & ~63ull;
      ++end;
    }
    if (need > e.budget) fail(ZPQ_E_NOMEM, "Out of memory: one block's model state exceeds the device budget");
    const size_t cnt = end - pos;
    const auto wave_begin = std::chrono::steady_clock::now();
    uint64_t arena_need = 0;
    for (size_t i = pos; i < end; ++i) arena_need += blocks[i].plan->hdr().arena_bytes;
    e.arena.ensure(arena_need);
    e.io_in.ensure(in_bytes + 64);
    e.io_out.ensure(out_bytes + 64);
    e.jobs.ensure(cnt * sizeof(BlockJob));
    e.results.ensure(cnt * sizeof(BlockResult));
    std::vector<uint32_t> order(cnt);
    for (size_t i = 0; i < cnt; ++i) order[i] = (uint32_t)(pos + i);
    std::set<const zpq_plan*> multi_segment;
    for (size_t i = pos; i < end; ++i) if (blocks[i].nseg > 1) multi_segment.insert(blocks[i].plan);
    std::vector<LaunchGroup> groups = make_groups(
        e, decode, order, [&](uint32_t b) { return blocks[b].plan; },
        [&](uint32_t b) { return blocks[b].in_len + blocks[b].prefix_len; }, &multi_segment);
    std::vector<BlockJob> jobs(cnt);
    // inputs are gathered into one buffer and sent with one copy: a page-locked buffer kept by the engine, filled by
    // several threads (+3 % on the headline's API figure in round 3); ZPAQ_AMD_PINNED_STAGE=0 = pageable memory
    std::unique_ptr<uint8_t[]> stage_buf;
    uint8_t* stage = nullptr;
    const char* pin_env = getenv("ZPAQ_AMD_PINNED_STAGE");          // (A/B of round 3: "0" = pageable staging)
    const bool pinned = !(pin_env && pin_env[0] == '0') && in_bytes >= (1u << 20) && e.pin_in.ensure(in_bytes + 64);
    if (pinned) stage = (uint8_t*)e.pin_in.p;
    else {
      // (not a std::vector: value-initialising a gigabyte costs a quarter of a second; the padding between blocks is
      // never read -- every kernel goes by in_len)
      stage_buf.reset(new uint8_t[in_bytes + 64]);
      stage = stage_buf.get();
    }
    std::vector<uint64_t> in_off_of(cnt);
    std::vector<uint64_t> out_off(cnt);
    // segment tables of the blocks that have several segments
    std::vector<SegRange> segtab;
    std::vector<size_t> seg_first(cnt, 0);
    for (size_t k = 0; k < cnt; ++k) {
      const HostBlock& hb = blocks[order[k]];
      if (hb.nseg <= 1) continue;
      const int kd = kind_of_sorted(groups, k);
      if (kd != (decode ? 3 : 4))
        fail(ZPQ_E_UNSUPPORTED, "blocks of several segments need the pipelined encoder / the per-header wavefront decoder");
      seg_first[k] = segtab.size();
      U32 at = 0;
      for (U32 sgi = 0; sgi < hb.nseg; ++sgi) {
        const U32 l = hb.seg_len[sgi] + (sgi == 0 ? hb.prefix_len : 0);
        segtab.push_back(SegRange{at, at + l, 0, 0});
        at += l;
      }
      if (at != hb.in_len + hb.prefix_len) fail(ZPQ_E_ARG, "segment lengths do not add up to the block");
    }
    if (!segtab.empty()) {
      e.segs.ensure(segtab.size() * sizeof(SegRange));
      HIP_CHECK(hipMemcpyAsync(e.segs.p, segtab.data(), segtab.size() * sizeof(SegRange), hipMemcpyHostToDevice, e.stream));
    }
    uint64_t a_off = 0, i_off = 0, o_off = 0;
    for (size_t k = 0; k < cnt; ++k) {
      const HostBlock& hb = blocks[order[k]];
      BlockJob& j = jobs[k];
      memset(&j, 0, sizeof(j));
      if (hb.nseg > 1) { j.nseg = hb.nseg; j.segs = (SegRange*)e.segs.p + seg_first[k]; }
      j.plan = plan_on_device(e, hb.plan);
      j.arena = (uint8_t*)e.arena.p + a_off;
      j.in = (const uint8_t*)e.io_in.p + i_off;
      j.out = (uint8_t*)e.io_out.p + o_off;
      j.in_len = hb.in_len + hb.prefix_len;
      j.out_cap = hb.out_cap;
      j.res_slot = (uint32_t)k;
      in_off_of[k] = i_off;
      out_off[k] = o_off;
      a_off += hb.plan->hdr().arena_bytes;
      i_off += ((uint64_t)hb.in_len + hb.prefix_len + 63) & ~63ull;
      o_off += ((uint64_t)hb.out_cap + 63) & ~63ull;
    }
    // Gather + copy.  The pipelined encoder reads input byte k no earlier than step k / chunk, so for a batch of equally long
    // blocks (the bulk case: compress() / compressBlocks over a cut-up stream) only the first kHeadBytes of every block are
    // gathered and copied before the kernels are launched; the rest is gathered by helper threads while the first steps run
    // and copied -- one strided copy -- when the launch loop reaches the first step that may read it (LateInput).
    static const uint32_t kHeadBytes = 64u << 10;
    const uint32_t len0 = cnt ? blocks[order[0]].in_len + blocks[order[0]].prefix_len : 0;
    bool split = pinned && !decode && cnt >= 64 && len0 >= 8 * kHeadBytes && multi_segment.empty() && !getenv("ZPAQ_AMD_PIPE_PROFILE");
    if (const char* sc = getenv("ZPAQ_AMD_SPLIT_COPY")) split = split && sc[0] != '0';      // (A/B aid: "0" = one copy up front)
    for (size_t k = 0; split && k < cnt; ++k)
      split = blocks[order[k]].in_len + blocks[order[k]].prefix_len == len0 && kind_of_sorted(groups, k) == 4;
    const uint64_t pitch = ((uint64_t)len0 + 63) & ~63ull;      // (= the distance between two blocks' inputs when all are len0 long)
    // bytes [from, to) of every block's input (prefix first, then the data) -> staging
    auto gather_range = [&](size_t t, size_t nt, uint32_t from, uint32_t to) {
      for (size_t k = t; k < cnt; k += nt) {
        const HostBlock& hb = blocks[order[k]];
        const uint32_t total = hb.prefix_len + hb.in_len, hi = std::min(to, total);
        uint8_t* dst = stage + in_off_of[k];
        if (from < hb.prefix_len) memcpy(dst + from, hb.prefix + from, std::min(hi, hb.prefix_len) - from);
        const uint32_t lo = std::max(from, hb.prefix_len);
        if (hi > lo) memcpy(dst + lo, hb.in + (lo - hb.prefix_len), hi - lo);
      }
    };
    auto gather_all = [&](uint32_t from, uint32_t to, std::vector<std::thread>& pool, bool wait) {
      // a gigabyte of memcpy for a full batch: one thread per 16 MiB, at most 8 (small batches: this thread alone)
      const size_t nt = std::max<size_t>(1, std::min<size_t>({(size_t)8, cnt, (size_t)(in_bytes >> 24) + 1}));
      size_t started = wait ? 1 : 0;
      try {
        for (; started < nt; ++started) pool.emplace_back(gather_range, started, nt, from, to);
      } catch (...) {}                                  // no more threads to be had: this one does the rest
      if (wait) {
        gather_range(0, nt, from, to);
        for (size_t t = started; t < nt; ++t) gather_range(t, nt, from, to);
        for (auto& th : pool) th.join();
        pool.clear();
      } else {
        for (size_t t = started; t < nt; ++t) gather_range(t, nt, from, to);     // (what no thread could be started for)
      }
    };
    std::vector<std::thread> tail_pool;
    struct JoinAll { std::vector<std::thread>& p; ~JoinAll() { for (auto& t : p) if (t.joinable()) t.join(); } } join_tail{tail_pool};
    if (split) {
      std::vector<std::thread> pool;
      gather_all(0, kHeadBytes, pool, true);
      HIP_CHECK(hipMemcpy2DAsync(e.io_in.p, pitch, stage, pitch, kHeadBytes, cnt, hipMemcpyHostToDevice, e.stream));
      gather_all(kHeadBytes, 0xFFFFFFFFu, tail_pool, false);        // in the background, joined by LateInput::arrive
    } else {
      std::vector<std::thread> pool;
      gather_all(0, 0xFFFFFFFFu, pool, true);
      HIP_CHECK(hipMemcpyAsync(e.io_in.p, stage, in_bytes, hipMemcpyHostToDevice, e.stream));
    }
    HIP_CHECK(hipMemcpyAsync(e.jobs.p, jobs.data(), cnt * sizeof(BlockJob), hipMemcpyHostToDevice, e.stream));
    // SHA-1 of the blocks whose caller asked for it: one lane per block, on a side stream beside the coder
    std::vector<Sha1Job> shj;
    std::vector<size_t> sh_of;
    for (size_t k = 0; k < cnt; ++k) {
      const HostBlock& hb = blocks[order[k]];
      if (!hb.sha1_out || decode) continue;
      shj.push_back(Sha1Job{jobs[k].in + hb.prefix_len, hb.in_len, (uint32_t)shj.size()});
      sh_of.push_back(order[k]);
    }
    Event sha_done, staged, tail_arrived;
    if (!shj.empty() || split) {
      if (e.side.empty()) { hipStream_t s2; HIP_CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); e.side.push_back(s2); }
    }
    if (!shj.empty()) {
      e.sha_jobs.ensure(shj.size() * sizeof(Sha1Job));
      e.sha_out.ensure(shj.size() * 20);
      HIP_CHECK(hipMemcpyAsync(e.sha_jobs.p, shj.data(), shj.size() * sizeof(Sha1Job), hipMemcpyHostToDevice, e.stream));
    }
    HIP_CHECK(hipEventRecord(staged, e.stream));
    // hashing needs whole blocks: behind the input's tail when that comes later (same side stream, in order)
    auto hash_blocks = [&] {
      if (shj.empty()) return;
      HIP_CHECK(hipStreamWaitEvent(e.side[0], staged, 0));
      HIP_CHECK(launch_sha1((const Sha1Job*)e.sha_jobs.p, (uint32_t)shj.size(), (uint8_t*)e.sha_out.p, e.side[0]));
      HIP_CHECK(hipEventRecord(sha_done, e.side[0]));
    };
    LateInput late;
    bool tail_sent = false;
    if (split) {
      late.from_byte = kHeadBytes;
      late.arrive = [&]() -> hipEvent_t {
        if (!tail_sent) {
          for (auto& t : tail_pool) t.join();
          tail_pool.clear();
          HIP_CHECK(hipStreamWaitEvent(e.side[0], staged, 0));          // (the buffers are this call's from here on)
          HIP_CHECK(hipMemcpy2DAsync((uint8_t*)e.io_in.p + kHeadBytes, pitch, stage + kHeadBytes, pitch, pitch - kHeadBytes, cnt,
                                     hipMemcpyHostToDevice, e.side[0]));
          HIP_CHECK(hipEventRecord(tail_arrived, e.side[0]));
          hash_blocks();
          tail_sent = true;
        }
        return tail_arrived;
      };
    } else {
      hash_blocks();
    }
    Timing before = e.last;
    const auto wave_t0 = std::chrono::steady_clock::now();
    e.last_kind = groups.empty() ? 0 : groups[0].pick.kind;
    const std::function<void()> hashing_done = [&]() {
      if (!shj.empty()) HIP_CHECK(hipStreamWaitEvent(e.stream, sha_done, 0));        // (enqueued above, or by late.arrive() just now)
    };
    launch_all(e, decode, (const BlockJob*)e.jobs.p, (BlockResult*)e.results.p, groups, (uint32_t)cnt, max_arena,
               e.stream, true, split ? &late : nullptr, jobs.data(), &hashing_done);
    if (split && !tail_sent) HIP_CHECK(hipStreamWaitEvent(e.stream, late.arrive(), 0));   // (no pipelined group after all)
    e.last.init_ms += before.init_ms;
    e.last.code_ms += before.code_ms;
    e.last.blocks += before.blocks;
    std::vector<BlockResult> res(cnt);
    std::vector<uint8_t> digests(shj.size() * 20);
    if (!shj.empty()) {
      HIP_CHECK(hipStreamWaitEvent(e.stream, sha_done, 0));
      HIP_CHECK(hipMemcpyAsync(digests.data(), e.sha_out.p, digests.size(), hipMemcpyDeviceToHost, e.stream));
    }
    HIP_CHECK(hipMemcpyAsync(res.data(), e.results.p, cnt * sizeof(BlockResult), hipMemcpyDeviceToHost, e.stream));
    HIP_CHECK(hipStreamSynchronize(e.stream));
    const auto wave_results = std::chrono::steady_clock::now();
    for (size_t i = 0; i < sh_of.size(); ++i) memcpy(blocks[sh_of[i]].sha1_out, digests.data() + 20 * i, 20);
    if (!segtab.empty()) {
      HIP_CHECK(hipMemcpyAsync(segtab.data(), e.segs.p, segtab.size() * sizeof(SegRange), hipMemcpyDeviceToHost, e.stream));
      HIP_CHECK(hipStreamSynchronize(e.stream));
      for (size_t k = 0; k < cnt; ++k) {
        const HostBlock& hb = blocks[order[k]];
        if (hb.nseg <= 1 || !hb.seg_out_end) continue;
        for (U32 sgi = 0; sgi < hb.nseg; ++sgi) hb.seg_out_end[sgi] = segtab[seg_first[k] + sgi].out_end;
        if (decode && !res[k].status)
          for (U32 sgi = 0; sgi < hb.nseg; ++sgi)
            if (segtab[seg_first[k] + sgi].status) res[k].consumed = 0;      // capacity reached before the last end-of-stream
      }
    }
    // Outputs.  A copy into the caller's pageable buffer is staged by the runtime and returns when it is done: a thousand of
    // them cost a few hundred ms however small they are.  A large batch's outputs therefore go -- asynchronously, back to
    // back -- into a page-locked buffer kept by the engine and are handed out from there by a few threads.
    uint64_t out_total = 0;
    std::vector<uint64_t> pin_off(cnt, 0);
    for (size_t k = 0; k < cnt; ++k) {
      const HostBlock& hb = blocks[order[k]];
      results[order[k]] = res[k];
      pin_off[k] = out_total;
      if (hb.out) out_total += ((uint64_t)std::min(res[k].out_len, hb.out_cap) + 15) & ~15ull;
    }
    const bool pin_outputs = pinned && cnt >= 64 && out_total >= (1u << 20) && e.pin_out.ensure(out_total + 64);
    for (size_t k = 0; k < cnt; ++k) {
      const HostBlock& hb = blocks[order[k]];
      const uint32_t got = std::min(res[k].out_len, hb.out_cap);
      if (got && hb.out)
        HIP_CHECK(hipMemcpyAsync(pin_outputs ? (uint8_t*)e.pin_out.p + pin_off[k] : hb.out, (const uint8_t*)e.io_out.p + out_off[k], got,
                                 hipMemcpyDeviceToHost, e.stream));
    }
    HIP_CHECK(hipStreamSynchronize(e.stream));
    if (pin_outputs) {
      const size_t nt = std::max<size_t>(1, std::min<size_t>({(size_t)8, cnt, (size_t)(out_total >> 24) + 1}));
      auto hand_out = [&](size_t t) {
        for (size_t k = t; k < cnt; k += nt) {
          const HostBlock& hb = blocks[order[k]];
          const uint32_t got = std::min(res[k].out_len, hb.out_cap);
          if (got && hb.out) memcpy(hb.out, (const uint8_t*)e.pin_out.p + pin_off[k], got);
        }
      };
      std::vector<std::thread> pool;
      size_t started = 1;
      try {
        for (; started < nt; ++started) pool.emplace_back(hand_out, started);
      } catch (...) {}
      hand_out(0);
      for (size_t t = started; t < nt; ++t) hand_out(t);
      for (auto& th : pool) th.join();
    }
    if (getenv("ZPAQ_AMD_LOG")) {     // one line per device batch: what a caller-side pool of threads really hands over, and where its time went
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
      };
      fprintf(stderr, "[zpaq_amd] %s batch of %zu blocks (%llu bytes in): %zu launch group(s), kernel kind %d mode %d; buffers + gather + "
              "copy enqueued %.1f ms, launches to results %.1f ms (init %.1f, coding %.1f on the device), outputs %.1f ms%s\n",
              decode ? "decode" : "encode", cnt, (unsigned long long)in_bytes, groups.size(), e.last_kind,
              groups.empty() ? 0 : groups[0].pick.mode, ms(wave_begin, wave_t0), ms(wave_t0, wave_results), e.last.init_ms, e.last.code_ms,
              ms(wave_results, std::chrono::steady_clock::now()), split ? " (input tail copied behind the first steps)" : "");
    }
    pos = end;
  }
}
}  // namespace

void engine_code_device(bool decode, const zpq_plan* const* plans, bool one_plan, const void* d_in,
                        const uint64_t* in_off, const uint32_t* in_len, uint32_t nblocks, void* d_out,
                        const uint64_t* out_off, const uint32_t* out_cap, BlockResult* d_res, void* stream, bool timed) {
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e);
  bind_device(e);
  wait_in_flight(e);
  hipStream_t st = stream ? (hipStream_t)stream : e.stream;
  e.jit_left = jit_budget();
  auto plan_of = [&](uint32_t b) { return one_plan ? plans[0] : plans[b]; };
  uint64_t need = 0, max_arena = 0;
  for (uint32_t b = 0; b < nblocks; ++b) {
    const uint64_t a = plan_of(b)->hdr().arena_bytes;
    need += a;
    max_arena = std::max(max_arena, a);
  }
  // group blocks by (kernel, plan); results keep the caller's block order through res_slot
  std::vector<uint32_t> order(nblocks);
  for (uint32_t b = 0; b < nblocks; ++b) order[b] = b;
  std::vector<LaunchGroup> groups = make_groups(e, decode, order, plan_of, [&](uint32_t b) { return in_len[b]; }, nullptr, timed);
  uint64_t pipe_need = 0;
  for (const LaunchGroup& gr : groups)
    if (gr.pick.kind == 4) pipe_need += pipe_bytes(gr.plan, gr.count, gr.pick.mode);
  if (need + pipe_need > e.budget) fail(ZPQ_E_NOMEM, "Out of memory: batch state exceeds the device budget (split the batch)");
  e.arena.ensure(need);
  e.jobs.ensure((size_t)nblocks * sizeof(BlockJob));
  std::vector<BlockJob> jobs(nblocks);
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
  }
  HIP_CHECK(hipMemcpyAsync(e.jobs.p, jobs.data(), (size_t)nblocks * sizeof(BlockJob), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipStreamSynchronize(st));   // jobs vector goes out of scope below
  e.last = Timing{};
  e.last_kind = groups.empty() ? 0 : groups[0].pick.kind;
  launch_all(e, decode, (const BlockJob*)e.jobs.p, d_res, groups, nblocks, max_arena, st, timed, nullptr, jobs.data());
  if (!timed) mark_in_flight(e, st);
}

bool engine_pcomp(const U8* code, size_t codelen, int ph, int pm, std::vector<PcompSeg>& segs, std::string& note) {
  if (segs.empty()) return true;
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e);
  bind_device(e);
  wait_in_flight(e);
  PcompKernel* k = pcomp_kernel_for(code, codelen, ph, pm, note);
  if (!k) return false;
  const size_t n = segs.size();
  const uint64_t mbytes = ((1ull << pm) + 255) & ~255ull, hbytes = ((4ull << ph) + 255) & ~255ull;
  std::vector<uint64_t> cap(n);
  for (size_t i = 0; i < n; ++i) cap[i] = (segs[i].hint ? segs[i].hint : 8ull * segs[i].in_len) + 65536;
  for (int attempt = 0; attempt < 2; ++attempt) {
    uint64_t in_bytes = 0, out_bytes = 0;
    for (size_t i = 0; i < n; ++i) {
      if (cap[i] > 0xFFFFFFF0ull) { note = "segment output beyond the device kernel's 32-bit range"; return false; }   // the host runs it (the size hint is a comment the reference ignores)
      in_bytes += ((uint64_t)segs[i].in_len + 63) & ~63ull;
      out_bytes += (cap[i] + 63) & ~63ull;
    }
    const uint64_t work = (uint64_t)n * (mbytes + hbytes + 1024);
    if (in_bytes + out_bytes + work > e.budget) { note = "post-processor state exceeds the device budget"; return false; }
    e.io_in.ensure(in_bytes + 64);
    e.io_out.ensure(out_bytes + 64);
    e.arena.ensure(work);
    e.jobs.ensure(n * sizeof(PcompJob));
    e.results.ensure(n * 8);
    std::vector<uint8_t> stage(in_bytes + 64);
    std::vector<PcompJob> jobs(n);
    std::vector<uint64_t> ooff(n);
    uint64_t io = 0, oo = 0;
    for (size_t i = 0; i < n; ++i) {
      if (segs[i].in_len) memcpy(stage.data() + io, segs[i].in, segs[i].in_len);
      PcompJob& j = jobs[i];
      j.in = (const uint8_t*)e.io_in.p + io;
      j.out = (uint8_t*)e.io_out.p + oo;
      uint8_t* w = (uint8_t*)e.arena.p + (uint64_t)i * (mbytes + hbytes + 1024);
      j.M = w;
      j.H = (uint32_t*)(w + mbytes);
      j.R = (uint32_t*)(w + mbytes + hbytes);
      j.in_len = segs[i].in_len;
      j.out_cap = (uint32_t)cap[i];
      j.result = (uint32_t*)e.results.p + 2 * i;
      ooff[i] = oo;
      io += ((uint64_t)segs[i].in_len + 63) & ~63ull;
      oo += (cap[i] + 63) & ~63ull;
    }
    HIP_CHECK(hipMemsetAsync(e.arena.p, 0, work, e.stream));
    HIP_CHECK(hipMemcpyAsync(e.io_in.p, stage.data(), in_bytes, hipMemcpyHostToDevice, e.stream));
    HIP_CHECK(hipMemcpyAsync(e.jobs.p, jobs.data(), n * sizeof(PcompJob), hipMemcpyHostToDevice, e.stream));
    const PcompJob* d_jobs = (const PcompJob*)e.jobs.p;
    unsigned nn = (unsigned)n;
    void* args[2] = {(void*)&d_jobs, (void*)&nn};
    HIP_CHECK(hipModuleLaunchKernel(k->fn, (unsigned)((n + 63) / 64), 1, 1, 64, 1, 1, 0, e.stream, args, nullptr));
    std::vector<uint32_t> res(2 * n);
    HIP_CHECK(hipMemcpyAsync(res.data(), e.results.p, n * 8, hipMemcpyDeviceToHost, e.stream));
    HIP_CHECK(hipStreamSynchronize(e.stream));
    bool again = false;
    for (size_t i = 0; i < n; ++i) {
      // a device status is not a verdict: the translated program has a fixed budget of backward jumps, so the host
      // post-processor (which owns the ZPAQL-error decision, like the reference's) runs these segments again
      if (res[2 * i + 1]) { note = "device post-processor stopped (status " + std::to_string(res[2 * i + 1]) + "): host fallback"; return false; }
      if (res[2 * i] > cap[i]) { cap[i] = res[2 * i]; again = true; }
    }
    if (again && attempt == 0) continue;
    for (size_t i = 0; i < n; ++i) {
      segs[i].out->resize(res[2 * i]);
      if (res[2 * i])
        HIP_CHECK(hipMemcpyAsync(segs[i].out->data(), (const uint8_t*)e.io_out.p + ooff[i], res[2 * i], hipMemcpyDeviceToHost, e.stream));
    }
    HIP_CHECK(hipStreamSynchronize(e.stream));
    return true;
  }
  return false;
}

// SHA-1 of n host buffers on the device (one lane per buffer); 20 bytes each into out.
void engine_sha1_host(const uint8_t* const* in, const uint32_t* len, uint32_t n, uint8_t* out) {
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e);
  bind_device(e);
  wait_in_flight(e);
  uint64_t bytes = 0;
  for (uint32_t i = 0; i < n; ++i) bytes += ((uint64_t)len[i] + 63) & ~63ull;
  e.io_in.ensure(bytes + 64);
  e.sha_jobs.ensure((size_t)n * sizeof(Sha1Job));
  e.sha_out.ensure((size_t)n * 20);
  std::vector<uint8_t> stage(bytes + 64);
  std::vector<Sha1Job> jobs(n);
  uint64_t off = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (len[i]) memcpy(stage.data() + off, in[i], len[i]);
    jobs[i] = Sha1Job{(const uint8_t*)e.io_in.p + off, len[i], i};
    off += ((uint64_t)len[i] + 63) & ~63ull;
  }
  HIP_CHECK(hipMemcpyAsync(e.io_in.p, stage.data(), bytes, hipMemcpyHostToDevice, e.stream));
  HIP_CHECK(hipMemcpyAsync(e.sha_jobs.p, jobs.data(), (size_t)n * sizeof(Sha1Job), hipMemcpyHostToDevice, e.stream));
  HIP_CHECK(launch_sha1((const Sha1Job*)e.sha_jobs.p, n, (uint8_t*)e.sha_out.p, e.stream));
  HIP_CHECK(hipMemcpyAsync(out, e.sha_out.p, (size_t)n * 20, hipMemcpyDeviceToHost, e.stream));
  HIP_CHECK(hipStreamSynchronize(e.stream));
}

bool engine_suffix_arrays(const std::vector<std::pair<const U8*, U32>>& blocks, std::vector<std::vector<U32>>& sa, std::string& note) {
  const size_t n = blocks.size();
  sa.assign(n, std::vector<U32>());
  uint64_t total = 0;
  uint32_t max_len = 0;
  for (auto& b : blocks) { total += b.second; max_len = std::max(max_len, b.second); }
  if (!total) return true;
  if (n > 65535 || max_len >= (1u << 24) || total >= (1ull << 31)) { note = "batch outside the device sorter's range"; return false; }
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e);
  bind_device(e);
  wait_in_flight(e);
  const size_t ws = sa_workspace_bytes(total, (uint32_t)n);
  const uint64_t in_bytes = (total + 255) & ~255ull;
  if (ws + in_bytes + 4 * total + (1u << 20) > e.budget) { note = "suffix sort workspace exceeds the device budget"; return false; }
  // inputs back to back in io_in, the arrays in io_out, the sorter's workspace in the arena buffer (idle between batches)
  e.io_in.ensure(in_bytes + 64);
  e.io_out.ensure(4 * total + 64);
  e.arena.ensure(ws);
  e.jobs.ensure((n + 2) * 16 + 64);
  std::vector<const uint8_t*> ptrs(n);
  std::vector<uint64_t> off(n + 1, 0);
  const bool pinned = in_bytes >= (1u << 20) && e.pin_in.ensure(in_bytes + 64);
  std::unique_ptr<uint8_t[]> pageable;
  uint8_t* stage = pinned ? (uint8_t*)e.pin_in.p : (pageable.reset(new uint8_t[in_bytes + 64]), pageable.get());
  for (size_t i = 0; i < n; ++i) {
    ptrs[i] = (const uint8_t*)e.io_in.p + off[i];
    if (blocks[i].second) memcpy(stage + off[i], blocks[i].first, blocks[i].second);
    off[i + 1] = off[i] + blocks[i].second;
  }
  uint8_t* meta = (uint8_t*)e.jobs.p;
  HIP_CHECK(hipMemcpyAsync(e.io_in.p, stage, total, hipMemcpyHostToDevice, e.stream));
  HIP_CHECK(hipMemcpyAsync(meta, ptrs.data(), n * 8, hipMemcpyHostToDevice, e.stream));
  HIP_CHECK(hipMemcpyAsync(meta + ((n * 8 + 15) & ~15ull), off.data(), (n + 1) * 8, hipMemcpyHostToDevice, e.stream));
  uint32_t rounds = 0;
  const hipError_t rc = build_suffix_arrays((const uint8_t* const*)meta, (const uint64_t*)(meta + ((n * 8 + 15) & ~15ull)), (uint32_t)n, total, max_len,
                                            (uint32_t*)e.io_out.p, e.arena.p, e.arena.cap, e.stream, &rounds);
  if (rc != hipSuccess) { (void)hipGetLastError(); note = std::string("device suffix sort failed: ") + hipGetErrorString(rc); return false; }
  for (size_t i = 0; i < n; ++i) {
    sa[i].resize(blocks[i].second);
    if (blocks[i].second)
      HIP_CHECK(hipMemcpyAsync(sa[i].data(), (const uint32_t*)e.io_out.p + off[i], 4ull * blocks[i].second, hipMemcpyDeviceToHost, e.stream));
  }
  HIP_CHECK(hipStreamSynchronize(e.stream));
  note = "device, " + std::to_string(rounds) + " doubling rounds";
  return true;
}

bool engine_sort_preprocess(const std::vector<SortJob>& jobs, std::vector<SortOut>& out, std::string& note) {
  const size_t n = jobs.size();
  out.assign(n, SortOut());
  uint64_t total = 0, ntok = 0, bwt_bytes = 0;
  uint32_t max_len = 0;
  bool any_lz = false, any_bwt = false;
  std::vector<LzBlock> blk(n);
  for (size_t i = 0; i < n; ++i) {
    const SortJob& j = jobs[i];
    LzBlock& B = blk[i];
    memset(&B, 0, sizeof(B));
    B.off = total;
    B.n = j.n;
    B.kind = j.n ? j.kind : 0u;
    B.min_match = j.min_match; B.lookahead = j.lookahead; B.bucket = j.bucket; B.checkbits = j.checkbits;
    if (B.kind == 1 || B.kind == 2) {
      if (j.min_match < 1 || j.lookahead > 255 || j.checkbits < 1 || j.checkbits > 31) { note = "LZ77 parameters outside the device parser's range"; return false; }
      B.tok_off = ntok;
      B.tok_cap = j.n / j.min_match + 2;
      ntok += B.tok_cap;
      any_lz = true;
    } else if (B.kind == 3) {
      any_bwt = true;
    } else if (B.kind != 0) { note = "unknown pre-processor kind"; return false; }
    total += j.n;
    max_len = std::max(max_len, j.n);
  }
  if (!total) return true;
  bwt_bytes = any_bwt ? total + n : 0;
  if (n > 65535 || max_len >= (1u << 24) || total >= (1ull << 31)) { note = "batch outside the device sorter's range"; return false; }
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e);
  bind_device(e);
  wait_in_flight(e);
  const size_t ws = sa_workspace_bytes(total, (uint32_t)n);
  const uint64_t in_bytes = (total + 255) & ~255ull;
  // behind the arrays in io_out: decisions (16 B per element), tokens, BWT bytes, counts and indices, the block table
  const uint64_t o_res = (4 * total + 255) & ~255ull;
  const uint64_t o_tok = o_res + (any_lz ? 16 * total : 0);
  const uint64_t o_bwt = o_tok + 16 * ntok;
  const uint64_t o_cnt = (o_bwt + bwt_bytes + 255) & ~255ull;
  const uint64_t o_idx = o_cnt + 4 * n;
  const uint64_t o_blk = (o_idx + 4 * n + 255) & ~255ull;
  const uint64_t out_bytes = o_blk + n * sizeof(LzBlock) + 256;
  if (ws + in_bytes + out_bytes + (1u << 20) > e.budget) { note = "sort + parse workspace exceeds the device budget"; return false; }
  e.io_in.ensure(in_bytes + 64);
  e.io_out.ensure(out_bytes);
  e.arena.ensure(ws);
  e.jobs.ensure((n + 2) * 16 + 64);
  std::vector<const uint8_t*> ptrs(n);
  std::vector<uint64_t> off(n + 1, 0);
  const bool pinned = in_bytes >= (1u << 20) && e.pin_in.ensure(in_bytes + 64);
  std::unique_ptr<uint8_t[]> pageable;
  uint8_t* stage = pinned ? (uint8_t*)e.pin_in.p : (pageable.reset(new uint8_t[in_bytes + 64]), pageable.get());
  for (size_t i = 0; i < n; ++i) {
    ptrs[i] = (const uint8_t*)e.io_in.p + off[i];
    if (jobs[i].n) memcpy(stage + off[i], jobs[i].data, jobs[i].n);
    off[i + 1] = off[i] + jobs[i].n;
  }
  uint8_t* meta = (uint8_t*)e.jobs.p;
  uint8_t* const ob = (uint8_t*)e.io_out.p;
  HIP_CHECK(hipMemcpyAsync(e.io_in.p, stage, total, hipMemcpyHostToDevice, e.stream));
  HIP_CHECK(hipMemcpyAsync(meta, ptrs.data(), n * 8, hipMemcpyHostToDevice, e.stream));
  HIP_CHECK(hipMemcpyAsync(meta + ((n * 8 + 15) & ~15ull), off.data(), (n + 1) * 8, hipMemcpyHostToDevice, e.stream));
  HIP_CHECK(hipMemcpyAsync(ob + o_blk, blk.data(), n * sizeof(LzBlock), hipMemcpyHostToDevice, e.stream));
  HIP_CHECK(hipMemsetAsync(ob + o_cnt, 0, 8 * n, e.stream));
  uint32_t rounds = 0;
  SaSideArrays side;
  hipError_t rc = build_suffix_arrays((const uint8_t* const*)meta, (const uint64_t*)(meta + ((n * 8 + 15) & ~15ull)), (uint32_t)n, total, max_len,
                                      (uint32_t*)ob, e.arena.p, e.arena.cap, e.stream, &rounds, &side);
  if (rc == hipSuccess)
    rc = launch_sort_preprocessors((const uint8_t*)e.io_in.p, (const uint32_t*)ob, side, (const LzBlock*)(ob + o_blk), (uint32_t)n, total, any_lz, any_bwt,
                                   ob + o_res, (LzTok*)(ob + o_tok), (uint32_t*)(ob + o_cnt), ob + o_bwt, (uint32_t*)(ob + o_idx), e.stream);
  if (rc != hipSuccess) { (void)hipGetLastError(); note = std::string("device sort / parse failed: ") + hipGetErrorString(rc); return false; }
  std::vector<uint32_t> cnt(2 * n);
  HIP_CHECK(hipMemcpyAsync(cnt.data(), ob + o_cnt, 8 * n, hipMemcpyDeviceToHost, e.stream));
  HIP_CHECK(hipStreamSynchronize(e.stream));
  static_assert(sizeof(LzTok) == sizeof(LzToken) && sizeof(LzTok) == 16, "token layouts");
  for (size_t i = 0; i < n; ++i) {
    const LzBlock& B = blk[i];
    if (B.kind == 1 || B.kind == 2) {
      if (cnt[i] > B.tok_cap) { note = "LZ77 token list overflowed"; return false; }
      out[i].toks.resize(cnt[i]);
      if (cnt[i]) HIP_CHECK(hipMemcpyAsync(out[i].toks.data(), ob + o_tok + 16 * B.tok_off, 16ull * cnt[i], hipMemcpyDeviceToHost, e.stream));
    } else if (B.kind == 3) {
      out[i].bwt.resize((size_t)B.n + 5);
      HIP_CHECK(hipMemcpyAsync(out[i].bwt.data(), ob + o_bwt + B.off + i, (size_t)B.n + 1, hipMemcpyDeviceToHost, e.stream));
      uint32_t idx = cnt[n + i];
      for (int k = 0; k < 4; ++k) { out[i].bwt[(size_t)B.n + 1 + k] = (U8)idx; idx >>= 8; }
    }
  }
  HIP_CHECK(hipStreamSynchronize(e.stream));
  note = "device, " + std::to_string(rounds) + " doubling rounds";
  return true;
}

int engine_jit_threads() { return jit_threads(); }

int engine_selftest(int32_t out[8]) {
  Engine& e = eng();
  std::lock_guard<std::mutex> g(e.mu);
  require_ready(e);
  bind_device(e);
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
