// Loads (or JIT-compiles) the per-header specialised kernels.
#include "spec_loader.hpp"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>

#include <algorithm>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <vector>

#include "../host/codegen.hpp"

namespace zpq {

namespace {

std::string lib_dir() {
  Dl_info info;
  if (dladdr((const void*)&spec_kernel_for, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    if (k != std::string::npos) return p.substr(0, k);
  }
  return ".";
}

bool read_file(const std::string& path, std::string& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

std::string hex20(const U8* d) {
  char hex[41];
  for (int i = 0; i < 20; ++i) snprintf(hex + 2 * i, 3, "%02x", d[i]);
  return std::string(hex, 40);
}

// Code objects compiled by spec_precompile() in this process that no loader has picked up yet (and those the cache
// directory could not take): key -> code object.
struct MemStore {
  std::mutex mu;
  std::map<std::string, std::vector<char>> code;
};
MemStore& mem_store() {
  static MemStore s;
  return s;
}

bool mem_take(const std::string& key, std::vector<char>& code, bool on_disk) {
  MemStore& m = mem_store();
  std::lock_guard<std::mutex> g(m.mu);
  const auto it = m.code.find(key);
  if (it == m.code.end()) return false;
  code = it->second;
  if (on_disk) m.code.erase(it);          // the cache file serves later loads (other devices, later processes)
  return true;
}

bool file_exists(const std::string& path) {
  struct stat sb;
  return ::stat(path.c_str(), &sb) == 0 && sb.st_size > 0;
}

}  // namespace

std::string spec_include_dir() {
  if (const char* e = getenv("ZPAQ_AMD_DEVICE_INCLUDE")) return e;
  return lib_dir() + "/csrc/device";
}
std::string spec_cache_dir() {
  if (const char* e = getenv("ZPAQ_AMD_SPEC_CACHE")) return e;
  return lib_dir() + "/spec_cache";
}

int spec_variant_forced() {
  const char* e = getenv("ZPAQ_AMD_SPEC_WAVES");
  if (!e || !e[0]) return -1;
  const int w = atoi(e);
  return w == 8 ? 1 : 0;
}

bool spec_source_and_key(const zpq_plan& plan, int variant, std::string& source, std::string& key, std::string& why_not) {
  // variant 0 / 1: one block per wavefront, 4 / 8 blocks per workgroup; 2: the decoder with two blocks per wavefront;
  // 3: the lockstep decoder (row / mixer wavefronts, device/spec_team_kernel.h)
  if (!generate_spec_source(plan, variant >= 1 ? 8 : 4, source, why_not, variant == 2 ? 1 : (variant == 3 ? 2 : 0))) return false;
  std::string h1, h2, h3, h4;
  const std::string inc = spec_include_dir();
  if (!read_file(inc + "/spec_kernel.h", h1) || !read_file(inc + "/layout.h", h2) ||
      (variant >= 2 && !read_file(inc + "/spec_dual_kernel.h", h3)) ||
      (variant == 3 && !read_file(inc + "/spec_team_kernel.h", h4))) {
    why_not = "kernel template headers not found under " + inc;
    return false;
  }
  Sha1 s;
  s.update(source.data(), source.size());
  s.update(h1.data(), h1.size());
  s.update(h2.data(), h2.size());
  s.update(h3.data(), h3.size());
  s.update(h4.data(), h4.size());
  if (const char* defs = getenv("ZPAQ_AMD_SPEC_DEFS")) s.update(defs, strlen(defs));   // e.g. -DZPQ_PROF
  key = hex20(s.result());
  return true;
}

bool pipe_source_and_key(const zpq_plan& plan, const PipeOptions& opt, std::string& source, std::string& key, std::string& why_not) {
  if (!generate_pipe_source(plan, opt, source, why_not)) return false;
  std::string h1, h2, h3, h4;
  const std::string inc = spec_include_dir();
  if (!read_file(inc + "/spec_kernel.h", h1) || !read_file(inc + "/layout.h", h2) || !read_file(inc + "/pipe_kernel.h", h3) ||
      !read_file(inc + "/pipe_persist.h", h4)) {
    why_not = "kernel template headers not found under " + inc;
    return false;
  }
  Sha1 s;
  s.update(source.data(), source.size());
  s.update(h1.data(), h1.size());
  s.update(h2.data(), h2.size());
  s.update(h3.data(), h3.size());
  s.update(h4.data(), h4.size());
  if (const char* defs = getenv("ZPAQ_AMD_SPEC_DEFS")) s.update(defs, strlen(defs));
  key = hex20(s.result());
  return true;
}

// -simplifycfg-sink-common=false: store sinking across the kernel's big if/else ladders
// otherwise forces register state into scratch (see prebuild.py, same flags)
static std::vector<std::string> jit_options() {
  std::vector<std::string> o = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + spec_include_dir(), "-Wno-unused-label",
                                "-mllvm", "-simplifycfg-sink-common=false"};
  const char* defs = getenv("ZPAQ_AMD_SPEC_DEFS");
  if (defs && defs[0]) {                       // extra options separated by blanks, e.g. "-DZPQ_PROF -DZPQ_TEAM_EARLY2=0"
    std::string cur;
    for (const char* c = defs;; ++c) {
      if (*c == ' ' || *c == 0) {
        if (!cur.empty()) o.push_back(cur);
        cur.clear();
        if (*c == 0) break;
      } else {
        cur.push_back(*c);
      }
    }
  }
  return o;
}

static bool compile_hiprtc(const std::string& source, std::vector<char>& code, std::string& log) {
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, source.c_str(), "zpq_spec.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    log = "hiprtcCreateProgram failed";
    return false;
  }
  const std::vector<std::string> o = jit_options();
  std::vector<const char*> opts;
  for (const std::string& x : o) opts.push_back(x.c_str());
  const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  size_t ls = 0;
  if (hiprtcGetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
    log.resize(ls);
    hiprtcGetProgramLog(prog, &log[0]);
  }
  if (r != HIPRTC_SUCCESS) { hiprtcDestroyProgram(&prog); return false; }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  code.resize(cs);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  return true;
}

static bool write_cache_file(const std::string& key, const std::vector<char>& code) {
  ::mkdir(spec_cache_dir().c_str(), 0755);
  const std::string path = spec_cache_dir() + "/" + key + ".hsaco";
  // (a name of its own per writer: two threads or processes may compile the same key at the same time)
  char suffix[64];
  snprintf(suffix, sizeof suffix, ".tmp%ld_%p", (long)getpid(), (const void*)&code);
  std::ofstream f(path + suffix, std::ios::binary);
  if (!f) return false;
  f.write(code.data(), (std::streamsize)code.size());
  f.close();
  if (!f) { ::unlink((path + suffix).c_str()); return false; }
  return ::rename((path + suffix).c_str(), path.c_str()) == 0;
}

extern "C" char** environ;

struct JitItem { std::string source, key; };

// hipRTC compiles one program at a time per process, so several headers are compiled by several helper processes
// (zpq_jitc, next to the library): sources and code objects travel through a scratch directory.  Returns the number of
// code objects made, or -1 when the helper is not there.
static int precompile_in_processes(const std::vector<JitItem>& todo, int procs, std::string* log) {
  const std::string helper = lib_dir() + "/zpq_jitc";
  if (::access(helper.c_str(), X_OK) != 0) return -1;
  const char* tmp = getenv("TMPDIR");
  std::string dir = std::string(tmp && tmp[0] ? tmp : "/tmp") + "/zpq_jit_XXXXXX";
  if (!::mkdtemp(&dir[0])) return -1;
  const std::vector<std::string> o = jit_options();
  struct Child { pid_t pid = -1; size_t item = 0; };
  std::vector<Child> running;
  size_t next = 0;
  int done = 0;
  auto src_of = [&](size_t i) { return dir + "/" + todo[i].key + ".hip"; };
  auto out_of = [&](size_t i) { return dir + "/" + todo[i].key + ".hsaco"; };
  auto err_of = [&](size_t i) { return dir + "/" + todo[i].key + ".log"; };
  auto reap = [&](const Child& c, int status) {
    std::string blob;
    // (status -1: the host application ignores SIGCHLD, the exit status is gone -- the helper renames its output into
    // place only when it is complete, so the file says whether it worked)
    if ((status == -1 || (WIFEXITED(status) && WEXITSTATUS(status) == 0)) && read_file(out_of(c.item), blob) && !blob.empty()) {
      std::vector<char> code(blob.begin(), blob.end());
      (void)write_cache_file(todo[c.item].key, code);
      MemStore& m = mem_store();
      std::lock_guard<std::mutex> g(m.mu);
      m.code[todo[c.item].key] = std::move(code);
      ++done;
    } else if (log) {
      std::string l;
      read_file(err_of(c.item), l);
      *log += todo[c.item].key + ": " + l.substr(0, 2000) + "\n";
    }
    ::unlink(src_of(c.item).c_str());
    ::unlink(out_of(c.item).c_str());
    ::unlink(err_of(c.item).c_str());
  };
  while (next < todo.size() || !running.empty()) {
    while (next < todo.size() && (int)running.size() < procs) {
      const size_t i = next++;
      { std::ofstream f(src_of(i), std::ios::binary); f.write(todo[i].source.data(), (std::streamsize)todo[i].source.size()); }
      const std::string src = src_of(i), out = out_of(i), err = err_of(i);
      std::vector<char*> argv = {(char*)helper.c_str(), (char*)src.c_str(), (char*)out.c_str()};
      for (const std::string& x : o) argv.push_back((char*)x.c_str());
      argv.push_back(nullptr);
      posix_spawn_file_actions_t fa;
      posix_spawn_file_actions_init(&fa);
      posix_spawn_file_actions_addopen(&fa, 2, err.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
      posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);
      Child c;
      c.item = i;
      const int rc = posix_spawn(&c.pid, helper.c_str(), &fa, nullptr, argv.data(), environ);
      posix_spawn_file_actions_destroy(&fa);
      if (rc != 0) {
        if (log) *log += todo[i].key + ": cannot start " + helper + "\n";
        ::unlink(src.c_str());
        continue;
      }
      running.push_back(c);
    }
    if (running.empty()) break;
    // wait for ONE OF OURS (the host application may have children of its own: never wait for "any child")
    bool reaped = false;
    for (size_t k = 0; k < running.size(); ++k) {
      int status = 0;
      const pid_t r = ::waitpid(running[k].pid, &status, WNOHANG);
      if (r == running[k].pid || (r < 0 && errno != EINTR)) {
        reap(running[k], r < 0 ? -1 : status);
        running.erase(running.begin() + (long)k);
        reaped = true;
        break;
      }
    }
    if (!reaped) ::usleep(5000);
  }
  ::rmdir(dir.c_str());
  return done;
}

int spec_precompile(const std::vector<const zpq_plan*>& plans, bool pipe, int variant, int max_compiles, int threads,
                    std::string* log, const std::vector<int>* modes) {
  // one caller at a time in the whole process: every call starts up to `threads` helper processes (hipRTC + the code
  // generator's libraries each), and the engines of a multi-device process all come here at once with the same chains --
  // whoever waited finds them in the in-process store afterwards
  static std::mutex one_at_a_time;
  std::lock_guard<std::mutex> turn(one_at_a_time);
  std::vector<JitItem> todo;
  std::vector<std::string> seen;
  for (size_t pi = 0; pi < plans.size(); ++pi) {
    const zpq_plan* p = plans[pi];
    if ((int)todo.size() >= max_compiles) break;
    if (!p || !p->hdr().wave_ok) continue;
    JitItem it;
    std::string why;
    bool have = false;
    if (pipe) have = pipe_source_and_key(*p, pipe_options(modes && pi < modes->size() ? (*modes)[pi] : 0), it.source, it.key, why);
    // a chain the wanted decoder shape does not take (more than 32 components, an ISSE fed from afar ...) gets the next
    // one down, as kernel_kind() will choose it: 3 lockstep -> 2 two blocks per wavefront -> 1 one block per wavefront
    for (int v = variant; !pipe && !have && v >= 1; --v) have = spec_source_and_key(*p, v, it.source, it.key, why);
    if (!have && (pipe || variant < 1)) have = spec_source_and_key(*p, variant, it.source, it.key, why);
    if (!have) continue;
    bool dup = false;
    for (const std::string& k : seen) dup = dup || k == it.key;
    if (dup) continue;
    seen.push_back(it.key);
    if (file_exists(spec_cache_dir() + "/" + it.key + ".hsaco")) continue;
    {
      MemStore& m = mem_store();
      std::lock_guard<std::mutex> g(m.mu);
      if (m.code.count(it.key)) continue;
    }
    todo.push_back(std::move(it));
  }
  if (todo.empty()) return 0;
  if (todo.size() > 1 && threads > 1) {
    const int n = precompile_in_processes(todo, threads, log);
    if (n >= 0) return n;                   // (< 0: no helper here -- compile in this process, one after the other)
  }
  int done = 0;
  for (const JitItem& it : todo) {
    std::vector<char> code;
    std::string l;
    if (!compile_hiprtc(it.source, code, l)) {
      if (log) *log += it.key + ": " + l.substr(0, 2000) + "\n";
      continue;                           // the loader will try again on its own and report the failure in the plan's note
    }
    (void)write_cache_file(it.key, code);
    MemStore& m = mem_store();
    std::lock_guard<std::mutex> g(m.mu);
    m.code[it.key] = std::move(code);
    ++done;
  }
  return done;
}

size_t spec_jit_compile_only(const zpq_plan& plan, int variant, std::string& log) {
  std::string source, key, why;
  if (!spec_source_and_key(plan, variant, source, key, why)) { log = why; return 0; }
  std::vector<char> code;
  if (!compile_hiprtc(source, code, log)) return 0;
  return code.size();
}

SpecKernel* spec_kernel_for(zpq_plan* plan, int variant, bool allow_jit, bool* jit_deferred, bool* did_jit) {
  if (variant < 0 || variant > 3) variant = 0;
  if (plan->cur().spec_state[variant] > 0) return (SpecKernel*)plan->cur().spec[variant];
  if (plan->cur().spec_state[variant] < 0) return nullptr;
  plan->cur().spec_state[variant] = -1;
  std::string source, key, why;
  if (!spec_source_and_key(*plan, variant, source, key, why)) { plan->cur().spec_note = why; return nullptr; }
  std::vector<char> code;
  std::string origin;
  const std::string path = spec_cache_dir() + "/" + key + ".hsaco";
  std::string blob;
  if (mem_take(key, code, file_exists(path))) {
    origin = "hiprtc";                      // compiled by spec_precompile() in this process
  } else if (read_file(path, blob) && !blob.empty()) {
    code.assign(blob.begin(), blob.end());
    origin = "cache:" + key;
  } else {
    if (!allow_jit) {                       // not prebuilt and the caller's JIT budget is spent
      plan->cur().spec_state[variant] = 0;
      plan->cur().spec_note = "hipRTC compile deferred (JIT budget of this batch spent)";
      if (jit_deferred) *jit_deferred = true;
      return nullptr;
    }
    if (did_jit) *did_jit = true;
    std::string log;
    if (!compile_hiprtc(source, code, log)) {
      plan->cur().spec_note = "hipRTC compile failed: " + log.substr(0, 2000);
      return nullptr;
    }
    origin = "hiprtc";
    ::mkdir(spec_cache_dir().c_str(), 0755);
    std::ofstream f(path + ".tmp", std::ios::binary);
    if (f) {
      f.write(code.data(), (std::streamsize)code.size());
      f.close();
      ::rename((path + ".tmp").c_str(), path.c_str());
    }
  }
  SpecKernel* k = new SpecKernel;
  const bool team = variant == 3;           // the decoder alone, 8 blocks of a workgroup in lockstep (device/spec_team_kernel.h)
  const bool dual = variant == 2 || team;   // the decoder alone, two blocks per wavefront (device/spec_dual_kernel.h)
  if (hipModuleLoadData(&k->module, code.data()) != hipSuccess ||
      (!dual && hipModuleGetFunction(&k->encode, k->module, "zpq_spec_encode") != hipSuccess) ||
      hipModuleGetFunction(&k->decode, k->module, team ? "zpq_spec_decode3" : (dual ? "zpq_spec_decode2" : "zpq_spec_decode")) != hipSuccess) {
    plan->cur().spec_note = "hipModuleLoadData failed for " + origin;
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
    return nullptr;
  }
  int maxthr = 0;
  if (team) { k->waves = 8; k->threads = team_threads(*plan); }
  else if (dual) { k->waves = 8; k->threads = 256; }
  else if (hipFuncGetAttribute(&maxthr, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, k->encode) == hipSuccess && maxthr >= 64) {
    k->waves = maxthr / 64;
    k->threads = 64 * k->waves;
  }
  k->origin = origin;
  plan->cur().spec[variant] = k;
  plan->cur().spec_state[variant] = 1;
  plan->cur().spec_note = origin;
  return k;
}

PipeKernel* pipe_kernel_for(zpq_plan* plan, int mode, bool allow_jit, bool* did_jit) {
  if (mode < 0 || mode >= kPipeVariants) mode = 0;
  if (plan->cur().pipe_state[mode] > 0) return (PipeKernel*)plan->cur().pipe[mode];
  if (plan->cur().pipe_state[mode] < 0) return nullptr;
  plan->cur().pipe_state[mode] = -1;
  std::string source, key, why;
  if (!pipe_source_and_key(*plan, pipe_options(mode), source, key, why)) { plan->cur().pipe_note = why; return nullptr; }
  std::vector<char> code;
  std::string origin, blob;
  const std::string path = spec_cache_dir() + "/" + key + ".hsaco";
  if (mem_take(key, code, file_exists(path))) {
    origin = "hiprtc";                      // compiled by spec_precompile() in this process
  } else if (read_file(path, blob) && !blob.empty()) {
    code.assign(blob.begin(), blob.end());
    origin = "cache:" + key;
  } else {
    if (!allow_jit) {
      plan->cur().pipe_state[mode] = 0;
      plan->cur().pipe_note = "hipRTC compile deferred (JIT budget of this batch spent)";
      return nullptr;
    }
    if (did_jit) *did_jit = true;
    std::string log;
    if (!compile_hiprtc(source, code, log)) {
      plan->cur().pipe_note = "hipRTC compile failed: " + log.substr(0, 2000);
      return nullptr;
    }
    origin = "hiprtc";
    ::mkdir(spec_cache_dir().c_str(), 0755);
    std::ofstream f(path + ".tmp", std::ios::binary);
    if (f) {
      f.write(code.data(), (std::streamsize)code.size());
      f.close();
      ::rename((path + ".tmp").c_str(), path.c_str());
    }
  }
  PipeKernel* k = new PipeKernel;
  static const char* names[6] = {"zpq_pipe_hcomp", "zpq_pipe_rows", "zpq_pipe_light", "zpq_pipe_icm", "zpq_pipe_isse", "zpq_pipe_mix"};
  bool ok = hipModuleLoadData(&k->module, code.data()) == hipSuccess;
  for (int i = 0; ok && i < 6; ++i) ok = hipModuleGetFunction(&k->fn[i], k->module, names[i]) == hipSuccess;
  ok = ok && hipModuleGetFunction(&k->repack, k->module, "zpq_pipe_repack") == hipSuccess;
  if (!ok) {
    plan->cur().pipe_note = "hipModuleLoadData failed for " + origin;
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
    return nullptr;
  }
  if (hipModuleGetFunction(&k->persist, k->module, "zpq_pipe_persist") != hipSuccess) { (void)hipGetLastError(); k->persist = nullptr; }
  {
    PipeLayout PL;
    std::string why2;
    if (pipe_layout(*plan, pipe_options(mode), PL, why2)) for (int v : PL.mix_packed) k->any_packed = k->any_packed || v != 0;
  }
  k->origin = origin;
  plan->cur().pipe[mode] = k;
  plan->cur().pipe_state[mode] = 1;
  plan->cur().pipe_note = origin;
  return k;
}

bool pcomp_source_and_key(const U8* code, size_t len, int ph, int pm, std::string& source, std::string& key, std::string& why_not) {
  if (!generate_pcomp_source(code, len, ph, pm, source, why_not)) return false;
  std::string h1, h2, h3;
  const std::string inc = spec_include_dir();
  if (!read_file(inc + "/spec_kernel.h", h1) || !read_file(inc + "/layout.h", h2) || !read_file(inc + "/pcomp_kernel.h", h3)) {
    why_not = "kernel template headers not found under " + inc;
    return false;
  }
  Sha1 s;
  s.update(source.data(), source.size());
  s.update(h1.data(), h1.size());
  s.update(h2.data(), h2.size());
  s.update(h3.data(), h3.size());
  key = hex20(s.result());
  return true;
}

PcompKernel* pcomp_kernel_for(const U8* code, size_t len, int ph, int pm, std::string& note) {
  static std::mutex mu;
  static std::map<std::pair<int, std::string>, PcompKernel*> loaded;     // (device, cache key)
  std::string source, key, why;
  if (!pcomp_source_and_key(code, len, ph, pm, source, key, why)) { note = why; return nullptr; }
  const int dev = plan_device_index();
  std::lock_guard<std::mutex> g(mu);
  const auto it = loaded.find({dev, key});
  if (it != loaded.end()) { note = it->second ? it->second->origin : "unavailable"; return it->second; }
  std::vector<char> bin;
  std::string origin, blob;
  const std::string path = spec_cache_dir() + "/" + key + ".hsaco";
  if (read_file(path, blob) && !blob.empty()) {
    bin.assign(blob.begin(), blob.end());
    origin = "cache:" + key;
  } else {
    std::string log;
    if (!compile_hiprtc(source, bin, log)) {
      note = "hipRTC compile failed: " + log.substr(0, 2000);
      loaded[{dev, key}] = nullptr;
      return nullptr;
    }
    origin = "hiprtc";
    ::mkdir(spec_cache_dir().c_str(), 0755);
    std::ofstream f(path + ".tmp", std::ios::binary);
    if (f) {
      f.write(bin.data(), (std::streamsize)bin.size());
      f.close();
      ::rename((path + ".tmp").c_str(), path.c_str());
    }
  }
  PcompKernel* k = new PcompKernel;
  if (hipModuleLoadData(&k->module, bin.data()) != hipSuccess ||
      hipModuleGetFunction(&k->fn, k->module, "zpq_pcomp_run") != hipSuccess) {
    note = "hipModuleLoadData failed for " + origin;
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
    loaded[{dev, key}] = nullptr;
    return nullptr;
  }
  k->origin = origin;
  note = origin;
  loaded[{dev, key}] = k;
  return k;
}

void spec_kernel_release(zpq_plan* plan) {
  for (int m = 0; plan && m < kPipeVariants; ++m) {
    if (!plan->cur().pipe[m]) continue;
    PipeKernel* k = (PipeKernel*)plan->cur().pipe[m];
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
    plan->cur().pipe[m] = nullptr;
    plan->cur().pipe_state[m] = 0;
  }
  for (int v = 0; plan && v < 4; ++v) {
    if (!plan->cur().spec[v]) continue;
    SpecKernel* k = (SpecKernel*)plan->cur().spec[v];
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
    plan->cur().spec[v] = nullptr;
    plan->cur().spec_state[v] = 0;
  }
}

}  // namespace zpq
