// Loads (or JIT-compiles) the per-header specialised kernels.
#include "spec_loader.hpp"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>

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
  if (!generate_spec_source(plan, variant == 1 ? 8 : 4, source, why_not)) return false;
  std::string h1, h2;
  const std::string inc = spec_include_dir();
  if (!read_file(inc + "/spec_kernel.h", h1) || !read_file(inc + "/layout.h", h2)) {
    why_not = "kernel template headers not found under " + inc;
    return false;
  }
  Sha1 s;
  s.update(source.data(), source.size());
  s.update(h1.data(), h1.size());
  s.update(h2.data(), h2.size());
  if (const char* defs = getenv("ZPAQ_AMD_SPEC_DEFS")) s.update(defs, strlen(defs));   // e.g. -DZPQ_PROF
  key = hex20(s.result());
  return true;
}

bool pipe_source_and_key(const zpq_plan& plan, std::string& source, std::string& key, std::string& why_not) {
  if (!generate_pipe_source(plan, source, why_not)) return false;
  std::string h1, h2, h3;
  const std::string inc = spec_include_dir();
  if (!read_file(inc + "/spec_kernel.h", h1) || !read_file(inc + "/layout.h", h2) || !read_file(inc + "/pipe_kernel.h", h3)) {
    why_not = "kernel template headers not found under " + inc;
    return false;
  }
  Sha1 s;
  s.update(source.data(), source.size());
  s.update(h1.data(), h1.size());
  s.update(h2.data(), h2.size());
  s.update(h3.data(), h3.size());
  if (const char* defs = getenv("ZPAQ_AMD_SPEC_DEFS")) s.update(defs, strlen(defs));
  key = hex20(s.result());
  return true;
}

static bool compile_hiprtc(const std::string& source, std::vector<char>& code, std::string& log) {
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, source.c_str(), "zpq_spec.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    log = "hiprtcCreateProgram failed";
    return false;
  }
  const std::string inc = "-I" + spec_include_dir();
  // -simplifycfg-sink-common=false: store sinking across the kernel's big if/else ladders
  // otherwise forces register state into scratch (see prebuild.py, same flags)
  std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", inc.c_str(), "-Wno-unused-label",
                                   "-mllvm", "-simplifycfg-sink-common=false"};
  const char* defs = getenv("ZPAQ_AMD_SPEC_DEFS");
  if (defs && defs[0]) opts.push_back(defs);   // a single extra option, e.g. -DZPQ_PROF
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

size_t spec_jit_compile_only(const zpq_plan& plan, int variant, std::string& log) {
  std::string source, key, why;
  if (!spec_source_and_key(plan, variant, source, key, why)) { log = why; return 0; }
  std::vector<char> code;
  if (!compile_hiprtc(source, code, log)) return 0;
  return code.size();
}

SpecKernel* spec_kernel_for(zpq_plan* plan, int variant, bool allow_jit, bool* jit_deferred, bool* did_jit) {
  if (variant < 0 || variant > 1) variant = 0;
  if (plan->cur().spec_state[variant] > 0) return (SpecKernel*)plan->cur().spec[variant];
  if (plan->cur().spec_state[variant] < 0) return nullptr;
  plan->cur().spec_state[variant] = -1;
  if (getenv("ZPAQ_AMD_NO_SPEC")) { plan->cur().spec_note = "disabled by ZPAQ_AMD_NO_SPEC"; return nullptr; }
  std::string source, key, why;
  if (!spec_source_and_key(*plan, variant, source, key, why)) { plan->cur().spec_note = why; return nullptr; }
  std::vector<char> code;
  std::string origin;
  const std::string path = spec_cache_dir() + "/" + key + ".hsaco";
  std::string blob;
  if (read_file(path, blob) && !blob.empty()) {
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
  if (hipModuleLoadData(&k->module, code.data()) != hipSuccess ||
      hipModuleGetFunction(&k->encode, k->module, "zpq_spec_encode") != hipSuccess ||
      hipModuleGetFunction(&k->decode, k->module, "zpq_spec_decode") != hipSuccess) {
    plan->cur().spec_note = "hipModuleLoadData failed for " + origin;
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
    return nullptr;
  }
  int maxthr = 0;
  if (hipFuncGetAttribute(&maxthr, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, k->encode) == hipSuccess && maxthr >= 64)
    k->waves = maxthr / 64;
  k->origin = origin;
  plan->cur().spec[variant] = k;
  plan->cur().spec_state[variant] = 1;
  plan->cur().spec_note = origin;
  return k;
}

PipeKernel* pipe_kernel_for(zpq_plan* plan, bool allow_jit, bool* did_jit) {
  if (plan->cur().pipe_state > 0) return (PipeKernel*)plan->cur().pipe;
  if (plan->cur().pipe_state < 0) return nullptr;
  plan->cur().pipe_state = -1;
  if (getenv("ZPAQ_AMD_NO_PIPE")) { plan->cur().pipe_note = "disabled by ZPAQ_AMD_NO_PIPE"; return nullptr; }
  std::string source, key, why;
  if (!pipe_source_and_key(*plan, source, key, why)) { plan->cur().pipe_note = why; return nullptr; }
  std::vector<char> code;
  std::string origin, blob;
  const std::string path = spec_cache_dir() + "/" + key + ".hsaco";
  if (read_file(path, blob) && !blob.empty()) {
    code.assign(blob.begin(), blob.end());
    origin = "cache:" + key;
  } else {
    if (!allow_jit) {
      plan->cur().pipe_state = 0;
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
  if (!ok) {
    plan->cur().pipe_note = "hipModuleLoadData failed for " + origin;
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
    return nullptr;
  }
  k->origin = origin;
  plan->cur().pipe = k;
  plan->cur().pipe_state = 1;
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
  if (plan && plan->cur().pipe) {
    PipeKernel* k = (PipeKernel*)plan->cur().pipe;
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
    plan->cur().pipe = nullptr;
    plan->cur().pipe_state = 0;
  }
  for (int v = 0; plan && v < 2; ++v) {
    if (!plan->cur().spec[v]) continue;
    SpecKernel* k = (SpecKernel*)plan->cur().spec[v];
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
    plan->cur().spec[v] = nullptr;
    plan->cur().spec_state[v] = 0;
  }
}

}  // namespace zpq
