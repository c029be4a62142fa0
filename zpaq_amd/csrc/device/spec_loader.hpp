#pragma once
#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "plan.hpp"
#include "../host/codegen.hpp"

namespace zpq {

struct SpecKernel {
  hipModule_t module = nullptr;
  hipFunction_t encode = nullptr;
  hipFunction_t decode = nullptr;
  int waves = 4;        // blocks per workgroup
  int threads = 256;    // lanes per workgroup (64 per block; 32 per block for the decoder with two blocks per wavefront)
  std::string origin;   // "cache:<file>" or "hiprtc"
};

// The specialised kernel of `plan` for the current device, or nullptr when it
// is not available (then plan->spec_note says why and the generic kernels run).
// Looks in the in-tree cache (zpaq_amd/spec_cache/<key>.hsaco, filled by
// zpaq_amd/prebuild.py at build time), else compiles with hipRTC and stores the
// code object back into the cache directory when that is writable.
// allow_jit = false: only the in-tree cache is consulted; a miss leaves the plan untried (a later call may
// compile it) and *jit_deferred is set.
// variant 0: 4 blocks per workgroup, variant 1: 8 (see zpq_plan::spec)
SpecKernel* spec_kernel_for(zpq_plan* plan, int variant, bool allow_jit = true, bool* jit_deferred = nullptr,
                            bool* did_jit = nullptr);
// ZPAQ_AMD_SPEC_WAVES=4|8 forces one workgroup shape (tests, experiments, prebuild): its variant, or -1 when unset
int spec_variant_forced();
void spec_kernel_release(zpq_plan* plan);

// The pipelined encoder of a plan (device/pipe_kernel.h): six kernels of one module.
struct PipeKernel {
  hipModule_t module = nullptr;
  hipFunction_t fn[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // hcomp, rows, light, icm, isse, mix
  hipFunction_t repack = nullptr;      // rewrites Predictor::init's MIX tables as packed rows where the chain keeps them so (pipe_repack_body)
  bool any_packed = false;             // ... and whether it has anything to do
  hipFunction_t persist = nullptr;     // the persistent launch (device/pipe_persist.h), when the chain can be packed
  int persist_wg_per_cu = 0;           // workgroups of it a compute unit holds (occupancy API; 0: not asked yet)
  std::string origin;
};
// variant: 0 throughput, 1 latency, 2 latency with long steps (host/codegen.hpp pipe_options); one code object per (header, variant)
PipeKernel* pipe_kernel_for(zpq_plan* plan, int variant, bool allow_jit = true, bool* did_jit = nullptr);
bool pipe_source_and_key(const zpq_plan& plan, const PipeOptions& opt, std::string& source, std::string& key, std::string& why_not);

// A block's PCOMP post-processor translated for the device (device/pcomp_kernel.h), per (program, ph, pm) and device;
// loaded kernels live until the process ends.  nullptr + note when the program cannot be translated or compiled.
struct PcompKernel {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  std::string origin;
};
PcompKernel* pcomp_kernel_for(const U8* code, size_t len, int ph, int pm, std::string& note);
bool pcomp_source_and_key(const U8* code, size_t len, int ph, int pm, std::string& source, std::string& key, std::string& why_not);

// Source text + cache key (with the template-header digest) for prebuilding.
bool spec_source_and_key(const zpq_plan& plan, int variant, std::string& source, std::string& key, std::string& why_not);
// hipRTC compile only (no device needed, nothing loaded or cached): returns the code object size or 0, log filled.
size_t spec_jit_compile_only(const zpq_plan& plan, int variant, std::string& log);
// Compiles (hipRTC, no device needed) the code objects the listed plans would need -- the pipelined encoder when
// `pipe` (and the chain has one; modes[i] = the mode plan i will run in, throughput when `modes` is null), the wavefront
// kernel of `variant` otherwise -- that are neither in the cache directory nor compiled
// earlier in this process, on up to `threads` host threads at once, at most `max_compiles` of them.  Results go to the
// cache directory (when writable) and to an in-process store the loaders look into first.  Returns the number compiled.
int spec_precompile(const std::vector<const zpq_plan*>& plans, bool pipe, int variant, int max_compiles, int threads,
                    std::string* log = nullptr, const std::vector<int>* modes = nullptr);
std::string spec_include_dir();
std::string spec_cache_dir();

}  // namespace zpq
