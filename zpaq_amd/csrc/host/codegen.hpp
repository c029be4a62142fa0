#pragma once
#include <string>

#include "../device/plan.hpp"
#include "common.hpp"

namespace zpq {

static const int kCodegenVersion = 3;

// Emits the specialised translation unit for `plan`.  Returns false (with a
// reason) when the chain cannot be specialised (n > 64, MIX wider than a wave,
// untranslatable HCOMP): such plans run on the generic kernels.
// `waves` = blocks per workgroup the kernel is laid out for (4: every side table that fits 30 KiB in LDS,
// one workgroup per CU; 8: half the LDS per block, two wavefronts per SIMD)
bool generate_spec_source(const zpq_plan& plan, int waves, std::string& source, std::string& why_not);

// Cache key of a generated source: SHA-1 over the text (which embeds the codegen
// version) -- the loader extends it with a digest of the kernel template headers.
std::string spec_cache_key(const std::string& source);

}  // namespace zpq
