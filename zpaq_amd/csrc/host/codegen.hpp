#pragma once
#include <string>

#include "../device/plan.hpp"
#include "common.hpp"

namespace zpq {

static const int kCodegenVersion = 2;

// Emits the specialised translation unit for `plan`.  Returns false (with a
// reason) when the chain cannot be specialised (n > 64, MIX wider than a wave,
// untranslatable HCOMP): such plans run on the generic kernels.
bool generate_spec_source(const zpq_plan& plan, std::string& source, std::string& why_not);

// Cache key of a generated source: SHA-1 over the text (which embeds the codegen
// version) -- the loader extends it with a digest of the kernel template headers.
std::string spec_cache_key(const std::string& source);

}  // namespace zpq
