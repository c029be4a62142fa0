#pragma once
#include <string>
#include <utility>
#include <vector>

#include "../device/plan.hpp"
#include "common.hpp"

namespace zpq {

static const int kCodegenVersion = 4;

// Emits the specialised translation unit for `plan`.  Returns false (with a
// reason) when the chain cannot be specialised (n > 64, MIX wider than a wave,
// untranslatable HCOMP): such plans run on the generic kernels.
// `waves` = blocks per workgroup the kernel is laid out for (4: every side table that fits 30 KiB in LDS,
// one workgroup per CU; 8: half the LDS per block, two wavefronts per SIMD)
bool generate_spec_source(const zpq_plan& plan, int waves, std::string& source, std::string& why_not);

// ---- pipelined encoder (device/pipe_kernel.h) ----
// Dataflow plan of a chain: the level of every unit, the stream/state layout of one group of 64 blocks and the
// unit lists of the five kernels.  The generator prints it into the source as constants; the engine uses the
// same numbers to size buffers and grids.
struct PipeLayout {
  int n = 0;
  int C = 512;                 // chunk: input bytes per unit per step
  int G = 32;                  // blocks per group = active lanes per wavefront (8 / 16 / 32 / 64; measured best: 32)
  int S = 0;                   // ring slots = highest level + 1
  int nctx = 0, nrow = 0, nstate = 0;
  int level[64], ctx[64], row[64], state[64];
  int coder_level = 0, coder_state = 0, hcomp_state = 0;
  int hcomp_lanes = 64;        // blocks per HCOMP workgroup
  bool hcomp_h_lds = false;    // H staged in LDS
  std::vector<std::pair<int, int>> light;   // (PipeKind, component): CONS, CM, MATCH, AVG, MIX2, SSE units and the coder
  std::vector<int> rows, icm, isse, mix, mix_ql;
  std::vector<int> light_sub;  // per light unit: which eighth of the group's blocks (units with a lane per bit position), else 0
  uint64_t off_ctx = 0, off_bh = 0, off_p = 0, off_state = 0, group_bytes = 0;
  // kernel-level dataflow (0 hcomp, 1 rows, 2 light, 3 icm, 4 isse, 5 mix): consumes[c][p] = some unit of kernel c reads a
  // stream some unit of kernel p writes; slack = ring slots beyond the minimum, i.e. how many steps a producer kernel
  // may run ahead of its slowest consumer
  bool consumes[6][6] = {};
  int slack = 3;
  int mix_split = 1;           // MIX wavefronts carry 1 / mix_split of the lanes they could (more, emptier wavefronts)
  // MIX with a lane per (block, bit position, weight quad) -- pipe_kernel.h::pipe_mix_bits_body; needs every MIX of the
  // chain to keep the whole partial byte in its row index and at most 32 inputs (ZPAQ_AMD_PIPE_MIX_BITS=1; off by default:
  // emulator-exact, not yet measured on the MI355X)
  int mix_bits = 0;
  int mix_depth = 3;           // bytes a bit-lane MIX fetches ahead (ZPAQ_AMD_PIPE_MIX_DEPTH, 1..4)
  // CM / MIX2 / SSE with a lane per (block, bit position): workgroups of 64 lanes = 8 blocks, G / 8 of them per group and unit (ZPAQ_AMD_PIPE_LIGHT_BITS=7; off
  // by default: emulator-exact, not yet measured on the MI355X)
  int map_ilp = 1;             // blocks per lane in the ICM / ISSE maps (ZPAQ_AMD_PIPE_MAP_ILP=2|4; experimental, off = 1)
  int full_squash = 0;         // squash from the whole 4096-entry table in LDS (ZPAQ_AMD_PIPE_FULL_SQUASH=1; experimental, off)
  int light_bits = 0;          // 1 CM | 2 MIX2 | 4 SSE
  int light_depth = 3;         // bytes such a unit fetches ahead (ZPAQ_AMD_PIPE_LIGHT_DEPTH, 1..4)
  // ROW units with a lane per (block, nibble): workgroups of 2 x G lanes (ZPAQ_AMD_PIPE_ROW_NIBBLES=1, G <= 32; off by default:
  // emulator-exact, not yet measured on the MI355X)
  int row_nibbles = 0;
  int row_flat = 0;            // one-lane ROW unit with the candidate row picked by masks instead of branches (ZPAQ_AMD_PIPE_ROW_FLAT=1)
  int row_depth = 2;           // bytes such a unit fetches its candidate rows ahead (ZPAQ_AMD_PIPE_ROW_DEPTH, 1..4)
  int light_threads() const { return light_bits ? 64 : G; }                               // workgroup size of the light kernel
  int rows_threads() const { return row_nibbles ? 2 * G : G; }                            // workgroup size of the rows kernel
  int mix_waves_of(int ql) const { return mix_bits ? G * ql / 8 : ql * mix_split; }       // wavefronts per group of one MIX
  int mix_waves_per_group() const { int s = 0; for (int q : mix_ql) s += mix_waves_of(q); return s; }
  int mix_threads() const { return mix_bits ? 64 : G; }                                   // workgroup size of the mix kernel
};
// false + reason when the chain cannot run on the pipelined encoder (then the per-wavefront kernels code it)
bool pipe_layout(const zpq_plan& plan, PipeLayout& out, std::string& why_not);
bool generate_pipe_source(const zpq_plan& plan, std::string& source, std::string& why_not);

// PCOMP translated for the device (device/pcomp_kernel.h); code = PCOMP bytes without the 2 length bytes
bool generate_pcomp_source(const U8* code, size_t len, int ph, int pm, std::string& source, std::string& why_not);

// Cache key of a generated source: SHA-1 over the text (which embeds the codegen
// version) -- the loader extends it with a digest of the kernel template headers.
std::string spec_cache_key(const std::string& source);

}  // namespace zpq
