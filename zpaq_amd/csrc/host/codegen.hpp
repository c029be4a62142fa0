#pragma once
#include <string>
#include <utility>
#include <vector>

#include "../device/plan.hpp"
#include "common.hpp"

namespace zpq {

static const int kCodegenVersion = 6;

// Emits the specialised translation unit for `plan`.  Returns false (with a
// reason) when the chain cannot be specialised (n > 64, MIX wider than a wave,
// untranslatable HCOMP): such plans run on the generic kernels.
// `waves` = blocks per workgroup the kernel is laid out for (4: every side table that fits 30 KiB in LDS,
// one workgroup per CU; 8: half the LDS per block, two wavefronts per SIMD)
// shape 1 ("dual"): the decoder with two blocks per wavefront (device/spec_dual_kernel.h; chains of up to 32 components, waves = 8's LDS plan)
// shape 2 ("team"): the decoder with the 8 blocks of a workgroup in lockstep, ICM / ISSE components on row wavefronts and the
//                   rest on mixer wavefronts (device/spec_team_kernel.h; the same LDS plan; see team_threads for the workgroup size)
bool generate_spec_source(const zpq_plan& plan, int waves, std::string& source, std::string& why_not, int shape = 0);
// threads per workgroup of the lockstep decoder for this chain (384: up to 16 ICM / ISSE components, 512: up to 32)
int team_threads(const zpq_plan& plan);

// ---- pipelined encoder (device/pipe_kernel.h) ----
// Dataflow plan of a chain: the level of every unit, the stream/state layout of one group of blocks and the
// unit lists of the six kernels.  The generator prints it into the source as constants; the engine uses the
// same numbers to size buffers and grids.
// How the generator lays a chain out.  `mode` is the one real choice (device/pipe_kernel.h, "two shapes of one unit"):
//   0 THROUGHPUT  a lane per block for everything but SSE -- the batch fills the machine and every HBM transaction counts
//   1 LATENCY     MIX, CM and MIX2 with a lane per (block, bit position) as well: 6 x shorter per-byte chains, more
//                 wavefronts and more requests; faster below ~400 blocks of a chain, slower above (profiles/r03)
// The engine picks the mode per chain from the number of blocks the batch holds of it.  chunk / group are test and
// experiment parameters (0 = the defaults: 512 bytes per step, 32 blocks per group).
struct PipeOptions {
  int mode = 0;
  int chunk = 0;
  int group = 0;
  bool persist = true;         // also emit the persistent launch (device/pipe_persist.h) when the chain can be packed
  bool wide = false;           // latency shape with a wavefront per SIMD (workgroups of 4): twice the workgroups per group
};
// The product's variants of a chain's encoder: 0 = throughput shape, 1 = latency shape, 2 = latency shape with steps
// of 2048 bytes instead of 512.  A step costs a fixed ~0.3 ms of launches and dependencies between the six streams
// whatever it holds; a batch that fills the GPU hides that behind 2 ms of work, a small one does not (-m5 on 64 blocks:
// 0.98 ms per 512-byte step against 0.65 ms for the longest chain), so blocks long enough to fill a long pipeline
// (128 KiB and more: 16 levels x 2048 bytes of fill) take four times fewer, four times longer steps -- as long as the
// streams of one step stay cache-sized (engine.cpp::pipe_mode_for has the rule and the measurements).  3 = latency shape with
// a wavefront per SIMD (workgroups of 4 wavefronts: twice the workgroups per group), taken by the persistent launch while those
// all fit the device (round 6; a chain of up to 32 unit wavefronts has that shape as its variant 1 already).
static const int kPipeVariants = 4;
PipeOptions pipe_options(int variant);

struct PipeLayout {
  int n = 0;
  int C = 512;                 // chunk: input bytes per unit per step
  int G = 32;                  // blocks per group = active lanes per wavefront (8 / 16 / 32 / 64; measured best: 32)
  int S = 0;                   // ring slots = highest level + 1 + slack
  int nctx = 0, nrow = 0, nstate = 0;
  int level[64], ctx[64], row[64], state[64];
  int coder_level = 0, coder_state = 0, hcomp_state = 0;
  int hcomp_lanes = 64;        // blocks per HCOMP workgroup
  bool hcomp_h_lds = false;    // H staged in LDS
  bool hcomp_m_lds = false;    // persistent launch: M (up to 256 bytes) lives in LDS beside H
  std::vector<std::pair<int, int>> light;   // (PipeKind, component): CONS, CM, MATCH, AVG, MIX2, SSE units and the coder
  std::vector<int> rows, icm, isse, mix, mix_ql;
  std::vector<int> mix_packed;   // per MIX role: 1 = weight rows as 24-bit quads (device/pipe_kernel.h pipe_mix_packed_unit); the arena holds them so
  std::vector<int> mix_lds_rows; // per MIX role, persistent launch: rows [0, n) of the packed table live in the unit's LDS (0: none)
  std::vector<int> light_sub;  // per light unit: which eighth of the group's blocks (units with a lane per bit position), else 0
  uint64_t off_ctx = 0, off_bh = 0, off_p = 0, off_state = 0, group_bytes = 0;
  // kernel-level dataflow (0 hcomp, 1 rows, 2 light, 3 icm, 4 isse, 5 mix): consumes[c][p] = some unit of kernel c reads a
  // stream some unit of kernel p writes; slack = ring slots beyond the minimum, i.e. how many steps a producer kernel
  // may run ahead of its slowest consumer
  bool consumes[6][6] = {};
  int slack = 3;
  int mode = 0;
  int mix_bits = 0;            // MIX with a lane per (block, bit position, weight quad): latency mode, when every MIX of the chain allows it
  int light_bits = 4;          // which light units have a lane per (block, bit position): 1 CM | 2 MIX2 | 4 SSE (SSE always: it wins everywhere)
  int depth = 3;               // bytes a bit-lane unit fetches ahead (measured 1..4 on the MI355X: flat, 3 never worse)
  // ---- the persistent launch (device/pipe_persist.h): the units of a group packed into workgroups that are resident together ----
  struct Slot {
    int kind = -1;             // 0 hcomp, 1 row, 2 light, 3 icm, 4 isse, 5 mix (= the six kernels); -1: an idle wavefront
    int role = 0;              // index in that kernel's role list (rows / light / icm / isse / mix)
    int sub = 0;               // which wavefront of a unit that has several (MIX lane groups, HCOMP with fewer lanes than a group has blocks)
    int unit = 0;              // progress counter the wavefront bumps
    int lds = 0;               // bytes of private LDS (HCOMP: H, ICM / ISSE: the side tables of the group)
    int lds_off = 0;           // where in the workgroup's LDS
    float cost = 0;            // relative time per chunk (packing heuristic only)
    float lines = 0;           // distinct memory lines of model state the wavefront asks for per input byte, lines that stay on the die at half weight (packing: what a compute unit can have in flight is what paces its wavefronts)
  };
  struct Dep { int unit, lag, mult; };   // wait for progress[unit] >= mult * (chunk + 1 - lag)
  bool persist_ok = false;
  std::string persist_why;
  int ps_waves = 0;            // wavefronts per workgroup
  int ps_wpg = 0;              // workgroups per group
  int ps_nunit = 0;
  int ps_lds_bytes = 0;        // per workgroup: shared tables + the largest flavour's private tables
  bool ps_coder_fast = false;  // latency shape: the coder with one store per bit (device pipe_coder_fast; 16 KiB of LDS)
  bool ps_row_ring = false;    // lane-per-block ROW units with the table two bytes ahead (device pipe_row_ring)
  int ps_ahead = 0;            // a small chain's units read their streams ps_ahead + 1 bytes ahead
  bool ps_wide = false;        // variant 3: workgroups of 4 wavefronts whatever the chain's size
  bool ps_icm_full = false;    // ICM maps with the whole stretch table behind their side table (not variant 1 of a larger chain)
  bool ps_row_halves = false;  // ROW units with a lane per nibble (the small chains proper; not variant 3 of a larger chain)
  bool ps_small = false;       // a chain of at most 16 unit wavefronts in the latency shape: one wavefront per SIMD, ISSE pairs unpacked
  int ps_mix_nh = 1;           // lane groups a MIX unit gives a block: 2 = bits 0 .. 3 and bits 4 .. 7 apart (half the chain per byte)
  std::vector<Slot> ps_slots;  // ps_wpg * ps_waves, flavour-major
  std::vector<std::vector<Dep>> ps_deps;   // per slot
  int light_threads() const { return 64; }                                                // workgroup size of the light kernel (bit-lane units: 8 blocks x 8 positions)
  int mix_waves_of(int ql) const { return mix_bits ? G * ql / 8 : ql; }                   // wavefronts per group of one MIX
  int mix_waves_per_group() const { int s = 0; for (int q : mix_ql) s += mix_waves_of(q); return s; }
  int mix_threads() const { return mix_bits ? 64 : G; }                                   // workgroup size of the mix kernel
};
// false + reason when the chain cannot run on the pipelined encoder (then the per-wavefront kernels code it)
bool pipe_layout(const zpq_plan& plan, const PipeOptions& opt, PipeLayout& out, std::string& why_not);
bool generate_pipe_source(const zpq_plan& plan, const PipeOptions& opt, std::string& source, std::string& why_not);

// PCOMP translated for the device (device/pcomp_kernel.h); code = PCOMP bytes without the 2 length bytes
bool generate_pcomp_source(const U8* code, size_t len, int ph, int pm, std::string& source, std::string& why_not);

// Cache key of a generated source: SHA-1 over the text (which embeds the codegen
// version) -- the loader extends it with a digest of the kernel template headers.
std::string spec_cache_key(const std::string& source);

}  // namespace zpq
