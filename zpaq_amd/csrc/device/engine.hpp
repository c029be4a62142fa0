#pragma once
#include <vector>

#include "../host/common.hpp"
#include "layout.h"
#include "plan.hpp"

namespace zpq {

struct HostBlock {
  const zpq_plan* plan;
  const U8* prefix;   // optional bytes coded before `in` (the PP header), may be null
  U32 prefix_len;
  const U8* in;
  U32 in_len;
  U8* out;        // host destination (may be null to discard)
  U32 out_cap;    // encode: capacity; decode: max bytes to decode
  // A block of several segments (model and coder state run on from one to the next): nseg > 1, seg_len[s] = the
  // segment's share of `in` (encode: input bytes, the prefix belongs to the first; decode: coded bytes incl. its
  // terminator), seg_out_end[s] <- where the segment's output ends in `out`.  Pipelined encoder / wavefront decoder only.
  U32 nseg = 0;
  const U32* seg_len = nullptr;
  U32* seg_out_end = nullptr;
  U8* sha1_out = nullptr;   // encode: if set, SHA-1 of `in` (without the prefix) is computed on the device into these 20 bytes
};

struct Timing {
  float init_ms = 0;   // init_arena_kernel (Predictor::init)
  float code_ms = 0;   // coding kernel(s)
  uint32_t blocks = 0;
};

void engine_init(int device);
// contiguous share [lo, hi) of n blocks for shard k of `parts` (how host batches are split over devices)
void engine_shard_range(uint64_t n, uint32_t parts, uint32_t k, uint64_t* lo, uint64_t* hi);
int engine_device_count();
int engine_count();                      // engines configured by zpq_init (one per device named; a device may be named twice)
void engine_shutdown();
void engine_set_budget(uint64_t bytes);
void engine_set_kernel(int which);
Timing engine_last_timing();
bool engine_last_persistent();
double engine_last_persist_abort_ms();
void engine_plan_release(zpq_plan* p);
// 4 pipelined encoder (compression only) / 3 specialised / 2 generic wave / 1 generic one-lane; note = origin of the specialised kernel or why not
int engine_plan_kernel_kind(zpq_plan* p, std::string& note, bool decode = false, uint32_t nblocks = 0, uint32_t block_bytes = 0);   // nblocks = 0: a batch that fills the GPU

// Host-buffer batch: copies in, runs (possibly in several residency waves), copies out.  Concurrent callers are
// coalesced into one device batch (see the submission queue in engine.cpp); a caller that will submit shortly
// announces itself with engine_caller_enter() and withdraws the announcement with engine_caller_leave() right
// before it submits (or gives up), so that the batch leader waits for it.
// announced = the caller called engine_caller_enter() before: the announcement is withdrawn once its blocks are queued
void engine_code_host(bool decode, const std::vector<HostBlock>& blocks, std::vector<BlockResult>& results,
                      bool* announced = nullptr);
void engine_caller_enter();
void engine_caller_leave();

// Device-resident batch; plans[0] for every block when one_plan, else plans[b] per block.  Results
// land in the device array d_res in the caller's block order.
void engine_code_device(bool decode, const zpq_plan* const* plans, bool one_plan, const void* d_in,
                        const uint64_t* in_off, const uint32_t* in_len, uint32_t nblocks, void* d_out,
                        const uint64_t* out_off, const uint32_t* out_cap, BlockResult* d_res, void* stream, bool timed);

// Post-processing on the device: every segment's stream (decoded bytes after the PP header) through its block's PCOMP
// program, one lane per segment.  Returns false (note says why) when the program cannot run there: the caller then
// uses the host interpreter.  out[i] receives segment i's data; hint[i] = expected size or 0.
struct PcompSeg { const U8* in; U32 in_len; U64 hint; std::vector<U8>* out; };
bool engine_pcomp(const U8* code, size_t codelen, int ph, int pm, std::vector<PcompSeg>& segs, std::string& note);
void engine_sha1_host(const uint8_t* const* in, const uint32_t* len, uint32_t n, uint8_t* out);
// Suffix arrays of host buffers, all in one device call (device/sa_kernels.hip: prefix doubling over the whole batch).
// false + note when the device declines (then the host sorts): a buffer of 2^24 bytes or more, too many buffers, memory.
bool engine_suffix_arrays(const std::vector<std::pair<const U8*, U32>>& blocks, std::vector<std::vector<U32>>& sa, std::string& note);
// The pre-processors behind the sort for a whole batch on the device (device/lz77_kernel.h): blocks of kind 1 / 2 come back as
// the LZ77 parse's list of matches (host/preproc.cpp lz77_serialize codes it), blocks of kind 3 as the BWT stream
// preprocess_block would make (n + 5 bytes).  false + note when the device declines (then the host does it all).
struct SortJob { const U8* data; U32 n; U32 kind, min_match, lookahead, bucket, checkbits; };
struct SortOut { std::vector<LzToken> toks; std::vector<U8> bwt; };
bool engine_sort_preprocess(const std::vector<SortJob>& jobs, std::vector<SortOut>& out, std::string& note);
// the job of a block whose method has these args (LZBuffer's parameters: libzpaq.cpp:6647-6692)
inline SortJob sort_job(const U8* data, U32 n, const int args[9]) {
  const U32 level = (U32)(args[1] & 3);
  if (level == 3) return SortJob{data, n, 3u, 0u, 0u, 0u, 0u};
  return SortJob{data, n, level, (U32)args[2], (U32)args[6], args[4] >= 0 && args[4] < 31 ? (1u << args[4]) - 1u : 0x7FFFFFFFu, (U32)(17 + args[0])};
}
int engine_selftest(int32_t out[8]);
int engine_jit_threads();      // host threads spec_precompile() uses by default (the host cores the process may use, at most 16)

}  // namespace zpq
