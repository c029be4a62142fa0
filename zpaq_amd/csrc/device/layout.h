// Device-visible data layout shared by the host engine and the HIP kernels.
//
// One in-flight ZPAQ block owns one contiguous ARENA in HBM:
//
//   [ component tables, each 256-B aligned, in COMP order ]   Predictor::init 1776-1846
//   [ H : U32[2^hh] ][ M : U8[2^hm] ][ R : U32[256] ]         ZPAQL::init 1012-1024
//   [ BlockRun : per-block scalar state + h[]/p[] + per-component run vars ]
//
// The plan (parsed header) is immutable and shared by all blocks coded with the
// same header; it lives in its own small device buffer.
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#else
// hipRTC has no C library headers; its built-in prelude keeps the fixed-width types in a namespace
using __hip_internal::int8_t;
using __hip_internal::int16_t;
using __hip_internal::int32_t;
using __hip_internal::int64_t;
using __hip_internal::uint8_t;
using __hip_internal::uint16_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
#endif

namespace zpq {

enum CompType : uint32_t { C_NONE = 0, C_CONS, C_CM, C_ICM, C_MATCH, C_AVG, C_MIX2, C_MIX, C_ISSE, C_SSE };

// How to initialise one arena segment (Predictor::init patterns).
enum FillKind : uint32_t {
  F_ZERO = 0,
  F_U32 = 1,       // every dword = value            (CM 0x80000000, MIX 65536/m, MIX2 0x80008000)
  F_SSE = 2,       // dword j = sse_row[j&31] | value (SSE, value = start count)
  F_ICM = 3,       // copy icm_init[256]
  F_ISSE = 4,      // copy isse_init[512]
  F_MATCHBUF = 5   // zeros, first byte = 1           (MATCH ht(0)=1, libzpaq.cpp:1801)
};

struct Segment {          // 24 bytes
  uint64_t off;           // byte offset in arena (256-B aligned)
  uint64_t bytes;         // multiple of 16
  uint32_t kind;
  uint32_t value;
};

struct CompDesc {         // 64 bytes
  uint32_t type;
  uint32_t a1, a2, a3, a4, a5;   // COMP argument bytes cp[1..5]
  uint32_t limit;         // CM/SSE: cp[]*4 count limit
  uint32_t mask0;         // t0 index mask (elements): CM 2^s-1, MATCH 2^a1-1, MIX/MIX2 2^s-1, SSE 32*2^s-1
  uint32_t mask1;         // t1 mask (bytes): ICM/ISSE ht_n-1, MATCH 2^a2-1
  uint32_t stride;        // MIX: words from one weight row to the next (>= m; see mix_row_stride), 0 otherwise
  uint64_t t0;            // arena offset of cm / a16
  uint64_t t1;            // arena offset of ht
  uint64_t pad1;
};

// MIX weight rows in HBM: row r of an m-input mixer starts at word r * stride.  The reference packs rows (stride = m); a
// row is then 4 m bytes at a 4-byte-aligned address and straddles 128-byte memory lines (m = 19: 76-byte rows, 1.6 lines
// per row on average).  The coder touches one row per bit and component at a random place of a table far larger than any
// cache, and the MI355X's memory system moves 128-byte lines whatever is asked of them (profiles/r03/gups.hip: 49 G
// random line reads per second, 24 G read-modify-writes), so rows are padded to the next power of two up to 32 words:
// one line per row, and 16-byte-aligned for the lane groups that load a row as 16-byte quads.
static inline constexpr uint32_t mix_row_stride(uint32_t m) {
  return m <= 1 ? 1u : (m <= 2 ? 2u : (m <= 4 ? 4u : (m <= 8 ? 8u : (m <= 16 ? 16u : ((m + 31u) & ~31u)))));
}

struct PlanHeader {       // followed in the same buffer by CompDesc[n], Segment[nseg], prog[prog_len]
  uint32_t n;             // components
  uint32_t hmask;         // 2^hh - 1 (elements)
  uint32_t mmask;         // 2^hm - 1 (bytes)
  uint32_t prog_len;      // HCOMP bytes incl. trailing 0
  uint32_t nseg;
  uint32_t wave_ok;       // 1 if the wave-parallel kernel supports this chain
  uint64_t off_H, off_M, off_R, off_run;   // arena offsets
  uint64_t arena_bytes;   // total, 4 KiB multiple
  uint32_t off_comp;      // byte offsets inside this buffer
  uint32_t off_seg;
  uint32_t off_prog;
  uint32_t total_bytes;
  uint64_t dep_mask;      // wave kernel: lanes whose predict needs earlier p[] (ISSE/AVG/MIX2/MIX/SSE)
  uint64_t mix_mask;      // wave kernel: MIX lanes
};

struct SegRange { uint32_t in_begin, in_end, out_end, status; };

// Per-block job descriptor (device array, one per block in the batch).
struct BlockJob {
  const uint8_t* plan;    // -> PlanHeader
  uint8_t* arena;
  const uint8_t* in;
  uint8_t* out;
  uint32_t in_len;
  uint32_t out_cap;       // encode: capacity; decode: max bytes to decode
  uint32_t res_slot;      // index of this block's BlockResult in the results array
  uint32_t nseg;          // 0 / 1: one segment.  > 1: a block of several segments (model and coder state run on,
                          // Compressor::postProcess / Decoder::decompress, libzpaq.cpp:2889-2891, 2129): `segs` has nseg entries
  SegRange* segs;         // encode: in_begin..in_end = the segment's input bytes (consecutive), out_end <- coded bytes so far
                          // decode: in_begin..in_end = the segment's coded bytes incl. terminator, out_end <- decoded bytes so far
};

struct BlockResult {      // 16 bytes
  uint32_t out_len;
  uint32_t consumed;
  int32_t status;
  uint32_t steps;         // coded bits (diagnostic)
};

// Constant tables as uploaded to the device (one buffer).
struct DeviceTables {
  int16_t stretch[32768];
  uint16_t squash[4096];
  int32_t dt[1024];
  int32_t dt2k[256];
  uint8_t ns[1024];
  uint32_t icm_init[256];
  uint32_t isse_init[512];
  uint32_t sse_row[32];
  uint32_t stretch_cb[2016];   // compact stretch (host/common.hpp Tables)
  int16_t stretch_top[256];
};

// One segment to post-process on the device (device/pcomp_kernel.h): the decoded stream after the PP header goes
// through the block's PCOMP program; H, M, R are the program's zeroed work arrays.
struct PcompJob {
  const uint8_t* in;
  uint8_t* out;
  uint8_t* M;
  uint32_t* H;
  uint32_t* R;          // 256 words
  uint32_t in_len, out_cap;
  uint32_t* result;     // [0] bytes produced (may exceed out_cap: then nothing past the capacity was stored), [1] status
};

// One block to hash on the device (sha1_blocks_kernel): digest 20 * slot in the output array.
struct Sha1Job {
  const uint8_t* p;
  uint32_t len;
  uint32_t slot;
};

// Argument block of the pipelined encoder's kernels (device/pipe_kernel.h), passed by value.
struct PipeArgs {
  const BlockJob* jobs;     // the blocks of one plan, consecutive; 64 consecutive blocks form a group
  BlockResult* res;
  uint32_t nblocks;
  const DeviceTables* tb;
  uint8_t* pipe;            // stream + state buffer, PIPE_GROUP_BYTES per group
  int32_t step;             // a unit of dataflow level L works on chunk step - L
  uint32_t wg0;             // added to blockIdx.x: lets a launch cover a sub-range of a kernel's units
  // Placement / timing trace (kernels built with -DZPQ_TRACE, engine run with ZPAQ_AMD_PIPE_TRACE=<file>): four 64-bit
  // words per workgroup and launch, record index = trace_base + blockIdx.x; null otherwise
  unsigned long long* trace;
  uint32_t trace_base;
  uint32_t arrive_need;           // persistent launch: ctl[3] counts the workgroups that have started; a unit begins only when it says
                                  // this many (every workgroup of this launch and of the run's earlier rounds); 0: no handshake
  // The persistent launch (device/pipe_persist.h): one workgroup set per group of blocks for the whole sequence
  uint32_t* prog;                 // progress counters, PS_NUNIT per group, zero at launch
  const uint32_t* group_chunks;   // per group: chunks of its longest block (at least 1)
  uint32_t* ctl;                  // [0] abort flag (zero at launch; 1: a unit's watchdog fired in mid-sequence, 2: the launch's workgroups
                                  // never became resident together -- nothing was touched), [1] the slot whose watchdog fired, [2] its
                                  // chunk, [3] workgroups that have started
  uint32_t group0, ngroups_here;  // the groups this launch serves
  uint32_t timeout_ticks;         // 100 MHz ticks a poller waits without progress before it raises the abort flag
  uint32_t arrive_ticks;          // 100 MHz ticks without a new arrival before a waiting wavefront gives the launch up (flag 2)
  uint32_t spread;                // 8: the workgroups of a group share an XCD (workgroup b of the first 8 * (ngroups / 8) groups serves group
                                  // b % 8 + 8 * (b / 8 / PS_WPG); the other groups' workgroups follow one after the other); 1: group b / PS_WPG
};

// LDS plan of the specialised kernel (spec_kernel.h), known to the host code
// generator: shared constant tables, then one region per wave (block).
static const int kSpecTablesBytes = 32768 + 2688 + 4096 + 512 + 1024;                       // 41088
static const int kSpecLdsBudget = 163840 - 256;                                             // gfx950: 160 KiB per workgroup
// W blocks (waves) per workgroup share the budget: 4 -> 30624 B each, 8 -> 15312 B each
static inline constexpr int spec_wave_lds_bytes(int waves) { return ((kSpecLdsBudget - kSpecTablesBytes) / waves) & ~15; }

// The lockstep decoder (spec_team_kernel.h) keeps 8 blocks' side tables in one workgroup's LDS, so it packs: the compact
// stretch table of the encoder instead of half the full one (8.4 instead of 32 KiB), ICM side-table entries as 16 + 8 bits
// (cm < 2^23), ISSE weight pairs as 2 x 16 + 8 bits (weights are clamped to +-2^19: libzpaq.cpp:2031-2039).  Of the -m5 chain's
// 18 side tables 14 then fit a block's region (7 in the unpacked plan above).
static const int kTeamTablesBytes = 2016 * 4 + 256 * 2 + 2688 + 4096 + 512 + 1024;         // 16896
static inline constexpr int team_block_lds_bytes() { return ((kSpecLdsBudget - kTeamTablesBytes) / 8) & ~15; }   // 18336
static const int kTeamIcmLds = 256 * 2 + 256, kTeamIsseLds = 256 * 4 + 256;                  // 768, 1280 bytes per table

// The pre-processors behind the suffix sort (device/lz77_kernel.h, device/sa_kernels.hip)
struct LzBlock {            // one per block of the batch
  uint64_t off;             // its first element in the batch's arrays (bytes, suffix array, ranks, decisions)
  uint64_t tok_off;         // its first slot in the token array
  uint32_t n;
  uint32_t tok_cap;
  uint32_t kind;            // 1 / 2: LZ77 with bit-packed / byte-aligned codes; 3: BWT; 0: nothing to do here
  uint32_t min_match, lookahead, bucket, checkbits;     // LZBuffer's parameters (args[2], args[6], 2^args[4] - 1, 17 + args[0])
  uint32_t pad;
};
struct LzTok { uint32_t i, off, len, blit; };            // = host/common.hpp LzToken

// Cap on HCOMP instructions per input byte: the reference has no limit (a
// hostile header can loop forever); a device kernel must not hang.
static const uint32_t kMaxVmSteps = 1u << 20;

}  // namespace zpq
