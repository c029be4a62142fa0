// Pipelined ENCODER: the kernel templates a GENERATED translation unit instantiates for ONE block header
// (host/codegen.cpp, generate_pipe_source).  Replaces Encoder::compress + Predictor::predict0/update0 +
// ZPAQL::run (libzpaq.cpp:2419-2447, 1854-2066, 1027-1262) on the compression side only.
//
// Why a second design.  spec_kernel.h maps one ZPAQ block to one wavefront and walks the whole COMP chain once
// per coded bit: ~400 dependent instructions, 23 of 64 lanes busy, 3.1 k cycles per bit, and every lever on that
// design was measured out in round 2 (profiles/r02_ab_matrix.txt).  The decoder has to work that way -- it
// learns each bit from the final probability.  The ENCODER does not: it knows every bit in advance, and
// libzpaq's model is strictly feed-forward --
//   * HCOMP contexts depend on input bytes only;
//   * CM / ICM / MATCH train on (own table, y) only;
//   * ISSE / AVG / MIX2 / MIX / SSE read predictions of EARLIER components and train on (own output, y);
//   * nothing ever reads the final probability except the arithmetic coder.
// So every component is an independent stream processor over the bit sequence, and the chain is a dataflow
// graph.  Here each component of each block gets its own lane:
//
//     wave = 64 consecutive blocks x ONE component (lane = block)      [MIX: a few lanes per block]
//
// Lanes of a wave run identical code on different blocks (no divergence, no cross-lane traffic, all 64 lanes
// busy), a lane's per-bit loop is 15-45 instructions instead of 400, and all table addresses of a byte are
// known before its first bit (no speculation, one candidate instead of two).  Components talk through streams in
// HBM, laid out [position][lane] so that every stream access of a wave is one contiguous 64 / 128 / 256-byte
// transaction:
//     ctx[slot][c][byte][lane]  u32      HCOMP's H[i] for the byte                       (producer: HCOMP unit)
//     bh [slot][r][byte][lane]  2 x u32  the 8 bit histories of the byte's two nibbles   (producer: ROW unit of an ICM/ISSE)
//     p  [slot][i][byte][lane]  8 x i16  stretch-domain predictions of component i       (producer: component i)
// so one 4 / 8 / 16-byte access per lane and input byte moves a unit's whole input or output for that byte.
// Time is cut into chunks of PIPE_C input bytes.  A unit of dataflow level L works on chunk (step - L) during
// launch `step`; one launch = one step, the kernel boundary is the barrier between producers and consumers and
// PIPE_S = maxlevel + 1 ring slots keep a chunk alive until its last consumer has run.  All units of a step are
// independent, so the step's six kernels (they differ only in their static LDS needs) run concurrently.
//
//   unit          level                   per-lane state between chunks
//   HCOMP         0                       VM registers (H, M, R live in the arena as before; H is staged in LDS)
//   ROW(i)        1                       none  (bit-history row probed, used for 4 bits, written back per nibble)
//   CONS/CM/MATCH 1                       MATCH: len/offset/pos/predicted byte
//   ICM map       2                       side table, staged in LDS [entry][lane] during the chunk
//   ISSE map      max(2, level(j)+1)      same (two words per entry)
//   AVG/MIX2/SSE  max(inputs)+1           none
//   MIX           max(inputs)+1           none; MIX_QL lanes per block, DPP reduction inside the lane group
//   CODER         level(n-1)+1            low, high, bytes written
//
// A lane's loop is a serial chain, so what it waits for decides the speed.  Every unit therefore (a) fetches the
// stream inputs of byte k+1 before it works on byte k (nothing it stores can alias them), (b) fetches all table
// words of a byte together before the byte's first bit (legal when the byte's 8 addresses are distinct, a
// compile-time property of the component), and (c) touches the table lines of byte k+1 one byte early so that
// (b) finds them in L2.  The ROW unit loads the candidate rows of both nibbles up front and forwards the row the
// first nibble rewrote; the ISSE/ICM maps read the next bit's LDS entry early and forward the entry just trained.
//
// Two shapes of one unit.  With all bits known, the 8 bits of a byte are independent for every unit whose table index
// contains the bit position and whose training touches only the indexed entry (CM, SSE, MIX2 / MIX with a full c0 mask):
// such a unit also exists with a lane per (block, bit position) -- an 8 x shorter chain per byte, 8 x the wavefronts and
// more memory requests.  What the MI355X says (profiles/r03): a batch that fills the machine is bound by HBM
// TRANSACTIONS -- every random 16-byte row costs a 128-byte line each way, 24 G read-modify-writes per second for the whole
// GPU (profiles/r03/gups.hip) -- and there the lane-per-block units win (fewer requests; only SSE is faster per bit
// position everywhere); a chain with few blocks in the batch is bound by the length of one wavefront's chain, and there
// the bit-position units win (-m5, 64 blocks: 1.6 x).  The generator emits one or the other (PIPE_MODE, PipeOptions in
// host/codegen.hpp) and the engine picks per chain from the number of its blocks in the batch.
//
// Integer arithmetic is the reference's, statement for statement; the parity tests compare the
// coded bytes with the oracle and with the reference.  The same source runs in tests/emu on the host.
#pragma once
#ifndef ZPQ_LANE_VM
#define ZPQ_LANE_VM 1          // one HCOMP machine per lane: its condition flag is per-lane data
#endif
#include "spec_kernel.h"       // CompK, address-space typedefs, clamps, static_for

namespace zpq {

typedef __attribute__((address_space(1))) short g_i16;
typedef __attribute__((address_space(1))) unsigned short g_u16;

// -DZPQ_TRACE: every workgroup of every launch records where it ran (HW_ID: SE / CU / SIMD / wave slot, XCC_ID) and when
// (the 100 MHz reference clock at entry and at exit) -- profiles/pipe_trace.py turns the records into the occupancy of
// every SIMD over a step.  Compiled out otherwise (the kernels' ISA is the same as without this block).
#ifdef ZPQ_TRACE
struct PipeTraceScope {
  unsigned long long* rec;
  __device__ __forceinline__ PipeTraceScope(const PipeArgs& a, int kernel) : rec(nullptr) {
    if (a.trace && threadIdx.x == 0) {
      rec = a.trace + 4ull * ((unsigned long long)a.trace_base + blockIdx.x);
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      rec[0] = ((unsigned long long)(unsigned)kernel << 56) | ((unsigned long long)((unsigned)a.step & 0xFFFFFFu) << 32) |
               (unsigned long long)(blockIdx.x + a.wg0);
      rec[1] = ((unsigned long long)xcc << 32) | hw;
      rec[2] = __builtin_amdgcn_s_memrealtime();
    }
  }
  __device__ __forceinline__ ~PipeTraceScope() {
    if (rec) rec[3] = __builtin_amdgcn_s_memrealtime();
  }
};
#define ZPQ_PIPE_TRACE(a, k) zpq::PipeTraceScope zpq_trace_scope(a, k)
#else
#define ZPQ_PIPE_TRACE(a, k)
#endif

enum PipeKind : int { PK_ROW = 1, PK_CONS, PK_CM, PK_MATCH, PK_AVG, PK_MIX2, PK_SSE, PK_CODER,
                      PK_CM_BITS, PK_MIX2_BITS, PK_SSE_BITS };      // ..._BITS: a lane per (block, bit position), 8 workgroups per group

__device__ __forceinline__ bool pipe_any(bool x) {
#ifdef ZPQ_EMU
  return emu::wave_any(x);
#else
  return __builtin_amdgcn_ballot_w64(x) != 0ull;
#endif
}

// Before a lane fetches a word ANOTHER lane of its wavefront may have stored: wait until the wavefront's stores have
// been acknowledged (the bit-lane units' rare re-fetch path; everything else reads only what the lane itself wrote).
__device__ __forceinline__ void pipe_stores_done() {
#ifndef ZPQ_EMU
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#endif
}

// The same between lanes of ONE wavefront when the re-fetch is simply issued behind the store: the hardware performs a
// wavefront's memory operations on one address in program order (a wavefront-scope fence emits no instruction); this only keeps
// the compiler from moving the load above the store.
__device__ __forceinline__ void pipe_wave_order() {
#ifndef ZPQ_EMU
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#endif
}

// A value the compiler must treat as per-lane data.  When all lanes of a wavefront serve one block the compiler proves
// stream addresses uniform, moves every loaded context into a scalar register with v_readfirstlane right behind the load
// -- and so waits for a fetch that was issued bytes ahead precisely in order not to be waited for.
__device__ __forceinline__ unsigned pipe_opaque(unsigned x) {
#ifndef ZPQ_EMU
  asm volatile("" : "+v"(x));
#endif
  return x;
}

// A value fetched on a rarely taken path inside a pipelined loop: consuming it inside the branch keeps the wait for it
// inside the branch.  (vmcnt counts in order: where the branch joins the main path the compiler would otherwise wait for
// the branch's fetch -- the youngest -- on every iteration, and with it for every fetch issued ahead.)
__device__ __forceinline__ uint4 pipe_settle(uint4 v) {
  v.x = pipe_opaque(v.x); v.y = pipe_opaque(v.y); v.z = pipe_opaque(v.z); v.w = pipe_opaque(v.w);
  return v;
}

typedef __attribute__((address_space(1))) uint2 g_u64v;

// ---- stores the PERSISTENT encoder (pipe_persist.h) hands from one workgroup to another inside a launch -------------
// A stream element is written once and read by other workgroups -- on other CUs, maybe on other XCDs, whose L2s are not
// coherent with the writer's -- a chunk later.  Write-through (sc1) stores put it where every reader finds it; the
// producer then drains its stores (s_waitcnt vmcnt(0)) and publishes its progress counter with an sc1 store, the consumer
// polls the counter relaxed, takes ONE agent-scope acquire and reads with plain loads (MI355X_MICROARCH.md, "Workgroup
// dispatch, XCD placement & inter-workgroup visibility": the drained-sc1 form).  The base is the group's buffer: wave-uniform.
#ifdef ZPQ_EMU
__device__ __forceinline__ void pipe_wt_store16(g_u8* base, unsigned off, const uint4& v) { *(g_u128*)(base + off) = v; }
__device__ __forceinline__ void pipe_wt_store8(g_u8* base, unsigned off, const uint2& v) { *(g_u64v*)(base + off) = v; }
__device__ __forceinline__ void pipe_wt_store4(g_u8* base, unsigned off, unsigned v) { *(g_u32*)(base + off) = v; }
__device__ __forceinline__ void pipe_wt_store2(g_u8* base, unsigned off, unsigned v) { *(g_u16*)(base + off) = (unsigned short)v; }
#else
typedef unsigned pipe_v4u __attribute__((ext_vector_type(4)));
typedef unsigned pipe_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pipe_rsrc(g_u8* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00027000);
}
__device__ __forceinline__ void pipe_wt_store16(g_u8* base, unsigned off, const uint4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(pipe_v4u{v.x, v.y, v.z, v.w}, pipe_rsrc(base), (int)off, 0, 16);     // aux 16 = sc1
}
__device__ __forceinline__ void pipe_wt_store8(g_u8* base, unsigned off, const uint2& v) {
  __builtin_amdgcn_raw_buffer_store_b64(pipe_v2u{v.x, v.y}, pipe_rsrc(base), (int)off, 0, 16);
}
__device__ __forceinline__ void pipe_wt_store4(g_u8* base, unsigned off, unsigned v) {
  __builtin_amdgcn_raw_buffer_store_b32(v, pipe_rsrc(base), (int)off, 0, 16);
}
__device__ __forceinline__ void pipe_wt_store2(g_u8* base, unsigned off, unsigned v) {
  __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, pipe_rsrc(base), (int)off, 0, 16);
}
#endif

template <class Chain, class = void> struct PipePersistOf { static constexpr bool value = false; };
template <class Chain> struct PipePersistOf<Chain, decltype((void)Chain::PIPE_PERSIST)> { static constexpr bool value = Chain::PIPE_PERSIST; };

// One lane's view of its block, its chunk and the group's streams.
template <class Chain>
struct PipeLane {
  static constexpr int N = Chain::N, C = Chain::PIPE_C, S = Chain::PIPE_S;
  static constexpr unsigned G = Chain::PIPE_G;     // blocks per group = active lanes of a wavefront
  unsigned gl;              // lane index inside the group (= position in every stream row)
  bool live;                // block exists
  g_u8* arena;
  const g_u8* in;
  g_u8* out;
  unsigned len, out_cap, rslot;
  unsigned nseg;
  SegRange* segs;
  g_u8* gb;                 // group base in the pipe buffer
  int chunk;                // chunk this unit works on in this step
  unsigned slot;            // chunk % S
  unsigned k0, nb;          // first input byte of the chunk, bytes of this lane in it

  static constexpr bool WT = PipePersistOf<Chain>::value;      // the unit runs inside the persistent launch: streams are stored write-through
  __device__ __forceinline__ void open(const PipeArgs& a, unsigned blk, int level) {
    bind(a, blk / G, blk % G, true);
    at_chunk(a.step - level);
  }
  // the same in two halves (the persistent launch binds a lane to its block once and then walks the chunks):
  // g = the group (wave-uniform there), lane_in_group = the block's position in every stream row
  __device__ __forceinline__ void bind(const PipeArgs& a, unsigned g, unsigned lane_in_group, bool lane_ok) {
    const unsigned blk = g * G + lane_in_group;
    live = lane_ok && blk < a.nblocks;
    const BlockJob job = a.jobs[live ? blk : 0];
    arena = (g_u8*)job.arena;
    in = (const g_u8*)job.in;
    out = (g_u8*)job.out;
    len = live ? job.in_len : 0u;
    out_cap = job.out_cap;
    rslot = job.res_slot;
    nseg = job.nseg;
    segs = job.segs;
    gl = lane_in_group;
    gb = (g_u8*)a.pipe + (unsigned long long)g * Chain::PIPE_GROUP_BYTES;
  }
  __device__ __forceinline__ void at_chunk(int chunk_) {
    chunk = chunk_;
    slot = chunk >= 0 ? (unsigned)chunk % (unsigned)S : 0u;
    k0 = chunk >= 0 ? (unsigned)chunk * (unsigned)C : 0u;
    nb = (chunk >= 0 && len > k0) ? min(len - k0, (unsigned)C) : 0u;
  }
  __device__ __forceinline__ void idle() { live = false; nb = 0; len = 0; }
  // streams (byte offsets fit 32 bits: a group's buffer is far below 4 GiB)
  __device__ __forceinline__ g_u32& ctx(int ci, unsigned k) const {
    return *(g_u32*)(gb + (unsigned)Chain::PIPE_OFF_CTX + ((((slot * Chain::PIPE_NCTX + ci) * C + k) * G + gl) << 2));
  }
  __device__ __forceinline__ g_u64v& bh(int ri, unsigned k) const {
    return *(g_u64v*)(gb + (unsigned)Chain::PIPE_OFF_BH + ((((slot * Chain::PIPE_NROW + ri) * C + k) * G + gl) << 3));
  }
  __device__ __forceinline__ g_u128& p(int i, unsigned k) const {
    return *(g_u128*)(gb + (unsigned)Chain::PIPE_OFF_P + ((((slot * N + i) * C + k) * G + gl) << 4));
  }
  __device__ __forceinline__ g_u32& state(int w) const {
    return *(g_u32*)(gb + (unsigned)Chain::PIPE_OFF_STATE + (((unsigned)w * G + gl) << 2));
  }
  // what a unit hands to OTHER units goes through these (plain stores between launches, write-through inside one)
  __device__ __forceinline__ unsigned off_ctx(int ci, unsigned k) const { return (unsigned)Chain::PIPE_OFF_CTX + ((((slot * Chain::PIPE_NCTX + ci) * C + k) * G + gl) << 2); }
  __device__ __forceinline__ unsigned off_bh(int ri, unsigned k) const { return (unsigned)Chain::PIPE_OFF_BH + ((((slot * Chain::PIPE_NROW + ri) * C + k) * G + gl) << 3); }
  __device__ __forceinline__ unsigned off_p(int i, unsigned k) const { return (unsigned)Chain::PIPE_OFF_P + ((((slot * N + i) * C + k) * G + gl) << 4); }
  __device__ __forceinline__ void put_ctx(int ci, unsigned k, unsigned v) const {
    if constexpr (WT) pipe_wt_store4(gb, off_ctx(ci, k), v); else ctx(ci, k) = v;
  }
  __device__ __forceinline__ void put_bh(int ri, unsigned k, const uint2& v) const {
    if constexpr (WT) pipe_wt_store8(gb, off_bh(ri, k), v); else bh(ri, k) = v;
  }
  __device__ __forceinline__ void put_bh_half(int ri, unsigned k, unsigned half, unsigned v) const {      // one nibble's four bit histories
    if constexpr (WT) pipe_wt_store4(gb, off_bh(ri, k) + 4u * half, v); else *(g_u32*)(gb + off_bh(ri, k) + 4u * half) = v;
  }
  __device__ __forceinline__ void put_p(int i, unsigned k, const uint4& v) const {
    if constexpr (WT) pipe_wt_store16(gb, off_p(i, k), v); else p(i, k) = v;
  }
  __device__ __forceinline__ void put_p64(int i, unsigned k, unsigned half, const uint2& v) const {      // bits 0 .. 3 or 4 .. 7 of the element
    if constexpr (WT) pipe_wt_store8(gb, off_p(i, k) + 8u * half, v); else *(g_u64v*)((g_u8*)&p(i, k) + 8u * half) = v;
  }
  __device__ __forceinline__ void put_p16(int i, unsigned k, unsigned B, int v) const {      // one bit position's half-word
    if constexpr (WT) pipe_wt_store2(gb, off_p(i, k) + 2u * B, (unsigned)v & 0xFFFFu); else *(g_i16*)((g_u8*)&p(i, k) + 2u * B) = (short)v;
  }
  __device__ __forceinline__ void put_state(int w, unsigned v) const {                        // a state word ANOTHER unit reads (HCOMP's status)
    if constexpr (WT) pipe_wt_store4(gb, (unsigned)Chain::PIPE_OFF_STATE + (((unsigned)w * G + gl) << 2), v); else state(w) = v;
  }
  __device__ __forceinline__ g_u32& A32(unsigned off) const { return *(g_u32*)(arena + off); }
  __device__ __forceinline__ g_u8& A8(unsigned off) const { return *(g_u8*)(arena + off); }
  __device__ __forceinline__ g_u128& A128(unsigned off) const { return *(g_u128*)(arena + off); }
  __device__ __forceinline__ unsigned byte_at(unsigned k) const { return in[k0 + k]; }
  __device__ __forceinline__ unsigned next(unsigned k) const { return min(k + 1u, nb - 1u); }   // index to prefetch
};

// 8 predictions of a byte in one 16-byte stream element: bit B in half-word B
__device__ __forceinline__ int pipe_p_get(const uint4& v, int B) {
  const unsigned w = B < 2 ? v.x : (B < 4 ? v.y : (B < 6 ? v.z : v.w));
  return (int)(short)(unsigned short)(w >> (16 * (B & 1)));
}
struct PipeP8 {
  unsigned w[4] = {0, 0, 0, 0};
  __device__ __forceinline__ void set(int B, int v) { w[B >> 1] |= ((unsigned)v & 0xFFFFu) << (16 * (B & 1)); }
  __device__ __forceinline__ uint4 get() const { return make_uint4(w[0], w[1], w[2], w[3]); }
};

// position-in-byte helpers: everything below is a pure function of the input byte (the encoder knows it)
__device__ __forceinline__ int pipe_y(unsigned byte, int B) { return (int)((byte >> (7 - B)) & 1u); }
__device__ __forceinline__ unsigned pipe_c8(unsigned byte, int B) { return (1u << B) | (byte >> (8 - B)); }     // libzpaq.cpp:2055
__device__ __forceinline__ unsigned pipe_hmap4(unsigned byte, int B) {                                          // libzpaq.cpp:2057-2065
  const unsigned hi = byte >> 4, lo = byte & 15u;
  return B < 4 ? ((1u << B) | (hi >> (4 - B))) : (256u + 16u * hi + ((1u << (B - 4)) | (lo >> (8 - B))));
}

// stretch from 8.5 KB of LDS: groups of 8 as (value at the group's start | seven 1-bit increments << 16) for
// x in [16384, 32512), the steep top end direct, the lower half by stretch(x) = -stretch(32767 - x)
struct PipeStretch {
  unsigned cb[2016];
  short top[256];
  __device__ __forceinline__ void load(const DeviceTables* tb, int lane) {
    for (int i = lane; i < 2016; i += (int)blockDim.x) cb[i] = tb->stretch_cb[i];
    for (int i = lane; i < 256; i += (int)blockDim.x) top[i] = tb->stretch_top[i];
  }
  __device__ __forceinline__ int operator()(unsigned x) const {   // x in 0..32767
    const bool lo = x < 16384u;
    const unsigned y = lo ? 32767u - x : x;
    const unsigned e = cb[min((y - 16384u) >> 3, 2015u)];
    const int mid = (int)(short)(unsigned short)e + __builtin_popcount((e >> 16) & ((1u << (y & 7u)) - 1u));
    const int hi = top[y >= 32512u ? y - 32512u : 0u];
    const int v = y >= 32512u ? hi : mid;
    return lo ? -v : v;
  }
};

// squash through the 1344 non-trivial entries held in LDS
struct PipeSquash {
  unsigned short mid[1344];
  __device__ __forceinline__ void load(const DeviceTables* tb, int lane) {
    for (int i = lane; i < 1344; i += (int)blockDim.x) mid[i] = tb->squash[1376 + i];
  }
  __device__ __forceinline__ int operator()(int p) const {   // p in -2048..2047
    const int i = p + 2048 - 1376;
    const int v = mid[min(max(i, 0), 1343)];
    return i < 0 ? 0 : (i > 1343 ? 32767 : v);
  }
};

// stretch through the whole table (64 KiB: a unit of a small chain, LDS to spare -- the compact form costs ~20 instructions a lookup)
struct PipeStretchFull {
  const short* t;
  __device__ __forceinline__ int operator()(unsigned x) const { return t[x]; }   // x in 0..32767
};

// squash through the whole table (a unit with LDS to spare: no range tests around the lookup)
struct PipeSquashFull {
  const unsigned short* t;
  __device__ __forceinline__ int operator()(int p) const { return t[p + 2048]; }   // p in -2048..2047
};

// Predictor::train (libzpaq.h:1151-1157)
__device__ __forceinline__ unsigned pipe_train(unsigned v, int y, unsigned dtv, unsigned limit) {
  const unsigned count = v & 0x3ffu;
  const int err = y * 32767 - (int)(v >> 17);
  return v + ((unsigned)__mul24(err, (int)dtv) & 0xFFFFFC00u) + (count < limit ? 1u : 0u);
}

// =====================================================================================================
// HCOMP unit: HL lanes per workgroup, H staged in LDS as [index][lane] when it fits.
template <class Chain>
struct PipeHLds {
  unsigned* base;
  int lane;
  __device__ __forceinline__ unsigned& operator[](unsigned i) const { return base[i * Chain::HCOMP_LANES + lane]; }
};

// M in LDS as [index][lane] bytes (the persistent launch, M of up to 256 bytes: Chain::HCOMP_M_LDS)
template <class Chain>
struct PipeMLds {
  unsigned char* base;
  int lane;
  struct Ref {
    unsigned char* p;
    __device__ __forceinline__ operator unsigned char() const { return *p; }
    __device__ __forceinline__ Ref& operator=(unsigned char v) { *p = v; return *this; }
  };
  __device__ __forceinline__ Ref operator[](unsigned i) const { return Ref{base + i * Chain::HCOMP_LANES + lane}; }
};
template <class Chain, class = void> struct PipeHcompMLds { static constexpr bool value = false; };
template <class Chain> struct PipeHcompMLds<Chain, decltype((void)Chain::HCOMP_M_LDS)> { static constexpr bool value = Chain::HCOMP_M_LDS; };

// One chunk of the HCOMP unit for the lanes of `L` (lane < HCOMP_LANES; the others are idle).  load_h / store_h: H is
// staged from / written back to the arena around this chunk (the persistent launch keeps it in LDS from chunk to chunk).
template <class Chain>
__device__ __forceinline__ void pipe_hcomp_unit(PipeLane<Chain>& L, unsigned* Hs, int lane, bool load_h, bool store_h) {
  constexpr bool HLDS = Chain::HCOMP_H_LDS;
  constexpr unsigned HW = Chain::HMASK + 1u;
  if (L.live && L.chunk == 0) L.put_state(Chain::HCOMP_STATE + 4, 0u);      // status word, read by the coder at the end
  unsigned st = L.live && L.chunk > 0 ? (unsigned)L.state(Chain::HCOMP_STATE + 4) : 0u;
  if (st) L.nb = 0;
  if (!pipe_any(L.nb > 0)) return;
  unsigned vb = 0, vc = 0, vd = 0, vf = 0;
  if (L.nb && L.chunk > 0) {
    vb = L.state(Chain::HCOMP_STATE + 0); vc = L.state(Chain::HCOMP_STATE + 1);
    vd = L.state(Chain::HCOMP_STATE + 2); vf = L.state(Chain::HCOMP_STATE + 3);
  }
  g_u8* const vm_M = L.arena + Chain::OFF_M;
  g_u32* const vm_R = (g_u32*)(L.arena + Chain::OFF_R);
  g_u32* const Hg = (g_u32*)(L.arena + Chain::OFF_H);
  PipeHLds<Chain> Hl{Hs, lane};
  if constexpr (HLDS) {
    if (L.nb && load_h) for (unsigned i = 0; i < HW; ++i) Hl[i] = Hg[i];
  }
  constexpr bool MLDS = PipeHcompMLds<Chain>::value;
  PipeMLds<Chain> Ml{(unsigned char*)Hs + (HLDS ? HW * (unsigned)Chain::HCOMP_LANES * 4u : 16u), lane};
  if constexpr (MLDS) {
    if (L.nb && load_h) for (unsigned i = 0; i <= Chain::MMASK; ++i) Ml[i] = (unsigned char)vm_M[i];
  }
  // the input four bytes at a time, a word ahead (a byte asked for one byte ahead is waited for behind the stream stores of the
  // byte before: they are written through, and vmcnt counts in order); a block's input may be read up to the next multiple of 64
  typedef unsigned __attribute__((aligned(1))) u32u;
  typedef const __attribute__((address_space(1))) u32u g_cu32u;
  const unsigned lastw = L.nb ? (L.nb - 1u) & ~3u : 0u;
  unsigned cur = L.nb ? *(g_cu32u*)(L.in + L.k0) : 0u;
  unsigned nxt = L.nb ? *(g_cu32u*)(L.in + L.k0 + min(4u, lastw)) : 0u;
  for (unsigned k = 0; k < L.nb; ++k) {
    if ((k & 3u) == 0u && k) { cur = nxt; nxt = *(g_cu32u*)(L.in + L.k0 + min(k + 4u, lastw)); }
    const unsigned ch = (cur >> (8u * (k & 3u))) & 255u;
    // contexts of byte k = H as left by the bytes before it (Predictor::update0, libzpaq.cpp:2049-2054): read before the
    // program runs, stored BEHIND it -- the stream stores are written through and vmcnt counts in order, so a load of the
    // program (its M array lives in the arena) issued behind them would wait for their acknowledgements (round 6, profiles/r06
    // call 16: mid.cfg's HCOMP unit set the pace of its launch at 2.1 us per byte)
    unsigned hv[Chain::PIPE_NCTX];
    static_for<0, Chain::N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      if constexpr (Chain::P_CTX[i] >= 0) {
        if constexpr (HLDS) hv[Chain::P_CTX[i]] = Hl[(unsigned)i & Chain::HMASK]; else hv[Chain::P_CTX[i]] = Hg[(unsigned)i & Chain::HMASK];
      }
    });
    int e;
    if constexpr (HLDS && MLDS) e = Chain::hcomp(ch, vb, vc, vd, vf, Ml, Hl, vm_R);
    else if constexpr (HLDS) e = Chain::hcomp(ch, vb, vc, vd, vf, vm_M, Hl, vm_R);
    else if constexpr (MLDS) e = Chain::hcomp(ch, vb, vc, vd, vf, Ml, Hg, vm_R);
    else e = Chain::hcomp(ch, vb, vc, vd, vf, vm_M, Hg, vm_R);
    static_for<0, Chain::N>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      if constexpr (Chain::P_CTX[i] >= 0) L.put_ctx(Chain::P_CTX[i], k, hv[Chain::P_CTX[i]]);
    });
    if (e) { st = (unsigned)e; break; }
  }
  if (L.nb) {
    if constexpr (HLDS) { if (store_h) for (unsigned i = 0; i < HW; ++i) Hg[i] = Hl[i]; }
    L.state(Chain::HCOMP_STATE + 0) = vb; L.state(Chain::HCOMP_STATE + 1) = vc;
    L.state(Chain::HCOMP_STATE + 2) = vd; L.state(Chain::HCOMP_STATE + 3) = vf;
    L.put_state(Chain::HCOMP_STATE + 4, st);
  }
}

template <class Chain>
__device__ __forceinline__ void pipe_hcomp_body(const PipeArgs& a) {
  constexpr int HL = Chain::HCOMP_LANES;
  constexpr bool HLDS = Chain::HCOMP_H_LDS;
  constexpr unsigned HW = Chain::HMASK + 1u;
  __shared__ unsigned Hs[HLDS ? HW * HL : 1];
  const int lane = threadIdx.x & 63;
  const bool mine = lane < HL;
  PipeLane<Chain> L;
  L.open(a, (blockIdx.x + a.wg0) * HL + (mine ? lane : 0), 0);
  if (!mine) { L.live = false; L.nb = 0; }
  pipe_hcomp_unit<Chain>(L, Hs, lane, true, true);
}

// =====================================================================================================
// ROW unit of ICM / ISSE component I: Predictor::find (libzpaq.cpp:2072-2088) once per nibble, the row's
// bit histories in registers for the nibble's 4 bits, next-state by table, row written back.
struct PipeRow { unsigned off, w0, w1, w2, w3; };

__device__ __forceinline__ PipeRow pipe_find(const uint4& r0, const uint4& r1, const uint4& r2, unsigned chk, unsigned h0) {
  const bool m0 = (r0.x & 255u) == chk, m1 = (r1.x & 255u) == chk, m2 = (r2.x & 255u) == chk;
  const unsigned p0 = (r0.x >> 8) & 255u, p1 = (r1.x >> 8) & 255u, p2 = (r2.x >> 8) & 255u;
  const bool hit = m0 || m1 || m2;
  // the row taken: the first that matches, else the victim (lowest priority, the earlier of equals).  As two flags, not as an
  // index: the compiler turned "pick == 0 ? .. : pick == 1 ? .." into a switch and the switch into ladders of exec-mask
  // branches -- ~100 scalar and branch instructions per find, two finds per byte in every ROW unit (round 6, profiles/unit_isa.py)
  const bool v0 = p0 <= p1 && p0 <= p2, v1 = !v0 && p1 < p2;
  const bool t0 = m0 || (!hit && v0);
  const bool t1 = !m0 && (m1 || (!hit && v1));
  PipeRow r;
  r.off = h0 ^ (t0 ? 0u : (t1 ? 16u : 32u));
  const unsigned x = t0 ? r0.x : (t1 ? r1.x : r2.x), y = t0 ? r0.y : (t1 ? r1.y : r2.y);
  const unsigned z = t0 ? r0.z : (t1 ? r1.z : r2.z), w = t0 ? r0.w : (t1 ? r1.w : r2.w);
  r.w0 = hit ? x : chk;
  r.w1 = hit ? y : 0u;
  r.w2 = hit ? z : 0u;
  r.w3 = hit ? w : 0u;
  return r;
}

// the nibble's 4 bits: slots 1, 2..3, 4..7, 8..15 (hmap4 & 15) = bytes 1..3 of w0, then w1, then w2 / w3;
// returns the 4 bit histories the predictor sees, lowest byte first
template <class NS>
__device__ __forceinline__ unsigned pipe_row_bits(PipeRow& r, unsigned bits, const NS& ns) {
  unsigned outw;
  {
    const unsigned s = (r.w0 >> 8) & 255u;
    outw = s;
    r.w0 = (r.w0 & 0xFFFF00FFu) | ((unsigned)ns[s * 4u + ((bits >> 3) & 1u)] << 8);
  }
  {
    const unsigned sh = 16u + 8u * ((bits >> 3) & 1u);
    const unsigned s = (r.w0 >> sh) & 255u;
    outw |= s << 8;
    r.w0 = (r.w0 & ~(255u << sh)) | ((unsigned)ns[s * 4u + ((bits >> 2) & 1u)] << sh);
  }
  {
    const unsigned sh = 8u * ((bits >> 2) & 3u);
    const unsigned s = (r.w1 >> sh) & 255u;
    outw |= s << 16;
    r.w1 = (r.w1 & ~(255u << sh)) | ((unsigned)ns[s * 4u + ((bits >> 1) & 1u)] << sh);
  }
  {
    const unsigned idx = (bits >> 1) & 7u, sh = 8u * (idx & 3u);
    const unsigned wsel = (idx & 4u) ? r.w3 : r.w2;
    const unsigned s = (wsel >> sh) & 255u;
    outw |= s << 24;
    const unsigned nw = (wsel & ~(255u << sh)) | ((unsigned)ns[s * 4u + (bits & 1u)] << sh);
    r.w2 = (idx & 4u) ? r.w2 : nw;
    r.w3 = (idx & 4u) ? nw : r.w3;
  }
  return outw;
}

template <class Chain, int I, class NS>
__device__ __forceinline__ void pipe_row(PipeLane<Chain>& L, const NS& ns) {
  constexpr CompK c = Chain::comp[I];
  constexpr unsigned sizebits = c.a1 + 2, rmask = c.mask1, ht = (unsigned)c.t1;
  constexpr int ci = Chain::P_CTX[I], ri = Chain::P_ROW[I];
  if (!L.nb) return;
  // candidate rows of a byte's two nibbles (c8 = 1, c8 = 16 + high nibble): three 16-byte rows inside one 64-byte line each
  auto lines = [&](unsigned hh, unsigned bytev, unsigned& ha, unsigned& hb) __attribute__((always_inline)) {
    ha = ((hh + 16u) * 16u) & (rmask - 15u);
    hb = ((hh + 16u * (16u + (bytev >> 4))) * 16u) & (rmask - 15u);
  };
  unsigned h = L.ctx(ci, 0), byte = L.byte_at(0);
  unsigned k1 = L.next(0);
  unsigned h1 = L.ctx(ci, k1), byte1 = L.byte_at(k1);
  unsigned ha, hb;
  lines(h, byte, ha, hb);
  uint4 a0 = L.A128(ht + ha), a1 = L.A128(ht + (ha ^ 16u)), a2 = L.A128(ht + (ha ^ 32u));
  uint4 b0 = L.A128(ht + hb), b1 = L.A128(ht + (hb ^ 16u)), b2 = L.A128(ht + (hb ^ 32u));
  for (unsigned k = 0; k < L.nb; ++k) {
    const unsigned k2 = min(k + 2u, L.nb - 1u);
    const unsigned h2 = L.ctx(ci, k2), byte2 = L.byte_at(k2);          // streams run two bytes ahead ...
    // ... the table one byte ahead: the next byte's six candidates are fetched before this byte's two rows are
    // written back, unless they share a 64-byte line with them (then they are fetched after the stores)
    unsigned han, hbn;
    lines(h1, byte1, han, hbn);
    const unsigned la = ha & ~63u, lb = hb & ~63u, lan = han & ~63u, lbn = hbn & ~63u;
    const bool clash = lan == la || lan == lb || lbn == la || lbn == lb;
    uint4 na0 = a0, na1 = a1, na2 = a2, nb0 = b0, nb1 = b1, nb2 = b2;
    if (!clash) {
      na0 = L.A128(ht + han); na1 = L.A128(ht + (han ^ 16u)); na2 = L.A128(ht + (han ^ 32u));
      nb0 = L.A128(ht + hbn); nb1 = L.A128(ht + (hbn ^ 16u)); nb2 = L.A128(ht + (hbn ^ 32u));
    }
    const unsigned cxa = h + 16u, cxb = h + 16u * (16u + (byte >> 4));
    PipeRow ra = pipe_find(a0, a1, a2, (cxa >> sizebits) & 255u, ha);
    uint2 o;
    o.x = pipe_row_bits(ra, byte >> 4, ns);
    const uint4 na = make_uint4(ra.w0, ra.w1, ra.w2, ra.w3);
    L.A128(ht + ra.off) = na;
    // the second nibble's candidates were fetched before that store: forward the row it rewrote
    if (ra.off == hb) b0 = na;
    if (ra.off == (hb ^ 16u)) b1 = na;
    if (ra.off == (hb ^ 32u)) b2 = na;
    PipeRow rb = pipe_find(b0, b1, b2, (cxb >> sizebits) & 255u, hb);
    o.y = pipe_row_bits(rb, byte & 15u, ns);
    L.A128(ht + rb.off) = make_uint4(rb.w0, rb.w1, rb.w2, rb.w3);
    L.put_bh(ri, k, o);
    if (clash) {
      na0 = L.A128(ht + han); na1 = L.A128(ht + (han ^ 16u)); na2 = L.A128(ht + (han ^ 32u));
      nb0 = L.A128(ht + hbn); nb1 = L.A128(ht + (hbn ^ 16u)); nb2 = L.A128(ht + (hbn ^ 32u));
    }
    a0 = na0; a1 = na1; a2 = na2; b0 = nb0; b1 = nb1; b2 = nb2;
    ha = han; hb = hbn;
    h = h1; byte = byte1; h1 = h2; byte1 = byte2;
  }
}

// The same unit with a lane per NIBBLE (round 6; small chains in the latency shape, whose wavefronts are half empty: lane b codes
// the first nibble of block b's bytes, lane 32 + b the second).  The encoder knows both contexts of a byte before its first bit,
// and the two finds of a byte touch different 64-byte lines whenever the table has 8 KiB or more (their addresses differ by
// 3 840 .. 7 680 bytes; the caller checks), so within a byte the halves do not meet; the next byte's candidates are fetched before
// this byte's row is stored unless their line is the one either half is about to store (then behind the stores, as above).
// Half the instruction stream per byte.
template <class Chain, int I, class NS>
__device__ __forceinline__ void pipe_row_halves(PipeLane<Chain>& L, const NS& ns, unsigned half) {
  constexpr CompK c = Chain::comp[I];
  constexpr unsigned sizebits = c.a1 + 2, rmask = c.mask1, ht = (unsigned)c.t1;
  constexpr int ci = Chain::P_CTX[I], ri = Chain::P_ROW[I];
  static_assert(rmask + 1u >= 8192u, "the two nibbles of a byte must not share a line");
  if (!L.nb) return;
  // the lines of a byte's two finds: this lane's and the other half's
  auto lines = [&](unsigned hh, unsigned bytev, unsigned& mine, unsigned& other) __attribute__((always_inline)) {
    const unsigned ha = ((hh + 16u) * 16u) & (rmask - 15u);
    const unsigned hb = ((hh + 16u * (16u + (bytev >> 4))) * 16u) & (rmask - 15u);
    mine = half ? hb : ha;
    other = half ? ha : hb;
  };
  auto same_line = [](unsigned x, unsigned y) __attribute__((always_inline)) { return ((x ^ y) & ~63u) == 0u; };
  // The table runs TWO bytes ahead, the streams four: a lone wavefront on an empty machine waits for every row it asks for, and
  // one byte of lead left a byte's work to cover a trip to HBM.  Everything a byte needs lives in one of four SLOTS (byte k in
  // slot k mod 4), the loop is unrolled four times and every slot is a fixed set of registers: a value handed from register to
  // register would have to be waited for on the spot, which is the lead gone.  The rows of byte k + 2 are asked for when byte k
  // is done; if their line is one that byte k or k + 1 stores into (either half's), they are asked for AGAIN behind byte k + 1's
  // store.  No lane leaves the loop early (the compiler's vmcnt bookkeeping gives up at a divergent branch): a lane whose block
  // ends inside the chunk goes on over its last byte's elements -- its tables and stream positions are dead by then, and a
  // chunk's length is a multiple of 4, so a block that goes on never overruns.
  static_assert(Chain::PIPE_C % 4 == 0, "ring of four slots");
  const unsigned last = L.nb - 1u;
  unsigned hq[4], byq[4], hmq[4], hoq[4];
  uint4 r0[4], r1[4], r2[4];
  bool late[4];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const unsigned kd = min((unsigned)sl, last);
    hq[sl] = L.ctx(ci, kd); byq[sl] = L.byte_at(kd);
    late[sl] = false;
  }
  lines(hq[0], byq[0], hmq[0], hoq[0]);
  lines(hq[1], byq[1], hmq[1], hoq[1]);
  hmq[2] = hmq[3] = hmq[1]; hoq[2] = hoq[3] = hoq[1];
  r0[0] = L.A128(ht + hmq[0]); r1[0] = L.A128(ht + (hmq[0] ^ 16u)); r2[0] = L.A128(ht + (hmq[0] ^ 32u));
  late[1] = same_line(hmq[1], hmq[0]) || same_line(hmq[1], hoq[0]);         // byte 1's rows: again behind byte 0's store?
  r0[1] = L.A128(ht + hmq[1]); r1[1] = L.A128(ht + (hmq[1] ^ 16u)); r2[1] = L.A128(ht + (hmq[1] ^ 32u));
  r0[2] = r0[3] = r0[0]; r1[2] = r1[3] = r1[0]; r2[2] = r2[3] = r2[0];
  for (unsigned kb = 0; pipe_any(kb < L.nb); kb += 4u) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int s1 = (sl + 1) % 4, s2 = (sl + 2) % 4;
      const unsigned k = kb + (unsigned)sl;
      // byte k: find, the nibble's four bits, the row back, the bit histories out
      const unsigned h = hq[sl], byte = byq[sl], hm = hmq[sl];
      const unsigned cx = half ? h + 16u * (16u + (byte >> 4)) : h + 16u;
      PipeRow r = pipe_find(r0[sl], r1[sl], r2[sl], (cx >> sizebits) & 255u, hm);
      const unsigned o = pipe_row_bits(r, half ? byte & 15u : byte >> 4, ns);
      L.A128(ht + r.off) = make_uint4(r.w0, r.w1, r.w2, r.w3);
      L.put_bh_half(ri, k, half, o);
      // byte k + 1's rows again, if their line was one this byte or the one before stored into -- the other half's store as well:
      // another lane's, but the same wavefront's, and a wavefront's memory operations on one address are performed in program
      // order (the re-fetch is issued behind the store; waiting for the store's acknowledgement first -- on text some lane of
      // the 64 is late at nearly every byte -- put a memory round trip into every byte: profiles/r06 call 23)
      if (pipe_any(late[s1])) {
        pipe_wave_order();
        if (late[s1]) { r0[s1] = L.A128(ht + hmq[s1]); r1[s1] = L.A128(ht + (hmq[s1] ^ 16u)); r2[s1] = L.A128(ht + (hmq[s1] ^ 32u)); }
      }
      // byte k + 2: its lines from the stream elements asked for two bytes ago, its rows now
      lines(hq[s2], byq[s2], hmq[s2], hoq[s2]);
      late[s2] = same_line(hmq[s2], hm) || same_line(hmq[s2], hoq[sl]) || same_line(hmq[s2], hmq[s1]) || same_line(hmq[s2], hoq[s1]);
      r0[s2] = L.A128(ht + hmq[s2]); r1[s2] = L.A128(ht + (hmq[s2] ^ 16u)); r2[s2] = L.A128(ht + (hmq[s2] ^ 32u));
      // byte k + 4's stream elements into the slot byte k leaves
      const unsigned k4 = min(k + 4u, last);
      hq[sl] = L.ctx(ci, k4); byq[sl] = L.byte_at(k4);
    }
  }
}

// The lane-per-block unit with the table TWO bytes ahead and the context stream four (round 6; pipe_row_halves says why and how:
// byte k in slot k mod 4, the loop unrolled four times, every slot a fixed set of registers, no lane leaving the loop early).
// The six candidate rows of byte k + 2 are asked for when byte k is done; if one of their lines is a line byte k or k + 1 stores
// into, they are asked for again behind byte k + 1's stores.
template <class Chain, int I, class NS>
__device__ __forceinline__ void pipe_row_ring(PipeLane<Chain>& L, const NS& ns) {
  constexpr CompK c = Chain::comp[I];
  constexpr unsigned sizebits = c.a1 + 2, rmask = c.mask1, ht = (unsigned)c.t1;
  constexpr int ci = Chain::P_CTX[I], ri = Chain::P_ROW[I];
  static_assert(Chain::PIPE_C % 4 == 0, "ring of four slots");
  if (!L.nb) return;
  auto lines = [&](unsigned hh, unsigned bytev, unsigned& ha, unsigned& hb) __attribute__((always_inline)) {
    ha = ((hh + 16u) * 16u) & (rmask - 15u);
    hb = ((hh + 16u * (16u + (bytev >> 4))) * 16u) & (rmask - 15u);
  };
  auto same_line = [](unsigned x, unsigned y) __attribute__((always_inline)) { return ((x ^ y) & ~63u) == 0u; };
  const unsigned last = L.nb - 1u;
  unsigned hq[4], byq[4], haq[4], hbq[4];
  uint4 a0[4], a1[4], a2[4], b0[4], b1[4], b2[4];
  bool late[4];
  auto fetch = [&](int sl) __attribute__((always_inline)) {
    a0[sl] = L.A128(ht + haq[sl]); a1[sl] = L.A128(ht + (haq[sl] ^ 16u)); a2[sl] = L.A128(ht + (haq[sl] ^ 32u));
    b0[sl] = L.A128(ht + hbq[sl]); b1[sl] = L.A128(ht + (hbq[sl] ^ 16u)); b2[sl] = L.A128(ht + (hbq[sl] ^ 32u));
  };
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const unsigned kd = min((unsigned)sl, last);
    hq[sl] = L.ctx(ci, kd); byq[sl] = L.byte_at(kd);
    late[sl] = false;
  }
  lines(hq[0], byq[0], haq[0], hbq[0]);
  lines(hq[1], byq[1], haq[1], hbq[1]);
  haq[2] = haq[3] = haq[1]; hbq[2] = hbq[3] = hbq[1];
  fetch(0);
  late[1] = same_line(haq[1], haq[0]) || same_line(haq[1], hbq[0]) || same_line(hbq[1], haq[0]) || same_line(hbq[1], hbq[0]);
  fetch(1);
  a0[2] = a0[3] = a0[0]; a1[2] = a1[3] = a1[0]; a2[2] = a2[3] = a2[0];
  b0[2] = b0[3] = b0[0]; b1[2] = b1[3] = b1[0]; b2[2] = b2[3] = b2[0];
  for (unsigned kb = 0; pipe_any(kb < L.nb); kb += 4u) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int s1 = (sl + 1) % 4, s2 = (sl + 2) % 4;
      const unsigned k = kb + (unsigned)sl;
      const unsigned h = hq[sl], byte = byq[sl], ha = haq[sl], hb = hbq[sl];
      const unsigned cxa = h + 16u, cxb = h + 16u * (16u + (byte >> 4));
      PipeRow ra = pipe_find(a0[sl], a1[sl], a2[sl], (cxa >> sizebits) & 255u, ha);
      uint2 o;
      o.x = pipe_row_bits(ra, byte >> 4, ns);
      const uint4 na = make_uint4(ra.w0, ra.w1, ra.w2, ra.w3);
      L.A128(ht + ra.off) = na;
      // the second nibble's candidates were fetched before that store: forward the row it rewrote
      uint4 c0 = b0[sl], c1 = b1[sl], c2 = b2[sl];
      if (ra.off == hb) c0 = na;
      if (ra.off == (hb ^ 16u)) c1 = na;
      if (ra.off == (hb ^ 32u)) c2 = na;
      PipeRow rb = pipe_find(c0, c1, c2, (cxb >> sizebits) & 255u, hb);
      o.y = pipe_row_bits(rb, byte & 15u, ns);
      L.A128(ht + rb.off) = make_uint4(rb.w0, rb.w1, rb.w2, rb.w3);
      L.put_bh(ri, k, o);
      // byte k + 1's rows again, if one of their lines was a line this byte or the one before stored into
      if (pipe_any(late[s1])) {
        pipe_wave_order();
        if (late[s1]) fetch(s1);
      }
      // byte k + 2: its lines from the stream elements asked for two bytes ago, its rows now
      lines(hq[s2], byq[s2], haq[s2], hbq[s2]);
      late[s2] = same_line(haq[s2], ha) || same_line(haq[s2], hb) || same_line(haq[s2], haq[s1]) || same_line(haq[s2], hbq[s1]) ||
                 same_line(hbq[s2], ha) || same_line(hbq[s2], hb) || same_line(hbq[s2], haq[s1]) || same_line(hbq[s2], hbq[s1]);
      fetch(s2);
      // byte k + 4's stream elements into the slot byte k leaves
      const unsigned k4 = min(k + 4u, last);
      hq[sl] = L.ctx(ci, k4); byq[sl] = L.byte_at(k4);
    }
  }
}

// CONS: a constant stream, so that consumers need no special case
template <class Chain, int I>
__device__ __forceinline__ void pipe_cons(PipeLane<Chain>& L) {
  constexpr unsigned v = (unsigned)(((int)Chain::comp[I].a1 - 128) * 4) & 0xFFFFu;
  for (unsigned k = 0; k < L.nb; ++k) L.put_p(I, k, make_uint4(v | v << 16, v | v << 16, v | v << 16, v | v << 16));
}

// CM (Predictor::predict0/update0 case CM, libzpaq.cpp:1869-1873, 1969-1971)
template <class Chain, int I, class DT>
__device__ __forceinline__ void pipe_cm(PipeLane<Chain>& L, const PipeStretch& stretch, const DT& dt) {
  constexpr CompK c = Chain::comp[I];
  constexpr int ci = Chain::P_CTX[I];
  constexpr bool batch = c.mask0 >= 511u;      // the 8 words of a byte are distinct: fetch them together, a byte ahead
  if (!L.nb) return;
  auto addr = [&](unsigned hh, unsigned bytev, int B) __attribute__((always_inline)) -> unsigned {
    return (unsigned)c.t0 + 4u * ((hh ^ pipe_hmap4(bytev, B)) & c.mask0);
  };
  unsigned h = L.ctx(ci, 0), byte = L.byte_at(0);
  const unsigned k1 = L.next(0);
  unsigned h1 = L.ctx(ci, k1), byte1 = L.byte_at(k1);
  unsigned v[8];
  if constexpr (batch) {
#pragma unroll
    for (int B = 0; B < 8; ++B) v[B] = L.A32(addr(h, byte, B));
  }
  for (unsigned k = 0; k < L.nb; ++k) {
    const unsigned k2 = min(k + 2u, L.nb - 1u);
    const unsigned h2 = L.ctx(ci, k2), byte2 = L.byte_at(k2);
    // Words of the next byte: same context -> only equal bit positions can coincide (the hmap4 ranges of different
    // positions are disjoint), forwarded below; contexts differing in the bits hmap4 covers -> fetched after the stores
    const bool same = h1 == h;
    const bool late = !same && (((h1 ^ h) & c.mask0) < 512u);
    unsigned vn[8], nv[8];
    if constexpr (batch) {
      if (!late) {
#pragma unroll
        for (int B = 0; B < 8; ++B) vn[B] = L.A32(addr(h1, byte1, B));
      }
    }
    PipeP8 out;
#pragma unroll
    for (int B = 0; B < 8; ++B) {
      const unsigned off = addr(h, byte, B);
      if constexpr (!batch) v[B] = L.A32(off);
      out.set(B, stretch(v[B] >> 17));
      nv[B] = pipe_train(v[B], pipe_y(byte, B), (unsigned)dt[v[B] & 0x3ffu], c.limit);
      L.A32(off) = nv[B];
    }
    L.put_p(I, k, out.get());
    if constexpr (batch) {
      if (late) {
#pragma unroll
        for (int B = 0; B < 8; ++B) vn[B] = L.A32(addr(h1, byte1, B));
      }
#pragma unroll
      for (int B = 0; B < 8; ++B) v[B] = (same && addr(h1, byte1, B) == addr(h, byte, B)) ? nv[B] : vn[B];
    }
    h = h1; byte = byte1; h1 = h2; byte1 = byte2;
  }
}

// CM whose table fits the LDS (persistent launch: the table of the group's blocks lives there as [entry][lane] for the whole
// sequence; staged from the arena at the first chunk).  A wavefront's LDS operations execute in order, so the next byte's
// reads follow this byte's writes without any of the forwarding the global version needs.
template <class Chain, int I, class DT>
__device__ __forceinline__ void pipe_cm_lds(PipeLane<Chain>& L, unsigned* tab, const PipeStretch& stretch, const DT& dt, int lane, bool load_tab) {
  constexpr unsigned G = Chain::PIPE_G;
  constexpr CompK c = Chain::comp[I];
  constexpr int ci = Chain::P_CTX[I];
  if (!L.nb) return;
  if (load_tab)
    for (unsigned e = 0; e <= c.mask0; e += 4) {
      const uint4 q = L.A128((unsigned)c.t0 + 4u * e);
      tab[e * G + lane] = q.x; tab[(e + 1) * G + lane] = q.y; tab[(e + 2) * G + lane] = q.z; tab[(e + 3) * G + lane] = q.w;
    }
  unsigned h = L.ctx(ci, 0), byte = L.byte_at(0);
  const unsigned k1 = L.next(0);
  unsigned h1 = L.ctx(ci, k1), byte1 = L.byte_at(k1);
  for (unsigned k = 0; k < L.nb; ++k) {
    const unsigned k2 = min(k + 2u, L.nb - 1u);
    const unsigned h2 = L.ctx(ci, k2), byte2 = L.byte_at(k2);
    PipeP8 out;
    unsigned v[8], ix[8];
#pragma unroll
    for (int B = 0; B < 8; ++B) ix[B] = ((h ^ pipe_hmap4(byte, B)) & c.mask0) * G + (unsigned)lane;
    if constexpr (c.mask0 >= 511u) {
      // the 8 words of a byte are distinct: all reads, then all table lookups, then all writes -- three LDS round trips per
      // byte instead of three per bit (the compiler cannot move a lookup above a store into the same LDS array by itself)
      unsigned d[8];
#pragma unroll
      for (int B = 0; B < 8; ++B) v[B] = tab[ix[B]];
#pragma unroll
      for (int B = 0; B < 8; ++B) { d[B] = (unsigned)dt[v[B] & 0x3ffu]; out.set(B, stretch(v[B] >> 17)); }
#pragma unroll
      for (int B = 0; B < 8; ++B) tab[ix[B]] = pipe_train(v[B], pipe_y(byte, B), d[B], c.limit);
    } else {
#pragma unroll
      for (int B = 0; B < 8; ++B) {        // (a table this small: two positions of a byte may share a word)
        v[B] = tab[ix[B]];
        out.set(B, stretch(v[B] >> 17));
        tab[ix[B]] = pipe_train(v[B], pipe_y(byte, B), (unsigned)dt[v[B] & 0x3ffu], c.limit);
      }
    }
    L.put_p(I, k, out.get());
    h = h1; byte = byte1; h1 = h2; byte1 = byte2;
  }
}

// MIX2 whose weights fit the LDS (persistent launch; see pipe_cm_lds)
template <class Chain, int I, class SQ>
__device__ __forceinline__ void pipe_mix2_lds(PipeLane<Chain>& L, unsigned* tab, const SQ& squash, int lane, bool load_tab) {
  constexpr unsigned G = Chain::PIPE_G;
  constexpr CompK c = Chain::comp[I];
  constexpr int ci = Chain::P_CTX[I];
  if (!L.nb) return;
  if (load_tab)
    for (unsigned e = 0; e <= c.mask0; e += 4) {
      const uint4 q = L.A128((unsigned)c.t0 + 4u * e);
      tab[e * G + lane] = q.x; tab[(e + 1) * G + lane] = q.y; tab[(e + 2) * G + lane] = q.z; tab[(e + 3) * G + lane] = q.w;
    }
  unsigned h = L.ctx(ci, 0), byte = L.byte_at(0);
  const unsigned k1 = L.next(0);
  unsigned h1 = L.ctx(ci, k1), byte1 = L.byte_at(k1);
  uint4 vj = L.p((int)c.a2, 0), vk = L.p((int)c.a3, 0);
  uint4 vj1 = L.p((int)c.a2, k1), vk1 = L.p((int)c.a3, k1);
  for (unsigned k = 0; k < L.nb; ++k) {
    const unsigned k2 = min(k + 2u, L.nb - 1u);
    const unsigned h2 = L.ctx(ci, k2), byte2 = L.byte_at(k2);
    const uint4 vj2 = L.p((int)c.a2, k2), vk2 = L.p((int)c.a3, k2);
    PipeP8 out;
    if constexpr (c.a5 == 255u && c.mask0 >= 255u) {
      // the 8 weights of a byte are distinct: reads, squash lookups and writes in three rounds (see pipe_cm_lds)
      unsigned ix[8];
      int w[8], pr[8], sq[8];
#pragma unroll
      for (int B = 0; B < 8; ++B) { ix[B] = ((h + pipe_c8(byte, B)) & c.mask0) * G + (unsigned)lane; w[B] = (int)tab[ix[B]]; }
#pragma unroll
      for (int B = 0; B < 8; ++B) {
        pr[B] = (__mul24(w[B], pipe_p_get(vj, B)) + __mul24(65536 - w[B], pipe_p_get(vk, B))) >> 16;
        out.set(B, pr[B]);
        sq[B] = squash(sp_clamp2k(pr[B]));
      }
#pragma unroll
      for (int B = 0; B < 8; ++B) {
        const int err = __mul24(pipe_y(byte, B) * 32767 - sq[B], (int)c.a4) >> 5;
        tab[ix[B]] = (unsigned)min(max(w[B] + ((__mul24(err, pipe_p_get(vj, B) - pipe_p_get(vk, B)) + (1 << 12)) >> 13), 0), 65535);
      }
    } else {
#pragma unroll
      for (int B = 0; B < 8; ++B) {
        const unsigned ix = ((h + (pipe_c8(byte, B) & c.a5)) & c.mask0) * G + (unsigned)lane;
        const int w = (int)tab[ix];
        const int pj = pipe_p_get(vj, B), pk = pipe_p_get(vk, B);
        const int pr = (__mul24(w, pj) + __mul24(65536 - w, pk)) >> 16;
        out.set(B, pr);
        const int err = __mul24(pipe_y(byte, B) * 32767 - squash(sp_clamp2k(pr)), (int)c.a4) >> 5;
        tab[ix] = (unsigned)min(max(w + ((__mul24(err, pj - pk) + (1 << 12)) >> 13), 0), 65535);
      }
    }
    L.put_p(I, k, out.get());
    h = h1; byte = byte1; h1 = h2; byte1 = byte2;
    vj = vj1; vk = vk1; vj1 = vj2; vk1 = vk2;
  }
}

// MATCH (libzpaq.cpp:1883-1892, 1985-2008): length / offset / position live in registers.  What update0 reads from
// memory at the end of a byte -- the index entry of the byte's context, the history behind the candidate it names
// (compared backwards with the bytes just coded) and the byte the match predicts next -- has addresses that are
// known when the byte STARTS, so all of it is fetched then and is in registers by the time the 8 bits are done:
//   * the index entry is fetched a byte ahead (forwarded when the next byte hashes to the entry just written);
//   * the 8 history bytes before the candidate come as one unaligned 64-bit load and are compared with the last 8
//     input bytes (which ARE the history on our side) in one xor / count-trailing-zeros; longer matches, candidates
//     that overlap the byte being written, and the buffer's wrap-around take the byte loop of the reference;
//   * both possible "predicted next byte" positions (continuing match, new match) are fetched as well.
// State words: len, offset, pos, predicted byte, 2048/len, last predicted bit, last 8 input bytes (2 words).
template <class Chain, int I, class DT2K>
__device__ __forceinline__ void pipe_match(PipeLane<Chain>& L, const PipeStretch& stretch, const DT2K& dt2k) {
  constexpr CompK c = Chain::comp[I];
  constexpr int ci = Chain::P_CTX[I], sw = Chain::P_STATE[I];
  constexpr unsigned off0 = (unsigned)c.t0, off1 = (unsigned)c.t1, mask = c.mask1;
  typedef unsigned long long __attribute__((aligned(1))) u64u;
  typedef __attribute__((address_space(1))) const u64u g_u64u;
  if (!L.nb) return;
  unsigned ra = 0, rb = 0, rlimit = 0, mpred = 0, mdd = 0, rc = 0;
  unsigned long long hist = 0;
  if (L.chunk > 0) {
    ra = L.state(sw + 0); rb = L.state(sw + 1); rlimit = L.state(sw + 2);
    mpred = L.state(sw + 3); mdd = L.state(sw + 4); rc = L.state(sw + 5);
    hist = (unsigned long long)(unsigned)L.state(sw + 6) | (unsigned long long)(unsigned)L.state(sw + 7) << 32;
  }
  unsigned h = L.ctx(ci, 0), byte = L.byte_at(0);
  const unsigned k1 = L.next(0);
  unsigned h1 = L.ctx(ci, k1), byte1 = L.byte_at(k1);
  unsigned cmv = L.A32(off0 + 4u * (h & c.mask0));
  int st_pos = stretch(mdd & 32767u), st_neg = stretch((0u - mdd) & 32767u);
  for (unsigned k = 0; k < L.nb; ++k) {
    const unsigned k2 = min(k + 2u, L.nb - 1u);
    const unsigned h2 = L.ctx(ci, k2), byte2 = L.byte_at(k2);
    const unsigned eo = off0 + 4u * (h & c.mask0), eon = off0 + 4u * (h1 & c.mask0);
    const unsigned cmvn_mem = L.A32(eon);                        // next byte's index entry (patched below if it is this one)
    // history behind the candidate (positions cmv-8 .. cmv-1), the byte at the candidate, the byte a continuing match predicts
    const unsigned cpos = (cmv - 8u) & mask;
    const bool wraps = cpos + 8u > mask + 1u || mask < 15u;
    const unsigned long long cand = *(g_u64u*)(L.arena + off1 + (wraps ? 0u : cpos));
    const unsigned at_cand = L.A8(off1 + (cmv & mask));
    const unsigned at_cont = L.A8(off1 + ((rlimit + 1u - rb) & mask));
    // (the prediction is one of three values for the whole byte -- stretch(+-2048 / len) or 0 -- looked up when the length changes,
    //  not once per bit: the bit loop is a select and a compare)
    PipeP8 out;
#pragma unroll
    for (int B = 0; B < 8; ++B) {
      const bool on = ra != 0;
      rc = on ? ((mpred >> (7 - B)) & 1u) : rc;
      out.set(B, on ? (rc ? st_neg : st_pos) : 0);                            // stretch(16384) = 0: "p[i]=0"
      ra = ((int)rc != pipe_y(byte, B)) ? 0u : ra;
    }
    L.put_p(I, k, out.get());
    const unsigned wpos = rlimit & mask;                         // where this byte goes
    L.A8(off1 + wpos) = (unsigned char)byte;
    hist = hist << 8 | byte;
    rlimit = (rlimit + 1) & mask;
    bool fresh = false;
    if (ra == 0) {
      rb = rlimit - cmv;
      if (rb & mask) {
        // fast path legal when the 8 candidate bytes were not touched by this byte's store and do not wrap
        // (before 8 bytes have been coded `hist` holds zeros where the reference reads the never-written end of the buffer)
        const bool overlap = ((cmv - 1u - wpos) & mask) < 8u;
        unsigned m = 0;
        if (!wraps && !overlap) {
          const unsigned long long diff = __builtin_bswap64(cand) ^ hist;
          m = diff ? (unsigned)(__builtin_ctzll(diff) >> 3) : 8u;
        }
        ra = m;
        // a match that goes on: 16 more bytes per round trip (four 8-byte loads in flight) while neither range wraps -- the
        // reference's byte loop spends a memory latency per byte, and every lane of the wavefront waits for the one that is in it
        if (!wraps && !overlap && m == 8u) {
          for (;;) {
            if (ra >= 255u) break;
            const unsigned pa = (rlimit - ra - 16u) & mask, pb = (rlimit - ra - rb - 16u) & mask;
            if (pa + 16u > mask + 1u || pb + 16u > mask + 1u) break;
            const unsigned long long a1 = *(g_u64u*)(L.arena + off1 + pa + 8u), b1 = *(g_u64u*)(L.arena + off1 + pb + 8u);
            const unsigned long long a0 = *(g_u64u*)(L.arena + off1 + pa), b0 = *(g_u64u*)(L.arena + off1 + pb);
            const unsigned long long d1 = __builtin_bswap64(a1) ^ __builtin_bswap64(b1), d0 = __builtin_bswap64(a0) ^ __builtin_bswap64(b0);
            if (d1) { ra += (unsigned)(__builtin_ctzll(d1) >> 3); break; }        // (the byte nearest the current position is the first compared)
            if (d0) { ra += 8u + (unsigned)(__builtin_ctzll(d0) >> 3); break; }
            ra += 16u;
          }
          ra = min(ra, 255u);
        }
        if (wraps || overlap || m == 8u)
          while (ra < 255 && L.A8(off1 + ((rlimit - ra - 1) & mask)) == L.A8(off1 + ((rlimit - ra - rb - 1) & mask))) ++ra;
      }
      fresh = true;
    } else ra += ra < 255;
    L.A32(eo) = rlimit;
    if (ra != 0) {
      const unsigned ppos = (rlimit - rb) & mask;                // = cmv for a fresh match, the continuing position otherwise
      const unsigned early = fresh ? at_cand : at_cont;
      mpred = ppos == wpos ? byte : early;
      mdd = dt2k[ra];
      st_pos = stretch(mdd & 32767u); st_neg = stretch((0u - mdd) & 32767u);
    }
    cmv = eon == eo ? rlimit : cmvn_mem;
    h = h1; byte = byte1; h1 = h2; byte1 = byte2;
  }
  L.state(sw + 0) = ra; L.state(sw + 1) = rb; L.state(sw + 2) = rlimit;
  L.state(sw + 3) = mpred; L.state(sw + 4) = mdd; L.state(sw + 5) = rc;
  L.state(sw + 6) = (unsigned)hist; L.state(sw + 7) = (unsigned)(hist >> 32);
}

// MATCH for blocks that fit the component's history buffer (every block compressBlock's methods make: the buffer is at
// least as long as the block).  The history IS the input -- position p of the buffer holds input byte p -- so everything
// update0 reads from it (the bytes behind a candidate, the byte a match predicts next) is read from the block's INPUT,
// which nobody writes: no load ever waits for a store of ours, and every address is known as soon as the index entry is.
// The unit runs a software pipeline over the bytes.  While byte k is coded,
//   * the context of byte k + 6 and the index entry of byte k + 4 are requested;
//   * the entry of byte k + 2 (requested two bytes ago) is final once the entries that bytes k - 2 .. k + 1 store into
//     the same slot have been forwarded arithmetically (byte j stores j + 1); the 24 bytes behind its candidate, the 4
//     bytes from the candidate on and 16 older bytes of our own side are requested;
//   * a running match requests the byte it will predict at the end of byte k + 2;
// and at the end of byte k everything a new match needs arrived two bytes ago.  The slow paths that remain: candidates
// within 24 bytes of the block's start (the reference compares with the never-written end of its buffer there: zeros),
// matches longer than 24 bytes (16 more bytes per round trip), blocks shorter than 24 bytes.  The history buffer in the
// arena is neither read nor written here; the index table is.  State words: len, distance, -, predicted byte,
// 2048 / len, last predicted bit, last 8 input bytes (2 words), the 4 bytes from the match's start, bytes since it started.
template <class Chain, int I, class DT2K>
__device__ __forceinline__ void pipe_match_in(PipeLane<Chain>& L, const PipeStretch& stretch, const DT2K& dt2k) {
  constexpr CompK c = Chain::comp[I];
  constexpr int ci = Chain::P_CTX[I], sw = Chain::P_STATE[I];
  constexpr unsigned off0 = (unsigned)c.t0;
  typedef unsigned long long __attribute__((aligned(1))) u64u;
  typedef __attribute__((address_space(1))) const u64u g_u64u;
  typedef unsigned __attribute__((aligned(1))) u32u;
  typedef __attribute__((address_space(1))) const u32u g_u32u;
  if (!L.nb) return;
  unsigned ra = 0, rd = 0, mpred = 0, mdd = 0, rc = 0, pwin = 0, age = 0;
  unsigned long long hist = 0;
  if (L.chunk > 0) {
    ra = L.state(sw + 0); rd = L.state(sw + 1); mpred = L.state(sw + 3); mdd = L.state(sw + 4); rc = L.state(sw + 5);
    hist = (unsigned long long)(unsigned)L.state(sw + 6) | (unsigned long long)(unsigned)L.state(sw + 7) << 32;
    pwin = L.state(sw + 8); age = L.state(sw + 9);
  }
  const unsigned last = L.nb - 1u, len = L.len, n0 = L.k0;
  const g_u8* const in = L.in;
  const bool big = len >= 24u;          // (shorter blocks: everything through the slow paths)
  // input byte i as the reference's buffer holds it: zero before the block's start (its never-written end)
  auto at = [&](int i) __attribute__((always_inline)) -> unsigned { return i < 0 ? 0u : (unsigned)in[(unsigned)i]; };
  auto slot_of = [&](unsigned hh) __attribute__((always_inline)) -> unsigned { return off0 + 4u * (hh & c.mask0); };
  auto ctx_at = [&](unsigned kk) __attribute__((always_inline)) -> unsigned { return L.ctx(ci, min(kk, last)); };
  struct Cand { unsigned long long a, b1, b0, o1, o0; unsigned pn; };      // candidate bytes -8..-1, -16..-9, -24..-17; ours -15..-8, -23..-16; the 4 bytes from it on
  // what the end of the byte at absolute index n needs when its candidate is cv (static input; clamped where a window
  // would leave the block -- the clamped cases take the slow path and do not look at it)
  // (no branch around a request: where a conditional load joins the main path the compiler waits for it -- the youngest
  //  request -- and with it for everything requested ahead; a block too short for the windows reads its arena instead)
  const g_u8* const src = big ? in : (const g_u8*)L.arena;
  const unsigned lim = big ? len : 64u;
  auto fetch = [&](unsigned n, unsigned cv) __attribute__((always_inline)) -> Cand {
    Cand r;
    const unsigned base = min(cv >= 24u ? cv : 24u, lim), nn = min(n >= 23u ? n : 23u, lim - 1u);      // (requests for the bytes past the block's end stay inside it)
    r.a = *(g_u64u*)(src + (base - 8u));
    r.b1 = *(g_u64u*)(src + (base - 16u));
    r.b0 = *(g_u64u*)(src + (base - 24u));
    r.o1 = *(g_u64u*)(src + (nn - 15u));
    r.o0 = *(g_u64u*)(src + (nn - 23u));
    const unsigned pa = min(cv, lim - 4u);
    r.pn = (unsigned)*(g_u32u*)(src + pa) >> (8u * min(cv - pa, 3u));
    return r;
  };
  // ---- prime the pipeline for the chunk's first bytes (everything earlier chunks stored is long done)
  unsigned h4 = ctx_at(4), h5 = ctx_at(5);
  unsigned em2 = 0xFFFFFFFFu, em1 = 0xFFFFFFFFu;                               // slots of the two bytes before byte k (their stores: n - 1, n)
  unsigned e0 = slot_of(ctx_at(0)), e1 = slot_of(ctx_at(1)), e2 = slot_of(ctx_at(2)), e3 = slot_of(ctx_at(3));
  unsigned b0 = L.byte_at(0), b1 = L.byte_at(min(1u, last));
  const unsigned r0 = L.A32(e0), r1 = L.A32(e1);
  unsigned ri2 = L.A32(e2), ri3 = L.A32(e3);                                   // raw entries of bytes k + 2, k + 3
  unsigned cm0 = r0, cm1 = e1 == e0 ? n0 + 1u : r1;                            // final candidates of bytes k, k + 1
  Cand c0 = fetch(n0, cm0), c1 = fetch(n0 + 1u, cm1);
  int st_pos = stretch(mdd & 32767u), st_neg = stretch((0u - mdd) & 32767u);
  unsigned sp0 = 0, sp1 = 0;                                                   // a running match's predicted byte for the end of bytes k, k + 1
  if (ra != 0u) { sp0 = at((int)min(n0 + 1u - rd, len - 1u)); sp1 = at((int)min(n0 + 2u - rd, len - 1u)); }
  for (unsigned k = 0; k < L.nb; ++k) {
    const unsigned n = n0 + k, n1 = n + 1u;
    const unsigned byte = b0;
    // ---- requests for the bytes ahead
    const unsigned h6 = ctx_at(k + 6u);
    const unsigned b2 = L.byte_at(min(k + 2u, last));
    const unsigned e4 = slot_of(h4);
    const unsigned ri4 = L.A32(e4);
    unsigned cm2 = ri2;                                                        // byte k + 2's candidate: the youngest store into its slot wins
    cm2 = e2 == em2 ? n - 1u : cm2;
    cm2 = e2 == em1 ? n : cm2;
    cm2 = e2 == e0 ? n + 1u : cm2;
    cm2 = e2 == e1 ? n + 2u : cm2;
    const Cand c2 = fetch(n + 2u, cm2);
    const unsigned sp2 = (unsigned)src[min(n + 3u - rd, lim - 1u)];                  // (meaningful while a match runs; rd <= n)
    // ---- the byte's 8 bits (libzpaq.cpp:1883-1892, 1985-1990)
    // (the prediction is one of three values for the whole byte -- stretch(+-2048 / len) or 0 -- looked up when the length changes,
    //  not once per bit: the bit loop is a select and a compare)
    PipeP8 out;
#pragma unroll
    for (int B = 0; B < 8; ++B) {
      const bool on = ra != 0;
      rc = on ? ((mpred >> (7 - B)) & 1u) : rc;
      out.set(B, on ? (rc ? st_neg : st_pos) : 0);                            // stretch(16384) = 0: "p[i]=0"
      ra = ((int)rc != pipe_y(byte, B)) ? 0u : ra;
    }
    L.put_p(I, k, out.get());
    hist = hist << 8 | byte;
    // ---- end of the byte (libzpaq.cpp:1991-2008)
    bool fresh = false;
    if (ra == 0u) {
      const unsigned cv = cm0;
      rd = n1 - cv;                                                            // (never 0: no earlier byte can have stored n + 1)
      // candidate 0 = a context never seen before (the index table starts zeroed): the reference compares with the zeros
      // at the end of its buffer -- by far the most frequent case besides a real candidate, and it needs no load at all
      const bool fast = big && cv >= 24u, zero = big && cv == 0u && n >= 23u;
      unsigned m = 0;
      if (fast || zero) {
        const unsigned long long d = (zero ? 0ull : __builtin_bswap64(c0.a)) ^ hist;
        m = d ? (unsigned)(__builtin_ctzll(d) >> 3) : 8u;
        if (m == 8u) {
          const unsigned long long d1 = (zero ? 0ull : __builtin_bswap64(c0.b1)) ^ __builtin_bswap64(c0.o1);
          if (d1) m = 8u + (unsigned)(__builtin_ctzll(d1) >> 3);
          else {
            const unsigned long long d0 = (zero ? 0ull : __builtin_bswap64(c0.b0)) ^ __builtin_bswap64(c0.o0);
            m = d0 ? 16u + (unsigned)(__builtin_ctzll(d0) >> 3) : 24u;
          }
        }
      }
      ra = m;
      if (fast && m == 24u) {
        for (;;) {             // a longer match: 16 bytes per round trip while the candidate's side stays inside the block
          if (ra >= 255u || cv < ra + 16u) break;
          const unsigned long long a1 = *(g_u64u*)(in + (n1 - ra - 8u)), x1 = *(g_u64u*)(in + (cv - ra - 8u));
          const unsigned long long a0 = *(g_u64u*)(in + (n1 - ra - 16u)), x0 = *(g_u64u*)(in + (cv - ra - 16u));
          const unsigned long long d1 = __builtin_bswap64(a1) ^ __builtin_bswap64(x1), d0 = __builtin_bswap64(a0) ^ __builtin_bswap64(x0);
          if (d1) { ra += (unsigned)(__builtin_ctzll(d1) >> 3); break; }
          if (d0) { ra += 8u + (unsigned)(__builtin_ctzll(d0) >> 3); break; }
          ra += 16u;
        }
        ra = min(ra, 255u);
      }
      if ((!fast && !zero) || (m == 24u && ra < 255u && (zero || cv < ra + 16u)))
        while (ra < 255u && at((int)(n1 - ra - 1u)) == at((int)cv - (int)ra - 1)) ++ra;
      fresh = true;
    } else ra += ra < 255u;
    L.A32(e0) = n1;
    if (ra != 0u) {
      if (fresh) { pwin = big ? c0.pn : 0u; age = 0u; }
      // the byte the match predicts next, in[n1 - rd]: from the window behind the candidate while the match is young, from the
      // request made two bytes ago afterwards
      mpred = !big ? at((int)min(n1 - rd, len - 1u)) : (age < 4u ? (pwin >> (8u * age)) & 255u : sp0);
      mdd = dt2k[ra];
      st_pos = stretch(mdd & 32767u); st_neg = stretch((0u - mdd) & 32767u);
      age += age < 255u;
    }
    // ---- the pipeline moves on by one byte
    em2 = em1; em1 = e0; e0 = e1; e1 = e2; e2 = e3; e3 = e4;
    ri2 = ri3; ri3 = ri4;
    h4 = h5; h5 = h6;
    cm0 = cm1; cm1 = cm2;
    c0 = c1; c1 = c2;
    sp0 = sp1; sp1 = sp2;
    b0 = b1; b1 = b2;
  }
  L.state(sw + 0) = ra; L.state(sw + 1) = rd; L.state(sw + 3) = mpred; L.state(sw + 4) = mdd; L.state(sw + 5) = rc;
  L.state(sw + 6) = (unsigned)hist; L.state(sw + 7) = (unsigned)(hist >> 32);
  L.state(sw + 8) = pwin; L.state(sw + 9) = age;
}

// the MATCH unit: from the input when the history ring of every block of the wavefront can never be read where it has
// been written a lap earlier (a property of the blocks' lengths: the same choice at every chunk), through the buffer in
// the arena otherwise.  update0 compares up to 255 bytes backwards from a candidate (libzpaq.cpp:1995-1998) with indices
// taken modulo the buffer size: behind a candidate near the block's start that reaches the END of the ring, which
// pipe_match_in takes for never written (zeros) -- true only while len + 255 <= buffer size.
template <class Chain, int I, class DT2K>
__device__ __forceinline__ void pipe_match_any(PipeLane<Chain>& L, const PipeStretch& stretch, const DT2K& dt2k) {
  constexpr CompK c = Chain::comp[I];
  if (pipe_any(L.live && L.len + 255u > c.mask1 + 1u)) pipe_match<Chain, I>(L, stretch, dt2k);
  else pipe_match_in<Chain, I>(L, stretch, dt2k);
}

// AVG (libzpaq.cpp:1894-1896)
template <class Chain, int I>
__device__ __forceinline__ void pipe_avg(PipeLane<Chain>& L) {
  constexpr CompK c = Chain::comp[I];
  for (unsigned k = 0; k < L.nb; ++k) {
    const uint4 vj = L.p((int)c.a1, k), vk = L.p((int)c.a2, k);
    PipeP8 out;
#pragma unroll
    for (int B = 0; B < 8; ++B) out.set(B, (__mul24(pipe_p_get(vj, B), (int)c.a3) + __mul24(pipe_p_get(vk, B), 256 - (int)c.a3)) >> 8);
    L.put_p(I, k, out.get());
  }
}

// MIX2 (libzpaq.cpp:1898-1908, 2010-2021).  Device layout: one weight per dword.
template <class Chain, int I, class SQ>
__device__ __forceinline__ void pipe_mix2(PipeLane<Chain>& L, const SQ& squash) {
  constexpr CompK c = Chain::comp[I];
  constexpr int ci = Chain::P_CTX[I];
  constexpr bool single = c.mask0 == 0u;                          // one weight: it stays in a register
  constexpr bool batch = !single && c.a5 == 255u && c.mask0 >= 255u;   // the 8 weights of a byte are distinct
  if (!L.nb) return;
  auto addr = [&](unsigned hh, unsigned bytev, int B) __attribute__((always_inline)) -> unsigned {
    return (unsigned)c.t0 + 4u * ((hh + (pipe_c8(bytev, B) & c.a5)) & c.mask0);
  };
  unsigned wreg = 0;
  if constexpr (single) wreg = L.A32((unsigned)c.t0);
  unsigned h = single ? 0u : (unsigned)L.ctx(ci, 0), byte = L.byte_at(0);
  const unsigned k1 = L.next(0);
  unsigned h1 = single ? 0u : (unsigned)L.ctx(ci, k1), byte1 = L.byte_at(k1);
  uint4 vj = L.p((int)c.a2, 0), vk = L.p((int)c.a3, 0);
  uint4 vj1 = L.p((int)c.a2, k1), vk1 = L.p((int)c.a3, k1);
  unsigned v[8];
  if constexpr (batch) {
#pragma unroll
    for (int B = 0; B < 8; ++B) v[B] = L.A32(addr(h, byte, B));
  }
  for (unsigned k = 0; k < L.nb; ++k) {
    const unsigned k2 = min(k + 2u, L.nb - 1u);
    const unsigned h2 = single ? 0u : (unsigned)L.ctx(ci, k2), byte2 = L.byte_at(k2);
    const uint4 vj2 = L.p((int)c.a2, k2), vk2 = L.p((int)c.a3, k2);
    // next byte's weights: same context -> only equal bit positions coincide (c8 ranges are disjoint), forwarded;
    // contexts less than 256 apart -> any position may coincide: fetched after the stores
    const bool same = h1 == h;
    const bool late = !same && (((h1 - h) & c.mask0) < 256u || ((h - h1) & c.mask0) < 256u);
    unsigned vn[8], nv[8];
    if constexpr (batch) {
      if (!late) {
#pragma unroll
        for (int B = 0; B < 8; ++B) vn[B] = L.A32(addr(h1, byte1, B));
      }
    }
    PipeP8 out;
#pragma unroll
    for (int B = 0; B < 8; ++B) {
      const unsigned off = addr(h, byte, B);
      const int w = (int)(single ? wreg : (batch ? v[B] : (unsigned)L.A32(off)));
      const int pj = pipe_p_get(vj, B), pk = pipe_p_get(vk, B);
      const int pr = (__mul24(w, pj) + __mul24(65536 - w, pk)) >> 16;   // 17-bit x 12-bit
      out.set(B, pr);
      const int err = __mul24(pipe_y(byte, B) * 32767 - squash(sp_clamp2k(pr)), (int)c.a4) >> 5;
      nv[B] = (unsigned)min(max(w + ((__mul24(err, pj - pk) + (1 << 12)) >> 13), 0), 65535);   // 19-bit x 13-bit
      if constexpr (single) wreg = nv[B]; else L.A32(off) = nv[B];
    }
    L.put_p(I, k, out.get());
    if constexpr (batch) {
      if (late) {
#pragma unroll
        for (int B = 0; B < 8; ++B) vn[B] = L.A32(addr(h1, byte1, B));
      }
#pragma unroll
      for (int B = 0; B < 8; ++B) v[B] = (same && addr(h1, byte1, B) == addr(h, byte, B)) ? nv[B] : vn[B];
    }
    h = h1; byte = byte1; h1 = h2; byte1 = byte2;
    vj = vj1; vk = vk1; vj1 = vj2; vk1 = vk2;
  }
  if constexpr (single) L.A32((unsigned)c.t0) = wreg;
}

// SSE (libzpaq.cpp:1933-1944, 2041-2044)
template <class Chain, int I, class DT>
__device__ __forceinline__ void pipe_sse(PipeLane<Chain>& L, const PipeStretch& stretch, const DT& dt) {
  constexpr CompK c = Chain::comp[I];
  constexpr int ci = Chain::P_CTX[I];
  constexpr bool batch = c.mask0 >= 32u * 256u - 1u;               // the 8 rows of a byte are distinct
  constexpr unsigned rowmask = c.mask0 >> 5;                         // rows of 32 entries
  if (!L.nb) return;
  // entry pair read for bit B: index of the lower one and the interpolation weight (libzpaq.cpp:1935-1939)
  auto index = [&](unsigned hh, unsigned bytev, const uint4& pv, int B, int& wt) __attribute__((always_inline)) -> unsigned {
    int pq = pipe_p_get(pv, B) + 992;
    pq = min(max(pq, 0), 1983);
    wt = pq & 63;
    return (((hh + pipe_c8(bytev, B)) * 32u) & c.mask0) + (unsigned)(pq >> 6);
  };
  unsigned h = L.ctx(ci, 0), byte = L.byte_at(0);
  const unsigned k1 = L.next(0);
  unsigned h1 = L.ctx(ci, k1), byte1 = L.byte_at(k1);
  uint4 vj = L.p((int)c.a2, 0), vj1 = L.p((int)c.a2, k1);
  unsigned e0[8], e1[8], ix[8];
  int wt[8];
#pragma unroll
  for (int B = 0; B < 8; ++B) {
    ix[B] = index(h, byte, vj, B, wt[B]);
    if constexpr (batch) {
      e0[B] = L.A32((unsigned)c.t0 + 4u * (ix[B] & c.mask0));
      e1[B] = L.A32((unsigned)c.t0 + 4u * ((ix[B] + 1u) & c.mask0));
    }
  }
  for (unsigned k = 0; k < L.nb; ++k) {
    const unsigned k2 = min(k + 2u, L.nb - 1u);
    const unsigned h2 = L.ctx(ci, k2), byte2 = L.byte_at(k2);
    const uint4 vj2 = L.p((int)c.a2, k2);
    // next byte's entries: same context -> only equal bit positions share a row, forwarded; rows less than 256 apart ->
    // fetched after the stores
    const bool same = h1 == h;
    const bool late = !same && (((h1 - h) & rowmask) < 256u || ((h - h1) & rowmask) < 256u);
    unsigned n0[8], n1[8], ixn[8], ti[8], nv[8];
    int wtn[8];
#pragma unroll
    for (int B = 0; B < 8; ++B) ixn[B] = index(h1, byte1, vj1, B, wtn[B]);
    if constexpr (batch) {
      if (!late) {
#pragma unroll
        for (int B = 0; B < 8; ++B) {
          n0[B] = L.A32((unsigned)c.t0 + 4u * (ixn[B] & c.mask0));
          n1[B] = L.A32((unsigned)c.t0 + 4u * ((ixn[B] + 1u) & c.mask0));
        }
      }
    }
    PipeP8 out;
#pragma unroll
    for (int B = 0; B < 8; ++B) {
      if constexpr (!batch) {
        e0[B] = L.A32((unsigned)c.t0 + 4u * (ix[B] & c.mask0));
        e1[B] = L.A32((unsigned)c.t0 + 4u * ((ix[B] + 1u) & c.mask0));
      }
      const int w = wt[B];
      out.set(B, stretch((__umul24(e0[B] >> 10, (unsigned)(64 - w)) + __umul24(e1[B] >> 10, (unsigned)w)) >> 13));
      const unsigned tv = (w >> 5) ? e1[B] : e0[B];
      ti[B] = (ix[B] + (unsigned)(w >> 5)) & c.mask0;
      nv[B] = pipe_train(tv, pipe_y(byte, B), (unsigned)dt[tv & 0x3ffu], c.limit);
      L.A32((unsigned)c.t0 + 4u * ti[B]) = nv[B];
    }
    L.put_p(I, k, out.get());
    if constexpr (batch) {
      if (late) {
#pragma unroll
        for (int B = 0; B < 8; ++B) {
          n0[B] = L.A32((unsigned)c.t0 + 4u * (ixn[B] & c.mask0));
          n1[B] = L.A32((unsigned)c.t0 + 4u * ((ixn[B] + 1u) & c.mask0));
        }
      }
#pragma unroll
      for (int B = 0; B < 8; ++B) {
        e0[B] = (same && (ixn[B] & c.mask0) == ti[B]) ? nv[B] : n0[B];
        e1[B] = (same && ((ixn[B] + 1u) & c.mask0) == ti[B]) ? nv[B] : n1[B];
      }
    }
#pragma unroll
    for (int B = 0; B < 8; ++B) { ix[B] = ixn[B]; wt[B] = wtn[B]; }
    h = h1; byte = byte1; h1 = h2; byte1 = byte2;
    vj = vj1; vj1 = vj2;
  }
}

// ---- CM / MIX2 / SSE with a lane per (block, BIT POSITION) ----------------------------------------------------------
// (SSE in both shapes of the encoder -- it is the longest chain of the light kernel and wins at every batch size measured;
//  CM and MIX2 in latency mode only: PipeOptions in host/codegen.hpp)
// Same idea as pipe_mix_bits_body: the table word a bit uses is indexed by the bit's position in the byte (hmap4 / c8
// are part of the index) and training touches that word only, so with all bits known the 8 bits of a byte are
// independent.  8 lanes per block, the 8 positions of a block in ONE wavefront (a workgroup = PIPE_G lanes = PIPE_G / 8
// blocks; a unit = several workgroups per group), table words / inputs / contexts fetched LIGHT_DEPTH bytes ahead, a word
// rewritten since its fetch taken from the lane's own history (same context) or fetched again (contexts close enough
// for DIFFERENT positions to meet).
template <int D>
struct PipeBitsWin {
  static constexpr int HN = D;      // bytes whose stores a fetched word may have missed: the D - 1 done since its fetch
                                    // and the one stored just before it (another lane's store is ordered only by the re-fetch)
};

__device__ __forceinline__ unsigned pipe_hmap4_at(unsigned byte, unsigned B) {
  const unsigned hi = byte >> 4, lo = byte & 15u;
  return B < 4u ? ((1u << B) | (hi >> (4u - B))) : (256u + 16u * hi + ((1u << (B - 4u)) | (lo >> (8u - B))));
}
__device__ __forceinline__ unsigned pipe_c8_at(unsigned byte, unsigned B) { return (1u << B) | (byte >> (8u - B)); }

// The 8 lanes of a block (lane & 7 = bit position) hold the 8 half-words of one stream element: gathered into the lane of
// position 0 by DPP shifts inside the row of 16 lanes and stored as ONE 16-byte element (every lane of the wavefront takes
// part in the shifts: call in wave-uniform control flow).
template <class Chain>
__device__ __forceinline__ void pipe_put_bits(const PipeLane<Chain>& L, int I, unsigned k, unsigned B, int pr, bool on) {
  const int h = pr & 0xFFFF;
  const int t = h | (__builtin_amdgcn_update_dpp(0, h, 0x101, 0xF, 0xF, true) << 16);        // row_shl:1 -- even lanes: positions B, B + 1
  const int y = __builtin_amdgcn_update_dpp(0, t, 0x102, 0xF, 0xF, true);                    // row_shl:2
  const int z = __builtin_amdgcn_update_dpp(0, t, 0x104, 0xF, 0xF, true);
  const int w = __builtin_amdgcn_update_dpp(0, t, 0x106, 0xF, 0xF, true);
  if (on && B == 0u) L.put_p(I, k, make_uint4((unsigned)t, (unsigned)y, (unsigned)z, (unsigned)w));
}

template <class Chain, int I, class DT>
__device__ __forceinline__ void pipe_cm_bits(PipeLane<Chain>& L, unsigned B, const PipeStretch& stretch, const DT& dt) {
  constexpr CompK c = Chain::comp[I];
  constexpr int ci = Chain::P_CTX[I], D = Chain::LIGHT_DEPTH, HN = PipeBitsWin<D>::HN, W = 2 * D;
  static_assert(c.mask0 >= 511u, "CM bit lanes need the 8 words of a byte to be distinct");
  if (!L.nb) return;
  const unsigned last = L.nb - 1u;
  // ring of W = 2 D slots (slot = byte index mod W, fixed registers): context and byte fetched W bytes ahead, the table
  // word D bytes ahead
  unsigned hx[W], bx[W], aq[W], vq[W];
  unsigned ha[HN], hh[HN], hv[HN];
  auto near = [&](int sl) __attribute__((always_inline)) {
    aq[sl] = (unsigned)c.t0 + 4u * ((hx[sl] ^ pipe_hmap4_at(bx[sl], B)) & c.mask0);
    vq[sl] = L.A32(aq[sl]);
  };
#pragma unroll
  for (int sl = 0; sl < W; ++sl) { const unsigned kk = min((unsigned)sl, last); hx[sl] = L.ctx(ci, kk); bx[sl] = L.byte_at(kk); }
#pragma unroll
  for (int sl = 0; sl < D; ++sl) near(sl);
#pragma unroll
  for (int i = 0; i < HN; ++i) { ha[i] = 0xFFFFFFFFu; hh[i] = hx[0]; hv[i] = 0u; }
  for (unsigned kb = 0; pipe_any(kb < L.nb); kb += (unsigned)W) {
#pragma unroll
    for (int sl = 0; sl < W; ++sl) {
      const unsigned k = kb + (unsigned)sl;
      const bool on = k < L.nb;
      const unsigned hcur = hx[sl], addr = aq[sl];
      const int y = (int)((bx[sl] >> (7u - B)) & 1u);
      unsigned v = vq[sl];
      {
        bool late = false;
#pragma unroll
        for (int i = HN - 1; i >= 0; --i) {
          v = addr == ha[i] ? hv[i] : v;
          late = late || (hcur != hh[i] && ((hcur ^ hh[i]) & c.mask0) < 512u);
        }
        if (pipe_any(late)) { pipe_stores_done(); if (late) v = pipe_opaque(L.A32(addr)); }
      }
      const int pr = stretch(v >> 17);
      const unsigned nv = pipe_train(v, y, (unsigned)dt[v & 0x3ffu], c.limit);
      if (on) L.A32(addr) = nv;
      pipe_put_bits(L, I, k, B, pr, on);
#pragma unroll
      for (int i = HN - 1; i > 0; --i) { ha[i] = ha[i - 1]; hh[i] = hh[i - 1]; hv[i] = hv[i - 1]; }
      ha[0] = addr; hh[0] = hcur; hv[0] = nv;
      {
        const unsigned kw = min(k + (unsigned)W, last);
        hx[sl] = L.ctx(ci, kw); bx[sl] = L.byte_at(kw);
        near((sl + D) % W);
      }
    }
  }
}

template <class Chain, int I, class SQ>
__device__ __forceinline__ void pipe_mix2_bits(PipeLane<Chain>& L, unsigned B, const SQ& squash) {
  constexpr CompK c = Chain::comp[I];
  constexpr int ci = Chain::P_CTX[I], D = Chain::LIGHT_DEPTH, HN = PipeBitsWin<D>::HN, W = 2 * D;
  static_assert(c.a5 == 255u && c.mask0 >= 255u, "MIX2 bit lanes need the 8 weights of a byte to be distinct");
  if (!L.nb) return;
  const unsigned last = L.nb - 1u;
  auto input = [&](int t, unsigned kk) __attribute__((always_inline)) -> int {
    return (int)*(const g_i16*)((const g_u8*)&L.p(t, kk) + 2u * B);
  };
  unsigned hx[W], bx[W], aq[W], vq[W];
  int pj[W], pk[W];
  unsigned ha[HN], hh[HN], hv[HN];
  auto near = [&](int sl, unsigned kk) __attribute__((always_inline)) {
    aq[sl] = (unsigned)c.t0 + 4u * ((hx[sl] + pipe_c8_at(bx[sl], B)) & c.mask0);
    vq[sl] = L.A32(aq[sl]);
    pj[sl] = input((int)c.a2, kk); pk[sl] = input((int)c.a3, kk);
  };
#pragma unroll
  for (int sl = 0; sl < W; ++sl) { const unsigned kk = min((unsigned)sl, last); hx[sl] = L.ctx(ci, kk); bx[sl] = L.byte_at(kk); }
#pragma unroll
  for (int sl = 0; sl < D; ++sl) near(sl, min((unsigned)sl, last));
#pragma unroll
  for (int i = 0; i < HN; ++i) { ha[i] = 0xFFFFFFFFu; hh[i] = hx[0]; hv[i] = 0u; }
  for (unsigned kb = 0; pipe_any(kb < L.nb); kb += (unsigned)W) {
#pragma unroll
    for (int sl = 0; sl < W; ++sl) {
      const unsigned k = kb + (unsigned)sl;
      const bool on = k < L.nb;
      const unsigned hcur = hx[sl], addr = aq[sl];
      const int y = (int)((bx[sl] >> (7u - B)) & 1u);
      unsigned v = vq[sl];
      {
        bool late = false;
#pragma unroll
        for (int i = HN - 1; i >= 0; --i) {
          v = addr == ha[i] ? hv[i] : v;
          late = late || (hcur != hh[i] && (((hcur - hh[i]) & c.mask0) < 256u || ((hh[i] - hcur) & c.mask0) < 256u));
        }
        if (pipe_any(late)) { pipe_stores_done(); if (late) v = pipe_opaque(L.A32(addr)); }
      }
      const int w = (int)v, qj = pj[sl], qk = pk[sl];
      const int pr = (__mul24(w, qj) + __mul24(65536 - w, qk)) >> 16;   // 17-bit x 12-bit
      const int err = __mul24(y * 32767 - squash(sp_clamp2k(pr)), (int)c.a4) >> 5;
      const unsigned nv = (unsigned)min(max(w + ((__mul24(err, qj - qk) + (1 << 12)) >> 13), 0), 65535);   // 19-bit x 13-bit
      if (on) L.A32(addr) = nv;
      pipe_put_bits(L, I, k, B, pr, on);
#pragma unroll
      for (int i = HN - 1; i > 0; --i) { ha[i] = ha[i - 1]; hh[i] = hh[i - 1]; hv[i] = hv[i - 1]; }
      ha[0] = addr; hh[0] = hcur; hv[0] = nv;
      {
        const unsigned kw = min(k + (unsigned)W, last);
        hx[sl] = L.ctx(ci, kw); bx[sl] = L.byte_at(kw);
        near((sl + D) % W, min(k + (unsigned)D, last));
      }
    }
  }
}

template <class Chain, int I, class DT>
__device__ __forceinline__ void pipe_sse_bits(PipeLane<Chain>& L, unsigned B, const PipeStretch& stretch, const DT& dt) {
  constexpr CompK c = Chain::comp[I];
  constexpr int ci = Chain::P_CTX[I], D = Chain::LIGHT_DEPTH, HN = PipeBitsWin<D>::HN, W = 2 * D;
  constexpr unsigned rowmask = c.mask0 >> 5;                         // rows of 32 entries
  static_assert(c.mask0 >= 32u * 256u - 1u, "SSE bit lanes need the 8 rows of a byte to be distinct");
  if (!L.nb) return;
  const unsigned last = L.nb - 1u;
  auto input = [&](unsigned kk) __attribute__((always_inline)) -> int {
    return (int)*(const g_i16*)((const g_u8*)&L.p((int)c.a2, kk) + 2u * B);
  };
  // the entry pair depends on the input prediction, so that travels with the context: all three W bytes ahead
  unsigned hx[W], bx[W], ix[W], wt[W], e0[W], e1[W];
  int px[W];
  unsigned ha[HN], hh[HN], hv[HN];           // entry trained (index), context, value stored
  // entry pair read for this position: index of the lower one and the interpolation weight (libzpaq.cpp:1935-1939)
  auto near = [&](int sl) __attribute__((always_inline)) {
    const int pq = min(max(px[sl] + 992, 0), 1983);
    wt[sl] = (unsigned)pq & 63u;
    ix[sl] = (((hx[sl] + pipe_c8_at(bx[sl], B)) * 32u) & c.mask0) + (unsigned)(pq >> 6);
    e0[sl] = L.A32((unsigned)c.t0 + 4u * (ix[sl] & c.mask0));
    e1[sl] = L.A32((unsigned)c.t0 + 4u * ((ix[sl] + 1u) & c.mask0));
  };
#pragma unroll
  for (int sl = 0; sl < W; ++sl) {
    const unsigned kk = min((unsigned)sl, last);
    hx[sl] = L.ctx(ci, kk); bx[sl] = L.byte_at(kk); px[sl] = input(kk);
  }
#pragma unroll
  for (int sl = 0; sl < D; ++sl) near(sl);
#pragma unroll
  for (int i = 0; i < HN; ++i) { ha[i] = 0xFFFFFFFFu; hh[i] = hx[0]; hv[i] = 0u; }
  for (unsigned kb = 0; pipe_any(kb < L.nb); kb += (unsigned)W) {
#pragma unroll
    for (int sl = 0; sl < W; ++sl) {
      const unsigned k = kb + (unsigned)sl;
      const bool on = k < L.nb;
      const unsigned hcur = hx[sl];
      const int y = (int)((bx[sl] >> (7u - B)) & 1u);
      const unsigned i0 = ix[sl] & c.mask0, i1 = (ix[sl] + 1u) & c.mask0;
      unsigned v0 = e0[sl], v1 = e1[sl];
      {
        bool late = false;
#pragma unroll
        for (int i = HN - 1; i >= 0; --i) {
          v0 = i0 == ha[i] ? hv[i] : v0;
          v1 = i1 == ha[i] ? hv[i] : v1;
          late = late || (hcur != hh[i] && (((hcur - hh[i]) & rowmask) < 256u || ((hh[i] - hcur) & rowmask) < 256u));
        }
        if (pipe_any(late)) {
          pipe_stores_done();
          if (late) { v0 = pipe_opaque(L.A32((unsigned)c.t0 + 4u * i0)); v1 = pipe_opaque(L.A32((unsigned)c.t0 + 4u * i1)); }
        }
      }
      const unsigned w = wt[sl];
      const int pr = stretch((__umul24(v0 >> 10, 64u - w) + __umul24(v1 >> 10, w)) >> 13);
      const unsigned tv = (w >> 5) ? v1 : v0;
      const unsigned ti = (w >> 5) ? i1 : i0;
      const unsigned nv = pipe_train(tv, y, (unsigned)dt[tv & 0x3ffu], c.limit);
      if (on) L.A32((unsigned)c.t0 + 4u * ti) = nv;
      pipe_put_bits(L, I, k, B, pr, on);
#pragma unroll
      for (int i = HN - 1; i > 0; --i) { ha[i] = ha[i - 1]; hh[i] = hh[i - 1]; hv[i] = hv[i - 1]; }
      ha[0] = ti; hh[0] = hcur; hv[0] = nv;
      {
        const unsigned kw = min(k + (unsigned)W, last);
        hx[sl] = L.ctx(ci, kw); bx[sl] = L.byte_at(kw); px[sl] = input(kw);
        near((sl + D) % W);
      }
    }
  }
}

// CODER: Encoder::compress / encode (libzpaq.cpp:2402-2447) fed by the last component's stream.
template <class Chain, class SQ>
__device__ __forceinline__ void pipe_coder(PipeLane<Chain>& L, const PipeArgs& a, const SQ& squash) {
  constexpr int sw = Chain::CODER_STATE;
  const unsigned nchunks = max((L.len + (unsigned)Chain::PIPE_C - 1u) / (unsigned)Chain::PIPE_C, 1u);
  const bool active = L.live && L.chunk >= 0 && (unsigned)L.chunk < nchunks;
  if (!active) return;
  unsigned low = 1, high = 0xFFFFFFFFu, n = 0;
  // a block of several segments: the end-of-segment code (and nothing else) separates them, the model never notices
  unsigned seg = 0, seg_end = 0xFFFFFFFFu;
  const bool multi = L.nseg > 1;
  if (L.chunk > 0) { low = L.state(sw + 0); high = L.state(sw + 1); n = L.state(sw + 2); seg = L.state(sw + 3); }
  typedef __attribute__((address_space(1))) SegRange g_seg;
  g_seg* const segs = (g_seg*)L.segs;
  if (multi) seg_end = segs[seg].in_end;
  // coded bytes leave through a small buffer: {ob_hi, ob_lo} holds the last n - nf bytes (the newest in the lowest place),
  // four go out in one store.  A byte store per coded byte was most of the shift-out loop -- which every lane of the
  // wavefront sits through whenever one of them has a byte to emit -- and the coder is the slowest unit of a short chain.
  // Lanes flush together once per input byte; inside a byte only a lane about to run out of room (rare) does.
  typedef unsigned __attribute__((aligned(1))) u32u;
  typedef __attribute__((address_space(1))) u32u g_u32u;
  unsigned ob_hi = 0, ob_lo = 0;
  unsigned nf = n;
  auto flush4 = [&]() __attribute__((always_inline)) {
    const unsigned sh = 8u * (n - nf - 4u);                                  // 0 .. 32: the four oldest bytes start there
    const unsigned w4 = sh >= 32u ? ob_hi : (unsigned)((((unsigned long long)ob_hi << 32) | ob_lo) >> sh);
    if (nf + 4u <= L.out_cap) *(g_u32u*)(L.out + nf) = __builtin_bswap32(w4);  // (the oldest byte is the top one)
    else for (unsigned i = 0; i < 4u; ++i) if (nf + i < L.out_cap) L.out[nf + i] = (unsigned char)(w4 >> (24u - 8u * i));
    nf += 4u;
  };
  auto flush_all = [&]() __attribute__((always_inline)) {
    if (n - nf >= 4u) flush4();
    for (; nf < n; ++nf) if (nf < L.out_cap) L.out[nf] = (unsigned char)(ob_lo >> (8u * (n - nf - 1u)));
  };
  // Encoder::encode (libzpaq.cpp:2402-2416).  Its loop "while ((high ^ low) < 2^24) { out(high >> 24); high = high << 8 | 255;
  // low <<= 8; low += (low == 0); }" in closed form: with 32 lanes in lockstep SOME lane shifts a byte out at nearly every bit
  // of incompressible data, so the loop's branches were on every bit's path (profiles/unit_isa.py: the coder is the longest
  // chain of a short model).  The loop runs k = (leading bytes high and low share) times, 0 .. 4: it emits the top k bytes of
  // high, leaves high << 8k with ones shifted in, and low << 8k -- except that a turn which finds low << 8 == 0 makes it 1:
  // that happens first at turn j = ceil((32 - ctz(low)) / 8) and never again (1 << 8 t != 0 for t < 4), so for j <= k the result
  // is 1 << 8 (k - j).  Checked against the loop on 4 x 10^8 states (low >= 1 always: it starts at 1, mid + 1 <= high, and
  // the rule above keeps it there; the OR below covers 0 as the loop would).
  auto encode = [&](int y, unsigned pr) __attribute__((always_inline)) {
    const unsigned mid = low + (unsigned)(((unsigned long long)(high - low) * pr) >> 16);
    if (y) high = mid; else low = mid + 1;
    const unsigned x = high ^ low;
    const unsigned k = x ? (unsigned)__builtin_clz(x) >> 3 : 4u;
    const unsigned sh = (8u * k) & 31u;
    const unsigned emitted = (unsigned)(((unsigned long long)high << (8u * k)) >> 32);       // the top k bytes of high
    const unsigned j = (39u - (unsigned)__builtin_ctz(low | 0x80000000u)) >> 3;
    const unsigned nh = k == 4u ? 0xFFFFFFFFu : ((high << sh) | ((1u << sh) - 1u));
    const unsigned nl = j <= k ? 1u << ((8u * (k - j)) & 31u) : low << sh;
    high = nh;
    low = nl;
    const unsigned long long ob = ((((unsigned long long)ob_hi << 32) | ob_lo) << (8u * k)) | emitted;
    ob_hi = (unsigned)(ob >> 32);
    ob_lo = (unsigned)ob;
    n += k;
    if (n - nf > 4u) flush4();                 // (an encode emits at most 4 bytes and finds at most 4 waiting: the buffer holds 8)
  };
  if (L.nb) {
    unsigned byte = L.byte_at(0);
    uint4 v = L.p(Chain::N - 1, 0);
    for (unsigned k = 0; k < L.nb; ++k) {
      const unsigned kn = L.next(k);
      const unsigned byten = L.byte_at(kn);
      const uint4 vn = L.p(Chain::N - 1, kn);
      if (multi) {
        while (seg + 1 < L.nseg && L.k0 + k == seg_end) {      // (empty segments: several boundaries at one byte)
          encode(1, 0);
          segs[seg].out_end = n;
          ++seg;
          seg_end = segs[seg].in_end;
        }
      }
      // (the 8 probabilities first: their table lookups do not depend on the coder's state, and behind the data-dependent
      //  loops of encode() the compiler would leave each where it is used)
      unsigned prs[8];
#pragma unroll
      for (int B = 0; B < 8; ++B) prs[B] = (unsigned)squash(sp_clamp2k(pipe_p_get(v, B))) * 2u + 1u;
      encode(0, 0);
#pragma unroll
      for (int B = 0; B < 8; ++B) {
        encode(pipe_y(byte, B), prs[B]);
      }
      if (n - nf >= 4u) flush4();
      byte = byten; v = vn;
    }
  }
  if ((unsigned)L.chunk == nchunks - 1u) {
    int status = (int)(unsigned)L.state(Chain::HCOMP_STATE + 4);
    if (multi && !status) {
      while (seg + 1 < L.nseg) { encode(1, 0); segs[seg].out_end = n; ++seg; }     // trailing empty segments
    }
    if (!status) encode(1, 0);
    flush_all();
    if (multi) segs[seg].out_end = n;
    if (!status && n > L.out_cap) status = 3;
    BlockResult r;
    r.out_len = n; r.consumed = L.len; r.status = status; r.steps = 8u * L.len;
    a.res[L.rslot] = r;
  } else {
    flush_all();
    L.state(sw + 0) = low; L.state(sw + 1) = high; L.state(sw + 2) = n; L.state(sw + 3) = seg;
  }
}

// CODER of the latency shape's persistent launch (round 6).  A chain of a few components on a batch that leaves the machine
// empty goes at the pace of its longest per-bit instruction stream, and that was this unit's: ~50 executed instructions per bit
// (profiles/r06_results.md section 5: a 64-bit multiply, the closed form's case analysis, the 64-bit output window, a flush test).
// Here a bit is ~27:
//   * p comes from a private LDS table of 4096 words that already hold (squash(p) * 2 + 1) << 16, so that
//     (high - low) * p >> 16 is ONE v_mul_hi_u32;
//   * the shift-out is done for the case that covers all but ~2^-16 of the bits -- high and low differ somewhere and the
//     low 16 bits of low are not all zero: then k = clz(high ^ low) / 8 <= 3 turns of the reference's loop leave
//     high << 8k | ones and max(low << 8k, 1) (only the LAST of three turns can find low << 8 == 0 when bits 0 .. 15 are not
//     all zero) -- and every bit keeps min(high ^ low, low & 0xFFFF) in an accumulator: if it is 0 after a byte's 9 codes,
//     the lane takes the byte again from the saved state with the reference's loop (tests/cpp/coder_norm_check.c checks the
//     fast form against the loop on every state that passes the test);
//   * output: the coded bytes of ONE input byte collect in a 32-bit window (window << 8k | top k bytes of high: no test) and
//     leave with ONE unaligned 4-byte store per input byte; the bytes of the store that are not yet final are overwritten by
//     the next (a lane's stores to one address arrive in program order).  A byte that codes into more than 4 bytes (a model
//     badly wrong nine times in a row) takes the careful way as well, and so does a lane within 40 bytes of its capacity
//     (byte stores, each tested).  What lies between out_len and out_cap afterwards is undefined, as the C ABI says.
//     (The first version stored the 4 top bytes of high once per BIT: the fastest coder alone, 386 -> 271 ms on configs[1], but
//     9 partial stores per input byte and block into one line slowed every other unit of the launch -- mid.cfg 110 -> 92 MB/s,
//     -m5 on 256 blocks 142 -> 106; profiles/r06 calls 15, 16.)
// pt: 4096 words of LDS owned by this wavefront; load_tab: fill it (first chunk).  PD: how many bytes ahead the stream is read.
template <class Chain, int PD>
__device__ __forceinline__ void pipe_coder_fast(PipeLane<Chain>& L, const PipeArgs& a, unsigned* pt, int lane, bool load_tab) {
  constexpr int sw = Chain::CODER_STATE;
  if (load_tab) {
    for (int i = lane; i < 4096; i += 64) pt[i] = (((unsigned)a.tb->squash[i] * 2u) + 1u) << 16;
    (void)pipe_any(true);          // (every lane reads what every lane wrote: the lanes meet here -- the emulator runs them one by one)
  }
  const unsigned nchunks = max((L.len + (unsigned)Chain::PIPE_C - 1u) / (unsigned)Chain::PIPE_C, 1u);
  const bool active = L.live && L.chunk >= 0 && (unsigned)L.chunk < nchunks;
  if (!active) return;
  unsigned low = 1, high = 0xFFFFFFFFu, n = 0;
  unsigned seg = 0, seg_end = 0xFFFFFFFFu;
  const bool multi = L.nseg > 1;
  if (L.chunk > 0) { low = L.state(sw + 0); high = L.state(sw + 1); n = L.state(sw + 2); seg = L.state(sw + 3); }
  typedef __attribute__((address_space(1))) SegRange g_seg;
  g_seg* const segs = (g_seg*)L.segs;
  if (multi) seg_end = segs[seg].in_end;
  typedef unsigned __attribute__((aligned(1))) u32u;
  typedef __attribute__((address_space(1))) u32u g_u32u;
  // Encoder::encode (libzpaq.cpp:2402-2416) as the reference writes it; p16 = the 16-bit probability
  auto encode_loop = [&](int y, unsigned p16) __attribute__((always_inline)) {
    const unsigned mid = low + (unsigned)(((unsigned long long)(high - low) * p16) >> 16);
    if (y) high = mid; else low = mid + 1u;
    while (((high ^ low) & 0xFF000000u) == 0u) {
      if (n < L.out_cap) L.out[n] = (unsigned char)(high >> 24);
      ++n;
      high = high << 8 | 255u;
      low <<= 8;
      low += (low == 0u);
    }
  };
  unsigned acc = 0xFFFFFFFFu, ob = 0u;
  // the common case of the same (see above)
  auto shift_out = [&](unsigned high1, unsigned low1) __attribute__((always_inline)) {
    const unsigned x = high1 ^ low1;
    acc = min(acc, min(x, low1 & 0xFFFFu));
    const unsigned sh = (unsigned)__builtin_clz(x | 1u) & 24u;
#ifdef ZPQ_EMU
    const unsigned top = sh ? high1 >> (32u - sh) : 0u;
#else
    const unsigned top = __builtin_amdgcn_ubfe(high1, 0u - sh, sh);      // (offset 32 - sh mod 32, width sh: 0 when sh = 0)
#endif
    ob = (ob << sh) | top;
    n += sh >> 3;
    high = (high1 << sh) | ((1u << sh) - 1u);
    low = max(low1 << sh, 1u);
  };
  auto encode_fast = [&](bool y, unsigned P) __attribute__((always_inline)) {            // P = p16 << 16
    const unsigned mid = low + (unsigned)(((unsigned long long)(high - low) * P) >> 32);
    shift_out(y ? mid : high, y ? low : mid + 1u);
  };
  if (L.nb) {
    const unsigned last = L.nb - 1u;
    constexpr int U = PD + 1;            // (ring of U slots, byte k in slot k mod U, unrolled U times: see pipe_icm_unit)
    static_assert(Chain::PIPE_C % U == 0, "ring of U slots");
    unsigned bq[U];
    uint4 vq[U];
#pragma unroll
    for (int sl = 0; sl < U; ++sl) { const unsigned kd = min((unsigned)sl, last); bq[sl] = L.byte_at(kd); vq[sl] = L.p(Chain::N - 1, kd); }
    for (unsigned kb = 0; pipe_any(kb < L.nb); kb += (unsigned)U) {
#pragma unroll
      for (int sl = 0; sl < U; ++sl) {
        const unsigned k = kb + (unsigned)sl;
        {
          // (no lane leaves the loop early: see pipe_icm_unit.  A lane past its last byte codes nothing: it takes neither way)
          const bool on = k < L.nb;
          const unsigned byte = bq[sl];
          const uint4 v = vq[sl];
          if (multi && on) {
            while (seg + 1 < L.nseg && L.k0 + k == seg_end) {      // (empty segments: several boundaries at one byte)
              encode_loop(1, 0);
              segs[seg].out_end = n;
              ++seg;
              seg_end = segs[seg].in_end;
            }
          }
          unsigned P[8];
#pragma unroll
          for (int B = 0; B < 8; ++B) P[B] = pt[(unsigned)(sp_clamp2k(pipe_p_get(v, B)) + 2048)];
          bool yb[8];                        // (the byte's bits as lane masks, all eight before the first is used)
#pragma unroll
          for (int B = 0; B < 8; ++B) yb[B] = pipe_y(byte, B) != 0;
          const unsigned low0 = low, high0 = high, n0 = n;
          const bool room = n + 40u <= L.out_cap;
          acc = 0xFFFFFFFFu;
          ob = 0u;
          shift_out(high, low + 1u);                               // encode(0, 0): mid = low
#pragma unroll
          for (int B = 0; B < 8; ++B) encode_fast(yb[B], P[B]);
          const unsigned cnt = n - n0;                             // coded bytes of this input byte: the low cnt bytes of ob, the oldest on top
          const bool fine = room && acc != 0u && cnt <= 4u;
          if (fine && on) *(g_u32u*)(L.out + n0) = __builtin_bswap32(ob << ((32u - 8u * cnt) & 31u));     // (cnt = 0: four bytes the next store overwrites)
          if (!(fine && on)) { low = low0; high = high0; n = n0; }
          if (pipe_any(on && !fine)) {
            if (on && !fine) {
              encode_loop(0, 0);
              for (int B = 0; B < 8; ++B) encode_loop(pipe_y(byte, B), P[B] >> 16);
            }
          }
          const unsigned kf = min(k + (unsigned)U, last);
          bq[sl] = L.byte_at(kf); vq[sl] = L.p(Chain::N - 1, kf);
        }
      }
    }
  }
  if ((unsigned)L.chunk == nchunks - 1u) {
    int status = (int)(unsigned)L.state(Chain::HCOMP_STATE + 4);
    if (multi && !status) {
      while (seg + 1 < L.nseg) { encode_loop(1, 0); segs[seg].out_end = n; ++seg; }     // trailing empty segments
    }
    if (!status) encode_loop(1, 0);
    if (multi) segs[seg].out_end = n;
    if (!status && n > L.out_cap) status = 3;
    BlockResult r;
    r.out_len = n; r.consumed = L.len; r.status = status; r.steps = 8u * L.len;
    a.res[L.rslot] = r;
  } else {
    L.state(sw + 0) = low; L.state(sw + 1) = high; L.state(sw + 2) = n; L.state(sw + 3) = seg;
  }
}

// =====================================================================================================
// Rows kernel: the ROW unit of every ICM / ISSE (1 KB of LDS: the state table).  One wavefront per (unit, group).
template <class Chain>
__device__ __forceinline__ void pipe_rows_body(const PipeArgs& a) {
  __shared__ unsigned char ns[1024];
  const int lane = threadIdx.x & 63;
  const unsigned ngroups = (a.nblocks + Chain::PIPE_G - 1u) / Chain::PIPE_G;
  const unsigned wg = blockIdx.x + a.wg0;
  const unsigned role = wg / ngroups, g = wg % ngroups;
  for (int i = lane; i < 256; i += (int)blockDim.x) ((unsigned*)ns)[i] = ((const unsigned*)a.tb->ns)[i];
  __syncthreads();
  static_for<0, Chain::NROWU>([&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    if (role != (unsigned)r) return;
    PipeLane<Chain> L;
    L.open(a, g * Chain::PIPE_G + (unsigned)lane, 1);
    if (L.chunk < 0 || !pipe_any(L.nb > 0)) return;
    pipe_row<Chain, Chain::ROW_COMP[r]>(L, ns);
  });
}

// Light kernel: every other unit that needs no big LDS table (CONS, CM, MATCH, AVG, MIX2, SSE, CODER); the small
// shared tables (dt, dt2k, squash, compact stretch: 16 KB) are loaded by the units that use them.
template <class Chain>
__device__ __forceinline__ void pipe_light_body(const PipeArgs& a) {
  __shared__ int dt[1024];
  __shared__ unsigned short dt2k[256];
  __shared__ PipeSquash squash;
  __shared__ PipeStretch stretch;
  const int lane = threadIdx.x & 63;
  const unsigned ngroups = (a.nblocks + Chain::PIPE_G - 1u) / Chain::PIPE_G;
  const unsigned wg = blockIdx.x + a.wg0;
  const unsigned role = wg / ngroups, g = wg % ngroups;
  static_for<0, Chain::NLIGHT>([&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    if (role != (unsigned)r) return;
    constexpr int kind = Chain::LIGHT_KIND[r], I = Chain::LIGHT_COMP[r];
    constexpr int level = kind == PK_CODER ? Chain::CODER_LEVEL : Chain::P_LEVEL[I];
    PipeLane<Chain> L;
    if constexpr (kind >= PK_CM_BITS) {
      // workgroup LIGHT_SUB[r] of the unit's: LIGHT_THREADS / 8 blocks, lane = (block, bit position)
      constexpr unsigned BPW = (unsigned)Chain::LIGHT_THREADS / 8u;
      L.open(a, g * Chain::PIPE_G + (unsigned)Chain::LIGHT_SUB[r] * BPW + ((unsigned)lane >> 3), level);
      if (L.chunk < 0 || !pipe_any(L.nb > 0)) return;
      const unsigned B = (unsigned)lane & 7u;
      if constexpr (kind == PK_CM_BITS) {
        for (int i = lane; i < 1024; i += (int)blockDim.x) dt[i] = a.tb->dt[i];
        stretch.load(a.tb, lane);
        __syncthreads();
        pipe_cm_bits<Chain, I>(L, B, stretch, dt);
      } else if constexpr (kind == PK_MIX2_BITS) {
        squash.load(a.tb, lane);
        __syncthreads();
        pipe_mix2_bits<Chain, I>(L, B, squash);
      } else {
        for (int i = lane; i < 1024; i += (int)blockDim.x) dt[i] = a.tb->dt[i];
        stretch.load(a.tb, lane);
        __syncthreads();
        pipe_sse_bits<Chain, I>(L, B, stretch, dt);
      }
      return;
    }
    L.open(a, g * Chain::PIPE_G + (unsigned)lane, level);
    if constexpr ((unsigned)Chain::LIGHT_THREADS > Chain::PIPE_G) {
      // (workgroups sized for the bit-lane units: a lane-per-block unit uses the first PIPE_G lanes)
      if ((unsigned)lane >= Chain::PIPE_G) { L.live = false; L.nb = 0; L.len = 0; }
    }
    if (L.chunk < 0) return;
    if constexpr (kind != PK_CODER) { if (!pipe_any(L.nb > 0)) return; }
    if constexpr (kind == PK_CONS) {
      pipe_cons<Chain, I>(L);
    } else if constexpr (kind == PK_CM) {
      for (int i = lane; i < 1024; i += (int)blockDim.x) dt[i] = a.tb->dt[i];
      stretch.load(a.tb, lane);
      __syncthreads();
      pipe_cm<Chain, I>(L, stretch, dt);
    } else if constexpr (kind == PK_MATCH) {
      for (int i = lane; i < 256; i += (int)blockDim.x) dt2k[i] = (unsigned short)a.tb->dt2k[i];
      stretch.load(a.tb, lane);
      __syncthreads();
      pipe_match_any<Chain, I>(L, stretch, dt2k);
    } else if constexpr (kind == PK_AVG) {
      pipe_avg<Chain, I>(L);
    } else if constexpr (kind == PK_MIX2) {
      squash.load(a.tb, lane);
      __syncthreads();
      pipe_mix2<Chain, I>(L, squash);
    } else if constexpr (kind == PK_SSE) {
      for (int i = lane; i < 1024; i += (int)blockDim.x) dt[i] = a.tb->dt[i];
      stretch.load(a.tb, lane);
      __syncthreads();
      pipe_sse<Chain, I>(L, stretch, dt);
    } else if constexpr (kind == PK_CODER) {
      squash.load(a.tb, lane);
      __syncthreads();
      pipe_coder<Chain>(L, a, squash);
    }
  });
}

// the 8 bit histories of a byte out of a bh stream element
__device__ __forceinline__ unsigned pipe_bh_get(const uint2& w, int B) { return ((B < 4 ? w.x : w.y) >> (8 * (B & 3))) & 255u; }

// =====================================================================================================
// ICM map (libzpaq.cpp:1875-1881, 1973-1977): side table cm[256] of 64 blocks in LDS as [entry][lane].
// The entry of the NEXT bit is read before this bit's entry is written and patched when they coincide, so the
// LDS round trip is off the lane's serial chain.
// one chunk of the ICM map of component I; tab = [256][G] words of LDS; load_tab / store_tab: the side table is staged from /
// written back to the arena around this chunk (the persistent launch keeps it in LDS from chunk to chunk)
// PD: the streams are read PD + 1 bytes ahead (0: as the step kernels do.  A lone wavefront of the persistent launch stores its
// output stream write-through and vmcnt counts in order, so a load issued right behind a byte's store -- the next byte's
// element, one byte ahead -- cannot be waited for without waiting for that store's acknowledgement; two more bytes of lead and
// the wait only covers stores that are two bytes old)
template <class Chain, int I, class ST, int PD = 0>
__device__ __forceinline__ void pipe_icm_unit(PipeLane<Chain>& L, unsigned* tab, const ST& stretch, int lane, bool load_tab, bool store_tab) {
  constexpr unsigned G = Chain::PIPE_G;
  constexpr CompK c = Chain::comp[I];
  constexpr int ri = Chain::P_ROW[I];
  if (!L.nb) return;
  if (load_tab)
    for (int e = 0; e < 256; e += 4) {
      const uint4 q = L.A128((unsigned)c.t0 + 4u * e);
      tab[e * G + lane] = q.x; tab[(e + 1) * G + lane] = q.y; tab[(e + 2) * G + lane] = q.z; tab[(e + 3) * G + lane] = q.w;
    }
  const unsigned last = L.nb - 1u;
  unsigned s, v;
  // the 8 bits of one byte: its bit histories w, the next byte's wn (bit 7 reads ahead), the p element out
  auto do_byte = [&](unsigned k, unsigned byte, const uint2& w, const uint2& wn) __attribute__((always_inline)) {
    PipeP8 out;
#pragma unroll
    for (int B = 0; B < 8; ++B) {
      const unsigned sn = B < 7 ? pipe_bh_get(w, B + 1) : pipe_bh_get(wn, 0);
      const unsigned vn = tab[sn * G + lane];
      out.set(B, stretch(v >> 8));
      const unsigned nv = v + (unsigned)((int)((unsigned)(pipe_y(byte, B) * 32767) - (v >> 8)) >> 2);
      tab[s * G + lane] = nv;
      v = sn == s ? nv : vn;
      s = sn;
    }
    L.put_p(I, k, out.get());
  };
  if constexpr (PD > 0) {
    // ring of U slots, byte k in slot k mod U, the loop unrolled U times: every slot a fixed set of registers (a value handed
    // from register to register would be waited for on the spot)
    constexpr int U = PD + 1;
    unsigned bq[U];
    uint2 wq[U];
#pragma unroll
    for (int sl = 0; sl < U; ++sl) { const unsigned kd = min((unsigned)sl, last); bq[sl] = L.byte_at(kd); wq[sl] = L.bh(ri, kd); }
    s = pipe_bh_get(wq[0], 0);
    v = tab[s * G + lane];
    // (no lane leaves the loop early -- the compiler's vmcnt bookkeeping gives up at a divergent branch --: a lane whose block
    //  ends inside the chunk goes on over its last byte's elements, its table column and stream positions are dead by then;
    //  a chunk's length is a multiple of U, so a block that goes on never overruns)
    static_assert(Chain::PIPE_C % U == 0, "ring of U slots");
    for (unsigned kb = 0; pipe_any(kb < L.nb); kb += (unsigned)U) {
#pragma unroll
      for (int sl = 0; sl < U; ++sl) {
        const unsigned k = kb + (unsigned)sl;
        do_byte(k, bq[sl], wq[sl], wq[(sl + 1) % U]);
        const unsigned kf = min(k + (unsigned)U, last);
        bq[sl] = L.byte_at(kf); wq[sl] = L.bh(ri, kf);
      }
    }
  } else {
    unsigned byte = L.byte_at(0);
    uint2 w = L.bh(ri, 0);
    s = pipe_bh_get(w, 0);
    v = tab[s * G + lane];
    for (unsigned k = 0; k < L.nb; ++k) {
      const unsigned kn = L.next(k);
      const unsigned byten = L.byte_at(kn);
      const uint2 wn = L.bh(ri, kn);
      do_byte(k, byte, w, wn);
      byte = byten; w = wn;
    }
  }
  if (store_tab)
    for (int e = 0; e < 256; e += 4)
      L.A128((unsigned)c.t0 + 4u * e) = make_uint4(tab[e * G + lane], tab[(e + 1) * G + lane], tab[(e + 2) * G + lane], tab[(e + 3) * G + lane]);
}

template <class Chain>
__device__ __forceinline__ void pipe_icm_body(const PipeArgs& a) {
  constexpr unsigned G = Chain::PIPE_G;
  __shared__ unsigned tab[256 * G];
  __shared__ PipeStretch stretch;
  const int lane = threadIdx.x & 63;
  const unsigned ngroups = (a.nblocks + Chain::PIPE_G - 1u) / Chain::PIPE_G;
  const unsigned wg = blockIdx.x + a.wg0;
  const unsigned role = wg / ngroups, g = wg % ngroups;
  static_for<0, Chain::NICM>([&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    if (role != (unsigned)r) return;
    constexpr int I = Chain::ICM_COMP[r];
    PipeLane<Chain> L;
    L.open(a, g * Chain::PIPE_G + (unsigned)lane, Chain::P_LEVEL[I]);
    if (L.chunk < 0 || !pipe_any(L.nb > 0)) return;
    stretch.load(a.tb, lane);
    __syncthreads();
    pipe_icm_unit<Chain, I>(L, tab, stretch, lane, true, true);
  });
}

// ISSE map inside the persistent launch: the weight pairs PACKED -- both weights are clamped to +-2^19 (libzpaq.cpp:2031-2039),
// 40 bits per pair: a word [entry][lane] with w0 and the low 12 bits of w1, and w1's high 8 bits in a byte of the word
// [entry / 4][lane] -- 40 KiB instead of 64 for a group of 32 blocks, which is what decides how many workgroups a group needs.
template <class Chain, int I, class SQ, int PD = 0>
__device__ __forceinline__ void pipe_isse_packed_unit(PipeLane<Chain>& L, unsigned* tab, const SQ& squash, int lane, bool load_tab) {
  constexpr unsigned G = Chain::PIPE_G;
  constexpr CompK c = Chain::comp[I];
  constexpr int ri = Chain::P_ROW[I], J = (int)c.a2;
  if (!L.nb) return;
  unsigned* const lo = tab;
  unsigned char* const hi = (unsigned char*)(tab + 256u * G);
  auto hi_at = [&](unsigned s) __attribute__((always_inline)) -> unsigned { return (((s >> 2) * G + (unsigned)lane) << 2) + (s & 3u); };
  auto w0_of = [&](unsigned l) __attribute__((always_inline)) -> int { return (int)(l << 12) >> 12; };
  auto w1_of = [&](unsigned l, unsigned h) __attribute__((always_inline)) -> int { return (int)(((l >> 20) | (h << 12)) << 12) >> 12; };
  if (load_tab)
    for (unsigned e = 0; e < 256u; e += 2) {
      const uint4 q = L.A128((unsigned)c.t0 + 8u * e);                 // (w0, w1) of entries e, e + 1
      lo[e * G + lane] = (q.x & 0xFFFFFu) | (q.y << 20);
      hi[hi_at(e)] = (unsigned char)(q.y >> 12);
      lo[(e + 1u) * G + lane] = (q.z & 0xFFFFFu) | (q.w << 20);
      hi[hi_at(e + 1u)] = (unsigned char)(q.w >> 12);
    }
  const unsigned last = L.nb - 1u;
  unsigned s;
  int w0, w1;
  auto do_byte = [&](unsigned k, unsigned byte, const uint2& w, const uint2& wn, const uint4& vj) __attribute__((always_inline)) {
    PipeP8 out;
#pragma unroll
    for (int B = 0; B < 8; ++B) {
      const unsigned sn = B < 7 ? pipe_bh_get(w, B + 1) : pipe_bh_get(wn, 0);
      const unsigned ln = lo[sn * G + lane], hn = hi[hi_at(sn)];
      const int pj = pipe_p_get(vj, B);
      const int pr = sp_clamp2k((__mul24(w0, pj) + w1 * 64) >> 16);          // 20-bit x 12-bit
      out.set(B, pr);
      const int err = pipe_y(byte, B) * 32767 - squash(pr);
      const int u0 = sp_clamp512k(w0 + ((__mul24(err, pj) + (1 << 12)) >> 13));
      const int u1 = sp_clamp512k(w1 + ((err + 16) >> 5));
      lo[s * G + lane] = ((unsigned)u0 & 0xFFFFFu) | ((unsigned)u1 << 20);
      hi[hi_at(s)] = (unsigned char)((unsigned)u1 >> 12);
      w0 = sn == s ? u0 : w0_of(ln);
      w1 = sn == s ? u1 : w1_of(ln, hn);
      s = sn;
    }
    L.put_p(I, k, out.get());
  };
  if constexpr (PD > 0) {
    constexpr int U = PD + 1;            // (ring of U slots, unrolled U times, no lane leaving the loop early: see pipe_icm_unit)
    static_assert(Chain::PIPE_C % U == 0, "ring of U slots");
    unsigned bq[U];
    uint2 wq[U];
    uint4 vq[U];
#pragma unroll
    for (int sl = 0; sl < U; ++sl) { const unsigned kd = min((unsigned)sl, last); bq[sl] = L.byte_at(kd); wq[sl] = L.bh(ri, kd); vq[sl] = L.p(J, kd); }
    s = pipe_bh_get(wq[0], 0);
    { const unsigned l0 = lo[s * G + lane], h0 = hi[hi_at(s)]; w0 = w0_of(l0); w1 = w1_of(l0, h0); }
    for (unsigned kb = 0; pipe_any(kb < L.nb); kb += (unsigned)U) {
#pragma unroll
      for (int sl = 0; sl < U; ++sl) {
        const unsigned k = kb + (unsigned)sl;
        do_byte(k, bq[sl], wq[sl], wq[(sl + 1) % U], vq[sl]);
        const unsigned kf = min(k + (unsigned)U, last);
        bq[sl] = L.byte_at(kf); wq[sl] = L.bh(ri, kf); vq[sl] = L.p(J, kf);
      }
    }
  } else {
    unsigned byte = L.byte_at(0);
    uint2 w = L.bh(ri, 0);
    uint4 vj = L.p(J, 0);
    s = pipe_bh_get(w, 0);
    { const unsigned l0 = lo[s * G + lane], h0 = hi[hi_at(s)]; w0 = w0_of(l0); w1 = w1_of(l0, h0); }
    for (unsigned k = 0; k < L.nb; ++k) {
      const unsigned kn = L.next(k);
      const unsigned byten = L.byte_at(kn);
      const uint2 wn = L.bh(ri, kn);
      const uint4 vjn = L.p(J, kn);
      do_byte(k, byte, w, wn, vj);
      byte = byten; w = wn; vj = vjn;
    }
  }
}

// ISSE map (libzpaq.cpp:1923-1931, 2031-2039): weight pairs of 64 blocks in LDS as [2 entry + w][lane].
template <class Chain, int I, class SQ, int PD = 0>
__device__ __forceinline__ void pipe_isse_unit(PipeLane<Chain>& L, unsigned* tab, const SQ& squash, int lane, bool load_tab, bool store_tab) {
  constexpr unsigned G = Chain::PIPE_G;
  constexpr CompK c = Chain::comp[I];
  constexpr int ri = Chain::P_ROW[I], J = (int)c.a2;
  if (!L.nb) return;
  if (load_tab)
    for (int e = 0; e < 512; e += 4) {
      const uint4 q = L.A128((unsigned)c.t0 + 4u * e);
      tab[e * G + lane] = q.x; tab[(e + 1) * G + lane] = q.y; tab[(e + 2) * G + lane] = q.z; tab[(e + 3) * G + lane] = q.w;
    }
  const unsigned last = L.nb - 1u;
  unsigned s;
  int w0, w1;
  auto do_byte = [&](unsigned k, unsigned byte, const uint2& w, const uint2& wn, const uint4& vj) __attribute__((always_inline)) {
    PipeP8 out;
#pragma unroll
    for (int B = 0; B < 8; ++B) {
      const unsigned sn = B < 7 ? pipe_bh_get(w, B + 1) : pipe_bh_get(wn, 0);
      const int n0 = (int)tab[(2u * sn) * G + lane], n1 = (int)tab[(2u * sn + 1u) * G + lane];
      const int pj = pipe_p_get(vj, B);
      const int pr = sp_clamp2k((__mul24(w0, pj) + w1 * 64) >> 16);          // 20-bit x 12-bit
      out.set(B, pr);
      const int err = pipe_y(byte, B) * 32767 - squash(pr);
      const int u0 = sp_clamp512k(w0 + ((__mul24(err, pj) + (1 << 12)) >> 13));
      const int u1 = sp_clamp512k(w1 + ((err + 16) >> 5));
      tab[(2u * s) * G + lane] = (unsigned)u0;
      tab[(2u * s + 1u) * G + lane] = (unsigned)u1;
      w0 = sn == s ? u0 : n0;
      w1 = sn == s ? u1 : n1;
      s = sn;
    }
    L.put_p(I, k, out.get());
  };
  if constexpr (PD > 0) {
    constexpr int U = PD + 1;            // (ring of U slots, unrolled U times: see pipe_icm_unit)
    unsigned bq[U];
    uint2 wq[U];
    uint4 vq[U];
#pragma unroll
    for (int sl = 0; sl < U; ++sl) { const unsigned kd = min((unsigned)sl, last); bq[sl] = L.byte_at(kd); wq[sl] = L.bh(ri, kd); vq[sl] = L.p(J, kd); }
    s = pipe_bh_get(wq[0], 0);
    w0 = (int)tab[(2u * s) * G + lane]; w1 = (int)tab[(2u * s + 1u) * G + lane];
    static_assert(Chain::PIPE_C % U == 0, "ring of U slots");          // (no lane leaves the loop early: see pipe_icm_unit)
    for (unsigned kb = 0; pipe_any(kb < L.nb); kb += (unsigned)U) {
#pragma unroll
      for (int sl = 0; sl < U; ++sl) {
        const unsigned k = kb + (unsigned)sl;
        do_byte(k, bq[sl], wq[sl], wq[(sl + 1) % U], vq[sl]);
        const unsigned kf = min(k + (unsigned)U, last);
        bq[sl] = L.byte_at(kf); wq[sl] = L.bh(ri, kf); vq[sl] = L.p(J, kf);
      }
    }
  } else {
    unsigned byte = L.byte_at(0);
    uint2 w = L.bh(ri, 0);
    uint4 vj = L.p(J, 0);
    s = pipe_bh_get(w, 0);
    w0 = (int)tab[(2u * s) * G + lane]; w1 = (int)tab[(2u * s + 1u) * G + lane];
    for (unsigned k = 0; k < L.nb; ++k) {
      const unsigned kn = L.next(k);
      const unsigned byten = L.byte_at(kn);
      const uint2 wn = L.bh(ri, kn);
      const uint4 vjn = L.p(J, kn);
      do_byte(k, byte, w, wn, vj);
      byte = byten; w = wn; vj = vjn;
    }
  }
  if (store_tab)
    for (int e = 0; e < 512; e += 4)
      L.A128((unsigned)c.t0 + 4u * e) = make_uint4(tab[e * G + lane], tab[(e + 1) * G + lane], tab[(e + 2) * G + lane], tab[(e + 3) * G + lane]);
}

template <class Chain>
__device__ __forceinline__ void pipe_isse_body(const PipeArgs& a) {
  constexpr unsigned G = Chain::PIPE_G;
  __shared__ unsigned tab[512 * G];
  __shared__ PipeSquash squash;
  const int lane = threadIdx.x & 63;
  const unsigned ngroups = (a.nblocks + Chain::PIPE_G - 1u) / Chain::PIPE_G;
  const unsigned wg = blockIdx.x + a.wg0;
  const unsigned role = wg / ngroups, g = wg % ngroups;
  static_for<0, Chain::NISSE>([&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    if (role != (unsigned)r) return;
    constexpr int I = Chain::ISSE_COMP[r];
    PipeLane<Chain> L;
    L.open(a, g * Chain::PIPE_G + (unsigned)lane, Chain::P_LEVEL[I]);
    if (L.chunk < 0 || !pipe_any(L.nb > 0)) return;
    squash.load(a.tb, lane);
    __syncthreads();
    pipe_isse_unit<Chain, I>(L, tab, squash, lane, true, true);
  });
}

// =====================================================================================================
// MIX (libzpaq.cpp:1910-1921, 2023-2029): QL lanes per block, lane q owning weights 4q .. 4q+3 of the selected
// row, so a row travels as ONE 16-byte access per lane (a row of m weights = ceil(m/4) lanes, rows are only 4-byte
// aligned).  The dot product is reduced inside the lane group by DPP (quad_perm, row_half_mirror, row_mirror),
// which leaves the sum in EVERY lane of the group, so each lane computes the error itself and trains its own
// weights.
typedef uint4 __attribute__((aligned(4))) pipe_u128a4;
typedef __attribute__((address_space(1))) pipe_u128a4 g_u128a4;

template <int QL>
__device__ __forceinline__ int pipe_group_sum(int v) {
  if constexpr (QL >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  if constexpr (QL >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  if constexpr (QL >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
  if constexpr (QL >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror
  return v;
}

// MIX with a lane per (block, BIT POSITION, weight quad) -- latency mode (chains with few blocks in the batch: the MI355X
// measurements in profiles/r03 put the crossover between 256 and 512 blocks).  A MIX whose context mask keeps the whole partial byte
// (c.a5 == 255, at least 256 rows) selects a different weight row for each of a byte's 8 bits -- c8 is part of the row
// index -- and training touches the selected row only, so with all bits known the 8 bits of a byte are independent:
// 8 x QL lanes work on one byte at a time, the per-byte chain is one bit's ~70 instructions instead of eight bits' ~1000,
// and the unit runs 8 x the wavefronts.  All bit positions of a block sit in ONE wavefront (workgroups of 64 threads,
// QL <= 8), so two bytes whose rows coincide through DIFFERENT contexts are ordered by the wavefront's instruction
// order.  Rows, inputs and contexts are fetched MIX_DEPTH bytes ahead; a row rewritten since its fetch is taken from
// the lane's own history (same context: same bit position, same lane) or fetched again (contexts less than a byte's
// row range apart: any lane may have written it).
// one chunk of MIX role r with a lane per (block, bit position, weight quad): q = the lane's quad, B = its bit position
template <class Chain, int r, class SQ>
__device__ __forceinline__ void pipe_mix_bits_unit(PipeLane<Chain>& L, unsigned q, unsigned B, const SQ& squash) {
  constexpr int I = Chain::MIX_COMP[r], QL = Chain::MIX_QL[r];
  constexpr CompK c = Chain::comp[I];
  constexpr int m = (int)c.a3, J = (int)c.a2, ci = Chain::P_CTX[I];
  constexpr int NQ = (m + 3) / 4, TAIL = m % 4, D = Chain::MIX_DEPTH, HN = D;
  static_assert(NQ <= QL && c.a5 == 255u && c.mask0 >= 255u, "MIX bit lanes need the 8 rows of a byte to be distinct");
  if (!L.nb) return;
  // A row padded to at least 4 words per lane of the group (layout.h mix_row_stride: every m but 2) is loaded and stored
  // as whole quads by EVERY lane: a quad past the row's end lies in the padding, its inputs read as 0, its words go back as
  // they came -- no lane is masked, the 8 bits of a byte are straight-line code and the scheduler interleaves them.
  constexpr bool FULL = c.stride >= 4u * (unsigned)QL;
  const bool act = FULL || q < (unsigned)NQ;                       // lanes that hold weights
  const bool tail = !FULL && TAIL != 0 && q == (unsigned)(NQ - 1); // the lane whose quad is cut short by the row's end
  const unsigned qoff = 16u * (act ? q : 0u);
  bool have[4];
  int tin[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const int t = (int)q * 4 + x;
    have[x] = t < m;
    tin[x] = J + (have[x] ? t : 0);
  }
  auto row_of = [&](unsigned hh, unsigned bytev) __attribute__((always_inline)) -> unsigned {
    const unsigned c8 = (1u << B) | (bytev >> (8u - B));           // pipe_c8 for a lane's own position
    return (unsigned)c.t0 + 4u * __umul24((hh + c8) & c.mask0, (unsigned)c.stride) + qoff;
  };
  auto input = [&](int x, unsigned kk) __attribute__((always_inline)) -> int {      // this position's half-word of the stream element
    return (int)*(const g_i16*)((const g_u8*)&L.p(tin[x], kk) + 2u * B);
  };
  const unsigned last = L.nb - 1u;
  // Ring of W = 2 D slots, slot = byte index mod W, every slot a fixed set of registers (the loop is unrolled W times):
  // a byte's context and value are fetched W bytes ahead, its row address / weights / inputs D bytes ahead.  A register is
  // written by one fetch and read D or more bytes later -- nothing is copied while a fetch is in flight.
  constexpr int W = 2 * D;
  unsigned hx[W], bx[W];                  // context, input byte
  unsigned rq[W];                         // row address of this lane's quad
  uint4 wq[W];                            // weights as fetched
  int pq[W][4];                           // inputs as fetched (masked when used)
  // the last D bytes done: row, context, weights as stored
  unsigned hr[HN], hh[HN];
  uint4 hw[HN];
  auto near = [&](int sl, unsigned kk) __attribute__((always_inline)) {
    rq[sl] = row_of(hx[sl], bx[sl]);
    wq[sl] = *(g_u128a4*)(L.arena + rq[sl]);
#pragma unroll
    for (int x = 0; x < 4; ++x) pq[sl][x] = input(x, kk);
  };
#pragma unroll
  for (int sl = 0; sl < W; ++sl) {
    const unsigned kk = min((unsigned)sl, last);
    hx[sl] = L.ctx(ci, kk);
    bx[sl] = L.byte_at(kk);
  }
#pragma unroll
  for (int sl = 0; sl < D; ++sl) near(sl, min((unsigned)sl, last));
#pragma unroll
  for (int i = 0; i < HN; ++i) { hr[i] = 0xFFFFFFFFu; hh[i] = hx[0]; hw[i] = make_uint4(0u, 0u, 0u, 0u); }
  for (unsigned kb = 0; pipe_any(kb < L.nb); kb += (unsigned)W) {
#pragma unroll
    for (int sl = 0; sl < W; ++sl) {
      const unsigned k = kb + (unsigned)sl;
      const bool on = k < L.nb;
      const unsigned hcur = hx[sl], bcur = bx[sl], row = rq[sl];
      uint4 w = wq[sl];
      {
        bool late = false;
#pragma unroll
        for (int i = HN - 1; i >= 0; --i) {                       // oldest first: the most recent store wins
          const bool fw = row == hr[i];
          w.x = fw ? hw[i].x : w.x; w.y = fw ? hw[i].y : w.y; w.z = fw ? hw[i].z : w.z; w.w = fw ? hw[i].w : w.w;
          late = late || (hcur != hh[i] && (((hcur - hh[i]) & c.mask0) < 256u || ((hh[i] - hcur) & c.mask0) < 256u));
        }
        if (pipe_any(late)) {
          pipe_stores_done();
          if (late) w = pipe_settle(*(g_u128a4*)(L.arena + row));   // after every store so far, in this wavefront's order
        }
      }
      const int w0 = (int)w.x, w1 = (int)w.y, w2 = (int)w.z, w3 = (int)w.w;
      // (lanes without weights and the slots past a row's end have zero inputs: they add nothing)
      const int p0 = have[0] ? pq[sl][0] : 0, p1 = have[1] ? pq[sl][1] : 0, p2 = have[2] ? pq[sl][2] : 0, p3 = have[3] ? pq[sl][3] : 0;
      const int dot = __mul24(w0 >> 8, p0) + __mul24(w1 >> 8, p1) + __mul24(w2 >> 8, p2) + __mul24(w3 >> 8, p3);
      const int pr = sp_clamp2k(pipe_group_sum<QL>(dot) >> 8);
      const int y = (int)((bcur >> (7u - B)) & 1u);
      const int err = __mul24(y * 32767 - squash(pr), (int)c.a4) >> 4;
      uint4 nw;
      nw.x = (unsigned)sp_clamp512k(w0 + ((__mul24(err, p0) + (1 << 12)) >> 13));
      nw.y = (unsigned)sp_clamp512k(w1 + ((__mul24(err, p1) + (1 << 12)) >> 13));
      nw.z = (unsigned)sp_clamp512k(w2 + ((__mul24(err, p2) + (1 << 12)) >> 13));
      nw.w = (unsigned)sp_clamp512k(w3 + ((__mul24(err, p3) + (1 << 12)) >> 13));
      if (on) {
        if constexpr (FULL) *(g_u128a4*)(L.arena + row) = nw;
        else if (act && !tail) *(g_u128a4*)(L.arena + row) = nw;
        if constexpr (TAIL != 0 && !FULL) {
          if (tail) {
            L.A32(row) = nw.x;
            if constexpr (TAIL >= 2) L.A32(row + 4u) = nw.y;
            if constexpr (TAIL >= 3) L.A32(row + 8u) = nw.z;
          }
        }
        if (q == 0) L.put_p16(I, k, B, pr);
      }
#pragma unroll
      for (int i = HN - 1; i > 0; --i) { hr[i] = hr[i - 1]; hh[i] = hh[i - 1]; hw[i] = hw[i - 1]; }
      hr[0] = row; hh[0] = hcur; hw[0] = nw;
      // this slot now takes byte k + W; the slot D ahead gets its row, weights and inputs (its context came D bytes ago)
      {
        const unsigned kw = min(k + (unsigned)W, last);
        hx[sl] = L.ctx(ci, kw);
        bx[sl] = L.byte_at(kw);
        near((sl + D) % W, min(k + (unsigned)D, last));
      }
    }
  }
}

template <class Chain>
__device__ __forceinline__ void pipe_mix_bits_body(const PipeArgs& a) {
  __shared__ PipeSquash squash;
  const int lane = threadIdx.x & 63;
  const unsigned ngroups = (a.nblocks + Chain::PIPE_G - 1u) / Chain::PIPE_G;
  const unsigned wg = blockIdx.x + a.wg0;
  squash.load(a.tb, lane);
  __syncthreads();
  static_for<0, Chain::NMIXR>([&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    constexpr int I = Chain::MIX_COMP[r], QL = Chain::MIX_QL[r], first = Chain::MIX_FIRST[r];   // first: wavefronts per group of earlier roles
    constexpr int PPW = 64 / QL, BPW = PPW / 8;                  // (block, bit) pairs and blocks per wavefront
    static_assert(QL <= 8 && BPW >= 1 && (int)Chain::PIPE_G % BPW == 0, "MIX bit lanes: a block's 8 positions share a wavefront");
    constexpr int WPG = (int)Chain::PIPE_G / BPW;                // wavefronts per group
    if (wg < (unsigned)first * ngroups || wg >= (unsigned)(first + WPG) * ngroups) return;
    const unsigned wi = wg - (unsigned)first * ngroups;
    const unsigned g = wi / (unsigned)WPG, sub = wi % (unsigned)WPG;
    const unsigned pair = (unsigned)lane / QL, q = (unsigned)lane % QL, B = pair & 7u;
    PipeLane<Chain> L;
    L.open(a, pipe_opaque(g * Chain::PIPE_G + sub * BPW + (pair >> 3)), Chain::P_LEVEL[I]);
    if (L.chunk < 0 || !pipe_any(L.nb > 0)) return;
    pipe_mix_bits_unit<Chain, r>(L, q, B, squash);
  });
}

// one chunk of MIX role r with QL lanes per block and byte PART: q = the lane's weight quad.  NH = 1: a lane group codes all 8
// bits of its block's byte.  NH = 2 (the persistent launch's throughput shape): two lane groups per block, `half` 0 codes bits
// 0 .. 3 and half 1 bits 4 .. 7 -- with all bits known the positions of a byte are independent for a mixer whose row index
// contains the bit position (see pipe_mix_bits_unit), so the per-byte chain is half as long for twice the wavefronts; both
// halves of a block sit in ONE wavefront, so a row another half wrote is ordered by the wavefront's instruction order.
template <class Chain, int r, int NH, class SQ>
__device__ __forceinline__ void pipe_mix_unit(PipeLane<Chain>& L, unsigned q, unsigned half, const SQ& squash) {
  constexpr int I = Chain::MIX_COMP[r], QL = Chain::MIX_QL[r];
  constexpr CompK c = Chain::comp[I];
  constexpr int m = (int)c.a3, J = (int)c.a2, ci = Chain::P_CTX[I];
  constexpr int NQ = (m + 3) / 4, TAIL = m % 4, NB = 8 / NH;
  static_assert(NQ <= QL, "MIX lane group");
  constexpr bool batch = c.a5 == 255u && c.mask0 >= 255u;      // the 8 rows of a byte are distinct
  static_assert(NH == 1 || (NH == 2 && batch), "a byte is split over lane groups only when its 8 rows are distinct");
  if (!L.nb) return;
  // A row padded to at least 4 words per lane of the group (layout.h mix_row_stride: every m but 2) is loaded and stored
  // as whole quads by EVERY lane: a quad past the row's end lies in the padding, its inputs read as 0, its words go back as
  // they came -- no lane is masked, the 8 bits of a byte are straight-line code and the scheduler interleaves them.
  constexpr bool FULL = c.stride >= 4u * (unsigned)QL;
  const bool act = FULL || q < (unsigned)NQ;                       // lanes that hold weights
  const bool tail = !FULL && TAIL != 0 && q == (unsigned)(NQ - 1); // the lane whose quad is cut short by the row's end
  const unsigned qoff = 16u * (act ? q : 0u);
  bool have[4];
  int tin[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const int t = (int)q * 4 + x;
    have[x] = t < m;
    tin[x] = J + (have[x] ? t : 0);
  }
  // bit b of this lane's part of the byte: its c8 and its value
  auto c8_of = [&](unsigned bytev, int b) __attribute__((always_inline)) -> unsigned {
    if constexpr (NH == 1) return pipe_c8(bytev, b);
    else return ((half ? 16u : 1u) << b) | ((half ? bytev : bytev >> 4) >> (4 - b));
  };
  auto y_of = [&](unsigned bytev, int b) __attribute__((always_inline)) -> int {
    if constexpr (NH == 1) return pipe_y(bytev, b);
    else return (int)(((half ? bytev : bytev >> 4) >> (3 - b)) & 1u);
  };
  auto row_of = [&](unsigned hh, unsigned bytev, int b) __attribute__((always_inline)) -> unsigned {
    return (unsigned)c.t0 + 4u * __umul24((hh + (c8_of(bytev, b) & c.a5)) & c.mask0, (unsigned)c.stride) + qoff;   // s <= 24
  };
  unsigned h = L.ctx(ci, 0), byte = L.byte_at(0);
  const unsigned k1 = L.next(0);
  unsigned h1 = L.ctx(ci, k1), byte1 = L.byte_at(k1);
  // inputs of the weights a lane does not have read as 0 (masked once per byte, not once per bit); NH = 2: the two words
  // of the lane's half in .x / .y
  auto inputs = [&](int x, unsigned kk) __attribute__((always_inline)) -> uint4 {
    const uint4 v = L.p(tin[x], kk);
    const unsigned mk = have[x] ? 0xFFFFFFFFu : 0u;
    if constexpr (NH == 1) return make_uint4(v.x & mk, v.y & mk, v.z & mk, v.w & mk);
    else return make_uint4((half ? v.z : v.x) & mk, (half ? v.w : v.y) & mk, 0u, 0u);
  };
  uint4 pv[4], pv1[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) { pv[x] = inputs(x, 0); pv1[x] = inputs(x, k1); }
  uint4 w[NB];
  unsigned rowc[NB];
#pragma unroll
  for (int B = 0; B < NB; ++B) rowc[B] = row_of(h, byte, B);
  if constexpr (batch) {
#pragma unroll
    for (int B = 0; B < NB; ++B) w[B] = *(g_u128a4*)(L.arena + rowc[B]);
  }
  for (unsigned k = 0; k < L.nb; ++k) {
    const unsigned k2 = min(k + 2u, L.nb - 1u);
    const unsigned h2 = L.ctx(ci, k2), byte2 = L.byte_at(k2);
    uint4 pv2[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) pv2[x] = inputs(x, k2);
    unsigned rown[NB];
#pragma unroll
    for (int B = 0; B < NB; ++B) rown[B] = row_of(h1, byte1, B);
    // next byte's rows: same context -> only equal bit positions select the same row (c8 ranges are disjoint),
    // forwarded below; contexts less than 256 apart -> any position may coincide: fetched after the stores
    const bool same = h1 == h;
    const bool late = !same && (((h1 - h) & c.mask0) < 256u || ((h - h1) & c.mask0) < 256u);
    uint4 wn[NB], nw[NB];
    if constexpr (batch) {
      if (!late) {
#pragma unroll
        for (int B = 0; B < NB; ++B) wn[B] = *(g_u128a4*)(L.arena + rown[B]);
      }
    }
    PipeP8 out;
#pragma unroll
    for (int B = 0; B < NB; ++B) {
      const unsigned row = rowc[B];
      if constexpr (!batch) w[B] = *(g_u128a4*)(L.arena + row);
      const int w0 = (int)w[B].x, w1 = (int)w[B].y, w2 = (int)w[B].z, w3 = (int)w[B].w;
      const int p0 = pipe_p_get(pv[0], B), p1 = pipe_p_get(pv[1], B), p2 = pipe_p_get(pv[2], B), p3 = pipe_p_get(pv[3], B);
      // (lanes without weights and the slots past a row's end have zero inputs: they add nothing)
      const int dot = __mul24(w0 >> 8, p0) + __mul24(w1 >> 8, p1) + __mul24(w2 >> 8, p2) + __mul24(w3 >> 8, p3);
      const int pr = sp_clamp2k(pipe_group_sum<QL>(dot) >> 8);
      out.set(B, pr);
      const int err = __mul24(y_of(byte, B) * 32767 - squash(pr), (int)c.a4) >> 4;
      nw[B].x = (unsigned)sp_clamp512k(w0 + ((__mul24(err, p0) + (1 << 12)) >> 13));
      nw[B].y = (unsigned)sp_clamp512k(w1 + ((__mul24(err, p1) + (1 << 12)) >> 13));
      nw[B].z = (unsigned)sp_clamp512k(w2 + ((__mul24(err, p2) + (1 << 12)) >> 13));
      nw[B].w = (unsigned)sp_clamp512k(w3 + ((__mul24(err, p3) + (1 << 12)) >> 13));
      if constexpr (FULL) *(g_u128a4*)(L.arena + row) = nw[B];
      else if (act && !tail) *(g_u128a4*)(L.arena + row) = nw[B];
      if constexpr (TAIL != 0 && !FULL) {
        if (tail) {
          L.A32(row) = nw[B].x;
          if constexpr (TAIL >= 2) L.A32(row + 4u) = nw[B].y;
          if constexpr (TAIL >= 3) L.A32(row + 8u) = nw[B].z;
        }
      }
    }
    if (q == 0) {
      if constexpr (NH == 1) L.put_p(I, k, out.get());
      else L.put_p64(I, k, half, make_uint2(out.w[0], out.w[1]));
    }
    if constexpr (batch) {
      if constexpr (NH == 2) {
        if (pipe_any(late)) pipe_stores_done();       // (the other half's stores of this byte: ordered before the re-fetch)
      }
      if (late) {
#pragma unroll
        for (int B = 0; B < NB; ++B) wn[B] = *(g_u128a4*)(L.arena + rown[B]);
      }
#pragma unroll
      for (int B = 0; B < NB; ++B) {
        // (the tail lane's words past the row's end are never used: their inputs are 0 and they are not stored)
        const bool fw = same && rown[B] == rowc[B];
        w[B].x = fw ? nw[B].x : wn[B].x; w[B].y = fw ? nw[B].y : wn[B].y;
        w[B].z = fw ? nw[B].z : wn[B].z; w[B].w = fw ? nw[B].w : wn[B].w;
      }
    }
    h = h1; byte = byte1; h1 = h2; byte1 = byte2;
#pragma unroll
    for (int B = 0; B < NB; ++B) rowc[B] = rown[B];
#pragma unroll
    for (int x = 0; x < 4; ++x) { pv[x] = pv1[x]; pv1[x] = pv2[x]; }
  }
}

// ---- MIX with PACKED weight rows ------------------------------------------------------------------------------------
// Requests, not bytes, bound the encoder at 1024 blocks, and a MIX touches one row per bit.  Weights are
// clamped to +-2^19 (clamp512k, libzpaq.cpp:2028-2030; predict reads wt >> 8, 1913-1917), so four of them fit 3 dwords:
//   d0 = w0 | w1 << 24      d1 = w1 >> 8 | w2 << 16      d2 = w2 >> 16 | w3 << 8          (24 bits each, two's complement)
// and lane q's quad of a row sits at byte 12 q of it.  A row of up to 20 weights is then 60 bytes in a 64-byte stride where the
// padded 32-bit form takes a 128-byte line (MIX_PSTRIDE: the next power of two that holds 12 bytes per weight quad) -- half the
// bytes each way per touch -- and, packed, the rows of a SMALL table that a byte's first bits select fit the LDS:
// rows [0, LROWS) of the mixer live there for the whole sequence inside the persistent launch (m8 of compressBlock's level 5:
// the row is the partial byte c0, rows below 128 are bits 0 .. 6 of every byte -- 7 of the mixer's 8 row touches per byte
// never leave the compute unit), as [dword][row][quad][block of the wavefront] so that what a wavefront reads in one
// instruction is spread over the banks by the blocks' different rows.
// The arena holds the table in this form from the start: zpq_pipe_repack (below) rewrites Predictor::init's 65536 / m
// pattern once per block before the encoder runs; nothing but these units reads the table.
// Same structure as pipe_mix_unit (NH = 1, the byte's 8 rows distinct): a byte's HBM rows are fetched one byte ahead and
// patched by the forwarding rules there; LDS rows are read behind the previous byte's stores, in program order.
struct PipeW3 { unsigned x, y, z; };
typedef __attribute__((address_space(1))) PipeW3 g_w3;

__device__ __forceinline__ int pipe_sext24(unsigned v) { return (int)(v << 8) >> 8; }
__device__ __forceinline__ void pipe_unpack4(const PipeW3& d, int& w0, int& w1, int& w2, int& w3) {
  w0 = pipe_sext24(d.x);
  w1 = pipe_sext24((d.x >> 24) | (d.y << 8));
  w2 = pipe_sext24((d.y >> 16) | (d.z << 16));
  w3 = (int)d.z >> 8;
}
__device__ __forceinline__ PipeW3 pipe_pack4(int w0, int w1, int w2, int w3) {
  PipeW3 d;
  d.x = ((unsigned)w0 & 0xFFFFFFu) | ((unsigned)w1 << 24);
  d.y = (((unsigned)w1 >> 8) & 0xFFFFu) | ((unsigned)w2 << 16);
  d.z = (((unsigned)w2 >> 16) & 0xFFu) | ((unsigned)w3 << 8);
  return d;
}

// bytes from one packed row to the next: 12 per weight quad, rounded up to a power of two (at most the padded 32-bit row)
__device__ __forceinline__ constexpr unsigned pipe_mix_pstride(unsigned m) {
  const unsigned need = 12u * ((m + 3u) / 4u);
  unsigned p = 16u;
  while (p < need) p *= 2u;
  return p;
}

template <class Chain, class = void> struct PipeMixLdsRows { static constexpr int of(int) { return 0; } };
template <class Chain> struct PipeMixLdsRows<Chain, decltype((void)Chain::MIX_LDS_ROWS)> { static constexpr int of(int r) { return Chain::MIX_LDS_ROWS[r]; } };

// one chunk of MIX role r on packed rows.  q = the lane's weight quad, bl = the lane's block among the BPW of its wavefront;
// lds = the unit's private region (LROWS > 0 only), stage = first chunk: copy rows [0, LROWS) from the arena.  NH = 2: two lane
// groups per block, `half` 0 codes bits 0 .. 3 and half 1 bits 4 .. 7 (pipe_mix_unit: both halves of a block in ONE wavefront).
template <class Chain, int r, int LROWS, int BPW, int NH, class SQ>
__device__ __forceinline__ void pipe_mix_packed_unit(PipeLane<Chain>& L, unsigned q, unsigned half, unsigned bl, unsigned* lds, bool stage, const SQ& squash) {
  constexpr int I = Chain::MIX_COMP[r], QL = Chain::MIX_QL[r];
  constexpr CompK c = Chain::comp[I];
  constexpr int m = (int)c.a3, J = (int)c.a2, ci = Chain::P_CTX[I];
  constexpr int NQ = (m + 3) / 4, NB = 8 / NH;
  constexpr unsigned PS = pipe_mix_pstride((unsigned)m);
  static_assert(NQ <= QL && c.a5 == 255u && c.mask0 >= 255u, "packed MIX rows: the 8 rows of a byte are distinct");
  static_assert(PS <= 4u * c.stride, "a packed row fits the padded row's place");
  static_assert(NH == 1 || NH == 2, "lane groups per block");
  if (!L.nb) return;
  const bool act = q < (unsigned)NQ;                   // lanes that hold weights (the others: zero inputs, no stores)
  const unsigned qoff = 12u * (act ? q : 0u);
  bool have[4];
  int tin[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const int t = (int)q * 4 + x;
    have[x] = t < m;
    tin[x] = J + (have[x] ? t : 0);
  }
  // bit b of this lane's part of the byte: its c8 and its value
  auto c8_of = [&](unsigned bytev, int b) __attribute__((always_inline)) -> unsigned {
    if constexpr (NH == 1) return pipe_c8(bytev, b);
    else return ((half ? 16u : 1u) << b) | ((half ? bytev : bytev >> 4) >> (4 - b));
  };
  auto y_of = [&](unsigned bytev, int b) __attribute__((always_inline)) -> int {
    if constexpr (NH == 1) return pipe_y(bytev, b);
    else return (int)(((half ? bytev : bytev >> 4) >> (3 - b)) & 1u);
  };
  // row index of bit b of a byte, and where its quad lives
  auto row_of = [&](unsigned hh, unsigned bytev, int b) __attribute__((always_inline)) -> unsigned { return (hh + c8_of(bytev, b)) & c.mask0; };
  auto addr_of = [&](unsigned row) __attribute__((always_inline)) -> unsigned { return (unsigned)c.t0 + row * PS + qoff; };
  // LDS: [dword][row][quad][block]
  constexpr unsigned LW = (unsigned)(LROWS > 0 ? LROWS : 1) * (unsigned)NQ * (unsigned)BPW;      // words from one dword plane to the next
  const unsigned lbase = (act ? q : 0u) * (unsigned)BPW + bl;
  auto lds_at = [&](unsigned row) __attribute__((always_inline)) -> unsigned { return row * (unsigned)(NQ * BPW) + lbase; };
  if constexpr (LROWS > 0) {
    if (stage && act && (NH == 1 || half == 0u))
      for (unsigned row = 0; row < (unsigned)LROWS && row <= c.mask0; ++row) {
        const PipeW3 d = *(g_w3*)(L.arena + addr_of(row));
        const unsigned a = lds_at(row);
        lds[a] = d.x; lds[a + LW] = d.y; lds[a + 2u * LW] = d.z;
      }
  }
  unsigned h = L.ctx(ci, 0), byte = L.byte_at(0);
  const unsigned k1 = L.next(0);
  unsigned h1 = L.ctx(ci, k1), byte1 = L.byte_at(k1);
  auto inputs = [&](int x, unsigned kk) __attribute__((always_inline)) -> uint4 {
    const uint4 v = L.p(tin[x], kk);
    const unsigned mk = have[x] ? 0xFFFFFFFFu : 0u;
    if constexpr (NH == 1) return make_uint4(v.x & mk, v.y & mk, v.z & mk, v.w & mk);
    else return make_uint4((half ? v.z : v.x) & mk, (half ? v.w : v.y) & mk, 0u, 0u);
  };
  uint4 pv[4], pv1[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) { pv[x] = inputs(x, 0); pv1[x] = inputs(x, k1); }
  // a row that lives in the LDS is not fetched from the arena: its request goes to the table's first line instead (a line
  // this wavefront keeps in the L1: no branch around a request -- see pipe_match_in -- and no transaction at the L2)
  auto in_lds = [&](unsigned row) __attribute__((always_inline)) -> bool { return LROWS > 0 && row < (unsigned)LROWS; };
  auto fetch = [&](unsigned row) __attribute__((always_inline)) -> PipeW3 { return *(g_w3*)(L.arena + addr_of(in_lds(row) ? 0u : row)); };
  auto from_lds = [&](PipeW3& d, unsigned row) __attribute__((always_inline)) {
    const unsigned a = lds_at(in_lds(row) ? row : 0u);
    const unsigned x = lds[a], y = lds[a + LW], z = lds[a + 2u * LW];
    const bool li = in_lds(row);
    d.x = li ? x : d.x; d.y = li ? y : d.y; d.z = li ? z : d.z;
  };
  PipeW3 w[NB];
  unsigned rowc[NB];
#pragma unroll
  for (int B = 0; B < NB; ++B) { rowc[B] = row_of(h, byte, B); w[B] = fetch(rowc[B]); }
  if constexpr (LROWS > 0) {
    if constexpr (NH == 2) (void)pipe_any(stage);      // (the other half staged the rows: a wavefront's LDS operations are in order; the emulator's lanes meet here)
#pragma unroll
    for (int B = 0; B < NB; ++B) from_lds(w[B], rowc[B]);
  }
  for (unsigned k = 0; k < L.nb; ++k) {
    const unsigned k2 = min(k + 2u, L.nb - 1u);
    const unsigned h2 = L.ctx(ci, k2), byte2 = L.byte_at(k2);
    uint4 pv2[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) pv2[x] = inputs(x, k2);
    unsigned rown[NB];
#pragma unroll
    for (int B = 0; B < NB; ++B) rown[B] = row_of(h1, byte1, B);
    // next byte's rows: same context -> only equal bit positions select the same row (c8 ranges are disjoint), forwarded
    // below; contexts less than 256 apart -> any position may coincide: fetched after the stores
    const bool same = h1 == h;
    const bool late = !same && (((h1 - h) & c.mask0) < 256u || ((h - h1) & c.mask0) < 256u);
    PipeW3 wn[NB], nw[NB];
    if (!late) {
#pragma unroll
      for (int B = 0; B < NB; ++B) wn[B] = fetch(rown[B]);
    }
    PipeP8 out;
#pragma unroll
    for (int B = 0; B < NB; ++B) {
      int w0, w1, w2, w3;
      pipe_unpack4(w[B], w0, w1, w2, w3);
      const int p0 = pipe_p_get(pv[0], B), p1 = pipe_p_get(pv[1], B), p2 = pipe_p_get(pv[2], B), p3 = pipe_p_get(pv[3], B);
      const int dot = __mul24(w0 >> 8, p0) + __mul24(w1 >> 8, p1) + __mul24(w2 >> 8, p2) + __mul24(w3 >> 8, p3);
      const int pr = sp_clamp2k(pipe_group_sum<QL>(dot) >> 8);
      out.set(B, pr);
      const int err = __mul24(y_of(byte, B) * 32767 - squash(pr), (int)c.a4) >> 4;
      nw[B] = pipe_pack4(sp_clamp512k(w0 + ((__mul24(err, p0) + (1 << 12)) >> 13)), sp_clamp512k(w1 + ((__mul24(err, p1) + (1 << 12)) >> 13)),
                         sp_clamp512k(w2 + ((__mul24(err, p2) + (1 << 12)) >> 13)), sp_clamp512k(w3 + ((__mul24(err, p3) + (1 << 12)) >> 13)));
      if (act) {
        if (in_lds(rowc[B])) { const unsigned a = lds_at(rowc[B]); lds[a] = nw[B].x; lds[a + LW] = nw[B].y; lds[a + 2u * LW] = nw[B].z; }
        else *(g_w3*)(L.arena + addr_of(rowc[B])) = nw[B];
      }
    }
    if (q == 0) {
      if constexpr (NH == 1) L.put_p(I, k, out.get());
      else L.put_p64(I, k, half, make_uint2(out.w[0], out.w[1]));
    }
    if constexpr (NH == 2) {
      if (pipe_any(late)) pipe_stores_done();       // (the other half's stores of this byte: ordered before the re-fetch)
    }
    if (late) {
#pragma unroll
      for (int B = 0; B < NB; ++B) wn[B] = fetch(rown[B]);
    }
#pragma unroll
    for (int B = 0; B < NB; ++B) {
      const bool fw = same && rown[B] == rowc[B];
      w[B].x = fw ? nw[B].x : wn[B].x; w[B].y = fw ? nw[B].y : wn[B].y; w[B].z = fw ? nw[B].z : wn[B].z;
    }
    if constexpr (LROWS > 0) {      // (behind this byte's LDS stores, in program order: no forwarding needed)
#pragma unroll
      for (int B = 0; B < NB; ++B) from_lds(w[B], rown[B]);
    }
    h = h1; byte = byte1; h1 = h2; byte1 = byte2;
#pragma unroll
    for (int B = 0; B < NB; ++B) rowc[B] = rown[B];
#pragma unroll
    for (int x = 0; x < 4; ++x) { pv[x] = pv1[x]; pv1[x] = pv2[x]; }
  }
}

// which MIX roles of a chain keep packed rows (host/codegen.cpp decides: MIX_PACKED in the generated chain)
template <class Chain, class = void> struct PipeMixPacked { static constexpr bool of(int) { return false; } };
template <class Chain> struct PipeMixPacked<Chain, decltype((void)Chain::MIX_PACKED)> { static constexpr bool of(int r) { return Chain::MIX_PACKED[r] != 0; } };

// Predictor::init wrote 65536 / m into every dword of a MIX table (libzpaq.cpp:1822-1826); the packed units want the same
// weights as 24-bit quads.  One workgroup per block, before the encoder's first launch (engine.cpp; tests/emu does the same).
template <class Chain>
__device__ __forceinline__ void pipe_repack_body(const PipeArgs& a) {
  const unsigned blk = blockIdx.x;
  if (blk >= a.nblocks) return;
  g_u8* const arena = (g_u8*)a.jobs[blk].arena;
  static_for<0, Chain::NMIXR>([&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    if constexpr (PipeMixPacked<Chain>::of(r)) {
      constexpr CompK c = Chain::comp[Chain::MIX_COMP[r]];
      constexpr unsigned m = c.a3, PS = pipe_mix_pstride(m), NQ = (m + 3u) / 4u;
      constexpr int v = (int)(65536u / m);
      const PipeW3 d = pipe_pack4(v, v, v, v);
      const unsigned per_row = PS / 4u;                                        // dwords
      const unsigned long long words = (unsigned long long)(c.mask0 + 1u) * per_row;
      for (unsigned long long i = threadIdx.x; i < words; i += blockDim.x) {
        const unsigned j = (unsigned)(i % per_row);
        const unsigned val = j < 3u * NQ ? (j % 3u == 0u ? d.x : (j % 3u == 1u ? d.y : d.z)) : 0u;
        *(g_u32*)(arena + (unsigned long long)c.t0 + 4ull * i) = val;
      }
    }
  });
}

template <class Chain>
__device__ __forceinline__ void pipe_mix_body(const PipeArgs& a) {
  if constexpr (Chain::MIX_BITS != 0) {
    pipe_mix_bits_body<Chain>(a);
    return;
  } else {
  __shared__ PipeSquash squash;
  const int lane = threadIdx.x & 63;
  const unsigned ngroups = (a.nblocks + Chain::PIPE_G - 1u) / Chain::PIPE_G;
  const unsigned wg = blockIdx.x + a.wg0;
  squash.load(a.tb, lane);
  __syncthreads();
  static_for<0, Chain::NMIXR>([&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    constexpr int I = Chain::MIX_COMP[r], QL = Chain::MIX_QL[r], first = Chain::MIX_FIRST[r];   // first = sum of QL of earlier roles
    const unsigned per_group = (unsigned)QL;                                                  // wavefronts per group
    if (wg < (unsigned)first * ngroups || wg >= (unsigned)(first + QL) * ngroups) return;
    constexpr int BPW = (int)Chain::PIPE_G / QL;
    static_assert(BPW >= 1, "MIX lane group");
    const unsigned wi = wg - (unsigned)first * ngroups;
    const unsigned g = wi / per_group, sub = wi % per_group;
    const unsigned bl = (unsigned)lane / QL, q = (unsigned)lane % QL;
    PipeLane<Chain> L;
    L.open(a, g * Chain::PIPE_G + sub * BPW + bl, Chain::P_LEVEL[I]);
    if (bl >= (unsigned)BPW) { L.live = false; L.nb = 0; }          // lanes beyond this wavefront's blocks
    if (L.chunk < 0 || !pipe_any(L.nb > 0)) return;
    if constexpr (PipeMixPacked<Chain>::of(r)) pipe_mix_packed_unit<Chain, r, 0, 1, 1>(L, q, 0u, 0u, nullptr, false, squash);
    else pipe_mix_unit<Chain, r, 1>(L, q, 0u, squash);
  });
  }
}

}  // namespace zpq
